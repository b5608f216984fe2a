#!/usr/bin/env python3
"""bench.py -- throughput of the frontend hot path (detect + describe + match-to-previous) on
N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p_2k|c2|c3|c4] [--batch B]

A "step" is one pass of the hot path over one batch of B synthetic frames that are already
resident in HBM: FAST detect -> exact std::sort order -> greedy NMS -> blur -> orientation +
steered BRIEF -> brute-force Hamming match of every frame against its predecessor (frame 0 of
the batch against the batch's last frame, standing in for the previous batch's last frame, so
every one of the B frames is detected, described AND matched).

Workloads (BASELINE.json): the metric is quoted on "1080p @ 2k feats/frame", which is the
default; `c2` = configs[1] (640x480, ~2000 keypoints), `c3` = configs[2] (1080p, ~10k keypoints).
The reference has no feature-count knob, so counts are reached through the synthetic corner
density and SuppressionWindowSize; the realised candidates M / keypoints N are in the JSON.
Frames per step (512 / 1024 / 256): the sort and the NMS run one workgroup per frame, so a batch
must be a multiple of the 256 CUs -- and of 2 x 256 where two of them fit a CU -- to keep the chip full.

`c4` = configs[3]: the 500-frame sequence's sliding-window Gauss-Newton (one window of 10 poses x 2048 landmarks
per frame, J^T J / J^T r and the landmark elimination on the FP64 matrix cores); a step there is one
Gauss-Newton iteration (linearise + accumulate, solve, back-substitute, apply) over all windows of the
rank; the cost after the last step is evaluated once at the end of the job, inside the timed region.

Multi-GPU: frames shard embarrassingly (one independent stream per rank, seed + rank), no
data-path collective; at the end of the timed job every rank triangulates its newest frame pair
and the map points are exchanged with one RCCL all_gather (counts + padded points).  One process per
GPU: started by `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when
`--gpus N` is given with no launcher around it -- by this script itself (`launch_ranks`: N fresh children,
started before torch or HIP are touched in the parent).  `--backend gloo --share-device` rehearses the N > 1
path on one card, `--dry-run` the rank plumbing with no GPU at all.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
dominant kernel and `cpu_baseline` (the CPU oracle timed on this host, rank 0, N == 1 only).  The default
run on one GPU also carries `other_workloads`: short runs of c3 / c3p / c4 (child processes, after the timed
region), so that BASELINE.json's other single-GPU configs are measured by whoever runs the default command.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

WORKLOADS = {
    # name: rows, cols, window, n_rect, rect_min, rect_max, batch, max_candidates, max_keypoints
    "1080p_2k": dict(rows=1080, cols=1920, window=12, n_rect=6800, rmin=3, rmax=24, batch=512, max_cand=32768, max_kp=4096,
                     desc="1920x1080, ~2k keypoints/frame (shipped YAML: t=20 N=12 window=12), detect+describe+match-to-previous"),
    "c2": dict(rows=480, cols=640, window=6, n_rect=5000, rmin=3, rmax=16, batch=1024, max_cand=32768, max_kp=4096,
               desc="640x480, ~2k keypoints/frame (SuppressionWindowSize 6), detect+describe+match-to-previous"),
    # configs[0], the reference's own CPU-runnable case: 50 synthetic 640x480 frames, ~500 keypoints per frame with the
    # SHIPPED YAML (test/data/feature_detector.yml:3-7: t=20 N=12 window=12); one step = the 50-frame job
    "c1": dict(rows=480, cols=640, window=12, n_rect=1950, rmin=3, rmax=24, batch=50, max_cand=16384, max_kp=2048,
               desc="50 x 640x480, ~500 keypoints/frame (shipped YAML: t=20 N=12 window=12), detect+describe+match-to-previous"),
    "c3": dict(rows=1080, cols=1920, window=8, n_rect=32000, rmin=3, rmax=16, batch=256, max_cand=131072, max_kp=16384,
               desc="1920x1080, ~10k keypoints/frame (SuppressionWindowSize 8), detect+describe+match-to-previous"),
    # configs[2] as BASELINE.json words it: "1920x1080 8-level pyramid, 10k feats/frame" -- the pyramid is a build-side
    # extension (the reference has none), levels >= 1 are parity-unpinned; reported separately from c3
    # (512 frames per step since round 4: two NMS workgroups share a CU and the sorts run two full rounds -- 21.1k ->
    # 22.8k frames/s against 256 frames per step, the packing a single launch over all levels would give)
    "c3p": dict(rows=1080, cols=1920, window=12, n_rect=6800, rmin=3, rmax=24, batch=512, max_cand=32768, max_kp=16384,
                levels=8, scale=1.2,
                desc="1920x1080, 8-level pyramid (scale 1.2), ~12k keypoints/frame over all levels (the default workload's frames, "
                     "shipped YAML), detect+describe on every level + match-to-previous on the merged sets"),
}
GN_WORKLOAD = dict(windows=500, cams=10, landmarks=2048, lam=1e-4, n_fixed=2,
                   desc="500-frame sequence, one sliding window per frame: 10 poses x 2048 landmarks, one Gauss-Newton "
                        "iteration per step (J^T J / J^T r + landmark elimination on FP64 MFMA, reduced solve, back-substitution)")
# FP64 matrix peak of MI355X: 78.6 TFLOP/s (AMD spec; v_mfma_f64_16x16x4_f64 = 2048 flop in 64 cycles per
# SIMD at 2.4 GHz on 1024 SIMDs = 78.6; tools/microbench/mfma_f64_peak.hip measures the sustained rate)
MFMA_F64_PEAK_TFLOPS = 78.6
SEED = 0x5EED0051
# The scene of `--with-pose` (tests/pose_scene.py:loop_sequence).  The 6-point DLT of LoopClosure::solvePnP has no refit
# and a minimal sample amplifies keypoint noise: at 100 hypotheses (loop_closure.yml) and a 2 px threshold the calibrated
# form finds the correct correspondences reliably up to ~0.02 px of keypoint noise (0.15 px: a few per cent of them) --
# measured with a numpy restatement before choosing these; the essential-matrix half does not care.
# A PnP correspondence is right when two consecutive matches are: 0.85^2 = 72 % of them at 15 % wrong matches, and a
# 6-point sample is all-right with probability 14 %.
POSE_SCENE = dict(outliers=0.15, noise_px=0.02, keep=0.85)
METRIC = "frames/s + matched-keypoints/s, 1080p @ 2k feats/frame, 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
# Sustained rate of the Hamming inner loop's instruction mix (v_xor_b32 + v_bcnt_u32_b32 with fused
# accumulate) measured on MI355X with tools/microbench/valu_peak.hip: 42.1 Tera lane-ops/s
# (v_bcnt is half rate; full-rate int32 pairs reach 65.8).  SURVEY.md 8d's nominal 3.9e13 agrees.
VALU_HAMMING_PEAK_GOPS = 42100.0
# int8 matrix-core peak (dense): ~5 POP/s (MI355X_MICROARCH.md; v_mfma_i32_16x16x64_i8 = 32768 ops in 16 cycles per SIMD)
MFMA_I8_PEAK_TOPS = 5000.0


def load_product():
    name = "slam_cin0051_amd"
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "slam-cin0051_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def shard_seed(rank):
    """One independent synthetic stream per rank."""
    return SEED + rank


def gather_summary(dist, values, device):
    """The one collective of the path: every rank contributes its result summary
    [keypoints, matches, candidates] and receives everyone's (RCCL all_gather on the GPU box,
    gloo in the CPU tests).  Returns the per-rank rows as a list of lists."""
    import torch

    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return [[float(x) for x in g.tolist()] for g in gathered]


def gather_map_points(dist, xyz, device, cap):
    """The one exchange of the path: RCCL all_gather of every rank's triangulated map points (N_g x 3
    doubles, variable N_g) as ONE collective -- a fixed-capacity payload of cap + 1 rows per rank whose row 0 carries the
    count (`cap` is a capacity every rank knows without asking: the batch's max_keypoints), one copy of the gathered block
    to the host, no host synchronisation between collectives (until round 4: counts first, N .item() calls, then the
    points padded to the largest count).  Returns the list of per-rank arrays."""
    import numpy as np
    import torch

    world = dist.get_world_size()
    n = int(xyz.shape[0])
    if n > cap:
        raise RuntimeError("gather_map_points: %d points exceed the agreed capacity of %d rows" % (n, cap))
    host = np.zeros((cap + 1, 3), np.float64)
    host[0, 0] = float(n)
    if n > 0:
        host[1:n + 1] = xyz
    mine = torch.from_numpy(host).to(device)
    out = torch.empty((world * (cap + 1), 3), dtype=torch.float64, device=device)  # (concatenated: the form gloo takes too)
    dist.all_gather_into_tensor(out, mine)
    got = out.cpu().numpy().reshape(world, cap + 1, 3)  # the one synchronisation: the gathered block comes to the host
    return [got[r, 1:int(got[r, 0, 0]) + 1].copy() for r in range(world)]


def triangulate_last_pair(ctx, batch, matcher, B, cols, rows):
    """Map points of the newest frame pair (PoseEstimator::triangulatePoints with a fixed synthetic
    relative pose): the payload of the final gather."""
    import numpy as np

    kq, _ = batch.download(B - 1)
    kt, _ = batch.download(B - 2)
    m = batch.download_matches(matcher, B - 1)
    f = 0.9 * cols
    K = np.array([[f, 0, cols / 2.0], [0, f, rows / 2.0], [0, 0, 1.0]])
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = K @ np.hstack([np.eye(3), np.array([[-0.2], [0.0], [0.0]])])
    p1 = np.stack([kt["x"][m["trainIdx"]], kt["y"][m["trainIdx"]]], 1).astype(np.float32)
    p2 = np.stack([kq["x"][m["queryIdx"]], kq["y"][m["queryIdx"]]], 1).astype(np.float32)
    if len(m) == 0:
        return np.zeros((0, 3))
    return ctx.triangulate(P1, P2, p1, p2)[1]


def pmc_traffic_bytes(kernel, batch):
    """2 x FETCH_SIZE + WRITE_SIZE of one launch of `kernel`: measured offline at 256 frames per launch
    (tools/collect_profiles.sh) and scaled to `batch` frames -- the per-frame traffic of these kernels
    does not depend on the batch.  Returns (bytes or None, the profiles/ file the figure came from or None)."""
    # a STORED figure (profiles/README.md): the newest round's counter file that names the kernel, not a measurement of this run
    scale = batch / 256.0
    name = kernel.split("<")[0]
    for r in (5, 4, 3, 2):
        path = os.path.join(ROOT, "profiles", "r0%d_pmc_hbm_traffic_b256.csv" % r)
        if not os.path.exists(path):
            continue
        for line in open(path).read().splitlines()[1:]:
            parts = line.rsplit(",", 2)
            if len(parts) == 3 and parts[0].split("<")[0] == name:
                # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section):
                # doubled here as that guide prescribes; WRITE_SIZE is exact
                return (2.0 * float(parts[1]) + float(parts[2])) * 1024.0 * scale, os.path.relpath(path, ROOT)
    return None, None


def pcie_inclusive(mod, wl, B, frames, main, local_rank, use_keypoints, steps=12, n_lanes=4):
    """SURVEY.md 8d's END-TO-END definition of frames/s: the same step with every frame coming from the host AND every
    result going back to it.  The batch is copied from page-locked host memory (slamx_batch_upload_async) before it
    is processed; after it, a device-side pack step gathers the VALID keypoints, descriptors and matches of all its
    frames and stores them into page-locked memory (slamx_batch_download_packed_async: the pack kernel's own stores
    over PCIe, or with SLAMX_PACKED_DOWNLOAD=copy one queued copy; the host range is sized from the previous batch of
    the lane plus 10 %, the tail call covers a wrong guess).  n_lanes contexts (streams, batches) rotate, so the copies
    of one batch overlap the kernels and copies of the others (measured: 22.7k / 24.4k / 24.8k frames/s with 2 / 3 / 4
    of them).  tools/e2e_breakdown.py cuts the step short stage by stage: uploads alone 57.6 GB/s (27.8k frames/s,
    the link), + detect 53.8, + match 52.9, + the 60 MB of results 50.2 GB/s -- the same with either download form.
    Reported next to `value`, never as it (the timed region of `value` starts with the frames resident in HBM, as
    the task's measurement rule says)."""
    pin = mod.pinned_array(frames.shape)
    pin[...] = frames
    lanes = [main]  # (context, batch, matcher, pyramid or None)
    for _ in range(n_lanes - 1):
        c = mod.Context(local_rank)
        d = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, wl["window"], 31, 256), c)
        m = mod.FeatureMatcher(mod.MatcherConfig(0, 0, 20, 1, 0.5), c)
        if "levels" in wl:
            py = mod.Pyramid(d, wl["rows"], wl["cols"], B, wl["levels"], wl["scale"], wl["max_cand"], wl["max_kp"])
            lanes.append((c, py.level(0), m, py))
        else:
            lanes.append((c, mod.Batch(d, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"]), m, None))
    outs = [lane[1].packed_buffer(B) for lane in lanes]
    guess = [outs[0].nbytes] * len(lanes)  # first round: the worst case
    sent = [0] * len(lanes)
    tails = 0

    def step(i):
        k = i % len(lanes)
        _, b, m, py = lanes[k]
        b.upload_async(pin)
        if py is not None:
            py.detect_and_compute(B)
        else:
            b.detect_and_compute(B)
        b.match_consecutive(m, B, use_keypoints=use_keypoints, first_train_frame=B - 1)
        sent[k] = b.download_packed_async(B, outs[k], guess[k])

    def collect(k):
        """what a consumer does with a finished batch of lane k: check the header, fetch a tail if the guess was low"""
        nonlocal tails
        lanes[k][0].synchronize()
        total = int(mod.packed_header(outs[k])["total_bytes"])
        if total > sent[k]:
            lanes[k][1].fetch_packed_tail(outs[k], sent[k])
            tails += 1
        guess[k] = min(outs[k].nbytes, int(total * 1.1) + 4096)
        return total

    for i in range(len(lanes)):
        step(i)
    for k in range(len(lanes)):
        collect(k)
    t0 = time.perf_counter()
    total_out = 0
    for i in range(steps):
        if i >= len(lanes):
            total_out += collect(i % len(lanes))  # the lane's previous batch is consumed before its buffers are reused
        step(i)
    for k in range(len(lanes)):
        if k < steps:
            total_out += collect(k)
    dt = time.perf_counter() - t0
    gb = steps * frames.nbytes / 1e9
    kp_first, m_first, status, kps, desc, matches = mod.unpack_results(outs[0])
    assert kp_first[-1] > 0 and m_first[-1] > 0 and not status.any()  # the results did arrive
    # the same lanes with the frames left where they are: what two batches in flight (two streams) add to the
    # HBM-resident rate of `value`, which queues its steps on one stream
    t1 = time.perf_counter()
    for i in range(2 * steps):
        _, b, m, py = lanes[i % 2]
        if py is not None:
            py.detect_and_compute(B)
        else:
            b.detect_and_compute(B)
        b.match_consecutive(m, B, use_keypoints=use_keypoints, first_train_frame=B - 1)
    for lane in lanes[:2]:
        lane[0].synchronize()
    two_in_flight = 2 * steps * B / (time.perf_counter() - t1)
    return {"value": steps * B / dt, "unit": "frames/s", "steps": steps, "contexts": len(lanes),
            "hbm_resident_two_batches_in_flight": two_in_flight,
            "host_to_hbm_GBps": gb / dt, "hbm_to_host_GBps": total_out / 1e9 / dt,
            "result_bytes_per_frame": total_out / (steps * B), "tail_fetches": tails,
            "note": "every batch uploaded from pinned host memory before it is processed; its valid results (counts, "
                    "keypoints, descriptors, matches) packed on the device and downloaded with one copy after; %d "
                    "contexts rotate" % len(lanes)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set
    (a GPU box hands a one-GPU job a share of the host: 256 CPUs visible, ~16 granted)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(mod, wl, frames, budget_s=12.0, budget_mt_s=8.0):
    """The restated reference CPU path (oracle/) on a bounded sample of the SAME frames:
    detectAndCompute + match(frame k, frame k-1) with keypoints.  Two numbers: ONE thread -- the reference's own
    execution model (it has no threading on this path), the primary baseline -- and ALL host cores this process
    may use, frames sharded over std::threads (oracle/cpu_sequence.cpp).  Pyramid workloads: one thread only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding

    o = oracle_binding.load()
    P = oracle_binding.DetectorParams(20, 12, 1, wl["window"], 31, 256)
    M = oracle_binding.MatcherParams(0, 20, 1, 0.5)
    cores = usable_cores()
    host = {"host_cpus": os.cpu_count(), "usable_cpus": cores, "cpu_model": cpu_model()}
    if "levels" in wl:
        def dac(img):
            return o.pyramid_detect_and_compute(img, P, wl["levels"], wl["scale"])

        t0 = time.perf_counter()
        prev = dac(frames[-1])  # predecessor of frame 0, not counted
        o.match(prev[1], prev[1], prev[0], prev[0], M)  # (sizes the sample: matching is O(N^2) and dominates at 10k keypoints)
        t_first = time.perf_counter() - t0
        n = int(max(2, min(len(frames), budget_s / max(t_first, 1e-3))))
        t0 = time.perf_counter()
        kps_total = 0
        for k in range(n):
            cur = dac(frames[k])
            o.match(cur[1], prev[1], cur[0], prev[0], M)
            kps_total += len(cur[0])
            prev = cur
        dt = time.perf_counter() - t0
        out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same workload, CPU restatement of the reference path over the pyramid "
                         "(oracle/, g++ -O3 -ffp-contract=off, 1 thread), %.1f s, %.0f keypoints/frame" % (n, dt, kps_total / n),
               "matched_keypoints_per_s": kps_total / dt}
        out.update(host)
        return out
    # size the samples from two frames
    dt2, _ = o.sequence_mt(frames[:2], P, M, True, 1)
    t_frame = max(dt2 / 3.0, 1e-4)  # 2 frames + the halo's detectAndCompute
    n1 = int(max(2, min(len(frames), budget_s / t_frame)))
    reps1 = int(max(1, budget_s / (t_frame * n1)))  # short jobs (c1: 50 frames) go round several times
    dt1, per = o.sequence_mt(frames[:n1], P, M, True, 1, reps1)
    kps1 = float(per[:, 0].sum()) * reps1
    n1 *= reps1
    out = {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same workload, CPU restatement of the reference path (oracle/slam_oracle.cpp, "
                     "g++ -O3 -ffp-contract=off, 1 thread = the reference's execution model), %.1f s, %.0f keypoints/frame"
                     % (n1, dt1, kps1 / n1),
           "matched_keypoints_per_s": kps1 / dt1}
    out.update(host)
    if cores > 1:
        T = min(cores, len(frames), 128)
        per_thread = max(1, min(len(frames) // T, int(budget_mt_s / t_frame)))  # (c3: 3.5 s per frame -- two per thread)
        nmt = per_thread * T
        # one frame per thread first: what a frame costs with every core busy (clocks, shared caches, a quota the
        # cgroup files did not show), then as many passes as the budget allows
        dt_cal, _ = o.sequence_mt(frames[:T], P, M, True, T, 1)
        t_busy = max(dt_cal / 2.0, 1e-4)  # a thread did its halo + one frame
        reps = int(max(1, min(10000, budget_mt_s / (t_busy * per_thread))))
        dtm, perm = o.sequence_mt(frames[:nmt], P, M, True, T, reps)
        out["all_cores"] = {"value": nmt * reps / dtm, "unit": "frames/s", "cores": T, "kind": "port",
                            "sample": "%d frames x %d passes sharded over %d std::threads (oracle/cpu_sequence.cpp; the "
                                      "reference itself is single-threaded), %.1f s" % (nmt, reps, T, dtm),
                            "matched_keypoints_per_s": float(perm[:, 0].sum()) * reps / dtm,
                            "speedup_over_1_thread": (nmt * reps / dtm) / (n1 / dt1)}
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the launcher.  It starts N
    fresh children of this script -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment exactly as `torch.distributed.run` would set them -- BEFORE torch is imported or anything touches
    HIP here (the parent never initialises the GPU and never execs), relays rank 0's JSON line, and fails if any
    rank fails.  A rank that dies takes the others down with it (they would wait in the rendezvous for ever):
    the parent ends exactly the processes it started."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    rc = 0
    try:
        alive = set(range(n))
        import selectors

        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        open0 = True
        while alive:
            if open0:
                for key, _ in sel.select(timeout=0.2):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        sel.unregister(key.fileobj)
                        open0 = False
            else:
                time.sleep(0.2)
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                    for q in alive:
                        procs[q].terminate()
        if open0:
            out0 += procs[0].stdout.read()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def main_dry_run(args, rank, world, dist):
    """--dry-run: the rank plumbing of the multi-GPU path with NO GPU and NO kernels (gloo, CPU tensors): every rank
    builds its own synthetic stream (seed + rank), "map points" of a rank-dependent count stand in for the
    triangulated ones, and the two collectives of the real run (gather_map_points inside the timed region,
    gather_summary after it) and the max-over-ranks timing run as they do on the GPUs.  `value` is null: nothing
    was measured."""
    import numpy as np
    import torch

    mod = load_product()  # the host-side generator only; no context is created
    seed = shard_seed(rank)
    frames = [mod.synth_frame(seed, k, 48, 64, 20, 3, 12) for k in range(3)]
    checksum = float(sum(int(f.sum()) for f in frames))
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    xyz = (np.arange((5 + 3 * rank) * 3, dtype=np.float64).reshape(-1, 3) + 1000.0 * rank)
    n_map_points = xyz.shape[0]
    if dist is not None:
        n_map_points = sum(a.shape[0] for a in gather_map_points(dist, xyz, "cpu", 64))
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rows = [[float(rank), float(seed), checksum, float(xyz.shape[0])]]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows = gather_summary(dist, rows[0], "cpu")
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "steps": 0, "warmup": 0,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic", "dry_run": True,
            "config": {"workload": "dry run of the rank plumbing: no GPU, no kernels, gloo", "name": "dry_run",
                       "parallelism": "frames sharded one stream per rank (dp%d), no data-path collective" % world},
            "map_points_gathered": n_map_points,
            "ranks": [{"rank": int(r[0]), "seed": int(r[1]), "stream_checksum": r[2], "map_points": int(r[3])} for r in rows],
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main_gn(args, rank, local_rank, world, dist, torch, coll_device="cuda"):
    """--workload c4: BASELINE.json configs[3].  Windows shard across ranks like frames do (independent
    windows, no data-path collective); the gather at the end of the job exchanges the optimised poses."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gn_scene

    mod = load_product()
    g = GN_WORKLOAD
    nw = args.batch or g["windows"]
    W, L = g["cams"], g["landmarks"]
    ctx = mod.Context(local_rank)
    wins = [gn_scene.make_window(shard_seed(rank) * 1000 + k, W, L) for k in range(nw)]
    poses0 = np.stack([w["poses0"] for w in wins])
    for k, w in enumerate(wins):
        poses0[k, : g["n_fixed"]] = w["poses_gt"][: g["n_fixed"]]  # the held cameras carry the gauge
    X0 = np.stack([w["X0"] for w in wins])
    obs = np.stack([w["obs"] for w in wins])
    Kc = wins[0]["K"]
    n_obs = int(np.isfinite(obs[..., 0]).sum())
    gn = mod.GaussNewton(nw, W, L, ctx)
    gn.upload(poses0, X0, obs)

    def run_steps(n):  # slamx_gn_iterate takes at most 64 iterations per call
        while n > 0:
            gn.iterate(Kc, g["lam"], g["n_fixed"], min(n, 64))
            n -= min(n, 64)

    run_steps(max(args.warmup, 1))
    ctx.synchronize()
    poses_w, _, cost_w, status = gn.download()
    assert not status.any(), "a window's reduced system was not positive definite"
    if dist is not None:
        gather_map_points(dist, poses_w.reshape(-1, 3), coll_device, nw * W * 4)  # untimed first use of the collective (lazy RCCL set-up)
    K = args.steps
    gn.set_profiling(True, every=4 if K >= 8 else 1)  # hipEvents around the kernels of every 4th iteration only
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    run_steps(K)  # K steps queued back to back + the cost evaluation after the last one
    poses, _, cost, status = gn.download()  # end of the job: results leave the device
    n_gathered = nw
    if dist is not None:
        n_gathered = sum(a.shape[0] for a in gather_map_points(dist, poses.reshape(-1, 3), coll_device, nw * W * 4)) // (W * 4)
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms, launches = gn.stage_ms()
    if rank == 0:
        n_gpus = world
        chunks = (L + 3) // 4
        mfma_flops = 40 * 2048.0 * chunks * nw         # 10 F F^T tiles x 3 + 5 Gram tiles x 2 k-steps per 4 landmarks
        survey_flops = 360.0 * n_obs                   # 2x9 block, J^T J + J^T r per observation
        acc_s = ms["accumulate"] * 1e-3
        achieved = mfma_flops / acc_s / 1e12 if acc_s > 0 else 0.0
        err = float(np.abs(poses - np.stack([w["poses_gt"] for w in wins])).max())
        out = {
            # configs[3]: one sliding window per frame of the 500-frame sequence, so a window-iteration is the "frame"
            # of this workload -- but it is NOT the 1080p frames/s metric: own metric text and unit
            "metric": "window-iterations/s, sliding-window Gauss-Newton (BASELINE.json configs[3]), 1/2/4/8 MI355X",
            "value": nw * K * n_gpus / elapsed, "unit": "window-iterations/s", "n_gpus": n_gpus, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": g["desc"], "name": "c4", "windows_per_step_per_gpu": nw, "cameras": W, "landmarks": L,
                       "observations_per_window": n_obs / nw, "lambda": g["lam"], "held_cameras": g["n_fixed"],
                       "parallelism": "windows sharded across GPUs (dp%d), no data-path collective" % n_gpus},
            "observations_per_s": float(n_obs) * K * n_gpus / elapsed,
            "windows_gathered": n_gathered,
            "stage_ms_per_launch": ms, "launches_timed": launches,
            "roofline": {"kernel": "gn_accumulate_kernel", "bound": "mfma", "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / MFMA_F64_PEAK_TFLOPS, "algorithmic_per_launch": mfma_flops,
                         "survey_flops_per_launch": survey_flops,
                         "survey_achieved": survey_flops / acc_s / 1e12 if acc_s > 0 else 0.0,
                         "avg_launch_ms": ms["accumulate"], "traffic": None,
                         "note": "algorithmic = FP64 MFMA flops the formulation needs (15 accumulator tiles per 4 landmarks); "
                                 "SURVEY.md 8d's per-observation figure (360 flop) is reported next to it"},
            "cost_first_step": float(cost_w[:, 0].sum()) if cost_w is not None else None,
            "cost_now": float(cost[:, -1].sum()), "max_pose_error_vs_generating": err,
            "device": ctx.device_name,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            import oracle_binding

            o = oracle_binding.load()
            t1 = time.perf_counter()
            done = 0
            while done < nw and time.perf_counter() - t1 < 15.0:
                w = wins[done]
                o.gn_step_schur(poses0[done], w["X0"], w["obs"], Kc, g["lam"], g["n_fixed"])
                done += 1
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": done / dt, "unit": "window-iterations/s", "cores": 1, "kind": "port",
                                   "sample": "%d windows of the same workload, one Gauss-Newton step each (CPU restatement "
                                             "oracle/gn_oracle.cpp:oracle_gn_step_schur, g++ -O3, 1 thread), %.1f s" % (done, dt),
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="1080p_2k", choices=sorted(WORKLOADS) + ["c4"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-upload-inclusive leg (N == 1 only)")
    ap.add_argument("--no-keypoints", action="store_true", help="diagnostic: match without the jump-radius penalty")
    ap.add_argument("--with-pose", action="store_true",
                    help="configs[2]'s second half inside the timed step: essential-matrix RANSAC, simpleRecoverPose, "
                         "triangulatePoints and PnP RANSAC for every frame pair of the batch (slamx_batch_pose)")
    ap.add_argument("--pose-iterations", type=int, default=256, help="essential-matrix hypotheses per frame pair")
    ap.add_argument("--pnp-iterations", type=int, default=100, help="PnP hypotheses per frame (loop_closure.yml: 100); 0 = off")
    ap.add_argument("--pose-source", default="scene", choices=["scene", "frontend"],
                    help="what the pose stage of --with-pose works on: `scene` (default) = keypoint / match arrays of a camera "
                         "moving through a 3-D point cloud at the workload's size, resident in a second batch (the workload's own "
                         "frames are a planar pattern under an image translation: E is degenerate, nothing has depth); `frontend` "
                         "= the match lists the front end just produced (diagnostic)")
    ap.add_argument("--pose-points", type=int, default=10000, help="keypoints per frame of the pose scene")
    ap.add_argument("--geometry", default="calibrated", choices=["calibrated", "reference"],
                    help="form of the PnP hypotheses in the timed step (include/slamx.h SLAMX_PNP_*): `reference` = "
                         "LoopClosure::solvePnP as written, which scores no inliers on real geometry; the other mode is "
                         "measured once after the timed region and reported next to it")
    ap.add_argument("--no-extras", action="store_true",
                    help="default workload only: skip the short runs of c3 / c3p / c4 reported under other_workloads")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 run: nccl (= RCCL, the product path) or gloo (rehearsal: "
                         "the collectives run on CPU tensors)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: every rank uses device 0 (needs --backend gloo; "
                         "RCCL refuses two ranks on one device)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank plumbing only: no GPU, no kernels, gloo (tests/test_distributed_cpu.py)")
    args = ap.parse_args(argv)
    # dmabuf IPC only on these hosts (RCCL and device-tensor sharing need it): also when a foreign launcher provides
    # RANK / WORLD_SIZE and this process is a rank, and before torch or HIP are loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        args.backend = "gloo"
    if args.share_device and args.backend != "gloo":
        ap.error("--share-device needs --backend gloo")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become it (children are started before torch / HIP are touched in this process)
        return launch_ranks(args, list(sys.argv[1:] if argv is None else argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0 and (world > 1 or args.gpus > 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d; running (and reporting) %d ranks\n"
                         % (args.gpus, world, world))
    import torch  # loaded first: libslamx then binds to the HIP runtime already in the process

    dist = None
    coll_device = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    if args.dry_run:
        return main_dry_run(args, rank, world, dist)
    n_gpus = world
    if args.workload == "c4":
        return main_gn(args, rank, local_rank, world, dist, torch, coll_device)

    mod = load_product()
    import numpy as np

    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    ctx = mod.Context(local_rank)  # no CPU fallback: raises without a HIP device
    det = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, wl["window"], 31, 256), ctx)
    matcher = mod.FeatureMatcher(mod.MatcherConfig(0, 0, 20, 1, 0.5), ctx)  # FilterMatches 0
    seed = shard_seed(rank)
    frames = np.stack([mod.synth_frame(seed, k, wl["rows"], wl["cols"], wl["n_rect"], wl["rmin"], wl["rmax"]) for k in range(B)])
    pyr = None
    if "levels" in wl:
        pyr = mod.Pyramid(det, wl["rows"], wl["cols"], B, wl["levels"], wl["scale"], wl["max_cand"], wl["max_kp"])
        batch = pyr.level(0)  # frames in, merged results out
    else:
        batch = mod.Batch(det, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"])
    batch.upload(frames)
    ctx.synchronize()
    pose = None
    pose_truth = None
    pose_other = None
    if args.with_pose:
        # a pinhole camera for the synthetic frames (the reference's camera.yml belongs to its 1392 x 512 KITTI frames)
        f = 0.9 * wl["cols"]
        Kcam = np.array([[f, 0, wl["cols"] / 2.0], [0, f, wl["rows"] / 2.0], [0, 0, 1.0]])
        pose_batch = batch
        if args.pose_source == "scene":
            # the stage is timed on a scene WITH DEPTH of the workload's size, resident in a second batch
            # of the same context (uploaded once, untimed -- as the frames are); the front end is still timed on the
            # workload's frames in the same step
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import pose_scene

            sc, sc_R, sc_t = pose_scene.loop_sequence(mod, np.random.default_rng(seed), B, args.pose_points, Kcam,
                                                      outliers=POSE_SCENE["outliers"], noise=POSE_SCENE["noise_px"],
                                                      keep=POSE_SCENE["keep"])
            pose_batch = mod.Batch(det, 64, 64, B, 1024, max(wl["max_kp"], args.pose_points))
            for k, (kps_k, m_k) in enumerate(sc):
                pose_batch.upload_results(k, kps_k, m_k)
            pose_truth = (pose_scene, sc_R, sc_t)
        modes = {"calibrated": mod.GEOMETRY_CALIBRATED, "reference": mod.GEOMETRY_REFERENCE}
        pose_kw = dict(iterations=args.pose_iterations, threshold_px=1.0, seed=SEED & 0xFFFFFFFF,
                       pnp_iterations=args.pnp_iterations, pnp_threshold=2.0, pnp_min_correspondences=20)
        pose = mod.PoseStage(pose_batch, Kcam, geometry_mode=modes[args.geometry], **pose_kw)
        if args.pnp_iterations > 0:
            other = "reference" if args.geometry == "calibrated" else "calibrated"
            pose_other = (other, mod.PoseStage(pose_batch, Kcam, geometry_mode=modes[other], **pose_kw))

    def step():
        if pyr is not None:
            pyr.detect_and_compute(B)
        else:
            batch.detect_and_compute(B)
        batch.match_consecutive(matcher, B, use_keypoints=not args.no_keypoints, first_train_frame=B - 1)
        if pose is not None:
            pose.run(B, B - 1)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ncand, nkp, nmatch = batch.counts(B)  # also surfaces capacity overflows loudly
    if args.warmup > 0:
        # the job's one-off end (downloads of the newest pair, triangulation) once untimed as well: its first call
        # pays for lazily created buffers, 1.5 - 2.6 ms of the timed region otherwise
        xyz_warm = triangulate_last_pair(ctx, batch, matcher, B, wl["cols"], wl["rows"])
        if dist is not None:
            # ... and the gather: the first collective of a kind pays for RCCL's lazy channel set-up, which must not land
            # in the timed region of the N > 1 runs only
            gather_map_points(dist, xyz_warm, coll_device, wl["max_kp"])

    K = args.steps
    # hipEvents between the kernels of every 8th timed step (every 4th / every step when there are few): an
    # event between two kernels keeps the second from starting while the first drains (round 1 measured ~9 % of
    # such a step; over the 20 default steps the events now cost about 1 %: 2.97 ms per step without any, 3.00 with)
    every = 8 if K >= 16 else (4 if K >= 8 else 1)
    sets = min((K + every - 1) // every, 256)
    batch.set_profiling(sets, every)
    if pose is not None:
        pose.set_profiling(every)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    # end of the job: triangulate the newest pair and gather the map points of all ranks
    ctx.synchronize()
    t_tail = time.perf_counter()
    xyz = triangulate_last_pair(ctx, batch, matcher, B, wl["cols"], wl["rows"])
    n_map_points = xyz.shape[0]
    if dist is not None:
        n_map_points = sum(a.shape[0] for a in gather_map_points(dist, xyz, coll_device, wl["max_kp"]))
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    end_of_job_ms = (time.perf_counter() - t_tail) * 1e3  # inside `elapsed`: downloads of the newest pair, triangulation, gather
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # result summary of every rank (the map points themselves were gathered inside the timed region)
        rows = gather_summary(dist, [nkp.sum(), nmatch.sum(), ncand.sum()], coll_device)
        tot_kp = sum(r[0] for r in rows)
        tot_match = sum(r[1] for r in rows)
        tot_cand = sum(r[2] for r in rows)
    else:
        tot_kp, tot_match, tot_cand = float(nkp.sum()), float(nmatch.sum()), float(ncand.sum())

    # per-stage device time, averaged over the timed steps (events on the kernels' own stream)
    stage = {s: 0.0 for s in mod.STAGES}
    for i in range(sets):
        ms = batch.stage_ms(i)
        for s in mod.STAGES:
            stage[s] += ms[s] / sets

    if pyr is not None:
        # level 0's BRIEF bracket of a pyramid also holds all of levels 1.. and the merges (slamx_pyramid_detect_and_compute)
        stage["pyramid_levels_1_up"] = stage["brief"]
        stage["brief"] = 0.0
    if rank == 0:
        frames_total = B * K * n_gpus
        fps = frames_total / elapsed
        M = tot_cand / (B * n_gpus)
        N = tot_kp / (B * n_gpus)
        px = wl["rows"] * wl["cols"]
        # Algorithmic work per launch (one launch = the whole batch of B frames), SURVEY.md 8d / DESIGN.md 4.
        # Bytes for the HBM-bound kernels, int32 ops for the Hamming kernel.
        live_words = 2  # 46 surviving BRIEF pairs -> 2 non-zero 32-bit words per descriptor
        # level 0's OWN keypoints (a pyramid's level-0 arrays hold the merged set of all levels, its describe kernel ran on its own)
        N0 = N - sum(float(pyr.level(l).counts(B)[1].sum()) for l in range(1, wl["levels"])) / B if pyr is not None else N
        greedy_nms = wl["window"] <= 24  # nms_greedy_kernel: no NMS rounds are launched (nms.hip)
        hbm_bytes = {
            "fast": px + px / 8.0 + px / 64.0 + 4.0 * M,  # frame read; bitmap, per-word counts, per-tile corner lists written
            "scan": 5.0 * px / 64,                 # per-word count bytes read, prefix written
            "expand": 24.0 * M,                    # list entry read, bitmap word + prefix gathered, (x,y) + key written
            "sort": 16.0 * M,                      # keys read, sorted keys + sorted (x,y) written, (x,y) gathered
            "nms_build": 8.0 * M + 32.0 * N0,      # greedy NMS: sorted key + (x,y) read; keypoint, slot and visiting list written
            "nms_rounds": 0.0 if greedy_nms else 4.0 * M,  # (only the generic NMS path launches rounds)
            "blur": 2.0 * px,                      # frame read, blurred frame written
            # orientation + BRIEF are one kernel (describe_fused_kernel, reported under "moments").  COMPULSORY bytes: the
            # blurred frame once (the 43 x 43 neighbourhoods of a few thousand keypoints cover it) + per keypoint the slot
            # and coordinates read (12 B) and angle + descriptor written (32 B).  (Until round 3 this line counted the
            # 43 x 48-byte tile of EVERY keypoint -- L2 traffic, several times the frame -- and printed fractions above the
            # kernel's real one: VERDICT r3 W5.)
            "moments": px + 44.0 * N0, "trig": 0.0, "brief": 0.0,
            "emit": 24.0 * N,
        }
        dom = max((k for k in stage if k != "pyramid_levels_1_up"), key=lambda k: stage[k])  # (a kernel, not a bracket)
        kernel_of = {"fast": "fast_detect_kernel", "match": "match_kernel<%d,2>" % live_words, "blur": "blur_kernel",
                     "brief": "brief_kernel", "nms_build": "nms_greedy_kernel", "sort": "sort_kernel",
                     "moments": "describe_fused_kernel", "expand": "expand_kernel", "nms_rounds": "nms_round/finish_kernel"}
        thr = mod.load_library().slamx_matcher_mfma_threshold()
        # the device's rule (scan_choice_kernel, per train frame; N is the batch average): >= thr keypoints, or >= 768 of them
        # spread over more than 1000 px -- synthetic frames are covered edge to edge, so the spread is the frame's diagonal
        wide = (wl["rows"] ** 2 + wl["cols"] ** 2) ** 0.5 >= 1000.0 + 64.0
        thr_eff = min(thr, 768) if wide else thr
        mfma_match = wl["max_kp"] >= thr_eff and N >= thr_eff
        if mfma_match:
            kernel_of["match"] = "match_spatial_kernel"
        if dom == "match" and mfma_match:
            # large train sets: Hamming distances as int8 dot products on the matrix cores (match.hip): one
            # v_mfma_i32_16x16x64_i8 per 16 x 16 pairs, K = 64 of which 48 carry data (46 bits + 2 augmenting terms)
            ops = 2.0 * 64.0 * N * N * B
            achieved = ops / (stage[dom] * 1e-3) / 1e12
            roof = {"kernel": kernel_of[dom], "bound": "mfma", "achieved": achieved, "peak": MFMA_I8_PEAK_TOPS,
                    "unit": "TOP/s", "frac": achieved / MFMA_I8_PEAK_TOPS, "algorithmic_per_launch": ops,
                    "pairs_per_s": N * N * B / (stage[dom] * 1e-3),
                    "popcount_equivalent_Gops": 2.0 * live_words * N * N * B / (stage[dom] * 1e-3) / 1e9,
                    "survey_ops_per_launch_256bit": 16.0 * N * N * B, "traffic": None,
                    "note": "int8 MFMA ops issued (K padded to 64); the kernel is bound by the vector ALU's per-query "
                            "bookkeeping beside the MFMAs, not by the matrix cores (DESIGN.md 4.12)"}
        elif dom == "match":
            # executed algorithmic work: xor + popcount-accumulate per live 32-bit word per pair;
            # SURVEY.md 8d's figure (16 ops per 256-bit pair) is reported next to it
            ops = 2.0 * live_words * N * N * B
            achieved = ops / (stage[dom] * 1e-3) / 1e9
            roof = {"kernel": kernel_of[dom], "bound": "valu-int32 (neither hbm nor mfma: SURVEY.md 8d)",
                    "achieved": achieved, "peak": VALU_HAMMING_PEAK_GOPS, "unit": "Gop/s",
                    "frac": achieved / VALU_HAMMING_PEAK_GOPS, "algorithmic_per_launch": ops,
                    "survey_ops_per_launch_256bit": 16.0 * N * N * B, "traffic": None}
        else:
            byts = hbm_bytes[dom] * B
            achieved = byts / (stage[dom] * 1e-3) / 1e9
            roof = {"kernel": kernel_of.get(dom, dom), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_per_launch": byts, "traffic": None}
        roof["avg_launch_ms"] = stage[dom]
        # HBM traffic per launch from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
        # separate runs of this same command by tools/collect_profiles.sh ->
        # profiles/r02_pmc_hbm_traffic_b256.csv), FETCH_SIZE doubled as the guide prescribes for
        # gfx950.  Check on fast_detect_kernel, whose compulsory traffic is known (every frame byte
        # once): 2 x 259 MB = 519 MB against 531 MB of frames.
        roof["traffic"], roof["traffic_source"] = pmc_traffic_bytes(roof["kernel"], B)
        if roof["traffic_source"] is not None:
            roof["traffic_source"] += " (a stored counter figure scaled to this batch, not a measurement of this run)"
        # every other kernel against its own roofline, for the record
        others = {}
        for k in stage:
            if stage[k] <= 0 or k not in hbm_bytes and k != "match":
                continue
            if k == "match" and mfma_match:
                o = 2.0 * 64.0 * N * N * B / (stage[k] * 1e-3) / 1e12  # int8 ops the matrix cores were issued (K padded to 64)
                others[k] = {"kernel": "match_spatial_kernel (+ spatial_order, scan_choice)", "ms": stage[k], "achieved": o,
                             "unit": "TOP/s", "peak": MFMA_I8_PEAK_TOPS, "frac": o / MFMA_I8_PEAK_TOPS}
            elif k == "match":
                o = 2.0 * live_words * N * N * B / (stage[k] * 1e-3) / 1e9
                others[k] = {"kernel": kernel_of["match"], "ms": stage[k], "achieved": o, "unit": "Gop/s",
                             "peak": VALU_HAMMING_PEAK_GOPS, "frac": o / VALU_HAMMING_PEAK_GOPS}
            elif hbm_bytes[k] > 0:
                g = hbm_bytes[k] * B / (stage[k] * 1e-3) / 1e9
                others[k] = {"kernel": kernel_of.get(k, k + "_kernel"), "ms": stage[k], "achieved": g, "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": g / HBM_PEAK_GBS}
        # a fraction above 1 (frames of a small workload resident in L2 / Infinity Cache, timer noise on a short kernel)
        # is flagged in the line, not fatal: one such value must not lose the whole run (tests assert on the flag)
        over = {k: v["frac"] for k, v in others.items() if v["frac"] > 1.0}
        if roof["frac"] > 1.0:
            over["dominant:" + str(roof["kernel"])] = roof["frac"]
        # the HBM-bound front-end as a whole: B_fe = 4WH + 16M + 52N bytes per frame
        fe_ms = sum(stage[k] for k in stage if k not in ("match", "emit", "pyramid_levels_1_up"))
        b_fe = (4.0 * px + 16.0 * M + 52.0 * N) * B
        out = {
            "metric": METRIC,
            "value": fps, "unit": "frames/s", "n_gpus": n_gpus, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.workload, "frames_per_step_per_gpu": B,
                       "candidates_per_frame": M, "keypoints_per_frame": N, "matches_per_frame": tot_match / (B * n_gpus),
                       "parallelism": "frames sharded one stream per GPU (dp%d), no data-path collective" % n_gpus},
            "matched_keypoints_per_s": N * fps,
            "map_points_gathered": n_map_points,
            "stage_ms_per_step": stage, "end_of_job_ms": end_of_job_ms,
            "stage_ms_note": "device time between hipEvents on the kernels' stream, averaged over the EVENTED steps only (every "
                             "%d-th timed step); those steps run 1-2 %% slower than the others, so the sum may exceed ms_per_step" % every,
            "collective": {"backend": ("rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo") if dist is not None else None,
                           "world_size_of_the_process_group": dist.get_world_size() if dist is not None else 1,
                           "data_path_collectives": 0, "end_of_job": "one all_gather_into_tensor of the map points (count in row 0)"},
            "roofline": roof,
            "kernel_rooflines": others,
            "roofline_over_1": over,
            "frontend_hbm": {"achieved": b_fe / (fe_ms * 1e-3) / 1e9 if fe_ms > 0 else None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "algorithmic_bytes_per_step": b_fe, "ms": fe_ms},
            "device": ctx.device_name,
        }
        if pose is not None:
            out["pose"] = pose_report(mod, pose, B, args, elapsed / K * 1e3, pose_truth, pose_other)
            out["config"]["workload"] += (" + pose stage per frame pair (essential-matrix RANSAC %d hypotheses, simpleRecoverPose, "
                                          "triangulatePoints, PnP RANSAC %d hypotheses)" % (args.pose_iterations, args.pnp_iterations))
            out["config"]["pose_input"] = (
                "scene with depth: %d keypoints per frame, a camera on a closed loop through a 3-D point cloud, %.0f %% of the "
                "query keypoints matched, %.0f %% of the matches wrong, keypoint noise %.2f px (tests/pose_scene.py:loop_sequence), "
                "resident in a second batch; the front end of the same step runs on the workload's frames"
                % (args.pose_points, 100 * POSE_SCENE["keep"], 100 * POSE_SCENE["outliers"], POSE_SCENE["noise_px"])
                if args.pose_source == "scene" else "the match lists of the workload's own frames (planar pattern under an image "
                "translation: a degenerate essential matrix, no depth)")
        if pyr is not None:
            out["config"]["pyramid"] = {"levels": wl["levels"], "scale_factor": wl["scale"], "level_shapes": pyr.shapes,
                                        "note": "stage_ms_per_step / roofline are level 0's kernels (candidates_per_frame too) plus "
                                                "pyramid_levels_1_up = everything of the other levels and the merges; "
                                                "keypoints_per_frame and the matcher cover all levels"}
        if n_gpus == 1 and not args.no_pcie:
            batch.set_profiling(0, 1)
            # (three lanes for the workloads with 256-frame batches of up to 15 MB of scratch per frame)
            out["pcie_inclusive"] = pcie_inclusive(mod, wl, B, frames, (ctx, batch, matcher, pyr), local_rank,
                                                   not args.no_keypoints, steps=12 if args.workload in ("1080p_2k", "c2", "c1") else 9,
                                                   n_lanes=4 if args.workload in ("1080p_2k", "c2", "c1") else 3)
            # SURVEY.md 8d defines frames/s END TO END (host frames in, host results out): that is this leg
            out["end_to_end"] = {"value": out["pcie_inclusive"]["value"], "unit": "frames/s",
                                 "definition": "SURVEY.md 8d: H2D of every frame + detect/describe/match + D2H of every "
                                               "result, batched (= pcie_inclusive); `value` is the HBM-resident rate"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mod, wl, frames)
        if n_gpus == 1 and args.workload == "1080p_2k" and not args.no_extras and not args.no_cpu_baseline:
            out["facade_latency"] = facade_latency()
        if n_gpus == 1 and args.workload == "1080p_2k" and not args.no_extras:
            out["other_workloads"] = other_workloads()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def facade_latency(reps=50):
    """Per-call latency of the reference-signature API on the reference's own frames -- what its test programs print
    (test/frontend/test_feature_detector.cpp:34-50, test_feature_matcher.cpp:41-68): slam::FeatureDetector::detect /
    compute / detectAndCompute and slam::FeatureMatcher::match through the header-only C++ facade (tools/facade_latency.cpp,
    a child process with its own context; host buffers in, host buffers out), median of `reps` calls after warm-up, next
    to the CPU restatement's time for the same calls on the same inputs (1 thread)."""
    import subprocess
    import numpy as np

    ref = os.path.join(ROOT, "tests", "golden", "ref_data")
    exe = os.path.join(ROOT, "tools", "bin", "facade_latency")
    imgs = [os.path.join(ref, "images", "000000000%d.png" % i) for i in (0, 1)]
    out = {"inputs": "tests/golden/ref_data/images/0000000000.png, 0000000001.png (the reference's KITTI pair, 1392 x 512), "
                     "shipped feature_detector.yml / feature_matcher.yml"}
    try:
        p = subprocess.run([exe, os.path.join(ref, "feature_detector.yml"), os.path.join(ref, "feature_matcher.yml")] + imgs + [str(reps)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        out["gpu_facade"] = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        out["gpu_facade"] = {"error": repr(e)[:200]}
    # the same facade objects over a whole sequence: slam::FrameBatch (include/slam/frontend/frame_batch.hpp) against one
    # detectAndCompute + one match call per frame, on the reference's ten KITTI frames cycled 25 times; the tool also
    # compares the two results byte for byte (tools/test_frame_batch.cpp)
    try:
        import re

        seq = sorted(os.path.join(ref, "images", n) for n in os.listdir(os.path.join(ref, "images")) if n.endswith(".png"))
        p = subprocess.run([os.path.join(ROOT, "tools", "bin", "test_frame_batch"), os.path.join(ref, "feature_detector.yml"),
                            os.path.join(ref, "feature_matcher.yml"), "25"] + seq, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
        text = p.stdout.decode()
        rates = [float(x) for x in re.findall(r"([0-9.]+) frames/s", text)]
        out["gpu_facade_sequence"] = {"frames": 25 * len(seq), "frame_by_frame_fps": rates[0], "frame_batch_fps": rates[2],
                                      "frame_batch_unpipelined_fps": rates[1],
                                      "results_identical": "batch == frame by frame: yes" in text,
                                      "note": "host images in, keypoints + descriptors + matches of every frame out, both ways"}
    except Exception as e:
        out["gpu_facade_sequence"] = {"error": repr(e)[:200]}
    # ... and on frames of the headline workload (1080p, ~2k keypoints), to set beside end_to_end: the C++ facade's sequence
    # interface against the bench's own rotating lanes over the C ABI
    try:
        p = subprocess.run([os.path.join(ROOT, "tools", "bin", "test_frame_batch"), os.path.join(ref, "feature_detector.yml"),
                            os.path.join(ref, "feature_matcher.yml"), "1", "synth:1080x1920:6800:3:24:512"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
        text = p.stdout.decode()
        rates = [float(x) for x in re.findall(r"([0-9.]+) frames/s", text)]
        out["gpu_facade_sequence_1080p"] = {"frames": 512, "frame_by_frame_fps": rates[0], "frame_batch_fps": rates[2],
                                            "frame_batch_unpipelined_fps": rates[1],
                                            "results_identical": "batch == frame by frame: yes" in text,
                                            "note": "512 synthetic 1080p frames of the headline workload through slam::FrameBatch::processSequence + "
                                                    "fetch; compare with end_to_end.value"}
    except Exception as e:
        out["gpu_facade_sequence_1080p"] = {"error": repr(e)[:200]}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        from PIL import Image

        o = oracle_binding.load()
        P = oracle_binding.DetectorParams(20, 12, 1, 12, 31, 256)
        M = oracle_binding.MatcherParams(1, 20, 1, 0.5)
        a, b = (np.ascontiguousarray(np.asarray(Image.open(f), np.uint8)) for f in imgs)
        ka, da = o.detect_and_compute(a, P)
        kb, db = o.detect_and_compute(b, P)

        def med(f, n=5):
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                f()
                ts.append((time.perf_counter() - t0) * 1e6)
            return sorted(ts)[len(ts) // 2]

        kd = o.detect(a, P)
        out["cpu_restatement_us"] = {"detect": med(lambda: o.detect(a, P)), "compute": med(lambda: o.compute(a, kd, P)),
                                     "detectAndCompute": med(lambda: o.detect_and_compute(a, P)),
                                     "match": med(lambda: o.match(da, db, ka, kb, M)),
                                     "note": "oracle/slam_oracle.cpp through ctypes, median of 5 calls, 1 thread"}
    except Exception as e:
        out["cpu_restatement_us"] = {"error": repr(e)[:200]}
    return out


# vector FP64 peak of MI355X: 78.6 TFLOP/s counting an FMA as two (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz); the
# parity path rounds every operation separately (no FMA contraction), so its instruction-rate ceiling is half of that
VALU_F64_PEAK_TFLOPS = 78.6


def pose_report(mod, pose, B, args, ms_per_step, truth=None, other=None):
    """Per-kernel times of the pose stage (hipEvents on the context stream around every launch of the recorded steps),
    what the stage found, and the kernels against their ceilings: the two scoring kernels are FP64 vector arithmetic
    (18 FMA-chain instructions = 33 flops per hypothesis and correspondence for the Sampson test; 3 x 8 + 20 flops and three
    IEEE divisions for the reprojection test), the refit and the DLT solves are the FP64 matrix-core contractions of the stage."""
    import numpy as np

    ms, nsets = pose.stage_ms()
    res = pose.download(B)
    n = res["n_matches"].astype(np.float64)
    m = res["pnp_correspondences"].astype(np.float64)
    H, Hp = args.pose_iterations, args.pnp_iterations
    total = sum(ms.values())
    out = {"ms_per_step": total, "share_of_step": total / ms_per_step if ms_per_step > 0 else None,
           "stage_ms_per_step": ms, "steps_timed": nsets,
           "hypotheses_per_pair": H, "pnp_hypotheses_per_frame": Hp,
           "matches_per_pair": float(n.mean()), "essential_inliers_per_pair": float(res["n_inliers"].mean()),
           "pairs_with_a_pose": int((res["status"] & (mod.POSE_TOO_FEW_MATCHES | mod.POSE_NO_TRAIN_FRAME) == 0).sum()),
           "pnp_correspondences_per_frame": float(m.mean()), "pnp_inliers_per_frame": float(res["pnp_inliers"].mean()),
           "pnp_frames_skipped": int((res["status"] & mod.POSE_PNP_SKIPPED != 0).sum())}
    rl = {}
    if ms["score"] > 0:
        fl = 33.0 * H * float(n.sum())
        a = fl / (ms["score"] * 1e-3) / 1e12
        rl["sampson_score_kernel"] = {"bound": "valu-f64", "achieved": a, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": a / VALU_F64_PEAK_TFLOPS, "algorithmic_per_launch": fl, "avg_launch_ms": ms["score"],
                                      "note": "33 flops per (hypothesis, correspondence) issued as 18 FP64 instructions (15 FMAs in chains, "
                                              "3 multiplies): the instruction-rate ceiling of that mix is 33 / 36 of the peak"}
    if ms["refit"] > 0:
        fl = 2.0 * 81.0 * float(res["n_inliers"].sum())
        issued = 2.0 * 256.0 * float((np.ceil(n / 64.0) * 64.0).sum())
        a = issued / (ms["refit"] * 1e-3) / 1e12
        rl["essential_refit_kernel"] = {"bound": "mfma", "achieved": a, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": a / MFMA_F64_PEAK_TFLOPS, "algorithmic_per_launch": fl, "issued_per_launch": issued,
                                        "avg_launch_ms": ms["refit"],
                                        "note": "one workgroup of four waves per pair: the 9 x 9 normal matrix over all inliers as "
                                                "16 x 16 x 4 FP64 MFMAs; the latency of its chunk loop and of the 9 x 9 Jacobi "
                                                "iteration bounds it, not the matrix cores"}
    if ms["pnp_solve"] > 0 and Hp > 0:
        nh = float((res["status"] & mod.POSE_PNP_SKIPPED == 0).sum()) * Hp
        # one lane per hypothesis: Householder QR of the 12 x 12 DLT matrix (4/3 n^3 = 2304 flops) + inverse iteration with
        # R^T R (2 n^2 = 288 per step, ~4 steps) -- FP64 VECTOR work, no matrix cores (the wave-per-hypothesis MFMA form,
        # which slamx_pnp_dlt_batch keeps, took 60 x as long); SURVEY 8d's bookkeeping figure is 576 n = 3456 flops per
        # Gram pass of the 2n x 12 system
        fl = (2304.0 + 4 * 288.0) * nh
        a = fl / (ms["pnp_solve"] * 1e-3) / 1e12
        rl["pnp_hyp_kernel"] = {"bound": "valu-f64 / LDS latency", "achieved": a, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": a / VALU_F64_PEAK_TFLOPS, "algorithmic_per_launch": fl,
                                "survey_flops_per_launch": 3456.0 * nh, "avg_launch_ms": ms["pnp_solve"],
                                "note": "a lane per hypothesis, its 12 x 12 matrix in LDS (72 KB per wave: two waves per CU): "
                                        "400 waves of dependent FP64 chains, bound by their latency"}
    out["rooflines"] = rl
    out["geometry_mode"] = args.geometry
    out["all_pairs_have_a_pose"] = out["pairs_with_a_pose"] == B
    out["pnp_inlier_share_of_correspondences"] = float(res["pnp_inliers"].sum()) / max(float(m.sum()), 1.0)
    if truth is not None:
        # what the stage recovered against the poses that generated the scene (frame f against f - 1 on the closed loop;
        # the PnP pose is relative to the camera the train frame's map points live in, f - 2, in units of that pair's baseline)
        ps, Rw, tw = truth
        rot_e, dir_e, prot_e, ptr_e = [], [], [], []
        for f in range(B):
            tf, tff = (f - 1) % B, (f - 2) % B
            Rr, tr = ps.relative_pose(Rw, tw, tf, f)
            if res["status"][f] & (mod.POSE_TOO_FEW_MATCHES | mod.POSE_NO_TRAIN_FRAME) == 0:
                rot_e.append(ps.rotation_angle_deg(res["R"][f], Rr))
                dir_e.append(ps.direction_angle_deg(res["t"][f], tr))
            if res["status"][f] & mod.POSE_PNP_SKIPPED == 0 and res["pnp_inliers"][f] > 0:
                R2, t2 = ps.relative_pose(Rw, tw, tff, f)
                base = float(np.linalg.norm(ps.relative_pose(Rw, tw, tff, tf)[1]))
                prot_e.append(ps.rotation_angle_deg(res["pnp_R"][f], R2))
                ptr_e.append(float(np.linalg.norm(res["pnp_t"][f] - t2 / base) / max(np.linalg.norm(t2 / base), 1e-300)))
        med = lambda a: float(np.median(a)) if len(a) else None  # noqa: E731
        top = lambda a: float(np.max(a)) if len(a) else None  # noqa: E731
        out["recovered_motion"] = {
            "essential": {"pairs": len(rot_e), "rotation_error_deg_median": med(rot_e), "rotation_error_deg_max": top(rot_e),
                          "translation_direction_error_deg_median": med(dir_e), "translation_direction_error_deg_max": top(dir_e)},
            "pnp": {"frames": len(prot_e), "rotation_error_deg_median": med(prot_e), "rotation_error_deg_max": top(prot_e),
                    "translation_relative_error_median": med(ptr_e), "translation_relative_error_max": top(ptr_e)},
            "note": "against the generating poses of the scene; PnP poses only of frames where a hypothesis scored"}
    if other is not None:
        # the other hypothesis form on the same arrays, once, after the timed region (same samples: same seed)
        name, st = other
        st.set_profiling(1)
        st.run(B, B - 1)
        st.run(B, B - 1)
        oms, _ = st.stage_ms()
        ores = st.download(B)
        out["other_geometry_mode"] = {
            "geometry_mode": name, "pnp_inliers_per_frame": float(ores["pnp_inliers"].mean()),
            "pnp_inlier_share_of_correspondences": float(ores["pnp_inliers"].sum()) / max(float(ores["pnp_correspondences"].sum()), 1.0),
            "pnp_solve_ms": oms["pnp_solve"], "pnp_score_ms": oms["pnp_score"],
            "note": ("LoopClosure::solvePnP exactly as written (loop_closure.cpp:238-274: pixel coordinates in the DLT, column-major "
                     "map of the row-major vector, t0 / ||R0||_F): no hypothesis reprojects a real scene within the threshold -- the "
                     "reference's own behaviour on such input, reproduced (tests/test_pose_batch_gpu.py compares the counts with the oracle)")
            if name == "reference" else "the build-defined corrected solve (SLAMX_GEOMETRY_CALIBRATED, include/slamx.h)"}
    return out


def other_workloads():
    """Short runs of BASELINE.json's other single-GPU configs (configs[2] with and without the pyramid, configs[3]),
    each in a child process after this run's timed region: their headline numbers travel with the default line.
    Full lines with the CPU baselines: profiles/r02_bench_c3.json, _c3p.json, _c4.json."""
    import subprocess

    res = {}
    for name, extra in (("c3", ["--steps", "10", "--warmup", "2"]),
                        ("c3_pose", ["--with-pose", "--steps", "10", "--warmup", "2"]),
                        ("c3p", ["--steps", "3", "--warmup", "1"]),
                        ("c4", ["--steps", "20", "--warmup", "3"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name.split("_")[0], "--no-cpu-baseline", "--no-pcie",
               "--no-extras"] + extra
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            res[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                         "workload": d["config"]["workload"], "dtype": d["dtype"],
                         "roofline": {k: d["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}}
            for k in ("keypoints_per_frame", "candidates_per_frame", "observations_per_window", "pose_input"):
                if k in d["config"]:
                    res[name][k] = d["config"][k]
            if "pose" in d:
                res[name]["pose"] = d["pose"]
        except Exception as e:  # a failed extra must not take the headline line down
            res[name] = {"error": repr(e)[:200]}
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
