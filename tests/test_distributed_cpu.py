"""world_size-2 gloo test of the multi-GPU plumbing of bench.py: frames shard one
independent stream per rank (no data-path collective) and the single collective gathers every
rank's result summary.  Runs on CPU; the GPU box uses the same code with backend nccl (RCCL)."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = _load_bench()
    prod = bench.load_product()
    seed = bench.shard_seed(rank)
    # host-side synthetic stream of this rank (the generator needs no GPU)
    frames = [prod.synth_frame(seed, k, 48, 64, 20, 3, 12) for k in range(3)]
    summary = [float(sum(int(f.sum()) for f in frames)), float(rank + 1), float(seed)]
    rows = bench.gather_summary(dist, summary, "cpu")
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(rows))
    # variable-length map-point gather as ONE collective (count in row 0 of a fixed-capacity payload): rank r
    # contributes 3 + 2r points; then ragged counts with a rank that has NO points, and a full payload
    pts = np.arange((3 + 2 * rank) * 3, dtype=np.float64).reshape(-1, 3) + 100 * rank
    parts = bench.gather_map_points(dist, pts, "cpu", 8)
    np.save(os.path.join(out_dir, "points%d.npy" % rank), np.concatenate(parts))
    empty = np.zeros((0, 3)) if rank == 0 else np.full((4, 3), 7.0)
    parts = bench.gather_map_points(dist, empty, "cpu", 4)
    assert [p.shape for p in parts] == [(0, 3), (4, 3)] and (parts[1] == 7.0).all()
    parts = bench.gather_map_points(dist, np.zeros((0, 3)), "cpu", 0)  # nobody has a point, capacity 0
    assert [p.shape for p in parts] == [(0, 3), (0, 3)]
    try:
        bench.gather_map_points(dist, np.zeros((5, 3)), "cpu", 4)  # more points than the agreed capacity: loud, before
        raise AssertionError("an over-full payload must be refused")  # any rank enters the collective
    except RuntimeError:
        pass
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1) and r0.shape == (2, 3)  # every rank sees every row
    bench = _load_bench()
    assert r0[0, 2] == bench.shard_seed(0) and r0[1, 2] == bench.shard_seed(1)
    assert r0[0, 2] != r0[1, 2]  # independent streams
    assert r0[0, 0] != r0[1, 0]  # ... with different pixels
    assert r0[:, 1].tolist() == [1.0, 2.0]
    p0 = np.load(tmp_path / "points0.npy")
    p1 = np.load(tmp_path / "points1.npy")
    expect = np.concatenate([np.arange(9.0).reshape(-1, 3), np.arange(15.0).reshape(-1, 3) + 100])
    assert np.array_equal(p0, expect) and np.array_equal(p1, expect)  # counts + padding handled


def _run_bench(argv, env_extra=None, timeout=180):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr.decode()


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start the two ranks itself (VERDICT r2 W1:
    --gpus used to be parsed and ignored).  Dry run: gloo, no GPU, the real collectives."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1, "exactly ONE JSON line, printed by rank 0"
    d = lines[0]
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    bench = _load_bench()
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert [r["seed"] for r in ranks] == [bench.shard_seed(0), bench.shard_seed(1)]
    assert ranks[0]["stream_checksum"] != ranks[1]["stream_checksum"]  # two different streams
    assert [r["map_points"] for r in ranks] == [5, 8]
    assert d["map_points_gathered"] == 13  # one all_gather of the count-prefixed payloads of both ranks


def test_bench_under_a_launcher_keeps_its_world_size():
    """Started the way the driver does for N > 1 (RANK / WORLD_SIZE in the environment) bench.py must NOT spawn
    again: here a world of one with --gpus 1."""
    rc, lines, err = _run_bench(["--gpus", "1", "--dry-run"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["map_points_gathered"] == 5


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies must end the whole job with a non-zero code instead of leaving the others waiting in the
    rendezvous: without a GPU the real (non dry-run) ranks fail at context creation -- there is no CPU fallback."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    rc, lines, err = _run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--batch", "2"])
    assert rc != 0 and lines == []
    assert "exited with code" in err
