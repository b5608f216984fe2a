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
    # variable-length map-point gather: rank r contributes 3 + 2r points
    pts = np.arange((3 + 2 * rank) * 3, dtype=np.float64).reshape(-1, 3) + 100 * rank
    parts = bench.gather_map_points(dist, pts, "cpu")
    np.save(os.path.join(out_dir, "points%d.npy" % rank), np.concatenate(parts))
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
