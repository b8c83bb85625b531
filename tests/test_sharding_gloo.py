"""Multi-GPU path on CPU: world_size-2 gloo rehearsal of bench.py's sharding logic.

The path shards by independent stereo pairs: rank r owns pairs
[r*n, (r+1)*n) with seeds 1234 + global index, no data-path collective; the only collective is
the MAX over ranks of the elapsed time.  Here the per-rank "engine" is the CPU oracle (test
infrastructure) so the plumbing -- rank-disjoint seeds, barrier, max-reduce, whole-job
aggregate -- runs without a GPU.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    for p in (os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle_lib
    import stereo_synthetic as syn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, K, D = 48, 80, 2, 16
    L, R = syn.make_batch(n, H, W, D, K, first_index=rank * n)          # bench.py's shard rule
    cfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    o = oracle_lib.get()
    dist.barrier()
    outs = np.stack([o.run(cfg, L[i], R[i]) for i in range(n)])
    elapsed = torch.tensor([0.25 + rank], dtype=torch.float64)            # deterministic stand-in
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)                        # timing only
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), left=L, out=outs, elapsed=elapsed.numpy())
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world, n = 2, 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # disjoint shards: no pair is processed twice, and shard 1 continues where shard 0 ends
    for i in range(n):
        for j in range(n):
            assert not np.array_equal(r0["left"][i], r1["left"][j])
    sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
    import stereo_synthetic as syn
    L_all, _ = syn.make_batch(world * n, 48, 80, 16, 2, first_index=0)
    assert np.array_equal(np.concatenate([r0["left"], r1["left"]]), L_all)
    # both ranks agree on the max-over-ranks time; whole-job throughput = all pairs / that time
    assert float(r0["elapsed"][0]) == float(r1["elapsed"][0]) == 1.25
    assert r0["out"].shape == r1["out"].shape == (n, 48, 80)
