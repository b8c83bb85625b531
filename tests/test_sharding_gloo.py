"""Multi-GPU path on CPU: world_size-2 rehearsal of bench.py's OWN sharding and timing code.

The path shards by independent stereo pairs: pair i -> device i mod N
(`sharding.shard_indices`), no data-path collective; the ranks share only a gloo barrier and the
MAX of the elapsed time (`bench.TimingGroup`, `bench.timed_steps`).  Here the per-rank "engine"
is the CPU oracle (test infrastructure), so everything except the HIP calls runs without a GPU.
"""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stereo-depth_amd")


def _load_bench():
    spec = importlib.util.spec_from_file_location("smx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, total, out_dir):
    for p in (PKG, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import oracle_lib
    import sharding
    import stereo_synthetic as syn
    bench = _load_bench()
    assert bench.rank_env() == (rank, rank, world)
    Hh, Ww, Kk, Dd = 48, 80, 2, 16
    mine = sharding.shard_indices(total, world, rank)                   # the rule bench.py uses
    L = np.stack([syn.make_pair(Hh, Ww, Dd, Kk, i)[0] for i in mine])
    R = np.stack([syn.make_pair(Hh, Ww, Dd, Kk, i)[1] for i in mine])
    cfg = oracle_lib.OracleConfig(height=Hh, width=Ww, downscale_factor=Kk, min_disparity=0, max_disparity=Dd - 1)
    o = oracle_lib.get()
    group = bench.TimingGroup(rank, world)
    outs = {}
    steps_run = []

    def step():
        steps_run.append(1)
        for c in sharding.calls_for_shard(len(mine), 2):                # batch calls of <= 2 pairs
            for k in c:
                outs[mine[k]] = o.run(cfg, L[k], R[k])

    elapsed = bench.timed_steps(step, lambda: None, group, steps=2, warmup=1)
    slow = group.max(10.0 + rank)                                       # MAX really is over the ranks
    group.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=np.array(mine), left=L,
             out=np.stack([outs[i] for i in mine]), elapsed=elapsed, slow=slow, steps=len(steps_run))


def test_two_rank_sharding_uses_bench_code(tmp_path):
    bench = _load_bench()
    world, total = 2, 7                                                 # ragged: 4 + 3 pairs
    mp.spawn(_worker, args=(world, bench.free_port(), total, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    # disjoint, complete, round-robin
    assert list(r[0]["idx"]) == [0, 2, 4, 6] and list(r[1]["idx"]) == [1, 3, 5]
    sys.path[:0] = [PKG, os.path.join(ROOT, "oracle")]
    import oracle_lib
    import stereo_synthetic as syn
    cfg = oracle_lib.OracleConfig(height=48, width=80, downscale_factor=2, min_disparity=0, max_disparity=15)
    for k in range(world):
        for j, i in enumerate(r[k]["idx"]):
            l, rr, _ = syn.make_pair(48, 80, 16, 2, int(i))
            assert np.array_equal(r[k]["left"][j], l)                   # seed = 1234 + GLOBAL index
            assert np.array_equal(r[k]["out"][j], oracle_lib.get().run(cfg, l, rr))
    # both ranks report the same (max-over-ranks) time; 1 warm-up + 2 timed steps ran
    assert float(r[0]["elapsed"]) == float(r[1]["elapsed"]) > 0
    assert float(r[0]["slow"]) == float(r[1]["slow"]) == 11.0
    assert int(r[0]["steps"]) == int(r[1]["steps"]) == 3


def test_shard_plan_properties():
    sys.path.insert(0, PKG)
    import sharding
    for total, world in ((512, 8), (512, 1), (64, 4), (5, 8), (0, 2)):
        parts = [sharding.shard_indices(total, world, r) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(total))
        assert all(i % world == r for r, p in enumerate(parts) for i in p)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert [list(c) for c in sharding.calls_for_shard(5, 2)] == [[0, 1], [2, 3], [4]]
    assert sharding.calls_for_shard(0, 4) == []
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 2, 2)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks_cpu_dry_run(world):
    """`python bench.py --gpus N` without any wrapper (N = 2, and N = 8 as the driver's scaling run uses it): the parent
    starts N rank processes, each pins itself to its share of the host cores, they rendezvous over gloo (no GPU needed
    for that) and then refuse to run because the HIP path has no CPU fallback.  With GPUs the same command prints the
    JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the self-launch is exercised by the real bench run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--pairs", "2", "--steps", "1",
                        "--warmup", "0", "--quick"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    for r in range(world):
        assert f"rank {r}/{world}: needs a GPU" in p.stderr, p.stderr[-2000:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]        # no result line without a GPU


def test_rank_cpu_slices_partition_the_cores_next_to_each_gpu():
    """bench.py pins every rank's host threads before its first GPU call: ranks whose GPUs share a NUMA node split that
    node's cores; without topology information the allowed cores are cut into contiguous 1/N slices."""
    sys.path.insert(0, PKG)
    import sharding
    node0, node1 = list(range(0, 48)), list(range(48, 96))
    lists = [node0] * 4 + [node1] * 4                                    # 8 GPUs, 4 per socket
    slices = [sharding.rank_cpu_slice(r, 8, allowed=list(range(96)), local_lists=lists) for r in range(8)]
    assert sorted(c for s in slices for c in s) == list(range(96))       # disjoint and complete
    assert all(set(slices[r]) <= set(lists[r]) and len(slices[r]) == 12 for r in range(8))
    # a cgroup that only allows part of the machine
    part = [sharding.rank_cpu_slice(r, 8, allowed=list(range(16, 80)), local_lists=lists) for r in range(8)]
    assert all(part[r] and set(part[r]) <= set(lists[r]) & set(range(16, 80)) for r in range(8))
    # no topology: contiguous slices; fewer cores than ranks: everybody gets something
    flat = [sharding.rank_cpu_slice(r, 4, allowed=list(range(10)), local_lists=[]) for r in range(4)]
    assert sorted(c for s in flat for c in s) == list(range(10))
    assert all(sharding.rank_cpu_slice(r, 8, allowed=[3, 4], local_lists=[]) for r in range(8))
    assert sharding._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    with pytest.raises(ValueError):
        sharding.rank_cpu_slice(8, 8)
