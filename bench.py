#!/usr/bin/env python3
"""bench.py -- disparity maps/sec of the MI355X-native stereo-matching hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json): synthetic 1242x375 grayscale pairs, D=128, K=2 (config C2's shape);
one "step" = one pass of the whole hot path (prologue -> cost volume + aggregation + WTA ->
secondary matching -> bilateral fills) over a batch of PAIRS_PER_GPU independent pairs that
are already resident in HBM (float32 gray, as B_alg = 12*H*W bytes/pair assumes).  With
N > 1 (one process per GPU under torch.distributed.run) every rank owns its own engine and
its own pairs -- independent units, no data-path collective -- so scaling is weak; at N = 8
a step is BASELINE config 3 (512 pairs).  Rank 0 prints ONE JSON line.

The HIP library is the only compute path; the CPU oracle is used solely for the reported
`cpu_baseline` (rank 0, N = 1, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]

H, W, K, D = 375, 1242, 2, 128
PAIRS_PER_GPU = 64
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
B_ALG_PER_PAIR = 12 * H * W      # 2 gray f32 inputs + 1 f32 output


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (OpenMP build of oracle/stereo_oracle.c) timed on the host cores: a
    reported baseline only.  The reference ships no CPU path for this algorithm."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    import stereo_synthetic as syn
    o = oracle_lib.get(parallel=True)
    # the GPU box gives one GPU's job a 16-core share of the host (more threads only oversubscribe)
    cores = min(o.max_threads(), os.cpu_count() or 1, 16)
    o.set_threads(cores)
    cfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    left, right, _ = syn.make_pair(H, W, D, K, 0)
    t0 = time.perf_counter()
    o.run(cfg, left, right)
    first = time.perf_counter() - t0
    n = max(1, min(16, int(budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n):
        o.run(cfg, left, right)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} x C2 pair (1242x375, D=128, K=2) through oracle/stereo_oracle.c "
                      f"(-O3 -mavx2 -fopenmp, {cores} threads); the reference has no CPU implementation"}


def measured_traffic(kernel: str, pairs: int):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under
    profiles/ (tools/pmc_traffic.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same
    command, gfx950 corrections per MI355X_MICROARCH.md).  None if no matching record exists."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    k = rec.get("kernels", {}).get(kernel)
    if not k or rec.get("pairs_per_launch") != pairs:
        return None
    return k.get("hbm_bytes_per_launch")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per GPU per step")
    ap.add_argument("--mode", default="auto", choices=["auto", "exact_order", "fast_grid"])
    ap.add_argument("--engines", type=int, default=1,
                    help="engines (each on its own HIP stream, an equal share of the step's pairs) per GPU; 2 lets the "
                         "tail of one engine's kernels overlap the other's (+~8 %%) but blurs per-kernel durations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency", action="store_true",
                    help="also time config C2 literally (one pair per call, back to back); off by default so "
                         "that a rocprofv3 run of the default command sees batch launches only")
    args = ap.parse_args()

    import numpy as np
    import torch
    import cuda_depth
    import stereo_synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    # SMX_BENCH_FORCE_DIST=1: rehearse the RCCL rendezvous / barrier / MAX-reduce with a single rank
    if world > 1 or (os.environ.get("SMX_BENCH_FORCE_DIST") and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n = args.pairs
    # distinct synthetic pairs per rank (seed 1234 + global index); generate a few, tile to n
    uniq = min(n, 8)
    L, R = syn.make_batch(uniq, H, W, D, K, first_index=rank * n)
    reps = (n + uniq - 1) // uniq
    left = torch.from_numpy(np.concatenate([L] * reps)[:n]).cuda()
    right = torch.from_numpy(np.concatenate([R] * reps)[:n]).cuda()
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K,
                                                 min_disparity=0, max_disparity=D - 1)
    E = max(1, args.engines)
    if n % E:
        raise SystemExit(f"--pairs {n} must be a multiple of --engines {E}")
    per = n // E
    engines = [cuda_depth.StereoMatching(cfg, max_batch=per, match_mode=args.mode, device=local_rank) for _ in range(E)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(E - 1)]
    sm = engines[0]
    out = torch.empty((n, H, W), dtype=torch.float32, device="cuda")

    def step():
        for i, (eng, st) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(st):
                eng.compute_disparity_map_batch(left[i * per:(i + 1) * per], right[i * per:(i + 1) * per],
                                                out[i * per:(i + 1) * per])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    sm.profile_begin(args.steps)          # HIP events around every kernel, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    prof = sm.profile_end()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # timing only, not part of the data path
        elapsed = float(t.item())

    # single-pair latency (config C2 as written: one pair per call), rank 0 only
    lat_us = None
    if rank == 0 and args.latency:
        sm1 = cuda_depth.StereoMatching(cfg, max_batch=1, match_mode=args.mode, device=local_rank)
        for _ in range(20):
            sm1.compute_disparity_map_gray(left[0], right[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        iters = 200
        for _ in range(iters):
            sm1.compute_disparity_map_gray(left[0], right[0])
        torch.cuda.synchronize()
        lat_us = (time.perf_counter() - t1) / iters * 1e6

    if rank == 0:
        pairs = n * world * args.steps
        value = pairs / elapsed
        dominant = max((k for k in prof if prof[k][1] > 0), key=lambda k: prof[k][0])
        dom_ms, dom_launches = prof[dominant]
        achieved = (B_ALG_PER_PAIR * per) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0   # per launch: `per` pairs
        line = {
            "metric": "disparity maps/sec (stereo pairs/sec) at 1242x375 D=128",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2 shape 1242x375 gray f32, D=128, K=2, {n} device-resident pairs per GPU "
                                   f"per step through the batch C ABI (= config C3 at 8 GPUs); match_mode={args.mode}" + (f", {E} engines x {per} pairs on {E} streams" if E > 1 else ""),
                       "pairs_per_gpu_per_step": n, "parallelism": f"independent pairs x{world} (no collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(dominant, per),
                         "kernel": dominant, "kernel_ms": dom_ms, "launches": dom_launches,
                         "algorithmic_bytes_per_launch": B_ALG_PER_PAIR * per},
            "kernel_ms": {k: round(v[0], 5) for k, v in prof.items() if v[1] > 0},
            "whole_path_hbm_frac": (B_ALG_PER_PAIR * value / world) / 1e9 / HBM_PEAK_GBPS,
            "bound_note": "the HBM fraction is a ceiling indicator only: k_match_fast is VALU-issue bound (no MFMA: "
                          "integer abs-diff reductions), ~83 % of its SIMD issue slots are busy over a launch "
                          "(profiles/r01_sq_counters.txt, DESIGN.md section 3.4)",
            "single_pair_latency_us": lat_us,
            "match_mode_used": sm.last_match_mode(),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
