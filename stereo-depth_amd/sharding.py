"""Static shard plan for a batch of independent stereo pairs.

The path has no exchange step: pair i of a job goes to device i mod G, every device owns
one engine and its own streams, and nothing crosses xGMI.  The reference processes frames
serially on one GPU (python/pipeline/depth_estimation_pipeline_runner.py:51-52); this is
the loop that is being parallelised.

`bench.py`, the multi-process tests and any caller that feeds several GPUs use these two
functions, so the rule exists once.
"""
from __future__ import annotations

import glob
import os
from typing import List, Optional


def shard_indices(total_pairs: int, world_size: int, rank: int) -> List[int]:
    """Global pair indices owned by `rank`: round-robin, i -> i mod world_size."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    if total_pairs < 0:
        raise ValueError("total_pairs must be non-negative")
    return list(range(rank, total_pairs, world_size))


def calls_for_shard(n_pairs: int, max_batch: int) -> List[range]:
    """Splits a shard of n_pairs (positions in the rank's local order) into batch-ABI calls of at
    most `max_batch` pairs: [range(0, b), range(b, 2b), ...]; the last call may be ragged."""
    if max_batch < 1:
        raise ValueError("max_batch must be positive")
    return [range(s, min(s + max_batch, n_pairs)) for s in range(0, n_pairs, max_batch)]


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpulists(sysfs: str = "/sys/class/drm") -> List[List[int]]:
    """CPU lists next to every AMD GPU, in PCI-address order (the order HIP enumerates devices in on one node), read
    from sysfs WITHOUT touching the GPU runtime: .../renderD*/device/{vendor, local_cpulist}."""
    found = []
    for node in glob.glob(os.path.join(sysfs, "renderD*")):
        dev = os.path.join(node, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip().lower() != "0x1002":
                continue
            found.append((os.path.basename(os.path.realpath(dev)), _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())))
        except (OSError, ValueError):
            continue
    return [cpus for _, cpus in sorted(found)]


def rank_cpu_slice(local_rank: int, world_size: int, allowed: Optional[List[int]] = None,
                   local_lists: Optional[List[List[int]]] = None) -> List[int]:
    """Host cores for the rank that drives GPU `local_rank`: an equal share of the cores next to its GPU (the ranks
    whose GPUs hang off the same NUMA node split that node's cores between them); without topology information a
    contiguous 1/world_size slice of the allowed cores.  Used by bench.py BEFORE the first GPU call, so that the
    uploads of the PCIe-inclusive C3 leg come from memory near the device."""
    if world_size < 1 or not (0 <= local_rank < world_size):
        raise ValueError(f"local_rank {local_rank} outside [0, {world_size})")
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    if local_lists is None:
        local_lists = gpu_local_cpulists()
    if len(local_lists) >= world_size and local_lists[local_rank]:
        mine = [c for c in local_lists[local_rank] if c in set(allowed)]
        peers = [r for r in range(world_size) if local_lists[r] == local_lists[local_rank]]
        k, m = peers.index(local_rank), len(peers)
        share = mine[k * len(mine) // m:(k + 1) * len(mine) // m]
        if share:
            return share
    n = len(allowed)
    share = allowed[local_rank * n // world_size:(local_rank + 1) * n // world_size]
    return share or allowed
