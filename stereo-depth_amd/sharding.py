"""Static shard plan for a batch of independent stereo pairs.

The path has no exchange step: pair i of a job goes to device i mod G, every device owns
one engine and its own streams, and nothing crosses xGMI.  The reference processes frames
serially on one GPU (python/pipeline/depth_estimation_pipeline_runner.py:51-52); this is
the loop that is being parallelised.

`bench.py`, the multi-process tests and any caller that feeds several GPUs use these two
functions, so the rule exists once.
"""
from __future__ import annotations

from typing import List


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
