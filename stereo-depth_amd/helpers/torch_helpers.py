"""cuda_perf_clock: mirrors /root/reference/src/python/helpers/torch_helpers.py:19-28
(the reference's only profiling hook on this path)."""
from contextlib import contextmanager
from time import time

import torch


@contextmanager
def cuda_perf_clock(name: str, do_log: bool = True):
    start = time()
    try:
        yield
    finally:
        if do_log:
            torch.cuda.synchronize()
            end = time()
            print(f"{name} took: {(end - start) * 1000:.3f} ms.")
