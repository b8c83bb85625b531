"""The oracle under AddressSanitizer + UBSan on the CPU (odd sizes, K = 3, dmin > 0, windows that
wrap several times): the reference itself reads and writes out of bounds there (SURVEY Appendix A,
Q1/Q9/Q11/Q12); the oracle's safe rules must not."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_is_memory_safe_under_asan_ubsan():
    r = subprocess.run(["make", "-C", ORACLE, "asan"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "asan driver ok" in r.stdout
