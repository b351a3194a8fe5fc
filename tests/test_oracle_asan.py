"""The CPU oracle under AddressSanitizer + UBSan (`make -C oracle asan`): the checker itself must not read or write
out of bounds on the edge shapes the parity tests feed it (sanitizers: CPU build only)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r"""
import ctypes as C, numpy as np, sys
lib = C.CDLL(sys.argv[1])
lib.nsof_ref_farneback_u8.restype = C.c_int
lib.nsof_ref_farneback_u8.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p,
                                      C.c_ssize_t, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
rng = np.random.default_rng(0)
for (h, w), p in [((97, 131), (0.5, 3, 15, 3, 5, 1.2)), ((64, 80), (0.6, 3, 3, 3, 10, 1.05)), ((2, 2), (0.5, 3, 15, 3, 5, 1.2)),
                  ((1, 9), (0.6, 3, 4, 2, 1, 1.05)), ((40, 33), (0.75, 5, 8, 1, 3, 0.9))]:
    big = rng.integers(0, 256, (2, h + 4, w + 7), dtype=np.uint8)
    a, b = big[0, 2:2 + h, 3:3 + w], big[1, 2:2 + h, 3:3 + w]          # strided views, like the ROI crops
    out = np.empty((h, w, 2), np.float32)
    rc = lib.nsof_ref_farneback_u8(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], w, h, out.ctypes.data,
                                   out.strides[0], p[0], p[1], p[2], p[3], p[4], p[5], 0)
    assert rc == 0 and np.isfinite(out).all(), (h, w, rc)
print("asan-ok")
"""


def test_oracle_clean_under_asan(tmp_path):
    gcc_asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not gcc_asan or not os.path.isabs(gcc_asan) or not os.path.exists(gcc_asan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "_build", "libnsof_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=gcc_asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86")
    r = subprocess.run([sys.executable, "-c", SCRIPT, so], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
