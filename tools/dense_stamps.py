#!/usr/bin/env python3
"""Where a block of conv_bwd_dense_x3_kernel spends its cycles (diagnostic build -DRBNN_DENSE_STAMPS=1|2, tools/dense_stamps.sh): runs a bench
workload in this process, then reads the s_memtime sums of waves 0 and 3 (rbnn_debug_dense_stamps) and prints cycles per block and segment.
usage: RBNN_ALLOW_ABLATION=1 python tools/dense_stamps.py <bench args...>"""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustbnns_amd import _hip                                           # noqa: E402

lib = C.CDLL(_hip.LIB_PATH)
lib.rbnn_debug_dense_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
_hip.load()
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
buf = (C.c_ulonglong * 64)()
assert lib.rbnn_debug_dense_stamps(buf, 1) == 0
for w, base in ((os.environ.get("STAMP_WA", "0"), 0), (os.environ.get("STAMP_WB", "3"), 32)):
    v = [buf[base + i] for i in range(32)]
    nb = max(v[25], 1)
    print(f"wave {w}: blocks {v[25]}  cycles/block {v[24] / nb:9.0f}")
    for p in range(3):
        if v[8 * p + 1]:
            print(f"   pass {p}: prologue {v[8 * p] / nb:8.0f}   K loop {v[8 * p + 1] / nb:9.0f}   col2im {v[8 * p + 2] / nb:8.0f}")
    if v[26] or v[27]:
        print(f"   inside the K loops: barrier {v[26] / nb:8.0f}   weight-tile waits {v[27] / nb:8.0f}")
