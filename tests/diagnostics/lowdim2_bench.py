#!/usr/bin/env python3
"""Per-pass time of the fc2 lowdim path on the reference's half-moons grid cells (grid_search_halfMoons.py:159-169: fc2, hidden 32 .. 512,
250 samples, 100 test points): HIP events around back-to-back FGSM / expected-gradient passes, labels already int32 on the device.
usage: python tests/diagnostics/lowdim2_bench.py [hidden ...]      (run it under `rocprofv3 --kernel-trace --stats` for the per-kernel durations)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # tests/diagnostics/ -> repository root
from oracle import bnn_oracle as O                                     # noqa: E402  (synthetic posterior generator only)
from robustbnns_amd import AttackEngine, StackedPosterior               # noqa: E402

S, N, DEV = 250, 100, "cuda:0"
hidden = [int(v) for v in sys.argv[1:]] or [32, 128, 256, 512]
x, y = O.synthetic_inputs(N, (1, 2, 1), 2, seed=9)
xd, lab = x.to(DEV), y.argmax(-1).to(device=DEV, dtype=torch.int32)
for h in hidden:
    sp = StackedPosterior("fc2", "leaky", (1, 2, 1), 2, h, O.synthetic_posterior("fc2", 2, h, 2, S, 0.3), DEV)
    for prec in ("auto", "exact"):
        eng = AttackEngine(sp, precision=prec)
        row = []
        for fn in (lambda: eng.fgsm(xd, lab, S, 0.3), lambda: eng.loss_gradients(xd, lab, S), lambda: eng.pgd(xd, lab, S, 0.3, iters=40)):
            for _ in range(3):
                fn()
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / reps * 1e3)
        print(f"hidden {h:3d} [{eng.precision:6s}] FGSM pass {row[0]:8.1f} us   expected-gradient pass {row[1]:8.1f} us   PGD T=40 {row[2]:9.1f} us = {row[2] / 40:7.1f} us / iteration", flush=True)
