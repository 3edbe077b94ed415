"""Diagnostic (GPU): the fp64-oracle case fc / leaky / (1, 2, 1) / C 2 / H 16 / S 4 / N 300 on the fp32-MFMA kernels, per loss mode, per point, twice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bnn_oracle as O                                     # noqa: E402
from robustbnns_amd import AttackEngine, StackedPosterior, _hip        # noqa: E402

arch, act, shape, C, H, S, N, std = "fc", "leaky", (1, 2, 1), 2, 16, 4, 300, 0.5
D = 2
post = O.synthetic_posterior(arch, D, H, C, S, std)
x, y = O.synthetic_inputs(N, shape, C, seed=H + N)
lab = y.argmax(-1)
p64 = O.cast(post, torch.float64)
ok = O.kink_margin(x.double(), p64, arch, act, S) > 2e-6
for rep in range(2):
    eng = AttackEngine(StackedPosterior(arch, act, shape, C, H, post, "cuda:0"), precision="fast")
    labd = lab.int().to("cuda:0")
    for mode, kind in ((_hip.LOSS_MEAN_PROB, "bnn"), (_hip.LOSS_MEAN_LOGIT, "ensemble")):
        G = eng.gradient(eng.pad_inputs(x), labd, None, S, mode)[:, :D].cpu().reshape(N, -1).double()
        ref = O.meanprob_gradients(x.double(), lab, p64, arch, act, S, kind=kind).reshape(N, -1)
        e = (G - ref).abs().max(1)[0] / ref.abs().max(1)[0]
        e_ok = torch.where(ok, e, torch.zeros_like(e))
        w = int(e_ok.argmax())
        print(f"rep {rep} {kind}: global rel err {float((G[ok] - ref[ok]).abs().max() / ref[ok].abs().max()):.3e}; per-point worst {float(e_ok.max()):.3e} at {w}: "
              f"G {G[w].tolist()} ref {ref[w].tolist()}; points beyond 1e-5: {int((e_ok > 1e-5).sum())}; kink margin there {float(O.kink_margin(x.double(), p64, arch, act, S)[w]):.3e}")
