import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as G; G.build()
from oracle import bnn_oracle as O
from robustbnns_amd import AttackEngine, StackedPosterior, _hip
arch, act, shape, C, H, S, N, std = "fc", "leaky", (1, 10, 10), 2, 128, 3, 300, 0.3
D = int(np.prod(shape))
post = O.synthetic_posterior(arch, D, H, C, S, std)
x, y = O.synthetic_inputs(N, shape, C, seed=H + N)
p64 = O.cast(post, torch.float64)
ref = O.loss_gradients(x.double(), y, p64, arch, act, S).reshape(N, -1)
km = O.kink_margin(x.double(), p64, arch, act, S)
for mode in ("exact", "split"):
    eng = AttackEngine(StackedPosterior(arch, act, shape, C, H, post, "cuda:0"), precision=mode)
    g = eng.loss_gradients(x, y, S).cpu().reshape(N, -1).double()
    rel = (g - ref).abs().max(1)[0] / ref.abs().max(1)[0]
    bad = (rel > 1e-5).nonzero().flatten()
    print(mode, "max rel", float(rel.max()), "n bad", bad.numel(), bad[:20].tolist())
    for n in bad[:6].tolist():
        print("   point", n, "rel", float(rel[n]), "kink margin", float(km[n]), "max|g|", float(ref[n].abs().max()))
    ws = eng.workspace(N, S)
    if mode == "exact":
        m_exact = ws["mask1"].clone(); dz_exact = ws["dZ"].clone()
    else:
        d = (ws["mask1"] ^ m_exact)
        print("   mask words differing:", int((d != 0).sum()), " dZ max diff", float((ws["dZ"] - dz_exact).abs().max()), "max|dZ|", float(dz_exact.abs().max()))
        print("   gscale range", float(ws["split"]["g_scale"][:N].min()), float(ws["split"]["g_scale"][:N].max()))
