import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import bnn_oracle as O
from robustbnns_amd import AttackEngine, StackedPosterior, _hip
arch, act, shape, C, H, S, N, std = ("fc", "leaky", (1, 28, 28), 10, 512, 7, 333, 0.05)
D = 784
post = O.synthetic_posterior(arch, D, H, C, S, std)
x, y = O.synthetic_inputs(N, shape, C, seed=H + N)
p64 = O.cast(post, torch.float64)
ref = O.loss_gradients(x.double(), y, p64, arch, act, S).reshape(N, -1)
def perr(g):
    g = g.cpu().double().reshape(N, -1)
    e = (g - ref).abs().max(1)[0] / ref.abs().max(1)[0]
    bad = (e > 1e-5).nonzero().flatten()
    return float(e.max()), bad.tolist()[:20], len(bad)
for order in ("fresh", "after_fwd", "after_logits"):
    eng = AttackEngine(StackedPosterior(arch, act, shape, C, H, post, "cuda:0"))
    if order != "fresh": eng.forward(x, S)
    if order == "after_logits": eng.forward(x, S, logits=True)
    g = eng.loss_gradients(x, y, S)
    print(order, perr(g))
    g2 = eng.loss_gradients(x, y, S)
    print(order, "again", perr(g2), "bit-equal", torch.equal(g, g2))
# which columns / structure of error for a bad point
eng = AttackEngine(StackedPosterior(arch, act, shape, C, H, post, "cuda:0"))
eng.forward(x, S); eng.forward(x, S, logits=True)
g = eng.loss_gradients(x, y, S).cpu().double().reshape(N, -1)
e = (g - ref).abs()
n = int(e.max(1)[0].argmax()); print("worst point", n, "err cols", (e[n] > 1e-8).nonzero().flatten().tolist()[:40], "count", int((e[n] > 1e-8).sum()))
