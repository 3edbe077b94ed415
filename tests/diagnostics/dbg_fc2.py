import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import bnn_oracle as O
from robustbnns_amd import AttackEngine, StackedPosterior
torch.manual_seed(0)
for H in (128, 512):
    S, N = 3, 300
    post = O.synthetic_posterior("fc2", 784, H, 10, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), 10, seed=1)
    sp = StackedPosterior("fc2", "leaky", (1, 28, 28), 10, H, post, "cuda:0")
    p64 = O.bnn_forward(x.double(), O.cast(post, torch.float64), "fc2", "leaky", S)
    pt = AttackEngine(sp, precision="triple").forward(x, S).cpu().double()
    pe = AttackEngine(sp, precision="exact").forward(x, S).cpu().double()
    et = ((pt - p64).abs().max(1)[0] / p64.abs().max(1)[0])
    ee = ((pe - p64).abs().max(1)[0] / p64.abs().max(1)[0])
    print(H, "triple max %.2e median %.2e | exact max %.2e" % (et.max(), et.median(), ee.max()), "worst rows", et.topk(5).indices.tolist(), "rows>1e-5:", int((et > 1e-5).sum()))
