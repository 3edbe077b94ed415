"""GPU check of the split-half (f16x3) kernels against the exact-fp32 kernels and an fp64 torch reference.
usage: python tests/tools/gpu_check_split.py [N] [S] [H]"""
import math
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
G.build()
from oracle import bnn_oracle as O
from robustbnns_amd import AttackEngine, StackedPosterior
from robustbnns_amd import _hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
D, C = 784, 10
dev = "cuda:0"
post = O.synthetic_posterior("fc", D, H, C, S, 0.05)
x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed=0)
sp = StackedPosterior("fc", "leaky", (1, 28, 28), C, H, post, dev)
eng = AttackEngine(sp)
K = eng.k
Xp = eng.pad_inputs(x)
sidx, _ = eng.sample_index(S)
ws = eng.workspace(N, S)

def exp_for(t, target=14):
    m = float(t.abs().max())
    return target - math.ceil(math.log2(m)) if m > 0 else 0

ld = (D + 31) // 32 * 32
w1_exp = exp_for(sp.W1)
x_exp = exp_for(Xp)
W1s = torch.empty(sp.W1.shape[0] * H, ld * 2, dtype=torch.int16, device=dev)
K.split_rows(sp.W1, D, w1_exp, W1s, ld)
Xs = torch.empty(N, ld * 2, dtype=torch.int16, device=dev)
K.split_rows(Xp, D, x_exp, Xs, ld)
img = _hip.SplitImages()
Dp = sp.Dp
W1c = torch.empty(sp.W1.shape[0] * (H // 32) * 8 * Dp * 8, dtype=torch.int16, device=dev)
K.split_cols(sp.W1, H, D, w1_exp, W1c, Dp)
w2_exp = exp_for(sp.W2)
W2g = torch.empty(sp.W2.shape[0] * (H // 16) * 64 * 8, dtype=torch.int16, device=dev)
K.split_w2gen(sp.W2, C, H, w2_exp, W2g)
img.W1_rows = W1s.data_ptr(); img.W1_cols = W1c.data_ptr(); img.W2_gen = W2g.data_ptr()
img.ld_rows = ld; img.ld_cols = Dp; img.w1_exp = w1_exp; img.w2_exp = w2_exp
print("w1_exp", w1_exp, "x_exp", x_exp, "ld", ld)

# check the split image itself
v = (sp.W1[:, :, :D].reshape(-1, D) * 2.0 ** w1_exp)
hl = W1s.view(torch.float16).reshape(-1, ld // 8, 2, 8)
rec = (hl[:, :, 0].float() + hl[:, :, 1].float()).reshape(-1, ld)[:, :D]
print("split image rel err", float(((rec.double() - v.double()).abs().max()) / v.abs().max()))

K.fc_forward(sp, Xp, sidx, S, _hip.OUT_PROBS, ws)
P_exact = ws["P"].reshape(S, N, 16).clone(); m_exact = ws["mask1"].clone()
ws["P"].zero_(); ws["mask1"].zero_()
K.fc_forward_split(sp, img, Xs, ld, x_exp, N, sidx, S, _hip.OUT_PROBS, ws)
torch.cuda.synchronize()
P_split = ws["P"].reshape(S, N, 16).clone(); m_split = ws["mask1"].clone()
# fp64 reference on the GPU (check tool only)
W1 = sp.W1[:, :, :D].double(); Xd = Xp[:, :D].double()
errs_e, errs_s = 0.0, 0.0
for s in range(min(S, 8)):
    a = Xd @ W1[s].T + sp.b1[s].double()
    h = torch.where(a > 0, a, 0.01 * a)
    z = h @ sp.W2[s].double().T + sp.b2[s].double()
    p = torch.softmax(z, -1)
    errs_e = max(errs_e, float((P_exact[s, :, :C].double() - p).abs().max()))
    errs_s = max(errs_s, float((P_split[s, :, :C].double() - p).abs().max()))
print(f"max |P - P64|: exact {errs_e:.3e}  split {errs_s:.3e}   max|P_split - P_exact| {float((P_split - P_exact).abs().max()):.3e}")
diffbits = (m_exact ^ m_split)
nb = sum(int(((diffbits >> b) & 1).sum()) for b in range(32))
print("mask bits differing:", nb, "of", S * H * N)

def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
te = timeit(lambda: K.fc_forward(sp, Xp, sidx, S, _hip.OUT_PROBS, ws))
tsp = timeit(lambda: K.fc_forward_split(sp, img, Xs, ld, x_exp, N, sidx, S, _hip.OUT_PROBS, ws))
tx = timeit(lambda: K.split_rows(Xp, D, x_exp, Xs, ld))
fl = 2.0 * N * S * H * (D + 16)
print(f"forward exact {te:.3f} ms ({fl/te/1e9:.1f} TF)   split {tsp:.3f} ms ({fl/tsp/1e9:.1f} TF fp32-equivalent, {3*fl/tsp/1e9:.0f} TF f16)   split_rows(X) {tx:.3f} ms")

# ---------------- backward ----------------
lab = y.argmax(-1).to(dev, torch.int32)
K.fc_forward(sp, Xp, sidx, S, _hip.OUT_PROBS, ws)
K.reduce_samples(ws["P"], S, N, C, 1.0, ws["Psum"])
K.loss_dlogits(_hip.LOSS_MEAN_PROB, ws["P"], ws["Psum"], None, lab, S, 1.0 / S, N, C, ws["dZ"])
nsl = K.fc_input_grad(sp, sidx, S, N, ws["chunk"], ws)
Ge = ws["slabs"].reshape(nsl, N, Dp).sum(0)[:, :D].clone()
ssz = K.split_workspace_sizes(sp, img, N, S)
sws = {"dZ_gen": torch.empty(ssz["dZ_gen"] // 2, dtype=torch.int16, device=dev),
       "g_scale": torch.empty(ssz["g_scale"] // 4, dtype=torch.float32, device=dev), "X_split": Xs}
ws["slabs"].zero_()
nsl2 = K.fc_input_grad_split(sp, img, sidx, S, N, ws["chunk"], ws, sws)
torch.cuda.synchronize()
Gs = ws["slabs"].reshape(nsl2, N, Dp).sum(0)[:, :D].clone()
mx = Ge.abs().max(1)[0].clamp_min(1e-30)
rel = ((Gs - Ge).abs().max(1)[0] / mx)
print(f"grad split vs exact: max rel-to-point-max {float(rel.max()):.3e}  median {float(rel.median()):.3e}  slabs {nsl}/{nsl2}")
# fp64 backward GEMM from the same dZ and stash, first 256 points
M = 256
dZ = ws["dZ"].reshape(S, N, 16)[:, :M, :C].double()
mk = ws["mask1"].reshape(S, H // 32, -1)[:, :, :M]
bits = torch.stack([((mk >> b) & 1) for b in range(32)], 2).reshape(S, H, M).permute(0, 2, 1).double()   # [S, M, H]
act = bits + (1 - bits) * 0.01
g64 = torch.zeros(M, D, dtype=torch.float64, device=dev)
for s in range(S):
    g64 += ((dZ[s] @ sp.W2[s].double()) * act[s]) @ sp.W1[s, :, :D].double()
m64 = g64.abs().max(1)[0]
print(f"vs fp64 (first {M} points): exact {float(((Ge[:M].double() - g64).abs().max(1)[0] / m64).max()):.3e}"
      f"  split {float(((Gs[:M].double() - g64).abs().max(1)[0] / m64).max()):.3e}")
tge = timeit(lambda: K.fc_input_grad(sp, sidx, S, N, ws["chunk"], ws))
tgs = timeit(lambda: K.fc_input_grad_split(sp, img, sidx, S, N, ws["chunk"], ws, sws))
flg = 2.0 * N * S * H * (D + 16)
print(f"grad exact {tge:.3f} ms ({flg/tge/1e9:.1f} TF)   split {tgs:.3f} ms ({flg/tgs/1e9:.1f} TF fp32-equivalent)")
bad = (rel > 1e-4).nonzero().flatten()
print("bad points:", bad.numel(), bad[:40].tolist())
if bad.numel():
    n = int(bad[0]); dd = (Gs[n] - Ge[n]).abs() / mx[n]
    print("point", n, "bad cols", (dd > 1e-4).nonzero().flatten()[:40].tolist(), "gscale", float(sws["g_scale"][n]), "max|Ge|", float(mx[n]))
    print("hist n%256:", torch.bincount(bad % 256, minlength=256).nonzero().flatten()[:64].tolist())
    print("hist n//256:", torch.bincount(bad // 256).tolist())
bad = (rel > 1e-4).nonzero().flatten()
print("bad points:", bad.numel(), bad[:40].tolist())
