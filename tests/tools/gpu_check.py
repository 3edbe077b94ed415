#!/usr/bin/env python3
"""Stage-by-stage check of the HIP path against the fp64 oracle + a first timing.  GPU box only.
usage: python tests/tools/gpu_check.py [--big]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import bnn_oracle as O
from robustbnns_amd import AttackEngine, StackedPosterior, _hip
from robustbnns_amd._hip import *

def rel(a, b):
    a = a.double().reshape(a.shape[0], -1); b = b.double().reshape(b.shape[0], -1)
    return float(((a - b).abs().max(1)[0] / b.abs().max(1)[0].clamp_min(1e-300)).max())

def check(arch, act, shape, C, H, S, N, std, seed=0):
    D = 1
    for v in shape: D *= v
    post = O.synthetic_posterior(arch, D, H, C, S, std)
    x, y = O.synthetic_inputs(N, shape, C, seed)
    lab = y.argmax(-1)
    sp = StackedPosterior(arch, act, shape, C, H, post, "cuda:0")
    eng = AttackEngine(sp)
    p64 = O.cast(post, torch.float64)
    xd = x.double()
    # stage 1: per-sample probabilities + mask
    Xp = eng.pad_inputs(x)
    ws = eng.workspace(N, S)
    eng.k.fc_forward(sp, Xp, None, S, OUT_PROBS, ws)
    torch.cuda.synchronize()
    P = ws["P"].view(S, N, 16)[:, :, :C].cpu()
    z64 = O.nn_logits(xd, p64, arch, act)
    P64 = torch.softmax(z64, -1)
    print(f"[{arch}/{act} D={D} H={H} C={C} S={S} N={N}] P rel err {rel(P.reshape(S*N,-1), P64.reshape(S*N,-1)):.2e}", flush=True)
    if arch == "fc" and act in ("relu", "leaky"):
        layers = O.mlp_layers(p64, arch)
        _, pre = O._mlp_forward_cache(xd.reshape(N, -1), layers, act)
        Hp = sp.Hp
        Npad = (N + 255) // 256 * 256
        m = ws["mask1"].view(S, Hp // 32, Npad)[:, :, :N].permute(0, 2, 1).contiguous().cpu()
        bits = ((m.unsqueeze(-1) >> torch.arange(32, dtype=torch.int32)) & 1).reshape(S, N, Hp)[:, :, :H].bool()
        ref = pre[0] > 0
        near = pre[0].abs() < 1e-6
        print(f"    mask mismatches (excluding |a|<1e-6): {int(((bits != ref) & ~near).sum())} of {bits.numel()}", flush=True)
    pm = eng.forward(x, S).cpu()
    print(f"    mean-prob rel err {rel(pm, P64.mean(0)):.2e}")
    for mode, name in ((LOSS_MEAN_PROB, "mean_prob"), (LOSS_PER_SAMPLE, "per_sample")):
        G = eng.gradient(eng.pad_inputs(x), to_lab(lab), None, S, mode)[:, :D].cpu().reshape(x.shape)
        ref = O._input_grad(xd, lab, p64, arch, act, name)
        print(f"    {name} input-gradient rel err {rel(G, ref):.2e}   max|g| {float(ref.abs().max()):.3e}", flush=True)
    return eng, x, y

def to_lab(lab):
    return lab.to("cuda:0", torch.int32)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    check("fc", "leaky", (1, 2, 1), 2, 64, 10, 100, 0.5)
    check("fc", "leaky", (1, 28, 28), 10, 32, 8, 8, 0.05)
    check("fc", "relu", (1, 28, 28), 10, 512, 5, 300, 0.05)
    check("fc", "leaky", (1, 28, 28), 10, 512, 7, 333, 0.05)
    check("fc", "sigm", (1, 28, 28), 10, 64, 3, 40, 0.05)
    check("fc", "tanh", (1, 28, 28), 10, 128, 3, 70, 0.05)
    check("fc", "leaky", (1, 28, 28), 10, 256, 3, 150, 0.05)
    check("fc", "leaky", (1, 28, 28), 10, 1024, 2, 100, 0.05)
    check("fc2", "leaky", (1, 28, 28), 10, 32, 4, 6, 0.08)
    check("fc2", "leaky", (1, 28, 28), 10, 512, 3, 200, 0.05)
    check("fc2", "tanh", (1, 2, 1), 2, 32, 6, 40, 0.4)
    if "--big" in sys.argv:
        D, H, C, S, N = 784, 512, 10, 100, 10000
        g = torch.Generator().manual_seed(0)
        post = {"model.1.weight": torch.randn(S, H, D, generator=g) * 0.05, "model.1.bias": torch.randn(S, H, generator=g) * 0.05,
                "model.3.weight": torch.randn(S, C, H, generator=g) * 0.05, "model.3.bias": torch.randn(S, C, generator=g) * 0.05}
        x, y = O.synthetic_inputs(N, (1, 28, 28), C, 0)
        sp = StackedPosterior("fc", "leaky", (1, 28, 28), C, H, post, "cuda:0")
        eng = AttackEngine(sp)
        xg = x.cuda()
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            adv = eng.fgsm(xg, y, S, 0.3)
            torch.cuda.synchronize(); dt = time.time() - t0
            print(f"FGSM C2 N={N} S={S}: {dt*1e3:.2f} ms -> {N*S/dt:.3e} attack-samples/s, {N*S*1626112/dt/1e12:.1f} TFLOP/s", flush=True)
        ws = eng.workspace(N, S)
        print("chunk", ws["chunk"], "n_slabs", ws["n_slabs"])
        Xp = eng.pad_inputs(xg); lab = to_lab(y.argmax(-1))
        for name, fn in (("fc_forward", lambda: eng.k.fc_forward(sp, Xp, None, S, OUT_PROBS, ws)),
                         ("fc_input_grad", lambda: eng.k.fc_input_grad(sp, None, S, N, ws["chunk"], ws))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"{name}: {ms:.3f} ms  -> {N*S*2*D*H/ms/1e9:.1f} TFLOP/s (algorithmic, this GEMM only)", flush=True)
