#!/usr/bin/env python3
"""Stage-by-stage check of the conv HIP path against the fp64 oracle.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import bnn_oracle as O
from robustbnns_amd import _hip
from robustbnns_amd.conv import ConvStackedPosterior, ConvEngine
from robustbnns_amd._hip import *

def rel(a, b):
    a = a.double().reshape(a.shape[0], -1); b = b.double().reshape(b.shape[0], -1)
    return float(((a - b).abs().max(1)[0] / b.abs().max(1)[0].clamp_min(1e-300)).max())

def check(act, C, Hc, S, N, std, seed=0, grad="--grad" in sys.argv):
    post = O.synthetic_posterior("conv", 784, Hc, C, S, std)
    x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed)
    sp = ConvStackedPosterior(act, (1, 28, 28), C, Hc, post, "cuda:0")
    eng = ConvEngine(sp)
    p64 = O.cast(post, torch.float64); xd = x.double()
    Xp = eng.pad_inputs(x); ws = eng.workspace(N, S)
    eng.k.conv_forward(sp, Xp, None, S, OUT_PROBS, ws); torch.cuda.synchronize()
    # stage checks
    P1ref, Q2ref = [], []
    for s in range(S):
        h = F.conv2d(xd, p64["model.0.weight"][s], p64["model.0.bias"][s]); h = F.max_pool2d(O._act(h, act), 2); P1ref.append(h)
        h = F.conv2d(h, p64["model.3.weight"][s], p64["model.3.bias"][s]); h = F.max_pool2d(O._act(h, act), 2, stride=1); Q2ref.append(h)
    P1ref = torch.stack(P1ref).reshape(S * N, -1); Q2ref = torch.stack(Q2ref).reshape(S * N, -1)
    P1 = ws["P1"].view(S * N, -1).cpu(); Q2 = ws["Q2"].view(S * N, -1).cpu()
    P = ws["P"].view(S, N, 16)[:, :, :C].cpu().reshape(S * N, C)
    Pref = torch.softmax(O.nn_logits(xd, p64, "conv", act), -1).reshape(S * N, C)
    print(f"[conv/{act} Hc={Hc} C={C} S={S} N={N}] P1 {rel(P1, P1ref):.2e}  Q2 {rel(Q2, Q2ref):.2e}  P {rel(P, Pref):.2e}", flush=True)
    pm = eng.forward(x, S).cpu()
    print(f"    mean-prob rel err {rel(pm, Pref.reshape(S, N, C).mean(0)):.2e}", flush=True)
    if grad:
        lab = y.argmax(-1)
        for mode, name in ((LOSS_MEAN_PROB, "mean_prob"), (LOSS_PER_SAMPLE, "per_sample")):
            G = eng.gradient(Xp, lab.to("cuda:0", torch.int32), None, S, mode)[:, :784].cpu().reshape(x.shape)
            ref = O._input_grad(xd, lab, p64, "conv", act, name)
            print(f"    {name} input-gradient rel err {rel(G, ref):.2e}   max|g| {float(ref.abs().max()):.3e}", flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    check("leaky", 10, 16, 2, 4, 0.05)
    check("relu", 10, 32, 2, 19, 0.05)
    check("leaky", 10, 64, 3, 33, 0.05)
    check("leaky", 10, 512, 2, 20, 0.03)
    check("leaky", 3, 272, 1, 5, 0.05)
