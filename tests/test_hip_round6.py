"""Round 6 (GPU): conv1 + pool on the f16 matrix pipe (conv1_pool_x3_kernel) against the fp32 VALU kernel it replaces in the triple mode; the
mean-logit loss tail against fp64 on saturated outputs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bnn_oracle as O                                     # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from robustbnns_amd import _hip
    _hip.load()


def _conv1_state(eng, x, S, N):
    """forward, then the pooled conv1 image and its stash exactly as the kernels left them: P1 [S, N, 32 * P1W^2] fp32, st1 the same shape in bytes"""
    eng.forward(x, S)
    ws, p = eng.workspace(N, S), eng.post
    per = 32 * p.P1W * p.P1W
    return ws["P1"].view(-1)[:S * N * per].view(S, N, per).clone().cpu(), ws["st1"].view(-1)[:S * N * per].view(S, N, per).clone().cpu()


@pytest.mark.parametrize("pw", ["0", "3"])
@pytest.mark.parametrize("act", ["leaky", "relu", "sigm", "tanh"])
@pytest.mark.parametrize("shape,N", [((1, 28, 28), 37), ((3, 32, 32), 21)])
def test_conv1_on_the_f16_pipe_equals_the_fp32_kernel(shape, N, act, pw, monkeypatch):
    """model_nn.py:98-100 (Conv2d(Cin, 32, 5) -> act -> MaxPool2d(2)): conv1_pool_x3_kernel — triple-split arithmetic on v_mfma_f32_16x16x32_f16, the
    pooling window's two columns as two shifted weight sets over one image fragment — writes the same P1 image and the same stash bytes (argmax of
    the 2x2 window, first maximum winning, | sign bit) as the fp32 packed-FMA kernel: values within 2e-6 of the point's largest, decisions equal
    except at numerical ties (where the two candidates' values agree to 1e-5).  Ragged point counts, several points per wave (pw), a sample subset."""
    from robustbnns_amd.conv import ConvEngine, ConvStackedPosterior
    Hc, Cn, S = 16, 10, 3
    Din = shape[0] * shape[1] * shape[2]
    q2 = ((shape[1] - 4) // 2) - 5
    post = O.synthetic_posterior("conv", Din, Hc, Cn, S, 0.05, in_ch=shape[0], head=q2 * q2 * Hc)
    x, _ = O.synthetic_inputs(N, shape, Cn, seed=11)
    x[3] = 0.0                                                          # an all-zero image (MNIST's background): exact ties in every window
    x[5] *= 1e-3                                                        # a dim image: the per-point scale
    sp = ConvStackedPosterior(act, shape, Cn, Hc, post, DEV)
    monkeypatch.setenv("RBNN_CONV1_X3_PW", pw)
    monkeypatch.setenv("RBNN_CONV1_X3", "0")
    p_ref, s_ref = _conv1_state(ConvEngine(sp, precision="triple"), x, S, N)
    p_exact, s_exact = _conv1_state(ConvEngine(sp, precision="exact"), x, S, N)
    assert torch.equal(p_ref, p_exact) and torch.equal(s_ref, s_exact)   # RBNN_CONV1_X3=0 IS the fp32 kernel
    monkeypatch.setenv("RBNN_CONV1_X3", "1")
    p_new, s_new = _conv1_state(ConvEngine(sp, precision="triple"), x, S, N)
    top = p_ref.abs().amax(2, keepdim=True).clamp_min(1e-30)
    err = ((p_new - p_ref).abs() / top).amax()
    diff = s_new != s_ref
    print(f"[conv1 x3 vs fp32 {shape} {act} pw={pw}] max |dP1| / max|P1| per point {float(err):.2e}; stash bytes differing {int(diff.sum())} of {diff.numel()}")
    assert float(err) < 2e-6
    assert int(diff.sum()) <= 2e-4 * diff.numel()
    # the all-zero image: every window is an exact tie — the first candidate wins in both kernels, and nothing is positive
    assert torch.equal(s_new[:, 3], s_ref[:, 3])
    # the per-sample conv1 output of the fp64 oracle (model.0 + act + pool) agrees with both
    w, b = post["model.0.weight"].double(), post["model.0.bias"].double()
    for s in range(S):
        o = torch.nn.functional.conv2d(x.double(), w[s], b[s])
        o = {"leaky": torch.nn.functional.leaky_relu, "relu": torch.relu, "sigm": torch.sigmoid, "tanh": torch.tanh}[act](o)
        ref = torch.nn.functional.max_pool2d(o, 2).reshape(N, -1)
        e = ((p_new[s].double() - ref).abs() / ref.abs().amax(1, keepdim=True).clamp_min(1e-30)).amax()
        assert float(e) < 2e-6, (s, float(e))
