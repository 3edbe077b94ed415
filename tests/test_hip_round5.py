"""GPU parity tests of round 5 (-m gpu), through the C-ABI:

  lowdim fc2 strides   rbnn_lowdim_run(ATTACK, iters > 1) with a padded X beside a compact out (the iterate lives in `out` with out's stride)
  shared forward       AttackEngine.loss_gradients_and_fgsm — BASELINE config 4's step on ONE forward — against the two separate calls: bit-identical
  eps grid             build_eps_attacks_df on a stored posterior: one gradient pass + one clean forward per n_samples, rows equal to the per-cell loop's
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import bnn_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("built_library")]
TOL, TAU, DEV = 1e-5, 1e-3, "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"


# ------------------------------------------------------------------ rbnn_lowdim_run: independent strides of X and out (fc2, several iterations)
@pytest.mark.parametrize("arch,Hn", [("fc2", 32), ("fc2", 128), ("fc", 64)])
def test_lowdim_attack_with_padded_inputs_and_compact_output(arch, Hn):
    """A C-ABI caller may hand X / X0 with a padded row stride (ldx = 16) and a compact `out` (ldo = D).  From iteration 1 on the iterate is read
    back from `out`: with out's stride, not X's (round 4 read it with ldx — wrong iterates and reads past `out`; the Python wrapper always
    passes equal strides, so only a direct call sees it)."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    S, N, D, iters = 6, 37, 2, 5
    post = O.synthetic_posterior(arch, D, Hn, 2, S, 0.5)
    x, y = O.synthetic_inputs(N, (1, 2, 1), 2, seed=3)
    sp = StackedPosterior(arch, "leaky", (1, 2, 1), 2, Hn, post, DEV)
    eng = AttackEngine(sp)
    assert eng.precision == "lowdim"
    lab = y.argmax(-1).int().to(DEV)
    Xc = x.reshape(N, D).contiguous().to(DEV)
    ref = eng._lowdim_attack(Xc, Xc, lab, None, S, _hip.LOSS_MEAN_PROB, None, 0.0, True, 0.2, True, iters)
    Xpad = torch.full((N, 16), float("nan"), device=DEV)
    Xpad[:, :D] = Xc
    out = torch.full((N * D + 64,), -7.0, device=DEV)                       # compact rows + a guard band behind them
    scratch = eng._low_scratch(N, S)
    lib = eng.k.lib
    rc = lib.rbnn_lowdim_run(C.byref(sp.descriptor()), eng.k.LOWDIM_ATTACK, _hip.LOSS_MEAN_PROB, 0, _hip.ptr(Xpad), _hip.ptr(Xpad), 16, N, None, S,
                             _hip.ptr(lab), 1.0 / S, 1.0, 0.2, None, 0.0, 1, 1, iters, _hip.ptr(scratch), _hip.ptr(out), D, None, None,
                             _hip.stream_of(Xpad))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out[:N * D].view(N, D), ref)
    assert bool((out[N * D:] == -7.0).all())                                # nothing written (or, with the old stride, read-modify-written) past the rows


# ------------------------------------------------------------------ the carried PGD image belongs to ONE tensor
def test_forward_on_other_inputs_inside_a_pgd_loop_does_not_take_the_iterates_image():
    """Inside pgd() every step writes the next iterate's triple image and the next forward on that workspace skips the image builder.  A
    callback that runs a forward on OTHER inputs of the same (N, S) must get those inputs' image (round 4 keyed the hand-over on a boolean:
    the foreign forward silently used the iterate's image — ADVICE r4), and the attack must come out as without the callback."""
    from robustbnns_amd import AttackEngine, StackedPosterior
    S, N = 7, 300
    post = O.synthetic_posterior("fc", 784, 128, 10, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), 10, seed=11)
    other, _ = O.synthetic_inputs(N, (1, 28, 28), 10, seed=12)
    sp = StackedPosterior("fc", "leaky", (1, 28, 28), 10, 128, post, DEV)
    eng = AttackEngine(sp, precision="triple")
    want_other = eng.forward(other.to(DEV), S).clone()
    plain = eng.pgd(x, y, S, 0.1, iters=5)
    seen = []
    with_cb = eng.pgd(x, y, S, 0.1, iters=5, before_step=lambda: seen.append(eng.forward(other.to(DEV), S).clone()))
    assert len(seen) == 4 and all(torch.equal(s, want_other) for s in seen)
    assert torch.equal(with_cb, plain)
