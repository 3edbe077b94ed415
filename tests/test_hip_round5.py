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


# ------------------------------------------------------------------ BASELINE config 4's step on one forward
@pytest.mark.parametrize("arch,Hn,Cn,S,N,act,precision", [
    ("fc", 512, 10, 23, 1000, "leaky", "triple"), ("fc", 128, 3, 5, 77, "relu", "triple"), ("fc2", 256, 10, 9, 300, "leaky", "triple"),
    ("fc", 128, 7, 12, 257, "tanh", "triple"), ("fc2", 128, 10, 6, 130, "sigm", "triple"),
    ("fc", 512, 10, 11, 600, "leaky", "exact"), ("fc2", 32, 10, 4, 50, "tanh", "exact"), ("fc", 16, 10, 3, 40, "leaky", "exact"),
    ("fc", 256, 10, 10, 513, "leaky", "split"), ("fc2", 128, 10, 5, 100, "relu", "split")])
def test_shared_forward_is_bit_identical_to_loss_gradients_then_fgsm(arch, Hn, Cn, S, N, act, precision):
    """AttackEngine.loss_gradients_and_fgsm: rbnn_fc_forward* once, then (per-sample loss -> backward GEMM -> 1/S sum) and (mean-probability
    loss -> backward GEMM -> sign step) from the same P and stash.  The backward kernels must leave the forward's state untouched — every
    architecture, activation kind (1-bit stash / stored derivative), precision mode: expected gradients and FGSM images equal, bit for bit,
    those of loss_gradients() followed by fgsm(); also with a sample-index call, and repeated (nothing stale in the workspace)."""
    from robustbnns_amd import AttackEngine, StackedPosterior
    post = O.synthetic_posterior(arch, 784, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=Hn + N)
    sp = StackedPosterior(arch, act, (1, 28, 28), Cn, Hn, post, DEV)
    eng = AttackEngine(sp, precision=precision)
    assert eng.precision == precision
    launches = []
    for fn in ("fc_forward", "fc_forward_triple", "fc_forward_split"):
        real = getattr(eng.k, fn)
        setattr(eng.k, fn, (lambda real: lambda *a, **kw: (launches.append("fwd"), real(*a, **kw))[1])(real))
    lg, adv = eng.loss_gradients(x, y, S).cpu(), eng.fgsm(x, y, S, 0.2).cpu()
    n_sep = len(launches)
    del launches[:]
    lg2, adv2 = eng.loss_gradients_and_fgsm(x, y, S, 0.2)
    assert 2 * len(launches) == n_sep
    assert torch.equal(lg2.cpu(), lg) and torch.equal(adv2.cpu(), adv)
    seeds = [S - 1, 0, S // 2]
    a, b = eng.loss_gradients_and_fgsm(x, y, 3, 0.1, seeds=seeds)
    assert torch.equal(a.cpu(), eng.loss_gradients(x, y, 3, seeds=seeds).cpu()) and torch.equal(b.cpu(), eng.fgsm(x, y, 3, 0.1, seeds=seeds).cpu())
    lg3, adv3 = eng.loss_gradients_and_fgsm(x, y, S, 0.2)
    assert torch.equal(lg3.cpu(), lg) and torch.equal(adv3.cpu(), adv)
    # sanity against the fp64 oracle (the separate calls are pinned elsewhere; this guards the comparison itself against "both empty")
    ref = O.loss_gradients(x[:8].double(), y[:8].double(), O.cast(post, torch.float64), arch, act, S)
    assert rel_err(lg2[:8].cpu(), ref.float()) < 1e-4


def test_caller_level_loss_gradients_and_fgsm_on_an_hmc_bnn(golden):
    """lossGradients.loss_gradients_and_fgsm(net, ...) on the reference's fixture: the reference's own loss_gradients and fgsm arrays."""
    from robustbnns_amd import lossGradients
    from robustbnns_amd.model_bnn import BNN
    g = golden("mnist_fc_h512_s8_n8_leaky"); m = g.meta
    bnn = BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], 0, tuple(m["shape"]), m["n_classes"])
    bnn.set_posterior_samples(g.posterior(), DEV)
    x, y = g.t("x"), g.t("y")
    grads, adv = lossGradients.loss_gradients_and_fgsm(bnn, x, y, DEV, m["S"], {"epsilon": m["eps"]})
    assert rel_err(grads.cpu(), g.t("loss_gradients")) < TOL
    ref_g = g.t("meanprob_grad").reshape(len(x), -1)
    safe = ref_g.abs() > TAU * ref_g.abs().max(dim=1, keepdim=True)[0]
    assert not (((adv.cpu().reshape(len(x), -1) - g.t("fgsm").reshape(len(x), -1)).abs() > 1e-6) & safe).any()


# ------------------------------------------------------------------ an FGSM eps grid as one resident job
def _hmc_bnn(g):
    from robustbnns_amd.model_bnn import BNN
    m = g.meta
    bnn = BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], 0, tuple(m["shape"]), m["n_classes"])
    bnn.set_posterior_samples(g.posterior(), DEV)
    return bnn


@pytest.mark.parametrize("name,dataset", [("trained_halfmoons_fc_h32_m10", "half_moons"), ("trained_halfmoons_fc2_h32_m10", "half_moons"),
                                          ("trained_mnistshaped_fc_h128_m5", "mnist"), ("trained_mnistshaped_fc2_h128_m3", "mnist"),
                                          ("trained_mnistshaped_conv_h16_m3", "mnist")])
def test_fgsm_eps_grid_is_one_gradient_and_one_clean_forward_per_n_samples(golden, name, dataset, tmp_path, monkeypatch):
    """build_eps_attacks_df (plot_eps_attacks.py:16-33) for FGSM on stored samples through adversarialAttacks.FgsmGrid: the gradient and the
    clean-set forward are computed once per n_samples and reused by every epsilon.  Every cell's adversarial set is the SAME TENSOR, bit for
    bit, as attack() returns for it (lowdim, triple and conv engines: the cached gradient is the sum the one-launch attack takes the sign
    of), the evaluation triple equals attack_evaluation()'s, and the rows equal the reference's DataFrame; the cost of both forms is printed."""
    import time
    from robustbnns_amd import adversarialAttacks as AA, plot_eps_attacks
    g = golden(name); m = g.meta; x, y = g.t("x"), g.t("y")
    bnn = _hmc_bnn(g)
    monkeypatch.chdir(tmp_path)
    eps_list, ns_list = m["eps_list"], m["ns_list"]
    grid = AA.FgsmGrid(bnn, x, y, dataset, DEV)
    assert grid.shared
    for eps in eps_list:
        for ns in ns_list:
            a = grid.attack(eps, ns, filename=bnn.name)
            b = AA.attack(net=bnn, x_test=x, y_test=y, dataset_name=dataset, device=DEV, method="fgsm", filename=bnn.name, n_samples=ns,
                          hyperparams={"epsilon": eps})
            assert torch.equal(a, b), f"eps={eps} ns={ns}: {int((a != b).sum())} pixels differ"
            ra = grid.evaluate(a, ns)
            rb = AA.attack_evaluation(net=bnn, x_test=x, x_attack=b, y_test=y, device=DEV, n_samples=ns)
            assert ra[:2] == rb[:2] and torch.equal(ra[2], rb[2])
    assert (grid.gradient_passes, grid.clean_forwards) == (len(ns_list), len(ns_list))
    # the driver: the reference's rows (where the fixture holds its DataFrame), and what the grid cost
    torch.cuda.synchronize(); t0 = time.perf_counter()
    df = plot_eps_attacks.build_eps_attacks_df(bnn=bnn, dataset=dataset, device=DEV, method="fgsm", x_test=x, y_test=y, epsilon_list=eps_list,
                                               n_samples_list=ns_list, savedir=bnn.name)
    torch.cuda.synchronize(); t_grid = time.perf_counter() - t0
    cost = df.attrs["grid_cost"]
    assert cost["gradient_passes"] == cost["clean_forwards"] == len(ns_list) and cost["cells"] == len(eps_list) * len(ns_list)
    if "fgsm_df_epsilon" in g.arr:
        for col in ("epsilon", "test_acc", "adv_acc", "n_samples"):
            assert np.array_equal(df[col].to_numpy().astype("float64"), g.arr["fgsm_df_" + col]), col
        assert np.abs(df["softmax_rob"].to_numpy() - g.arr["fgsm_df_softmax_rob"]).max() < TOL
    # what a cell costs without the PNG / pickle side effects (tens of ms of matplotlib each, the same in both forms): host clock around the
    # two loops, device synchronised at both ends
    monkeypatch.setattr(AA, "plot_save_grid_images", lambda **kw: None)
    monkeypatch.setattr(AA, "save_to_pickle", lambda **kw: None)

    def per_cell(form):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            grid2 = AA.FgsmGrid(bnn, x, y, dataset, DEV)
            for eps in eps_list:
                for ns in ns_list:
                    if form == "grid":
                        grid2.evaluate(grid2.attack(eps, ns, filename=bnn.name), ns)
                    else:
                        adv = AA.attack(net=bnn, x_test=x, y_test=y, dataset_name=dataset, device=DEV, method="fgsm", filename=bnn.name, n_samples=ns,
                                        hyperparams={"epsilon": eps})
                        AA.attack_evaluation(net=bnn, x_test=x, x_attack=adv, y_test=y, device=DEV, n_samples=ns)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / (3 * cost["cells"])

    per_cell("grid")
    t_g, t_c = per_cell("grid"), per_cell("cells")
    print(f"\n   {name}: {cost['cells']} cells — resident grid: {cost['gradient_passes']} gradient passes + {cost['clean_forwards']} clean forwards, "
          f"{t_g:.3f} ms per cell; per-cell loop: {cost['cells']} + {cost['cells']}, {t_c:.3f} ms per cell (host clock, side effects off; "
          f"the driver with them: {1e3 * t_grid / cost['cells']:.1f} ms per cell)")


# ------------------------------------------------------------------ a pending lazy draw and an index buffer that leaves its coverage
def test_lazy_draw_with_an_index_buffer_beyond_the_drawn_samples_materialises_first():
    """A lazy draw of the first 4 of 10 samples (seeded: 4 keys), then a forward that names sample 7: the fused launch would generate a sample
    the draw does not cover and read sample_keys[7] past the 4-element key tensor.  Such a call runs the draw for real and reads the
    stack — the same numbers as the non-lazy sequence; an index buffer inside the coverage still takes the fused launch (the stack stays
    untouched); the pending record owns a copy of the keys (an in-place edit of the caller's tensor does not change the draw)."""
    from robustbnns_amd import AttackEngine, StackedPosterior
    from robustbnns_amd.posterior import SviGuide
    shape, H, C, S, N = (1, 2, 1), 32, 2, 10, 50
    g = torch.Generator().manual_seed(5)
    names = {"model.1.weight": (H, 2), "model.1.bias": (H,), "model.3.weight": (C, H), "model.3.bias": (C,)}
    loc = {k: torch.randn(*v, generator=g) * 0.5 for k, v in names.items()}
    scl = {k: -2.0 + 0.3 * torch.randn(*v, generator=g) for k, v in names.items()}
    x, _ = O.synthetic_inputs(N, shape, C, seed=1)

    def run(lazy):
        keys = torch.tensor([11, 5, 7, 3], dtype=torch.int64, device=DEV)
        post = StackedPosterior.for_guide(SviGuide(loc, scl, "fc", DEV), "leaky", shape, C, S)
        eng = AttackEngine(post)
        post.redraw(1, 0)                                                   # all ten samples hold a first draw
        post.redraw(0, 2, n_samples=4, sample_keys=keys, lazy=lazy)         # the first four are redrawn (lazily)
        keys.fill_(99)                                                      # the caller reuses its tensor
        out = {"inside": eng.forward(x, 2, seeds=[3, 1]).cpu()}
        if lazy:
            assert post.__dict__["_lazy"] is not None                      # covered: generated inside the launch, still pending
        out["beyond"] = eng.forward(x, 2, seeds=[7, 0]).cpu()
        if lazy:
            assert post.__dict__["_lazy"] is None                          # not covered: the draw was run, the stack read
        out["W1"] = post.W1.clone().cpu()
        return out

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
