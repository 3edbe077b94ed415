"""GPU parity tests of round 4 (-m gpu), through the C-ABI:

  fused step tail    rbnn_step_tail_triple (sum over samples + loss + dZ generator image in one launch) and rbnn_attack_step_triple (step +
                     the new iterate's triple image) against the separate kernels they replace: BIT-IDENTICAL images, gradients and attacks
  conv SVI, 3x32x32  the in-place conv draw on BASELINE config 5's geometry (the bench runs exactly that)
  point-sharded      the zero-communication spelling of a multi-GPU job (`bench.py --shard points`): two processes, real kernels
"""
import os

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


# ------------------------------------------------------------------ the fused tail of a step
@pytest.mark.parametrize("arch,Hn,Cn,S,N,act", [("fc", 512, 10, 23, 1000, "leaky"), ("fc", 128, 3, 5, 77, "relu"), ("fc2", 256, 10, 9, 300, "leaky"),
                                              ("fc", 128, 7, 40, 257, "tanh")])
def test_fused_step_tail_is_bit_identical_to_the_separate_kernels(arch, Hn, Cn, S, N, act, monkeypatch):
    """One launch (rbnn_step_tail_triple) instead of rbnn_reduce_samples + rbnn_loss_dlogits + the dZ re-scaling of rbnn_fc_input_grad_triple:
    the generator image and the per-point scales it leaves are the same BITS for every loss, so are the gradients, the FGSM images, and — with
    rbnn_attack_step_triple writing each new iterate's triple image instead of a rbnn_triple_rows_grouped launch per iteration — the PGD
    images; ragged point counts (the image pads to 256 points), fewer than 10 classes, a sample-index call."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    post = O.synthetic_posterior(arch, 784, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=Hn + N)
    sp = StackedPosterior(arch, act, (1, 28, 28), Cn, Hn, post, DEV)
    lab = y.argmax(-1).int().to(DEV)
    seeds = [S - 1, 0, S // 2, 1]
    res = {}
    for tag in ("separate", "fused"):
        monkeypatch.setenv("RBNN_FUSED_TAIL", "0" if tag == "separate" else "1")
        eng = AttackEngine(sp, precision="triple")
        out = {}
        for name, mode in (("mean_prob", _hip.LOSS_MEAN_PROB), ("per_sample", _hip.LOSS_PER_SAMPLE), ("mean_logit", _hip.LOSS_MEAN_LOGIT)):
            out["G_" + name] = eng.gradient(eng.pad_inputs(x), lab, None, S, mode).clone()
            ws = eng.workspace(N, S)
            out["gen_" + name] = ws["triple"]["dZ_gen"].clone()
            out["gs_" + name] = ws["triple"]["g_scale"][:N].clone()
        out["fgsm"] = eng.fgsm(x, y, S, 0.2)
        out["fgsm_seeds"] = eng.fgsm(x, y, len(seeds), 0.1, seeds=seeds)
        out["pgd"] = eng.pgd(x, y, S, 0.1, iters=6)
        out["pgd_default"] = eng.pgd(x, y, S, 0.5, alpha=2 / 225, iters=4, mode=_hip.LOSS_MEAN_LOGIT)
        out["lossgrad"] = eng.loss_gradients(x, y, S)
        out["again"] = eng.pgd(x, y, S, 0.1, iters=6)
        res[tag] = {k: v.cpu() for k, v in out.items()}
    sep, fus = res["separate"], res["fused"]
    n_pad = (N + 255) // 256 * 256
    for k in sep:
        if k.startswith("gen_"):       # [S][n_pad][64 B]: compare the records of the N real points (rows beyond belong to no point)
            a, b = sep[k][:S * n_pad * 32].view(S, n_pad, 32)[:, :N], fus[k][:S * n_pad * 32].view(S, n_pad, 32)[:, :N]
            assert torch.equal(a, b), k
        else:
            assert torch.equal(sep[k], fus[k]), k
    assert torch.equal(fus["pgd"], fus["again"])
    # and against the fp64 oracle (mean-probability gradient, points away from activation kinks)
    p64 = O.cast(post, torch.float64)
    M = min(N, 96)
    ok = O.kink_margin(x[:M].double(), p64, arch, act, S) > 2e-6 if act in ("relu", "leaky") else torch.ones(M, dtype=torch.bool)
    ref = O.meanprob_gradients(x[:M].double(), y[:M].argmax(-1), p64, arch, act, S).reshape(M, -1)
    assert int(ok.sum()) >= 8 and rel_err(fus["G_mean_prob"][:M, :784][ok], ref[ok]) < TOL


def test_pgd_loop_launches_no_image_builder_after_the_first_iteration(monkeypatch):
    """Inside pgd() the forward of iteration k > 0 reads the image rbnn_attack_step_triple wrote: exactly ONE rbnn_triple_rows_grouped call per
    attack (counted on the binding), none of the separate tail kernels, and the flag does not leak into a later call on the same workspace."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    S, N, Hn, Cn = 6, 300, 128, 10
    post = O.synthetic_posterior("fc", 784, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=3)
    eng = AttackEngine(StackedPosterior("fc", "leaky", (1, 28, 28), Cn, Hn, post, DEV), precision="triple")
    calls = {"triple_rows": 0, "reduce_samples": 0, "loss_dlogits": 0, "attack_step": 0, "attack_step_triple": 0, "step_tail_triple": 0}
    for name in calls:
        real = getattr(eng.k, name)

        def spy(*a, _real=real, _name=name, **k):
            calls[_name] += 1
            return _real(*a, **k)
        monkeypatch.setattr(eng.k, name, spy)
    adv = eng.pgd(x, y, S, 0.1, iters=5).cpu()
    assert calls == {"triple_rows": 1, "reduce_samples": 0, "loss_dlogits": 0, "attack_step": 0, "attack_step_triple": 5, "step_tail_triple": 5}
    p1 = eng.forward(x, S).cpu()                                 # a later call on the same workspace builds its own image
    assert calls["triple_rows"] == 2
    fresh = AttackEngine(eng.post, precision="triple")
    assert torch.equal(p1, fresh.forward(x, S).cpu()) and torch.equal(adv, fresh.pgd(x, y, S, 0.1, iters=5).cpu())
    eng.fgsm(x, y, S, 0.1)                                       # FGSM: the plain step (its result's image is never read)
    assert calls["attack_step"] == 1 and calls["attack_step_triple"] == 5


# ------------------------------------------------------------------ conv weight images in one launch
@pytest.mark.parametrize("Hc,S", [(512, 3), (48, 2), (16, 2), (272, 1)])
def test_conv_weight_images_kernel_equals_the_standalone_builders(Hc, S, monkeypatch):
    """rbnn_conv_weight_images (forward grouped tap-major rows image + dense conv2^T image of model.3.weight from the fp32 stack, one launch:
    what every SVI redraw of a conv posterior runs) against the stand-alone builders (permuted copies + rbnn_triple_rows): the same bits —
    including channel counts that are not multiples of 32 (the dense image's zero-padded K step)."""
    from robustbnns_amd.conv import ConvStackedPosterior
    post = O.synthetic_posterior("conv", 784, Hc, 10, S, 0.05)
    out = {}
    for tag, env in (("standalone", "0"), ("fused", "1")):
        monkeypatch.setenv("RBNN_CONV_FUSED_IMAGES", env)
        sp = ConvStackedPosterior("leaky", (1, 28, 28), 10, Hc, post, DEV)
        rows, k2_exp, dense, _ = sp.triple_images()
        out[tag] = (rows.cpu(), dense.cpu(), k2_exp)
    assert out["fused"][2] == out["standalone"][2]
    assert torch.equal(out["fused"][0], out["standalone"][0]) and torch.equal(out["fused"][1], out["standalone"][1])


# ------------------------------------------------------------------ the SVI draw fused into the lowdim launch (BASELINE config 1)
@pytest.mark.parametrize("shape,H,C,S,N,act", [((1, 2, 1), 64, 2, 10, 100, "leaky"), ((1, 7, 1), 32, 10, 4, 37, "tanh"), ((1, 2, 1), 16, 2, 3, 500, "relu")])
def test_lazy_svi_draw_inside_the_lowdim_launch_equals_draw_then_run(shape, H, C, S, N, act):
    """redraw(lazy=True) only RECORDS (key, draw id); rbnn_lowdim_run_svi generates the same weights inside the pass (svi_draw_kernel's Philox
    counters): forward, expected gradients, FGSM, a PGD iteration and seeded draws are BIT-IDENTICAL to redraw() + rbnn_lowdim_run, the stack is
    untouched until somebody reads it — and reading it (attribute, state_dict, another engine) materialises exactly those weights."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    from robustbnns_amd.posterior import SviGuide
    D = shape[0] * shape[1] * shape[2]
    g = torch.Generator().manual_seed(H + N)
    names = {"model.1.weight": (H, D), "model.1.bias": (H,), "model.3.weight": (C, H), "model.3.bias": (C,)}
    loc = {k: torch.randn(*v, generator=g) * 0.5 for k, v in names.items()}
    scl = {k: -2.0 + 0.3 * torch.randn(*v, generator=g) for k, v in names.items()}
    x, y = O.synthetic_inputs(N, shape, C, seed=N)
    keys = torch.tensor([11, 5, 7, 3, 2, 9, 1, 8, 4, 6][:S], dtype=torch.int64, device=DEV)

    def run(lazy):
        post = StackedPosterior.for_guide(SviGuide(loc, scl, "fc", DEV), act, shape, C, S)
        eng = AttackEngine(post)
        assert eng.precision == "lowdim" and post.lazy_capable()
        out = {}
        post.redraw(0xABCDEF0123, 3, lazy=lazy)
        assert (post.__dict__["_lazy"] is not None) == lazy
        if lazy:
            assert float(post.__dict__["_t_W1"].abs().max()) == 0.0          # nothing has been written to the stack yet
        out["probs"] = eng.forward(x, S).cpu()
        out["lg"] = eng.loss_gradients(x, y, S).cpu()
        out["fgsm"] = eng.fgsm(x, y, S, 0.2).cpu()
        out["pgd1"] = eng.pgd_continue(x, x, y, S, 0.1).cpu()
        out["sub"] = eng.forward(x, 2, seeds=[S - 1, 0]).cpu()
        if lazy:
            assert float(post.__dict__["_t_W1"].abs().max()) == 0.0 and post.__dict__["_lazy"] is not None
        post.redraw(0, 1, sample_keys=keys, lazy=lazy)                       # seeded draws: one key per sample
        out["seeded"] = eng.forward(x, S).cpu()
        out["W1"] = post.W1.clone().cpu()                                    # reading the stack materialises the pending draw
        assert post.__dict__["_lazy"] is None
        out["sd"] = post.state_dict(S - 1)["model.3.weight"]
        out["exact"] = AttackEngine(post, precision="exact").forward(x, S).cpu()
        return out

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(a["W1"].abs().max()) > 0 and rel_err(a["exact"], a["seeded"]) < TOL


@pytest.mark.parametrize("arch", ["fc", "fc2"])
def test_images_only_svi_draw_for_triple_engines(arch):
    """redraw(lazy=True) on a posterior with triple images = rbnn_svi_draw_images: the images, biases and W2 of a full draw — bit for bit —
    while the fp32 W1 / Wm stack and its pack_rows4 copy (read by no triple kernel) stay untouched until somebody reads them; forward, gradients
    and a PGD attack equal the full draw's; an fp32-MFMA engine on the same posterior (which DOES read the stack) sees the materialised weights."""
    from robustbnns_amd import AttackEngine, StackedPosterior
    from robustbnns_amd.posterior import SviGuide
    D, H, C, S, N = 784, 128, 10, 5, 200
    g = torch.Generator().manual_seed(9)
    names = {"model.1.weight": (H, D), "model.1.bias": (H,)}
    names.update({"model.3.weight": (H, H), "model.3.bias": (H,), "model.5.weight": (C, H), "model.5.bias": (C,)} if arch == "fc2"
                 else {"model.3.weight": (C, H), "model.3.bias": (C,)})
    loc = {k: torch.randn(*v, generator=g) * 0.05 for k, v in names.items()}
    scl = {k: torch.full(v, -3.0) for k, v in names.items()}
    x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed=1)

    def run(lazy):
        post = StackedPosterior.for_guide(SviGuide(loc, scl, arch, DEV), "leaky", (1, 28, 28), C, S)
        eng = AttackEngine(post)
        assert eng.precision == "triple"
        post.triple_images()
        assert post.lazy_capable()
        post.redraw(0x1234, 2, lazy=lazy)
        out = {"img": [t.clone().cpu() for t in post._triple[1]], "b1": post.__dict__["_t_b1"].clone().cpu(), "W2": post.__dict__["_t_W2"].clone().cpu()}
        if lazy:
            assert float(post.__dict__["_t_W1"].abs().max()) == 0.0 and post.__dict__["_lazy"] is not None
        out["probs"] = eng.forward(x, S).cpu()
        out["lg"] = eng.loss_gradients(x, y, S).cpu()
        out["pgd"] = eng.pgd(x, y, S, 0.1, iters=3).cpu()
        if lazy:
            assert float(post.__dict__["_t_W1"].abs().max()) == 0.0          # three calls later the stack is still untouched
        out["exact"] = AttackEngine(post, precision="exact").forward(x, S).cpu()     # reads the stack: materialises
        assert post.__dict__["_lazy"] is None
        out["W1"], out["W1p"] = post.W1.clone().cpu(), post.W1p.clone().cpu()
        return out

    a, b = run(False), run(True)
    for k in ("b1", "W2", "probs", "lg", "pgd", "exact", "W1", "W1p"):
        assert torch.equal(a[k], b[k]), k
    assert all(torch.equal(p, q) for p, q in zip(a["img"], b["img"]))
    assert float(a["W1"].abs().max()) > 0 and rel_err(a["exact"], a["probs"]) < TOL


# ------------------------------------------------------------------ wide nets and replaced guide tensors
def test_svi_draw_covers_wide_nets_and_falls_back_beyond():
    """rbnn_svi_draw stages W2 [C, H] in dynamic LDS: hidden 2048 x 10 classes (80 KB) needed more than the 64 KB default and raised
    RBNN_ERR_SHAPE — the limit is now the CU's 160 KB (hidden 4096), rbnn_svi_draw_supported says so, and BNN falls back to
    rbnn_svi_materialize + a new stack beyond it instead of raising."""
    import ctypes as C
    from robustbnns_amd import _hip
    from robustbnns_amd.model_bnn import BNN, set_rng_seed
    lib = _hip.load()
    d = _hip.Posterior()
    d.arch, d.activation, d.in_features, d.in_stride, d.n_classes, d.n_stored = 0, 1, 784, 784, 10, 2
    for H, want in ((2048, 1), (4096, 1), (8192, 0)):
        d.hidden = H
        assert lib.rbnn_svi_draw_supported(C.byref(d), 0) == want
    d.hidden = 64
    assert lib.rbnn_svi_draw_supported(C.byref(d), 0) == 1 and lib.rbnn_svi_draw_supported(C.byref(d), 1) == 0      # triple images: hidden % 128
    x, _ = O.synthetic_inputs(8, (1, 28, 28), 10, seed=2)
    for H, in_place in ((2048, True), (8192, False)):
        g = torch.Generator().manual_seed(H)
        shapes = {"model.1.weight": (H, 784), "model.1.bias": (H,), "model.3.weight": (10, H), "model.3.bias": (10,)}
        loc = {k: torch.randn(*v, generator=g) * 0.02 for k, v in shapes.items()}
        scl = {k: torch.full(v, -4.0) for k, v in shapes.items()}
        bnn = BNN("mnist", H, "leaky", "fc", "svi", 5, 0.01, None, None, (1, 28, 28), 10)
        bnn.set_variational_params(loc, scl, DEV)
        assert bnn._in_place() == in_place
        set_rng_seed(3)
        p = bnn.forward(x, n_samples=2).cpu()
        assert torch.isfinite(p).all() and float((p.sum(-1) - 1).abs().max()) < 1e-5
        mean = O.bnn_forward(x.double(), {k: v.double().unsqueeze(0) for k, v in loc.items()}, "fc", "leaky", 1)
        assert float((p - mean.float()).abs().max()) < 0.2                  # two draws around the guide's mean (sigma = 0.018)


def test_replacing_a_guide_tensor_invalidates_the_cached_guide():
    """`net.svi_loc[k] = new_tensor` (version 0 again, another address) without set_variational_params() must drop the guide's
    bounds, its resident stacks and its seeded draws — they key on (data_ptr, _version) of every variational tensor."""
    from robustbnns_amd.model_bnn import BNN
    H = 128
    g = torch.Generator().manual_seed(1)
    shapes = {"model.1.weight": (H, 784), "model.1.bias": (H,), "model.3.weight": (10, H), "model.3.bias": (10,)}
    loc = {k: torch.randn(*v, generator=g) * 0.05 for k, v in shapes.items()}
    scl = {k: torch.full(v, -3.0) for k, v in shapes.items()}
    bnn = BNN("mnist", H, "leaky", "fc", "svi", 5, 0.01, None, None, (1, 28, 28), 10)
    bnn.set_variational_params(loc, scl, DEV)
    x, _ = O.synthetic_inputs(16, (1, 28, 28), 10, seed=4)
    p1 = bnn.forward(x, n_samples=3, seeds=[1, 2, 3]).cpu()
    guide1 = bnn._guide
    bnn.svi_loc["model.3.weight"] = (bnn.svi_loc["model.3.weight"] * 40.0).contiguous()       # a NEW tensor: 40x larger output weights
    p2 = bnn.forward(x, n_samples=3, seeds=[1, 2, 3]).cpu()
    assert bnn._guide is not guide1 and not torch.equal(p1, p2)
    assert float(bnn._guide.bound["W2"]) > 10 * float(guide1.bound["W2"])                       # the image scale follows the new bound
    assert torch.isfinite(p2).all()
