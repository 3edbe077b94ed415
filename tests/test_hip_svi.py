"""GPU tests (-m gpu) of the SVI path: the in-place draw kernel rbnn_svi_draw against the oracle's restatement of
its generator (Philox4x32-10, pinned by Random123's known-answer vectors in tests/test_oracle_svi.py, + Box-Muller), every image it
writes against the stand-alone image builders, and the host side: one resident posterior / engine / workspace redrawn in place, no
device->host synchronisation inside an SVI PGD attack.  PARITY UNPINNED against pyro-ppl 1.3.0's own RNG stream."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import bnn_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("built_library")]
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from robustbnns_amd import _hip
    _hip.load()


def guide_tensors(arch, D, H, C, seed, std=0.05, scale=-3.0):
    g = torch.Generator().manual_seed(seed)
    shapes = {"model.1.weight": (H, D), "model.1.bias": (H,)}
    if arch == "fc2":
        shapes.update({"model.3.weight": (H, H), "model.3.bias": (H,), "model.5.weight": (C, H), "model.5.bias": (C,)})
    else:
        shapes.update({"model.3.weight": (C, H), "model.3.bias": (C,)})
    loc = {k: torch.randn(*s, generator=g) * std for k, s in shapes.items()}
    scl = {k: scale + 0.5 * torch.randn(*s, generator=g) for k, s in shapes.items()}
    return loc, scl


def by_role(d, arch):
    keys = ("model.1", "model.3", "model.5") if arch == "fc2" else ("model.1", "model.3")
    out = {"W1": d[keys[0] + ".weight"], "b1": d[keys[0] + ".bias"], "W2": d[keys[-1] + ".weight"], "b2": d[keys[-1] + ".bias"]}
    if arch == "fc2":
        out["Wm"], out["bm"] = d[keys[1] + ".weight"], d[keys[1] + ".bias"]
    return out


CASES = [("fc", (1, 28, 28), 128, 10, 3), ("fc", (1, 28, 28), 512, 10, 2), ("fc2", (1, 28, 28), 128, 10, 2), ("fc", (1, 2, 1), 64, 2, 4),
         ("fc2", (1, 2, 1), 32, 2, 3), ("fc", (1, 28, 28), 16, 10, 2), ("fc", (1, 5, 3), 32, 7, 3)]


@pytest.mark.parametrize("arch,shape,H,C,S", CASES)
def test_draw_kernel_matches_the_oracle_generator_and_the_image_builders(arch, shape, H, C, S):
    from robustbnns_amd import _hip
    from robustbnns_amd.posterior import StackedPosterior, SviGuide
    D = int(np.prod(shape))
    loc, scl = guide_tensors(arch, D, H, C, seed=H + D)
    guide = SviGuide(loc, scl, arch, DEV)
    post = StackedPosterior.for_guide(guide, "leaky", shape, C, S)
    tri = post.triple_supported()
    assert tri == (max(32, H) % 128 == 0)
    if tri:
        post.triple_images()
    key, draw = 0x1234567887654321, 7
    post.redraw(key, draw)
    torch.cuda.synchronize()
    W64, E = O.svi_draw_philox(by_role(loc, arch), by_role(scl, arch), key, draw, S)
    assert max(float(e.abs().max()) for e in E.values()) <= _hip.SVI_EPS_MAX
    Hp = post.Hp
    got = {"W1": post.W1[:, :H, :D], "b1": post.b1[:, :H], "W2": post.W2[:, :, :H], "b2": post.b2}
    if arch == "fc2":
        got.update(Wm=post.Wm[:, :H, :H], bm=post.bm[:, :H])
    for name, w in got.items():
        sp = torch.nn.functional.softplus(by_role(scl, arch)[name].double())
        tol = 2e-5 * sp.unsqueeze(0) * (1 + E[name].abs()) + 1e-7 * W64[name].abs()     # v_log_f32 / v_sin_f32 / v_cos_f32 vs numpy's double precision
        assert bool(((w.cpu().double() - W64[name]).abs() <= tol).all()), name
    # padding stays zero (hidden 16 -> 32 rows, D -> D_pad columns)
    assert float(post.W1[:, H:, :].abs().max() if Hp > H else 0.0) == 0.0 and float(post.W1[:, :, D:].abs().max() if post.Dp > D else 0.0) == 0.0
    assert float(post.W2[:, :, H:].abs().max() if Hp > H else 0.0) == 0.0
    # every image the kernel wrote == the stand-alone builder applied to the fp32 stack it wrote, bit for bit
    k = _hip.HipKernels()
    ref = torch.empty_like(post.W1)
    k.pack_rows4(post.W1, ref)
    assert torch.equal(ref, post.W1p)
    if arch == "fc2":
        k.pack_rows4(post.Wm, ref := torch.empty_like(post.Wm))
        assert torch.equal(ref, post.Wmp)
    if tri:
        img, keep = post._triple
        chk = [torch.empty_like(t) for t in keep]
        k.triple_rows(post.W1, D, img.w1_exp, chk[0], img.ld_rows, grouped=True)
        k.triple_cols(post.W1, Hp, D, img.w1_exp, chk[1], post.Dp)
        k.triple_w2gen(post.W2, C, Hp, img.w2_exp, chk[2])
        if arch == "fc2":
            k.triple_rows(post.Wm, Hp, img.wm_exp, chk[3], Hp, grouped=True)
            k.triple_cols(post.Wm, Hp, Hp, img.wm_exp, chk[4], Hp)
        for i, (a, b) in enumerate(zip(chk, keep)):
            assert torch.equal(a, b), f"triple image {i}"
        # the fixed per-guide scale is a true bound: the largest drawn weight times 2^exp stays within fp16's exact window
        assert float(post.W1.abs().max()) * 2.0 ** img.w1_exp <= 2.0 ** 14


def test_seeded_draws_do_not_depend_on_position_and_draw_ids_differ():
    from robustbnns_amd.posterior import StackedPosterior, SviGuide
    loc, scl = guide_tensors("fc", 784, 128, 10, seed=3)
    guide = SviGuide(loc, scl, "fc", DEV)
    a = StackedPosterior.for_guide(guide, "leaky", (1, 28, 28), 10, 3)
    b = StackedPosterior.for_guide(guide, "leaky", (1, 28, 28), 10, 2)
    a.redraw(0, 0, sample_keys=torch.tensor([5, 9, 2], dtype=torch.int64, device=DEV))
    b.redraw(0, 0, sample_keys=torch.tensor([2, 5], dtype=torch.int64, device=DEV))
    assert torch.equal(a.W1[0], b.W1[1]) and torch.equal(a.W1[2], b.W1[0]) and torch.equal(a.W2[2], b.W2[0]) and torch.equal(a.b1[0], b.b1[1])
    W64, _ = O.svi_draw_philox(by_role(loc, "fc"), by_role(scl, "fc"), None, 0, 3, sample_keys=[5, 9, 2])
    assert float((a.W1[:, :, :784].cpu().double() - W64["W1"]).abs().max()) < 5e-5
    w0 = a.W1.clone()
    a.redraw(77, 1)
    w1 = a.W1.clone()
    a.redraw(77, 2)
    assert not torch.equal(w0, w1) and not torch.equal(w1, a.W1)
    a.redraw(77, 1)
    assert torch.equal(w1, a.W1)                                   # a pure function of (key, draw id)
    # the samples of one draw differ from each other and have the guide's moments
    loc1, sp1 = loc["model.1.weight"], torch.nn.functional.softplus(scl["model.1.weight"])
    big = StackedPosterior.for_guide(guide, "leaky", (1, 28, 28), 10, 64)
    big.redraw(1, 1)
    z = ((big.W1[:, :, :784].cpu() - loc1) / sp1)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1) < 5e-3 and abs(float((z ** 3).mean())) < 2e-2 and abs(float((z ** 4).mean()) - 3) < 5e-2
    assert abs(float((z[0] * z[1]).mean())) < 1e-2                 # independent across samples


def make_svi_bnn(arch, H, shape=(1, 28, 28), C=10, seed=11):
    from robustbnns_amd.model_bnn import BNN
    D = int(np.prod(shape))
    loc, scl = guide_tensors(arch, D, H, C, seed)
    bnn = BNN("mnist" if D == 784 else "half_moons", H, "leaky", arch, "svi", 5, 0.01, None, None, shape, C)
    bnn.set_variational_params(loc, scl, DEV)
    return bnn, loc, scl


@pytest.mark.parametrize("arch,H", [("fc", 512), ("fc2", 128), ("fc", 64)])
def test_bnn_redraws_in_place_and_images_agree_with_the_fp32_stack(arch, H):
    """BNN.forward on an SVI net: the same posterior / engine object call after call, fresh weights every call, reproducible under
    set_rng_seed; and the result of the default (auto) engine equals the fp32-MFMA engine run on the very weights that were drawn."""
    from robustbnns_amd.factory import make_engine, posterior_from_stacked
    from robustbnns_amd.model_bnn import set_rng_seed
    bnn, loc, scl = make_svi_bnn(arch, H)
    x, _ = O.synthetic_inputs(24, (1, 28, 28), 10, seed=5)
    set_rng_seed(0)
    p1 = bnn.forward(x, n_samples=6).cpu()
    post, eng = bnn._slots[6]
    ptr = post.W1.data_ptr()
    w_first = post.W1.clone()
    p2 = bnn.forward(x, n_samples=6).cpu()
    assert bnn._slots[6][0] is post and bnn._slots[6][1] is eng and post.W1.data_ptr() == ptr and len(eng._ws_cache) == 1
    assert not torch.equal(w_first, post.W1) and not torch.equal(p1, p2)
    set_rng_seed(0)
    assert torch.equal(bnn.forward(x, n_samples=6).cpu(), p1) and torch.equal(post.W1, w_first)
    assert eng.precision == ("triple" if H % 128 == 0 else "exact")
    # the drawn stack, read back as stored samples, through the fp32-MFMA kernels and through the fp64 oracle
    stacked = {k: torch.stack([post.state_dict(i)[k] for i in range(6)]) for k in post.state_dict(0)}
    ex = make_engine(posterior_from_stacked(arch, "leaky", (1, 28, 28), 10, H, stacked, DEV), precision="exact")
    assert rel_err(ex.forward(x, 6).cpu(), p1) < 1e-5
    p64 = O.bnn_forward(x.double(), O.cast(stacked, torch.float64), arch, "leaky", 6)
    assert rel_err(p1, p64) < 1e-5
    # seeded forwards: reproducible, a prefix property, and independent of the unseeded stream
    q1 = bnn.forward(x, n_samples=3, seeds=[4, 1, 7]).cpu()
    bnn.forward(x, n_samples=6)
    assert torch.equal(bnn.forward(x, n_samples=3, seeds=[4, 1, 7]).cpu(), q1)
    q2 = bnn.forward(x, n_samples=1, seeds=[1]).cpu()
    g1 = bnn.hot_path(1, seeds=[1])[0].post
    assert torch.equal(g1.W1[0], bnn.hot_path(3, seeds=[4, 1, 7])[0].post.W1[1]) and q2.shape == (24, 10)
    # a reloaded guide is never served an older guide's draw
    loc2, scl2 = guide_tensors(arch, 784, H, 10, seed=99)
    bnn.set_variational_params(loc2, scl2, DEV)
    assert bnn._drawn is None and bnn._slots == {}
    assert not torch.equal(bnn.forward(x, n_samples=3, seeds=[4, 1, 7]).cpu(), q1)


def test_svi_pgd_iteration_has_no_device_to_host_sync():
    """An SVI PGD attack redraws before every iteration (adversarialAttacks.py:95-97 -> model_bnn.py:230-232): with the resident stack
    that is one extra launch per iteration — no new posterior / engine / workspace, and NO synchronising call (torch's sync debug mode
    raises on any)."""
    from robustbnns_amd import adversarialAttacks as AA
    bnn, _, _ = make_svi_bnn("fc", 512)
    x, y = O.synthetic_inputs(256, (1, 28, 28), 10, seed=6)
    xd, lab = x.to(DEV), y.argmax(-1).to(DEV)
    hp = {"epsilon": 0.1, "iters": 3}
    adv0 = AA.pgd_attack(bnn, xd, lab, hp, n_samples=5)                       # warm-up: allocates the stack, the engine, the workspace
    post, eng = bnn._slots[5]
    n_ws, ptr, draws = len(eng._ws_cache), post.W1.data_ptr(), bnn._draws
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        adv = AA.pgd_attack(bnn, xd, lab, hp, n_samples=5)
        adv2 = AA.fgsm_attack(bnn, xd.clone(), lab, {"epsilon": 0.1}, n_samples=5)
        p = bnn.forward(xd, n_samples=5)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert bnn._slots[5][0] is post and post.W1.data_ptr() == ptr and len(eng._ws_cache) == n_ws
    assert bnn._draws == draws + 3 + 1 + 1                                    # one draw per PGD iteration, one per FGSM, one per forward
    assert adv.shape == x.shape and float((adv - xd).abs().max()) <= 0.1 + 1e-6 and float(adv.min()) >= 0 and float(adv.max()) <= 1
    assert adv2.shape == x.shape and p.shape == (256, 10) and not torch.equal(adv, adv0)


def test_svi_attack_and_evaluation_through_the_call_surface(tmp_path, monkeypatch):
    """attack() / attack_evaluation() on an SVI net (fc2, the reference's saved model_5 shape at a small hidden size)."""
    from robustbnns_amd import adversarialAttacks as AA
    monkeypatch.chdir(tmp_path)
    bnn, _, _ = make_svi_bnn("fc2", 128)
    x, y = O.synthetic_inputs(64, (1, 28, 28), 10, seed=8)
    for method in ("fgsm", "pgd"):
        adv = AA.attack(net=bnn, x_test=x, y_test=y, dataset_name="mnist", device=DEV, method=method, filename=bnn.name, n_samples=4,
                        hyperparams={"epsilon": 0.2, "iters": 4})
        assert adv.shape == x.shape and float((adv.cpu() - x).abs().max()) <= 0.2 + 1e-6
        oa, aa, rob = AA.attack_evaluation(net=bnn, x_test=x, x_attack=adv, y_test=y, device=DEV, n_samples=4)
        assert 0 <= aa <= 100 and 0 <= oa <= 100 and rob.shape == (64,) and float(rob.min()) >= 0 and float(rob.max()) <= 1
    assert len(bnn._slots) == 1


# ------------------------------------------------------------------------------------------------ conv SVI nets (the reference's saved model_0/2/4/6/8)
def conv_guide_tensors(Cin, Hc, C, q2, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = {"model.0.weight": (32, Cin, 5, 5), "model.0.bias": (32,), "model.3.weight": (Hc, 32, 5, 5), "model.3.bias": (Hc,),
              "model.7.weight": (C, q2 * q2 * Hc), "model.7.bias": (C,)}
    loc = {k: torch.randn(*s, generator=g) * 0.05 for k, s in shapes.items()}
    scl = {k: -3.0 + 0.5 * torch.randn(*s, generator=g) for k, s in shapes.items()}
    return loc, scl


@pytest.mark.parametrize("act,Hc,shape", [("leaky", 32, (1, 28, 28)), ("tanh", 16, (1, 28, 28)), ("leaky", 64, (3, 32, 32))])
def test_conv_svi_redraws_in_place(act, Hc, shape, tmp_path, monkeypatch):
    """(3, 32, 32): BASELINE config 5 = "CIFAR-10 conv-BNN, SVI" — the geometry bench.py --workload c5 redraws in place every PGD iteration."""
    from robustbnns_amd import adversarialAttacks as AA
    from robustbnns_amd.conv import ConvSviGuide, ConvStackedPosterior
    from robustbnns_amd.model_bnn import BNN, set_rng_seed
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("RBNN_CIFAR_CONV", "1")
    C, S = 10, 3
    q2 = (shape[1] - 4) // 2 - 5
    dataset = "mnist" if shape[0] == 1 else "cifar"
    loc, scl = conv_guide_tensors(shape[0], Hc, C, q2, seed=Hc)
    # the flat draw against the oracle's generator, tensor by tensor (element e = component e % 4 of block e / 4)
    post = ConvStackedPosterior.for_guide(ConvSviGuide(loc, scl, DEV), act, shape, C, Hc, S)
    post.triple_images()
    post.redraw(0xFEEDFACE12345678, 5)
    got = post.state_dict
    for k, tid in ConvSviGuide.TENSOR_IDS.items():
        n = loc[k].numel()
        sp = torch.nn.functional.softplus(scl[k].double()).reshape(-1)
        for s in range(S):
            eps = torch.from_numpy(O.philox_normals(0xFEEDFACE12345678, 5, tid, s, 1, n)).reshape(-1)
            want = loc[k].double().reshape(-1) + sp * eps
            assert float(((got(s)[k].double().reshape(-1) - want).abs() / (sp * (1 + eps.abs()))).max()) < 2e-5, (k, s)
    # the derived images follow the drawn stack: the default engine (triple conv2) agrees with the fp32-MFMA engine on stored copies
    from robustbnns_amd.factory import make_engine, posterior_from_stacked
    x, y = O.synthetic_inputs(8, shape, C, seed=4)
    stacked = {k: torch.stack([post.state_dict(i)[k] for i in range(S)]) for k in ConvSviGuide.TENSOR_IDS}
    ex = make_engine(posterior_from_stacked("conv", act, shape, C, Hc, stacked, DEV), precision="exact")
    tri = make_engine(post)
    assert tri.precision == "triple"
    assert rel_err(tri.forward(x, S).cpu(), ex.forward(x, S).cpu()) < 1e-5
    assert rel_err(tri.forward(x, S).cpu(), O.bnn_forward(x.double(), O.cast(stacked, torch.float64), "conv", act, S)) < 1e-5
    # through the call surface: one resident stack, redrawn per call / per PGD iteration, no sync inside the attack
    bnn = BNN(dataset, Hc, act, "conv", "svi", 5, 0.01, None, None, shape, C)
    bnn.set_variational_params(loc, scl, DEV)
    set_rng_seed(0)
    p1 = bnn.forward(x, n_samples=S).cpu()
    slot = bnn._slots[S]
    p2 = bnn.forward(x, n_samples=S).cpu()
    set_rng_seed(0)
    assert bnn._slots[S] is slot and not torch.equal(p1, p2) and torch.equal(bnn.forward(x, n_samples=S).cpu(), p1)
    xd, lab = x.to(DEV), y.argmax(-1).to(DEV)
    AA.pgd_attack(bnn, xd, lab, {"epsilon": 0.1, "iters": 2}, n_samples=S)
    draws = bnn._draws
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        adv = AA.pgd_attack(bnn, xd, lab, {"epsilon": 0.1, "iters": 3}, n_samples=S)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bnn._draws == draws + 3 and bnn._slots[S] is slot
    assert float((adv - xd).abs().max()) <= 0.1 + 1e-6
    q = bnn.forward(x, n_samples=2, seeds=[7, 3]).cpu()
    assert torch.equal(q, bnn.forward(x, n_samples=2, seeds=[7, 3]).cpu())


def test_prefetched_draws_equal_sequential_draws(monkeypatch):
    """StackedPosterior.prefetch / flip (the next draw on a side stream into a second buffer set, under the current iteration's kernels)
    gives bit for bit what redraw() between the iterations gives for the same keys; an SVI PGD attack through the call surface uses it."""
    from robustbnns_amd import adversarialAttacks as AA
    from robustbnns_amd.engine import AttackEngine
    from robustbnns_amd.model_bnn import set_rng_seed
    from robustbnns_amd.posterior import StackedPosterior, SviGuide
    loc, scl = guide_tensors("fc", 784, 512, 10, seed=21)
    x, y = O.synthetic_inputs(300, (1, 28, 28), 10, seed=3)
    keys = [11, 22, 33, 44]

    def run(prefetch):
        post = StackedPosterior.for_guide(SviGuide(loc, scl, "fc", DEV), "leaky", (1, 28, 28), 10, 6)
        eng = AttackEngine(post)
        post.triple_images()
        post.redraw(keys[0], 0)
        it = iter(keys[1:])
        if prefetch:
            post.prefetch(next(it))

            def before():
                post.flip()
                k = next(it, None)
                if k is not None:
                    post.prefetch(k)
        else:
            before = lambda: post.redraw(next(it), 0)
        adv = eng.pgd(x, y, 6, 0.1, iters=4, before_step=before).cpu()
        return adv, post.W1.clone(), eng.forward(x, 6).cpu()

    a1, w1, p1 = run(False)
    a2, w2, p2 = run(True)
    assert torch.equal(a1, a2) and torch.equal(w1, w2) and torch.equal(p1, p2)
    # the call surface (opt-in: RBNN_SVI_PREFETCH=1): same generator state -> same attack, prefetched or not
    monkeypatch.setenv("RBNN_SVI_PREFETCH", "1")
    bnn, _, _ = make_svi_bnn("fc", 512)
    xd, lab = x.to(DEV), y.argmax(-1).to(DEV)
    set_rng_seed(5)
    adv_a = AA.pgd_attack(bnn, xd, lab, {"epsilon": 0.1, "iters": 4}, n_samples=5)
    post = bnn._slots[5][0]
    assert post._back is not None and not post._prefetched
    set_rng_seed(5)
    adv_b = AA.pgd_attack(bnn, xd, lab, {"epsilon": 0.1, "iters": 4}, n_samples=5)
    assert torch.equal(adv_a, adv_b)
    set_rng_seed(5)
    monkeypatch.setenv("RBNN_SVI_PREFETCH", "0")                        # the default, sequential path: redraw between the iterations
    adv_c = AA.pgd_attack(bnn, xd, lab, {"epsilon": 0.1, "iters": 4}, n_samples=5)
    assert torch.equal(adv_a, adv_c)


def test_conv_redraw_rebuilds_unread_images_on_demand(monkeypatch):
    """redraw() only marks stale what no kernel of the running configuration reads (the fp32 kernels' input-channel regrouping): whoever needs
    it afterwards gets it rebuilt from the CURRENT draw."""
    from robustbnns_amd.conv import ConvSviGuide, ConvStackedPosterior
    from robustbnns_amd.factory import make_engine, posterior_from_stacked
    C, S, Hc, act = 10, 3, 32, "leaky"
    loc, scl = conv_guide_tensors(1, Hc, C, 7, seed=77)
    post = ConvStackedPosterior.for_guide(ConvSviGuide(loc, scl, DEV), act, (1, 28, 28), C, Hc, S)
    post.triple_images()
    x, y = O.synthetic_inputs(8, (1, 28, 28), C, seed=5)
    tri = make_engine(post)
    assert tri.precision == "triple"
    post.redraw(0x1234, 0)
    tri.loss_gradients(x, y, S)                                   # the triple kernels: the fp32 regrouping stays stale
    post.redraw(0x5678, 1)
    g_dense = tri.loss_gradients(x, y, S).cpu()
    assert post._k2ci_stale
    stacked = {k: torch.stack([post.state_dict(i)[k] for i in range(S)]) for k in ConvSviGuide.TENSOR_IDS}
    ref = make_engine(posterior_from_stacked("conv", act, (1, 28, 28), C, Hc, stacked, DEV), precision="exact").loss_gradients(x, y, S).cpu()
    monkeypatch.setenv("RBNN_CONV_BWD_EXACT", "1")               # the fp32 gather-form backward behind the triple forward: K2ci rebuilt from draw 0x5678 first
    g_gather = tri.loss_gradients(x, y, S).cpu()
    assert not post._k2ci_stale
    monkeypatch.delenv("RBNN_CONV_BWD_EXACT")
    post.redraw(0x5678, 1)                                        # the same draw again: stale once more
    assert post._k2ci_stale
    ex = make_engine(post, precision="exact")                     # the fp32 kernels on the same resident posterior: K2ci rebuilt first
    g_exact = ex.loss_gradients(x, y, S).cpu()
    assert not post._k2ci_stale
    for g in (g_dense, g_gather, g_exact):
        assert rel_err(g, ref) < 1e-5
    assert torch.equal(g_exact, ref)
