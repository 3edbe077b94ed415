"""GPU tests (-m gpu) of rbnn_lowdim_run — the one-launch hot path of low-dimensional fc nets (half-moons), what `precision="auto"`
resolves to for in_features <= 16: against the fp64 oracle, against the 7-kernel fp32-MFMA path on the same posterior, and the
T-iteration attack in one launch against T one-iteration launches.  (The reference's half-moons fixtures run through this path in
tests/test_hip_parity.py — GOLDEN_MODES' "auto" entries: forward_probs, loss_gradients, meanprob_grad, FGSM / PGD images and the evaluation
triple of halfmoons_fc_h64_s10_n100 and halfmoons_fc2_h32_s6_n40 on the lowdim kernels — and, the trained posteriors, in
tests/test_hip_round3.py, whose engines are built with the default precision.)"""
import pytest
import torch

from conftest import assert_close_to_truth, rel_err, saturation_noise
from oracle import bnn_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("built_library")]
TOL = 1e-5
TAU = 1e-3
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"


def engines(shape, H, C, S, act, std=0.5, arch="fc"):
    from robustbnns_amd import AttackEngine, StackedPosterior
    D = shape[0] * shape[1] * shape[2]
    post = O.synthetic_posterior(arch, D, H, C, S, std)
    sp = StackedPosterior(arch, act, shape, C, H, post, DEV)
    low = AttackEngine(sp)
    assert low.precision == "lowdim"
    return post, low, AttackEngine(sp, precision="exact")


CASES = [((1, 2, 1), 64, 2, 10, 100, "leaky"), ((1, 2, 1), 32, 2, 3, 7, "relu"), ((1, 2, 1), 512, 2, 50, 300, "leaky"),
         ((1, 3, 1), 128, 5, 6, 33, "tanh"), ((1, 7, 1), 64, 10, 4, 50, "sigm"), ((1, 4, 4), 256, 3, 300, 20, "leaky"),
         ((1, 2, 1), 16, 2, 5, 1000, "leaky"), ((1, 2, 1), 64, 2, 1, 5000, "leaky")]


# fc2 (round 4; rbnn_lowdim.hip's low2 kernels: the H x H layer on the fp32 MFMA, 4 launches per pass issued by one C call): every tile
# plan (hidden 32 / 64 / 128 / 256 / 512; 16 is stored padded to 32), point counts around the 112 / 64 / 32-point blocks, all activations
CASES2 = [((1, 2, 1), 32, 2, 6, 40, "leaky", 0.5), ((1, 2, 1), 128, 2, 25, 100, "leaky", 0.2), ((1, 2, 1), 64, 2, 5, 113, "relu", 0.3),
          ((1, 2, 1), 256, 2, 9, 130, "leaky", 0.12), ((1, 2, 1), 512, 2, 7, 70, "leaky", 0.08), ((1, 3, 1), 128, 5, 4, 33, "tanh", 0.2),
          ((1, 7, 1), 64, 10, 3, 50, "sigm", 0.3), ((1, 4, 4), 256, 3, 5, 20, "leaky", 0.12), ((1, 2, 1), 16, 2, 5, 300, "leaky", 0.5)]
ALL_CASES = [c + (0.5, "fc") for c in CASES] + [c + ("fc2",) for c in CASES2]


@pytest.mark.parametrize("shape,H,C,S,N,act,std,arch", ALL_CASES)
def test_lowdim_against_fp64_oracle_and_the_mfma_path(shape, H, C, S, N, act, std, arch):
    from robustbnns_amd import _hip
    post, low, ex = engines(shape, H, C, S, act, std, arch)
    p64 = O.cast(post, torch.float64)
    x, y = O.synthetic_inputs(N, shape, C, seed=H + N)
    lab = y.argmax(-1)
    # forward: probabilities, logits, a seeds subset
    assert rel_err(low.forward(x, S).cpu(), O.bnn_forward(x.double(), p64, arch, act, S)) < TOL
    assert rel_err(low.forward(x, S, logits=True).cpu(), O.ensemble_forward(x.double(), p64, arch, act, S)) < TOL
    sub = [S - 1, 0, S // 2]
    assert rel_err(low.forward(x, 3, seeds=sub).cpu(), O.bnn_forward(x.double(), p64, arch, act, 3, seeds=sub)) < TOL
    # expected gradients: per-sample loss (lossGradients.py) with fused norms, mean-probability and mean-logit losses
    ok = O.kink_margin(x.double(), p64, arch, act, S) > 2e-6 if act in ("relu", "leaky") else torch.ones(N, dtype=torch.bool)
    ref = O.loss_gradients(x.double(), y, p64, arch, act, S)
    g, linf, l2 = low.loss_gradients(x, y, S, norms=True)
    # std-0.5 weights on 512 hidden units saturate the softmax (p = 1 - 1e-6): fp32's own floor there is 2^-24 / (1 - p) (conftest.saturation_noise)
    noise = saturation_noise(x, post, arch, act, S)
    assert_close_to_truth(g.cpu(), ref, TOL, noise, "per-sample-loss gradient", rows=ok, fp32_yardstick=O.loss_gradients(x, y, post, arch, act, S))
    flat = g.cpu().reshape(N, -1)
    assert torch.allclose(linf.cpu(), flat.abs().max(1)[0], rtol=1e-6, atol=0) and torch.allclose(l2.cpu(), flat.norm(dim=1), rtol=1e-5, atol=1e-30)
    Xp, labd = low.pad_inputs(x), lab.int().to(DEV)
    for mode, kind in ((_hip.LOSS_MEAN_PROB, "bnn"), (_hip.LOSS_MEAN_LOGIT, "ensemble")):
        ref = O.meanprob_gradients(x.double(), lab, p64, arch, act, S, kind=kind)
        G = low.gradient(Xp, labd, None, S, mode).cpu()[:, :x[0].numel()].reshape(x.shape).clone()
        nz = noise if kind == "bnn" else saturation_noise(x, post, arch, act, S, "ensemble")
        y32 = O.meanprob_gradients(x, lab, post, arch, act, S, kind=kind)
        assert_close_to_truth(G, ref, TOL, nz, f"{kind} gradient", rows=ok, fp32_yardstick=y32)
        Ge = ex.gradient(ex.pad_inputs(x), labd, None, S, mode).cpu()[:, :x[0].numel()].reshape(x.shape)
        assert_close_to_truth(Ge, ref, TOL, nz, f"{kind} gradient, fp32-MFMA path", rows=ok, fp32_yardstick=y32)   # the 7-kernel path, same posterior, same bar
        # FGSM, and PGD with 7 iterations in one launch
        safe = ref.abs() > TAU * ref.abs().reshape(N, -1).max(1)[0].reshape(N, 1, 1, 1)
        adv = low.fgsm(x, y, S, 0.1, mode=mode).cpu()
        want = torch.clamp(x + 0.1 * ref.sign().float(), 0, 1)
        assert not (((adv - want).abs() > 1e-6) & safe)[ok].any()
        pg = low.pgd(x, y, S, 0.15, iters=7, mode=mode).cpu()
        assert float((pg - x).abs().max()) <= 0.15 + 1e-6 and float(pg.min()) >= 0 and float(pg.max()) <= 1
        step = x.clone()
        for _ in range(7):                                                     # one launch per iteration: the same iterates, bit for bit
            step = low.pgd_continue(step, x, y, S, 0.15, mode=mode).cpu()
        assert torch.equal(step, pg)
        same = ((pg - ex.pgd(x, y, S, 0.15, iters=7, mode=mode).cpu()).abs().reshape(N, -1).max(1)[0] <= 1e-6).double().mean()
        assert float(same) > 0.9                                               # vs the MFMA path: identical except sign flips of noise-level components
    assert torch.equal(low.pgd(x, y, S, 0.3, alpha=2 / 225, iters=5).cpu(), low.pgd(x, y, S, 0.3, alpha=2 / 225, iters=5).cpu())   # deterministic
    # evaluation and the autograd hook (the hook's upstream-gradient backward runs the fp32-MFMA kernels)
    oa, aa, rob, o, a_ = low.evaluate(x, adv, y, S)
    oe, ae, robe, _, _ = ex.evaluate(x, adv, y, S)
    assert abs(oa - oe) <= 100.0 / N and abs(aa - ae) <= 100.0 / N and float((rob - robe).abs().max()) < 1e-5
    xg = x.clone().to(DEV).requires_grad_(True)
    torch.nn.CrossEntropyLoss(reduction="sum")(low.forward(xg, S), lab.to(DEV)).backward()
    ref = O.meanprob_gradients(x.double(), lab, p64, arch, act, S)
    assert_close_to_truth(xg.grad.cpu(), ref, TOL, noise, "autograd hook", rows=ok, fp32_yardstick=O.meanprob_gradients(x, lab, post, arch, act, S))


def test_auto_is_lowdim_only_where_it_applies():
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    mk = lambda arch, shape, H: StackedPosterior(arch, "leaky", shape, 2, H, O.synthetic_posterior(arch, shape[1], H, 2, 2, 0.5), DEV)
    assert AttackEngine(mk("fc", (1, 2, 1), 64)).precision == "lowdim"
    assert AttackEngine(mk("fc", (1, 16, 1), 128)).precision == "lowdim"       # wins over triple: 16 columns are one MFMA K step of padding
    assert AttackEngine(mk("fc", (1, 17, 1), 128)).precision == "triple"
    assert AttackEngine(mk("fc2", (1, 2, 1), 64)).precision == "lowdim"        # round 4: the reference's half-moons grid (fc2) has its own kernels
    assert AttackEngine(mk("fc2", (1, 2, 1), 1024)).precision == "triple"       # no fc2 tile plan beyond hidden 512
    assert AttackEngine(mk("fc2", (1, 17, 1), 128)).precision == "triple"
    with pytest.raises(_hip.HipError):
        AttackEngine(mk("fc2", (1, 2, 1), 1024), precision="lowdim")
