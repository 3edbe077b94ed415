"""The oracle against the round-3 fixtures: TRAINED posteriors (tests/golden/make_golden_trained.py: the reference's own NN.train,
attack, attack_evaluation, build_eps_attacks_df), where adversarial accuracy is not degenerate — the clean accuracy is 90-99 % and
the adversarial accuracy walks down with eps — and the reference's own PGD iterates, step by step.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_close_to_reference, rel_err, saturation_noise
from oracle import bnn_oracle as O

TOL = 1e-5
TAU = 1e-3          # |g| below tau * max|g| of that point may legitimately flip sign
HALFMOONS = ["trained_halfmoons_fc_h32_m10", "trained_halfmoons_fc2_h32_m10"]


def marginal_ok(adv, ref, grad, what=""):
    """Images equal except components whose gradient is within noise of zero; returns the number of (marginal) pixels that differ."""
    adv, ref, grad = (torch.as_tensor(v).reshape(len(ref), -1) for v in (adv, ref, grad))
    safe = grad.abs() > TAU * grad.abs().max(dim=1, keepdim=True)[0]
    diff = (adv - ref).abs() > 1e-6
    assert not (diff & safe).any(), f"{what}: {int((diff & safe).sum())} non-marginal pixels differ"
    return int(diff.sum())


@pytest.mark.parametrize("name", HALFMOONS)
def test_fixture_is_not_degenerate(golden, name):
    g = golden(name)
    assert g.arr["bnn_fgsm_orig_acc"].min() > 80.0
    aa = g.arr["bnn_fgsm_adv_acc"]                        # [eps, n_samples]
    assert (aa[0] > aa[1]).all() and (aa[1] > aa[2]).all() and aa[0].min() > 75 and aa[-1].max() < 20


@pytest.mark.parametrize("name", HALFMOONS)
@pytest.mark.parametrize("kind", ["bnn", "ens"])
def test_trained_fgsm_grid(golden, name, kind):
    g = golden(name); m = g.meta; post = g.posterior(); x, y = g.t("x"), g.t("y"); lab = y.argmax(-1)
    okind = "bnn" if kind == "bnn" else "ensemble"
    for k, ns in enumerate(m["ns_list"]):
        grad = O.meanprob_gradients(x, lab, post, m["arch"], m["act"], ns, kind=okind)
        ref_g = g.t(kind + "_fgsm_grad")[k]
        g64 = O.meanprob_gradients(x.double(), lab, O.cast(post, torch.float64), m["arch"], m["act"], ns, kind=okind)
        # the fp64 evaluation reproduces the reference wherever the reference's own fp32 is well conditioned; the fp32 closed form too
        assert_close_to_reference(g64, ref_g, g64, TOL, None, f"{kind} ns={ns} fp64")
        assert_close_to_reference(grad, ref_g, g64, TOL, saturation_noise(x, post, m["arch"], m["act"], ns, okind), f"{kind} ns={ns}")
        for e, eps in enumerate(m["eps_list"]):
            ref_adv = g.t(kind + "_fgsm_adv")[e, k]
            adv = O.fgsm_attack(x, lab, post, m["arch"], m["act"], ns, {"epsilon": eps}, kind=okind)
            n_marg = marginal_ok(adv, ref_adv, ref_g, f"{kind} eps={eps} ns={ns}")
            # the reference's adversarial set scored by the oracle: the accuracies are the reference's, softmax_rob to 1e-5 (measured 2e-7)
            oa, aa, rob = O.attack_evaluation(x, ref_adv, y, post, m["arch"], m["act"], ns, kind=okind)
            assert (oa, aa) == (float(g.arr[kind + "_fgsm_orig_acc"][e, k]), float(g.arr[kind + "_fgsm_adv_acc"][e, k]))
            assert float((rob - g.t(kind + "_fgsm_rob")[e, k]).abs().max()) < TOL
            if n_marg == 0:                               # end to end: the oracle's own attack scores the same
                oa2, aa2, _ = O.attack_evaluation(x, adv, y, post, m["arch"], m["act"], ns, kind=okind)
                assert (oa2, aa2) == (oa, aa)


@pytest.mark.parametrize("name", HALFMOONS)
def test_trained_pgd_grid(golden, name):
    g = golden(name); m = g.meta; post = g.posterior(); x, y = g.t("x"), g.t("y"); lab = y.argmax(-1)
    for kind, okind, eps_l, ns_l in (("bnn", "bnn", m["pgd_eps"], m["pgd_ns"]), ("ens", "ensemble", m["ens_pgd_eps"], m["ens_pgd_ns"])):
        for e, eps in enumerate(eps_l):
            for k, ns in enumerate(ns_l):
                ref_adv = g.t(kind + "_pgd_adv")[e, k]
                oa, aa, rob = O.attack_evaluation(x, ref_adv, y, post, m["arch"], m["act"], ns, kind=okind)
                assert (oa, aa) == (float(g.arr[kind + "_pgd_orig_acc"][e, k]), float(g.arr[kind + "_pgd_adv_acc"][e, k]))
                assert float((rob - g.t(kind + "_pgd_rob")[e, k]).abs().max()) < TOL
                adv = O.pgd_attack(x, lab, post, m["arch"], m["act"], ns, {"epsilon": eps}, kind=okind)
                same = ((adv - ref_adv).abs().reshape(len(x), -1).max(1)[0] <= 1e-6)
                assert float(same.double().mean()) > 0.97          # a reported statistic; the exact statement is the trajectory test
                oa2, aa2, rob2 = O.attack_evaluation(x[same], adv[same], y[same], post, m["arch"], m["act"], ns, kind=okind)
                assert float((rob2 - g.t(kind + "_pgd_rob")[e, k][same]).abs().max()) < TOL


def test_eps_grid_dataframe_matches_the_grid(golden):
    """build_eps_attacks_df (plot_eps_attacks.py:9-39) on the trained posterior = the same attack / attack_evaluation cells, one row per point."""
    g = golden("trained_halfmoons_fc_h32_m10"); m = g.meta
    N, E, K = m["N"], len(m["eps_list"]), len(m["ns_list"])
    assert g.arr["fgsm_df_epsilon"].shape == (E * K * N,)
    rows = lambda col: g.arr["fgsm_df_" + col].reshape(E, K, N)
    np.testing.assert_array_equal(rows("test_acc")[:, :, 0], g.arr["bnn_fgsm_orig_acc"])
    np.testing.assert_array_equal(rows("adv_acc")[:, :, 0], g.arr["bnn_fgsm_adv_acc"])
    np.testing.assert_array_equal(rows("softmax_rob").astype("float32"), g.arr["bnn_fgsm_rob"])


MNIST_SHAPED = ["trained_mnistshaped_fc_h128_m5", "trained_mnistshaped_fc2_h128_m3", "trained_mnistshaped_conv_h16_m3"]


@pytest.mark.parametrize("name", MNIST_SHAPED)
def test_trained_mnist_shaped(golden, name):
    g = golden(name); m = g.meta; post = g.posterior(); x, y = g.t("x"), g.t("y"); lab = y.argmax(-1)
    arch, act = m["arch"], m["act"]
    oa_, aa_ = g.arr["bnn_fgsm_orig_acc"], g.arr["bnn_fgsm_adv_acc"]
    assert oa_[:, -1].min() > 90 and 20 < aa_[2, -1] < 90 and aa_[-1].max() == 0 and (aa_[0] > aa_[2]).all()      # not a degenerate fixture
    for k, ns in enumerate(m["ns_list"]):
        grad = O.meanprob_gradients(x, lab, post, arch, act, ns)
        sign = g.t("bnn_fgsm_sign")[k].float()
        if ns == m["ns_list"][-1]:
            g64 = O.meanprob_gradients(x.double(), lab, O.cast(post, torch.float64), arch, act, ns)
            assert_close_to_reference(grad, g.t(f"bnn_fgsm_grad_ns{ns}"), g64, TOL, saturation_noise(x, post, arch, act, ns), f"{name} gradient")
        safe = grad.abs() > TAU * grad.abs().reshape(len(x), -1).max(1)[0].reshape(-1, 1, 1, 1)
        assert not ((grad.sign() != sign) & safe).any()
        for e, eps in enumerate(m["eps_list"]):
            ref_adv = torch.clamp(x + eps * sign, 0, 1)    # the reference's image, bit for bit (asserted when the fixture was written)
            oa, aa, rob = O.attack_evaluation(x, ref_adv, y, post, arch, act, ns)
            assert (oa, aa) == (float(oa_[e, k]), float(aa_[e, k]))
            assert float((rob - g.t("bnn_fgsm_rob")[e, k]).abs().max()) < TOL
    P = m["pgd_points"]
    oa, aa, rob = O.attack_evaluation(x[:P], g.t("bnn_pgd_adv"), y[:P], post, arch, act, m["pgd_ns"])
    assert (oa, aa) == (float(g.arr["bnn_pgd_orig_acc"]), float(g.arr["bnn_pgd_adv_acc"]))
    assert float((rob - g.t("bnn_pgd_rob")).abs().max()) < TOL


TRAJ = [(n, "bnn") for n in ["trained_halfmoons_fc_h32_m10", "pgd_traj_mnist_fc_h512_s8_n8"] + MNIST_SHAPED] + \
       [("pgd_traj_halfmoons_fc2_h32_m10", "bnn"), ("pgd_traj_halfmoons_fc2_h32_m10", "ens"), ("pgd_traj_halfmoons_fc2_h32_m10", "nn0"),
        ("pgd_traj_det_ens_fc_h32_m4_n6", "ens"), ("pgd_traj_det_ens_fc_h32_m4_n6", "nn0")]


@pytest.mark.parametrize("name,kind", TRAJ)
def test_pgd_single_steps_along_the_reference_trajectory(golden, name, kind):
    """adversarialAttacks.py:95-105 is a 40-step chaotic map: compare ONE step at a time, from the reference's own iterate k to its
    iterate k+1 — zero non-marginal mismatches over all 40 steps.  kind "bnn": mean of probabilities over the samples; "ens": the
    reference's Ensemble_NN (mean of logits, model_ensemble.py:57-67); "nn0": one deterministic NN (n_samples=None) — round 4's fixtures."""
    g = golden(name); m = g.meta; post = g.posterior()
    pre = "" if kind == "bnn" else kind + "_"
    traj, tg = g.t(pre + "traj"), g.t(pre + "traj_grad")   # [41, P, *shape], [40, P, *shape]
    P = traj.shape[1]
    x0, lab = traj[0], g.t("y")[:P].argmax(-1)
    assert torch.equal(x0, g.t("x")[:P])
    eps, alpha, _ = O.pgd_params(x0, {"epsilon": m["traj_eps"]})
    S = 1 if kind == "nn0" else m["traj_ns"]
    marginal = 0
    for k in range(40):
        nxt = O.pgd_step(traj[k], x0, lab, post, m["arch"], m["act"], S, eps, alpha, kind="bnn" if kind == "bnn" else "ensemble")
        marginal += marginal_ok(nxt, traj[k + 1], tg[k], f"step {k}")
    print(f"{name} [{kind}]: {marginal} marginal pixels differ over 40 steps x {P} points")
