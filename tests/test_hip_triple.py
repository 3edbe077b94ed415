"""GPU parity tests of the triple-split ("f16x6") mode (-m gpu): full-width fp32 operands on the f16 matrix pipe.

  images       the three fp16 pieces of every operand sum to the fp32 value BIT FOR BIT (|v| >= 2^-15 max|tensor|)
  oracle       forward / loss gradients / mean-probability gradients vs the fp64 oracle on ragged sizes, every tile configuration
  golden       the reference's fixtures through the reference's call surface (RBNN_PRECISION=triple), attack -> evaluation triples
  full size    C2 (S=100) FGSM and C3-shaped PGD: 288 rows on 256-point tile edges vs the fp64 oracle with the kernels' own
               activation decisions; triple vs exact mode: the triple mode's error vs fp64 is NOT larger than the fp32 MFMA's
Everything goes through the C-ABI (robustbnns_amd._hip); the oracle is the checker only.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import bnn_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("built_library")]
TOL, TAU, KINK, DEV = 1e-5, 1e-3, 2e-6, "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from robustbnns_amd import _hip
    _hip.load()


def _halves(t):
    """int16 storage -> float64 values of the fp16 bit patterns."""
    return torch.from_numpy(t.cpu().numpy().view(np.float16).astype(np.float64))


# ------------------------------------------------------------------ images
def test_triple_rows_image_is_bit_exact():
    """rbnn_triple_rows: p0 + p1 + p2 == v * 2^e exactly for every |v| >= 2^-15 * max|v| (fp32's 24 bits fit in 3 x 11 + 2 signs; the
    last fp16 subnormal bit is 2^-24 and max|v * 2^e| >= 2^13); below that the error is at most 2^-25 in scaled units."""
    from robustbnns_amd import _hip
    from robustbnns_amd.posterior import scale_exp
    k = _hip.HipKernels()
    g = torch.Generator().manual_seed(5)
    rows, cols, ld = 37, 100, 128
    v = torch.randn(rows, cols, generator=g) * torch.exp(torch.randn(rows, cols, generator=g) * 3.0)     # ~5 decades of magnitudes
    v[0, :8] = torch.tensor([1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e-5, -7.1e-9, 0.0, 255.0 / 255, 1.0 / 255])
    e = scale_exp(float(v.abs().max()))
    out = torch.zeros(rows, ld * 3, dtype=torch.int16, device=DEV)
    k.triple_rows(v.to(DEV).contiguous(), cols, e, out, ld)
    torch.cuda.synchronize()
    img = _halves(out).reshape(rows, ld // 32, 3, 32)
    assert torch.isfinite(img).all()
    rec = img.sum(2).reshape(rows, ld)                          # fp64 sum of three fp16 values: exact
    want = torch.zeros(rows, ld, dtype=torch.float64)
    want[:, :cols] = v.double() * 2.0 ** e
    big = want.abs() >= 0.5
    assert int(big.sum()) > 0.1 * rows * cols and int((~big).sum()) > 0.1 * rows * cols      # both regimes are exercised
    assert torch.equal(rec[big], want[big])                     # bit for bit
    assert float((rec - want).abs().max()) <= 2.0 ** -25
    assert float(img[:, :, 0].abs().max()) <= 2.0 ** 14
    # the grouped order (the fc forward's operands) holds the same pieces: [row group][stage][piece][16 rows][32]
    rp = (rows + 15) // 16 * 16
    outg = torch.zeros(rp, ld * 3, dtype=torch.int16, device=DEV)
    k.triple_rows(v.to(DEV).contiguous(), cols, e, outg, ld, grouped=True)
    torch.cuda.synchronize()
    imgg = _halves(outg).reshape(rp // 16, ld // 32, 3, 16, 32).permute(0, 3, 1, 2, 4).reshape(rp, ld // 32, 3, 32)
    assert torch.equal(imgg[:rows], img) and float(imgg[rows:].abs().max()) == 0.0      # rows past the last one: untouched


def test_triple_cols_and_generator_images():
    """rbnn_triple_cols / rbnn_triple_w2gen: layouts documented in include/robustbnns_hip.h and rbnn_triple.hip, values exact."""
    from robustbnns_amd import _hip
    from robustbnns_amd.posterior import scale_exp
    k = _hip.HipKernels()
    g = torch.Generator().manual_seed(6)
    S, Hn, Dn, ldc, Cn = 2, 64, 40, 48, 10
    W = torch.randn(S, Hn, Dn, generator=g) * 0.05
    e = scale_exp(float(W.abs().max()))
    out = torch.zeros(S * (Hn // 32) * 12 * ldc * 8, dtype=torch.int16, device=DEV)
    k.triple_cols(W.to(DEV).contiguous(), Hn, Dn, e, out, ldc)
    img = _halves(out).reshape(S, Hn // 32, 4, 3, ldc, 8).sum(3)               # [S, hb, lg, d, j]
    j = torch.arange(8)
    for lg in range(4):
        h = 16 * (j >> 2) + 4 * lg + (j & 3)                                    # unit within the 32-block of K slot 8*lg + j
        for hb in range(Hn // 32):
            want = (W[:, 32 * hb + h, :].double() * 2.0 ** e).permute(0, 2, 1)  # [S, d, j]
            assert torch.equal(img[:, hb, lg, :Dn, :], want)
            assert float(img[:, hb, lg, Dn:, :].abs().max()) == 0.0
    W2 = torch.randn(S, Cn, Hn, generator=g) * 0.05
    e2 = scale_exp(float(W2.abs().max()))
    gen = torch.zeros(S * (Hn // 16) * 1024, dtype=torch.int16, device=DEV)
    k.triple_w2gen(W2.to(DEV).contiguous(), Cn, Hn, e2, gen)
    gi = _halves(gen).reshape(S, Hn // 16, 2, 4, 16, 8)                         # [S, t, mfma, lg, li, slot]
    p = O_split3(W2.double() * 2.0 ** e2)                                       # [3, S, C, H]
    for t in range(Hn // 16):
        hs = slice(16 * t, 16 * t + 16)
        w = lambda piece, c: p[piece][:, c, hs]                                 # [S, 16]
        for (mf, lg, piece) in [(0, 0, 0), (0, 1, 0), (0, 2, 1), (1, 0, 1), (1, 1, 0), (1, 2, 2)]:
            for c in range(8):
                assert torch.equal(gi[:, t, mf, lg, :, c], w(piece, c))
        tail1 = [(0, 8), (0, 9), (0, 8), (0, 9), (0, 8), (0, 9), (1, 8), (1, 9)]
        tail2 = [(2, 8), (2, 9), (1, 8), (1, 9)]
        for sl, (piece, c) in enumerate(tail1):
            assert torch.equal(gi[:, t, 0, 3, :, sl], w(piece, c))
        for sl, (piece, c) in enumerate(tail2):
            assert torch.equal(gi[:, t, 1, 3, :, sl], w(piece, c))
        assert float(gi[:, t, 1, 3, :, 4:].abs().max()) == 0.0


def test_triple_forward_reproduces_single_products_bit_for_bit():
    """One input pixel = 1, W2 = identity on the first units, relu, logits out: logit c = W1[c, 0] * 1 — a single fp32 product.
    The triple forward must return the fp32 weight BIT FOR BIT for every weight down to 2^-15 of the largest one: all three
    pieces — including the fp16-subnormal low pieces of the small weights — go through the f16 MFMA unflushed and their sum is
    exact in the fp32 accumulator.  (The two-piece split mode cannot do this: it drops the last 1-2 bits.)"""
    from robustbnns_amd import AttackEngine, StackedPosterior
    Dn, Hn, Cn, S, N = 784, 128, 10, 2, 16
    g = torch.Generator().manual_seed(11)
    W1 = torch.zeros(S, Hn, Dn)
    W1[:, :, 1:] = torch.rand(S, Hn, Dn - 1, generator=g) - 0.5            # multiplied by x = 0
    W1[:, 0, 1] = 1.0                                                      # the largest weight of each sample: fixes the image scale at 2^14
    expo = torch.stack([torch.arange(0, 10), torch.arange(5, 15)]).float()                 # weights from 2^0 down to 2^-14 of the largest
    mant = 1.0 + torch.randint(0, 2 ** 23, (S, Cn), generator=g).float() * 2.0 ** -23      # full 24-bit significands in [1, 2)
    w = mant * 2.0 ** (-expo - 1.0)                                                        # < 1 = the largest weight
    W1[:, :Cn, 0] = w
    W2 = torch.zeros(S, Cn, Hn)
    W2[:, torch.arange(Cn), torch.arange(Cn)] = 1.0
    post = {"model.1.weight": W1, "model.1.bias": torch.zeros(S, Hn), "model.3.weight": W2, "model.3.bias": torch.zeros(S, Cn)}
    x = torch.zeros(N, 1, 28, 28)
    x[:, 0, 0, 0] = 1.0
    eng = AttackEngine(StackedPosterior("fc", "relu", (1, 28, 28), Cn, Hn, post, DEV), precision="triple")
    for si in range(S):
        z = eng.forward(x, 1, seeds=[si], logits=True).cpu()
        bad = (z != w[si].expand(N, Cn)).any(0)
        assert not bad.any(), f"sample {si}: weights at 2^-(1+{expo[si][bad].tolist()}) of the largest are not reproduced exactly: {z[0][bad]} vs {w[si][bad]}"


def O_split3(v):
    """fp64 restatement of the device split: three round-to-nearest fp16 pieces."""
    f16 = lambda t: torch.from_numpy(t.numpy().astype(np.float32).astype(np.float16).astype(np.float64))
    p0 = f16(v)
    p1 = f16(v - p0)
    p2 = f16(v - p0 - p1)
    return p0, p1, p2


# ------------------------------------------------------------------ oracle, ragged sizes
CASES = [  # arch, act, H, C, S, N
    ("fc", "leaky", 512, 10, 5, 300), ("fc", "relu", 512, 10, 3, 77), ("fc", "leaky", 256, 10, 7, 257), ("fc", "leaky", 128, 3, 4, 129),
    ("fc", "relu", 384, 7, 2, 40), ("fc", "leaky", 1024, 10, 2, 130),
    # fc2 = the reference's saved model_1 / 3 / 5 / 7 family: layer 1 writes the hidden activations as a triple image, the backward
    # runs in two steps through the fp32 dhid1 buffer
    ("fc2", "leaky", 512, 10, 3, 300), ("fc2", "relu", 256, 10, 4, 77), ("fc2", "leaky", 128, 4, 2, 257), ("fc2", "leaky", 1024, 10, 2, 40),
    # sigmoid / tanh: act' travels as an fp32 stream instead of the 1-bit stash
    ("fc", "sigm", 512, 10, 3, 130), ("fc", "tanh", 256, 10, 2, 77), ("fc2", "tanh", 512, 10, 2, 90), ("fc2", "sigm", 128, 5, 3, 40),
]


@pytest.mark.parametrize("arch,act,Hn,Cn,S,N", CASES)
def test_triple_vs_fp64_oracle(arch, act, Hn, Cn, S, N):
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    Dn = 784
    post = O.synthetic_posterior(arch, Dn, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=7)
    sp = StackedPosterior(arch, act, (1, 28, 28), Cn, Hn, post, DEV)
    assert sp.triple_supported()
    eng = AttackEngine(sp, precision="triple")
    assert eng.precision == "triple"
    p64 = O.bnn_forward(x.double(), O.cast(post, torch.float64), arch, act, S)
    assert rel_err(eng.forward(x, S).cpu(), p64) < TOL
    margin = O.kink_margin(x.double(), O.cast(post, torch.float64), arch, act, S)
    ok = margin > KINK
    assert int(ok.sum()) >= 0.4 * N
    lab = y.argmax(-1)
    g64 = O.loss_gradients(x.double(), y, O.cast(post, torch.float64), arch, act, S)
    assert rel_err(eng.loss_gradients(x, y, S).cpu()[ok], g64[ok]) < TOL
    gm64 = O.meanprob_gradients(x.double(), lab, O.cast(post, torch.float64), arch, act, S)
    G = eng.gradient(eng.pad_inputs(x), lab.int().to(DEV), None, S, _hip.LOSS_MEAN_PROB)[:, :Dn].cpu().reshape(x.shape)
    assert rel_err(G[ok], gm64[ok]) < TOL
    adv = eng.fgsm(x, y, S, 0.3).cpu()
    ref_adv = torch.clamp(x + 0.3 * gm64.sign().float(), 0, 1)
    safe = (gm64.abs() > TAU * gm64.abs().reshape(N, -1).max(1)[0].reshape(N, 1, 1, 1)) & ok.reshape(N, 1, 1, 1)
    assert int((((adv - ref_adv).abs() > 1e-6) & safe).sum()) == 0
    # sample subsets through the index buffer, logits output
    idx = [S - 1, 0]
    p_idx = O.bnn_forward(x.double(), O.cast(post, torch.float64), arch, act, 2, seeds=idx)
    assert rel_err(eng.forward(x, 2, seeds=idx).cpu(), p_idx) < TOL


def test_triple_is_refused_where_it_does_not_apply():
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    # a posterior with one huge outlier weight: its small weights would be carried to 2^-39 of the OUTLIER — auto keeps the fp32 MFMA
    wild = O.synthetic_posterior("fc", 784, 512, 10, 2, 0.05)
    wild["model.1.weight"][0, 3, 5] = 1.0e5
    sp = StackedPosterior("fc", "leaky", (1, 28, 28), 10, 512, wild, DEV)
    assert not sp.triple_supported() and AttackEngine(sp).precision == "exact"
    with pytest.raises(_hip.HipError):
        AttackEngine(sp, precision="triple")
    post = O.synthetic_posterior("fc", 784, 512, 10, 2, 0.05)
    post = O.synthetic_posterior("fc", 784, 64, 10, 2, 0.05)
    with pytest.raises(_hip.HipError):
        AttackEngine(StackedPosterior("fc", "leaky", (1, 28, 28), 10, 64, post, DEV), precision="triple")
    from robustbnns_amd.conv import ConvEngine, ConvStackedPosterior
    cpost = O.synthetic_posterior("conv", 784, 16, 10, 1, 0.05, in_ch=1, head=49 * 16)
    cpost["model.3.weight"][0, 1, 2, 3, 4] = 1.0e4                                        # wild dynamic range -> fp32 MFMA
    csp = ConvStackedPosterior("leaky", (1, 28, 28), 10, 16, cpost, DEV)
    assert not csp.triple_supported() and ConvEngine(csp).precision == "exact"
    with pytest.raises(_hip.HipError):
        ConvEngine(csp, precision="triple")


# ------------------------------------------------------------------ golden fixtures through the reference's call surface
@pytest.mark.parametrize("name", ["mnist_fc_h512_s8_n8_leaky", "mnist_fc_h512_s8_n8_relu"])
def test_golden_cases_in_triple_mode(golden, name, monkeypatch):
    from robustbnns_amd import adversarialAttacks as A
    from robustbnns_amd import _hip
    from test_hip_parity import adv_equal, make_bnn
    monkeypatch.setenv("RBNN_PRECISION", "triple")
    g = golden(name); m = g.meta; bnn = make_bnn(g); x, y = g.t("x"), g.t("y")
    eng = bnn._engine
    assert eng.precision == "triple"
    # how far each arithmetic mode lands from the reference's own fp32 values (printed with -s: the modes are interchangeable here)
    from robustbnns_amd import AttackEngine
    for mode in ("triple", "exact"):
        e2 = AttackEngine(eng.post, precision=mode)
        print(f"[golden {name}] {mode}: forward {rel_err(e2.forward(x, m['S']).cpu(), g.t('forward_probs')):.2e}  "
              f"loss_gradients {rel_err(e2.loss_gradients(x, y, m['S']).cpu(), g.t('loss_gradients')):.2e} (rel. to the reference's values)")
    assert rel_err(bnn.forward(x.to(DEV), n_samples=m["S"]).cpu(), g.t("forward_probs")) < TOL
    seeds = [int(s) for s in g.arr["forward_seeds"]]
    assert rel_err(bnn.forward(x.to(DEV), n_samples=len(seeds), seeds=seeds).cpu(), g.t("forward_probs_seeds")) < TOL
    assert rel_err(eng.loss_gradients(x, y, m["S"]).cpu(), g.t("loss_gradients")) < TOL
    assert rel_err(eng.loss_gradients(x, y, m["S_half"]).cpu(), g.t("loss_gradients_half")) < TOL
    G = eng.gradient(eng.pad_inputs(x), y.argmax(-1).int().to(DEV), None, m["S"], _hip.LOSS_MEAN_PROB)
    assert rel_err(G[:, :eng.post.D].cpu().reshape(x.shape), g.t("meanprob_grad")) < TOL
    hyper = {"epsilon": m["eps"]}
    adv = A.attack(net=bnn, x_test=x, y_test=y, dataset_name=m["dataset"], device=DEV, method="fgsm", filename=bnn.name,
                   hyperparams=hyper, n_samples=m["S"])
    adv_equal(adv, g.t("fgsm"), g.t("meanprob_grad"))
    oa, aa, rob = A.attack_evaluation(net=bnn, x_test=x, x_attack=adv, y_test=y, device=DEV, n_samples=m["S"])
    assert (oa, aa) == (float(g.arr["eval_orig_acc"]), float(g.arr["eval_adv_acc"]))
    assert float((rob.cpu() - g.t("eval_softmax_rob")).abs().max()) < 1e-5
    idx = torch.from_numpy(g.arr["pgd_idx"])
    adv = A.attack(net=bnn, x_test=x[idx], y_test=y[idx], dataset_name=m["dataset"], device=DEV, method="pgd", filename=bnn.name,
                   hyperparams=hyper, n_samples=m["S"])
    oa, aa, rob = A.attack_evaluation(net=bnn, x_test=x[idx], x_attack=adv, y_test=y[idx], device=DEV, n_samples=m["S"])
    assert (oa, aa) == (float(g.arr["eval_pgd_orig_acc"]), float(g.arr["eval_pgd_adv_acc"]))
    err = (rob.cpu() - g.t("eval_pgd_softmax_rob")).abs()
    same = ((adv.cpu() - g.t("pgd")).abs().reshape(len(idx), -1) > 1e-6).sum(1) == 0
    assert float(err[same].max() if same.any() else 0.0) < 1e-5 and float(err.max()) < 1e-3


# ------------------------------------------------------------------ full size: C2 FGSM and C3-shaped PGD
@pytest.mark.parametrize("S,method", [(100, "fgsm"), (500, "pgd")])
def test_triple_full_size_against_oracle_and_exact_mode(S, method):
    """C2 (S=100, FGSM) and C3 (S=500, PGD T=40) at N=10 000.  (1) 288 rows on both sides of 256-point tile edges vs the fp64 oracle
    with the kernels' own activation decisions: < 1e-5 on every row, and the triple mode's error is not larger than the fp32-MFMA
    mode's (x1.25 slack: both are fp32-accumulation noise).  (2) Triple vs exact: same gradients on the points with identical
    decisions, adversarial images equal except noise-level gradient components, identical accuracies, robustness within 1e-5 on
    identical images.  (3) eps-ball / range / bit-determinism of the whole attack."""
    import test_hip_round2 as R2
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    N, Dn, Hn, eps = 10000, R2.D, R2.H, 0.3
    post = R2.big_posterior(S, seed=61 + S)
    x, y = O.synthetic_inputs(N, (1, 28, 28), R2.C, seed=62)
    sp = StackedPosterior("fc", "leaky", (1, 28, 28), R2.C, Hn, post, DEV)
    lab = y.argmax(-1)
    rows = R2.EDGE_ROWS
    res = {}
    for precision in ("exact", "triple"):
        eng = AttackEngine(sp, precision=precision)
        G = eng.gradient(eng.pad_inputs(x), lab.int().to(DEV), None, S, _hip.LOSS_MEAN_PROB)[:, :Dn].cpu()
        mask = R2.hip_activation_mask(eng, N, S, rows)
        words = eng.workspace(N, S)["mask1"].view(S, Hn // 32, -1)[:, :, :N].clone()
        pinned, margin, worst_flip = R2.oracle_gradients_fp64(x[rows], lab[rows], post, S, "mean_prob", hip_mask=mask)
        den = pinned.abs().max(1)[0]
        err = ((G[rows].double() - pinned).abs().max(1)[0] / den)
        assert float(err.max()) < TOL and worst_flip < KINK
        adv = eng.fgsm(x, y, S, eps) if method == "fgsm" else eng.pgd(x, y, S, eps, alpha=None, iters=40)
        a = adv.cpu()
        assert float((a - x).abs().max()) <= eps + 1e-6 and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
        res[precision] = dict(eng=eng, G=G, words=words, err=err, adv=adv)
    e_x, e_t = res["exact"]["err"], res["triple"]["err"]
    print(f"[triple vs exact] {method} S={S}: error vs fp64 (pinned decisions, 288 rows)  exact: median {float(e_x.median()):.2e} max "
          f"{float(e_x.max()):.2e}   triple: median {float(e_t.median()):.2e} max {float(e_t.max()):.2e}")
    assert float(e_t.median()) <= 1.25 * float(e_x.median()) and float(e_t.max()) <= 1.25 * float(e_x.max()) + 1e-7
    same_dec = (res["exact"]["words"] == res["triple"]["words"]).all(0).all(0).cpu()
    assert int((~same_dec).sum()) < 0.1 * N
    assert rel_err(res["triple"]["G"][same_dec], res["exact"]["G"][same_dec]) < TOL
    a_e, a_t = res["exact"]["adv"].cpu().reshape(N, -1), res["triple"]["adv"].cpu().reshape(N, -1)
    diff = (a_e - a_t).abs() > 1e-6
    print(f"    points with a differing activation decision {int((~same_dec).sum())}/{N}; pixels differing {int(diff.sum())} of "
          f"{diff.numel()}; images differing {int(diff.any(1).sum())}")
    if method == "fgsm":
        Gx = res["exact"]["G"]
        safe = Gx.abs() > TAU * Gx.abs().max(1, keepdim=True)[0]
        assert int((diff & safe)[same_dec].sum()) == 0
    oa_e, aa_e, rob_e, _, _ = res["exact"]["eng"].evaluate(x, res["exact"]["adv"], y, S)
    oa_x, aa_x, rob_x, _, _ = res["triple"]["eng"].evaluate(x, res["exact"]["adv"], y, S)          # both modes score the SAME images
    assert oa_e == oa_x and aa_e == aa_x and float((rob_e - rob_x).abs().max()) < 1e-5
    oa_t, aa_t, rob_t, _, _ = res["triple"]["eng"].evaluate(x, res["triple"]["adv"], y, S)
    same_img = ~diff.any(1)
    d_rob = (rob_e - rob_t).abs().cpu()
    print(f"    accuracy: original {oa_e} / {oa_t}, adversarial {aa_e} / {aa_t}; softmax_rob |diff| on identical images "
          f"{float(d_rob[same_img].max()) if same_img.any() else 0.0:.2e}, overall max {float(d_rob.max()):.2e}")
    assert oa_e == oa_t and abs(aa_e - aa_t) <= 0.1
    assert float(d_rob[same_img].max() if same_img.any() else 0.0) < 1e-5
    eng = res["triple"]["eng"]
    again = eng.fgsm(x, y, S, eps) if method == "fgsm" else eng.pgd(x, y, S, eps, alpha=None, iters=40)
    assert torch.equal(again, res["triple"]["adv"])                                               # bit-deterministic


# ------------------------------------------------------------------ fc2 and conv at bench-like sizes: triple against the fp32-MFMA mode
def test_fc2_bench_shape_triple_vs_exact():
    """fc2-512 (the reference's saved model_1 shape) at S=24, N=3000: the two-step triple backward (through the fp32 dhid1 buffer and the
    in-register A split) against the fp32-MFMA kernels on the same posterior — gradients on the points whose activation decisions
    agree in both layers, FGSM images, evaluation triple."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    S, N, Hn, Cn, Dn = 24, 3000, 512, 10, 784
    post = O.synthetic_posterior("fc2", Dn, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=71)
    sp = StackedPosterior("fc2", "leaky", (1, 28, 28), Cn, Hn, post, DEV)
    lab = y.argmax(-1).int().to(DEV)
    out = {}
    for mode in ("exact", "triple"):
        eng = AttackEngine(sp, precision=mode)
        G = eng.gradient(eng.pad_inputs(x), lab, None, S, _hip.LOSS_MEAN_PROB)[:, :Dn].cpu().clone()
        ws = eng.workspace(N, S)
        masks = torch.cat([ws["mask1"].view(S, Hn // 32, -1)[:, :, :N], ws["mask2"].view(S, Hn // 32, -1)[:, :, :N]], 1).clone()
        out[mode] = dict(G=G, masks=masks, probs=eng.forward(x, S).cpu(), adv=eng.fgsm(x, y, S, 0.3).cpu(), eng=eng)
    same = (out["exact"]["masks"] == out["triple"]["masks"]).all(0).all(0).cpu()
    assert int((~same).sum()) < 0.1 * N
    assert rel_err(out["triple"]["probs"], out["exact"]["probs"]) < 1e-6
    assert rel_err(out["triple"]["G"][same], out["exact"]["G"][same]) < TOL
    Gx = out["exact"]["G"]
    safe = (Gx.abs() > TAU * Gx.abs().max(1, keepdim=True)[0]) & same[:, None]
    diff = (out["exact"]["adv"] - out["triple"]["adv"]).abs().reshape(N, -1) > 1e-6
    assert int((diff & safe).sum()) == 0
    ev_e = out["exact"]["eng"].evaluate(x, out["exact"]["adv"], y, S)
    ev_t = out["triple"]["eng"].evaluate(x, out["exact"]["adv"], y, S)
    assert ev_e[0] == ev_t[0] and ev_e[1] == ev_t[1] and float((ev_e[2] - ev_t[2]).abs().max()) < 1e-5


@pytest.mark.parametrize("act,gf,gb", [("leaky", 5, 8), ("tanh", 7, 4), ("leaky", 1, 1)])
def test_fc2_sample_groups_reproduce_the_ungrouped_pass(act, gf, gb, monkeypatch):
    """fc2 in the triple mode run group after group through ONE reused hidden image / dhid1 buffer (AttackEngine._fc2_groups): the
    forward is bit-identical to the one-group pass (a sample's blocks do not depend on the others'); the gradients agree to fp32
    rounding (the per-point gradient scale e(n) is taken over a group's samples instead of all of them: the pieces stay exact, only
    sub-2^-39 tails move); a ragged last group, a sample-index call and the workspace's size are covered."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    S, N, Hn, Cn, Dn = 22, 700, 256, 10, 784
    post = O.synthetic_posterior("fc2", Dn, Hn, Cn, S, 0.05)
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=72)
    sp = StackedPosterior("fc2", act, (1, 28, 28), Cn, Hn, post, DEV)
    lab = y.argmax(-1).int().to(DEV)
    seeds = [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8]
    res = {}
    for tag, env in (("one", ("0", "0")), ("grouped", (str(gf), str(gb)))):
        monkeypatch.setenv("RBNN_FC2_GROUP_FWD", env[0])
        monkeypatch.setenv("RBNN_FC2_GROUP_BWD", env[1])
        eng = AttackEngine(sp, precision="triple")
        ws = eng.workspace(N, S)
        res[tag] = dict(groups=ws["fc2_groups"], hid=ws["triple"]["hid_triple"].numel(), dhid=ws["dhid1"].numel(), chunk=ws["chunk"],
                        probs=eng.forward(x, S).cpu(), G=eng.gradient(eng.pad_inputs(x), lab, None, S, _hip.LOSS_MEAN_PROB)[:, :Dn].cpu().clone(),
                        Gl=eng.loss_gradients(x, y, S).cpu(), adv=eng.fgsm(x, y, S, 0.3).cpu(),
                        ps=eng.forward(x, len(seeds), seeds=seeds).cpu(), gs=eng.fgsm(x, y, len(seeds), 0.1, seeds=seeds).cpu())
    one, grp = res["one"], res["grouped"]
    assert one["groups"] == (S, S) and grp["groups"][0] == gf and grp["groups"][1] % grp["chunk"] == 0 and grp["groups"][1] < S
    assert grp["hid"] * S == one["hid"] * gf and grp["dhid"] * S == one["dhid"] * grp["groups"][1]     # the workspace shrinks with the group
    assert torch.equal(one["probs"], grp["probs"]) and torch.equal(one["ps"], grp["ps"])
    assert rel_err(grp["G"], one["G"]) < 2e-6 and rel_err(grp["Gl"], one["Gl"]) < 2e-6
    safe = one["G"].abs() > TAU * one["G"].abs().max(1, keepdim=True)[0]
    assert int((((one["adv"] - grp["adv"]).abs().reshape(N, -1) > 1e-6) & safe).sum()) == 0
    assert float(((one["gs"] - grp["gs"]).abs() > 1e-6).float().mean()) < 1e-4
    p64 = O.cast(post, torch.float64)
    g64 = O.meanprob_gradients(x[:64].double(), y[:64].argmax(-1), p64, "fc2", act, S).reshape(64, -1)
    far = O.kink_margin(x[:64].double(), p64, "fc2", act, S) > KINK if act == "leaky" else torch.ones(64, dtype=torch.bool)
    assert int(far.sum()) >= 8 and rel_err(grp["G"][:64][far], g64[far]) < TOL


@pytest.mark.parametrize("shape,N,S", [((1, 28, 28), 300, 3), ((3, 32, 32), 150, 2)])
def test_conv_bench_shape_triple_vs_exact(shape, N, S):
    """conv-512 on both geometries at a few hundred points: the triple conv2 forward / conv2^T backward against the fp32-MFMA kernels on
    the same posterior — probabilities and gradients to 1e-5, the gradients on the points whose pooling / sign decisions (both byte stashes) are
    identical in the two modes; FGSM images equal there except noise-level gradient components."""
    from robustbnns_amd import _hip
    from robustbnns_amd.conv import ConvEngine, ConvStackedPosterior
    Hc, Cn = 512, 10
    Din = shape[0] * shape[1] * shape[2]
    q2 = ((shape[1] - 4) // 2) - 5
    post = O.synthetic_posterior("conv", Din, Hc, Cn, S, 0.03, in_ch=shape[0], head=q2 * q2 * Hc)
    x, y = O.synthetic_inputs(N, shape, Cn, seed=81)
    sp = ConvStackedPosterior("leaky", shape, Cn, Hc, post, DEV)
    lab = y.argmax(-1).int().to(DEV)
    out = {}
    for mode in ("exact", "triple"):
        eng = ConvEngine(sp, precision=mode)
        assert eng.precision == mode
        probs = eng.forward(x, S).cpu()
        G = eng.gradient(eng.pad_inputs(x), lab, None, S, _hip.LOSS_MEAN_PROB).cpu().reshape(N, -1).clone()
        ws = eng.workspace(N, S)
        st1, st2 = ws["st1"].view(S, N, -1).clone(), ws["st2"].view(S, N, -1).clone()
        out[mode] = dict(G=G, probs=probs, st1=st1, st2=st2, adv=eng.fgsm(x, y, S, 0.1).cpu().reshape(N, -1))
    same = ((out["exact"]["st1"] == out["triple"]["st1"]).all(2).all(0) & (out["exact"]["st2"] == out["triple"]["st2"]).all(2).all(0)).cpu()
    print(f"[conv {shape} triple vs exact] points with identical decisions: {int(same.sum())}/{N}")
    assert int(same.sum()) > 0.8 * N
    assert rel_err(out["triple"]["probs"], out["exact"]["probs"]) < TOL          # (each mode sits ~1e-6 from fp64 at 3x32x32: a 41 472-term head)
    assert rel_err(out["triple"]["G"][same], out["exact"]["G"][same]) < TOL
    Gx = out["exact"]["G"]
    safe = (Gx.abs() > TAU * Gx.abs().max(1, keepdim=True)[0]) & same[:, None]
    assert int((((out["exact"]["adv"] - out["triple"]["adv"]).abs() > 1e-6) & safe).sum()) == 0


def test_triple_mode_heavy_tails_and_sparse_inputs():
    """ONE power-of-two scale per tensor, taken from its largest magnitude: outlier weights (0.1 % at 100x the bulk — inside the
    dynamic-range guard), MNIST-like inputs (80 % exact zeros, saturated ones) and a few all-zero images.  The triple mode must stay
    at the fp32-MFMA kernels' own error against fp64 (outlier weights make the logits large: fp32 itself is the yardstick)."""
    from robustbnns_amd import AttackEngine, StackedPosterior, _hip
    Dn, Hn, Cn, S, N = 784, 256, 10, 5, 200
    post = O.synthetic_posterior("fc", Dn, Hn, Cn, S, 0.03)
    g = torch.Generator().manual_seed(17)
    for k in ("model.1.weight", "model.3.weight"):
        m = torch.rand(post[k].shape, generator=g) < 1e-3
        post[k] = torch.where(m, post[k] * 100.0, post[k])
    x, y = O.synthetic_inputs(N, (1, 28, 28), Cn, seed=8)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.8, torch.zeros_like(x), x)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.05, torch.ones_like(x), x)
    x[:3] = 0.0
    p64 = O.cast(post, torch.float64)
    sp = StackedPosterior("fc", "leaky", (1, 28, 28), Cn, Hn, post, DEV)
    assert sp.triple_supported()
    lab = y.argmax(-1)
    ref = O.meanprob_gradients(x.double(), lab, p64, "fc", "leaky", S)
    ok = O.kink_margin(x.double(), p64, "fc", "leaky", S) > KINK
    err = {}
    for mode in ("exact", "triple"):
        eng = AttackEngine(sp, precision=mode)
        assert rel_err(eng.forward(x, S).cpu(), O.bnn_forward(x.double(), p64, "fc", "leaky", S)) < TOL
        G = eng.gradient(eng.pad_inputs(x), lab.int().to(DEV), None, S, _hip.LOSS_MEAN_PROB)[:, :Dn].cpu().reshape(x.shape)
        err[mode] = rel_err(G[ok], ref[ok])
    print(f"heavy tails: triple {err['triple']:.2e}  fp32 MFMA {err['exact']:.2e}")
    assert err["triple"] < max(TOL, 1.25 * err["exact"])
