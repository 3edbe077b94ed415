"""CPU-only tests of the host side: the C-ABI library loads and exports everything include/*.h declares,
argument validation returns the documented codes without touching a GPU, and the reference-mirroring
modules keep the reference's names, signatures, error behaviour and file side effects.
Orchestration runs on tests/fake_kernels.py (a CPU test double); no HIP compute is called here."""
import ctypes as C
import inspect
import os
import re

import numpy as np
import pytest
import torch

import robustbnns_amd as R
from robustbnns_amd import _hip, adversarialAttacks, lossGradients, model_bnn, model_ensemble, model_nn
from robustbnns_amd.engine import AttackEngine
from robustbnns_amd.posterior import StackedPosterior
from conftest import rel_err
from fake_kernels import FakeKernels
from oracle import bnn_oracle as O

pytestmark = pytest.mark.usefixtures("built_library")


# ----------------------------------------------------------------------------- C-ABI
def test_library_exports_every_declared_symbol():
    hdr = open(_hip.HEADER_PATH).read()
    declared = set(re.findall(r"\b(rbnn_\w+)\s*\(", hdr))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib = _hip.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rbnn_abi_version() == _hip.ABI_VERSION == 9
    assert lib.rbnn_build_flags() == 0                       # a product build: no timing-only ablation switch in any translation unit
    assert lib.rbnn_strerror(0) == b"ok" and b"NULL" in lib.rbnn_strerror(-1)


def test_ablation_switches_are_fenced(tmp_path):
    """The timing-only switches compiled into the product kernels (RBNN_ABL, RBNN_*_ABL_*, RBNN_FAST_BUILD: wrong results by design)
    do not compile without -DRBNN_ALLOW_ABLATION, and a translation unit built with one makes rbnn_build_flags() non-zero — which
    _hip.load() refuses.  Host-side compile of the common header only (seconds, no GPU)."""
    import subprocess
    import __graft_entry__ as ge
    src = tmp_path / "probe.hip"
    src.write_text('#include "%s/rbnn_common.hpp"\n'
                   'extern __attribute__((weak)) int rbnn_ablation_build_marker;\n'
                   'int main() { return (&rbnn_ablation_build_marker != nullptr) ? 7 : 0; }\n' % ge.CSRC)
    base = [ge.HIPCC, "--offload-arch=gfx950", "-std=c++17", "--cuda-host-only", "-o", str(tmp_path / "probe"), str(src)]
    r = subprocess.run(base + ["-DRBNN_DENSE_ABL_NOEPI"], capture_output=True, text=True)
    assert r.returncode != 0 and "RBNN_ALLOW_ABLATION" in r.stderr
    for flag in ("-DRBNN_ABL=4", "-DRBNN_X3_L1_ABL_SMALL", "-DRBNN_FAST_BUILD", "-DRBNN_DENSE_STAMPS=1", "-DRBNN_DENSE_ABL_PARTA=16"):   # (round 4: the stamp build, the partial weight tiles)
        assert subprocess.run(base + [flag], capture_output=True).returncode != 0
    assert subprocess.run(base, capture_output=True).returncode == 0 and subprocess.run([str(tmp_path / "probe")]).returncode == 0
    assert subprocess.run(base + ["-DRBNN_DENSE_ABL_NOEPI", "-DRBNN_ALLOW_ABLATION"], capture_output=True).returncode == 0
    assert subprocess.run([str(tmp_path / "probe")]).returncode == 7          # the marker is planted: rbnn_build_flags() would say 1


def test_load_refuses_a_library_built_with_ablation_switches(monkeypatch):
    class Fake:
        def __getattr__(self, name):
            f = lambda *a: {"rbnn_abi_version": _hip.ABI_VERSION, "rbnn_build_flags": 1}.get(name, 0)
            return f
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip.C, "CDLL", lambda path: Fake())
    monkeypatch.delenv("RBNN_ALLOW_ABLATION", raising=False)
    with pytest.raises(_hip.HipError, match="ablation"):
        _hip.load()
    monkeypatch.setenv("RBNN_ALLOW_ABLATION", "1")
    assert _hip.load() is not None
    monkeypatch.setattr(_hip, "_lib", None)                  # the next load() binds the real library again


def _net(**kw):
    d = _hip.Posterior()
    d.arch, d.activation, d.in_features, d.in_stride, d.hidden, d.n_classes, d.n_stored = 0, 1, 784, 784, 512, 10, 4
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_workspace_query_sizes():
    lib = _hip.load()
    out = _hip.WorkspaceSizes()
    assert lib.rbnn_workspace_query(C.byref(_net()), 10000, 100, 0, C.byref(out)) == 0
    assert out.P == out.dZ == 100 * 10000 * 16 * 4
    assert out.mask1 == 100 * (512 // 32) * 10240 * 4 and out.dact1 == 0 and out.hid1 == 0    # rows padded to 256 points
    assert 1 <= out.chunk <= 8 and out.n_slabs == -(-100 // out.chunk)
    assert out.slabs == out.n_slabs * 10000 * 784 * 4
    assert lib.rbnn_workspace_query(C.byref(_net(activation=2, arch=1)), 64, 3, 2, C.byref(out)) == 0
    assert out.mask1 == 0 and out.dact1 == out.hid1 == out.dhid1 == out.dact2 == 3 * 64 * 512 * 4
    assert (out.chunk, out.n_slabs) == (2, 2)
    assert lib.rbnn_workspace_query(C.byref(_net(in_stride=780)), 8, 2, 0, C.byref(out)) == -2     # D_pad % 16
    assert lib.rbnn_workspace_query(C.byref(_net(hidden=48)), 8, 2, 0, C.byref(out)) == -2
    assert lib.rbnn_workspace_query(None, 8, 2, 0, C.byref(out)) == -1


def test_argument_validation_without_gpu():
    lib = _hip.load()
    ws = _hip.Workspace()
    # NULL weights -> RBNN_ERR_NULL before any launch
    assert lib.rbnn_fc_forward(C.byref(_net()), None, 784, 8, None, 2, 0, C.byref(ws), None) == -1
    assert lib.rbnn_fc_input_grad(C.byref(_net()), None, 2, 8, 0, C.byref(ws), None, None) == -1
    assert lib.rbnn_reduce_samples(None, 1, 1, 1, 1.0, None, 16, None) == -1
    assert lib.rbnn_loss_dlogits(7, C.c_void_p(16), None, 16, None, None, 1, 1.0, 1, 2, C.c_void_p(16), None) == -3
    assert lib.rbnn_sum_slabs(C.c_void_p(16), 1, 4, 18, 1.0, C.c_void_p(16), 18, None) == -2        # d_pad % 4
    assert lib.rbnn_attack_step(None, None, 16, None, 1, 0, 16, None, 0.1, 0.1, 0, 4, 2, None) == -1
    assert lib.rbnn_pgd_alpha(C.c_void_p(16), 1, 4, 2, C.c_void_p(16), None) == -2                  # ldx < D
    assert lib.rbnn_svi_materialize(None, None, None, 4, 1, None, None) == -1
    assert lib.rbnn_pack_rows4(C.c_void_p(16), 6, 8, C.c_void_p(16), None) == -2                    # rows % 4


def test_split_mode_validation_without_gpu():
    """Split-half (f16x3) entry points: argument checks and workspace sizes, no launch."""
    lib = _hip.load()
    ws, sws, img = _hip.Workspace(), _hip.SplitWorkspace(), _hip.SplitImages()
    img.ld_rows, img.ld_cols, img.w1_exp, img.w2_exp = 800, 784, 15, 14
    out = _hip.SplitWorkspaceSizes()
    assert lib.rbnn_split_workspace_query(C.byref(_net()), C.byref(img), 10000, 100, C.byref(out)) == 0
    assert out.X_split == 10000 * 800 * 4 and out.dZ_gen == 100 * 10240 * 64 and out.g_scale == 10240 * 4
    img.ld_rows = 784                                                                              # not a multiple of 32
    assert lib.rbnn_split_workspace_query(C.byref(_net()), C.byref(img), 8, 2, C.byref(out)) == -2
    img.ld_rows = 800
    assert lib.rbnn_split_rows(None, 4, 8, 8, 0, None, C.c_void_p(16), 32, None) == -1
    assert lib.rbnn_split_rows(C.c_void_p(16), 4, 8, 8, 0, None, C.c_void_p(16), 24, None) == -2          # ld_dst % 32
    assert lib.rbnn_split_rows(C.c_void_p(16), 4, 8, 8, 0, None, C.c_void_p(8), 32, None) == -5           # alignment
    assert lib.rbnn_input_scales(None, 4, 8, 8, 0.0, 0.0, 0.0, 1.0, C.c_void_p(16), None) == -1
    assert lib.rbnn_input_scales(C.c_void_p(16), 4, 8, 6, 0.0, 0.0, 0.0, 1.0, C.c_void_p(16), None) == -2     # ld < cols
    assert lib.rbnn_input_scales(C.c_void_p(16), 4, 8, 8, -1.0, 0.0, 0.0, 1.0, C.c_void_p(16), None) == -2    # negative floor
    assert lib.rbnn_input_scales(C.c_void_p(16), 4, 8, 8, 0.0, 0.0, 0.0, 1.0, C.c_void_p(8), None) == -5      # alignment
    assert lib.rbnn_split_cols(C.c_void_p(16), 1, 48, 8, 8, 0, C.c_void_p(16), 16, None) == -2      # rows % 32
    assert lib.rbnn_split_w2gen(C.c_void_p(16), 1, 11, 128, 0, C.c_void_p(16), None) == -2         # classes > 10
    assert lib.rbnn_fc_forward_split(C.byref(_net()), C.byref(img), None, 800, 14, None, 8, None, 2, 0, C.byref(ws), None) == -1
    ws.P = C.c_void_p(16); img.W1_rows = C.c_void_p(16)
    w = dict(W1=C.c_void_p(16), b1=C.c_void_p(16), W2=C.c_void_p(16), b2=C.c_void_p(16))
    assert lib.rbnn_fc_forward_split(C.byref(_net(arch=1, **w)), C.byref(img), C.c_void_p(16), 800, 14, None, 8, None, 2, 0, C.byref(ws), None) == -1   # fc2 without Wm_rows / bm / hid1
    assert lib.rbnn_fc_input_grad_split(C.byref(_net(arch=1, **w)), C.byref(img), None, 2, 8, 0, C.byref(ws), C.byref(sws), None, None) == -1   # fc2 without Wm_cols / mask2 / dhid1
    assert lib.rbnn_fc_forward_split(C.byref(_net(hidden=64, **w)), C.byref(img), C.c_void_p(16), 800, 14, None, 8, None, 2, 0, C.byref(ws), None) == -2  # hidden % 128
    assert lib.rbnn_fc_forward_split(C.byref(_net(**w)), C.byref(img), C.c_void_p(16), 784, 14, None, 8, None, 2, 0, C.byref(ws), None) == -2             # ldx != ld_rows
    assert lib.rbnn_fc_input_grad_split(C.byref(_net(**w)), C.byref(img), None, 2, 8, 0, C.byref(ws), C.byref(sws), None, None) == -1


def test_precision_resolution_and_scale_exponent(monkeypatch):
    from robustbnns_amd.posterior import scale_exp
    assert scale_exp(1.0) == 14 and scale_exp(0.999) == 14 and scale_exp(1.001) == 13 and scale_exp(3.0e-3) == 22
    assert scale_exp(0.0) == 0 and scale_exp(float("inf")) == 0 and scale_exp(1e-40) == 100
    for v in (1e-6, 0.3, 1.0, 7.5, 4096.0):
        assert v * 2.0 ** scale_exp(v) <= 2.0 ** 14 < v * 2.0 ** (scale_exp(v) + 1) * (1 + 1e-12)
    post = O.synthetic_posterior("fc", 784, 128, 10, 2, 0.05)
    sp = StackedPosterior("fc", "leaky", (1, 28, 28), 10, 128, post, "cpu")
    assert not sp.split_supported()                                # CPU tensors: nothing to build images from
    assert AttackEngine(sp, kernels=FakeKernels()).precision == "exact"      # injected test kernels never take the split path
    monkeypatch.setenv("RBNN_PRECISION", "exact")
    assert AttackEngine(sp, kernels=FakeKernels()).precision == "exact"
    with pytest.raises(_hip.HipError):
        AttackEngine(sp, kernels=FakeKernels(), precision="split")
    with pytest.raises(ValueError):
        AttackEngine(sp, kernels=FakeKernels(), precision="fp8")
    # the triple mode (what auto picks on the GPU) needs device-resident images and the real kernels: refused, never silently replaced
    assert not sp.triple_supported()
    monkeypatch.delenv("RBNN_PRECISION")
    assert AttackEngine(sp, kernels=FakeKernels(), precision="auto").precision == "exact"
    with pytest.raises(_hip.HipError):
        AttackEngine(sp, kernels=FakeKernels(), precision="triple")
    # a sharded posterior drops its images (they are rebuilt for the shard's own samples)
    sh = sp.shard(0, 2)
    assert sh.S == 1 and sh._triple is None and sh._split is None


def test_triple_abi_rejects_bad_arguments():
    """The triple entry points of the C-ABI validate like the others (host-side checks only: no kernel is launched)."""
    import ctypes as C
    lib = _hip.load()
    img, tws, ws = _hip.TripleImages(), _hip.TripleWorkspace(), _hip.Workspace()
    net = _hip.Posterior()
    assert lib.rbnn_fc_forward_triple(None, C.byref(img), C.byref(tws), 0, None, 8, None, 2, 0, C.byref(ws), None) == -1
    assert lib.rbnn_fc_forward_triple(C.byref(net), C.byref(img), C.byref(tws), 0, None, 8, None, 2, 0, C.byref(ws), None) == -1   # no images
    assert lib.rbnn_fc_input_grad_triple(C.byref(net), C.byref(img), None, 2, 8, 0, C.byref(ws), C.byref(tws), None, None) == -1
    assert lib.rbnn_triple_rows(None, 4, 32, 32, 0, None, None, 32, None) == -1
    assert lib.rbnn_triple_rows(C.c_void_p(16), 4, 40, 40, 0, None, C.c_void_p(16), 40, None) == -2          # ld_dst % 32
    assert lib.rbnn_triple_cols(C.c_void_p(16), 1, 48, 16, 16, 0, C.c_void_p(16), 16, None) == -2             # rows % 32
    assert lib.rbnn_triple_w2gen(C.c_void_p(16), 1, 11, 128, 0, C.c_void_p(16), None) == -2                   # > 10 classes
    sizes = _hip.TripleWorkspaceSizes()
    net.arch, net.in_features, net.hidden = 1, 784, 512
    img.ld_rows = 800
    assert lib.rbnn_triple_workspace_query(C.byref(net), C.byref(img), 100, 7, C.byref(sizes)) == 0
    assert (sizes.X_triple, sizes.hid_triple, sizes.g_scale) == (112 * 800 * 6, 7 * 112 * 512 * 4, 256 * 4)   # grouped images: whole 16-row groups (the hidden image: fp32 since round 5)
    cnet = _hip.ConvPosterior()
    assert lib.rbnn_conv_forward_triple(C.byref(cnet), None, 0, 0, None, None, 784, 4, None, 1, 0, None, None) != 0
    assert lib.rbnn_conv_input_grad_dense(C.byref(cnet), None, 0, 1.0, None, 1, 4, None, None) != 0


def test_lowdim_fc2_scratch_holds_the_sign_bit_stash():
    """rbnn_lowdim_scratch_bytes (host arithmetic only): fc = the per-sample outputs; fc2 = outputs + gradient slabs + the sum over samples
    + 16 bytes per thread of every low2_kernel block for the sign bits the forward launch leaves to the backward launch (round 4)."""
    import ctypes as C
    lib = _hip.load()
    net = _hip.Posterior()
    net.in_features, net.n_classes, net.hidden, net.in_stride = 2, 2, 128, 16
    N, S = 100, 250
    net.arch = 0                                                   # RBNN_ARCH_FC
    assert lib.rbnn_lowdim_scratch_bytes(C.byref(net), N, S) == S * N * 16 * 4
    net.arch = 1                                                   # RBNN_ARCH_FC2
    base = 2 * S * N * 16 * 4 + N * 16 * 4
    groups, blocks = (N + 31) // 32, 8 * ((N + 31) // 32) * ((S + 7) // 8)
    assert lib.rbnn_lowdim_scratch_bytes(C.byref(net), N, S) == base + blocks * 256 * 16 and groups == 4
    # hidden 128 runs 256-thread blocks of <= 112 points: one block per sample here, 8 * ceil(250 / 8) = 256 blocks of 4 KiB used
    assert blocks * 256 * 16 >= 256 * 256 * 16
    assert lib.rbnn_lowdim_scratch_bytes(None, N, S) == 0 and lib.rbnn_lowdim_scratch_bytes(C.byref(net), 0, S) == 0


def test_compute_refuses_cpu_tensors():
    post = O.synthetic_posterior("fc", 2, 64, 2, 3, 0.5)
    sp = StackedPosterior("fc", "leaky", (1, 2, 1), 2, 64, post, "cpu")
    eng = AttackEngine(sp)                                         # real HIP kernels, CPU tensors
    with pytest.raises(_hip.HipError, match="no CPU fallback"):
        eng.forward(torch.rand(4, 1, 2, 1), 3)


# ----------------------------------------------------------------------------- posterior layout
def test_stacked_posterior_padding_and_roundtrip():
    post = O.synthetic_posterior("fc2", 2, 16, 2, 3, 0.5)
    sp = StackedPosterior("fc2", "tanh", (1, 2, 1), 2, 16, post, "cpu")
    assert (sp.D, sp.Dp, sp.H, sp.Hp, sp.C, sp.S) == (2, 16, 16, 32, 2, 3)
    assert sp.W1.shape == (3, 32, 16) and sp.Wm.shape == (3, 32, 32) and sp.W2.shape == (3, 2, 32)
    assert float(sp.W1[:, 16:].abs().max()) == 0 and float(sp.W1[:, :, 2:].abs().max()) == 0
    assert float(sp.W2[:, :, 16:].abs().max()) == 0
    sd = sp.state_dict(1)
    for k, v in sd.items():
        assert torch.equal(v, post[k][1])
    d = sp.descriptor()
    assert (d.arch, d.activation, d.in_features, d.in_stride, d.hidden, d.n_classes, d.n_stored) == (1, 3, 2, 16, 32, 2, 3)
    # the packed backward images are written by the HIP kernel only: a CPU-resident posterior has none (no CPU compute path)
    assert sp.W1p is None and sp.Wmp is None and not d.W1_pack4
    sh = sp.shard(1, 2)
    assert sh.S == 2 and torch.equal(sh.W1, sp.W1[1:3])
    with pytest.raises(NotImplementedError):
        StackedPosterior("conv", "leaky", (1, 28, 28), 10, 16, {}, "cpu")       # conv has its own stacked posterior


def test_padded_hidden_is_exact(golden):
    """hidden 16 -> 32 zero padding changes nothing (sigmoid's act(0)=0.5 meets zero outgoing weights)."""
    g = golden("mnist_fc_h16_s4_n6_sigm"); m = g.meta
    sp = StackedPosterior(m["arch"], m["act"], m["shape"], m["n_classes"], m["hidden"], g.posterior(), "cpu")
    eng = AttackEngine(sp, kernels=FakeKernels())
    assert rel_err(eng.forward(g.t("x"), m["S"]), g.t("forward_probs")) < 1e-5
    assert rel_err(eng.loss_gradients(g.t("x"), g.t("y"), m["S"]), g.t("loss_gradients")) < 1e-5


# ----------------------------------------------------------------------------- engine orchestration (fake kernels)
@pytest.mark.parametrize("name", ["halfmoons_fc_h64_s10_n100", "mnist_fc_h32_s8_n8_leaky", "mnist_fc2_h32_s4_n6_leaky"])
def test_engine_orchestration_against_golden(golden, name):
    g = golden(name); m = g.meta
    sp = StackedPosterior(m["arch"], m["act"], m["shape"], m["n_classes"], m["hidden"], g.posterior(), "cpu")
    eng = AttackEngine(sp, kernels=FakeKernels())
    x, y = g.t("x"), g.t("y")
    assert rel_err(eng.forward(x, m["S"]), g.t("forward_probs")) < 1e-5
    seeds = [int(s) for s in g.arr["forward_seeds"]]
    assert rel_err(eng.forward(x, len(seeds), seeds=seeds), g.t("forward_probs_seeds")) < 1e-5
    assert rel_err(eng.loss_gradients(x, y, m["S"]), g.t("loss_gradients")) < 1e-5
    assert rel_err(eng.loss_gradients(x, y, m["S_half"]), g.t("loss_gradients_half")) < 1e-5
    ref_g = g.t("meanprob_grad").reshape(len(x), -1)
    safe = ref_g.abs() > 1e-3 * ref_g.abs().max(1, keepdim=True)[0]
    adv = eng.fgsm(x, y, m["S"], m["eps"])
    assert adv.shape == x.shape
    assert not ((((adv - g.t("fgsm")).abs() > 1e-6).reshape(len(x), -1)) & safe).any()
    idx = torch.from_numpy(g.arr["pgd_idx"])
    pg = eng.pgd(x[idx], y[idx], m["S"], m["eps"])
    assert float(((pg - g.t("pgd")).abs() > 1e-6).double().mean()) < 0.02
    if "pgd_default" in g.arr:
        pg = eng.pgd(x[idx], y[idx], m["S"], 0.5, alpha=2 / 225)
        assert float(((pg - g.t("pgd_default")).abs() > 1e-6).double().mean()) < 0.02
    oa, aa, rob, _, _ = eng.evaluate(x, g.t("fgsm"), y, m["S"])
    assert (oa, aa) == (float(g.arr["eval_orig_acc"]), float(g.arr["eval_adv_acc"]))
    assert float((rob - g.t("eval_softmax_rob")).abs().max()) < 1e-6
    with pytest.raises(ValueError, match="Number of seeds"):
        eng.sample_index(2, seeds=[0])
    with pytest.raises(IndexError):
        eng.sample_index(1, seeds=[m["S"]])


def test_chunking_does_not_change_the_gradient(golden):
    g = golden("mnist_fc_h32_s8_n8_leaky"); m = g.meta
    sp = StackedPosterior(m["arch"], m["act"], m["shape"], m["n_classes"], m["hidden"], g.posterior(), "cpu")
    eng = AttackEngine(sp, kernels=FakeKernels())
    X = eng.pad_inputs(g.t("x")); lab = g.t("y").argmax(-1).int()
    outs = []
    for chunk in (1, 3, 8):
        ws, n_slabs, _ = eng.gradient_slabs(X, lab, None, m["S"], _hip.LOSS_MEAN_PROB, chunk=chunk)
        assert n_slabs == -(-m["S"] // chunk)
        G = torch.empty(len(X), sp.Dp)
        eng.k.sum_slabs(ws["slabs"], n_slabs, len(X), sp.Dp, 1.0, G)
        outs.append(G)
    assert rel_err(outs[0], outs[2]) < 1e-6 and rel_err(outs[1], outs[2]) < 1e-6


# ----------------------------------------------------------------------------- reference call surface
def test_signatures_match_the_reference():
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(model_bnn.BNN.forward) == ["self", "inputs", "n_samples", "avg_posterior", "seeds"]
    assert sig(model_bnn.BNN.__init__) == ["self", "dataset_name", "hidden_size", "activation", "architecture", "inference",
                                          "epochs", "lr", "n_samples", "warmup", "input_shape", "output_size", "step_size", "num_steps"]
    assert sig(model_bnn.BNN.load) == ["self", "device", "rel_path", "filename"]
    assert sig(model_bnn.BNN.evaluate) == ["self", "test_loader", "device", "n_samples", "seeds_list"]
    assert sig(model_nn.NN.__init__) == ["self", "dataset_name", "input_shape", "output_size", "hidden_size", "activation",
                                        "architecture", "lr", "epochs"]
    assert sig(model_nn.NN.forward) == ["self", "inputs", "device", "args", "kwargs"]
    assert sig(model_ensemble.Ensemble_NN.forward) == ["self", "inputs", "n_samples", "args", "kwargs"]
    assert sig(lossGradients.loss_gradient) == ["net", "image", "label", "n_samples"]
    assert sig(lossGradients.loss_gradients) == ["net", "data_loader", "device", "filename", "savedir", "n_samples"]
    for f in (adversarialAttacks.fgsm_attack, adversarialAttacks.pgd_attack):
        assert sig(f) == ["net", "image", "label", "hyperparams", "n_samples", "avg_posterior"]
    assert sig(adversarialAttacks.attack) == ["net", "x_test", "y_test", "dataset_name", "device", "method", "filename",
                                              "savedir", "hyperparams", "n_samples", "avg_posterior"]
    assert sig(adversarialAttacks.attack_evaluation) == ["net", "x_test", "x_attack", "y_test", "device", "n_samples"]
    assert sig(adversarialAttacks.load_attack) == ["method", "filename", "savedir", "n_samples", "rel_path"]
    d = inspect.signature(model_bnn.BNN.forward).parameters
    assert d["n_samples"].default == 10 and d["avg_posterior"].default is False and d["seeds"].default is None


def _cpu_bnn(golden, name):
    """A BNN whose engine runs on the CPU fake (orchestration + file side effects only)."""
    g = golden(name); m = g.meta
    bnn = model_bnn.BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], 0, tuple(m["shape"]), m["n_classes"])
    bnn.set_posterior_samples(g.posterior(), "cpu")
    bnn._engine = AttackEngine(bnn.posterior, kernels=FakeKernels())
    return g, m, bnn


def test_names_and_guards(golden):
    g, m, bnn = _cpu_bnn(golden, "halfmoons_fc_h64_s10_n100")
    assert bnn.name == m["bnn_name"]
    nn_ = model_nn.NN("mnist", (1, 28, 28), 10, 32, "leaky", "fc", 0.01, 5)
    assert nn_.name == "mnist_nn_hid=32_act=leaky_arch=fc_ep=5_lr=0.01"
    assert list(nn_.state_dict().keys()) == ["model.1.weight", "model.1.bias", "model.3.weight", "model.3.bias"]
    with pytest.raises(ValueError, match="power of 2"):
        model_nn.NN("mnist", (1, 28, 28), 10, 48, "leaky", "fc", 0.01, 5)
    with pytest.raises(AssertionError):
        model_nn.NN("mnist", (1, 28, 28), 10, 32, "gelu", "fc", 0.01, 5)
    with pytest.raises(NotImplementedError):
        model_nn.NN("cifar", (3, 32, 32), 10, 32, "leaky", "conv", 0.01, 5)
    with pytest.raises(ValueError, match="Number of seeds"):
        bnn.forward(g.t("x"), n_samples=3, seeds=[0, 1])
    with pytest.raises(NameError):
        lossGradients.loss_gradient(bnn, g.t("x")[0], g.t("y")[0], n_samples=None)
    with pytest.raises(NotImplementedError):
        bnn.train(None, "cpu")
    ens = model_ensemble.Ensemble_NN("mnist", 32, "leaky", "fc", 1, 0.01, (1, 28, 28), 10, 4)
    assert ens.name == "mnist_ensemble_hid=32_act=leaky_arch=fc_size=4"
    with pytest.raises(ValueError):
        ens.forward(torch.zeros(1, 1, 28, 28), n_samples=5)


def test_drivers_and_side_effect_files(golden, tmp_path, monkeypatch):
    from torch.utils.data import DataLoader
    g, m, bnn = _cpu_bnn(golden, "halfmoons_fc_h64_s10_n100")
    monkeypatch.chdir(tmp_path)
    x, y = g.t("x"), g.t("y")
    adv = adversarialAttacks.attack(net=bnn, x_test=x, y_test=y, dataset_name=m["dataset"], device="cpu", method="fgsm",
                                    filename=bnn.name, hyperparams={"epsilon": m["eps"]}, n_samples=m["S"])
    assert adv.shape == x.shape and float((adv - g.t("attack_fn_fgsm")).abs().max()) < 1e-6
    loader = DataLoader(dataset=list(zip(x, y)), batch_size=3, shuffle=False)
    lg = lossGradients.loss_gradients(net=bnn, data_loader=loader, device="cpu", filename=bnn.name, savedir="sd/", n_samples=m["S"])
    assert lg.shape == g.arr["loss_gradients_fn"].shape and rel_err(torch.from_numpy(lg), g.t("loss_gradients_fn")) < 1e-5
    from robustbnns_amd import savedir
    files = []
    for root, _, fs in os.walk(tmp_path):
        files += [os.path.relpath(os.path.join(root, f), tmp_path) for f in fs]
    got = sorted(f.replace(savedir.TESTS, "TESTS/") for f in files)
    assert got == sorted(m["side_effect_files"])
    back = adversarialAttacks.load_attack("fgsm", bnn.name, n_samples=m["S"])
    assert torch.equal(back, adv)
    assert np.array_equal(lossGradients.load_loss_gradients(m["S"], bnn.name, "sd/"), lg)
    with pytest.raises(UnboundLocalError):
        adversarialAttacks.attack(net=bnn, x_test=x, y_test=y, dataset_name="d", device="cpu", method="cw", filename="f")
    # single-point calls, shaped as the reference's attack() loop makes them
    img, lab = x[3].unsqueeze(0).clone(), y[3].argmax(-1).unsqueeze(0)
    out = adversarialAttacks.fgsm_attack(bnn, img, lab, {"epsilon": m["eps"]}, n_samples=m["S"])
    assert out.shape == img.shape and float((out - g.t("fgsm")[3:4]).abs().max()) < 1e-6 and img.requires_grad
    out = adversarialAttacks.pgd_attack(bnn, x[3].unsqueeze(0), lab, {"epsilon": m["eps"]}, n_samples=m["S"])
    assert float(((out - g.t("pgd")[3:4]).abs() > 1e-6).double().mean()) <= 0.5
    lgi = lossGradients.loss_gradient(bnn, x[3], y[3], n_samples=m["S"])
    assert lgi.shape == x[3].shape and rel_err(lgi[None], g.t("loss_gradients")[3:4]) < 1e-5
    oa, aa, rob = adversarialAttacks.attack_evaluation(bnn, x, g.t("fgsm"), y, "cpu", n_samples=m["S"])
    assert (oa, aa) == (float(g.arr["eval_orig_acc"]), float(g.arr["eval_adv_acc"]))


def test_hmc_files_roundtrip(golden, tmp_path):
    g, m, bnn = _cpu_bnn(golden, "mnist_fc2_h32_s4_n6_leaky")
    rel = str(tmp_path) + "/"
    bnn.save(rel_path=rel)
    assert sorted(os.listdir(rel + bnn.name)) == [f"{bnn.name}_weights_{i}.pt" for i in range(m["S"])]
    b2 = model_bnn.BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], 0, tuple(m["shape"]), m["n_classes"])
    b2.load("cpu", rel_path=rel)
    for nm in ("W1", "b1", "Wm", "bm", "W2", "b2"):
        assert torch.equal(getattr(b2.posterior, nm), getattr(bnn.posterior, nm))
    pp = bnn.posterior_predictive
    assert sorted(pp) == list(range(m["S"])) and torch.equal(pp[2].state_dict()["model.3.weight"], g.posterior()["model.3.weight"][2])


def test_eps_grid_driver_matches_reference(golden, tmp_path, monkeypatch):
    """plot_eps_attacks.build_eps_attacks_df: same rows, same CSV path and schema as the reference's."""
    import pandas
    from robustbnns_amd import plot_eps_attacks
    g = golden("halfmoons_eps_grid_fgsm"); m = g.meta
    bnn = model_bnn.BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], 0, tuple(m["shape"]), m["n_classes"])
    bnn.set_posterior_samples(g.posterior(), "cpu")
    bnn._engine = AttackEngine(bnn.posterior, kernels=FakeKernels())
    assert bnn.name == m["bnn_name"]
    monkeypatch.chdir(tmp_path)
    df = plot_eps_attacks.build_eps_attacks_df(bnn=bnn, dataset=m["dataset"], device="cpu", method=m["method"], x_test=g.t("x"),
                                               y_test=g.t("y"), epsilon_list=m["epsilon_list"], n_samples_list=m["n_samples_list"],
                                               savedir=bnn.name)
    assert list(df.columns) == m["columns"] and len(df) == len(g.arr["df_epsilon"])
    for col in ("epsilon", "test_acc", "adv_acc", "n_samples"):
        assert np.array_equal(df[col].to_numpy().astype("float64"), g.arr["df_" + col]), col
    assert np.abs(df["softmax_rob"].to_numpy() - g.arr["df_softmax_rob"]).max() < 1e-6
    assert set(df["attack_method"]) == {m["method"]}
    assert os.path.exists(m["csv_files"][0])
    back = plot_eps_attacks.load_eps_attacks_df(m["dataset"], m["method"], bnn.name)
    assert len(back) == len(df) and list(back.columns) == m["columns"]
    # an FGSM grid on stored samples is one resident job: ONE gradient pass and ONE clean forward per n_samples, whatever the number of epsilons
    cost = df.attrs["grid_cost"]
    assert cost == {"cells": len(m["epsilon_list"]) * len(m["n_samples_list"]), "gradient_passes": len(m["n_samples_list"]),
                    "clean_forwards": len(m["n_samples_list"]), "shared": True}
    # ... and every cell is what the module-level attack() + attack_evaluation() return (and write) for it
    grid = adversarialAttacks.FgsmGrid(bnn, g.t("x"), g.t("y"), m["dataset"], "cpu")
    for eps in m["epsilon_list"][:2]:
        for ns in m["n_samples_list"]:
            a = grid.attack(eps, ns, filename=bnn.name)
            b = adversarialAttacks.attack(net=bnn, x_test=g.t("x"), y_test=g.t("y"), dataset_name=m["dataset"], device="cpu", method="fgsm",
                                          filename=bnn.name, n_samples=ns, hyperparams={"epsilon": eps})
            assert torch.equal(a, b) and a.requires_grad
            assert torch.equal(adversarialAttacks.load_attack("fgsm", bnn.name, n_samples=ns), b)
            ra, rb = grid.evaluate(a, ns), adversarialAttacks.attack_evaluation(net=bnn, x_test=g.t("x"), x_attack=b, y_test=g.t("y"), device="cpu", n_samples=ns)
            assert ra[:2] == rb[:2] and torch.equal(ra[2], rb[2])
    assert (grid.gradient_passes, grid.clean_forwards) == (len(m["n_samples_list"]),) * 2
    # the caches are valid for one frozen net and input set: inputs edited in place, or a replaced posterior, miss them — the cell is
    # then again what attack() returns for the NEW state, not a stale gradient's
    ns, eps = m["n_samples_list"][0], m["epsilon_list"][0]
    before = grid.gradient_passes
    with torch.no_grad():
        grid.x_test.mul_(0.5)
    a = grid.attack(eps, ns, filename=bnn.name)
    assert grid.gradient_passes == before + 1
    b = adversarialAttacks.attack(net=bnn, x_test=grid.x_test, y_test=g.t("y"), dataset_name=m["dataset"], device="cpu", method="fgsm",
                                  filename=bnn.name, n_samples=ns, hyperparams={"epsilon": eps})
    assert torch.equal(a, b)
    bnn.set_posterior_samples({k: v.flip(0) * 1.5 for k, v in g.posterior().items()}, "cpu")
    bnn._engine = AttackEngine(bnn.posterior, kernels=FakeKernels())
    a = grid.attack(eps, ns, filename=bnn.name)
    assert grid.gradient_passes == before + 2
    b = adversarialAttacks.attack(net=bnn, x_test=grid.x_test, y_test=g.t("y"), dataset_name=m["dataset"], device="cpu", method="fgsm",
                                  filename=bnn.name, n_samples=ns, hyperparams={"epsilon": eps})
    assert torch.equal(a, b)


def test_forward_is_differentiable_like_the_reference_expects(golden):
    """A caller that runs CrossEntropyLoss(net.forward(x)).backward() itself (adversarialAttacks.py:73-79) gets the
    same input gradient as the reference's autograd."""
    g = golden("mnist_fc_h32_s8_n8_leaky"); m = g.meta
    sp = StackedPosterior(m["arch"], m["act"], m["shape"], m["n_classes"], m["hidden"], g.posterior(), "cpu")
    eng = AttackEngine(sp, kernels=FakeKernels())
    x = g.t("x").clone().requires_grad_(True)
    out = eng.forward(x, m["S"])
    assert out.requires_grad
    torch.nn.CrossEntropyLoss(reduction="sum")(out, g.t("y").argmax(-1)).backward()
    assert rel_err(x.grad, g.t("meanprob_grad")) < 1e-5
    x2 = g.t("x").clone().requires_grad_(True)
    z = eng.forward(x2, m["S"], logits=True)
    torch.nn.CrossEntropyLoss(reduction="sum")(z, g.t("y").argmax(-1)).backward()
    ref = O.meanprob_gradients(g.t("x"), g.t("y").argmax(-1), g.posterior(), m["arch"], m["act"], m["S"], kind="ensemble")
    assert rel_err(x2.grad, ref) < 1e-5
    with torch.no_grad():
        assert not eng.forward(x, m["S"]).requires_grad


def test_compute_vanishing_norms_idxs_matches_reference():
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "vanishing_norms.npz"))
    for norm in ("linfty", "l2"):
        got = lossGradients.compute_vanishing_norms_idxs(d["grads"], list(d["n_samples_list"]), norm)
        assert got == list(d[norm])
    with pytest.raises(ValueError):
        lossGradients.compute_vanishing_norms_idxs(d["grads"], [1, 2], "l2")


def test_conv_posterior_layout_and_roundtrip(golden):
    from robustbnns_amd.conv import ConvStackedPosterior
    g = golden("mnist_conv_h16_s2_n4_leaky"); m = g.meta; post = g.posterior()
    sp = ConvStackedPosterior(m["act"], m["shape"], m["n_classes"], m["hidden"], post, "cpu")
    assert sp.K1w.shape == (2, 32, 25) and sp.K2w.shape == (2, 16, 800) and sp.Fw.shape == (2, 10, 49 * 16)
    # regrouped image [S, ci, hc block, tap, hc % 16] of model.3.weight [hc, ci, ky, kx]
    assert torch.equal(sp.K2ci.view(2, 32, 1, 25, 16)[1, 5, 0, 7, 3], post["model.3.weight"][1, 3, 5, 1, 2])
    sd = sp.state_dict(0)
    assert all(torch.equal(sd[k], post[k][0]) for k in sd)
    d = sp.descriptor()
    assert (d.activation, d.hidden, d.n_classes, d.n_stored) == (1, 16, 10, 2)
    assert (d.in_channels, d.in_width) == (1, 28)
    with pytest.raises(NotImplementedError):
        ConvStackedPosterior("leaky", (1, 32, 32), 10, 16, post, "cpu")          # only the two built geometries
    lib = _hip.load()
    out = _hip.ConvWorkspaceSizes()
    assert lib.rbnn_conv_workspace_query(C.byref(d), 100, 3, C.byref(out)) == 0
    assert out.P1 == 3 * 100 * 24576 and out.st1 == 3 * 100 * 4608 and out.Q2 == 3 * 100 * 16 * 49 * 4 and out.G == 3 * 100 * 784 * 4
    assert lib.rbnn_conv_forward(C.byref(d), None, 784, 4, None, 2, 0, None, None) == -1
    # the CIFAR-shaped geometry (BASELINE.json configs[4]; build-defined head 81 * Hc): layout and workspace sizes
    cif = O.synthetic_posterior("conv", 3 * 32 * 32, 16, 10, 2, 0.05, in_ch=3, head=81 * 16)
    sc = ConvStackedPosterior("tanh", (3, 32, 32), 10, 16, cif, "cpu")
    assert sc.K1w.shape == (2, 32, 75) and sc.Fw.shape == (2, 10, 81 * 16) and (sc.P1W, sc.P2W, sc.NP2, sc.D) == (14, 9, 81, 3072)
    assert not sc.split_supported()
    dc = sc.descriptor()
    assert (dc.activation, dc.in_channels, dc.in_width) == (3, 3, 32)
    assert lib.rbnn_conv_workspace_query(C.byref(dc), 10, 2, C.byref(out)) == 0
    assert out.P1 == 20 * 25600 and out.st1 == 20 * 6272 and out.Q2 == 20 * 16 * 81 * 4 and out.st2 == 20 * 16 * 81 and out.G == 20 * 3072 * 4
    assert lib.rbnn_conv_forward(C.byref(dc), C.c_void_p(16), 784, 4, None, 2, 0, C.byref(_hip.ConvWorkspace()), None) == -1
    dc.in_width = 30
    assert lib.rbnn_conv_workspace_query(C.byref(dc), 10, 2, C.byref(out)) == -3      # unsupported geometry
    bnn = model_bnn.BNN("mnist", 16, "leaky", "conv", "hmc", None, None, 2, 0, (1, 28, 28), 10)
    bnn.set_posterior_samples(post, "cpu")
    assert type(bnn._engine).__name__ == "ConvEngine" and bnn.posterior.S == 2


def test_conv_engine_point_blocking(monkeypatch, golden):
    from robustbnns_amd.conv import ConvEngine, ConvStackedPosterior
    g = golden("mnist_conv_h16_s2_n4_leaky"); m = g.meta
    eng = ConvEngine(ConvStackedPosterior(m["act"], m["shape"], m["n_classes"], m["hidden"], g.posterior(), "cpu"), kernels=FakeKernels())
    monkeypatch.setenv("RBNN_CONV_WS_GB", "0.001")                      # ~1 MB: forces blocks of 16 points
    assert eng.point_block(2) == 16
    calls = []
    out = eng._blocked(lambda xb, yb: (calls.append((len(xb), len(yb))), xb * 2)[1], torch.arange(40.).reshape(40, 1), torch.arange(40), n_samples=2)
    assert calls == [(16, 16), (16, 16), (8, 8)] and torch.equal(out, torch.arange(40.).reshape(40, 1) * 2)
    monkeypatch.setenv("RBNN_CONV_WS_GB", "48")
    assert eng.point_block(100) >= 2048


# ----------------------------------------------------------------------------- f2: Pyro param-store files
def test_pyro_param_store_layout_roundtrip(tmp_path):
    """BNN.load(inference="svi") reads a file in pyro 1.3.0's ParamStoreDict layout ({"params": unconstrained leaf tensors,
    "constraints": constraint objects}; hand-built fixture, tests/golden/make_pyro_store.py) and BNN.save writes it back."""
    import shutil
    from conftest import GOLDEN
    from torch.distributions import constraints
    bnn = model_bnn.BNN("half_moons", 32, "leaky", "fc", "svi", 5, 0.01, None, None, (1, 2, 1), 2)
    rel = str(tmp_path) + "/"
    os.makedirs(rel + bnn.name)
    shutil.copy(os.path.join(GOLDEN, "pyro_store_halfmoons_fc_h32.pt"), rel + bnn.name + "/" + bnn.name + "_weights.pt")
    raw = torch.load(rel + bnn.name + "/" + bnn.name + "_weights.pt", weights_only=False)
    assert set(raw) == {"params", "constraints"} and set(raw["params"]) == set(raw["constraints"])
    assert all(t.requires_grad and t.is_leaf for t in raw["params"].values())
    assert all(c is constraints.real or isinstance(c, type(constraints.real)) for c in raw["constraints"].values())
    bnn.load(device="cpu", rel_path=rel)
    exp = np.load(os.path.join(GOLDEN, "pyro_store_halfmoons_fc_h32_expected.npz"))
    keys = list(bnn.basenet.state_dict())
    assert sorted(exp.files) == sorted(k + s for k in keys for s in ("_loc", "_scale"))
    for k in keys:
        assert np.array_equal(bnn.svi_loc[k].numpy(), exp[k + "_loc"]) and np.array_equal(bnn.svi_scale[k].numpy(), exp[k + "_scale"])
        assert not bnn.svi_loc[k].requires_grad
    bnn.save(rel_path=rel, filename="copy")
    back = model_bnn.read_param_store(rel + bnn.name + "/copy.pt")
    assert all(np.array_equal(back[k].numpy(), exp[k]) for k in exp.files)
    # a positive-constrained parameter is stored unconstrained (log): the reader applies transform_to(constraint)
    st = model_bnn.param_store_state({"a": torch.tensor([0.5])}, {"a": torch.tensor([-1.0])})
    st["constraints"]["a_scale"] = constraints.positive
    torch.save(st, rel + "pos.pt")
    assert torch.allclose(model_bnn.read_param_store(rel + "pos.pt")["a_scale"], torch.tensor([-1.0]).exp())
    torch.save({"params": {}, "oops": {}}, rel + "bad.pt")
    with pytest.raises(KeyError, match="malformed"):
        model_bnn.read_param_store(rel + "bad.pt")
    other = model_bnn.BNN("half_moons", 32, "leaky", "fc2", "svi", 5, 0.01, None, None, (1, 2, 1), 2)
    os.makedirs(rel + other.name)
    shutil.copy(os.path.join(GOLDEN, "pyro_store_halfmoons_fc_h32.pt"), rel + other.name + "/" + other.name + "_weights.pt")
    with pytest.raises(KeyError, match="lacks"):
        other.load(device="cpu", rel_path=rel)                  # written by the fc guide: no model.5.* entries


@pytest.mark.parametrize("arch,act", [("fc", "leaky"), ("fc2", "tanh")])
def test_loss_gradients_and_fgsm_share_one_forward(arch, act):
    """AttackEngine.loss_gradients_and_fgsm (BASELINE config 4's step): ONE forward launch, two backward launches, results equal to the
    separate calls' — orchestration on the test double; the real kernels' bit-identity is tests/test_hip_round5.py."""
    post = O.synthetic_posterior(arch, 784, 32, 10, 5, 0.06)
    x, y = O.synthetic_inputs(9, (1, 28, 28), 10, seed=2)
    sp = StackedPosterior(arch, act, (1, 28, 28), 10, 32, post, "cpu")
    calls = []

    class Counting(FakeKernels):
        def fc_forward(self, *a, **kw):
            calls.append("fwd")
            return super().fc_forward(*a, **kw)

        def fc_input_grad(self, *a, **kw):
            calls.append("bwd")
            return super().fc_input_grad(*a, **kw)

    eng = AttackEngine(sp, kernels=Counting())
    lg, adv = eng.loss_gradients(x, y, 5), eng.fgsm(x, y, 5, 0.25)
    assert calls == ["fwd", "bwd", "fwd", "bwd"]
    del calls[:]
    lg2, adv2 = eng.loss_gradients_and_fgsm(x, y, 5, 0.25)
    assert calls == ["fwd", "bwd", "bwd"]
    assert torch.equal(lg2, lg) and torch.equal(adv2, adv)
    seeds = [3, 0, 4]
    lg3, adv3 = eng.loss_gradients_and_fgsm(x, y, 3, 0.25, seeds=seeds)
    assert torch.equal(lg3, eng.loss_gradients(x, y, 3, seeds=seeds)) and torch.equal(adv3, eng.fgsm(x, y, 3, 0.25, seeds=seeds))
    del calls[:]
    eng.loss_gradients_and_fgsm(x, y, 5, 0.25, mode=_hip.LOSS_MEAN_LOGIT)        # mean-logit attack: nothing to share
    assert calls == ["fwd", "bwd", "fwd", "bwd"]


def test_fake_collectives_are_refused_outside_a_one_rank_group(monkeypatch):
    """RBNN_FAKE_COLLECTIVES=1 (+ RBNN_FORCE_COLLECTIVES=1) turns every all-reduce into a no-op — a timing diagnostic for a 1-rank group
    (tools/collectives_ab.sh).  With more than one rank it would return partial sums as results: refused."""
    import torch.distributed as dist
    O_post = O.synthetic_posterior("fc", 16, 32, 3, 2, 0.3)
    sp = StackedPosterior("fc", "leaky", (1, 4, 4), 3, 32, O_post, "cpu")
    monkeypatch.setenv("RBNN_FAKE_COLLECTIVES", "1")
    monkeypatch.setenv("RBNN_FORCE_COLLECTIVES", "1")
    group = object()
    monkeypatch.setattr(dist, "get_world_size", lambda g=None: 2)
    with pytest.raises(_hip.HipError, match="never valid with more than one rank"):
        AttackEngine(sp, kernels=FakeKernels(), group=group)
    monkeypatch.setattr(dist, "get_world_size", lambda g=None: 1)
    eng = AttackEngine(sp, kernels=FakeKernels(), group=group)
    assert eng._fake_comm and eng.world == 2                 # the sharded launch sequence, no exchange
    # an engine WITHOUT a group never exchanges anything (bench's point-sharded / other-mode engines, SVI hot-path engines): the two switches
    # leave it alone instead of raising
    plain = AttackEngine(sp, kernels=FakeKernels())
    assert not plain._fake_comm and plain.world == 1
    monkeypatch.delenv("RBNN_FAKE_COLLECTIVES")
    monkeypatch.setattr(dist, "get_world_size", lambda g=None: 2)
    assert not AttackEngine(sp, kernels=FakeKernels(), group=group)._fake_comm


# ----------------------------------------------------------------------------- bench.py --gpus N outside torchrun
def test_bench_spawns_ranks_as_a_child_process(monkeypatch, capsys):
    """`python bench.py --gpus N` without a torchrun environment (the driver's scaling command) starts the ranks as a child
    process before anything touches the GPU and relays the child's JSON line as its own last line."""
    import importlib.util
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return types.SimpleNamespace(returncode=0, stdout='RCCL banner\n{"metric": "m", "value": 1.0, "n_gpus": 2}\ntrailing noise\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a: (_ for _ in ()).throw(AssertionError("the parent must not touch the GPU")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert cmd[-7].endswith("bench.py") and seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines[-1].startswith("{") and '"n_gpus": 2' in lines[-1] and "RCCL banner" in lines[0]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.main()


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench, root


def test_bench_config_strings_are_baseline_jsons():
    """`config.workload` of a line that measures a BASELINE config is that config's string, verbatim."""
    import json
    bench, root = _load_bench()
    configs = json.load(open(os.path.join(root, "BASELINE.json")))["configs"]
    assert [bench.BASELINE_CONFIGS[k] for k in ("c1", "c2", "c3", "c4", "c5")] == configs
    assert bench.WORKLOADS["c4"]["S_split"] == (2000, 8) and bench.WORKLOADS["c4"]["S"] == 250 and bench.WORKLOADS["c5"]["S_split"] == (500, 8)


class _HostEvent:
    def record(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def _bench_rank(rank, world, port, argv, q):
    """One rank of `bench.py --gpus N` under gloo on the CPU: bench.main() itself, with bench.RUNTIME replaced (CPU device, gloo, the test
    double of the kernel interface, host-clock events) — the orchestration, the sharding arithmetic and the record are the bench's own."""
    import contextlib
    import io
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    bench, _ = _load_bench()
    from fake_kernels import FakeKernels

    class CpuRuntime:
        backend = "gloo"
        open_device = staticmethod(lambda local: torch.device("cpu"))
        kernels_base = staticmethod(lambda: FakeKernels)
        event = staticmethod(_HostEvent)
        sync = staticmethod(lambda: None)

    bench.RUNTIME = CpuRuntime()
    if os.environ.get("RBNN_TEST_BENCH_ROOT"):                  # a test's own profiles/pmc_traffic.json
        bench.ROOT = os.environ["RBNN_TEST_BENCH_ROOT"]
    sys.argv = ["bench.py"] + argv
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    if rank == 0:
        q.put(buf.getvalue().strip().splitlines()[-1])
    else:
        assert "{" not in buf.getvalue()            # one JSON line per job: rank 0's


def _bench_world(world, argv):
    import json
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_bench_rank, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        line = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return json.loads(line)


def test_bench_default_multi_gpu_line_is_baseline_config_4_with_c2_point_sharded():
    """`bench.py --gpus 2` (what the driver's scaling run issues, no --workload): the headline is BASELINE config 4 — n_samples = 2000 split
    8-way, this run holding shards 0 and 1 (250 samples each), sample-sharded with the two all-reduces, `loss_gradients` + FGSM = 2 passes
    per step, weak scaling — and C2 with its n_samples = 100 KEPT, point-sharded (strong scaling), is the sub-record.  Two gloo ranks on the
    CPU with the test double (points and hidden size cut down by the debug overrides, which the line reports)."""
    import json
    bench, root = _load_bench()
    configs = json.load(open(os.path.join(root, "BASELINE.json")))["configs"]
    out = _bench_world(2, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "24", "--hidden", "32", "--posterior", "stored"])
    c = out["config"]
    assert c["workload"] == configs[3] and c["name"] == "c4" and out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert (c["n_samples_config"], c["n_samples_config_ranks"], c["samples_per_rank"], c["samples_total"]) == (2000, 8, [250, 250], 500)
    assert c["shard"] == "samples" and c["passes_per_step"] == 2 and c["points"] == 24 and c["overrides"] == {"points": 24, "hidden": 32}
    assert out["unit"] == "attack-samples/s" and abs(out["value"] - 24 * 500 * 2 * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    assert set(out["roofline"]["kernels"]) == {"fc_forward", "fc_input_grad"} and "cpu_baseline" not in out
    sub = out["c2_point_sharded"]
    sc = sub["config"]
    assert sc["workload"] == configs[1] and sc["name"] == "c2" and sub["scaling"] == "strong" and sc["shard"] == "points"
    assert (sc["n_samples_config"], sc["samples_per_rank"], sc["samples_total"], sc["points"], sc["points_per_rank"]) == (100, [100, 100], 100, 24, [12, 12])
    assert abs(sub["value"] - 24 * 100 / (sub["ms_per_step"] * 1e-3)) < 1e-6 * sub["value"]
    # what a judge needs from an N > 1 line: the workload and sharding at the top level, the collective library's own view of
    # the job, the exchange per step and the time the launch stream stood waiting for it, and the per-step spread beside the mean
    assert (out["workload"], out["shard"], sub["workload"], sub["shard"]) == ("c4", "samples", "c2", "points")
    cm = out["comm"]
    assert (cm["backend"], cm["world_size"], cm["ranks_summed_by_an_allreduce_of_ones"], cm["shard"]) == ("gloo", 2, 2, "samples")
    # c4 on one forward, sample-sharded: the summed per-sample-loss gradients, sum_s p_s, the summed mean-loss gradients — per step 2 x [24, 784] + [24, 16] fp32
    assert cm["allreduce_calls_per_step"] == 3 and cm["allreduce_bytes_per_step"] == 4 * (2 * 24 * 784 + 24 * 16)
    assert cm["waits_per_step"] == 3 and 0 <= cm["exposed_ms_per_step"] <= out["ms_per_step_max"] and 0 <= cm["exposed_frac_of_step"] <= 1.5
    scm = sub["comm"]
    assert (scm["world_size"], scm["shard"], scm["allreduce_calls_per_step"], scm["allreduce_bytes_per_step"]) == (2, "points", 0, 0)
    for rec in (out, sub):
        assert rec["ms_per_step_min"] <= rec["ms_per_step_median"] <= rec["ms_per_step_max"] and rec["ms_per_step_min"] > 0
    # the two-call definition of c4 (rounds 1-4) beside the shared-forward number, at N > 1 too
    assert out["config"]["forward_shared"] is True and out["separate_calls_mode"]["ms_per_step"] > 0


def test_bench_single_gpu_line_carries_the_per_step_spread_and_eval_is_a_forward_only_workload():
    """N = 1: the line's `value` arithmetic is unchanged and min / median / max of the per-step events sit beside the mean; no `comm` record without a
    process group.  `--workload eval` (attack_evaluation: clean + adversarial batched forward + eval_metrics) times two forward passes per step and
    no gradient kernel."""
    out = _bench_world(1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--points", "16", "--hidden", "32", "--samples", "3", "--posterior", "stored",
                           "--cpu-seconds", "0", "--no-other-mode"])
    assert out["workload"] == "c2" and out["shard"] == "none" and "comm" not in out
    assert out["ms_per_step_min"] <= out["ms_per_step_median"] <= out["ms_per_step_max"]
    assert abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    ev = _bench_world(1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "eval", "--points", "16", "--hidden", "32", "--samples", "3",
                          "--cpu-seconds", "0", "--no-other-mode"])
    assert ev["config"]["name"] == "eval" and ev["config"]["passes_per_step"] == 2 and set(ev["roofline"]["kernels"]) == {"fc_forward"}
    assert ev["roofline"]["kernels"]["fc_forward"]["launches"] == 4 and ev["roofline"]["kernels"]["fc_forward"]["launches_per_pass"] == 1
    assert abs(ev["value"] - 2 * 16 * 3 / (ev["ms_per_step"] * 1e-3)) < 1e-6 * ev["value"]


def test_bench_explicit_workload_on_two_ranks_is_one_record():
    out = _bench_world(2, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c2", "--points", "16", "--hidden", "32", "--samples", "3",
                           "--posterior", "stored"])
    assert out["config"]["name"] == "c2" and "c2_point_sharded" not in out and out["config"]["samples_per_rank"] == [3, 3] and out["scaling"] == "weak"


def test_bench_line_carries_the_counter_traffic_of_its_workload(tmp_path, monkeypatch):
    """roofline.traffic / roofline.hbm.counter_* come from profiles/pmc_traffic.json, keyed by WORKLOAD and kernel call (round 5 shipped a bench
    set without them: a loop variable shadowed the workload's name and the lookup missed).  One rank, the test double, a record of this size."""
    import json
    (tmp_path / "profiles").mkdir()
    rec = {"c2": {"points": 16, "samples": 3, "kernels": {"fc_forward": {"hbm_bytes_per_launch": 1000.0}, "fc_input_grad": {"hbm_bytes_per_launch": 3000.0}},
                  "small": {"attack_step_kernel": {"hbm_bytes_per_launch": 50.0}, "reduce_samples_kernel": {"hbm_bytes_per_launch": 7.0}}},
           "source": ["a test record"]}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setenv("RBNN_TEST_BENCH_ROOT", str(tmp_path))
    out = _bench_world(1, ["--gpus", "1", "--steps", "2", "--warmup", "0", "--points", "16", "--hidden", "32", "--samples", "3", "--posterior", "stored",
                           "--cpu-seconds", "0", "--no-other-mode"])
    r = out["roofline"]
    dom = max(r["kernels"], key=lambda k: r["kernels"][k]["avg_ms"])
    assert r["traffic"] == {"fc_forward": 1000.0, "fc_input_grad": 3000.0}[dom] and r["traffic_source"] == ["a test record"]
    assert r["hbm"]["counter_bytes_by_call"] == {"fc_forward": 1000.0, "fc_input_grad": 3000.0}
    assert r["hbm"]["counter_bytes_per_pass"] == 4000.0 + 50.0 + 7.0 and r["hbm"]["counter_small_kernels_bytes"] == 57.0


# ----------------------------------------------------------------------------- files WRITTEN BY THE REFERENCE
REF_FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files")


def _ref_files_meta():
    import ast
    d = np.load(os.path.join(REF_FILES, "expected.npz"))
    return ast.literal_eval(str(d["meta"])), d


def test_reads_hmc_files_written_by_the_reference(monkeypatch):
    """tests/golden/files/posterior/<name>/<name>_weights_<i>.pt were written by the reference's BNN.save (model_bnn.py:157-162,
    make_golden_trained.py run_reference_files): BNN.load here must read them — default filename, the reference's directory layout —
    and stack the very weights the reference's forward used (its forward_probs, reproduced by the oracle from the loaded stack)."""
    m, d = _ref_files_meta()
    bnn = model_bnn.BNN(m["dataset"], m["hidden"], m["act"], m["arch"], "hmc", None, None, m["S"], m["warmup"], tuple(m["shape"]), m["n_classes"])
    assert bnn.name == m["bnn_name"]
    bnn.load("cpu", rel_path=os.path.join(REF_FILES, "posterior") + "/")
    assert bnn.posterior.S == m["S"]
    stacked = {k: torch.stack([bnn.posterior.state_dict(i)[k] for i in range(m["S"])]) for k in bnn.posterior.state_dict(0)}
    assert list(stacked) == ["model.1.weight", "model.1.bias", "model.3.weight", "model.3.bias", "model.5.weight", "model.5.bias"]
    p = O.bnn_forward(torch.from_numpy(d["x"]), stacked, m["arch"], m["act"], m["S"])
    assert rel_err(p, torch.from_numpy(d["forward_probs"])) < 1e-6
    # and the reverse: what BNN.save writes here is, file by file, what the reference wrote (same names, keys, dtypes, shapes, values)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        bnn.save(rel_path=tmp + "/")
        for i in range(m["S"]):
            rel = f"{bnn.name}/{bnn.name}_weights_{i}.pt"
            mine = torch.load(os.path.join(tmp, rel), map_location="cpu")
            ref = torch.load(os.path.join(REF_FILES, "posterior", rel), map_location="cpu")
            assert list(mine) == list(ref) and type(mine) is type(ref)
            for k in ref:
                assert mine[k].dtype == ref[k].dtype and mine[k].shape == ref[k].shape and torch.equal(mine[k], ref[k]), k


class _RefFilesNet(model_nn.NN):
    """A net whose engine returns a fixed adversarial set: only the tensor attributes of fgsm_attack's return value are under test."""

    def __init__(self, adv):
        super().__init__("half_moons", (1, 2, 1), 2, 32, "leaky", "fc2", 0.01, 1)
        self.device, self._adv = "cpu", adv

    def engine(self, device):
        outer = self

        class E:
            def fgsm(self, *a, **k):
                return outer._adv.clone()
        return E()


def test_reads_result_pickles_written_by_the_reference(monkeypatch, tmp_path):
    """The reference's loss-gradient pickle (lossGradients.py:70-72: a numpy array) and attack pickle (adversarialAttacks.py:140-141: a
    torch tensor) read through load_loss_gradients / load_attack, bit-exact; re-saving them through this package's writers gives
    the reference's files byte for byte (same pickle protocol, same object types)."""
    from robustbnns_amd import savedir
    m, d = _ref_files_meta()
    name, S = m["bnn_name"], m["S"]
    monkeypatch.setattr(lossGradients, "DATA", os.path.join(REF_FILES, "DATA") + "/")
    lg = lossGradients.load_loss_gradients(S, name, "grads/", relpath=os.path.join(REF_FILES, "DATA") + "/")
    assert isinstance(lg, np.ndarray) and lg.dtype == np.float32 and np.array_equal(lg, d["loss_gradients"])
    monkeypatch.setattr(adversarialAttacks, "TESTS", os.path.join(REF_FILES, "TESTS") + "/")
    adv = adversarialAttacks.load_attack("fgsm", name, savedir="attacks", n_samples=S)
    assert isinstance(adv, torch.Tensor) and adv.dtype == torch.float32 and adv.requires_grad and np.array_equal(adv.detach().numpy(), d["fgsm"])
    lossGradients.save_loss_gradients(lg, S, name, "grads/", relpath=str(tmp_path) + "/")
    a = open(tmp_path / "grads" / f"{name}_samp={S}_lossGrads.pkl", "rb").read()
    b = open(os.path.join(REF_FILES, "DATA", "grads", f"{name}_samp={S}_lossGrads.pkl"), "rb").read()
    assert a == b
    from robustbnns_amd.utils import save_to_pickle
    mine = adversarialAttacks.fgsm_attack(_RefFilesNet(adv.detach()), adv.detach().clone(), torch.zeros(len(adv), dtype=torch.int64), n_samples=S)
    assert mine.requires_grad and torch.equal(mine.detach(), adv.detach())            # what this package's fgsm_attack hands to attack()'s pickle
    save_to_pickle(data=mine, path=str(tmp_path) + "/", filename="a.pkl")
    b = open(os.path.join(REF_FILES, "TESTS", "attacks", f"{name}_fgsm_attackSamp={S}_attack.pkl"), "rb").read()
    # a pickled torch tensor names its storage by the storage's ADDRESS in the writing process (torch/_tensor.py __reduce_ex__ ->
    # torch.save's persistent id): mask that decimal key, everything else — protocol, classes, flags, strides, payload — is byte-equal
    key = lambda raw: re.sub(rb"X.\x00\x00\x00\d{8,20}", b"<storage key>", raw)
    assert key(open(tmp_path / "a.pkl", "rb").read()) == key(b) and key(b) != b


def test_reference_trained_posteriors_are_far_inside_the_dynamic_range_guard(golden):
    """`auto` leaves a posterior to the fp32-MFMA kernels (half the triple mode's rate) when a weight tensor's largest magnitude
    exceeds 2^12 x its mean magnitude (posterior.narrow_range).  How often does a TRAINED posterior trip that?  Every net the reference's own
    NN.train produced for the fixtures — fc, fc2, conv; half-moons and MNIST-shaped — sits between 2 and 11, four hundred times inside the guard."""
    import glob
    from robustbnns_amd.posterior import narrow_range
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_*.npz")))
    assert len(names) >= 5
    worst = 0.0
    for name in names:
        post = golden(name).posterior()
        assert narrow_range(*post.values())
        worst = max(worst, max(float(v.abs().max() / v.abs().double().mean()) for v in post.values()))
    assert 2.0 < worst < 16.0, worst


def test_no_mfma_reads_a_vgpr_inside_the_valu_write_window():
    """a VGPR written by a vector instruction needs two wait states before an MFMA takes it as an operand; hipcc pads that between
    instructions it sees but not behind an inline-asm block (the pair splits end in v_fma_mixhi_f16).  The disassembly of the BUILT library is
    scanned for any such pair (tools/kernel_resources.py::mfma_operand_hazards) — whatever the scheduler did with the blocks in this build."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources as KR
    if not os.path.exists(KR.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    assert KR.mfma_operand_hazards() == []
    assert len(KR.mfma_operand_hazards(need=3)) > 1000       # the scan sees the kernels' MFMAs: hipcc's own padding sits at exactly two states
    # the reverse direction: a vector instruction reading an MFMA's result too early (the FIRST pair split of a generator result pads inside its asm block)
    assert KR.mfma_result_hazards() == []


def test_bench_path_kernels_use_no_scratch():
    """VERDICT r4 (C5's dominant kernel: 'scratch 0'): the kernels the BASELINE configs dispatch — triple-mode fc forward / gradient and the conv2 pair,
    relu and leaky — hold everything in registers; read from the code objects inside the built library (tools/kernel_resources.py: no GPU)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources as KR
    if not os.path.exists(KR.READELF):
        pytest.skip("llvm-readelf not in this image")
    res = KR.kernel_resources()
    assert len(res) > 100, len(res)
    hot = {n: r for n, r in res.items() if re.search(r"(conv_bwd_dense_x3_kernel|conv2_pool_x3_kernel|fc_forward_x3_kernel|fc_grad_x3_kernel|lowdim_kernel)<[01],", n)}
    assert len(hot) >= 20, sorted(hot)
    bad = {n: r["scratch"] for n, r in hot.items() if r["scratch"] or r["spill_vgpr"]}
    assert not bad, bad
    assert all(r["scratch"] == 0 for n, r in res.items() if "conv_bwd_dense_x3_kernel" in n)       # every activation, both geometries
    assert max(r["vgpr"] + r["agpr"] for r in res.values()) <= 512
