#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the
REFERENCE's own functions (imported, unmodified, from /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference imports `keras` and `pyro` at module top level (utils.py:10-11,
model_bnn.py:20-26); neither is installed here and neither is touched by the HMC
branch of the hot path, so both are replaced by inert stub modules before import
(SURVEY.md section 8c).  With a `BNN(inference="hmc")` whose `posterior_predictive`
dict is filled with seeded synthetic `NN` copies, the reference's own

    BNN.forward            model_bnn.py:198-258 (HMC branch :243-258)
    loss_gradient          lossGradients.py:20-50
    loss_gradients         lossGradients.py:52-68
    fgsm_attack            adversarialAttacks.py:69-83
    pgd_attack             adversarialAttacks.py:86-108
    attack                 adversarialAttacks.py:111-143
    attack_evaluation      adversarialAttacks.py:151-198
    Ensemble_NN.forward    model_ensemble.py:57-67
    NN.forward             model_nn.py:126-141

produce the arrays saved here.  Only arrays (inputs, weights or the seeds that
regenerate them, outputs) are written; no reference source is copied.
"""
import hashlib
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ROBUSTBNNS_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("inert stub called: this code path needs the real package")

    def to_categorical(y, num_classes):
        y = np.asarray(y).astype(int).ravel()
        out = np.zeros((len(y), num_classes), dtype="float32")
        out[np.arange(len(y)), y] = 1.0
        return out

    keras = mod("keras")
    keras.datasets = mod("keras.datasets", mnist=_Inert(), fashion_mnist=_Inert())
    keras.utils = mod("keras.utils", to_categorical=to_categorical)

    def set_rng_seed(seed):
        torch.manual_seed(seed)
        random.seed(seed)
        np.random.seed(seed)

    pyro = mod("pyro", __version__="1.3.0", set_rng_seed=set_rng_seed)
    pyro.poutine = mod("pyro.poutine")
    pyro.infer = mod("pyro.infer", SVI=_Inert, Trace_ELBO=_Inert,
                     TraceMeanField_ELBO=_Inert, Predictive=_Inert)
    pyro.infer.mcmc = mod("pyro.infer.mcmc", MCMC=_Inert, HMC=_Inert, NUTS=_Inert)
    pyro.optim = mod("pyro.optim")
    pyro.distributions = mod("pyro.distributions", OneHotCategorical=_Inert, Normal=_Inert,
                             Categorical=_Inert, Uniform=_Inert)
    pyro.nn = mod("pyro.nn", PyroModule=torch.nn.Module)

    import matplotlib
    matplotlib.use("Agg")


# ------------------------------------------------------------------- synthetic data
def synth_inputs(n, shape, n_classes, seed):
    """X ~ U[0,1) NCHW fp32, labels uniform, one-hot float (utils.py:100-110 ranges)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n,) + tuple(shape), generator=g, dtype=torch.float32)
    y = torch.randint(0, n_classes, (n,), generator=g)
    onehot = torch.zeros(n, n_classes, dtype=torch.float32)
    onehot[torch.arange(n), y] = 1.0
    return x, onehot


def fill_net(net, sample_idx, std):
    """HMC-style posterior sample #i: manual_seed(100+i), every tensor ~ N(0, std^2)."""
    torch.manual_seed(100 + sample_idx)
    with torch.no_grad():
        for p in net.parameters():
            p.normal_(0.0, std)


def state_arrays(nets):
    keys = list(nets[0].state_dict().keys())
    return {("w:" + k): np.stack([n.state_dict()[k].detach().numpy() for n in nets]) for k in keys}


def sha(arrs):
    h = hashlib.sha256()
    for k in sorted(arrs):
        h.update(np.ascontiguousarray(arrs[k]).tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------------- one case
def run_case(name, *, dataset, shape, n_classes, hidden, act, arch, S, N, std, seed,
             eps=0.3, store_weights=True, with_attack_fn=False, pgd_points=None,
             with_pgd_default=False):
    import model_bnn, model_nn, lossGradients, adversarialAttacks

    torch.manual_seed(seed)
    bnn = model_bnn.BNN(dataset_name=dataset, hidden_size=hidden, activation=act,
                        architecture=arch, inference="hmc", epochs=None, lr=None,
                        n_samples=S, warmup=0, input_shape=shape, output_size=n_classes)
    bnn.device = "cpu"
    bnn.basenet.device = "cpu"
    import copy
    nets = []
    for i in range(S):
        net = copy.deepcopy(bnn.basenet)
        fill_net(net, i, std)
        nets.append(net)
    bnn.posterior_predictive = {i: nets[i] for i in range(S)}

    x, y = synth_inputs(N, shape, n_classes, seed)
    out = {"x": x.numpy(), "y": y.numpy()}
    meta = dict(dataset=dataset, shape=list(shape), n_classes=n_classes, hidden=hidden,
                act=act, arch=arch, S=S, N=N, std=std, seed=seed, eps=eps)
    warr = state_arrays(nets)
    meta["weights_sha256"] = sha(warr)
    if store_weights:
        out.update(warr)

    # BNN.forward: mean probs over the first S samples; a seeds=[...] subset; S=1
    with torch.no_grad():
        out["forward_probs"] = bnn.forward(x, n_samples=S).numpy()
        sub = [S - 1, 0, S // 2]
        out["forward_seeds"] = np.array(sub)
        out["forward_probs_seeds"] = bnn.forward(x, n_samples=len(sub), seeds=sub).numpy()
        out["forward_probs_s1"] = bnn.forward(x, n_samples=1).numpy()

    # loss_gradient (per-sample CE, gradient mean) — lossGradients.py:20-50
    lg = [lossGradients.loss_gradient(net=bnn, image=x[i], label=y[i], n_samples=S) for i in range(N)]
    out["loss_gradients"] = torch.stack(lg).numpy()
    half = max(1, S // 2)
    lg = [lossGradients.loss_gradient(net=bnn, image=x[i], label=y[i], n_samples=half) for i in range(N)]
    out["loss_gradients_half"] = torch.stack(lg).numpy()
    meta["S_half"] = half

    # fgsm / pgd exactly as attack() drives them — adversarialAttacks.py:118-131
    def drive(fn, hyper, idxs, ns):
        res = []
        for idx in idxs:
            image = x[idx].unsqueeze(0).clone()
            label = y[idx].argmax(-1).unsqueeze(0)
            res.append(fn(net=bnn, image=image, label=label, hyperparams=hyper, n_samples=ns).detach())
        return torch.cat(res)

    hyper = {"epsilon": eps}
    fg = drive(adversarialAttacks.fgsm_attack, hyper, range(N), S)
    out["fgsm"] = fg.numpy()
    out["fgsm_default_eps"] = drive(adversarialAttacks.fgsm_attack, None, range(N), S).numpy()
    # the expected gradient the attack's sign() was taken of (for the |g|<tau sign-flip rule)
    gm = []
    for idx in range(N):
        image = x[idx].unsqueeze(0).clone().requires_grad_(True)
        label = y[idx].argmax(-1).unsqueeze(0)
        loss = torch.nn.CrossEntropyLoss()(bnn.forward(inputs=image, n_samples=S), label)
        loss.backward()
        gm.append(image.grad.detach().clone())
    out["meanprob_grad"] = torch.cat(gm).numpy()

    pidx = list(range(N)) if pgd_points is None else list(range(pgd_points))
    out["pgd_idx"] = np.array(pidx)
    out["pgd"] = drive(adversarialAttacks.pgd_attack, hyper, pidx, S).numpy()
    if with_pgd_default:
        out["pgd_default"] = drive(adversarialAttacks.pgd_attack, None, pidx, S).numpy()

    # attack_evaluation — adversarialAttacks.py:151-198
    oa, aa, rob = adversarialAttacks.attack_evaluation(net=bnn, x_test=x, x_attack=fg, y_test=y,
                                                       device="cpu", n_samples=S)
    out["eval_orig_acc"] = np.float64(oa)
    out["eval_adv_acc"] = np.float64(aa)
    out["eval_softmax_rob"] = rob.numpy()
    # ... and of the PGD images (the points pgd_idx): the triple a HIP attack -> HIP evaluation run must reproduce
    pg = torch.from_numpy(out["pgd"])
    oa, aa, rob = adversarialAttacks.attack_evaluation(net=bnn, x_test=x[pidx], x_attack=pg, y_test=y[pidx],
                                                       device="cpu", n_samples=S)
    out["eval_pgd_orig_acc"] = np.float64(oa)
    out["eval_pgd_adv_acc"] = np.float64(aa)
    out["eval_pgd_softmax_rob"] = rob.numpy()

    if with_attack_fn:
        # the full attack()/loss_gradients() drivers incl. their file side effects
        import savedir, utils
        from torch.utils.data import DataLoader
        with tempfile.TemporaryDirectory() as tmp:
            cwd = os.getcwd()
            os.chdir(tmp)
            try:
                adv = adversarialAttacks.attack(net=bnn, x_test=x, y_test=y, dataset_name=dataset,
                                                device="cpu", method="fgsm", filename=bnn.name,
                                                hyperparams=hyper, n_samples=S)
                out["attack_fn_fgsm"] = adv.detach().numpy()
                files = []
                for root, _, fs in os.walk(tmp):
                    files += [os.path.relpath(os.path.join(root, f), tmp) for f in fs]
                # TESTS is date-stamped (savedir.py:6): strip the date directory
                out_files = sorted(f.replace(savedir.TESTS, "TESTS/") for f in files)
                loader = DataLoader(dataset=list(zip(x, y)), batch_size=3, shuffle=False)
                lgs = lossGradients.loss_gradients(net=bnn, data_loader=loader, device="cpu",
                                                   filename=bnn.name, savedir="sd/", n_samples=S)
                out["loss_gradients_fn"] = lgs
                files2 = []
                for root, _, fs in os.walk(tmp):
                    files2 += [os.path.relpath(os.path.join(root, f), tmp) for f in fs]
                out_files += sorted(set(f for f in files2 if f.startswith("data/")))
                meta["side_effect_files"] = out_files
                meta["bnn_name"] = bnn.name
            finally:
                os.chdir(cwd)

    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def run_det_and_ensemble(name, *, shape, n_classes, hidden, act, arch, M, N, std, seed, eps=0.3):
    """Deterministic NN attack (n_samples=None) and Ensemble_NN mean-of-logits attack."""
    import model_nn, model_ensemble, adversarialAttacks
    x, y = synth_inputs(N, shape, n_classes, seed)
    ens = model_ensemble.Ensemble_NN(dataset_name="mnist", hidden_size=hidden, activation=act,
                                     architecture=arch, epochs=1, lr=0.01, input_shape=shape,
                                     output_size=n_classes, ensemble_size=M)
    ens.device = "cpu"
    nets = []
    for i in range(M):
        net = model_nn.NN(dataset_name="mnist", input_shape=shape, output_size=n_classes,
                          hidden_size=hidden, activation=act, architecture=arch, lr=0.01, epochs=1)
        net.device = "cpu"
        fill_net(net, i, std)
        ens.ensemble_models[str(i)] = net
        nets.append(net)
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(state_arrays(nets))
    meta = dict(shape=list(shape), n_classes=n_classes, hidden=hidden, act=act, arch=arch,
                M=M, N=N, std=std, seed=seed, eps=eps)

    def drive(net, fn, hyper, ns):
        res = []
        for idx in range(N):
            image = x[idx].unsqueeze(0).clone()
            label = y[idx].argmax(-1).unsqueeze(0)
            res.append(fn(net=net, image=image, label=label, hyperparams=hyper, n_samples=ns).detach())
        return torch.cat(res)

    hyper = {"epsilon": eps}
    with torch.no_grad():
        out["nn0_logits"] = nets[0].forward(x).numpy()
        out["ens_logits"] = ens.forward(x, n_samples=M).numpy()
        out["ens_logits_2"] = ens.forward(x, n_samples=2).numpy()
    out["nn0_fgsm"] = drive(nets[0], adversarialAttacks.fgsm_attack, hyper, None).numpy()
    out["nn0_pgd"] = drive(nets[0], adversarialAttacks.pgd_attack, hyper, None).numpy()
    out["ens_fgsm"] = drive(ens, adversarialAttacks.fgsm_attack, hyper, M).numpy()
    out["ens_pgd"] = drive(ens, adversarialAttacks.pgd_attack, hyper, M).numpy()
    fg = torch.from_numpy(out["nn0_fgsm"])
    oa, aa, rob = adversarialAttacks.attack_evaluation(net=nets[0], x_test=x, x_attack=fg, y_test=y,
                                                       device="cpu", n_samples=None)
    out["nn0_eval_orig_acc"], out["nn0_eval_adv_acc"] = np.float64(oa), np.float64(aa)
    out["nn0_eval_softmax_rob"] = rob.numpy()
    fg = torch.from_numpy(out["ens_fgsm"])
    oa, aa, rob = adversarialAttacks.attack_evaluation(net=ens, x_test=x, x_attack=fg, y_test=y,
                                                       device="cpu", n_samples=M)
    out["ens_eval_orig_acc"], out["ens_eval_adv_acc"] = np.float64(oa), np.float64(aa)
    out["ens_eval_softmax_rob"] = rob.numpy()
    for tag, net, ns in (("nn0", nets[0], None), ("ens", ens, M)):          # the PGD images' evaluation triples
        pg = torch.from_numpy(out[tag + "_pgd"])
        oa, aa, rob = adversarialAttacks.attack_evaluation(net=net, x_test=x, x_attack=pg, y_test=y, device="cpu", n_samples=ns)
        out[tag + "_eval_pgd_orig_acc"], out[tag + "_eval_pgd_adv_acc"] = np.float64(oa), np.float64(aa)
        out[tag + "_eval_pgd_softmax_rob"] = rob.numpy()
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def run_eps_grid(name, *, S, N, std, seed, epsilon_list, n_samples_list, method):
    """plot_eps_attacks.build_eps_attacks_df (plot_eps_attacks.py:9-39): the eps x n_samples grid and its CSV."""
    import copy
    import model_bnn, plot_eps_attacks
    shape, n_classes, hidden = (1, 2, 1), 2, 32
    bnn = model_bnn.BNN(dataset_name="half_moons", hidden_size=hidden, activation="leaky", architecture="fc",
                        inference="hmc", epochs=None, lr=None, n_samples=S, warmup=0, input_shape=shape,
                        output_size=n_classes)
    bnn.device = "cpu"; bnn.basenet.device = "cpu"
    nets = []
    for i in range(S):
        net = copy.deepcopy(bnn.basenet); fill_net(net, i, std); nets.append(net)
    bnn.posterior_predictive = {i: nets[i] for i in range(S)}
    x, y = synth_inputs(N, shape, n_classes, seed)
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            df = plot_eps_attacks.build_eps_attacks_df(bnn=bnn, dataset="half_moons", device="cpu", method=method,
                                                       x_test=x, y_test=y, epsilon_list=epsilon_list,
                                                       n_samples_list=n_samples_list, savedir=bnn.name)
            files = []
            for root, _, fs in os.walk(tmp):
                files += [os.path.relpath(os.path.join(root, f), tmp) for f in fs if f.endswith(".csv")]
        finally:
            os.chdir(cwd)
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(state_arrays(nets))
    for col in ("epsilon", "test_acc", "adv_acc", "softmax_rob", "n_samples"):
        out["df_" + col] = df[col].to_numpy().astype("float64")
    meta = dict(dataset="half_moons", shape=list(shape), n_classes=n_classes, hidden=hidden, act="leaky", arch="fc",
                S=S, N=N, std=std, seed=seed, epsilon_list=list(epsilon_list), n_samples_list=list(n_samples_list),
                method=method, columns=list(df.columns), csv_files=sorted(files), bnn_name=bnn.name)
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def run_vanishing_norms(name):
    """lossGradients.compute_vanishing_norms_idxs (lossGradients.py:78-127) on random + crafted gradients."""
    import contextlib, io
    import lossGradients as ref
    rng = np.random.default_rng(0)
    g = rng.normal(size=(40, 4, 1, 6, 6)).astype("float32")
    g[:, 1] *= np.linspace(0.2, 1.5, 40)[:, None, None, None]
    g[:, 2] *= 0.5
    g[:, 3] *= np.linspace(0.1, 1.2, 40)[::-1][:, None, None, None] * 0.5
    g[3] = 0
    g[7, 1:] = g[7, 0]                       # a null image and an all-equal one
    out = {}
    for norm in ("linfty", "l2"):
        with contextlib.redirect_stdout(io.StringIO()):
            out[norm] = np.array(ref.compute_vanishing_norms_idxs(g, [1, 10, 50, 100], norm))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), grads=g, n_samples_list=np.array([1, 10, 50, 100]), **out)
    print("wrote", name)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(1)          # one thread: bit-stable reductions for the fixtures
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "eps":
        run_eps_grid("halfmoons_eps_grid_fgsm", S=6, N=30, std=0.5, seed=12, epsilon_list=[0.1, 0.3],
                     n_samples_list=[1, 3, 6], method="fgsm")
        run_eps_grid("halfmoons_eps_grid_pgd", S=6, N=12, std=0.5, seed=13, epsilon_list=[0.1, 0.25],
                     n_samples_list=[2, 6], method="pgd")
        return
    mn = (1, 28, 28)
    hm = (1, 2, 1)
    # (1) BASELINE config C1: half-moons fc 2->64->2, S=10, N=100
    run_case("halfmoons_fc_h64_s10_n100", dataset="half_moons", shape=hm, n_classes=2, hidden=64,
             act="leaky", arch="fc", S=10, N=100, std=0.5, seed=1, with_pgd_default=True,
             with_attack_fn=True)
    # (2) MNIST-shaped fc, small hidden so the weights are stored
    run_case("mnist_fc_h32_s8_n8_leaky", dataset="mnist", shape=mn, n_classes=10, hidden=32,
             act="leaky", arch="fc", S=8, N=8, std=0.05, seed=2, with_pgd_default=True)
    run_case("mnist_fc_h32_s8_n8_relu", dataset="mnist", shape=mn, n_classes=10, hidden=32,
             act="relu", arch="fc", S=8, N=8, std=0.05, seed=3)
    run_case("mnist_fc_h16_s4_n6_sigm", dataset="mnist", shape=mn, n_classes=10, hidden=16,
             act="sigm", arch="fc", S=4, N=6, std=0.05, seed=4, pgd_points=2)
    run_case("mnist_fc_h16_s4_n6_tanh", dataset="mnist", shape=mn, n_classes=10, hidden=16,
             act="tanh", arch="fc", S=4, N=6, std=0.05, seed=5, pgd_points=2)
    # (3) the benchmark architecture 784->512->10; weights regenerated from seeds (sha256 in meta)
    run_case("mnist_fc_h512_s8_n8_leaky", dataset="mnist", shape=mn, n_classes=10, hidden=512,
             act="leaky", arch="fc", S=8, N=8, std=0.05, seed=6, store_weights=False, pgd_points=3)
    run_case("mnist_fc_h512_s8_n8_relu", dataset="mnist", shape=mn, n_classes=10, hidden=512,
             act="relu", arch="fc", S=8, N=8, std=0.05, seed=7, store_weights=False, pgd_points=2)
    # (4) fc2
    run_case("mnist_fc2_h32_s4_n6_leaky", dataset="mnist", shape=mn, n_classes=10, hidden=32,
             act="leaky", arch="fc2", S=4, N=6, std=0.08, seed=8, pgd_points=3)
    run_case("halfmoons_fc2_h32_s6_n40", dataset="half_moons", shape=hm, n_classes=2, hidden=32,
             act="leaky", arch="fc2", S=6, N=40, std=0.4, seed=9)
    # (5) conv on 1x28x28 (the only input size the reference's conv head is correct for)
    run_case("mnist_conv_h16_s2_n4_leaky", dataset="mnist", shape=mn, n_classes=10, hidden=16,
             act="leaky", arch="conv", S=2, N=4, std=0.05, seed=10, pgd_points=2)
    # (5b) conv with the two smooth activations the reference also allows (model_nn.py:66-75)
    run_case("mnist_conv_h16_s2_n4_sigm", dataset="mnist", shape=mn, n_classes=10, hidden=16,
             act="sigm", arch="conv", S=2, N=4, std=0.05, seed=14, pgd_points=1)
    run_case("mnist_conv_h16_s2_n4_tanh", dataset="mnist", shape=mn, n_classes=10, hidden=16,
             act="tanh", arch="conv", S=2, N=4, std=0.05, seed=15, pgd_points=1)
    # (7) the eps x n_samples grid driver
    run_eps_grid("halfmoons_eps_grid_fgsm", S=6, N=30, std=0.5, seed=12, epsilon_list=[0.1, 0.3],
                 n_samples_list=[1, 3, 6], method="fgsm")
    run_eps_grid("halfmoons_eps_grid_pgd", S=6, N=12, std=0.5, seed=13, epsilon_list=[0.1, 0.25],
                 n_samples_list=[2, 6], method="pgd")
    # (8) vanishing-gradient classification (host post-processing)
    run_vanishing_norms("vanishing_norms")
    # (6) deterministic NN + Ensemble_NN (mean of logits)
    run_det_and_ensemble("mnist_det_ens_fc_h32_m4_n6", shape=mn, n_classes=10, hidden=32, act="leaky",
                         arch="fc", M=4, N=6, std=0.05, seed=11)


if __name__ == "__main__":
    main()
