#!/usr/bin/env python3
"""Round-3 golden vectors: TRAINED posteriors, PGD trajectories and reference-WRITTEN files.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden_trained.py [trained|mnist|mnist2|traj|files|traj2]

Everything here is produced by the reference's own functions, imported unmodified from /root/reference behind the inert
keras / pyro stubs of make_golden.py (neither package is touched by these code paths):

    NN.train                      model_nn.py:175-219     plain Adam on CrossEntropyLoss — gives posteriors that CLASSIFY, so
                                                          that "(orig_acc, adv_acc) equal the reference's" is a real statement
                                                          (the random N(0, std^2) posteriors of make_golden.py score 0 / 0)
    load_half_moons               utils.py:67-92          sklearn make_moons, min-max normalised
    attack / fgsm / pgd           adversarialAttacks.py:69-143
    attack_evaluation             adversarialAttacks.py:151-198
    build_eps_attacks_df          plot_eps_attacks.py:9-39
    Ensemble_NN.forward           model_ensemble.py:57-67
    BNN.save (hmc branch)         model_bnn.py:157-162    the per-sample state_dict files BNN.load reads back
    save_loss_gradients           lossGradients.py:70-72  result pickles (numpy array)
    attack()'s pickle             adversarialAttacks.py:140-141 (torch tensor through utils.save_to_pickle, utils.py:242-248)

The PGD trajectories are the reference's own iterates: pgd_attack (adversarialAttacks.py:95-105) calls net.forward once per
iteration with the current iterate, so an observing wrapper around `forward` records x_0 .. x_39 and the return value is x_40 —
the function itself runs unmodified.

Only arrays and the files the reference wrote are stored; no reference source is copied.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                             # noqa: E402  (stubs + helpers)

FILES = os.path.join(HERE, "files")


@contextlib.contextmanager
def quiet(tmp=None):
    """The reference prints progress bars and writes PNGs / pickles under relative directories: run it in a scratch cwd."""
    cwd = os.getcwd()
    own = tmp is None
    tmp = tempfile.mkdtemp() if own else tmp
    os.chdir(tmp)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            yield tmp
    finally:
        os.chdir(cwd)
        if own:
            shutil.rmtree(tmp, ignore_errors=True)


def train_members(arch, hidden, M, x_train, y_train, shape, n_classes, dataset, n_sub, epochs, lr, seed0):
    """M networks trained by the reference's NN.train on different random subsets from different initialisations: a
    bootstrap "posterior" whose members all classify but disagree near the boundary."""
    import model_nn
    from torch.utils.data import DataLoader
    nets = []
    for i in range(M):
        torch.manual_seed(seed0 + i)                                 # initial weights (NN.train reseeds only afterwards)
        net = model_nn.NN(dataset_name=dataset, input_shape=shape, output_size=n_classes, hidden_size=hidden,
                          activation="leaky", architecture=arch, lr=lr, epochs=epochs)
        sel = np.random.default_rng(seed0 + i).choice(len(x_train), n_sub, replace=False)
        loader = DataLoader(dataset=list(zip(torch.as_tensor(x_train[sel]), torch.as_tensor(y_train[sel]))), batch_size=64, shuffle=False)
        with quiet():
            net.train(loader, "cpu", seed=i, save=False)
        nets.append(net)
    return nets


def as_bnn(nets, dataset, arch, hidden, shape, n_classes):
    import model_bnn
    bnn = model_bnn.BNN(dataset_name=dataset, hidden_size=hidden, activation="leaky", architecture=arch, inference="hmc",
                        epochs=None, lr=None, n_samples=len(nets), warmup=0, input_shape=shape, output_size=n_classes)
    bnn.device = "cpu"
    bnn.basenet.device = "cpu"
    bnn.posterior_predictive = {i: nets[i] for i in range(len(nets))}
    return bnn


def as_ensemble(nets, dataset, arch, hidden, shape, n_classes):
    import model_ensemble
    ens = model_ensemble.Ensemble_NN(dataset_name=dataset, hidden_size=hidden, activation="leaky", architecture=arch, epochs=1,
                                     lr=0.01, input_shape=shape, output_size=n_classes, ensemble_size=len(nets))
    ens.device = "cpu"
    for i, net in enumerate(nets):
        net.device = "cpu"
        ens.ensemble_models[str(i)] = net
    return ens


def meanprob_grads(net, x, y, ns):
    """The gradient fgsm_attack takes the sign of (adversarialAttacks.py:74-79), per point — for the |g| < tau sign-flip rule."""
    out = []
    for idx in range(len(x)):
        image = x[idx].unsqueeze(0).clone().requires_grad_(True)
        label = y[idx].argmax(-1).unsqueeze(0)
        loss = torch.nn.CrossEntropyLoss()(net.forward(inputs=image, n_samples=ns), label)
        loss.backward()
        out.append(image.grad.detach().clone())
    return torch.cat(out)


def attack_grid(net, x, y, dataset, method, eps_list, ns_list, tag, out):
    """attack() + attack_evaluation() over the eps x n_samples grid, exactly as build_eps_attacks_df drives them."""
    import adversarialAttacks as AA
    E, K, N = len(eps_list), len(ns_list), len(x)
    adv = np.zeros((E, K) + tuple(x.shape), dtype="float32")
    oacc, aacc, rob = np.zeros((E, K)), np.zeros((E, K)), np.zeros((E, K, N), dtype="float32")
    for e, eps in enumerate(eps_list):
        for k, ns in enumerate(ns_list):
            with quiet():
                xa = AA.attack(net=net, x_test=x, y_test=y, dataset_name=dataset, device="cpu", method=method, filename=net.name,
                               n_samples=ns, hyperparams={"epsilon": eps})
                oa, aa, r = AA.attack_evaluation(net=net, x_test=x, n_samples=ns, x_attack=xa, y_test=y, device="cpu")
            adv[e, k], oacc[e, k], aacc[e, k], rob[e, k] = xa.detach().numpy(), oa, aa, r.numpy()
            print(f"  {tag} {method} eps={eps} ns={ns}: orig {oa:.2f}  adv {aa:.2f}  rob {float(r.mean()):.4f}", flush=True)
    out[tag + "_" + method + "_adv"] = adv
    out[tag + "_" + method + "_orig_acc"], out[tag + "_" + method + "_adv_acc"], out[tag + "_" + method + "_rob"] = oacc, aacc, rob


def record_pgd_trajectory(net, x, y, eps, ns):
    """The reference's pgd_attack iterates for every row of x: [41, N, *shape] (x_0 = the clean image, x_40 = its return value) and
    the mean-prob gradient at x_0..x_39 [40, N, *shape]."""
    import adversarialAttacks as AA
    traj, grads = [], []
    real_forward = net.forward
    for idx in range(len(x)):
        seen = []

        def spy(*a, **k):
            seen.append(k["inputs"].detach().clone())
            return real_forward(*a, **k)
        net.forward = spy
        try:
            image = x[idx].unsqueeze(0).clone()
            label = y[idx].argmax(-1).unsqueeze(0)
            last = AA.pgd_attack(net=net, image=image, label=label, hyperparams={"epsilon": eps}, n_samples=ns).detach()
        finally:
            net.forward = real_forward
        assert len(seen) == 40
        traj.append(torch.cat(seen + [last]))
        grads.append(torch.cat([meanprob_grads(net, it, y[idx:idx + 1], ns) for it in seen]))
    return torch.stack(traj, 1).numpy(), torch.stack(grads, 1).numpy()


def eps_grid_df(bnn, x, y, dataset, method, eps_list, ns_list, out, tag):
    import plot_eps_attacks
    with quiet() as tmp:
        df = plot_eps_attacks.build_eps_attacks_df(bnn=bnn, dataset=dataset, device="cpu", method=method, x_test=x, y_test=y,
                                                   epsilon_list=eps_list, n_samples_list=ns_list, savedir=bnn.name)
        csv = [os.path.relpath(os.path.join(r, f), tmp) for r, _, fs in os.walk(tmp) for f in fs if f.endswith(".csv")]
    for col in ("epsilon", "test_acc", "adv_acc", "softmax_rob", "n_samples"):
        out[tag + "_df_" + col] = df[col].to_numpy().astype("float64")
    return sorted(csv)


# ------------------------------------------------------------------------------------------------ (1) half-moons, trained
def run_trained_halfmoons(name, arch, eps_list, ns_list, pgd_eps, pgd_ns, traj_points=0):
    import utils
    x_train, y_train, x_test, y_test, shape, C = utils.load_half_moons()
    hidden, M, N = 32, 10, 200
    nets = train_members(arch, hidden, M, x_train, y_train, shape, C, "half_moons", n_sub=1500, epochs=8, lr=0.02, seed0=1000)
    x, y = torch.from_numpy(x_test[:N]), torch.from_numpy(y_test[:N])
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(MG.state_arrays(nets))
    bnn = as_bnn(nets, "half_moons", arch, hidden, shape, C)
    ens = as_ensemble(nets, "half_moons", arch, hidden, shape, C)
    attack_grid(bnn, x, y, "half_moons", "fgsm", eps_list, ns_list, "bnn", out)
    attack_grid(bnn, x, y, "half_moons", "pgd", pgd_eps, pgd_ns, "bnn", out)
    out["bnn_fgsm_grad"] = np.stack([meanprob_grads(bnn, x, y, ns).numpy() for ns in ns_list])
    attack_grid(ens, x, y, "half_moons", "fgsm", eps_list, ns_list, "ens", out)
    attack_grid(ens, x, y, "half_moons", "pgd", pgd_eps[:1], pgd_ns[-1:], "ens", out)
    out["ens_fgsm_grad"] = np.stack([meanprob_grads(ens, x, y, ns).numpy() for ns in ns_list])
    csv = eps_grid_df(bnn, x, y, "half_moons", "fgsm", eps_list, ns_list, out, "fgsm")
    meta = dict(dataset="half_moons", shape=list(shape), n_classes=C, hidden=hidden, act="leaky", arch=arch, S=M, N=N,
                eps_list=list(eps_list), ns_list=list(ns_list), pgd_eps=list(pgd_eps), pgd_ns=list(pgd_ns),
                ens_pgd_eps=list(pgd_eps[:1]), ens_pgd_ns=list(pgd_ns[-1:]), csv_files=csv, bnn_name=bnn.name,
                trained="NN.train (model_nn.py:175-219), 8 epochs Adam lr 0.02 on 1500-point subsets, seeds 1000+i / i")
    if traj_points:
        t_eps, t_ns = pgd_eps[0], pgd_ns[-1]
        out["traj"], out["traj_grad"] = record_pgd_trajectory(bnn, x[:traj_points], y[:traj_points], t_eps, t_ns)
        meta.update(traj_eps=t_eps, traj_ns=t_ns, traj_points=traj_points)
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------------ (2) MNIST-shaped, trained
def synth_digits(n, seed, noise=0.35, amp=0.17):
    """A 10-class task on 1x28x28 images in [0,1] (no dataset can be downloaded here): class c = a fixed random 7x7 binary pattern,
    upsampled x4, at contrast `amp` around 0.5, plus N(0, noise^2) pixel noise, clamped.  Hard enough that one trained member
    scores ~75 % and five together ~99 %, and that FGSM with eps 0.01 .. 0.1 walks the accuracy from there to 0."""
    g = torch.Generator().manual_seed(777)
    proto = (torch.rand(10, 1, 7, 7, generator=g) > 0.5).float()
    proto = torch.nn.functional.interpolate(proto, scale_factor=4, mode="nearest")
    g2 = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, 10, (n,), generator=g2)
    x = (proto[lab] * amp + (0.5 - amp / 2) + noise * torch.randn(n, 1, 28, 28, generator=g2)).clamp(0, 1)
    y = torch.zeros(n, 10)
    y[torch.arange(n), lab] = 1
    return x, y


def run_trained_mnist_shaped(name, arch="fc", hidden=128, M=5, N=200, epochs=2, lr=0.001, seed0=2000, ns_list=(1, 3, 5), pgd_points=48, traj_points=4):
    """M nets of one architecture trained by the reference's NN.train on the synthetic 10-class task, attacked (FGSM over an eps x n_samples
    grid, PGD on a prefix) and scored by the reference.  fc 784-128-10 (the triple kernels' smallest hidden size), fc2 784-128-128-10,
    conv (hidden 16: the reference's conv on 1x28x28)."""
    shape, C = (1, 28, 28), 10
    x_train, y_train = synth_digits(3000, 1)
    x, y = synth_digits(N, 2)
    nets = train_members(arch, hidden, M, x_train.numpy(), y_train.numpy(), shape, C, "mnist", n_sub=1500, epochs=epochs, lr=lr, seed0=seed0)
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(MG.state_arrays(nets))
    bnn = as_bnn(nets, "mnist", arch, hidden, shape, C)
    eps_list, ns_list = [0.01, 0.02, 0.04, 0.06, 0.1], list(ns_list)
    import adversarialAttacks as AA
    E, K = len(eps_list), len(ns_list)
    oacc, aacc, rob = np.zeros((E, K)), np.zeros((E, K)), np.zeros((E, K, N), dtype="float32")
    sign = np.zeros((K, N) + shape, dtype="int8")
    for k, ns in enumerate(ns_list):
        grad = meanprob_grads(bnn, x, y, ns)
        if ns == ns_list[-1]:                                         # the fp32 gradient itself once (0.6 MB); the other n_samples keep its sign
            out[f"bnn_fgsm_grad_ns{ns}"] = grad.numpy()
        for e, eps in enumerate(eps_list):
            with quiet():
                xa = AA.attack(net=bnn, x_test=x, y_test=y, dataset_name="mnist", device="cpu", method="fgsm", filename=bnn.name,
                               n_samples=ns, hyperparams={"epsilon": eps})
                oa, aa, r = AA.attack_evaluation(net=bnn, x_test=x, n_samples=ns, x_attack=xa, y_test=y, device="cpu")
            # the adversarial set is x + eps * sign(grad) clamped: store it once per n_samples as the sign pattern the reference took
            # (checked here to reproduce its images bit for bit) instead of 15 x 200 images
            s = torch.sign(grad)
            assert torch.equal(torch.clamp(x + eps * s, 0, 1), xa.detach()), "fgsm image is not clamp(x + eps*sign(g))"
            sign[k] = s.numpy().astype("int8")
            oacc[e, k], aacc[e, k], rob[e, k] = oa, aa, r.numpy()
            print(f"  mnist-shaped {arch} fgsm eps={eps} ns={ns}: orig {oa:.2f}  adv {aa:.2f}  rob {float(r.mean()):.4f}", flush=True)
    out["bnn_fgsm_sign"], out["bnn_fgsm_orig_acc"], out["bnn_fgsm_adv_acc"], out["bnn_fgsm_rob"] = sign, oacc, aacc, rob
    # PGD on the first points, eps 0.04, all members
    P, p_eps = pgd_points, 0.04
    with quiet():
        xa = AA.attack(net=bnn, x_test=x[:P], y_test=y[:P], dataset_name="mnist", device="cpu", method="pgd", filename=bnn.name,
                       n_samples=M, hyperparams={"epsilon": p_eps})
        oa, aa, r = AA.attack_evaluation(net=bnn, x_test=x[:P], n_samples=M, x_attack=xa, y_test=y[:P], device="cpu")
    out["bnn_pgd_adv"], out["bnn_pgd_orig_acc"], out["bnn_pgd_adv_acc"], out["bnn_pgd_rob"] = xa.detach().numpy(), np.float64(oa), np.float64(aa), r.numpy()
    print(f"  mnist-shaped {arch} pgd eps={p_eps} ns={M} on {P} points: orig {oa:.2f}  adv {aa:.2f}", flush=True)
    out["traj"], out["traj_grad"] = record_pgd_trajectory(bnn, x[:traj_points], y[:traj_points], p_eps, M)
    meta = dict(dataset="mnist", shape=list(shape), n_classes=C, hidden=hidden, act="leaky", arch=arch, S=M, N=N, eps_list=eps_list,
                ns_list=ns_list, pgd_points=P, pgd_eps=p_eps, pgd_ns=M, traj_eps=p_eps, traj_ns=M, traj_points=traj_points,
                trained=f"NN.train (model_nn.py:175-219), {epochs} epochs Adam lr {lr} on 1500-image subsets of synth_digits(3000, 1), seeds {seed0}+i / i")
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------------ (3) PGD trajectory, fc-512
def run_traj_fc512(name):
    """The benchmark architecture 784->512->10, S=8 synthetic samples (regenerated from seeds like mnist_fc_h512_s8_n8_leaky)."""
    import copy
    import model_bnn
    shape, C, H, S, N, std, seed, eps = (1, 28, 28), 10, 512, 8, 8, 0.05, 6, 0.3
    bnn = model_bnn.BNN(dataset_name="mnist", hidden_size=H, activation="leaky", architecture="fc", inference="hmc", epochs=None,
                        lr=None, n_samples=S, warmup=0, input_shape=shape, output_size=C)
    bnn.device = "cpu"
    bnn.basenet.device = "cpu"
    nets = []
    for i in range(S):
        net = copy.deepcopy(bnn.basenet)
        MG.fill_net(net, i, std)
        nets.append(net)
    bnn.posterior_predictive = {i: nets[i] for i in range(S)}
    x, y = MG.synth_inputs(N, shape, C, seed)
    traj, grad = record_pgd_trajectory(bnn, x, y, eps, S)
    meta = dict(dataset="mnist", shape=list(shape), n_classes=C, hidden=H, act="leaky", arch="fc", S=S, N=N, std=std, seed=seed,
                traj_eps=eps, traj_ns=S, weights_sha256=MG.sha(MG.state_arrays(nets)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, x=x.numpy(), y=y.numpy(), traj=traj, traj_grad=grad.astype("float32"), meta=np.array(repr(meta)))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------------ (4) reference-written files
def run_reference_files():
    """Files WRITTEN by the reference (BNN.save hmc branch, save_loss_gradients, attack()'s pickle) for the package to READ, plus
    the numbers a reader must reproduce from them."""
    import copy
    import adversarialAttacks as AA
    import lossGradients
    import model_bnn
    import savedir
    from torch.utils.data import DataLoader
    shutil.rmtree(FILES, ignore_errors=True)
    os.makedirs(FILES)
    shape, C, H, S, N = (1, 2, 1), 2, 32, 5, 12
    bnn = model_bnn.BNN(dataset_name="half_moons", hidden_size=H, activation="leaky", architecture="fc2", inference="hmc", epochs=None,
                        lr=None, n_samples=S, warmup=3, input_shape=shape, output_size=C)
    bnn.device = "cpu"
    bnn.basenet.device = "cpu"
    nets = []
    for i in range(S):
        net = copy.deepcopy(bnn.basenet)
        MG.fill_net(net, i, 0.4)
        nets.append(net)
    bnn.posterior_predictive = {i: nets[i] for i in range(S)}
    x, y = MG.synth_inputs(N, shape, C, 21)
    expected = {"x": x.numpy(), "y": y.numpy()}
    with quiet() as tmp:
        bnn.save(rel_path="posterior/")                                                  # model_bnn.py:157-162
        with torch.no_grad():
            expected["forward_probs"] = bnn.forward(x, n_samples=S).numpy()
        # round trip through the reference's own reader: the files are what BNN.load expects
        bnn2 = model_bnn.BNN(dataset_name="half_moons", hidden_size=H, activation="leaky", architecture="fc2", inference="hmc",
                             epochs=None, lr=None, n_samples=S, warmup=3, input_shape=shape, output_size=C)
        bnn2.load(device="cpu", rel_path="posterior/")
        with torch.no_grad():
            assert np.array_equal(bnn2.forward(x, n_samples=S).numpy(), expected["forward_probs"])
        loader = DataLoader(dataset=list(zip(x, y)), batch_size=5, shuffle=False)
        lg = lossGradients.loss_gradients(net=bnn, data_loader=loader, device="cpu", filename=bnn.name, savedir="grads/", n_samples=S)
        expected["loss_gradients"] = lg
        adv = AA.attack(net=bnn, x_test=x, y_test=y, dataset_name="half_moons", device="cpu", method="fgsm", filename=bnn.name,
                        savedir="attacks", hyperparams={"epsilon": 0.2}, n_samples=S)
        expected["fgsm"] = adv.detach().numpy()
        copied = []
        for root, _, fs in os.walk(tmp):
            for f in fs:
                if f.endswith(".png"):
                    continue
                rel = os.path.relpath(os.path.join(root, f), tmp)
                # DATA / TESTS are date-stamped directories (savedir.py:5-6): file them under stable names
                dst = rel.replace(savedir.TESTS, "TESTS/").replace(savedir.DATA, "DATA/")
                os.makedirs(os.path.dirname(os.path.join(FILES, dst)), exist_ok=True)
                shutil.copyfile(os.path.join(root, f), os.path.join(FILES, dst))
                copied.append(dst)
    meta = dict(dataset="half_moons", shape=list(shape), n_classes=C, hidden=H, act="leaky", arch="fc2", S=S, N=N, warmup=3, eps=0.2,
                bnn_name=bnn.name, files=sorted(copied))
    expected["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(FILES, "expected.npz"), **expected)
    for f in sorted(copied):
        print("  wrote", os.path.join("tests/golden/files", f), os.path.getsize(os.path.join(FILES, f)), "B")


# ------------------------------------------------------------------------------------------------ (5) round 4: more PGD trajectories
def run_traj_halfmoons_fc2(name):
    """The reference's own PGD iterates on the trained half-moons fc2 posterior (the nets of trained_halfmoons_fc2_h32_m10: train_members is
    deterministic, checked against that fixture's weights) — as a BNN (mean of probabilities), as an Ensemble_NN (mean of logits) and for
    ONE deterministic member (n_samples=None), 8 points each."""
    import utils
    x_train, y_train, x_test, y_test, shape, C = utils.load_half_moons()
    hidden, M, P = 32, 10, 8
    nets = train_members("fc2", hidden, M, x_train, y_train, shape, C, "half_moons", n_sub=1500, epochs=8, lr=0.02, seed0=1000)
    have = np.load(os.path.join(HERE, "trained_halfmoons_fc2_h32_m10.npz"), allow_pickle=True)
    arrs = MG.state_arrays(nets)
    assert all(np.array_equal(arrs[k], have[k]) for k in arrs), "training is not reproducible: the trajectory would belong to other weights"
    x, y = torch.from_numpy(x_test[:P]), torch.from_numpy(y_test[:P])
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(arrs)
    bnn = as_bnn(nets, "half_moons", "fc2", hidden, shape, C)
    ens = as_ensemble(nets, "half_moons", "fc2", hidden, shape, C)
    eps = 0.1
    out["traj"], out["traj_grad"] = record_pgd_trajectory(bnn, x, y, eps, M)
    out["ens_traj"], out["ens_traj_grad"] = record_pgd_trajectory(ens, x, y, eps, M)
    out["nn0_traj"], out["nn0_traj_grad"] = record_pgd_trajectory(nets[0], x, y, eps, None)
    meta = dict(dataset="half_moons", shape=list(shape), n_classes=C, hidden=hidden, act="leaky", arch="fc2", S=M, N=P,
                traj_eps=eps, traj_ns=M, traj_points=P, weights_sha256=MG.sha(arrs))
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def run_traj_det_and_ensemble(name):
    """PGD iterates of the reference for an Ensemble_NN (mean of logits, model_ensemble.py:57-67) and for a deterministic NN
    (adversarialAttacks.py:287-350 path: n_samples=None) on the MNIST-shaped nets of make_golden.py's mnist_det_ens_fc_h32_m4_n6."""
    import model_ensemble
    import model_nn
    have = np.load(os.path.join(HERE, "mnist_det_ens_fc_h32_m4_n6.npz"), allow_pickle=True)
    import ast
    m = ast.literal_eval(str(have["meta"]))
    shape, C, H, M, P = tuple(m["shape"]), m["n_classes"], m["hidden"], m["M"], 4
    nets = []
    ens = model_ensemble.Ensemble_NN(dataset_name="mnist", hidden_size=H, activation=m["act"], architecture=m["arch"], epochs=1, lr=0.01,
                                     input_shape=shape, output_size=C, ensemble_size=M)
    ens.device = "cpu"
    for i in range(M):
        net = model_nn.NN(dataset_name="mnist", input_shape=shape, output_size=C, hidden_size=H, activation=m["act"], architecture=m["arch"],
                          lr=0.01, epochs=1)
        net.device = "cpu"
        MG.fill_net(net, i, m["std"])
        ens.ensemble_models[str(i)] = net
        nets.append(net)
    arrs = MG.state_arrays(nets)
    assert all(np.array_equal(arrs[k], have[k]) for k in arrs)
    x, y = torch.from_numpy(have["x"][:P]), torch.from_numpy(have["y"][:P])
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(arrs)
    out["ens_traj"], out["ens_traj_grad"] = record_pgd_trajectory(ens, x, y, m["eps"], M)
    out["nn0_traj"], out["nn0_traj_grad"] = record_pgd_trajectory(nets[0], x, y, m["eps"], None)
    assert np.array_equal(out["ens_traj"][40], have["ens_pgd"][:P]) and np.array_equal(out["nn0_traj"][40], have["nn0_pgd"][:P])
    meta = dict(m, N=P, S=M, traj_eps=m["eps"], traj_ns=M, traj_points=P, weights_sha256=MG.sha(arrs))
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.astype("float32") if k.endswith("traj_grad") else v) for k, v in out.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    MG._install_stubs()
    sys.path.insert(0, MG.REF)
    torch.set_num_threads(1)
    # attack() draws two PNG grids of the whole test set per call (adversarialAttacks.py:137-138: 7 s of matplotlib per grid cell and
    # no numbers; its file side effects are pinned by make_golden.py's with_attack_fn case): switched off for these grids
    import adversarialAttacks as AA
    AA.plot_save_grid_images = lambda *a, **k: None
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only in (None, "trained"):
        run_trained_halfmoons("trained_halfmoons_fc_h32_m10", "fc", eps_list=[0.05, 0.1, 0.2, 0.3], ns_list=[1, 5, 10],
                              pgd_eps=[0.1, 0.2], pgd_ns=[5, 10], traj_points=8)
        run_trained_halfmoons("trained_halfmoons_fc2_h32_m10", "fc2", eps_list=[0.05, 0.1, 0.2, 0.3], ns_list=[1, 5, 10],
                              pgd_eps=[0.1], pgd_ns=[10])
    if only in (None, "mnist"):
        run_trained_mnist_shaped("trained_mnistshaped_fc_h128_m5")
    if only in (None, "mnist2"):
        run_trained_mnist_shaped("trained_mnistshaped_fc2_h128_m3", arch="fc2", hidden=128, M=3, N=100, seed0=3000, ns_list=(1, 3), pgd_points=24,
                                 traj_points=2)
        run_trained_mnist_shaped("trained_mnistshaped_conv_h16_m3", arch="conv", hidden=16, M=3, N=100, epochs=3, seed0=3000, ns_list=(1, 3),
                                 pgd_points=16, traj_points=2)
    if only in (None, "traj"):
        run_traj_fc512("pgd_traj_mnist_fc_h512_s8_n8")
    if only in (None, "files"):
        run_reference_files()
    if only in (None, "traj2"):
        run_traj_halfmoons_fc2("pgd_traj_halfmoons_fc2_h32_m10")
        run_traj_det_and_ensemble("pgd_traj_det_ens_fc_h32_m4_n6")


if __name__ == "__main__":
    main()
