"""Expected loss gradients — the call surface of the reference's lossGradients.py:20-76.

The reference computes, per test point, one forward + one backward per posterior sample and averages the
gradients (lossGradients.py:29-40).  Here all points and samples go through the HIP kernels at once with
the per-sample loss (RBNN_LOSS_PER_SAMPLE).
"""
import numpy as np
import torch

from .savedir import DATA
from .utils import load_from_pickle, save_to_pickle


def loss_gradient(net, image, label, n_samples=None):
    """lossGradients.py:20-50.  image [C,H,W], label one-hot [n_classes] -> gradient, image's shape."""
    if not n_samples:
        # the reference's deterministic branch (:42-48) reads undefined names and raises NameError
        raise NameError("name 'net_copy' is not defined")
    # sample i is evaluated with seeds=[i] (lossGradients.py:29-33): deterministic, and the draws for n are a prefix of those for m > n
    eng, S, seeds, _ = net.hot_path(n_samples, seeds=list(range(n_samples)))
    return eng.loss_gradients(image.unsqueeze(0), label.unsqueeze(0), S, seeds=seeds)[0]


def loss_gradients(net, data_loader, device, filename, savedir, n_samples=None):
    """lossGradients.py:52-68 -> np.ndarray [N, ...squeezed], also pickled under DATA+savedir."""
    print(f"\n === Loss gradients on {len(data_loader.dataset)} input images:")
    if not n_samples:
        raise NameError("name 'net_copy' is not defined")
    images = torch.cat([im for im, _ in data_loader])
    labels = torch.cat([lb for _, lb in data_loader])
    eng, S, seeds, _ = net.hot_path(n_samples, seeds=list(range(n_samples)))       # seeds=[i] per sample, :29-33
    grads = eng.loss_gradients(images.to(device), labels, S, seeds=seeds)
    print(f"\nmin = {grads.min():.4f} \t max = {grads.max():.4f}")
    grads = grads.cpu().detach().numpy().squeeze()
    save_loss_gradients(grads, n_samples, filename, savedir)
    return grads


def loss_gradients_and_fgsm(net, x_test, y_test, device, n_samples, hyperparams=None):
    """`loss_gradients` (lossGradients.py:52-68) and an FGSM `attack` (adversarialAttacks.py:69-83, :111-131) on the same inputs and the
    same posterior samples — BASELINE config 4's job — from ONE forward pass (AttackEngine.loss_gradients_and_fgsm: three GEMMs instead of
    four; both results bit-identical to the two calls).  For nets whose two calls see the same weights: an HMC posterior (stored samples;
    seeds [0..n) are its first n samples either way).  An SVI net draws seeded weights for the gradients (:29-33) and fresh ones for the
    attack (model_bnn.py:230-232) — different samples, nothing to share: the two calls are made as they are.
    -> (expected loss gradients [N, ...] device tensor, adversarial inputs [N, ...] device tensor); no files are written."""
    if not n_samples:
        raise NameError("name 'net_copy' is not defined")
    epsilon = hyperparams["epsilon"] if hyperparams is not None else 0.3
    images, labels = x_test.to(device), y_test
    if getattr(net, "inference", None) == "hmc":
        eng, S, _, _ = net.hot_path(n_samples)
        return eng.loss_gradients_and_fgsm(images, labels, S, epsilon)
    from .adversarialAttacks import fgsm_attack
    eng, S, seeds, _ = net.hot_path(n_samples, seeds=list(range(n_samples)))
    grads = eng.loss_gradients(images, labels, S, seeds=seeds)
    lab = torch.as_tensor(labels)
    lab = lab.argmax(-1) if lab.dim() >= 2 else lab
    return grads, fgsm_attack(net, images, lab.to(device), hyperparams, n_samples=n_samples).detach()


def save_loss_gradients(loss_gradients, n_samples, filename, savedir, relpath=DATA):
    """lossGradients.py:70-72"""
    save_to_pickle(data=loss_gradients, path=relpath + savedir, filename=filename + "_samp=" + str(n_samples) + "_lossGrads.pkl")


def load_loss_gradients(n_samples, filename, savedir, relpath=DATA):
    """lossGradients.py:74-76"""
    return load_from_pickle(path=relpath + savedir + filename + "_samp=" + str(n_samples) + "_lossGrads.pkl")


def _vanishing_rule(norms):
    """The classification of lossGradients.py:89-121 on a table norms[image, j] (j along n_samples_list): an image is
    `vanishing` when its first norm is non-zero and every norm is <= the last ACCEPTED one (a running comparison, not a
    comparison of neighbours); `null` when the first norm is exactly 0; `increasing` otherwise."""
    vanishing, count_incr, count_null = [], 0, 0
    for image_idx in range(norms.shape[0]):
        gradient_norm = norms[image_idx, 0]
        if gradient_norm != 0.0:
            count = 0
            for j in range(norms.shape[1]):
                if norms[image_idx, j] <= gradient_norm:
                    gradient_norm = norms[image_idx, j]
                    count += 1
            if count == norms.shape[1]:
                vanishing.append(image_idx)
            else:
                count_incr += 1
        else:
            count_null += 1
    n = len(norms)
    print(f"vanishing gradients = {len(vanishing)/n} %")
    print(f"increasing gradients = {count_incr/n} %")
    print(f"null gradients = {count_null/n} %")
    print("\nvanishing_gradients_idxs = ", vanishing)
    return vanishing


def compute_vanishing_norms_idxs(loss_gradients, n_samples_list, norm):
    """lossGradients.py:78-127: indices of the images whose expected-gradient norm never increases along
    `n_samples_list`.  loss_gradients: np.ndarray [n_images, len(n_samples_list), ...] — host post-processing of pickled
    results; `expected_gradient_norms` + `_vanishing_rule` is the same computation with the norms taken on the GPU."""
    if loss_gradients.shape[1] != len(n_samples_list):
        raise ValueError("Second dimension should equal the length of `n_samples_list`")
    flat = np.asarray(loss_gradients).reshape(loss_gradients.shape[0], loss_gradients.shape[1], -1)
    if norm == "linfty":
        norms = np.abs(flat).max(axis=-1)
    elif norm == "l2":
        norms = np.stack([[np.linalg.norm(flat[i, j]) for j in range(flat.shape[1])] for i in range(flat.shape[0])])
    else:
        raise UnboundLocalError("local variable 'gradient_norm' referenced before assignment")
    return _vanishing_rule(norms)


def expected_gradient_norms(net, images, labels, n_samples_list, norm, device=None):
    """norms[image, j] of the expected loss gradient at n_samples_list[j] samples, taken INSIDE the gradient pass on the GPU
    (rbnn_sum_slabs_norms) — what compute_vanishing_norms_idxs derives from the stored gradients (lossGradients.py:91-105),
    without the [N, len(list), D] gradients ever leaving the device.  Returns (norms np.ndarray [N, len(list)] fp32,
    gradients list of device tensors).  Sample i is seeds=[i] as in loss_gradient, so the draws nest along the list."""
    if norm not in ("linfty", "l2"):
        raise UnboundLocalError("local variable 'gradient_norm' referenced before assignment")
    device = net.device if device is None else device
    cols, grads = [], []
    for n_samples in n_samples_list:
        eng, S, seeds, _ = net.hot_path(n_samples, seeds=list(range(n_samples)))
        g, linf, l2 = eng.loss_gradients(images.to(device), labels, S, seeds=seeds, norms=True)
        cols.append(linf if norm == "linfty" else l2)
        grads.append(g)
    return torch.stack(cols, dim=1).cpu().numpy(), grads


def vanishing_gradients_idxs(net, images, labels, n_samples_list, norm, device=None):
    """compute_vanishing_norms_idxs over a freshly computed grid, as one resident GPU job (no pickles, no host norms)."""
    norms, _ = expected_gradient_norms(net, images, labels, n_samples_list, norm, device)
    return _vanishing_rule(norms)
