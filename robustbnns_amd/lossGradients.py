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
    eng, S, _, _ = net.hot_path(n_samples)
    return eng.loss_gradients(image.unsqueeze(0), label.unsqueeze(0), S)[0]


def loss_gradients(net, data_loader, device, filename, savedir, n_samples=None):
    """lossGradients.py:52-68 -> np.ndarray [N, ...squeezed], also pickled under DATA+savedir."""
    print(f"\n === Loss gradients on {len(data_loader.dataset)} input images:")
    if not n_samples:
        raise NameError("name 'net_copy' is not defined")
    images = torch.cat([im for im, _ in data_loader])
    labels = torch.cat([lb for _, lb in data_loader])
    eng, S, _, _ = net.hot_path(n_samples)
    grads = eng.loss_gradients(images.to(device), labels, S)
    print(f"\nmin = {grads.min():.4f} \t max = {grads.max():.4f}")
    grads = grads.cpu().detach().numpy().squeeze()
    save_loss_gradients(grads, n_samples, filename, savedir)
    return grads


def save_loss_gradients(loss_gradients, n_samples, filename, savedir, relpath=DATA):
    """lossGradients.py:70-72"""
    save_to_pickle(data=loss_gradients, path=relpath + savedir, filename=filename + "_samp=" + str(n_samples) + "_lossGrads.pkl")


def load_loss_gradients(n_samples, filename, savedir, relpath=DATA):
    """lossGradients.py:74-76"""
    return load_from_pickle(path=relpath + savedir + filename + "_samp=" + str(n_samples) + "_lossGrads.pkl")


def compute_vanishing_norms_idxs(loss_gradients, n_samples_list, norm):
    """lossGradients.py:78-127: indices of the images whose expected-gradient norm never increases along
    `n_samples_list`.  loss_gradients: np.ndarray [n_images, len(n_samples_list), ...] (host post-processing of the
    pickled results; same classification rule, including the running `<=` comparison against the last accepted norm)."""
    if loss_gradients.shape[1] != len(n_samples_list):
        raise ValueError("Second dimension should equal the length of `n_samples_list`")
    flat = np.asarray(loss_gradients).reshape(loss_gradients.shape[0], loss_gradients.shape[1], -1)
    if norm == "linfty":
        norms = np.abs(flat).max(axis=-1)
    elif norm == "l2":
        norms = np.stack([[np.linalg.norm(flat[i, j]) for j in range(flat.shape[1])] for i in range(flat.shape[0])])
    else:
        raise UnboundLocalError("local variable 'gradient_norm' referenced before assignment")
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
