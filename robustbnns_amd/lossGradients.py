"""Expected loss gradients — the call surface of the reference's lossGradients.py:20-76.

The reference computes, per test point, one forward + one backward per posterior sample and averages the
gradients (lossGradients.py:29-40).  Here all points and samples go through the HIP kernels at once with
the per-sample loss (RBNN_LOSS_PER_SAMPLE).
"""
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
