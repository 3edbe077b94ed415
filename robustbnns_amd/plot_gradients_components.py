"""The gradient-collection part of the reference's plot_gradients_components.py (the figures themselves — seaborn strip plots and
heat maps, :17-114 — are out of scope: SURVEY.md section 2).

`_get_gradients` mirrors plot_gradients_components.py:125-141: one expected-loss-gradient array per entry of `n_samples_list`,
computed on the HIP path (and pickled where the reference pickles them) or loaded from those pickles.
`vanishing_gradients` is the numerical core of `vanishing_gradients_heatmaps` (:101-108): the transposed stack and the
vanishing-norm indices, with the norms taken inside the GPU gradient pass when the gradients are computed here.
"""
import numpy as np
import torch

from .lossGradients import (_vanishing_rule, compute_vanishing_norms_idxs, expected_gradient_norms, load_loss_gradients,
                            loss_gradients, save_loss_gradients)


def _get_gradients(args, bnn, test_loader, n_samples_list, relpath):
    """plot_gradients_components.py:125-141 (`args` needs .compute_grads and .device)."""
    filename = bnn.name
    loss_gradients_list = []
    for posterior_samples in n_samples_list:
        if args.compute_grads is True:
            loss_grads = loss_gradients(net=bnn, n_samples=posterior_samples, savedir=filename + "/", data_loader=test_loader,
                                        device=args.device, filename=filename)
        else:
            loss_grads = load_loss_gradients(n_samples=posterior_samples, filename=filename, relpath=relpath, savedir=filename + "/")
        loss_gradients_list.append(loss_grads)
    return loss_gradients_list


def vanishing_gradients(bnn, test_loader, device, n_samples_list, norm="linfty", save=False):
    """(transposed gradients [N, len(list), ...], vanishing indices): what vanishing_gradients_heatmaps derives before it plots
    (plot_gradients_components.py:101-106), as one resident GPU job with the norms fused into the gradient kernels' epilogue."""
    images = torch.cat([im for im, _ in test_loader])
    labels = torch.cat([lb for _, lb in test_loader])
    norms, grads = expected_gradient_norms(bnn, images, labels, n_samples_list, norm, device)
    stacked = np.stack([g.cpu().numpy().squeeze() for g in grads], axis=1)
    if stacked.shape[1] != len(n_samples_list):
        raise ValueError("Second dimension should contain the number of samples.")
    if save:
        for n_samples, g in zip(n_samples_list, grads):
            save_loss_gradients(g.cpu().numpy().squeeze(), n_samples, bnn.name, bnn.name + "/")
    return stacked, _vanishing_rule(norms)


__all__ = ["_get_gradients", "vanishing_gradients", "compute_vanishing_norms_idxs"]
