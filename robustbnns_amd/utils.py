"""The few helpers of the reference's utils.py that the hot path's drivers call.

Dataset download/loading (utils.py:67-235) is out of scope: callers pass
tensors shaped like its outputs — NCHW float32 in [0,1] and one-hot float labels.
"""
import os
import pickle as pkl

import numpy as np


def save_to_pickle(data, path, filename):
    """utils.py:242-247"""
    print("\nSaving pickle: ", path + filename)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path + filename, 'wb') as f:
        pkl.dump(data, f)


def load_from_pickle(path):
    """utils.py:250-258"""
    print("\nLoading from pickle: ", path)
    with open(path, 'rb') as f:
        u = pkl._Unpickler(f)
        u.encoding = 'latin1'
        return u.load()


def plot_save_grid_images(images, filename, savedir):
    """The PNG side effect of utils.py:276-290 (a square grid of at most 10 x 10 of the images, skipping images[0] and
    leaving the last cell empty, as the reference's `range(1, cols*rows)` does) — composed as ONE mosaic array and written
    with a single imsave call instead of one matplotlib axes per image."""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import cm, image as mpl_image
    side = min(int(np.sqrt(len(images))), 10)
    tiles = []
    for k in range(side * side - 1):
        t = np.squeeze(images[k + 1].detach().cpu().numpy()).astype(np.float64)
        t = t[None, :] if t.ndim == 1 else t                       # half-moons points are 2-vectors: one row
        t = np.moveaxis(t, 0, -1) if t.ndim == 3 else t            # CHW colour images -> HWC
        lo, hi = float(t.min()), float(t.max())
        t = (t - lo) / (hi - lo) if hi > lo else np.zeros_like(t)  # each cell on its own colour scale, like imshow
        tiles.append(cm.viridis(t)[..., :3] if t.ndim == 2 else t[..., :3])
    os.makedirs(os.path.dirname(savedir + "/"), exist_ok=True)
    if not tiles:                                                  # fewer than 4 images: the reference writes an empty figure
        mpl_image.imsave(savedir + filename, np.ones((8, 8, 3)))
        return
    h, w = tiles[0].shape[:2]
    mosaic = np.ones((side * (h + 1) + 1, side * (w + 1) + 1, 3))
    for k, t in enumerate(tiles):
        r, c = divmod(k, side)
        mosaic[1 + r * (h + 1):1 + r * (h + 1) + h, 1 + c * (w + 1):1 + c * (w + 1) + w] = t
    scale = max(1, 400 // max(mosaic.shape[:2]))                    # small inputs (2-vectors) are blown up to a visible size
    mpl_image.imsave(savedir + filename, np.kron(mosaic, np.ones((scale, scale, 1))))
