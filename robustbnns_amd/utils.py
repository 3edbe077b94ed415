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
    """utils.py:276-290: a grid of at most 10x10 images written as a PNG (needs matplotlib)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure(figsize=(8, 8))
    rows = cols = min(int(np.sqrt(len(images))), 10)
    for i in range(1, cols * rows):
        fig.add_subplot(rows, cols, i)
        image = np.squeeze(images[i].detach().cpu().numpy())
        if len(image.shape) == 1:
            image = np.expand_dims(image, axis=0)
        plt.imshow(image)
    os.makedirs(os.path.dirname(savedir + "/"), exist_ok=True)
    plt.savefig(savedir + filename)
    plt.close(fig)
