"""Deterministic network — the call surface of the reference's model_nn.NN (model_nn.py:34-173).

The module keeps the reference's `nn.Sequential` layout so `state_dict()` keys (`model.1.weight`, ...)
and the on-disk `<name>_weights[_<seed>].pt` files are interchangeable.  `forward` does not run the
Sequential: it stacks the parameters as a one-sample posterior and calls the HIP path (logits out),
so NN, Ensemble_NN and BNN share the same kernels.  Training (model_nn.py:175-219) is out of scope.
"""
import math
import os

import torch
from torch import nn

from .savedir import TESTS

saved_NNs = {"model_0": {"dataset": "mnist", "hidden_size": 512, "activation": "leaky", "architecture": "conv", "epochs": 5, "lr": 0.01},
             "model_5": {"dataset": "mnist", "hidden_size": 512, "activation": "leaky", "architecture": "fc2", "epochs": 10, "lr": 0.01},
             "model_6": {"dataset": "mnist", "hidden_size": 256, "activation": "leaky", "architecture": "conv", "epochs": 10, "lr": 0.05},
             "model_7": {"dataset": "mnist", "hidden_size": 1024, "activation": "leaky", "architecture": "fc2", "epochs": 5, "lr": 0.02},
             "model_8": {"dataset": "mnist", "hidden_size": 1024, "activation": "leaky", "architecture": "fc2", "epochs": 10, "lr": 0.02},
             "model_9": {"dataset": "mnist", "hidden_size": 1024, "activation": "leaky", "architecture": "conv", "epochs": 10, "lr": 0.01}}


def cifar_conv_enabled():
    """Opt-in (RBNN_CIFAR_CONV=1): the build-defined CIFAR-shaped conv net; the reference raises NotImplementedError (model_nn.py:95-96)."""
    return os.environ.get("RBNN_CIFAR_CONV", "0") == "1"


_ACTIV = {"relu": nn.ReLU, "leaky": nn.LeakyReLU, "sigm": nn.Sigmoid, "tanh": nn.Tanh}


class NN(nn.Module):

    def __init__(self, dataset_name, input_shape, output_size, hidden_size, activation, architecture, lr, epochs):
        if math.log(hidden_size, 2).is_integer() is False or hidden_size < 16:
            raise ValueError("\nhidden size should be a power of 2 greater than 16.")     # model_nn.py:39-40
        super(NN, self).__init__()
        self.dataset_name = dataset_name
        self.loss_func = nn.CrossEntropyLoss()
        self.architecture = architecture
        self.hidden_size = hidden_size
        self.output_size = output_size
        self.activation = activation
        self.input_shape = tuple(input_shape)
        self.lr, self.epochs = lr, epochs
        self.name = self.get_name(dataset_name, hidden_size, activation, architecture, lr, epochs)
        self.set_model(architecture, activation, input_shape, output_size, hidden_size)
        self._engine, self._engine_key = None, None

    def get_name(self, dataset_name, hidden_size, activation, architecture, lr, epochs):
        return str(dataset_name) + "_nn_hid=" + str(hidden_size) + "_act=" + str(activation) + \
               "_arch=" + str(architecture) + "_ep=" + str(epochs) + "_lr=" + str(lr)

    def set_model(self, architecture, activation, input_shape, output_size, hidden_size):
        """Same layer list as model_nn.py:60-124 (parameter container only)."""
        input_size = input_shape[0] * input_shape[1] * input_shape[2]
        in_channels = input_shape[0]
        if activation not in _ACTIV:
            raise AssertionError("\nWrong activation name.")
        activ = _ACTIV[activation]
        if architecture == "fc":
            self.model = nn.Sequential(nn.Flatten(), nn.Linear(input_size, hidden_size), activ(),
                                       nn.Linear(hidden_size, output_size))
        elif architecture == "fc2":
            self.model = nn.Sequential(nn.Flatten(), nn.Linear(input_size, hidden_size), activ(),
                                       nn.Linear(hidden_size, hidden_size), activ(),
                                       nn.Linear(hidden_size, output_size))
        elif architecture == "conv":
            if self.dataset_name not in ["mnist", "fashion_mnist"] and not cifar_conv_enabled():
                raise NotImplementedError()                            # model_nn.py:95-96
            # model_nn.py:106 sizes the head as int(hidden/16) * input_size, which equals the flattened conv output only for
            # 28x28 inputs (49 * hidden).  With RBNN_CIFAR_CONV=1 (BASELINE.json configs[4]; no working reference counterpart,
            # SURVEY 8a note) other datasets get the BUILD-DEFINED correctly sized head instead: ((W-4)/2 - 5)^2 * hidden.
            head = int(hidden_size / (4 * 4)) * input_size
            if self.dataset_name not in ["mnist", "fashion_mnist"]:
                head = (((input_shape[1] - 4) // 2) - 5) ** 2 * hidden_size
            self.model = nn.Sequential(nn.Conv2d(in_channels, 32, kernel_size=5), activ(), nn.MaxPool2d(kernel_size=2),
                                       nn.Conv2d(32, hidden_size, kernel_size=5), activ(),
                                       nn.MaxPool2d(kernel_size=2, stride=1), nn.Flatten(),
                                       nn.Linear(head, output_size))
        else:
            raise NotImplementedError()

    # -------------------------------------------------------------------------- HIP path
    def engine(self, device):
        """One-sample stacked posterior of the current parameters (rebuilt when they change)."""
        key = (str(device),) + tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())
        if self._engine is None or self._engine_key != key:
            from .factory import make_engine, posterior_from_modules
            self._engine, self._engine_key = make_engine(posterior_from_modules([self], device)), key
        return self._engine

    def forward(self, inputs, device=None, *args, **kwargs):
        """model_nn.py:126-141: logits [B, C]."""
        device = self.device if device is None else device      # AttributeError if unset, as in the reference
        return self.engine(device).forward(inputs.to(device), n_samples=1, logits=True)

    # -------------------------------------------------------------------------- files
    def save(self, savedir=None, seed=None):
        """model_nn.py:143-151"""
        name = self.name
        directory = name if savedir is None else savedir
        filename = name + "_weights.pt" if seed is None else name + "_weights_" + str(seed) + ".pt"
        os.makedirs(os.path.dirname(TESTS + directory + "/"), exist_ok=True)
        print("\nSaving: ", TESTS + directory + "/" + filename)
        torch.save(self.state_dict(), TESTS + directory + "/" + filename)

    def load(self, device, savedir=None, seed=None, rel_path=TESTS):
        """model_nn.py:158-168"""
        self.device = device
        name = self.name
        directory = name if savedir is None else savedir
        filename = name + "_weights.pt" if seed is None else name + "_weights_" + str(seed) + ".pt"
        print("\nLoading: ", rel_path + directory + "/" + filename)
        self.load_state_dict(torch.load(rel_path + directory + "/" + filename, map_location="cpu"))
        self._engine = None

    def train(self, *args, **kwargs):
        if args and isinstance(args[0], bool) or "mode" in kwargs:          # nn.Module.train(mode)
            return super().train(*args, **kwargs)
        raise NotImplementedError("training is outside the accelerated hot path (SURVEY.md section 2, row 8): "
                                  "train with the reference and load the weights here")

    def evaluate(self, test_loader, device, *args, **kwargs):
        """model_nn.py:221-240"""
        self.device = device
        correct = 0.0
        for x_batch, y_batch in test_loader:
            outputs = self.forward(x_batch.to(device))
            correct += float((outputs.argmax(-1) == y_batch.to(device).argmax(-1)).sum())
        accuracy = 100 * correct / len(test_loader.dataset)
        print("\nAccuracy: %.2f%%" % (accuracy))
        return accuracy
