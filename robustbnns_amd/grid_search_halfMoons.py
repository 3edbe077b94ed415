"""Half-moons model grid: expected loss gradients and attacks for every stored BNN of a hyper-parameter grid — the call
surface of the reference's grid_search_halfMoons.py (MoonsBNN :18-25, serial_compute_grads :94-102, grid_attack :133-153).

The reference trains the grid (`_train`, out of scope here: DESIGN.md section 7), then for every combination loads the HMC
posterior from disk and runs `loss_gradients` / `attack` — on CPU through 10 joblib processes (:58-59, :91-92, :129-131).
Here every model is one resident posterior on the GPU and every (model, n_samples) cell one batched run over all test
points; the grid itself is a plain loop (the work per cell is milliseconds).  Dataset loading is out of scope as well,
so the caller passes the test tensors (`x_test [N,1,2,1]`, `y_test [N,2]` one-hot, as utils.load_dataset returns them).
"""
import itertools

from torch.utils.data import DataLoader

from .adversarialAttacks import attack
from .lossGradients import loss_gradients
from .model_bnn import BNN
from .savedir import TESTS


class MoonsBNN(BNN):
    """grid_search_halfMoons.py:18-25: a BNN on half_moons whose name carries the training-set size."""

    def __init__(self, hidden_size, activation, architecture, inference, epochs, lr, n_samples, warmup, n_inputs,
                 input_shape, output_size):
        super(MoonsBNN, self).__init__("half_moons", hidden_size, activation, architecture, inference, epochs, lr, n_samples,
                                       warmup, input_shape, output_size, step_size=0.001)
        self.name = self.get_name(n_inputs)


def _combinations(*axes):
    return list(itertools.product(*axes))


def serial_compute_grads(hidden_size, activation, architecture, inference, epochs, lr, n_samples, warmup, n_inputs,
                         posterior_samples, rel_path, x_test, y_test, device="cuda"):
    """:94-102 (+ _compute_grads :66-78): loss_gradients of every model x posterior_samples; pickles land where the
    reference puts them (DATA + <bnn.name>/ + <bnn.name>_samp=<S>_lossGrads.pkl).  Returns {(bnn.name, S): ndarray}."""
    input_shape, output_size = tuple(x_test.shape[1:]), int(y_test.shape[-1])
    loader = DataLoader(dataset=list(zip(x_test, y_test)), batch_size=32, shuffle=False)
    out = {}
    for (h, act, arch, inf, ep, lr_, ns, wu, ninp, psamp) in _combinations(hidden_size, activation, architecture, inference, epochs,
                                                                       lr, n_samples, warmup, n_inputs, posterior_samples):
        bnn = MoonsBNN(h, act, arch, inf, ep, lr_, ns, wu, ninp, input_shape, output_size)
        bnn.load(device=device, rel_path=rel_path)
        out[(bnn.name, psamp)] = loss_gradients(net=bnn, n_samples=psamp, savedir=bnn.name + "/", data_loader=loader,
                                                device=device, filename=bnn.name)
    return out


def grid_attack(method, hidden_size, activation, architecture, inference, epochs, lr, n_samples, warmup, n_inputs,
                posterior_samples, x_test, y_test, device="cuda", rel_path=TESTS):
    """:133-153: every model is loaded once and attacked with each number of posterior samples.  Returns
    {(bnn.name, S): x_attack}; the attack pickles / PNGs are written by `attack` as in the reference."""
    input_shape, output_size = tuple(x_test.shape[1:]), int(y_test.shape[-1])
    out = {}
    for init in _combinations(hidden_size, activation, architecture, inference, epochs, lr, n_samples, warmup, n_inputs):
        bnn = MoonsBNN(*init, input_shape, output_size)
        bnn.load(device=device, rel_path=rel_path)
        for p_samp in posterior_samples:
            out[(bnn.name, p_samp)] = attack(net=bnn, x_test=x_test, y_test=y_test, dataset_name="half_moons", device=device,
                                             method=method, filename=bnn.name, n_samples=p_samp)
    return out
