"""Pick the stacked-posterior class and engine for an architecture (fc / fc2 -> MFMA GEMM path, conv -> conv path)."""
from .conv import ConvEngine, ConvStackedPosterior
from .engine import AttackEngine
from .posterior import StackedPosterior


def posterior_from_stacked(arch, activation, input_shape, n_classes, hidden, stacked, device):
    if arch == "conv":
        return ConvStackedPosterior(activation, input_shape, n_classes, hidden, stacked, device)
    return StackedPosterior(arch, activation, input_shape, n_classes, hidden, stacked, device)


def posterior_from_state_dicts(state_dicts, arch, activation, input_shape, n_classes, hidden, device):
    if arch == "conv":
        return ConvStackedPosterior.from_state_dicts(state_dicts, activation, input_shape, n_classes, hidden, device)
    return StackedPosterior.from_state_dicts(state_dicts, arch, activation, input_shape, n_classes, hidden, device)


def posterior_from_modules(nets, device):
    n0 = nets[0]
    return posterior_from_state_dicts([n.state_dict() for n in nets], n0.architecture, n0.activation, n0.input_shape,
                                      n0.output_size, n0.hidden_size, device)


def make_engine(post, kernels=None, group=None, total_samples=None, precision=None):
    if getattr(post, "arch", None) == "conv":
        return ConvEngine(post, kernels=kernels, group=group, total_samples=total_samples, precision=precision)
    return AttackEngine(post, kernels=kernels, group=group, total_samples=total_samples, precision=precision)
