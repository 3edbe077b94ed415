"""robustbnns_amd — MI355X-native Bayesian attack / expected-loss-gradient hot path.

Drop-in for the hot path of ginevracoal/robustBNNs (model_bnn.BNN.forward -> lossGradients.loss_gradient(s)
-> adversarialAttacks.{fgsm,pgd}_attack / attack / attack_evaluation): same module and function names,
same signatures, results within 1e-5 of the reference's CPU path; the arithmetic runs in hand-written
gfx950 HIP kernels behind the C-ABI of include/robustbnns_hip.h.  See DESIGN.md.
"""
from . import adversarialAttacks, grid_search_halfMoons, lossGradients, model_bnn, model_ensemble, model_nn, plot_baseline_attacks, plot_eps_attacks, plot_gradients_components   # noqa: F401
from .conv import ConvEngine, ConvStackedPosterior                                      # noqa: F401
from .engine import AttackEngine                                                        # noqa: F401
from .posterior import StackedPosterior                                                 # noqa: F401

__all__ = ["AttackEngine", "StackedPosterior", "adversarialAttacks", "lossGradients", "model_bnn",
           "model_ensemble", "model_nn", "plot_baseline_attacks", "plot_eps_attacks", "plot_gradients_components", "grid_search_halfMoons"]
