"""eps x n_samples attack grid — the call surface of the reference's plot_eps_attacks.py:9-42.

`build_eps_attacks_df` keeps the reference's loop over (epsilon, n_samples), the side effects and results of its `attack` +
`attack_evaluation` calls and its CSV schema (one row per test point: attack_method, epsilon, test_acc, adv_acc, softmax_rob,
n_samples); each grid cell is one batched GPU run over all points and the posterior stays resident across cells.  An FGSM grid on
a posterior of stored samples is ONE resident job (adversarialAttacks.FgsmGrid): the gradient does not depend on epsilon and neither
does the evaluation's forward of the clean inputs, so the reference's 5 x 3 grid (plot_eps_attacks.py:89-90) costs 3 gradient passes
and 3 clean forwards instead of 15 + 15; per cell what is left is one sign / clamp launch and the forward of the adversarial inputs.
Plotting (plot_eps_attacks.py:45-83) needs seaborn and is out of scope.
"""
import os

import pandas

from .adversarialAttacks import FgsmGrid, attack, attack_evaluation
from .savedir import DATA


def build_eps_attacks_df(bnn, dataset, device, method, x_test, y_test, epsilon_list, n_samples_list, savedir):
    """plot_eps_attacks.py:9-39"""
    rows = []
    grid = FgsmGrid(bnn, x_test, y_test, dataset, device) if method == "fgsm" else None
    for epsilon in epsilon_list:
        for n_samples in n_samples_list:
            if grid is not None:        # (an SVI net: FgsmGrid makes the two calls below as they are — fresh draws per forward, nothing to share)
                x_attack = grid.attack(epsilon, n_samples, filename=bnn.name)
                test_acc, adv_acc, softmax_rob = grid.evaluate(x_attack, n_samples)
            else:
                x_attack = attack(net=bnn, x_test=x_test, y_test=y_test, dataset_name=dataset, device=device, method=method,
                                  filename=bnn.name, n_samples=n_samples, hyperparams={"epsilon": epsilon})
                test_acc, adv_acc, softmax_rob = attack_evaluation(net=bnn, x_test=x_test, n_samples=n_samples,
                                                                   x_attack=x_attack, y_test=y_test, device=device)
            for pointwise_rob in softmax_rob.cpu().tolist():
                rows.append({"attack_method": method, "epsilon": epsilon, "test_acc": test_acc, "adv_acc": adv_acc,
                             "softmax_rob": pointwise_rob, "n_samples": n_samples})
    df = pandas.DataFrame(rows, columns=["attack_method", "epsilon", "test_acc", "adv_acc", "softmax_rob", "n_samples"])
    print("\nSaving:", df)
    os.makedirs(os.path.dirname(DATA + savedir + "/"), exist_ok=True)
    df.to_csv(DATA + savedir + "/" + str(dataset) + "_increasing_eps_" + str(method) + ".csv", index=False, header=True)
    if grid is not None:
        df.attrs["grid_cost"] = {"cells": len(epsilon_list) * len(n_samples_list), "gradient_passes": grid.gradient_passes if grid.shared else None,
                                 "clean_forwards": grid.clean_forwards if grid.shared else None, "shared": grid.shared}
    return df


def load_eps_attacks_df(dataset, method, savedir):
    """plot_eps_attacks.py:41-42"""
    return pandas.read_csv(DATA + savedir + "/" + str(dataset) + "_increasing_eps_" + str(method) + ".csv")
