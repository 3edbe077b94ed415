"""eps x n_samples attack grid — the call surface of the reference's plot_eps_attacks.py:9-42.

`build_eps_attacks_df` keeps the reference's loop over (epsilon, n_samples), its `attack` + `attack_evaluation`
calls and its CSV schema (one row per test point: attack_method, epsilon, test_acc, adv_acc, softmax_rob,
n_samples); each grid cell is one batched GPU run over all points, and the posterior stays resident across cells.
Plotting (plot_eps_attacks.py:45-83) needs seaborn and is out of scope.
"""
import os

import pandas

from .adversarialAttacks import attack, attack_evaluation
from .savedir import DATA


def build_eps_attacks_df(bnn, dataset, device, method, x_test, y_test, epsilon_list, n_samples_list, savedir):
    """plot_eps_attacks.py:9-39"""
    rows = []
    for epsilon in epsilon_list:
        for n_samples in n_samples_list:
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
    return df


def load_eps_attacks_df(dataset, method, savedir):
    """plot_eps_attacks.py:41-42"""
    return pandas.read_csv(DATA + savedir + "/" + str(dataset) + "_increasing_eps_" + str(method) + ".csv")
