"""Deterministic vs Bayesian vs ensemble attack table — the call surface of the reference's plot_baseline_attacks.py:10-145.

`build_baseline_attacks_df` keeps the reference's three sections (NN, BNN with attack_samples x defence_samples, Ensemble_NN
with n_samples in {1, 50, 100}), its `attack` + `attack_evaluation` calls, its DataFrame schema (one row per test point:
attack_method, epsilon, test_acc, adv_acc, softmax_rob, attack_samples, defence_samples, model_type) and CSV path.
The reference loads the dataset and the three saved models inside the function (`load_dataset`, `NN.load`, `BNN.load`,
`Ensemble_NN.load`, plot_baseline_attacks.py:24-34,56-57,100-107); dataset loading and training are out of scope here
(DESIGN.md section 7), so the caller passes the loaded nets and test tensors.  Every (model, n_samples) cell is one batched
GPU run over all points.  Plotting (:147-) needs seaborn and is out of scope.
"""
import os

import pandas

from .adversarialAttacks import attack, attack_evaluation
from .savedir import TESTS

COLUMNS = ["attack_method", "epsilon", "test_acc", "adv_acc", "softmax_rob", "attack_samples", "defence_samples", "model_type"]


def build_baseline_attacks_df(nn, bnn, ensemble, dataset_name, device, attack_method, x_test, y_test,
                              bayesian_attack_samples=(1,), bayesian_defence_samples=(1, 50, 100), n_samples_list=(1, 50, 100)):
    """plot_baseline_attacks.py:10-130.  `epsilon` is the attack default the reference records, 0.3 (:21; the attacks are
    called without hyperparams, so fgsm uses 0.3 and pgd its own defaults, adversarialAttacks.py:71,91)."""
    rows = []
    epsilon = 0.3

    def add(model_type, test_acc, adv_acc, softmax_rob, attack_samples, defence_samples):
        for pointwise_rob in softmax_rob.cpu().tolist():
            rows.append({"model_type": model_type, "attack_method": attack_method, "epsilon": epsilon, "test_acc": test_acc,
                         "adv_acc": adv_acc, "softmax_rob": pointwise_rob, "attack_samples": attack_samples,
                         "defence_samples": defence_samples})

    if nn is not None:                                                                   # :23-54
        nn_attack = attack(net=nn, x_test=x_test, y_test=y_test, dataset_name=dataset_name, device=device,
                           method=attack_method, filename=nn.name)
        add("nn", *attack_evaluation(net=nn, x_test=x_test, x_attack=nn_attack, y_test=y_test, device=device), 1, None)
    if bnn is not None:                                                                  # :56-88
        for attack_samples in bayesian_attack_samples:
            bnn_attack = attack(net=bnn, x_test=x_test, y_test=y_test, dataset_name=dataset_name, device=device,
                                method=attack_method, filename=bnn.name, n_samples=attack_samples)
            for defence_samples in bayesian_defence_samples:
                add("bnn", *attack_evaluation(net=bnn, x_test=x_test, x_attack=bnn_attack, y_test=y_test, device=device,
                                              n_samples=defence_samples), attack_samples, defence_samples)
    if ensemble is not None:                                                             # :90-125
        for n_samples in n_samples_list:
            ens_attack = attack(net=ensemble, x_test=x_test, y_test=y_test, dataset_name=dataset_name, device=device,
                                method=attack_method, filename=ensemble.name, n_samples=n_samples)
            add("ensemble", *attack_evaluation(net=ensemble, x_test=x_test, x_attack=ens_attack, y_test=y_test, device=device,
                                               n_samples=n_samples), n_samples, n_samples)
    df = pandas.DataFrame(rows, columns=COLUMNS)
    return _save_baseline_attacks_df(df, dataset_name, attack_method)


def _save_baseline_attacks_df(df, dataset_name, attack_method):
    """plot_baseline_attacks.py:132-139"""
    print("\nSaving:", df)
    os.makedirs(os.path.dirname(TESTS + "/"), exist_ok=True)
    df.to_csv(TESTS + "/" + str(dataset_name) + "_baseline_attacks_" + str(attack_method) + ".csv", index=False, header=True)
    return df


def load_baseline_attacks_df(dataset_name, attack_method, savedir):
    """plot_baseline_attacks.py:141-145"""
    df = pandas.read_csv(TESTS + savedir + "/" + str(dataset_name) + "_baseline_attacks_" + str(attack_method) + ".csv")
    print(df.head(300))
    return df
