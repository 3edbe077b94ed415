"""Deep-ensemble baseline — the call surface of the reference's model_ensemble.Ensemble_NN.

forward = mean of the LOGITS of the first n_samples members (model_ensemble.py:57-67): the same stacked
kernels as the BNN with RBNN_OUT_LOGITS / RBNN_LOSS_MEAN_LOGIT.
"""
from .model_nn import NN
from .savedir import TESTS


class Ensemble_NN(NN):

    def __init__(self, dataset_name, hidden_size, activation, architecture, epochs, lr, input_shape, output_size, ensemble_size):
        super(Ensemble_NN, self).__init__(dataset_name, input_shape, output_size, hidden_size, activation, architecture, lr, epochs)
        self.ensemble_size = ensemble_size
        self.random_seeds = range(0, ensemble_size)
        self.name = self.get_name(ensemble_size)
        self.ensemble_models = {}
        self._ens_engine = None

    def get_name(self, ensemble_size, *args, **kwargs):
        return str(self.dataset_name) + "_ensemble_hid=" + str(self.hidden_size) + "_act=" + str(self.activation) + \
               "_arch=" + str(self.architecture) + "_size=" + str(ensemble_size)

    def load(self, device, rel_path=TESTS):
        """model_ensemble.py:44-55"""
        self.device = device
        savedir = self.name + "/weights"
        for seed in self.random_seeds:
            net = NN(dataset_name=self.dataset_name, input_shape=self.input_shape, output_size=self.output_size,
                     hidden_size=self.hidden_size, activation=self.activation, architecture=self.architecture,
                     epochs=self.epochs, lr=self.lr)
            net.load(device=device, savedir=savedir, seed=seed, rel_path=rel_path)
            self.ensemble_models[str(seed)] = net
        self._ens_engine = None

    def engine(self, device):
        if self._ens_engine is None or str(self._ens_engine.device) != str(device):
            from .factory import make_engine, posterior_from_modules
            self._ens_engine = make_engine(posterior_from_modules(list(self.ensemble_models.values()), device))
        return self._ens_engine

    def forward(self, inputs, n_samples, *args, **kwargs):
        """model_ensemble.py:57-67"""
        if n_samples is not None:
            if n_samples > self.ensemble_size:
                raise ValueError("Maximum number of samples allowed is ", self.ensemble_size)
        n = len(self.ensemble_models) if n_samples is None else n_samples
        device = getattr(self, "device", inputs.device)
        return self.engine(device).forward(inputs.to(device), n_samples=n, logits=True)

    def evaluate(self, test_loader, device, n_samples, *args, **kwargs):
        """model_ensemble.py:85-106"""
        if n_samples > self.ensemble_size:
            raise ValueError("Maximum number of samples allowed is ", self.ensemble_size)
        self.device = device
        correct = 0.0
        for x_batch, y_batch in test_loader:
            outputs = self.forward(x_batch.to(device), n_samples=n_samples)
            correct += float((outputs.argmax(-1) == y_batch.to(device).argmax(-1)).sum())
        accuracy = 100 * correct / len(test_loader.dataset)
        print("\nAccuracy: %.2f%%" % (accuracy))
        return accuracy
