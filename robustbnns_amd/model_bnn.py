"""Bayesian network — the call surface of the reference's model_bnn.BNN (model_bnn.py:69-391).

`forward(inputs, n_samples, avg_posterior, seeds)` is the Monte-Carlo posterior predictive of
model_bnn.py:198-258, computed for the whole batch and all samples by the HIP kernels:

  hmc  the stored chain is one StackedPosterior; `seeds` index it (model_bnn.py:246-252);
  svi  weights are drawn as loc + softplus(scale) * eps (model_bnn.py:124-130).  On the GPU ONE kernel (fc / fc2:
       rbnn_svi_draw, writing the fp32 stack AND the packed / triple images; conv: rbnn_svi_draw_flat + the image
       builders; Philox eps in registers) redraws all S samples IN PLACE, so the posterior, its engine and its workspaces
       are reused draw after draw and a redraw contains no device->host sync.  RBNN_SVI_RNG=host: eps from torch's CPU
       generator in the guide's order -> rbnn_svi_materialize -> a new stacked posterior per draw.
       PARITY UNPINNED for the draw itself (pyro-ppl 1.3.0 is not available to check RNG order against); everything
       downstream of explicit weights is pinned.

Inference (SVI/HMC training, model_bnn.py:260-365) is out of scope: posteriors are inputs here.
"""
import os
import random

import numpy as np
import torch
from torch import nn

from . import _hip
from .factory import make_engine, posterior_from_stacked, posterior_from_state_dicts
from .model_nn import NN
from .savedir import TESTS

saved_BNNs = {"model_0": ["mnist", {"hidden_size": 512, "activation": "leaky", "architecture": "conv", "inference": "svi", "epochs": 5, "lr": 0.01, "n_samples": None, "warmup": None}],
              "model_1": ["mnist", {"hidden_size": 512, "activation": "leaky", "architecture": "fc2", "inference": "hmc", "epochs": None, "lr": None, "n_samples": 100, "warmup": 50}],
              "model_2": ["fashion_mnist", {"hidden_size": 1024, "activation": "leaky", "architecture": "conv", "inference": "svi", "epochs": 10, "lr": 0.001, "n_samples": None, "warmup": None}],
              "model_3": ["fashion_mnist", {"hidden_size": 1024, "activation": "leaky", "architecture": "fc2", "inference": "hmc", "epochs": None, "lr": None, "n_samples": 100, "warmup": 50}],
              "model_4": ["fashion_mnist", {"hidden_size": 1024, "activation": "leaky", "architecture": "conv", "inference": "svi", "epochs": 5, "lr": 0.01, "n_samples": None, "warmup": None}],
              "model_5": ["mnist", {"hidden_size": 512, "activation": "leaky", "architecture": "fc2", "inference": "svi", "epochs": 10, "lr": 0.01, "n_samples": None, "warmup": None}],
              "model_6": ["mnist", {"hidden_size": 256, "activation": "leaky", "architecture": "conv", "inference": "svi", "epochs": 10, "lr": 0.05, "n_samples": None, "warmup": None}],
              "model_7": ["mnist", {"hidden_size": 1024, "activation": "leaky", "architecture": "fc2", "inference": "svi", "epochs": 5, "lr": 0.02, "n_samples": None, "warmup": None}],
              "model_8": ["mnist", {"hidden_size": 1024, "activation": "leaky", "architecture": "conv", "inference": "svi", "epochs": 10, "lr": 0.02, "n_samples": None, "warmup": None}],
              "model_9": ["fashion_mnist", {"hidden_size": 512, "activation": "leaky", "architecture": "fc", "inference": "hmc", "epochs": None, "lr": None, "n_samples": 100, "warmup": 100}]}


def set_rng_seed(seed):
    """What pyro.set_rng_seed does (torch + random + numpy) — model_bnn.py:224,358,374."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def param_store_state(loc, scale):
    """The dict pyro 1.3.0's ParamStoreDict.save() pickles (pyro/params/param_store.py get_state(), [recalled]; the reference
    writes it at model_bnn.py:148-155): {"params": {name: UNCONSTRAINED value, a leaf tensor with requires_grad},
    "constraints": {name: torch.distributions constraint}}.  The guide's params `<key>_loc` / `<key>_scale` (:125-126) are
    registered without a constraint, i.e. constraints.real, for which unconstrained == constrained."""
    from torch.distributions import constraints
    params = {}
    for k, v in loc.items():
        params[k + "_loc"] = v.detach().cpu().clone().requires_grad_(True)
    for k, v in scale.items():
        params[k + "_scale"] = v.detach().cpu().clone().requires_grad_(True)
    return {"params": params, "constraints": {name: constraints.real for name in params}}


def read_param_store(path):
    """name -> CONSTRAINED fp32 tensor from a pyro param-store file (what pyro.get_param_store().load() + iteration yields,
    model_bnn.py:177-181).  Unconstrained values go through transform_to(constraint), the identity for constraints.real.
    Also accepts a bare {name: tensor} dict.  weights_only=False: the file pickles constraint objects."""
    from torch.distributions import constraints, transform_to
    store = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(store, dict):
        raise TypeError(f"{path}: malformed ParamStore state ({type(store).__name__})")
    if "params" not in store:
        return {k: v.detach().to(torch.float32) for k, v in store.items()}
    if set(store.keys()) != {"params", "constraints"}:
        raise KeyError(f"{path}: malformed ParamStore keys {sorted(store.keys())}")
    out = {}
    for name, value in store["params"].items():
        c = store["constraints"].get(name, constraints.real)
        c = constraints.real if isinstance(c, type(constraints.real)) else c       # pyro's own unpickling workaround
        value = value.detach()
        out[name] = (value if c is constraints.real else transform_to(c)(value)).to(torch.float32)
    return out


class BNN(nn.Module):

    def __init__(self, dataset_name, hidden_size, activation, architecture, inference, epochs, lr, n_samples, warmup,
                 input_shape, output_size, step_size=0.005, num_steps=10):
        super(BNN, self).__init__()
        self.dataset_name = dataset_name
        self.inference = inference
        self.architecture = architecture
        self.epochs = epochs
        self.lr = lr
        self.n_samples = n_samples
        self.warmup = warmup
        self.step_size = step_size
        self.num_steps = num_steps
        self.basenet = NN(dataset_name=dataset_name, input_shape=input_shape, output_size=output_size,
                          hidden_size=hidden_size, activation=activation, architecture=architecture, epochs=epochs, lr=lr)
        self.name = self.get_name()
        self.posterior = None                 # hmc: StackedPosterior of the chain
        self.svi_loc = self.svi_scale = None  # svi: dict key -> tensor (raw scale, softplus applied at draw)
        # SVI draws.  "device" (default): eps comes from the GPU generator — fresh from the live generator without seeds (the
        # reference draws from the live global RNG too), one generator re-seeded per seed otherwise (same seed -> same weights,
        # whatever the other seeds are).  "host": a CPU restatement of the guide's draw order (2 discarded tensors per key, then
        # one per parameter) — same-seed parity with pyro is unpinned either way, and at MNIST fc-512 the
        # host loop costs 3.7 s per 100 draws against 2.4 ms for the whole forward.
        self.svi_rng = os.environ.get("RBNN_SVI_RNG", "device")
        self._engine = None
        self._drawn = None                    # svi: (key, engine) of the last SEEDED draw (same seeds -> same weights: reused, not redrawn)
        self._guide = None                    # svi, fc / fc2: posterior.SviGuide of the current parameters (bounds fixed once per guide)
        self._slots = {}                      # svi: n_samples -> (StackedPosterior.for_guide, engine) redrawn in place by unseeded calls
        self._draws = 0                       # number of in-place redraws so far

    def get_name(self, n_inputs=None):
        """model_bnn.py:90-103"""
        name = str(self.dataset_name) + "_bnn_" + str(self.inference) + "_hid=" + str(self.basenet.hidden_size) + \
               "_act=" + str(self.basenet.activation) + "_arch=" + str(self.basenet.architecture)
        if n_inputs:
            name = name + "_inp=" + str(n_inputs)
        if self.inference == "svi":
            return name + "_ep=" + str(self.epochs) + "_lr=" + str(self.lr)
        elif self.inference == "hmc":
            return name + "_samp=" + str(self.n_samples) + "_warm=" + str(self.warmup) + \
                   "_stepsize=" + str(self.step_size) + "_numsteps=" + str(self.num_steps)

    # ------------------------------------------------------------------ posterior in
    def _make_posterior(self, stacked, device):
        b = self.basenet
        return posterior_from_stacked(b.architecture, b.activation, b.input_shape, b.output_size, b.hidden_size, stacked, device)

    def set_posterior_samples(self, samples, device):
        """hmc: `samples` = list of NN.state_dict()s (what load() reads from disk), or a dict key -> [S,...]."""
        self.device = device
        self.basenet.device = device
        if isinstance(samples, dict):
            self.posterior = self._make_posterior(samples, device)
        else:
            b = self.basenet
            self.posterior = posterior_from_state_dicts(samples, b.architecture, b.activation, b.input_shape,
                                                        b.output_size, b.hidden_size, device)
        self._engine = make_engine(self.posterior)

    def set_variational_params(self, loc, scale, device):
        """svi: dicts state_dict-key -> tensor, the `<key>_loc` / `<key>_scale` params of model_bnn.py:125-126."""
        self.device = device
        self.basenet.device = device
        self.svi_loc = {k: v.detach().to(device, torch.float32).contiguous() for k, v in loc.items()}
        self.svi_scale = {k: v.detach().to(device, torch.float32).contiguous() for k, v in scale.items()}
        self._engine = None
        self._drawn, self._guide, self._slots = None, None, {}     # nothing drawn from the previous guide may be served for this one

    @property
    def posterior_predictive(self):
        """dict idx -> NN carrying sample idx (the reference's attribute, model_bnn.py:186-190); built on demand."""
        import copy
        out = {}
        for i in range(self.posterior.S):
            net = copy.deepcopy(self.basenet)
            net.load_state_dict(self.posterior.state_dict(i))
            out[i] = net
        return out

    def save(self, rel_path=TESTS, filename=None):
        """hmc branch of model_bnn.py:138-165: one state_dict file per stored sample."""
        if filename is None:
            filename = self.name + "_weights"
        path = rel_path + self.name + "/"
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if self.inference == "hmc":
            for key, value in self.posterior_predictive.items():
                torch.save(value.state_dict(), path + filename + "_" + str(key) + ".pt")
        else:
            torch.save(param_store_state(self.svi_loc, self.svi_scale), path + filename + ".pt")

    def load(self, device, rel_path=TESTS, filename=None):
        """model_bnn.py:167-196"""
        if filename is None:
            filename = self.name + "_weights"
        path = rel_path + self.name + "/"
        if self.inference == "svi":
            params = read_param_store(path + filename + ".pt")
            keys = list(self.basenet.state_dict().keys())
            missing = [k + sfx for k in keys for sfx in ("_loc", "_scale") if k + sfx not in params]
            if missing:
                raise KeyError(f"param store {path + filename}.pt lacks {missing[:4]}{'...' if len(missing) > 4 else ''} "
                               f"(has {sorted(params)[:4]}...): not written by this architecture's guide (model_bnn.py:124-126)")
            self.set_variational_params({k: params[k + "_loc"] for k in keys}, {k: params[k + "_scale"] for k in keys}, device)
            print("\nLoading ", path + filename + ".pt\n")
        elif self.inference == "hmc":
            sds = [torch.load(path + filename + "_" + str(i) + ".pt", map_location="cpu") for i in range(self.n_samples)]
            if len(sds) != self.n_samples:
                raise AttributeError("wrong number of posterior models")
            self.set_posterior_samples(sds, device)

    # ------------------------------------------------------------------ svi draws
    def _svi_eps(self, n_samples, seeds):
        """eps per draw in the order the guide consumes the RNG ([recalled]): for every
        state_dict key two discarded randn_like (the eagerly evaluated pyro.param initialisers,
        model_bnn.py:125-126), then one standard normal per parameter in named_parameters() order."""
        shapes = [(k, tuple(v.shape)) for k, v in self.basenet.state_dict().items()]
        total = sum(int(np.prod(s)) for _, s in shapes)
        if self.svi_rng == "device" and torch.device(self.device).type == "cuda":
            if not seeds:
                return torch.randn(n_samples, total, device=self.device, dtype=torch.float32)
            eps = torch.empty(n_samples, total, device=self.device, dtype=torch.float32)
            gen = torch.Generator(device=self.device)
            for i in range(n_samples):
                gen.manual_seed(int(seeds[i]))
                torch.randn(total, generator=gen, device=self.device, dtype=torch.float32, out=eps[i])
            return eps
        eps = torch.empty(n_samples, total, dtype=torch.float32)
        for i in range(n_samples):
            if seeds:
                set_rng_seed(seeds[i])
            for _, shp in shapes:
                torch.randn(shp)
                torch.randn(shp)
            off = 0
            for _, shp in shapes:
                n = int(np.prod(shp))
                eps[i, off:off + n] = torch.randn(shp).reshape(-1)
                off += n
        return eps.to(self.device)

    def draw_posterior(self, n_samples, seeds=None):
        """n_samples weight draws as a StackedPosterior (model_bnn.py:124-130, :222-232)."""
        keys = list(self.basenet.state_dict().keys())
        loc = torch.cat([self.svi_loc[k].reshape(-1) for k in keys])
        scale = torch.cat([self.svi_scale[k].reshape(-1) for k in keys])
        eps = self._svi_eps(n_samples, seeds).contiguous()
        out = torch.empty_like(eps)
        _hip.HipKernels().svi_materialize(loc, scale, eps, out)
        stacked, off = {}, 0
        for k in keys:
            shp = tuple(self.svi_loc[k].shape)
            n = int(np.prod(shp))
            stacked[k] = out[:, off:off + n].reshape((n_samples,) + shp)
            off += n
        return self._make_posterior(stacked, self.device)

    # ------------------------------------------------------------------ svi: in-place redraw (fc / fc2 on the GPU)
    def _in_place(self):
        ok = (self.inference == "svi" and self.basenet.architecture in ("fc", "fc2", "conv") and self.svi_rng == "device"
              and torch.device(self.device).type == "cuda")
        if ok and self.basenet.architecture != "conv":
            # rbnn_svi_draw stages W2 [C, H] in LDS (160 KB: hidden <= 4096 at 10 classes); a wider net keeps the older path
            # (draw_posterior: rbnn_svi_materialize into a new stack), which has no such limit
            b = self.basenet
            ok = int(b.output_size) * max(32, int(b.hidden_size)) * 4 <= 160 * 1024
        return ok

    def _new_slot(self, n_samples):
        b = self.basenet
        if b.architecture == "conv":
            from .conv import ConvStackedPosterior, ConvSviGuide
            if self._guide is None:
                self._guide = ConvSviGuide(self.svi_loc, self.svi_scale, self.device)
            post = ConvStackedPosterior.for_guide(self._guide, b.activation, b.input_shape, b.output_size, b.hidden_size, n_samples)
        else:
            from .posterior import StackedPosterior, SviGuide
            if self._guide is None:
                self._guide = SviGuide(self.svi_loc, self.svi_scale, b.architecture, self.device)
            post = StackedPosterior.for_guide(self._guide, b.activation, b.input_shape, b.output_size, n_samples)
        eng = make_engine(post)
        if eng.precision == "triple":
            post.triple_images()                          # allocated once; every redraw writes them in the draw kernel itself
        elif eng.precision == "split":
            post.split_images()
        return post, eng

    def _fresh_key(self):
        """64 bits from torch's global CPU generator — the stream the reference's draws advance (model_bnn.py:230-232 under
        pyro.set_rng_seed): set_rng_seed(k) makes the following unseeded draws reproducible.  Host arithmetic only."""
        return int(torch.randint(-(2 ** 63), 2 ** 63 - 1, (1,), dtype=torch.int64).item())

    def redraw(self, n_samples):
        """Fresh weights for the `n_samples` samples of the resident SVI stack (what every un-seeded forward of the reference does,
        model_bnn.py:230-232): one kernel launch, nothing allocated, no device->host sync.  Returns the (unchanged) engine."""
        slot = self._slots.get(n_samples)
        if slot is None:
            if len(self._slots) >= 2:                         # e.g. an attack with S samples scored with S' defence samples
                self._slots.pop(next(iter(self._slots)))
            slot = self._slots[n_samples] = self._new_slot(n_samples)
        self._draws += 1
        # the key alone identifies the draw: reproducible under set_rng_seed.  lowdim engines (half-moons): the draw is left pending and generated
        # inside the next rbnn_lowdim_run launch (posterior.StackedPosterior.lazy_capable); anything else that reads the stack materialises it
        if getattr(slot[1], "precision", None) in ("lowdim", "triple") and hasattr(slot[0], "lazy_capable"):
            slot[0].redraw(self._fresh_key(), 0, lazy=True)
        else:
            slot[0].redraw(self._fresh_key(), 0)
        return slot[1]

    # ------------------------------------------------------------------ hot path handles
    def hot_path(self, n_samples, avg_posterior=False, seeds=None):
        """(engine, n_samples, seeds, logits) to run `n_samples` posterior samples through the kernels."""
        if seeds:
            if len(seeds) != n_samples:
                raise ValueError("Number of seeds should match number of samples.")     # model_bnn.py:200-202
        if self.inference == "hmc":
            return self._engine, n_samples, seeds, False
        if avg_posterior is True:                         # model_bnn.py:206-216: logits of the mean weights
            stacked = {k: v.unsqueeze(0) for k, v in self.svi_loc.items()}
            return make_engine(self._make_posterior(stacked, self.device)), 1, None, True
        # (identity, version) of every variational tensor: an in-place edit bumps _version, a REPLACED tensor (net.svi_loc[k] = new, version 0
        # again) changes data_ptr — either way the guide's bounds (hard bounds of the fp16 piece scaling), stacks and seeded draws are stale
        ver = tuple((t.data_ptr(), t._version) for t in list(self.svi_loc.values()) + list(self.svi_scale.values()))
        if ver != getattr(self, "_guide_ver", None):
            self._guide_ver, self._drawn, self._guide, self._slots = ver, None, None, {}
        if not seeds:                                     # the reference draws from the live RNG: fresh weights on every call
            if self._in_place():
                return self.redraw(n_samples), n_samples, None, False
            return make_engine(self.draw_posterior(n_samples, seeds)), n_samples, None, False
        # seeded draws are a pure function of (seeds, loc, scale): evaluate(), attack_evaluation() and the drivers call forward() batch
        # after batch with the same seeds — keep the last drawn posterior (weights, packed images, workspaces) instead of re-materialising
        # it per call.  set_variational_params() drops it, so a reloaded guide can never be served an older guide's draw.
        key = (int(n_samples), tuple(int(v) for v in seeds), self.svi_rng)
        if self._drawn is None or self._drawn[0] != key:
            if self._in_place():
                post, eng = self._new_slot(n_samples)
                keys = torch.tensor([int(v) for v in seeds], dtype=torch.int64).to(self.device)      # seed i -> Philox key i: the draw for a
                post.redraw(0, 0, sample_keys=keys)                                                  # seed does not depend on its position
                self._drawn = (key, eng)
            else:
                self._drawn = (key, make_engine(self.draw_posterior(n_samples, seeds)))
        return self._drawn[1], n_samples, None, False

    def forward(self, inputs, n_samples=10, avg_posterior=False, seeds=None):
        """model_bnn.py:198-258 -> mean probabilities [B, C] (raw logits if avg_posterior, :216)."""
        eng, S, sd, logits = self.hot_path(n_samples, avg_posterior, seeds)
        return eng.forward(inputs.to(self.device), S, seeds=sd, logits=logits)

    def train(self, *args, **kwargs):
        if args and isinstance(args[0], bool) or "mode" in kwargs:
            return super().train(*args, **kwargs)
        raise NotImplementedError("SVI/HMC inference is outside the accelerated hot path (SURVEY.md section 2, row 8): "
                                  "run it with the reference and load the posterior here")

    def evaluate(self, test_loader, device, n_samples=10, seeds_list=None):
        """model_bnn.py:367-391"""
        random.seed(0)
        set_rng_seed(0)
        bnn_seeds = list(range(n_samples)) if seeds_list is None else seeds_list
        correct = 0.0
        for x_batch, y_batch in test_loader:
            outputs = self.forward(x_batch.to(device), n_samples=n_samples, seeds=bnn_seeds)
            correct += float((outputs.argmax(-1) == y_batch.to(device).argmax(-1)).sum())
        accuracy = 100 * correct / len(test_loader.dataset)
        print("Accuracy: %.2f%%" % (accuracy))
        return accuracy
