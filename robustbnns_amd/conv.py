"""The reference's `conv` architecture (model_nn.py:93-106) on the HIP path: stacked posterior + batched engine.

Layout in HBM (all S samples resident): K1w [S,32,Cin*25], K1b [S,32], K2w [S,Hc,800] (k = ci*25 + ky*5 + kx, exactly
nn.Conv2d's [out,in,kh,kw] flattened), K2b [S,Hc], Fw [S,C,NP2*Hc], Fb [S,C].  Activations per (sample, point):
P1 18 KB, Q2 196*Hc B, two byte stashes — 148 KB at Hc=512 on 1x28x28, so large jobs are run in blocks of points.

Input geometries (include/robustbnns_hip.h): 1x28x28 — the only one the reference's conv accepts (model_nn.py:95-96) — and
3x32x32, BASELINE.json's CIFAR-shaped config 5, whose head Linear(81*Hc, C) is build-defined (SURVEY 8a note; parity unpinned).
"""
import ctypes as C
import os

import torch

from . import _hip
from .engine import AttackEngine, to_labels
from .posterior import scale_exp
from ._hip import OUT_LOGITS, OUT_PROBS

CONV_KEYS = ("model.0", "model.3", "model.7")
_U8 = {"st1", "st2"}
GEOMETRIES = {(1, 28, 28), (3, 32, 32)}


def conv_geometry(input_shape):
    """(pooled conv1 width, pooled conv2 width) of model_nn.py:98-106 on a Cin x W x W input: 5x5 convs, pool 2, pool 2 stride 1."""
    p1 = (int(input_shape[1]) - 4) // 2
    return p1, p1 - 4 - 1


class ConvSviGuide:
    """Variational parameters of a conv guide (model_bnn.py:124-126) on the device + the bounds that are fixed once per guide, so a
    redraw (ConvStackedPosterior.redraw) contains no pass over the drawn weights and no device->host sync — as posterior.SviGuide:
    |w| <= |loc| + RBNN_SVI_EPS_MAX * softplus(scale)."""
    TENSOR_IDS = {"model.0.weight": 0, "model.0.bias": 1, "model.3.weight": 2, "model.3.bias": 3, "model.7.weight": 4, "model.7.bias": 5}

    def __init__(self, loc, scale, device):
        self.device = torch.device(device)
        self.loc = {k: loc[k].detach().to(self.device, torch.float32).contiguous() for k in self.TENSOR_IDS}
        self.scale = {k: scale[k].detach().to(self.device, torch.float32).contiguous() for k in self.TENSOR_IDS}
        self.sigma = {k: torch.nn.functional.softplus(v).contiguous() for k, v in self.scale.items()}     # once per guide (as posterior.SviGuide)
        bnd = {k: self.loc[k].abs() + _hip.SVI_EPS_MAX * self.sigma[k] for k in self.loc}
        k2 = bnd["model.3.weight"]
        typ = (self.loc["model.3.weight"].abs() + 0.8 * self.sigma["model.3.weight"]).double().mean().float()
        Cn = self.loc["model.7.bias"].numel()
        rec = torch.stack([k2.max(), typ, bnd["model.0.weight"].reshape(32, -1).sum(-1).max(), bnd["model.0.bias"].max(),
                           bnd["model.7.weight"].reshape(Cn, -1).sum(0).max()]).cpu().tolist()           # the one sync, at load
        self.k2_max, self.p1_bound, self.fw_l1 = rec[0], (rec[2], rec[3]), rec[4]
        self.range_ok = 0.0 < rec[0] <= 4096.0 * rec[1] and rec[0] != float("inf")


class ConvStackedPosterior:
    arch = "conv"

    def __init__(self, activation, input_shape, n_classes, hidden, stacked, device):
        shape = tuple(int(v) for v in input_shape)
        if shape not in GEOMETRIES:
            raise NotImplementedError(f"conv on the HIP path is built for inputs {sorted(GEOMETRIES)}, not {shape}")
        if activation not in _hip.ACTIVATIONS:
            raise AssertionError("\nWrong activation name.")                      # model_nn.py:74-75
        self.activation, self.input_shape = activation, shape
        self.Cin, self.W = shape[0], shape[1]
        self.D = self.Dp = shape[0] * shape[1] * shape[2]
        self.P1W, self.P2W = conv_geometry(shape)
        self.NP2 = self.P2W * self.P2W
        self.H, self.C = int(hidden), int(n_classes)
        self.device = torch.device(device)
        S = stacked["model.0.weight"].shape[0]
        self.S = int(S)
        f = lambda k, shp: stacked[k].to(self.device, torch.float32).reshape((S,) + shp).contiguous()
        self.K1w, self.K1b = f("model.0.weight", (32, self.Cin * 25)), f("model.0.bias", (32,))
        self.K2w, self.K2b = f("model.3.weight", (self.H, 800)), f("model.3.bias", (self.H,))
        self.Fw, self.Fb = f("model.7.weight", (self.C, self.NP2 * self.H)), f("model.7.bias", (self.C,))
        # model.3.weight regrouped [S, 32 ci, Hc/16 blocks, 25 taps, 16 hc]: the backward GEMM's A operand, K-contiguous
        # with one K tile = one tap x 16 channels
        self.K2ci = torch.empty(S, 32, self.H * 25, dtype=torch.float32, device=self.device)
        self._regroup_k2ci()
        self._desc = None
        self._split = None
        self._triple = None
        self._guide = None                      # ConvSviGuide: a redrawable SVI stack (for_guide / redraw)
        self._dense = None                      # triple mode: model.3.weight image of the dense conv2^T kernel (geometry-independent)

    # ------------------------------------------------------------------ redrawable SVI stack
    @classmethod
    def for_guide(cls, guide, activation, input_shape, n_classes, hidden, S):
        """S zero samples to be filled IN PLACE by redraw(): buffers, descriptor and images keep their addresses across draws."""
        shapes = {k: tuple(v.shape) for k, v in guide.loc.items()}
        post = cls(activation, input_shape, n_classes, hidden, {k: torch.zeros((S,) + shp) for k, shp in shapes.items()}, guide.device)
        post._guide = guide
        return post

    def _regroup_k2ci(self):
        S, H = self.S, self.H
        self.K2ci.view(S, 32, H // 16, 25, 16).copy_(self.K2w.view(S, H // 16, 16, 32, 25).permute(0, 3, 1, 4, 2))

    def redraw(self, key, draw_id=0, n_samples=None, sample_keys=None):
        """All six tensors of all samples redrawn by ONE launch (rbnn_svi_draw_flat) into the resident fp32 stack; the images derived from
        model.3.weight (its input-channel regrouping, the triple / split images — at scales fixed per guide) are then rebuilt in place
        by their builders: a handful of asynchronous launches, no allocation of a new posterior, no device->host sync."""
        if self._guide is None:
            raise _hip.HipError("redraw() needs a posterior built by ConvStackedPosterior.for_guide")
        g, S = self._guide, (self.S if n_samples is None else int(n_samples))
        dst = {"model.0.weight": self.K1w, "model.0.bias": self.K1b, "model.3.weight": self.K2w, "model.3.bias": self.K2b,
               "model.7.weight": self.Fw, "model.7.bias": self.Fb}
        items = [(g.loc[k], g.sigma[k], dst[k], g.TENSOR_IDS[k]) for k in dst]
        _hip.HipKernels().svi_draw_flat(items, S, int(key), int(draw_id), sample_keys)
        # an image nobody may read before the next draw is only marked stale (a 100+ MB permuted copy at Hc = 512): the fp32 kernels'
        # input-channel regrouping (refresh_lazy_images)
        self._k2ci_stale = True
        if self._triple is not None:
            if self._fused_images():            # forward rows + dense conv2^T images by ONE kernel straight from the fp32 stack
                _hip.HipKernels().conv_weight_images(self.K2w, self.S, self.H, self._triple[1], self._triple[0], self._dense)
            else:
                self._build_triple(self._triple[0])
                self._build_dense(self._dense)
        if self._split is not None:
            self._build_split(self._split[0], self._split[4])
        return self

    def refresh_lazy_images(self, k2ci=False):
        """Rebuild an image that redraw() only marked stale, right before a kernel that reads it."""
        if k2ci and getattr(self, "_k2ci_stale", False):
            self._regroup_k2ci()
            self._k2ci_stale = False

    def _k2_max(self):
        return self._guide.k2_max if self._guide is not None else float(self.K2w.abs().max())

    def _fw_l1(self):
        return self._guide.fw_l1 if self._guide is not None else float(self.Fw.abs().sum(1).max())

    def _scratch(self):
        """Staging buffers of the image builders (tap-major conv2 weights, the 26-tap padded regrouping), allocated once."""
        if getattr(self, "_tmp", None) is None:
            S, H = self.S, self.H
            self._tmp = (torch.empty(S * H, 800, dtype=torch.float32, device=self.device),
                         torch.zeros(S, H, 32, 26, dtype=torch.float32, device=self.device),
                         torch.empty(S * 32, (H // 16) * 13 * 32, dtype=torch.float32, device=self.device))
        return self._tmp

    def _fused_images(self):
        """rbnn_conv_weight_images builds the forward and the dense triple image in one launch (RBNN_CONV_FUSED_IMAGES=0: the stand-alone
        builders — permuted copies + rbnn_triple_rows — kept as the reference the fused kernel is tested against)."""
        return os.environ.get("RBNN_CONV_FUSED_IMAGES", "1") != "0" and self.H % 16 == 0

    def _free_staging(self):
        """A stored (HMC / ensemble) posterior builds its images ONCE: the builders' staging buffers — a second copy of the triple rows image,
        three fp32 regroupings, the dense image's two — go back to the allocator (at Hc = 512 they are ~100 MB per 50 samples EACH, i.e. they
        would double or triple the resident size of the posterior).  A redrawable SVI stack (for_guide) keeps them: it rebuilds per draw."""
        if self._guide is None:
            self._tmp = self._dense_tmp = self._dense_stage = self._rows_stage = None

    def _build_dense(self, dense):
        """model.3.weight regrouped [S, K steps of 32 hc, 25 taps, 32 ci][32 hc] (hc zero-padded) as a triple-rows image."""
        S, H = self.S, self.H
        KS = (H + 31) // 32
        if getattr(self, "_dense_tmp", None) is None:
            self._dense_tmp = (torch.zeros(S, KS * 32, 32, 25, dtype=torch.float32, device=self.device),
                               torch.empty(S * KS * 25 * 32, 32, dtype=torch.float32, device=self.device))
        pad, rows = self._dense_tmp
        pad[:, :H].copy_(self.K2w.view(S, H, 32, 25))
        rows.view(S, KS, 25, 32, 32).copy_(pad.view(S, KS, 32, 32, 25).permute(0, 1, 4, 3, 2))         # [s, ks, tap, ci, hc]
        # triple rows ([row][3 pieces][64 B]) into a staging copy, then each 16-row tile piece-major and, inside a piece, FRAGMENT-major:
        # [3 pieces][4 chunks of 8 hc][16 rows][16 B] — lane (row li, chunk lg) of the dense kernel owns bytes 16 (16 lg + li) of every piece,
        # so a piece is one global_load_dwordx4 of 1 KiB contiguous straight into the MFMA's A registers
        if getattr(self, "_dense_stage", None) is None:
            self._dense_stage = torch.empty_like(dense)
        _hip.HipKernels().triple_rows(rows, 32, scale_exp(self._k2_max()), self._dense_stage, 32)
        T = dense.shape[0] // 16
        dense.view(T, 3, 4, 16, 8).copy_(self._dense_stage.view(T, 16, 3, 4, 8).permute(0, 2, 3, 1, 4))

    def _build_triple(self, rows):
        """The forward's image through the stand-alone builders (the fused kernel, rbnn_conv_weight_images, is tested against this)."""
        S, H = self.S, self.H
        k2 = self._scratch()[0]
        k = _hip.HipKernels()
        k2_exp = scale_exp(self._k2_max())
        k2.view(S, H, 25, 32).copy_(self.K2w.view(S, H, 32, 25).permute(0, 1, 3, 2))                 # k = tap*32 + ci
        # triple rows ([channel][tap][3 pieces][64 B]) into a staging copy, then grouped [16 channels][tap][3 pieces][16 rows][64 B]: the order
        # a 16-channel group of one tap has in the kernel's stage tile (its three LDS-DMA pieces then differ by 1 KiB on both sides)
        if getattr(self, "_rows_stage", None) is None:
            self._rows_stage = torch.empty_like(rows)
        k.triple_rows(k2, 800, k2_exp, self._rows_stage, 800)
        G = S * H // 16
        rows.view(G, 25, 3, 16, 32).copy_(self._rows_stage.view(G, 16, 25, 3, 32).permute(0, 2, 3, 1, 4))
        return k2_exp

    def _build_split(self, rows, bwd):
        S, H = self.S, self.H
        k2, w26, kb = self._scratch()
        k = _hip.HipKernels()
        k2_exp = scale_exp(self._k2_max())
        k2.view(S, H, 25, 32).copy_(self.K2w.view(S, H, 32, 25).permute(0, 1, 3, 2))
        k.split_rows(k2, 800, k2_exp, rows, 800)
        w26[..., :25].copy_(self.K2w.view(S, H, 32, 25))
        kb.view(S, 32, H // 16, 13, 2, 2, 8).copy_(w26.view(S, H // 16, 2, 8, 32, 13, 2).permute(0, 4, 1, 5, 6, 2, 3))
        k.split_rows(kb, kb.shape[1], k2_exp, bwd, kb.shape[1])
        return k2_exp

    # ------------------------------------------------------------------ triple-split ("f16x6") mode: full-width operands on the f16 pipe
    def triple_supported(self):
        """The triple conv2 kernels cover both geometries and all four activations, for posteriors whose conv2 weights have an ordinary
        dynamic range (posterior.narrow_range)."""
        if self.device.type != "cuda":
            return False
        if getattr(self, "_range_ok", None) is None:
            from .posterior import narrow_range
            self._range_ok = self._guide.range_ok if self._guide is not None else narrow_range(self.K2w)
        return self._range_ok

    def triple_images(self):
        """(K2 triple-rows image of model.3.weight regrouped tap-major [S*Hc, 25*32] for the forward, its exponent, the dense conv2^T image
        [S, K steps of 32 hc, 25 taps, 32 ci][32 hc], max_f sum_c |Fw[c,f]|) — built once, resident."""
        if self._triple is None:
            S, H = self.S, self.H
            rows = torch.empty(S * H, 800 * 3, dtype=torch.int16, device=self.device)
            self._dense = torch.empty(S * ((H + 31) // 32) * 25 * 32, 32 * 3, dtype=torch.int16, device=self.device)
            if self._fused_images():
                k2_exp = scale_exp(self._k2_max())
                _hip.HipKernels().conv_weight_images(self.K2w, S, H, k2_exp, rows, self._dense)
            else:
                k2_exp = self._build_triple(rows)
                self._build_dense(self._dense)
            self._triple = (rows, k2_exp, self._dense, self._fw_l1())
            self._free_staging()
        return self._triple

    # ------------------------------------------------------------------ split-half precision mode (forward conv2)
    def split_supported(self):
        """The split-half conv kernels are built for 1x28x28 inputs with relu / leaky (the reference's saved conv models)."""
        return self.device.type == "cuda" and self.input_shape == (1, 28, 28) and self.activation in ("relu", "leaky")

    def split_images(self):
        """(K2 split-rows image of model.3.weight regrouped tap-major [S*Hc, 25*32], its exponent, per-unit bound of the pooled
        conv1 activations |P1| <= max_c(sum_taps |K1w_c|) * max|x| + max|K1b|) — built once, resident."""
        if self._split is None:
            S, H = self.S, self.H
            rows = torch.empty(S * H, 800 * 2, dtype=torch.int16, device=self.device)
            bwd = torch.empty(S * 32, (H // 16) * 13 * 32 * 2, dtype=torch.int16, device=self.device)
            k2_exp = self._build_split(rows, bwd)
            w_l1, b_max = self._p1()
            self._split = (rows, k2_exp, w_l1, b_max, bwd, self._fw_l1())
            self._free_staging()
        return self._split

    @classmethod
    def from_state_dicts(cls, sds, activation, input_shape, n_classes, hidden, device):
        keys = [k + sfx for k in CONV_KEYS for sfx in (".weight", ".bias")]
        stacked = {k: torch.stack([sd[k].detach().to("cpu", torch.float32) for sd in sds]) for k in keys}
        return cls(activation, input_shape, n_classes, hidden, stacked, device)

    def scale_bounds(self):
        """(mul, add, cap) of rbnn_input_scales' record 1, the bound of the pooled conv1 ACTIVATIONS that fixes their fp16-piece scale:
        relu / leaky: |act(a)| <= |a| <= max_c(sum_taps |K1w_c|) * max|x| + max|K1b|; tanh: additionally <= 1; sigmoid lies in (0, 1)
        whatever the pre-activation — with tiny conv1 weights or all-zero images the |a| bound would be far BELOW sigmoid(a) ~ 0.5 and the
        scaled activations would overflow fp16 — so its bound is the constant 1, as in StackedPosterior.scale_bounds."""
        if self.activation == "sigm":
            return 0.0, 1.0, 1.0
        w_l1, b_max = self._p1()
        return w_l1, b_max, (1.0 if self.activation == "tanh" else float("inf"))

    def _p1(self):
        if self._guide is not None:
            return self._guide.p1_bound
        if not hasattr(self, "_p1_bound"):
            self._p1_bound = (float(self.K1w.abs().sum(-1).max()), float(self.K1b.abs().max()))
        return self._p1_bound

    def descriptor(self):
        if self._desc is None:
            d = _hip.ConvPosterior()
            d.activation, d.hidden, d.n_classes, d.n_stored = _hip.ACTIVATIONS[self.activation], self.H, self.C, self.S
            d.in_channels, d.in_width = self.Cin, self.W
            for name, t in (("K1w", self.K1w), ("K1b", self.K1b), ("K2w", self.K2w), ("K2b", self.K2b), ("Fw", self.Fw),
                            ("Fb", self.Fb), ("K2w_ci", self.K2ci)):
                setattr(d, name, C.c_void_p(t.data_ptr()))
            self._desc = d
        return self._desc

    def state_dict(self, i):
        sd = {"model.0.weight": self.K1w[i].view(32, self.Cin, 5, 5), "model.0.bias": self.K1b[i],
              "model.3.weight": self.K2w[i].view(self.H, 32, 5, 5), "model.3.bias": self.K2b[i],
              "model.7.weight": self.Fw[i], "model.7.bias": self.Fb[i]}
        return {k: v.detach().cpu().clone() for k, v in sd.items()}


class ConvEngine(AttackEngine):
    """AttackEngine whose forward / gradient calls go to the conv kernels; the attack loops, the reductions over
    samples, the loss kernels and the evaluation are inherited unchanged."""

    graph_safe = False                      # large jobs are cut into point blocks per call: no fixed launch sequence to capture
    fused_tail = False                      # the conv kernels read the fp32 dZ (rbnn_step_tail_triple builds the fc generator image only)
    pipelined_comm = False                  # one cached workspace: the sample-sharded step keeps the plain sequence
    shared_forward = False                  # the backward overwrites the forward's Q2 with dQ2: loss_gradients_and_fgsm runs the two calls

    def workspace(self, N, S, chunk=0, tag=0):
        """One workspace per (N, S); `chunk` / `tag` are the fc engine's knobs (slab plan, pipelined point blocks) and do not
        apply here.  Two entries stay cached — the full point block and a ragged tail block — so that a blocked job does not
        re-allocate its multi-GB buffers on every call."""
        key = (N, S)
        ws = self._ws_cache.get(key)
        if ws is None:
            sizes = self.k.conv_workspace_sizes(self.post, N, S)
            ws = {"n_slabs": S, "chunk": 1}
            for name in _hip.CONV_WS_KEYS:
                if name in _U8:
                    ws[name] = torch.empty(sizes[name], dtype=torch.uint8, device=self.device)
                else:
                    ws[name] = torch.empty(sizes[name] // 4, dtype=torch.float32, device=self.device)
            ws["slabs"] = ws["G"]                                          # per-sample gradients play the role of the slabs
            ws["Psum"] = torch.zeros(N, _hip.CPAD, dtype=torch.float32, device=self.device)
            ws["Gsum"] = torch.empty(N, self.post.D, dtype=torch.float32, device=self.device)
            while len(self._ws_cache) >= 2:
                self._ws_cache.pop(next(iter(self._ws_cache)))             # oldest entry out
            self._ws_cache[key] = ws
        return ws

    # -------------------------------------------------------------- point blocking
    # Activations cost 148 KB per (point, sample) at Hc=512 on 1x28x28 (P1 18 KB, Q2 100 KB, two byte stashes, per-sample
    # gradients), and the mean-probability loss needs every sample's forward before any backward — so large jobs are
    # split over POINTS (independent), never over samples.  Budget: RBNN_CONV_WS_GB (default 48 GB of the 288 GB).
    def point_block(self, S):
        p = self.post
        p1 = 32 * p.P1W * p.P1W
        per_point = S * (max(24576, (p1 + 255) // 256 * 1024) + p1 + p.H * p.NP2 * 5 + p.D * 4 + 2 * 64)
        budget = float(os.environ.get("RBNN_CONV_WS_GB", "48")) * 2 ** 30
        return max(16, int(budget // per_point) // 16 * 16)

    def _blocked(self, fn, x, *rest, n_samples, cat=True):
        nb = self.point_block(n_samples)
        if x.shape[0] <= nb:
            return fn(x, *rest)
        outs = [fn(x[i:i + nb], *[r[i:i + nb] if torch.is_tensor(r) and r.shape[:1] == x.shape[:1] else r for r in rest])
                for i in range(0, x.shape[0], nb)]
        return torch.cat(outs) if cat else outs

    def forward(self, x, n_samples, seeds=None, logits=False):
        if torch.is_grad_enabled() and x.requires_grad:
            return super().forward(x, n_samples, seeds, logits)
        return self._blocked(lambda xb: AttackEngine.forward(self, xb, n_samples, seeds, logits), x, n_samples=n_samples)

    def loss_gradients(self, x, y, n_samples, seeds=None, norms=False):
        y = torch.as_tensor(y)
        parts = self._blocked(lambda xb, yb: AttackEngine.loss_gradients(self, xb, yb, n_samples, seeds, norms), x, y,
                              n_samples=n_samples, cat=not norms)
        if not norms or isinstance(parts, tuple):
            return parts
        return tuple(torch.cat([p[i] for p in parts]) for i in range(3))

    def fgsm(self, x, y, n_samples, epsilon=0.3, seeds=None, mode=_hip.LOSS_MEAN_PROB):
        y = torch.as_tensor(y)
        return self._blocked(lambda xb, yb: AttackEngine.fgsm(self, xb, yb, n_samples, epsilon, seeds, mode), x, y, n_samples=n_samples)

    def pgd(self, x, y, n_samples, epsilon, alpha=None, iters=40, seeds=None, mode=_hip.LOSS_MEAN_PROB, before_step=None):
        """before_step (an SVI net's in-place redraw) runs before every iteration but the first of EVERY point block: blocks of one
        attack then see different draws — each point's marginal over draws is unchanged."""
        y = torch.as_tensor(y)
        return self._blocked(lambda xb, yb: AttackEngine.pgd(self, xb, yb, n_samples, epsilon, alpha, iters, seeds, mode, before_step),
                             x, y, n_samples=n_samples)

    def attack_gradient(self, x, y, n_samples, seeds=None, mode=_hip.LOSS_MEAN_PROB):
        y = torch.as_tensor(y)
        return self._blocked(lambda xb, yb: AttackEngine.attack_gradient(self, xb, yb, n_samples, seeds, mode), x, y, n_samples=n_samples)

    def clean_outputs(self, x, n_samples, logits=False):
        return self._blocked(lambda xb: AttackEngine.clean_outputs(self, xb, n_samples, logits), x, n_samples=n_samples)

    def evaluate(self, x, x_attack, y, n_samples, logits=False, clean=None):
        y = torch.as_tensor(y)
        if clean is not None:
            parts = self._blocked(lambda xb, ab, yb, cb: AttackEngine.evaluate(self, xb, ab, yb, n_samples, logits, cb), x, x_attack, y, clean,
                                  n_samples=n_samples, cat=False)
        else:
            parts = self._blocked(lambda xb, ab, yb: AttackEngine.evaluate(self, xb, ab, yb, n_samples, logits), x, x_attack, y,
                                  n_samples=n_samples, cat=False)
        if isinstance(parts, tuple):
            return parts
        n = x.shape[0]
        sizes = [p[2].shape[0] for p in parts]
        oa = sum(p[0] * m for p, m in zip(parts, sizes)) / n
        aa = sum(p[1] * m for p, m in zip(parts, sizes)) / n
        return oa, aa, torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts]), torch.cat([p[4] for p in parts])

    def _forward_kernels(self, Xp, sidx, S, out_kind, ws):
        if self.precision == "triple":
            rows, k2_exp = self.post.triple_images()[:2]
            ds = self._scales if self._scales is not None else self._input_scales(Xp, iterates=False)
            return self.k.conv_forward_triple(self.post, rows, k2_exp, 0, Xp, sidx, S, out_kind, ws, p1_dev_scale=ds[4:])
        if self.precision != "split":
            return self.k.conv_forward(self.post, Xp, sidx, S, out_kind, ws)
        rows, k2_exp = self.post.split_images()[:2]
        ds = self._scales if self._scales is not None else self._input_scales(Xp, iterates=False)
        self.k.conv_forward_split(self.post, rows, k2_exp, 0, Xp, sidx, S, out_kind, ws, p1_dev_scale=ds[4:])

    def _grad_kernels(self, sidx, S, N, ws, dz_ready=False):
        if self.precision == "triple" and os.environ.get("RBNN_CONV_BWD_EXACT") != "1":
            _, k2_exp, dense, fw_l1 = self.post.triple_images()                 # conv2^T: a GEMM per tap over the conv2 outputs + col2im
            return self.k.conv_input_grad_dense(self.post, dense, k2_exp, fw_l1, sidx, S, N, ws)
        if self.precision != "split" or os.environ.get("RBNN_CONV_BWD_EXACT") == "1":
            self.post.refresh_lazy_images(k2ci=True)
            return self.k.conv_input_grad(self.post, sidx, S, N, ws)
        _, k2_exp, _, _, bwd, fw_l1 = self.post.split_images()
        return self.k.conv_input_grad_split(self.post, bwd, k2_exp, fw_l1, sidx, S, N, ws)
