"""Stacked posterior: S weight samples of one fully-connected net, laid out for the HIP kernels.

The reference keeps an HMC posterior as a dict idx -> NN module (model_bnn.py:186-190) and an
ensemble as a dict seed -> NN (model_ensemble.py:44-55), and walks them in Python.  Here the S
state-dicts are stacked once into contiguous device tensors

    W1 [S, H, D_pad]   b1 [S, H]   (Wm [S, H, H]  bm [S, H])   W2 [S, C, H]   b2 [S, C]

padded to the kernels' tile contract (include/robustbnns_hip.h): D_pad = round_up(D, 16) with zero
columns, hidden 16 -> 32 with zero rows/columns (exact: a padded unit has no outgoing weight).
At MNIST fc-512 one sample is 1.63 MB, so S=2000 is 3.3 GB of the 288 GB HBM3E: the whole posterior
stays resident and every kernel indexes it by sample.
"""
import ctypes as C
import math

import torch

from . import _hip

# state_dict keys per Linear, in network order (model_nn.py:77-91)
LAYER_KEYS = {"fc": ("model.1", "model.3"), "fc2": ("model.1", "model.3", "model.5")}


def round_up(v, m):
    return (v + m - 1) // m * m


def padded_hidden(H):
    return max(32, H)


def narrow_range(*tensors, ratio=4096.0):
    """True when every tensor's largest magnitude is within `ratio` (2^12) of its mean magnitude (the mean, unlike the rms, is not
    carried by a single outlier).  The triple images carry a value exactly while |v| >= 2^-15 * max|tensor| and to 2^-39 * max|tensor|
    absolutely below that: with max <= 2^12 * mean|v| the absolute error of ANY element is <= 2^-27 of a typical element, i.e. below
    fp32's own resolution of it.  A posterior with a wilder dynamic range (one huge outlier weight) is left to the fp32 MFMA kernels,
    whose per-element precision does not depend on the rest of the tensor."""
    for t in tensors:
        if t is None:
            continue
        m = float(t.abs().max())
        r = float(t.abs().double().mean())
        if not (m <= ratio * r) or m == 0.0 or m != m or m == float("inf"):
            return False
    return True


def scale_exp(max_abs, target=14):
    """e with max_abs * 2^e <= 2^target: the power-of-two scale of a split-half image (rbnn_split_rows & co)."""
    if not (max_abs > 0.0) or math.isinf(max_abs):
        return 0
    return max(-100, min(100, target - math.ceil(math.log2(max_abs))))


class SviGuide:
    """The variational parameters of an fc / fc2 guide (model_bnn.py:124-126: `<key>_loc`, `<key>_scale`, raw scale — softplus is
    applied at the draw) resident on the device, plus everything about them that is fixed ONCE per guide, so that a redraw
    (rbnn_svi_draw) needs no pass over the drawn weights and no device->host sync:

      bound[k] = max(|loc| + RBNN_SVI_EPS_MAX * softplus(scale))   of W1, W2, Wm: the images' power-of-two scales
      h1       = (max_h sum_d bound(W1[h,d]), max bound(b1))       the fc2 hidden-activation bound (StackedPosterior.scale_bounds)
      range_ok = narrow_range on the bound against mean|loc| + 0.8 softplus(scale)  (the guard of the triple mode)

    Box-Muller on a 32-bit uniform cannot exceed 6.764 standard deviations, so these are bounds, not estimates."""

    def __init__(self, loc, scale, arch, device):
        if arch not in LAYER_KEYS:
            raise NotImplementedError(f"architecture {arch!r}: the in-place SVI draw covers fc and fc2")
        self.arch, self.device = arch, torch.device(device)
        keys = LAYER_KEYS[arch]
        f = lambda d, k: d[k].detach().to(self.device, torch.float32).contiguous()
        names = {"W1": keys[0] + ".weight", "b1": keys[0] + ".bias", "W2": keys[-1] + ".weight", "b2": keys[-1] + ".bias"}
        if arch == "fc2":
            names.update(Wm=keys[1] + ".weight", bm=keys[1] + ".bias")
        self.loc = {n: f(loc, k) for n, k in names.items()}
        self.scale = {n: f(scale, k) for n, k in names.items()}
        self.hidden = int(self.loc["b1"].numel())
        self.loc["W1"] = self.loc["W1"].reshape(self.hidden, -1)
        self.scale["W1"] = self.scale["W1"].reshape(self.hidden, -1)
        # the standard deviation softplus(raw scale) (model_bnn.py:18,127) is the same for every draw: taken ONCE here (the draw kernel was
        # ALU-bound while it re-evaluated expf + log1pf for each of the S x P weights)
        self.sigma = {n: torch.nn.functional.softplus(v).contiguous() for n, v in self.scale.items()}
        bnd = {n: self.loc[n].abs() + _hip.SVI_EPS_MAX * self.sigma[n] for n in self.loc}
        typ = {n: self.loc[n].abs() + 0.8 * self.sigma[n] for n in self.loc}                         # ~ E|w|
        mats = [n for n in ("W1", "W2", "Wm") if n in bnd]
        rec = torch.stack([bnd[n].max() for n in mats] + [typ[n].double().mean().float() for n in mats] +
                          [bnd["W1"].sum(-1).max(), bnd["b1"].max()]).cpu().tolist()                   # the one sync, at load
        k = len(mats)
        self.bound = dict(zip(mats, rec[:k]))
        self.h1_bound = (rec[2 * k], rec[2 * k + 1])
        self.range_ok = all(0.0 < m <= 4096.0 * t and m != float("inf") for m, t in zip(rec[:k], rec[k:2 * k]))
        self._desc = None

    def descriptor(self):
        if self._desc is None:
            d = _hip.SviGuide()
            for n in ("W1", "b1", "Wm", "bm", "W2", "b2"):
                setattr(d, n + "_loc", None if n not in self.loc else C.c_void_p(self.loc[n].data_ptr()))
                setattr(d, n + "_scale", None if n not in self.sigma else C.c_void_p(self.sigma[n].data_ptr()))
            d.hidden = self.hidden
            self._desc = d
        return self._desc


class StackedPosterior:
    def __init__(self, arch, activation, input_shape, n_classes, hidden, stacked, device):
        """`stacked`: dict state_dict-key -> tensor [S, ...] (unpadded, any device, fp32)."""
        if arch not in LAYER_KEYS:
            raise NotImplementedError(f"architecture {arch!r}: the HIP path covers fc and fc2 (conv is SURVEY 8f 'next')")
        if activation not in _hip.ACTIVATIONS:
            raise AssertionError("\nWrong activation name.")                      # model_nn.py:74-75
        if n_classes > _hip.CPAD:
            raise ValueError(f"n_classes={n_classes} > {_hip.CPAD}")
        self.arch, self.activation = arch, activation
        self.input_shape = tuple(int(v) for v in input_shape)
        self.D = int(torch.tensor(self.input_shape).prod())
        self.Dp = round_up(self.D, 16)
        self.H, self.Hp, self.C = int(hidden), padded_hidden(int(hidden)), int(n_classes)
        self.device = torch.device(device)
        keys = LAYER_KEYS[arch]
        S = stacked[keys[0] + ".weight"].shape[0]
        self.S = int(S)

        def pad(t, shape):
            out = torch.zeros((S,) + shape, dtype=torch.float32, device=self.device)
            out[(slice(None),) + tuple(slice(0, d) for d in t.shape[1:])] = t.to(self.device, torch.float32)
            return out.contiguous()

        w = lambda k: stacked[k]
        self.W1 = pad(w(keys[0] + ".weight").reshape(S, self.H, self.D), (self.Hp, self.Dp))
        self.b1 = pad(w(keys[0] + ".bias"), (self.Hp,))
        if arch == "fc2":
            self.Wm = pad(w(keys[1] + ".weight"), (self.Hp, self.Hp))
            self.bm = pad(w(keys[1] + ".bias"), (self.Hp,))
        else:
            self.Wm = self.bm = None
        self.W2 = pad(w(keys[-1] + ".weight"), (self.C, self.Hp))
        self.b2 = pad(w(keys[-1] + ".bias"), (self.C,))
        self._pack()
        self._desc = None
        self._split = None
        self._triple = None
        self._range_ok = None
        self._guide = None                      # SviGuide: this posterior is a redrawable SVI stack (for_guide / redraw)
        self._back = None                       # second buffer set of enable_prefetch()

    def _abs_max(self, name):
        """max |tensor| that fixes an image's power-of-two scale: taken from the stored weights (one sync, at load), or, for a
        redrawable SVI stack, the guide's a-priori bound (no pass over the drawn weights)."""
        if self._guide is not None:
            return self._guide.bound[name]
        return float(getattr(self, name).abs().max())

    def _h1(self):
        if self._guide is not None:
            return self._guide.h1_bound
        if not hasattr(self, "_h1_bound"):
            self._h1_bound = (float(self.W1.abs().sum(-1).max()), float(self.b1.abs().max()))
        return self._h1_bound

    # ------------------------------------------------------------------ split-half ("f16x3") precision mode
    def split_supported(self):
        """The split kernels cover fc and fc2 (all four activations) with hidden % 128 == 0 and <= 10 classes."""
        return self.arch in ("fc", "fc2") and self.Hp % 128 == 0 and self.C <= 10 and self.device.type == "cuda"

    def split_images(self):
        """rbnn_split_images of this posterior (built once, resident): W1 as split rows (forward A operand), W1 as
        split cols (backward B operand), W2 as the dA-generator image.  Same footprint as W1 each."""
        if self._split is None:
            k = _hip.HipKernels()
            S, H, Dp, D, Cn = self.S, self.Hp, self.Dp, self.D, self.C
            ld = round_up(D, 32)
            w1_exp = scale_exp(self._abs_max("W1"))
            w2_exp = scale_exp(self._abs_max("W2"))
            rows = torch.empty(S * H, ld * 2, dtype=torch.int16, device=self.device)
            cols = torch.empty(S * (H // 32) * 8 * Dp * 8, dtype=torch.int16, device=self.device)
            gen = torch.empty(S * (H // 16) * 512, dtype=torch.int16, device=self.device)
            k.split_rows(self.W1, D, w1_exp, rows, ld)
            k.split_cols(self.W1, H, D, w1_exp, cols, Dp)
            k.split_w2gen(self.W2, Cn, H, w2_exp, gen)
            img = _hip.SplitImages()
            img.W1_rows, img.W1_cols, img.W2_gen = rows.data_ptr(), cols.data_ptr(), gen.data_ptr()
            img.ld_rows, img.ld_cols, img.w1_exp, img.w2_exp = ld, Dp, w1_exp, w2_exp
            keep = [rows, cols, gen]
            if self.arch == "fc2":                               # forward of the middle layer: Wm as split rows [S*H, H]
                wm_exp = scale_exp(self._abs_max("Wm"))
                wm_rows = torch.empty(S * H, H * 2, dtype=torch.int16, device=self.device)
                k.split_rows(self.Wm, H, wm_exp, wm_rows, H)
                wm_cols = torch.empty(S * (H // 32) * 8 * H * 8, dtype=torch.int16, device=self.device)
                k.split_cols(self.Wm, H, H, wm_exp, wm_cols, H)
                img.Wm_rows, img.Wm_cols, img.wm_exp = wm_rows.data_ptr(), wm_cols.data_ptr(), wm_exp
                keep += [wm_rows, wm_cols]
            self._split = (img, keep)                            # the tensors keep the device memory alive
        return self._split[0]

    # ------------------------------------------------------------------ triple-split ("f16x6") mode: full-width operands
    def triple_supported(self):
        """The triple kernels cover fc and fc2 (all four activations) with hidden % 128 == 0 and <= 10 classes — and posteriors whose
        weight tensors have an ordinary dynamic range (narrow_range): anything else stays on the fp32 MFMA."""
        if not (self.arch in ("fc", "fc2") and self.Hp % 128 == 0 and self.C <= 10 and self.device.type == "cuda"):
            return False
        if self._range_ok is None:
            self._range_ok = self._guide.range_ok if self._guide is not None else narrow_range(self.W1, self.W2, self.Wm)
        return self._range_ok

    def triple_images(self):
        """rbnn_triple_images of this posterior (built once, resident): W1 as triple rows (forward A operand, 6 B per weight),
        W1 as triple cols (backward B operand), W2 as the dA-generator image."""
        if self._triple is None:
            k = _hip.HipKernels()
            S, H, Dp, D, Cn = self.S, self.Hp, self.Dp, self.D, self.C
            ld = round_up(D, 32)
            w1_exp = scale_exp(self._abs_max("W1"))
            w2_exp = scale_exp(self._abs_max("W2"))
            rows = torch.empty(S * H, ld * 3, dtype=torch.int16, device=self.device)
            cols = torch.empty(S * (H // 32) * 12 * Dp * 8, dtype=torch.int16, device=self.device)
            gen = torch.empty(S * (H // 16) * 1024, dtype=torch.int16, device=self.device)
            k.triple_rows(self.W1, D, w1_exp, rows, ld, grouped=True)
            k.triple_cols(self.W1, H, D, w1_exp, cols, Dp)
            k.triple_w2gen(self.W2, Cn, H, w2_exp, gen)
            img = _hip.TripleImages()
            img.W1_rows, img.W1_cols, img.W2_gen = rows.data_ptr(), cols.data_ptr(), gen.data_ptr()
            img.ld_rows, img.ld_cols, img.w1_exp, img.w2_exp = ld, Dp, w1_exp, w2_exp
            keep = [rows, cols, gen]
            if self.arch == "fc2":                               # the middle layer: Wm as triple rows [S*H, H] (forward) and triple cols (backward step 1)
                wm_exp = scale_exp(self._abs_max("Wm"))
                wm_rows = torch.empty(S * H, H * 3, dtype=torch.int16, device=self.device)
                k.triple_rows(self.Wm, H, wm_exp, wm_rows, H, grouped=True)
                wm_cols = torch.empty(S * (H // 32) * 12 * H * 8, dtype=torch.int16, device=self.device)
                k.triple_cols(self.Wm, H, H, wm_exp, wm_cols, H)
                img.Wm_rows, img.Wm_cols, img.wm_exp = wm_rows.data_ptr(), wm_cols.data_ptr(), wm_exp
                keep += [wm_rows, wm_cols]
            self._triple = (img, keep)                           # the tensors keep the device memory alive
        return self._triple[0]

    def scale_bounds(self):
        """(mul, add, cap) of rbnn_input_scales' record 1 — the bound of the fc2 hidden-activation image given max|x|:
        |h1| <= min(cap, mul * max|x| + add) with mul = max_h sum_d |W1[h,d]|, add = max|b1| (relu / leaky: |act(a)| <= |a|;
        |tanh(a)| <= min(|a|, 1); sigmoid <= 1).  fc has no such operand: the record is unused."""
        if self.arch != "fc2":
            return 0.0, 0.0, math.inf
        w_l1, b_max = self._h1()
        if self.activation == "sigm":
            return 0.0, 1.0, 1.0
        return w_l1, b_max, (1.0 if self.activation == "tanh" else math.inf)

    def _pack(self):
        """rbnn_pack_rows4 images [S, H/4, cols, 4] of W1 (and Wm): the backward GEMM's B-operand layout.  Built by the HIP
        kernel; a posterior that is not on the GPU has no images (nothing could read them: there is no CPU compute path)."""
        def pack(W):
            if W is None or W.device.type != "cuda":
                return None
            out = torch.empty_like(W)
            _hip.HipKernels().pack_rows4(W, out)
            return out
        self.W1p, self.Wmp = pack(self.W1), pack(self.Wm)

    # ------------------------------------------------------------------ redrawable SVI stack
    @classmethod
    def for_guide(cls, guide, activation, input_shape, n_classes, S):
        """S (still zero) samples laid out for the kernels, to be filled — again and again, IN PLACE — by redraw().  Buffers,
        descriptor and images keep their addresses across draws, so the engine and its workspaces are reused."""
        H, D = guide.hidden, guide.loc["W1"].shape[1]
        z = lambda *shape: torch.zeros((S,) + shape, dtype=torch.float32)
        keys = LAYER_KEYS[guide.arch]
        stacked = {keys[0] + ".weight": z(H, D), keys[0] + ".bias": z(H), keys[-1] + ".weight": z(n_classes, H), keys[-1] + ".bias": z(n_classes)}
        if guide.arch == "fc2":
            stacked[keys[1] + ".weight"], stacked[keys[1] + ".bias"] = z(H, H), z(H)
        post = cls(guide.arch, activation, input_shape, n_classes, H, stacked, guide.device)
        post._guide = guide
        return post

    def lazy_capable(self):
        """A draw can be left (partly) PENDING.  Lowdim kind — fc nets with in_features <= 16 and no weight images: nothing is launched, the draw
        is recorded as (key, draw id) and generated inside the next rbnn_lowdim_run launch (BASELINE config 1: a redraw + an FGSM pass at the launch
        floor of two kernels becomes one).  Triple kind — the triple images exist: see below.  Either way materialize() completes it."""
        if self._guide is None or self._split is not None or self._back is not None or self.device.type != "cuda":
            return False
        if self._triple is not None:            # the triple kind: the images ARE drawn now (rbnn_svi_draw_images), only the fp32 W1 / Wm stack and
            return True                         # its pack_rows4 copy — 40 % of a full draw's writes, read by no triple kernel — are left pending
        return self.arch == "fc" and self.D <= 16 and self.C <= 10

    def materialize(self):
        """Run a pending (lazy) draw for real: rbnn_svi_draw with the recorded (key, draw id) — the same weights the fused launches generated.
        Called by everything that reads the stack (the tensor attributes, descriptor(), state_dict(), shard())."""
        lazy = self.__dict__.get("_lazy")
        if lazy is not None:
            self.__dict__["_lazy"] = None
            self.redraw(lazy[0], lazy[1], lazy[2], lazy[3])
        return self

    def redraw(self, key, draw_id=0, n_samples=None, sample_keys=None, lazy=False):
        """W[s] = loc + softplus(scale) * eps(key, draw_id, s) for s < n_samples, written by ONE kernel (rbnn_svi_draw) into the fp32
        stack, the pack_rows4 images and — once they exist — the triple images.  sample_keys: int64 device tensor, one key per
        sample (seeded draws, model_bnn.py:222-226).  No device->host sync, no allocation.  lazy=True (callers about to run the lowdim
        kernels): only RECORD the draw where lazy_capable() — see materialize()."""
        if self._guide is None:
            raise _hip.HipError("redraw() needs a posterior built by StackedPosterior.for_guide")
        S = self.S if n_samples is None else int(n_samples)
        self.__dict__["_lazy"] = None           # a newer draw supersedes a pending one
        if lazy and self.lazy_capable():
            if self._triple is not None:
                _hip.HipKernels().svi_draw(self, self._triple[0], self._guide, S, int(key), int(draw_id), sample_keys, images_only=True)
            # (the pending record owns a COPY of the keys: an in-place edit of the caller's tensor before materialize() must not change the draw)
            self.__dict__["_lazy"] = (int(key), int(draw_id), S, None if sample_keys is None else sample_keys.clone())
            return self
        tri = self._triple[0] if self._triple is not None else None
        _hip.HipKernels().svi_draw(self, tri, self._guide, S, int(key), int(draw_id), sample_keys)
        if self._split is not None:             # the opt-in two-piece mode keeps its own images: rebuilt by its builders (same fixed scales)
            img, keep = self._split
            k = _hip.HipKernels()
            k.split_rows(self.W1, self.D, img.w1_exp, keep[0], img.ld_rows)
            k.split_cols(self.W1, self.Hp, self.D, img.w1_exp, keep[1], self.Dp)
            k.split_w2gen(self.W2, self.C, self.Hp, img.w2_exp, keep[2])
            if self.arch == "fc2":
                k.split_rows(self.Wm, self.Hp, img.wm_exp, keep[3], self.Hp)
                k.split_cols(self.Wm, self.Hp, self.Hp, img.wm_exp, keep[4], self.Hp)
        return self

    # ------------------------------------------------------------------ redraw overlapped with compute (second buffer set + side stream)
    _SETS = ("W1", "b1", "Wm", "bm", "W2", "b2", "W1p", "Wmp")

    def can_prefetch(self):
        """The next draw can be prepared while the kernels still read the current one: needs the resident SVI stack of for_guide and
        not the opt-in split mode (whose images are rebuilt by separate builders from the fp32 stack)."""
        return self._guide is not None and self._split is None and self.device.type == "cuda"

    def enable_prefetch(self):
        """A second ("back") set of every buffer rbnn_svi_draw writes — fp32 stack, pack_rows4 images, triple images — plus a side stream
        and two events.  prefetch(key) draws into the back set on the side stream, ordered after everything enqueued so far on the current
        stream (the last readers of those buffers); flip() makes the current stream wait for that draw and swaps the sets.  An SVI PGD
        attack redraws before every iteration (model_bnn.py:230-232): with this the draw of iteration t + 1 — HBM-write-bound — runs
        under the compute-bound GEMM kernels of iteration t instead of between them."""
        if getattr(self, "_back", None) is not None:
            return
        back = {n: (None if getattr(self, n) is None else torch.empty_like(getattr(self, n))) for n in self._SETS}
        self._back = {"bufs": back, "triple": None, "desc": None}
        self._side = torch.cuda.Stream(device=self.device)
        self._ev_main, self._ev_drawn = torch.cuda.Event(), torch.cuda.Event()
        self._prefetched = False

    def _back_triple(self):
        """The back set's triple images, allocated when the front set has them (they may be built after enable_prefetch)."""
        if self._triple is not None and self._back["triple"] is None:
            img, keep = self._triple
            keep2 = [torch.empty_like(t) for t in keep]
            img2 = _hip.TripleImages()
            C.memmove(C.byref(img2), C.byref(img), C.sizeof(img))
            img2.W1_rows, img2.W1_cols, img2.W2_gen = keep2[0].data_ptr(), keep2[1].data_ptr(), keep2[2].data_ptr()
            if self.arch == "fc2":
                img2.Wm_rows, img2.Wm_cols = keep2[3].data_ptr(), keep2[4].data_ptr()
            self._back["triple"] = (img2, keep2)

    def _swap_sets(self):
        b = self._back
        for n in self._SETS:
            cur = getattr(self, n)
            setattr(self, n, b["bufs"][n])
            b["bufs"][n] = cur
        self._triple, b["triple"] = b["triple"], self._triple
        self._desc, b["desc"] = b["desc"], self._desc

    def prefetch(self, key, draw_id=0, sample_keys=None):
        """Start the draw of the NEXT weight set into the back buffers on the side stream; returns at once."""
        self.enable_prefetch()
        self._back_triple()
        self._ev_main.record(torch.cuda.current_stream(self.device))
        self._swap_sets()                       # the draw kernel writes through `self`: aim it at the back set ...
        try:
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._ev_main)
                self.redraw(key, draw_id, sample_keys=sample_keys)
                self._ev_drawn.record(self._side)
        finally:
            self._swap_sets()                   # ... and leave the front set in place for the kernels about to be launched
        self._prefetched = True

    def flip(self):
        """Make the prefetched weight set current: the current stream waits for its draw, then every later launch reads it."""
        if not getattr(self, "_prefetched", False):
            raise _hip.HipError("flip() without a prefetch()")
        torch.cuda.current_stream(self.device).wait_event(self._ev_drawn)
        self._swap_sets()
        self._prefetched = False

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_state_dicts(cls, state_dicts, arch, activation, input_shape, n_classes, hidden, device):
        keys = [k + sfx for k in LAYER_KEYS[arch] for sfx in (".weight", ".bias")]
        stacked = {k: torch.stack([sd[k].detach().to("cpu", torch.float32) for sd in state_dicts]) for k in keys}
        return cls(arch, activation, input_shape, n_classes, hidden, stacked, device)

    @classmethod
    def from_modules(cls, nets, device):
        n0 = nets[0]
        return cls.from_state_dicts([n.state_dict() for n in nets], n0.architecture, n0.activation,
                                    n0.input_shape, n0.output_size, n0.hidden_size, device)

    # ------------------------------------------------------------------ C-ABI view
    def descriptor(self, lazy_ok=False):
        """lazy_ok: the caller does not read the weights behind the pointers (size queries) or generates them itself (the fused lowdim launch)."""
        if not lazy_ok:
            self.materialize()
        if self._desc is None:
            d = _hip.Posterior()
            d.arch, d.activation = _hip.ARCHS[self.arch], _hip.ACTIVATIONS[self.activation]
            d.in_features, d.in_stride, d.hidden, d.n_classes, d.n_stored = self.D, self.Dp, self.Hp, self.C, self.S
            raw = lambda n: self.__dict__.get("_t_" + n)      # (not through the attributes: they would materialise a pending draw)
            for name in ("W1", "b1", "Wm", "bm", "W2", "b2"):
                t = raw(name)
                setattr(d, name, None if t is None else C.c_void_p(t.data_ptr()))
            d.W1_pack4 = None if raw("W1p") is None else C.c_void_p(raw("W1p").data_ptr())
            d.Wm_pack4 = None if raw("Wmp") is None else C.c_void_p(raw("Wmp").data_ptr())
            self._desc = d
        return self._desc

    def state_dict(self, i):
        """Sample i as an unpadded CPU state_dict with the reference's keys (model_nn.py:77-91)."""
        keys = LAYER_KEYS[self.arch]
        H, D = self.H, self.D
        sd = {keys[0] + ".weight": self.W1[i, :H, :D], keys[0] + ".bias": self.b1[i, :H]}
        if self.arch == "fc2":
            sd[keys[1] + ".weight"], sd[keys[1] + ".bias"] = self.Wm[i, :H, :H], self.bm[i, :H]
        sd[keys[-1] + ".weight"], sd[keys[-1] + ".bias"] = self.W2[i, :, :H], self.b2[i]      # network order, as NN.state_dict() lists them
        return {k: v.detach().cpu().clone() for k, v in sd.items()}

    def nbytes(self):
        return sum(t.numel() * 4 for t in (self.W1, self.b1, self.Wm, self.bm, self.W2, self.b2, self.W1p, self.Wmp) if t is not None)

    def shard(self, rank, world):
        """Samples [rank*S/world, (rank+1)*S/world) as a new posterior (sample-sharded multi-GPU)."""
        self.materialize()
        lo, hi = rank * self.S // world, (rank + 1) * self.S // world
        out = object.__new__(StackedPosterior)
        out.__dict__.update(self.__dict__)
        for name in ("W1", "b1", "Wm", "bm", "W2", "b2"):
            t = getattr(self, name)
            setattr(out, name, None if t is None else t[lo:hi].contiguous())
        # the dynamic-range guard is decided ONCE, on the full posterior: every rank (and the single-process run) picks the same mode
        out.S, out._desc, out._split, out._triple, out._range_ok = hi - lo, None, None, None, self.triple_supported() and self._range_ok
        out._guide, out._back = None, None
        out._pack()
        return out


def _stack_tensor(name):
    """The stacked weight tensors as attributes that first materialise a pending lazy draw (StackedPosterior.materialize): whoever READS the stack
    sees the weights of the last redraw(), whether it ran as its own launch or only inside the fused lowdim launches so far."""
    def get(self):
        if self.__dict__.get("_lazy") is not None:
            self.materialize()
        return self.__dict__.get("_t_" + name)

    def put(self, value):
        self.__dict__["_t_" + name] = value
    return property(get, put)


for _n in ("W1", "b1", "Wm", "bm", "W2", "b2", "W1p", "Wmp"):
    setattr(StackedPosterior, _n, _stack_tensor(_n))
