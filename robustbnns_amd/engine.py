"""Batched driver of the hot path: every test point x every posterior sample per launch.

Reference loop nest (adversarialAttacks.py:118 -> :95 -> model_bnn.py:251, batch 1 at the leaves) vs here:

    per PGD iteration (1 for FGSM / loss_gradients), for ALL N points at once:
      rbnn_fc_forward      P[s,n,:] = softmax(NN_s(x_n))  + activation-derivative stash      (MFMA GEMM)
      rbnn_reduce_samples  Psum[n,:] = sum_s P[s,n,:]         -> all-reduce when samples are sharded
      rbnn_loss_dlogits    dZ[s,n,:] for the loss in use (mean-prob / per-sample / mean-logit)
      rbnn_fc_input_grad   slabs[k,n,:] = sum_{s in chunk k} dA_s[n,:] . W1_s                 (MFMA GEMM)
      rbnn_attack_step     x <- clamp(x0 + clamp(x + a*sign(sum_k slabs) - x0, +-eps), 0, 1)
                           (sharded: rbnn_sum_slabs -> all-reduce -> rbnn_attack_step)

x, x0, the posterior and every intermediate stay resident in HBM for the whole attack.

Precision modes of the two GEMMs (`precision=` / RBNN_PRECISION): "exact" = fp32 on the fp32 MFMA (rbnn_fc_forward /
rbnn_fc_input_grad); "triple" = the same full-width fp32 operands carried as three fp16 pieces, six exact product terms on the
f16 MFMA pipe, fp32 accumulation (rbnn_fc_forward_triple / rbnn_fc_input_grad_triple: nothing narrower than fp32, ~1.7x faster);
"auto" (default) = triple where those kernels cover the posterior, else exact; "split" (opt-in) = error-compensated half
pairs on the f16 MFMA pipe (rbnn_fc_forward_split / rbnn_fc_input_grad_split: 2^-22 per product, ~2.7x faster, same 1e-5
parity bar, operands narrower than fp32); "fast" = split where the split kernels cover the posterior, else exact.
Low-dimensional fc nets (in_features <= 16: half-moons) run NONE of the above: "auto" resolves to "lowdim", where a forward, an expected
gradient or a whole T-iteration attack is ONE launch of rbnn_lowdim_run (rbnn_lowdim.hip; fp32 FMA, the iterate resident in registers).
torch supplies device memory, the current HIP stream and torch.distributed (RCCL); all arithmetic is
in the HIP kernels behind `kernels` (robustbnns_amd._hip.HipKernels — there is no other backend in
this package; tests inject a CPU fake to exercise the multi-process orchestration under gloo).
"""
import os

import torch

from . import _hip
from ._hip import (LOSS_MEAN_LOGIT, LOSS_MEAN_PROB, LOSS_PER_SAMPLE, LOSS_UPSTREAM, LOSS_UPSTREAM_LOGIT, OUT_LOGITS, OUT_PROBS)

_WS_DTYPE = {"mask1": torch.int32, "mask2": torch.int32}


def to_labels(y, device):
    """one-hot float [N,C] (utils.py:90-91,110) or integer labels [N] -> int32 [N] on `device`."""
    y = torch.as_tensor(y)
    if y.dim() >= 2:
        y = y.argmax(-1)                       # lossGradients.py:23, adversarialAttacks.py:120
    return y.to(device=device, dtype=torch.int32).contiguous()


class _DifferentiableForward(torch.autograd.Function):
    """BNN/NN forward as an autograd node: output = mean over samples, backward = the hand-rolled HIP input gradient."""

    @staticmethod
    def forward(ctx, x, engine, sidx, S, logits):
        ctx.engine, ctx.sidx, ctx.S, ctx.logits = engine, sidx, S, logits
        ctx.save_for_backward(x.detach())
        out = engine.forward_padded(engine.pad_inputs(x), sidx, S, OUT_LOGITS if logits else OUT_PROBS)
        return out[:, :engine.post.C].clone()

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        g = ctx.engine.vjp(x, grad_out.contiguous(), ctx.sidx, ctx.S, ctx.logits)
        return g.to(x.device), None, None, None, None


class CommStats:
    """What the collectives of a run did, as the LAUNCH stream saw it (attached by bench.py: AttackEngine.comm_stats; None = nothing is recorded).
    `exposed`: event pairs on the launch stream around every point where that stream waits for an exchange — a blocking all-reduce, or the
    .wait() of an asynchronous one.  Nothing else is enqueued on the stream between the two events, so their distance is the time the compute
    stream stood still for the collective: the part of the exchange NOT hidden under kernels."""

    def __init__(self, event_factory):
        self.event = event_factory
        self.on = False
        self.calls = 0
        self.bytes = 0
        self.exposed = []

    def count(self, t):
        if self.on:
            self.calls += 1
            self.bytes += t.numel() * t.element_size()

    def around(self, fn):
        if not self.on:
            return fn()
        e0, e1 = self.event(), self.event()
        e0.record()
        r = fn()
        e1.record()
        self.exposed.append((e0, e1))
        return r


class AttackEngine:
    comm_stats = None                       # a CommStats while bench.py measures a sharded run

    def __init__(self, posterior, kernels=None, group=None, total_samples=None, precision=None):
        """posterior: StackedPosterior holding THIS rank's samples.  group: a torch.distributed process
        group when the posterior is sample-sharded across ranks; total_samples: samples over
        all ranks (default: all-reduced once)."""
        self.post = posterior
        self.k = kernels if kernels is not None else _hip.HipKernels()
        self.group = group
        self.world = 1
        if group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            if os.environ.get("RBNN_FORCE_COLLECTIVES") == "1":
                self.world = max(self.world, 2)    # diagnostics: run the all-reduce path even in a 1-rank group
        self._S_total = total_samples
        # diagnostics only (tools/collectives_ab.sh): the sharded launch sequence with NO exchange.  In a group of more than one rank that would
        # silently return every rank's partial sums as the result, so it is refused there.  An engine WITHOUT a group never exchanges anything
        # (bench's point-sharded and other-mode engines, SVI hot-path engines): the pair of switches leaves it alone
        self._fake_comm = (group is not None and os.environ.get("RBNN_FAKE_COLLECTIVES") == "1"
                           and os.environ.get("RBNN_FORCE_COLLECTIVES") == "1")
        if self._fake_comm:
            import torch.distributed as dist
            if dist.get_world_size(group) != 1:
                raise _hip.HipError("RBNN_FAKE_COLLECTIVES=1 skips every all-reduce: it is a timing diagnostic for a 1-rank group "
                                    "(with RBNN_FORCE_COLLECTIVES=1), never valid with more than one rank")
        self._ws_cache = {}
        self.precision = self._resolve_precision(precision)
        self._scales = None                     # device-resident operand scales of an attack's iterates (split mode), set by the attack loops
        self._carry_image = False               # inside pgd(): attack steps also write the next iterate's triple image

    def _resolve_precision(self, precision):
        """exact: both GEMMs in fp32 on the fp32 MFMA.  auto (the default): triple where it applies (below), else exact.
        split: error-compensated fp16 pairs on the f16 MFMA pipe (operands ~22-23 bits, i.e. narrower than fp32; ~2.7x faster;
        same 1e-5 parity bar, and the same adversarial accuracy in the split-vs-exact tests) — OPT-IN, raises where the split
        kernels do not cover the posterior.  fast: split where it applies, else exact.  RBNN_PRECISION sets the default."""
        want = (precision or os.environ.get("RBNN_PRECISION") or "auto").lower()
        if want not in ("auto", "exact", "triple", "split", "fast", "lowdim"):
            raise ValueError(f"precision={want!r}: expected 'auto', 'exact', 'triple', 'split', 'fast' or 'lowdim'")
        # lowdim: in_features <= 16 (half-moons).  The contractions are then a few FMAs per hidden unit and a pass is bound by the launch
        # floor of the 7 kernels above: one kernel does the forward / the expected gradient / ALL iterations of an attack in fp32 FMA
        # arithmetic (rbnn_lowdim.hip).  Single process only (the sample-sharded step needs its all-reduces between the phases); calls it
        # does not cover (autograd hooks with an upstream gradient) run the fp32-MFMA kernels.
        low = (isinstance(self.k, _hip.HipKernels) and self.world == 1 and getattr(self.post, "arch", None) in ("fc", "fc2")
               and self.device.type == "cuda" and self.k.lowdim_supported(self.post))
        if want == "lowdim" and not low:
            raise _hip.HipError("precision='lowdim' covers fc posteriors (fc2: hidden 32 ... 512, a power of two) with in_features <= 16 and "
                                "classes <= 10 on one GPU")
        if want == "lowdim" or (want == "auto" and low):
            return "lowdim"
        # triple: full-width fp32 operands as three fp16 pieces, six exact product terms on the f16 matrix pipe, fp32 accumulation
        # (rbnn_triple.hip).  Nothing is narrower than fp32 — the only rounding left is the fp32 accumulation, as on the fp32 MFMA; its
        # measured error against fp64 is BELOW the fp32-MFMA kernels' (tests/test_hip_triple.py) — and it is ~1.7x faster, so "auto"
        # takes it wherever it applies and falls back to the fp32 MFMA ("exact") elsewhere.
        tri = bool(getattr(self.post, "triple_supported", lambda: False)()) and isinstance(self.k, _hip.HipKernels)
        if want == "triple" and not tri:
            raise _hip.HipError("precision='triple' covers fc / fc2 posteriors with hidden % 128 == 0 and classes <= 10 and the conv "
                                "architecture, on the GPU, for weight tensors of ordinary dynamic range (posterior.narrow_range)")
        if want == "triple" or (want == "auto" and tri):
            return "triple"
        ok = bool(getattr(self.post, "split_supported", lambda: False)()) and isinstance(self.k, _hip.HipKernels)
        if want == "split" and not ok:
            raise _hip.HipError("precision='split' covers fc / fc2 posteriors with hidden % 128 == 0 and classes <= 10, and the conv "
                                "architecture, on the GPU")
        return "split" if (want in ("split", "fast") and ok) else "exact"

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return self.post.device

    class _NoWork:
        def wait(self):
            return True

    def _allreduce(self, t):
        if self.world > 1 and not self._fake_comm:
            import torch.distributed as dist
            cs = self.comm_stats
            if cs is None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            else:
                cs.count(t)
                cs.around(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))

    def _allreduce_async(self, t):
        """Start the all-reduce (RCCL runs it on its own stream) and return the work handle; _wait(handle) orders the current
        stream after it.  Kernels launched in between overlap the exchange."""
        if self._fake_comm:                      # diagnostics (RBNN_FAKE_COLLECTIVES=1 with RBNN_FORCE_COLLECTIVES=1): the sharded launch sequence
            return self._NoWork()                # without any collective — separates what the sequence costs from what RCCL costs
        import torch.distributed as dist
        if self.comm_stats is not None:
            self.comm_stats.count(t)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _wait(self, h):
        """The launch stream waits for an asynchronous all-reduce; with comm_stats attached the wait sits between two events (its exposed part)."""
        cs = self.comm_stats
        return h.wait() if cs is None else cs.around(h.wait)

    def total_samples(self, S_local):
        """Samples over all ranks taking part in this call."""
        if self.world == 1:
            return S_local
        if self._S_total is not None and S_local == self.post.S:
            return self._S_total
        t = torch.tensor([float(S_local)], device=self.device)
        self._allreduce(t)
        return int(round(t.item()))

    def pad_inputs(self, x, clone=False):
        """[N, *input_shape] -> contiguous fp32 [N, D_pad] on the device (zero columns beyond D)."""
        p = self.post
        xf = x.detach().reshape(x.shape[0], -1).to(self.device, torch.float32)
        if xf.shape[1] != p.D:
            raise ValueError(f"inputs flatten to {xf.shape[1]} features, posterior expects {p.D}")
        if p.Dp == p.D:
            return xf.clone() if clone else xf.contiguous()
        out = torch.zeros(x.shape[0], p.Dp, dtype=torch.float32, device=self.device)
        out[:, :p.D] = xf
        return out

    def unpad(self, Xp, like):
        return Xp[:, :self.post.D].reshape(like.shape).clone()

    def _flat(self, x):
        """[N, *input_shape] -> contiguous fp32 [N, D] on the device, UNPADDED and without a copy when x already is that: what the
        lowdim kernel reads (any row stride >= D; the MFMA kernels need the D_pad image of pad_inputs)."""
        xf = x.detach().reshape(x.shape[0], -1).to(self.device, torch.float32).contiguous()
        if xf.shape[1] != self.post.D:
            raise ValueError(f"inputs flatten to {xf.shape[1]} features, posterior expects {self.post.D}")
        return xf

    def sample_index(self, n_samples, seeds=None):
        """model_bnn.py:200-202,246-252: first n_samples stored samples, or `seeds` as indices."""
        if seeds:
            if len(seeds) != n_samples:
                raise ValueError("Number of seeds should match number of samples.")
        idx = list(range(n_samples)) if seeds is None else [int(s) for s in seeds]
        for i in idx:
            if i >= self.post.S or i < -self.post.S:
                raise IndexError("list index out of range")     # posterior_predictive[seed], model_bnn.py:252
        if idx == list(range(self.post.S)):
            return None, len(idx)
        if idx == list(range(len(idx))):
            return None, len(idx)                               # prefix: identity map, no index buffer
        t = torch.tensor([i % self.post.S for i in idx], dtype=torch.int32, device=self.device)
        t._rbnn_max_index = max(i % self.post.S for i in idx)          # known on the host: lets a pending lazy draw check its coverage without a sync
        return t, len(idx)

    def workspace(self, N, S, chunk=0, tag=0):
        key = (N, S, chunk, tag)
        ws = self._ws_cache.pop(key, None)
        if ws is not None:
            self._ws_cache[key] = ws            # re-inserted: the dict's order is the order of last use
        if ws is None:
            sizes = self.k.workspace_sizes(self.post, N, S, chunk)
            if self.precision == "triple":
                gb = self._fc2_groups(N, S, sizes["chunk"])[1]
                if gb < S:                                          # fc2, grouped backward: dL/d(pre-activation 1) of gb samples at a time
                    sizes["dhid1"] = sizes["dhid1"] // S * gb
            ws = {"n_slabs": sizes["n_slabs"], "chunk": sizes["chunk"]}
            for name in _hip.WS_KEYS:
                if sizes[name]:
                    ws[name] = torch.empty(sizes[name] // 4, dtype=_WS_DTYPE.get(name, torch.float32), device=self.device)
            ws["Psum"] = torch.zeros(N, _hip.CPAD, dtype=torch.float32, device=self.device)
            ws["G"] = torch.empty(N, self.post.Dp, dtype=torch.float32, device=self.device)
            if self.precision == "split":
                ssz = self.k.split_workspace_sizes(self.post, self.post.split_images(), N, S)
                ws["split"] = {"X_split": torch.empty(ssz["X_split"] // 2, dtype=torch.int16, device=self.device),
                               "dZ_gen": torch.empty(ssz["dZ_gen"] // 2, dtype=torch.int16, device=self.device),
                               "g_scale": torch.empty(ssz["g_scale"] // 4, dtype=torch.float32, device=self.device)}
            if self.precision == "triple":
                gf, gb = self._fc2_groups(N, S, ws["chunk"])
                tsz = self.k.triple_workspace_sizes(self.post, self.post.triple_images(), N, S)
                if gf < S:                                          # fc2, grouped: ONE hidden image of gf samples, reused group after group
                    tsz["hid_triple"] = tsz["hid_triple"] // S * gf
                # X_triple / hid_triple are ZEROED once: the forward kernel DMA-reads whole 16-row groups, and rows N .. ceil16(N) - 1 are never
                # written by the image builders — their products land in accumulator columns that are never stored, but they must not be
                # stale NaN / Inf halves of another call's data (invariant stated in include/robustbnns_hip.h)
                ws["triple"] = {k: (torch.zeros if k in ("X_triple", "hid_triple") else torch.empty)(max(1, v // 2), dtype=torch.int16, device=self.device)
                                for k, v in tsz.items() if v}
                ws["triple"]["g_scale"] = ws["triple"]["g_scale"].view(torch.float32)
                ws.pop("hid1", None)                                # the hidden activations live in the triple image instead
                ws["fc2_groups"] = (gf, gb)
            while len(self._ws_cache) > 6:      # evict the LEAST RECENTLY USED entry only: a sharded step holds at most RBNN_COMM_BLOCKS (<= 6) live
                self._ws_cache.pop(next(iter(self._ws_cache)))    # workspaces with pending all-reduce handles, all younger than it
            self._ws_cache[key] = ws
        return ws

    # ------------------------------------------------------------------ kernel hooks (overridden for the conv architecture)
    def _forward_kernels(self, Xp, sidx, S, out_kind, ws):
        if self.precision == "triple":
            img = self.post.triple_images()
            ds = self._scales if self._scales is not None else self._input_scales(Xp, iterates=False)
            # inside a PGD loop the previous attack_step_triple has already written the iterate's image — of THAT tensor: the record names the
            # buffer the image was built for, and any forward on this workspace consumes it (a forward on other inputs overwrites the image)
            if ws.pop("x_image_ready", None) != Xp.data_ptr():
                self.k.triple_rows(Xp, self.post.D, 0, ws["triple"]["X_triple"], img.ld_rows, dev_scale=ds, grouped=True)
            gf = ws.get("fc2_groups", (S, S))[0]
            if gf >= S:
                return self.k.fc_forward_triple(self.post, img, ws["triple"], 0, Xp.shape[0], sidx, S, out_kind, ws, dev_scales=ds)
            for s0 in range(0, S, gf):          # fc2: layer 1 -> layer 2 per group of samples through ONE reused hidden image (_fc2_groups)
                g = min(gf, S - s0)
                self.k.fc_forward_triple(self.post, img, ws["triple"], 0, Xp.shape[0], self._sample_slice(sidx, s0, g), g, out_kind,
                                         self._ws_samples(ws, s0, Xp.shape[0]), dev_scales=ds)
            return None
        if self.precision != "split":
            return self.k.fc_forward(self.post, Xp, sidx, S, out_kind, ws)
        img = self.post.split_images()
        ds = self._scales if self._scales is not None else self._input_scales(Xp, iterates=False)
        self.k.split_rows(Xp, self.post.D, 0, ws["split"]["X_split"], img.ld_rows, dev_scale=ds)
        self.k.fc_forward_split(self.post, img, ws["split"]["X_split"], img.ld_rows, 0, Xp.shape[0], sidx, S, out_kind, ws, dev_scales=ds)

    def _grad_kernels(self, sidx, S, N, ws, dz_ready=False):
        if self.precision == "triple":
            gb = ws.get("fc2_groups", (S, S))[1]
            if dz_ready:                        # the generator image of ALL samples was built by step_tail_triple: one call (its per-point scale spans them)
                return self.k.fc_input_grad_triple(self.post, self.post.triple_images(), sidx, S, N, ws["chunk"], ws, ws["triple"], dz_ready=True)
            if gb >= S:
                return self.k.fc_input_grad_triple(self.post, self.post.triple_images(), sidx, S, N, ws["chunk"], ws, ws["triple"])
            n_slabs = 0                         # fc2: step 1 -> step 2 per group of samples (a multiple of the slab chunk) through ONE reused dhid1
            for s0 in range(0, S, gb):
                g = min(gb, S - s0)
                n_slabs += self.k.fc_input_grad_triple(self.post, self.post.triple_images(), self._sample_slice(sidx, s0, g), g, N, ws["chunk"],
                                                       self._ws_samples(ws, s0, N, n_slabs), ws["triple"])
            return n_slabs
        if self.precision != "split" or (self.post.arch == "fc2" and os.environ.get("RBNN_FC2_BWD_EXACT") == "1"):
            return self.k.fc_input_grad(self.post, sidx, S, N, ws["chunk"], ws)
        return self.k.fc_input_grad_split(self.post, self.post.split_images(), sidx, S, N, ws["chunk"], ws, ws["split"])

    def _fc2_groups(self, N, S, chunk):
        """(forward, backward) sample-group sizes of the triple-mode fc2 path.  Layer 1 writes the hidden activations of every (sample,
        point) as an fp32 image for layer 2 to read back and split at its operand read (2 GB at C2's size, S x N x H x 4 B; rounds 2-4: the three pieces, 6 B), and the backward does the same with dhid1
        (fp32, 2 GB).  Run group after group through ONE buffer of g samples, the image a layer reads is the one the layer before it has
        just written — inside the 256 MB Infinity Cache when g x N x H x 4 B fits — and the workspace shrinks by (S - g) / S.
        RBNN_FC2_GROUP (both) / RBNN_FC2_GROUP_FWD / RBNN_FC2_GROUP_BWD override (0 = one group of S); the backward's group is rounded
        to a multiple of the slab chunk (a slab sums the samples of one chunk)."""
        if getattr(self.post, "arch", None) != "fc2":
            return S, S
        both = os.environ.get("RBNN_FC2_GROUP")
        want_f = os.environ.get("RBNN_FC2_GROUP_FWD", both)
        want_b = os.environ.get("RBNN_FC2_GROUP_BWD", both)
        per_sample = (N + 15) // 16 * 16 * self.post.Hp * 4
        auto = max(1, int(self._FC2_GROUP_BYTES // per_sample)) if self._FC2_GROUP_BYTES else S
        gf = int(want_f) if want_f not in (None, "") else auto
        gb = int(want_b) if want_b not in (None, "") else auto
        gf = S if gf <= 0 else min(S, gf)
        gb = S if gb <= 0 else min(S, max(chunk, gb // chunk * chunk))
        return gf, gb

    _FC2_GROUP_BYTES = 0                    # default group = this many bytes of hidden image (0: no grouping); set from the A/B in profiles/r04a

    def _sample_slice(self, sidx, s0, g):
        """The index buffer of samples [s0, s0 + g) of a call (the kernels take weights AND workspace rows by it / by position)."""
        if sidx is not None:
            return sidx[s0:s0 + g]
        if getattr(self, "_arange", None) is None or self._arange.numel() < s0 + g:
            self._arange = torch.arange(max(self.post.S, s0 + g), dtype=torch.int32, device=self.device)
        return self._arange[s0:s0 + g]

    def _ws_samples(self, ws, s0, N, slab0=None):
        """The workspace as samples [s0, ...) see it: every per-sample buffer advanced by s0 samples (the kernels index them from 0);
        hid_triple / dhid1 are group-local (not advanced); slabs advanced by the slabs already written."""
        H = self.post.Hp
        n_pad = (N + 255) // 256 * 256
        stride = {"P": N * _hip.CPAD, "dZ": N * _hip.CPAD, "mask1": (H // 32) * n_pad, "mask2": (H // 32) * n_pad, "dact1": N * H, "dact2": N * H}
        out = dict(ws)
        for k, st in stride.items():
            if k in ws:
                out[k] = ws[k][s0 * st:]
        if slab0:
            out["slabs"] = ws["slabs"][slab0 * N * self.post.Dp:]
        return out

    def _input_scales(self, X, iterates):
        """Split mode: the power-of-two scales of the inputs' half-pair image (and of the activations they bound: fc2 hidden
        layer / conv1 output), computed ON THE DEVICE by rbnn_input_scales — two 16-byte records the kernels read, no
        device->host sync.  `iterates`: later PGD iterates are clamp(., 0, 1) of something (adversarialAttacks.py:105), so
        |x| <= max(|x0|, 1) bounds them all and one record serves the whole attack; FGSM differentiates at x0 itself."""
        if self.precision not in ("split", "triple"):
            return None
        mul, add, cap = self.post.scale_bounds()
        out = torch.empty(8, dtype=torch.int32, device=self.device)
        return self.k.input_scales(X, self.post.D, 1.0 if iterates else 0.0, mul, add, cap, out)

    # ------------------------------------------------------------------ forward
    def forward_padded(self, Xp, sidx, S, out_kind=OUT_PROBS, out=None):
        """mean over samples of P (probabilities or logits) -> [N, 16] buffer (columns >= C are zero)."""
        N = Xp.shape[0]
        if out is None:
            out = torch.zeros(N, _hip.CPAD, dtype=torch.float32, device=self.device)
        if self.precision == "lowdim":
            self.k.lowdim_run(self.post, self.k.LOWDIM_FORWARD, 0, out_kind, Xp, None, sidx, S, None, 1.0, 1.0 / S, 0.0, None, 0.0, False, False,
                              1, self._low_scratch(N, S) if self.post.arch == "fc2" else None, out)
            return out
        ws = self.workspace(N, S)
        self._forward_kernels(Xp, sidx, S, out_kind, ws)
        if self.world == 1:
            self.k.reduce_samples(ws["P"], S, N, self.post.C, 1.0 / S, out)         # model_bnn.py:257
        else:       # sample-sharded: every rank scales its partial sum by 1 / (samples over all ranks); the all-reduce finishes the mean
            self.k.reduce_samples(ws["P"], S, N, self.post.C, 1.0 / self.total_samples(S), out)
            self._allreduce(out)
        return out

    def forward(self, x, n_samples, seeds=None, logits=False):
        sidx, S = self.sample_index(n_samples, seeds)
        if torch.is_grad_enabled() and x.requires_grad:
            # callers that differentiate the forward themselves (the reference's own fgsm_attack does:
            # adversarialAttacks.py:73-79): the backward is the HIP input-gradient path with the upstream dL/dout
            return _DifferentiableForward.apply(x, self, sidx, S, bool(logits))
        xin = self._flat(x) if self.precision == "lowdim" else self.pad_inputs(x)
        out = self.forward_padded(xin, sidx, S, OUT_LOGITS if logits else OUT_PROBS)
        return out[:, :self.post.C]

    def vjp(self, x, grad_out, sidx, S, logits):
        """d<grad_out, forward(x)>/dx for forward = mean over samples of probabilities (or logits)."""
        N, C = x.shape[0], self.post.C
        Xp = self.pad_inputs(x)
        ws = self.workspace(N, S)
        S_tot = self.total_samples(S)
        self._forward_kernels(Xp, sidx, S, OUT_LOGITS if logits else OUT_PROBS, ws)
        gup = torch.zeros(N, _hip.CPAD, dtype=torch.float32, device=self.device)
        gup[:, :C] = grad_out.to(self.device, torch.float32)
        # probabilities: dZ_s = softmax backward of grad/S through p_s; logits (d mean_s z_s): dZ_s = grad/S for every sample
        self.k.loss_dlogits(LOSS_UPSTREAM_LOGIT if logits else LOSS_UPSTREAM, ws["P"], None, gup, None, S, 1.0 / S_tot, N, C, ws["dZ"])
        n_slabs = self._grad_kernels(sidx, S, N, ws)
        G = ws["Gsum"] if "Gsum" in ws else ws["G"]
        self.k.sum_slabs(ws["slabs"], n_slabs, N, self.post.Dp, 1.0, G)
        self._allreduce(G)
        return self.unpad(G, x)

    # ------------------------------------------------------------------ expected input gradient
    def gradient_slabs(self, Xp, labels, sidx, S, mode, G_up=None, chunk=0):
        """Runs forward + loss + backward; leaves n_slabs partial gradients [N, D_pad] in ws['slabs']."""
        N = Xp.shape[0]
        ws = self.workspace(N, S, chunk)
        S_tot = self.total_samples(S)
        self._forward_kernels(Xp, sidx, S, OUT_LOGITS if mode == LOSS_MEAN_LOGIT else OUT_PROBS, ws)
        return ws, self._loss_backward(ws, labels, sidx, S, N, mode, S_tot, G_up), S_tot

    def _loss_backward(self, ws, labels, sidx, S, N, mode, S_tot, G_up=None):
        """From a finished forward — P[s, n, :] and the activation-derivative stash in `ws` — the loss in use -> dL/dlogits -> the backward
        GEMM; returns the slab count.  Reads the forward's state without changing it (fc / fc2): two losses can be differentiated from ONE
        forward (loss_gradients_and_fgsm)."""
        C = self.post.C
        # per-sample losses are averaged at the very end (lossGradients.py:40), the others inside the loss
        inv_S = 1.0 if mode == LOSS_PER_SAMPLE else 1.0 / S_tot
        if self._fused_tail(mode, G_up) and ws.get("fc2_groups", (S, S))[1] >= S:
            # one launch: sum over samples + loss + the dZ generator image (rbnn_step_tail_triple), bit-identical to the three it replaces;
            # the fp32 dZ buffer is never written
            self.k.step_tail_triple(mode, ws["P"], labels, S, inv_S, N, C, ws["triple"])
            return self._grad_kernels(sidx, S, N, ws, dz_ready=True)
        Psum = None
        if mode in (LOSS_MEAN_PROB, LOSS_MEAN_LOGIT):
            Psum = ws["Psum"]
            self.k.reduce_samples(ws["P"], S, N, C, 1.0, Psum)
            self._allreduce(Psum)                               # 64 B per point: the only exchange before the backward
        self.k.loss_dlogits(mode, ws["P"], Psum, G_up, labels, S, inv_S, N, C, ws["dZ"])
        return self._grad_kernels(sidx, S, N, ws)

    fused_tail = True                       # ConvEngine: its kernels read the fp32 dZ

    def _fused_tail(self, mode, G_up):
        """The step's tail between the two GEMM kernels as ONE launch: triple mode, one GPU (the sharded step all-reduces between the sum
        over samples and the loss), a label loss.  RBNN_FUSED_TAIL=0 keeps the three separate kernels (the A/B of tests and profiles)."""
        return (self.fused_tail and self.precision == "triple" and self.world == 1 and G_up is None and isinstance(self.k, _hip.HipKernels)
                and mode in (LOSS_MEAN_PROB, LOSS_PER_SAMPLE, LOSS_MEAN_LOGIT) and self.post.C <= 10
                and os.environ.get("RBNN_FUSED_TAIL", "1") != "0")

    def gradient(self, Xp, labels, sidx, S, mode, G_up=None, norms=None):
        """Summed (and, sample-sharded, all-reduced) expected input gradient [N, D_pad].  norms = (linf [N], l2 [N]) device
        buffers: filled with the per-point norms of that gradient in the same pass as the slab sum (rbnn_sum_slabs_norms)."""
        if self.precision == "lowdim" and G_up is None and mode in (LOSS_MEAN_PROB, LOSS_PER_SAMPLE, LOSS_MEAN_LOGIT):
            per = mode == LOSS_PER_SAMPLE
            G = self._low_out(Xp.shape[0])
            self.k.lowdim_run(self.post, self.k.LOWDIM_GRADIENT, mode, 0, Xp, None, sidx, S, labels, 1.0 if per else 1.0 / S, 1.0 / S if per else 1.0,
                              0.0, None, 0.0, False, False, 1, self._low_scratch(Xp.shape[0], S), G, None if norms is None else norms[0],
                              None if norms is None else norms[1])
            return G
        ws, n_slabs, S_tot = self.gradient_slabs(Xp, labels, sidx, S, mode, G_up)
        scale = 1.0 / S_tot if mode == LOSS_PER_SAMPLE else 1.0
        N, p = Xp.shape[0], self.post
        G = ws["Gsum"] if "Gsum" in ws else ws["G"]
        if norms is not None and self.world == 1:
            self.k.sum_slabs_norms(ws["slabs"], n_slabs, N, p.Dp, p.D, scale, G, norms[0], norms[1])
            return G
        self.k.sum_slabs(ws["slabs"], n_slabs, N, p.Dp, scale, G)
        self._allreduce(G)                                      # N x D_pad fp32: the one large exchange
        if norms is not None:                                   # sharded: the norms are those of the all-reduced gradient
            out = torch.empty_like(G)
            self.k.sum_slabs_norms(G, 1, N, p.Dp, p.D, 1.0, out, norms[0], norms[1])
            return out
        return G

    def loss_gradients(self, x, y, n_samples, seeds=None, norms=False):
        """lossGradients.loss_gradient for every row of x (lossGradients.py:20-40) -> x's shape; norms=True: also the per-point
        Linf and L2 norms of those gradients ([N] each, device), fused into the gradient pass (lossGradients.py:91-105)."""
        sidx, S = self.sample_index(n_samples, seeds)
        nb = None
        if norms:
            nb = (torch.empty(x.shape[0], dtype=torch.float32, device=self.device),
                  torch.empty(x.shape[0], dtype=torch.float32, device=self.device))
        xin = self._flat(x) if self.precision == "lowdim" else self.pad_inputs(x)
        G = self.gradient(xin, to_labels(y, self.device), sidx, S, LOSS_PER_SAMPLE, norms=nb)
        return (self.unpad(G, x), nb[0], nb[1]) if norms else self.unpad(G, x)

    # ------------------------------------------------------------------ attacks
    def _step(self, X, X0, labels, sidx, S, mode, alpha, alpha_scalar, eps, project):
        p = self.post
        if self.world > 1 and self.pipelined_comm:
            return self._step_sharded(X, X0, labels, sidx, S, mode, alpha, alpha_scalar, eps, project)
        ws, n_slabs, _ = self.gradient_slabs(X, labels, sidx, S, mode)
        if (self.world == 1 and self._carry_image and self._fused_tail(mode, None) and self._scales is not None and "triple" in ws
                and p.D % 4 == 0):
            # PGD: the step also writes the NEW iterate's triple image (rbnn_attack_step_triple) — the next forward launches no builder
            self.k.attack_step_triple(X, X0, ws["slabs"], n_slabs, X.shape[0] * p.Dp, p.Dp, alpha, alpha_scalar, eps, project, p.D,
                                      self._scales, ws["triple"]["X_triple"], self.post.triple_images().ld_rows)
            ws["x_image_ready"] = X.data_ptr()
        elif self.world == 1:
            self.k.attack_step(X, X0, ws["slabs"], n_slabs, X.shape[0] * p.Dp, p.Dp, alpha, alpha_scalar, eps, project, p.D)
        else:
            G = ws["Gsum"] if "Gsum" in ws else ws["G"]
            self.k.sum_slabs(ws["slabs"], n_slabs, X.shape[0], p.Dp, 1.0, G)
            self._allreduce(G)
            self.k.attack_step(X, X0, G, 1, 0, p.Dp, alpha, alpha_scalar, eps, project, p.D)

    pipelined_comm = True                   # ConvEngine (one cached workspace) keeps the plain sequence

    def _comm_blocks(self, N, S=None):
        """Point blocks of the sample-sharded step.  Pipelining hides at most the exchange of all blocks but the last under compute, and
        costs 2-7 % by itself (half-size launches: measured on one rank with the collectives forced, profiles/r03t, r03u).  The exchange
        is 8 D_pad bytes per point against 4 S (D H + H C) flops per point of compute: about 2200 / (S H) of the step at xGMI / MI355X
        rates — 4 % at C2 (S H = 51 200), where one block (the exchange exposed) is cheaper than two; below S H = 16 384 (exchange > 13 %)
        the step is cut in two.  RBNN_COMM_BLOCKS overrides."""
        env = os.environ.get("RBNN_COMM_BLOCKS")
        if env is not None:
            want = int(env)
        else:
            want = 2 if (S is not None and S * getattr(self.post, "Hp", 1 << 30) < 16384) else 1
        return max(1, min(want, N // max(1, int(os.environ.get("RBNN_COMM_MIN_POINTS", "512")))))

    def _step_sharded(self, X, X0, labels, sidx, S, mode, alpha, alpha_scalar, eps, project):
        """One attack step with the posterior sample-sharded over the ranks, pipelined over point blocks: the
        points are independent, so block b's two exchanges (sum_s p_s [n,16] before the loss, the summed gradients
        [n,D_pad] before the step) run on RCCL's stream while block b+1's forward / backward kernels run on ours.
        Every rank does the same blocks in the same order; each block has its own workspace."""
        p, N, C = self.post, X.shape[0], self.post.C
        nb = self._comm_blocks(N, S)
        # block boundaries on multiples of 256 points (the gradient kernels' point tile; the forward's is 128): the blocks together then
        # launch exactly the tiles of the unsplit step — an even split of 10 000 points would add a nearly empty tile to every kernel
        bounds = [0] + [min(N, (N * i // nb + 255) // 256 * 256) for i in range(1, nb)] + [N]
        bounds = sorted(set(bounds))
        nb = len(bounds) - 1
        S_tot = self.total_samples(S)
        need_psum = mode in (LOSS_MEAN_PROB, LOSS_MEAN_LOGIT)
        inv_S = 1.0 if mode == LOSS_PER_SAMPLE else 1.0 / S_tot
        blocks = []
        for b in range(nb):                                     # forward + local sum over samples; start the small exchange
            lo, hi = bounds[b], bounds[b + 1]
            ws = self.workspace(hi - lo, S, 0, tag=b)
            self._forward_kernels(X[lo:hi], sidx, S, OUT_LOGITS if mode == LOSS_MEAN_LOGIT else OUT_PROBS, ws)
            h = None
            if need_psum:
                self.k.reduce_samples(ws["P"], S, hi - lo, C, 1.0, ws["Psum"])
                h = self._allreduce_async(ws["Psum"])
            blocks.append((lo, hi, ws, h))
        pending = []
        for lo, hi, ws, h in blocks:                            # loss + backward; start the large exchange
            if h is not None:
                self._wait(h)
            self.k.loss_dlogits(mode, ws["P"], ws["Psum"] if need_psum else None, None, labels[lo:hi], S, inv_S, hi - lo, C, ws["dZ"])
            n_slabs = self._grad_kernels(sidx, S, hi - lo, ws)
            self.k.sum_slabs(ws["slabs"], n_slabs, hi - lo, p.Dp, 1.0, ws["G"])
            pending.append(self._allreduce_async(ws["G"]))
        for (lo, hi, ws, _), h in zip(blocks, pending):         # identical sign / project / clamp on every rank's replica of x
            self._wait(h)
            self.k.attack_step(X[lo:hi], None if X0 is None else X0[lo:hi], ws["G"], 1, 0, p.Dp,
                               None if alpha is None else alpha[lo:hi], alpha_scalar, eps, project, p.D)

    def fgsm(self, x, y, n_samples, epsilon=0.3, seeds=None, mode=LOSS_MEAN_PROB):
        """adversarialAttacks.fgsm_attack on every row of x (adversarialAttacks.py:69-83)."""
        sidx, S = self.sample_index(n_samples, seeds)
        if self.precision == "lowdim":          # one launch; the inputs are read where they are, the result is a fresh tensor
            out = self._lowdim_attack(self._flat(x), None, to_labels(y, self.device), sidx, S, mode, None, 0.0, False, float(epsilon), False, 1)
            return out.reshape(x.shape)
        X = self.pad_inputs(x, clone=True)
        self._scales = self._input_scales(X, iterates=False)
        try:
            self._step(X, None, to_labels(y, self.device), sidx, S, mode, None, float(epsilon), 0.0, False)
        finally:
            self._scales = None
        return self.unpad(X, x)

    def attack_gradient(self, x, y, n_samples, seeds=None, mode=LOSS_MEAN_PROB):
        """The summed (sample-sharded: all-reduced) input gradient whose SIGN fgsm takes, [N, D_pad] on the device.  epsilon enters an
        FGSM attack only after the sign (adversarialAttacks.py:81-82): one gradient serves every epsilon of a grid (fgsm_from_gradient)."""
        sidx, S = self.sample_index(n_samples, seeds)
        xin = self._flat(x) if self.precision == "lowdim" else self.pad_inputs(x)
        return self.gradient(xin, to_labels(y, self.device), sidx, S, mode).clone()

    def fgsm_from_gradient(self, x, G, epsilon):
        """clamp(x + epsilon * sign(G), 0, 1) for every row of x (adversarialAttacks.py:81-82; rbnn_attack_step's operation order) from a
        gradient of attack_gradient(): no forward, no backward."""
        p = self.post
        X = self.pad_inputs(x, clone=True)
        self.k.attack_step(X, None, G, 1, 0, p.Dp, None, float(epsilon), 0.0, False, p.D)
        return self.unpad(X, x)

    shared_forward = True                   # ConvEngine: its backward overwrites the forward's activations in place (Q2 -> dQ2)

    def loss_gradients_and_fgsm(self, x, y, n_samples, epsilon=0.3, seeds=None, mode=LOSS_MEAN_PROB):
        """loss_gradients(x, y) AND fgsm(x, y) on the same inputs and samples — BASELINE config 4's step ("expected_loss_gradients + FGSM") —
        from ONE forward.  The two differ only in the loss (lossGradients.py:33-34: CE of every sample's own prediction, averaged at the
        end :40; adversarialAttacks.py:74-76: CE of the mean prediction): P[s, n, :] and the activation-derivative stash are the same, so
        the forward GEMM runs once and the tail + backward GEMM twice — three GEMMs instead of four.  Every kernel call, its arguments and
        its order of operations are those of the two separate calls: both results are BIT-IDENTICAL to them (tests/test_hip_round5.py).
        Returns (expected loss gradients, adversarial inputs), both of x's shape.  Sample-sharded: the three exchanges of the separate
        calls (summed per-sample-loss gradients — overlapped with the second backward; sum_s p_s; summed mean-loss gradients)."""
        if not self.shared_forward or self.precision == "lowdim" or mode != LOSS_MEAN_PROB:
            # (an ensemble's / a deterministic net's attack differentiates the mean LOGITS: its forward leaves logits, the per-sample loss needs
            # probabilities — nothing to share; the lowdim kernels recompute the forward inside their one launch)
            return self.loss_gradients(x, y, n_samples, seeds), self.fgsm(x, y, n_samples, epsilon, seeds, mode)
        sidx, S = self.sample_index(n_samples, seeds)
        labels = to_labels(y, self.device)
        p = self.post
        X = self.pad_inputs(x, clone=True)
        N = X.shape[0]
        self._scales = self._input_scales(X, iterates=False)
        try:
            ws = self.workspace(N, S)
            S_tot = self.total_samples(S)
            self._forward_kernels(X, sidx, S, OUT_PROBS, ws)
            G = ws["Gsum"] if "Gsum" in ws else ws["G"]
            n_slabs = self._loss_backward(ws, labels, sidx, S, N, LOSS_PER_SAMPLE, S_tot)
            if self.world == 1:
                self.k.sum_slabs(ws["slabs"], n_slabs, N, p.Dp, 1.0 / S_tot, G)
                expected = self.unpad(G, x)
                n_slabs = self._loss_backward(ws, labels, sidx, S, N, mode, S_tot)
                self.k.attack_step(X, None, ws["slabs"], n_slabs, N * p.Dp, p.Dp, None, float(epsilon), 0.0, False, p.D)
            else:
                # sample-sharded: the exchange of the per-sample-loss gradients (N x D_pad fp32, its own buffer) is STARTED here and runs on the
                # collective's stream under the second tail + backward GEMM; the mean-loss pass then has its two exchanges as in a plain step
                G1 = ws.get("G_expected")
                if G1 is None:
                    G1 = ws["G_expected"] = torch.empty_like(G)
                self.k.sum_slabs(ws["slabs"], n_slabs, N, p.Dp, 1.0 / S_tot, G1)
                pending = self._allreduce_async(G1)
                n_slabs = self._loss_backward(ws, labels, sidx, S, N, mode, S_tot)
                self.k.sum_slabs(ws["slabs"], n_slabs, N, p.Dp, 1.0, G)
                self._allreduce(G)
                self.k.attack_step(X, None, G, 1, 0, p.Dp, None, float(epsilon), 0.0, False, p.D)
                self._wait(pending)
                expected = self.unpad(G1, x)
        finally:
            self._scales = None
        return expected, self.unpad(X, x)

    def pgd(self, x, y, n_samples, epsilon, alpha=None, iters=40, seeds=None, mode=LOSS_MEAN_PROB, before_step=None):
        """adversarialAttacks.pgd_attack on every row of x (adversarialAttacks.py:86-108).
        alpha=None: 2/max(image) per image (:89); a float: the same step for all (2/225, :91).
        before_step: called before every iteration but the first (an SVI net redraws its resident weights in place there: the
        reference draws fresh weights at every forward, model_bnn.py:230-232; the caller drew the first set)."""
        sidx, S = self.sample_index(n_samples, seeds)
        labels = to_labels(y, self.device)
        if self.precision == "lowdim":          # all `iters` iterations in ONE launch (an SVI net redraws in between: one launch per iteration)
            a = 0.0 if alpha is None else float(alpha)
            X0 = self._flat(x)
            if before_step is None:
                return self._lowdim_attack(X0, X0, labels, sidx, S, mode, None, a, alpha is None, float(epsilon), True, iters).reshape(x.shape)
            X = X0
            for it in range(iters):
                if it:
                    before_step()
                X = self._lowdim_attack(X, X0, labels, sidx, S, mode, None, a, alpha is None, float(epsilon), True, 1)
            return X.reshape(x.shape)
        X0 = self.pad_inputs(x, clone=True)
        X = X0.clone()
        alpha_t = None
        if alpha is None:
            alpha_t = torch.empty(X.shape[0], dtype=torch.float32, device=self.device)
            self.k.pgd_alpha(X0, self.post.D, alpha_t)
        self._scales = self._input_scales(X0, iterates=True)
        step = lambda: self._step(X, X0, labels, sidx, S, mode, alpha_t, 0.0 if alpha is None else float(alpha), float(epsilon), True)
        self._carry_image = self.graph_safe     # (ConvEngine cuts a job into point blocks with one workspace: no image to carry)
        try:
            done = 0
            if before_step is not None:
                for it in range(iters):
                    if it:
                        before_step()
                    step()
                done = iters
            elif iters >= 3 and self._graph_capturable():
                # RBNN_HIPGRAPH=1 (opt-in).  One iteration is a fixed sequence of 7-9 launches on fixed buffers (x is updated in
                # place): run it once eagerly (allocates the workspace), capture it once in a HIP graph, replay it for the other
                # iterations.  Bit-identical to the eager loop (tests); measured gain on MI355X: none — the asynchronous
                # launches already run ahead of the GPU (half-moons 40.2 -> 39.1 us per iteration, MNIST N=1000 560 -> 574 us),
                # the small cases are bound by the ~5 us floor of each of the 7 kernels, not by launch overhead.
                step()
                done = 1
                graph = self._capture(step)
                if graph is not None:
                    for _ in range(iters - done):
                        graph.replay()
                    done = iters
            for _ in range(iters - done):
                step()
        finally:
            self._scales = None
            self._carry_image = False
            for w in self._ws_cache.values():   # the last step's image belongs to no later forward
                if isinstance(w, dict):
                    w.pop("x_image_ready", None)
        return self.unpad(X, x)

    def _low_out(self, N):
        """[N, D_pad] gradient buffer of the lowdim path, cached per N (the generic path keeps its own in the workspace)."""
        key = ("lowG", N)
        t = self._ws_cache.pop(key, None)
        if t is None:
            t = torch.empty(N, self.post.Dp, dtype=torch.float32, device=self.device)
        self._ws_cache[key] = t
        return t

    def _low_scratch(self, N, S):
        """rbnn_lowdim_run's scratch (per-sample outputs; fc2: + per-sample gradient slabs + the sum over samples), cached per (N, S)."""
        key = ("low", N, S)
        t = self._ws_cache.pop(key, None)
        if t is None:
            t = torch.empty(self.k.lowdim_scratch_bytes(self.post, N, S) // 4, dtype=torch.float32, device=self.device)
            while len(self._ws_cache) > 6:
                self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = t
        return t

    def _lowdim_attack(self, X, X0, labels, sidx, S, mode, alpha_t, alpha_scalar, alpha_per_image, eps, project, iters):
        """rbnn_lowdim_run(ATTACK): `iters` iterations from X (around X0) in one launch -> a new [N, D_pad] tensor.  FGSM: project False,
        the step is eps itself."""
        out = torch.empty_like(X)               # the kernel writes every column d < D of every row; X is the unpadded [N, D] view
        if not project:
            alpha_scalar, alpha_per_image = eps, False
        self.k.lowdim_run(self.post, self.k.LOWDIM_ATTACK, mode, 0, X, X0, sidx, S, labels, 1.0 / S, 1.0, eps, alpha_t, alpha_scalar, alpha_per_image,
                          project, iters, self._low_scratch(X.shape[0], S), out)
        return out

    graph_safe = True                       # ConvEngine (per-call point blocking) turns this off

    def _graph_capturable(self):
        return (self.graph_safe and self.world == 1 and self.device.type == "cuda" and isinstance(self.k, _hip.HipKernels)
                and os.environ.get("RBNN_HIPGRAPH", "0") == "1")

    def _capture(self, fn):
        """Record fn's launches (all on torch's current stream, through the C-ABI) into a HIP graph; None if capture fails."""
        try:
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fn()
            return graph
        except Exception as exc:            # capture is an optimisation: report and run eagerly
            import warnings
            warnings.warn(f"HIP graph capture of the PGD iteration failed ({exc}); running eagerly")
            return None

    def pgd_continue(self, x, x0, y, n_samples, epsilon, alpha=None, mode=LOSS_MEAN_PROB):
        """ONE PGD iteration from x towards the eps-ball around x0 (SVI: the caller redraws weights between iterations)."""
        sidx, S = self.sample_index(n_samples)
        if self.precision == "lowdim":
            out = self._lowdim_attack(self._flat(x), self._flat(x0), to_labels(y, self.device), sidx, S, mode, None,
                                      0.0 if alpha is None else float(alpha), alpha is None, float(epsilon), True, 1)
            return out.reshape(x.shape)
        X0 = self.pad_inputs(x0, clone=True)
        X = self.pad_inputs(x, clone=True)
        alpha_t = None
        if alpha is None:
            alpha_t = torch.empty(X.shape[0], dtype=torch.float32, device=self.device)
            self.k.pgd_alpha(X0, self.post.D, alpha_t)
        self._step(X, X0, to_labels(y, self.device), sidx, S, mode, alpha_t, 0.0 if alpha is None else float(alpha), float(epsilon), True)
        return self.unpad(X, x)

    # ------------------------------------------------------------------ evaluation
    def clean_outputs(self, x, n_samples, logits=False):
        """The mean output [N, 16] of the CLEAN inputs — attack_evaluation's first forward (adversarialAttacks.py:177-181) — to be handed to
        evaluate(clean=...) by callers that score many attacks of the same inputs with the same samples (an epsilon grid)."""
        sidx, S = self.sample_index(n_samples)
        prep = self._flat if self.precision == "lowdim" else self.pad_inputs
        return self.forward_padded(prep(x), sidx, S, OUT_LOGITS if logits else OUT_PROBS)

    def evaluate(self, x, x_attack, y, n_samples, logits=False, clean=None):
        """attack_evaluation's numbers (adversarialAttacks.py:173-196): (orig acc %, adv acc %, rob [N]).  clean: clean_outputs(x, ...) of
        the same inputs / samples, computed once by the caller — the forward of the clean set is then skipped."""
        sidx, S = self.sample_index(n_samples)
        kind = OUT_LOGITS if logits else OUT_PROBS
        labels = to_labels(y, self.device)
        prep = self._flat if self.precision == "lowdim" else self.pad_inputs
        o = clean if clean is not None else self.forward_padded(prep(x), sidx, S, kind)
        a = self.forward_padded(prep(x_attack), sidx, S, kind)
        counts = torch.zeros(2, dtype=torch.int32, device=self.device)
        rob = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        self.k.eval_metrics(o, a, labels, self.post.C, counts, rob)
        c = counts.cpu()
        n = x.shape[0]
        return 100 * float(c[0]) / n, 100 * float(c[1]) / n, rob, o[:, :self.post.C], a[:, :self.post.C]
