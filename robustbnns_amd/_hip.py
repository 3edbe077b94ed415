"""ctypes binding of librbnn_hip.so (include/robustbnns_hip.h) — the only compute backend.

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
torch is used for device memory and streams only; every kernel is launched through the C-ABI
with raw device pointers (`tensor.data_ptr()`) on torch's current HIP stream.
"""
import ctypes as C
import os

import torch          # must be imported first: the .so then binds to torch's already-loaded libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librbnn_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "robustbnns_hip.h")

CPAD = 16
ACTIVATIONS = {"relu": 0, "leaky": 1, "sigm": 2, "tanh": 3}          # model_nn.py:66-75
ARCHS = {"fc": 0, "fc2": 1}                                           # model_nn.py:77-91
OUT_PROBS, OUT_LOGITS = 0, 1
LOSS_MEAN_PROB, LOSS_PER_SAMPLE, LOSS_MEAN_LOGIT, LOSS_UPSTREAM, LOSS_UPSTREAM_LOGIT = 0, 1, 2, 3, 4

_fp = C.c_void_p


class Posterior(C.Structure):
    _fields_ = [("arch", C.c_int32), ("activation", C.c_int32), ("in_features", C.c_int32),
                ("in_stride", C.c_int32), ("hidden", C.c_int32), ("n_classes", C.c_int32),
                ("n_stored", C.c_int32), ("reserved", C.c_int32),
                ("W1", _fp), ("b1", _fp), ("Wm", _fp), ("bm", _fp), ("W2", _fp), ("b2", _fp),
                ("W1_pack4", _fp), ("Wm_pack4", _fp)]


class Workspace(C.Structure):
    _fields_ = [(k, _fp) for k in ("P", "dZ", "mask1", "dact1", "hid1", "mask2", "dact2", "dhid1", "slabs")]


class WorkspaceSizes(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("P", "dZ", "mask1", "dact1", "hid1", "mask2", "dact2", "dhid1", "slabs")] + \
               [("n_slabs", C.c_int32), ("chunk", C.c_int32)]


class ConvPosterior(C.Structure):
    _fields_ = [("activation", C.c_int32), ("hidden", C.c_int32), ("n_classes", C.c_int32), ("n_stored", C.c_int32),
                ("in_channels", C.c_int32), ("in_width", C.c_int32),
                ("K1w", _fp), ("K1b", _fp), ("K2w", _fp), ("K2b", _fp), ("Fw", _fp), ("Fb", _fp), ("K2w_ci", _fp)]


CONV_WS_KEYS = ("P", "dZ", "P1", "st1", "Q2", "st2", "G")


class ConvWorkspace(C.Structure):
    _fields_ = [(k, _fp) for k in CONV_WS_KEYS]


class ConvWorkspaceSizes(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in CONV_WS_KEYS]


class SplitImages(C.Structure):
    _fields_ = [("W1_rows", _fp), ("W1_cols", _fp), ("W2_gen", _fp), ("ld_rows", C.c_int32), ("ld_cols", C.c_int32),
                ("w1_exp", C.c_int32), ("w2_exp", C.c_int32), ("Wm_rows", _fp), ("Wm_cols", _fp), ("wm_exp", C.c_int32),
                ("h1_exp", C.c_int32)]


class TripleImages(C.Structure):
    _fields_ = [("W1_rows", _fp), ("W1_cols", _fp), ("W2_gen", _fp), ("ld_rows", C.c_int32), ("ld_cols", C.c_int32),
                ("w1_exp", C.c_int32), ("w2_exp", C.c_int32), ("Wm_rows", _fp), ("Wm_cols", _fp), ("wm_exp", C.c_int32),
                ("h1_exp", C.c_int32)]


class SviGuide(C.Structure):
    _fields_ = [(k + sfx, _fp) for k in ("W1", "b1", "Wm", "bm", "W2", "b2") for sfx in ("_loc", "_scale")] + \
               [("hidden", C.c_int32), ("reserved", C.c_int32)]


class SviFlatTensor(C.Structure):
    _fields_ = [("loc", _fp), ("sigma", _fp), ("out", _fp), ("n_elem", C.c_int64), ("out_sample_stride", C.c_int64),
                ("tensor_id", C.c_int32), ("reserved", C.c_int32)]


SVI_EPS_MAX = 6.77                                                     # RBNN_SVI_EPS_MAX: Box-Muller on a 32-bit uniform cannot exceed it

TRIPLE_WS_KEYS = ("X_triple", "dZ_gen", "g_scale", "hid_triple")


class TripleWorkspace(C.Structure):
    _fields_ = [(k, _fp) for k in TRIPLE_WS_KEYS]


class TripleWorkspaceSizes(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in TRIPLE_WS_KEYS]


SPLIT_WS_KEYS = ("X_split", "dZ_gen", "g_scale")


class SplitWorkspace(C.Structure):
    _fields_ = [(k, _fp) for k in SPLIT_WS_KEYS]


class SplitWorkspaceSizes(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in SPLIT_WS_KEYS]


WS_KEYS = ("P", "dZ", "mask1", "dact1", "hid1", "mask2", "dact2", "dhid1", "slabs")

_i32, _f32, _sz, _i64 = C.c_int32, C.c_float, C.c_size_t, C.c_int64
_PP, _PW, _PS = C.POINTER(Posterior), C.POINTER(Workspace), C.POINTER(WorkspaceSizes)

# name -> (restype, argtypes): exactly the declarations of include/robustbnns_hip.h
SIGNATURES = {
    "rbnn_abi_version": (_i32, []),
    "rbnn_build_flags": (_i32, []),
    "rbnn_strerror": (C.c_char_p, [_i32]),
    "rbnn_workspace_query": (_i32, [_PP, _i32, _i32, _i32, _PS]),
    "rbnn_fc_forward": (_i32, [_PP, _fp, _i32, _i32, _fp, _i32, _i32, _PW, _fp]),
    "rbnn_reduce_samples": (_i32, [_fp, _i32, _i32, _i32, _f32, _fp, _i32, _fp]),
    "rbnn_loss_dlogits": (_i32, [_i32, _fp, _fp, _i32, _fp, _fp, _i32, _f32, _i32, _i32, _fp, _fp]),
    "rbnn_fc_input_grad": (_i32, [_PP, _fp, _i32, _i32, _i32, _PW, C.POINTER(_i32), _fp]),
    "rbnn_sum_slabs": (_i32, [_fp, _i32, _i32, _i32, _f32, _fp, _i32, _fp]),
    "rbnn_sum_slabs_norms": (_i32, [_fp, _i32, _i32, _i32, _i32, _f32, _fp, _i32, _fp, _fp, _fp]),
    "rbnn_pgd_alpha": (_i32, [_fp, _i32, _i32, _i32, _fp, _fp]),
    "rbnn_attack_step": (_i32, [_fp, _fp, _i32, _fp, _i32, _sz, _i32, _fp, _f32, _f32, _i32, _i32, _i32, _fp]),
    "rbnn_eval_metrics": (_i32, [_fp, _fp, _i32, _fp, _i32, _i32, _fp, _fp, _fp]),
    "rbnn_pack_rows4": (_i32, [_fp, _i64, _i32, _fp, _fp]),
    "rbnn_conv_workspace_query": (_i32, [C.POINTER(ConvPosterior), _i32, _i32, C.POINTER(ConvWorkspaceSizes)]),
    "rbnn_conv_input_grad": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _i32, C.POINTER(ConvWorkspace), _fp]),
    "rbnn_conv_forward": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _i32, _fp, _i32, _i32, C.POINTER(ConvWorkspace), _fp]),
    "rbnn_svi_materialize": (_i32, [_fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "rbnn_conv_input_grad_split": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _f32, _fp, _i32, _i32, C.POINTER(ConvWorkspace), _fp]),
    "rbnn_conv_forward_split": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _i32, _fp, _fp, _i32, _i32, _fp, _i32, _i32,
                                       C.POINTER(ConvWorkspace), _fp]),
    "rbnn_conv_input_grad_dense": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _f32, _fp, _i32, _i32, C.POINTER(ConvWorkspace), _fp]),
    "rbnn_conv_weight_images": (_i32, [_fp, _i32, _i32, _i32, _fp, _fp, _fp]),
    "rbnn_conv_forward_triple": (_i32, [C.POINTER(ConvPosterior), _fp, _i32, _i32, _fp, _fp, _i32, _i32, _fp, _i32, _i32,
                                        C.POINTER(ConvWorkspace), _fp]),
    "rbnn_input_scales": (_i32, [_fp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _fp, _fp]),
    "rbnn_split_rows": (_i32, [_fp, _i64, _i32, _i32, _i32, _fp, _fp, _i32, _fp]),
    "rbnn_fc_forward_split": (_i32, [_PP, C.POINTER(SplitImages), _fp, _i32, _i32, _fp, _i32, _fp, _i32, _i32, _PW, _fp]),
    "rbnn_split_cols": (_i32, [_fp, _i64, _i32, _i32, _i32, _i32, _fp, _i32, _fp]),
    "rbnn_split_w2gen": (_i32, [_fp, _i32, _i32, _i32, _i32, _fp, _fp]),
    "rbnn_split_workspace_query": (_i32, [_PP, C.POINTER(SplitImages), _i32, _i32, C.POINTER(SplitWorkspaceSizes)]),
    "rbnn_fc_input_grad_split": (_i32, [_PP, C.POINTER(SplitImages), _fp, _i32, _i32, _i32, _PW, C.POINTER(SplitWorkspace),
                                        C.POINTER(_i32), _fp]),
    "rbnn_triple_workspace_query": (_i32, [_PP, C.POINTER(TripleImages), _i32, _i32, C.POINTER(TripleWorkspaceSizes)]),
    "rbnn_triple_rows": (_i32, [_fp, _i64, _i32, _i32, _i32, _fp, _fp, _i32, _fp]),
    "rbnn_triple_rows_grouped": (_i32, [_fp, _i64, _i32, _i32, _i32, _fp, _fp, _i32, _fp]),
    "rbnn_triple_cols": (_i32, [_fp, _i64, _i32, _i32, _i32, _i32, _fp, _i32, _fp]),
    "rbnn_triple_w2gen": (_i32, [_fp, _i32, _i32, _i32, _i32, _fp, _fp]),
    "rbnn_fc_forward_triple": (_i32, [_PP, C.POINTER(TripleImages), C.POINTER(TripleWorkspace), _i32, _fp, _i32, _fp, _i32, _i32, _PW, _fp]),
    "rbnn_fc_input_grad_triple": (_i32, [_PP, C.POINTER(TripleImages), _fp, _i32, _i32, _i32, _PW, C.POINTER(TripleWorkspace),
                                         C.POINTER(_i32), _fp]),
    "rbnn_step_tail_triple": (_i32, [_i32, _fp, _fp, _i32, _f32, _i32, _i32, _fp, _i32, C.POINTER(TripleWorkspace), _fp]),
    "rbnn_attack_step_triple": (_i32, [_fp, _fp, _i32, _fp, _i32, _sz, _i32, _fp, _f32, _f32, _i32, _i32, _i32, _fp, _fp, _i32, _fp]),
    "rbnn_lowdim_supported": (_i32, [_PP]),
    "rbnn_lowdim_scratch_bytes": (_sz, [_PP, _i32, _i32]),
    "rbnn_lowdim_run": (_i32, [_PP, _i32, _i32, _i32, _fp, _fp, _i32, _i32, _fp, _i32, _fp, _f32, _f32, _f32, _fp, _f32, _i32, _i32, _i32,
                               _fp, _fp, _i32, _fp, _fp, _fp]),
    "rbnn_lowdim_fused_draw_supported": (_i32, [_PP, _i32, _i32]),
    "rbnn_lowdim_run_svi": (_i32, [_PP, C.POINTER(SviGuide), _fp, C.c_uint64, C.c_uint32, _i32, _i32, _i32, _fp, _fp, _i32, _i32, _fp, _i32, _fp, _f32,
                                   _f32, _f32, _fp, _f32, _i32, _i32, _i32, _fp, _fp, _i32, _fp, _fp, _fp]),
    "rbnn_svi_draw_flat": (_i32, [C.POINTER(SviFlatTensor), _i32, _i32, _fp, C.c_uint64, C.c_uint32, _fp]),
    "rbnn_svi_draw": (_i32, [_PP, C.POINTER(TripleImages), C.POINTER(SviGuide), _i32, _fp, C.c_uint64, C.c_uint32, _fp]),
    "rbnn_svi_draw_supported": (_i32, [_PP, _i32]),
    "rbnn_svi_draw_images": (_i32, [_PP, C.POINTER(TripleImages), C.POINTER(SviGuide), _i32, _fp, C.c_uint64, C.c_uint32, _fp]),
}

_lib = None
ABI_VERSION = 9


class HipError(RuntimeError):
    pass


def load():
    """Load librbnn_hip.so (built by __graft_entry__.build()).  Raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                           "robustbnns_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)            # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.rbnn_abi_version() != ABI_VERSION:
            raise HipError(f"librbnn_hip.so ABI version {lib.rbnn_abi_version()} != {ABI_VERSION}: rebuild it (__graft_entry__.build())")
        flags = lib.rbnn_build_flags()
        if flags != 0 and os.environ.get("RBNN_ALLOW_ABLATION") != "1":
            raise HipError(f"librbnn_hip.so was built with timing-only ablation switches (rbnn_build_flags() = {flags}): its kernels compute "
                           "wrong results by design.  Rebuild it (`python __graft_entry__.py --force`); the scripts under tools/ that build "
                           "such variants set RBNN_ALLOW_ABLATION=1 for their own runs.")
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise HipError(f"{what} failed: {load().rbnn_strerror(rc).decode()} ({rc})")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(t, name="tensor"):
    if t.device.type != "cuda":
        raise HipError(f"{name} is on {t.device}: the MI355X HIP kernels are the only compute path "
                       "(no CPU fallback); move it to 'cuda'.")
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise HipError(f"{name} must be float32/int32, got {t.dtype}")
    if not t.is_contiguous():
        raise HipError(f"{name} must be contiguous")


class HipKernels:
    """Tensor-level façade over the C-ABI.  `net` is a robustbnns_amd.posterior.StackedPosterior,
    `ws` a dict of torch tensors keyed like rbnn_workspace."""

    name = "hip"

    def __init__(self):
        self.lib = load()

    # -- host-only --------------------------------------------------------------------------------
    def workspace_sizes(self, net, N, S, chunk=0):
        out = WorkspaceSizes()
        check(self.lib.rbnn_workspace_query(C.byref(net.descriptor(lazy_ok=True)), N, S, chunk, C.byref(out)), "rbnn_workspace_query")
        d = {k: getattr(out, k) for k in WS_KEYS}
        d["n_slabs"], d["chunk"] = out.n_slabs, out.chunk
        return d

    @staticmethod
    def _ws(ws):
        w = Workspace()
        for k in WS_KEYS:
            setattr(w, k, ptr(ws.get(k)))
        return w

    # -- device -----------------------------------------------------------------------------------
    def fc_forward(self, net, X, sidx, S, out_kind, ws):
        require_gpu(X, "X")
        w = self._ws(ws)
        check(self.lib.rbnn_fc_forward(C.byref(net.descriptor()), ptr(X), X.stride(0), X.shape[0], ptr(sidx), S,
                                       out_kind, C.byref(w), stream_of(X)), "rbnn_fc_forward")

    def reduce_samples(self, P, S, N, Cn, scale, out):
        require_gpu(P, "P")
        check(self.lib.rbnn_reduce_samples(ptr(P), S, N, Cn, scale, ptr(out), out.stride(0), stream_of(P)), "rbnn_reduce_samples")

    def loss_dlogits(self, mode, P, Psum, G_up, labels, S, inv_S, N, Cn, dZ):
        require_gpu(P, "P")
        ldp = Psum.stride(0) if Psum is not None else (G_up.stride(0) if G_up is not None else CPAD)
        check(self.lib.rbnn_loss_dlogits(mode, ptr(P), ptr(Psum), ldp, ptr(G_up), ptr(labels), S, inv_S, N, Cn,
                                         ptr(dZ), stream_of(P)), "rbnn_loss_dlogits")

    def fc_input_grad(self, net, sidx, S, N, chunk, ws):
        w = self._ws(ws)
        n = C.c_int32(0)
        check(self.lib.rbnn_fc_input_grad(C.byref(net.descriptor()), ptr(sidx), S, N, chunk, C.byref(w), C.byref(n),
                                          stream_of(ws["dZ"])), "rbnn_fc_input_grad")
        return n.value

    def sum_slabs(self, slabs, K, N, Dp, scale, out):
        require_gpu(slabs, "slabs")
        check(self.lib.rbnn_sum_slabs(ptr(slabs), K, N, Dp, scale, ptr(out), out.stride(0), stream_of(slabs)), "rbnn_sum_slabs")

    def sum_slabs_norms(self, slabs, K, N, Dp, D, scale, out, linf, l2):
        """sum_slabs + per-point Linf / L2 norms of the result over the D real columns, one pass."""
        require_gpu(slabs, "slabs")
        check(self.lib.rbnn_sum_slabs_norms(ptr(slabs), K, N, Dp, D, scale, ptr(out), out.stride(0), ptr(linf), ptr(l2),
                                            stream_of(slabs)), "rbnn_sum_slabs_norms")

    def pgd_alpha(self, X0, D, alpha):
        require_gpu(X0, "X0")
        check(self.lib.rbnn_pgd_alpha(ptr(X0), X0.stride(0), X0.shape[0], D, ptr(alpha), stream_of(X0)), "rbnn_pgd_alpha")

    def attack_step(self, X, X0, G, K, slab_stride, ldg, alpha, alpha_scalar, eps, project, D):
        require_gpu(X, "X")
        check(self.lib.rbnn_attack_step(ptr(X), ptr(X0), X.stride(0), ptr(G), K, slab_stride, ldg, ptr(alpha),
                                        alpha_scalar, eps, int(project), X.shape[0], D, stream_of(X)), "rbnn_attack_step")

    def eval_metrics(self, A, B, labels, Cn, counts, rob):
        require_gpu(A, "outputs")
        check(self.lib.rbnn_eval_metrics(ptr(A), ptr(B), A.stride(0), ptr(labels), A.shape[0], Cn, ptr(counts), ptr(rob),
                                         stream_of(A)), "rbnn_eval_metrics")

    def pack_rows4(self, W, out):
        """[rows, cols] row-major -> [rows/4, cols, 4] (rows = leading dims of W flattened)."""
        require_gpu(W, "W")
        cols = W.shape[-1]
        check(self.lib.rbnn_pack_rows4(ptr(W), W.numel() // cols, cols, ptr(out), stream_of(W)), "rbnn_pack_rows4")

    # -- split-half ("f16x3") precision mode ---------------------------------------------------------
    def input_scales(self, X, cols, floor_abs, mul, add, cap, out):
        """Device-resident operand scales of the inputs X [N, ld] (record 0) and of the activations they bound (record 1):
        out = int32[8] on the device (two rbnn_dev_scale records).  No host sync."""
        require_gpu(X, "X")
        check(self.lib.rbnn_input_scales(ptr(X), X.shape[0], cols, X.stride(0), floor_abs, mul, add, cap, ptr(out), stream_of(X)),
              "rbnn_input_scales")
        return out

    def split_rows(self, src, cols, scale_exp, out, ld_dst, dev_scale=None):
        """src: [..., ld_src] fp32 rows -> out: split-rows image (int16 storage [rows, ld_dst*2]).  dev_scale: an
        rbnn_dev_scale record on the device that replaces scale_exp."""
        require_gpu(src, "src")
        ld_src = src.shape[-1]
        check(self.lib.rbnn_split_rows(ptr(src), src.numel() // ld_src, cols, ld_src, scale_exp, ptr(dev_scale), ptr(out), ld_dst,
                                       stream_of(src)), "rbnn_split_rows")

    def fc_forward_split(self, net, images, Xs, ld, x_exp, N, sidx, S, out_kind, ws, dev_scales=None):
        w = self._ws(ws)
        check(self.lib.rbnn_fc_forward_split(C.byref(net.descriptor()), C.byref(images), ptr(Xs), ld, x_exp, ptr(dev_scales), N,
                                             ptr(sidx), S, out_kind, C.byref(w), stream_of(Xs)), "rbnn_fc_forward_split")

    def split_cols(self, W, rows, cols, scale_exp, out, ld_dst):
        """W: [n_mats, rows, ld_src] fp32 -> out: split-cols image."""
        require_gpu(W, "W")
        check(self.lib.rbnn_split_cols(ptr(W), W.numel() // (rows * W.shape[-1]), rows, cols, W.shape[-1], scale_exp, ptr(out),
                                       ld_dst, stream_of(W)), "rbnn_split_cols")

    def split_w2gen(self, W2, Cn, H, scale_exp, out):
        require_gpu(W2, "W2")
        check(self.lib.rbnn_split_w2gen(ptr(W2), W2.numel() // (Cn * H), Cn, H, scale_exp, ptr(out), stream_of(W2)), "rbnn_split_w2gen")

    def split_workspace_sizes(self, net, images, N, S):
        out = SplitWorkspaceSizes()
        check(self.lib.rbnn_split_workspace_query(C.byref(net.descriptor()), C.byref(images), N, S, C.byref(out)),
              "rbnn_split_workspace_query")
        return {k: getattr(out, k) for k in SPLIT_WS_KEYS}

    def fc_input_grad_split(self, net, images, sidx, S, N, chunk, ws, sws):
        w = self._ws(ws)
        sw = SplitWorkspace()
        for k in SPLIT_WS_KEYS:
            setattr(sw, k, ptr(sws.get(k)))
        n = C.c_int32(0)
        check(self.lib.rbnn_fc_input_grad_split(C.byref(net.descriptor()), C.byref(images), ptr(sidx), S, N, chunk, C.byref(w),
                                                C.byref(sw), C.byref(n), stream_of(ws["dZ"])), "rbnn_fc_input_grad_split")
        return n.value

    # -- triple-split ("f16x6") mode: full-width fp32 operands on the f16 matrix pipe ---------------------
    def triple_rows(self, src, cols, scale_exp, out, ld_dst, dev_scale=None, grouped=False):
        """src: [..., ld_src] fp32 rows -> out: triple-rows image (int16 storage, 3 halves per element).  grouped=True: the fc forward's
        operand order (16-row groups, [3 pieces][16 rows][64 B] per K stage; out sized for ceil16(rows) rows)."""
        require_gpu(src, "src")
        ld_src = src.shape[-1]
        fn = self.lib.rbnn_triple_rows_grouped if grouped else self.lib.rbnn_triple_rows
        check(fn(ptr(src), src.numel() // ld_src, cols, ld_src, scale_exp, ptr(dev_scale), ptr(out), ld_dst, stream_of(src)),
              "rbnn_triple_rows_grouped" if grouped else "rbnn_triple_rows")

    def triple_cols(self, W, rows, cols, scale_exp, out, ld_dst):
        require_gpu(W, "W")
        check(self.lib.rbnn_triple_cols(ptr(W), W.numel() // (rows * W.shape[-1]), rows, cols, W.shape[-1], scale_exp, ptr(out),
                                        ld_dst, stream_of(W)), "rbnn_triple_cols")

    def triple_w2gen(self, W2, Cn, H, scale_exp, out):
        require_gpu(W2, "W2")
        check(self.lib.rbnn_triple_w2gen(ptr(W2), W2.numel() // (Cn * H), Cn, H, scale_exp, ptr(out), stream_of(W2)), "rbnn_triple_w2gen")

    def triple_workspace_sizes(self, net, images, N, S):
        out = TripleWorkspaceSizes()
        check(self.lib.rbnn_triple_workspace_query(C.byref(net.descriptor(lazy_ok=True)), C.byref(images), N, S, C.byref(out)),
              "rbnn_triple_workspace_query")
        return {k: getattr(out, k) for k in TRIPLE_WS_KEYS}

    @staticmethod
    def _tws(tws):
        t = TripleWorkspace()
        for k in TRIPLE_WS_KEYS:
            setattr(t, k, ptr(tws.get(k)))
        return t

    def fc_forward_triple(self, net, images, tws, x_exp, N, sidx, S, out_kind, ws, dev_scales=None):
        w, t = self._ws(ws), self._tws(tws)
        # (lazy_ok: a pending images-only draw left the fp32 W1 / Wm stale, which the triple kernels never read)
        check(self.lib.rbnn_fc_forward_triple(C.byref(net.descriptor(lazy_ok=True)), C.byref(images), C.byref(t), x_exp, ptr(dev_scales), N,
                                              ptr(sidx), S, out_kind, C.byref(w), stream_of(tws["X_triple"])), "rbnn_fc_forward_triple")

    def step_tail_triple(self, mode, P, labels, S, inv_S, N, Cn, tws, Psum=None):
        """reduce over samples + loss + dZ generator image in one launch (rbnn_step_tail_triple); the fp32 dZ is not written."""
        require_gpu(P, "P")
        t = self._tws(tws)
        check(self.lib.rbnn_step_tail_triple(mode, ptr(P), ptr(labels), S, inv_S, N, Cn, ptr(Psum), 0 if Psum is None else Psum.stride(0), C.byref(t),
                                             stream_of(P)), "rbnn_step_tail_triple")

    def attack_step_triple(self, X, X0, G, K, slab_stride, ldg, alpha, alpha_scalar, eps, project, D, dev_scale, X_triple, ld_rows):
        """attack_step + the grouped triple-rows image of the new iterate in one launch (rbnn_attack_step_triple)."""
        require_gpu(X, "X")
        check(self.lib.rbnn_attack_step_triple(ptr(X), ptr(X0), X.stride(0), ptr(G), K, slab_stride, ldg, ptr(alpha), alpha_scalar, eps, int(project),
                                               X.shape[0], D, ptr(dev_scale), ptr(X_triple), ld_rows, stream_of(X)), "rbnn_attack_step_triple")

    def fc_input_grad_triple(self, net, images, sidx, S, N, chunk, ws, tws, dz_ready=False):
        """dz_ready: tws['dZ_gen'] / tws['g_scale'] were built by step_tail_triple — the fp32 dZ is not read."""
        w, t = self._ws(ws), self._tws(tws)
        if dz_ready:
            w.dZ = None
        n = C.c_int32(0)
        check(self.lib.rbnn_fc_input_grad_triple(C.byref(net.descriptor(lazy_ok=True)), C.byref(images), ptr(sidx), S, N, chunk, C.byref(w),
                                                 C.byref(t), C.byref(n), stream_of(ws["slabs"])), "rbnn_fc_input_grad_triple")
        return n.value

    # -- conv architecture ---------------------------------------------------------------------------
    def conv_workspace_sizes(self, net, N, S):
        out = ConvWorkspaceSizes()
        check(self.lib.rbnn_conv_workspace_query(C.byref(net.descriptor()), N, S, C.byref(out)), "rbnn_conv_workspace_query")
        return {k: getattr(out, k) for k in CONV_WS_KEYS}

    @staticmethod
    def _conv_ws(ws):
        w = ConvWorkspace()
        for k in CONV_WS_KEYS:
            setattr(w, k, ptr(ws.get(k)))
        return w

    def conv_forward(self, net, X, sidx, S, out_kind, ws):
        require_gpu(X, "X")
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_forward(C.byref(net.descriptor()), ptr(X), X.stride(0), X.shape[0], ptr(sidx), S, out_kind,
                                         C.byref(w), stream_of(X)), "rbnn_conv_forward")

    def conv_forward_split(self, net, K2_rows, k2_exp, p1_exp, X, sidx, S, out_kind, ws, p1_dev_scale=None):
        require_gpu(X, "X")
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_forward_split(C.byref(net.descriptor()), ptr(K2_rows), k2_exp, p1_exp, ptr(p1_dev_scale), ptr(X),
                                               X.stride(0), X.shape[0],
                                               ptr(sidx), S, out_kind, C.byref(w), stream_of(X)), "rbnn_conv_forward_split")

    def conv_forward_triple(self, net, K2_triple, k2_exp, p1_exp, X, sidx, S, out_kind, ws, p1_dev_scale=None):
        require_gpu(X, "X")
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_forward_triple(C.byref(net.descriptor()), ptr(K2_triple), k2_exp, p1_exp, ptr(p1_dev_scale), ptr(X),
                                                X.stride(0), X.shape[0], ptr(sidx), S, out_kind, C.byref(w), stream_of(X)),
              "rbnn_conv_forward_triple")

    def conv_input_grad_dense(self, net, K2_dense, k2_exp, fw_l1, sidx, S, N, ws):
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_input_grad_dense(C.byref(net.descriptor()), ptr(K2_dense), k2_exp, fw_l1, ptr(sidx), S, N, C.byref(w),
                                                  stream_of(ws["dZ"])), "rbnn_conv_input_grad_dense")
        return S

    def conv_weight_images(self, K2w, S, H, k2_exp, rows=None, dense=None):
        """model.3.weight's forward (grouped tap-major rows) and dense conv2^T triple images from the fp32 stack in one launch."""
        require_gpu(K2w, "K2w")
        check(self.lib.rbnn_conv_weight_images(ptr(K2w), S, H, k2_exp, ptr(rows), ptr(dense), stream_of(K2w)), "rbnn_conv_weight_images")

    def conv_input_grad_split(self, net, K2_bwd, k2_exp, fw_l1, sidx, S, N, ws):
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_input_grad_split(C.byref(net.descriptor()), ptr(K2_bwd), k2_exp, fw_l1, ptr(sidx), S, N, C.byref(w),
                                                  stream_of(ws["dZ"])), "rbnn_conv_input_grad_split")
        return S

    def conv_input_grad(self, net, sidx, S, N, ws):
        w = self._conv_ws(ws)
        check(self.lib.rbnn_conv_input_grad(C.byref(net.descriptor()), ptr(sidx), S, N, C.byref(w), stream_of(ws["dZ"])),
              "rbnn_conv_input_grad")
        return S                                                          # one "slab" per sample

    def svi_materialize(self, loc, scale_raw, eps, out):
        require_gpu(loc, "loc")
        check(self.lib.rbnn_svi_materialize(ptr(loc), ptr(scale_raw), ptr(eps), loc.numel(), eps.shape[0], ptr(out),
                                            stream_of(loc)), "rbnn_svi_materialize")

    def svi_draw_supported(self, net, with_triple_images):
        return bool(self.lib.rbnn_svi_draw_supported(C.byref(net.descriptor()), int(bool(with_triple_images))))

    def svi_draw(self, net, images, guide, S, key, draw_id, sample_keys=None, images_only=False):
        """One launch: samples [0, S) of the stacked posterior `net` and all its weight images redrawn IN PLACE from the guide
        (robustbnns_amd.posterior.SviGuide).  images: the posterior's TripleImages or None.  sample_keys: int64 device tensor [S].
        images_only (rbnn_svi_draw_images): W1 / Wm into the triple images only — their fp32 stack and pack_rows4 copies are skipped."""
        w1 = net.__dict__.get("_t_W1", None) if hasattr(net, "__dict__") else None
        w1 = net.W1 if w1 is None else w1
        require_gpu(w1, "W1")
        fn = self.lib.rbnn_svi_draw_images if images_only else self.lib.rbnn_svi_draw
        check(fn(C.byref(net.descriptor(lazy_ok=True)), None if images is None else C.byref(images), C.byref(guide.descriptor()), S,
                 ptr(sample_keys), C.c_uint64(key & 0xFFFFFFFFFFFFFFFF), C.c_uint32(draw_id & 0xFFFFFFFF),
                 stream_of(w1)), "rbnn_svi_draw_images" if images_only else "rbnn_svi_draw")

    # -- low-dimensional fc nets: the whole hot path in one launch (rbnn_lowdim.hip) -------------------------
    LOWDIM_FORWARD, LOWDIM_GRADIENT, LOWDIM_ATTACK = 0, 1, 2

    def lowdim_supported(self, net):
        return bool(self.lib.rbnn_lowdim_supported(C.byref(net.descriptor(lazy_ok=True))))

    def lowdim_scratch_bytes(self, net, N, S):
        return int(self.lib.rbnn_lowdim_scratch_bytes(C.byref(net.descriptor(lazy_ok=True)), N, S))

    def lowdim_fused_draw_supported(self, net, N, S):
        return bool(self.lib.rbnn_lowdim_fused_draw_supported(C.byref(net.descriptor(lazy_ok=True)), N, S))

    def lowdim_run(self, net, op, loss_mode, out_kind, X, X0, sidx, S, labels, inv_S, out_scale, eps, alpha, alpha_scalar, alpha_per_image,
                   project, iters, P, out, linf=None, l2=None):
        require_gpu(X, "X")
        lazy = getattr(net, "_lazy", None)
        covered = sidx is None or getattr(sidx, "_rbnn_max_index", 1 << 62) < (lazy[2] if lazy is not None else 0)
        if (lazy is not None and getattr(net, "_triple", None) is None and covered and S <= lazy[2]
                and self.lowdim_fused_draw_supported(net, X.shape[0], S)):
            # a pending (lazy) SVI draw: the weights are generated inside this launch — no rbnn_svi_draw launch, the stack stays as it was.
            # Only when every sample the call names is one the draw covers: the identity map over the first S <= drawn samples, or an index
            # buffer whose largest index is KNOWN on the host (AttackEngine.sample_index records it) and below the drawn count — the kernel
            # reads sample_keys[sidx[s]].  Any other index buffer materialises the draw first (net.descriptor() below does) and reads the stack
            key, draw_id, _, sample_keys = lazy
            check(self.lib.rbnn_lowdim_run_svi(C.byref(net.descriptor(lazy_ok=True)), C.byref(net._guide.descriptor()), ptr(sample_keys),
                                               C.c_uint64(key & 0xFFFFFFFFFFFFFFFF), C.c_uint32(draw_id & 0xFFFFFFFF), op, loss_mode, out_kind, ptr(X),
                                               ptr(X0), X.stride(0), X.shape[0], ptr(sidx), S, ptr(labels), inv_S, out_scale, eps, ptr(alpha), alpha_scalar,
                                               int(alpha_per_image), int(project), iters, ptr(P), ptr(out), out.stride(0), ptr(linf), ptr(l2),
                                               stream_of(X)), "rbnn_lowdim_run_svi")
            return
        check(self.lib.rbnn_lowdim_run(C.byref(net.descriptor()), op, loss_mode, out_kind, ptr(X), ptr(X0), X.stride(0), X.shape[0], ptr(sidx), S,
                                       ptr(labels), inv_S, out_scale, eps, ptr(alpha), alpha_scalar, int(alpha_per_image), int(project), iters,
                                       ptr(P), ptr(out), out.stride(0), ptr(linf), ptr(l2), stream_of(X)), "rbnn_lowdim_run")

    def svi_draw_flat(self, items, S, key, draw_id, sample_keys=None):
        """items: list of (loc, sigma = softplus(raw scale), out [S, ...], tensor_id) — every tensor of a net redrawn in place by ONE launch."""
        arr = (SviFlatTensor * len(items))()
        for i, (loc, scl, out, tid) in enumerate(items):
            require_gpu(out, "out")
            arr[i].loc, arr[i].sigma, arr[i].out = loc.data_ptr(), scl.data_ptr(), out.data_ptr()
            arr[i].n_elem, arr[i].out_sample_stride, arr[i].tensor_id = loc.numel(), out.stride(0), tid
        check(self.lib.rbnn_svi_draw_flat(arr, len(items), S, ptr(sample_keys), C.c_uint64(key & 0xFFFFFFFFFFFFFFFF), C.c_uint32(draw_id & 0xFFFFFFFF),
                                          stream_of(items[0][2])), "rbnn_svi_draw_flat")
