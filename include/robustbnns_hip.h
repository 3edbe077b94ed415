/*
 * robustbnns_hip.h — C-ABI of the MI355X (gfx950) Bayesian attack / expected-loss-gradient
 * hot path.  Plain pointers and sizes only: no torch types, no C++ in the signatures.
 *
 * The reference (ginevracoal/robustBNNs) has NO native boundary for this path: it is a
 * Python loop nest that dispatches stock torch ops with batch 1.  Each entry point below
 * therefore cites the Python statements it replaces (file:line in the reference); the
 * reference-side binding a maintainer would add is the ctypes stub in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.empty(device="cuda")) unless
 *     marked [host]; all floating-point data is IEEE fp32, row-major, 16-byte aligned;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it,
 *     re-entrant, keeps no global state and allocates nothing: the caller owns every buffer;
 *   - return value: RBNN_OK (0) or a negative rbnn_status; nothing throws;
 *   - a *posterior* is S_total stacked weight samples (HMC chain / SVI draws / ensemble
 *     members); `sample_idx` (int32[S], may be NULL = 0..S-1) selects which stored samples
 *     a call uses, in order — model_bnn.py:246-252 (`seeds` index the stored samples).
 *
 * Padding contract (lets every kernel run whole 16-wide MFMA tiles, no tail code in the
 * K loops): D_pad = round_up(D,16) is the row stride of X, W1 and of gradient buffers;
 * columns [D, D_pad) are ZERO.  hidden is a multiple of 32 (the reference only allows
 * powers of two >= 16, model_nn.py:39-40; 16 is zero-padded to 32 by the host, which is
 * exact: a padded unit has zero outgoing weights).  n_classes <= 16.
 */
#ifndef ROBUSTBNNS_HIP_H
#define ROBUSTBNNS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBNN_ABI_VERSION 9
#define RBNN_CPAD 16               /* class axis of P / dZ buffers is padded to 16 floats */

typedef enum rbnn_status {
    RBNN_OK = 0,
    RBNN_ERR_NULL = -1,            /* required pointer is NULL                      */
    RBNN_ERR_SHAPE = -2,           /* dimension violates the padding contract       */
    RBNN_ERR_UNSUPPORTED = -3,     /* arch / activation / mode not implemented      */
    RBNN_ERR_LAUNCH = -4,          /* HIP reported an error at launch               */
    RBNN_ERR_ALIGN = -5            /* pointer not 16-byte aligned                   */
} rbnn_status;

/* model_nn.py:66-75 */
typedef enum rbnn_activation { RBNN_ACT_RELU = 0, RBNN_ACT_LEAKY = 1, RBNN_ACT_SIGM = 2, RBNN_ACT_TANH = 3 } rbnn_activation;
/* model_nn.py:77-91 */
typedef enum rbnn_arch { RBNN_ARCH_FC = 0, RBNN_ARCH_FC2 = 1 } rbnn_arch;

/* What the stacked forward leaves in `P`:
 *   RBNN_OUT_PROBS  softmax(logits)  — BNN.forward, model_bnn.py:133-134, :253-255
 *   RBNN_OUT_LOGITS raw logits       — NN.forward / Ensemble_NN.forward, model_nn.py:139, model_ensemble.py:64 */
typedef enum rbnn_out_kind { RBNN_OUT_PROBS = 0, RBNN_OUT_LOGITS = 1 } rbnn_out_kind;

/* Which loss the input gradient is taken of:
 *   RBNN_LOSS_MEAN_PROB   CE(mean_s p_s, y)        fgsm/pgd on a BNN   adversarialAttacks.py:74-78, :97-101
 *   RBNN_LOSS_PER_SAMPLE  mean_s CE(p_s, y)        loss_gradient       lossGradients.py:29-40
 *   RBNN_LOSS_MEAN_LOGIT  CE(mean_s z_s, y)        fgsm/pgd on NN / Ensemble_NN (P holds logits)
 *   RBNN_LOSS_UPSTREAM    caller supplies dL/d(mean_s p_s) [N,C] (autograd hook on BNN.forward: the softmax backward is applied)
 *   RBNN_LOSS_UPSTREAM_LOGIT  caller supplies dL/d(mean_s z_s) [N,C] (autograd hook on NN / Ensemble_NN.forward: P holds logits) */
typedef enum rbnn_loss_mode { RBNN_LOSS_MEAN_PROB = 0, RBNN_LOSS_PER_SAMPLE = 1, RBNN_LOSS_MEAN_LOGIT = 2, RBNN_LOSS_UPSTREAM = 3,
                              RBNN_LOSS_UPSTREAM_LOGIT = 4 } rbnn_loss_mode;

/* Stacked posterior of a fully-connected net (model_nn.py:77-91; state_dict keys in comments). */
typedef struct rbnn_posterior {
    int32_t arch;                  /* rbnn_arch                                                  */
    int32_t activation;            /* rbnn_activation                                            */
    int32_t in_features;           /* D  (784 MNIST, 2 half-moons)                               */
    int32_t in_stride;             /* D_pad, multiple of 16: row stride of W1                    */
    int32_t hidden;                /* H, multiple of 32                                          */
    int32_t n_classes;             /* C <= 16                                                    */
    int32_t n_stored;              /* S_total                                                    */
    int32_t reserved;
    const float *W1, *b1;          /* model.1.weight [S_total,H,D_pad], model.1.bias [S_total,H] */
    const float *Wm, *bm;          /* fc2 only: model.3.weight [S_total,H,H], model.3.bias       */
    const float *W2, *b2;          /* output layer (model.3 for fc, model.5 for fc2): [S_total,C,H], [S_total,C] */
    /* rbnn_pack_rows4() images of W1 / Wm: [S_total, H/4, cols, 4] — four consecutive hidden units interleaved per
     * column, so the backward GEMM's B operand (4 K steps of one column) is one 16-byte LDS read.  Required by
     * rbnn_fc_input_grad (Wm_pack4 for fc2 only); the forward reads the plain row-major W1 / Wm. */
    const float *W1_pack4, *Wm_pack4;
} rbnn_posterior;

/* Caller-owned scratch for one (N, S) problem; sizes from rbnn_workspace_query(). */
typedef struct rbnn_workspace {
    float    *P;                   /* [S,N,16]      per-sample probabilities (or logits)         */
    float    *dZ;                  /* [S,N,16]      dL/dlogits per sample                        */
    uint32_t *mask1;               /* [S,H/32,N_pad] bit h%32 of word [s][h/32][n] = (pre-activation > 0); N_pad = round_up(N,256) */
    float    *dact1;               /* [S,N,H]       act'(pre-activation), sigm/tanh only         */
    float    *hid1;                /* [S,N,H]       fc2: first hidden activations                */
    uint32_t *mask2;               /* fc2: as mask1 for the second hidden layer                  */
    float    *dact2;               /* fc2, sigm/tanh                                             */
    float    *dhid1;               /* [S,N,H]       fc2: dL/d(pre-activation 1)                  */
    float    *slabs;               /* [n_slabs,N,D_pad] partial input gradients                  */
} rbnn_workspace;

typedef struct rbnn_workspace_sizes {  /* bytes; 0 = not needed for this posterior */
    size_t P, dZ, mask1, dact1, hid1, mask2, dact2, dhid1, slabs;
    int32_t n_slabs;               /* slabs rbnn_fc_input_grad will write for (S, chunk)         */
    int32_t chunk;                 /* samples accumulated per slab (chosen if chunk<=0 passed)   */
} rbnn_workspace_sizes;

int rbnn_abi_version(void);
/* 0 for a product build.  Bit 0: a translation unit of this library was compiled with a timing-only ablation switch (RBNN_ABL,
 * RBNN_*_ABL_*, RBNN_FAST_BUILD — csrc/rbnn_common.hpp; such kernels compute WRONG results by design and need -DRBNN_ALLOW_ABLATION
 * to compile at all).  robustbnns_amd._hip.load() refuses a library whose flags are not 0 unless RBNN_ALLOW_ABLATION=1. */
int rbnn_build_flags(void);
const char *rbnn_strerror(int status);

/* Sizes of every workspace buffer for N points x S used samples.  chunk<=0: library picks. [host] */
int rbnn_workspace_query(const rbnn_posterior *net, int32_t n_points, int32_t n_samples,
                         int32_t chunk, rbnn_workspace_sizes *out);

/* Stacked per-sample forward: for every used sample s, P[s] = softmax(NN_s(X)) (or logits),
 * and the activation-derivative stash the backward needs.
 * Replaces: the `for seed in seeds: net.forward(inputs); softmax` loop, model_bnn.py:251-255,
 * with NN.forward = model_nn.py:126-141 (fc :77-82, fc2 :84-91), batched over all N points.
 * X: [N, ldx], ldx >= D_pad, multiple of 4. */
int rbnn_fc_forward(const rbnn_posterior *net, const float *X, int32_t ldx, int32_t n_points,
                    const int32_t *sample_idx, int32_t n_samples, int32_t out_kind,
                    const rbnn_workspace *ws, void *stream);

/* out[n,c] = scale * sum_s P[s,n,c]   (c < C; out row stride ldo floats).
 * Replaces torch.stack(preds).mean(0), model_bnn.py:257 (scale = 1/S); with scale = 1 it is the
 * local partial sum that the multi-GPU path all-reduces. */
int rbnn_reduce_samples(const float *P, int32_t n_samples, int32_t n_points, int32_t n_classes,
                        float scale, float *out, int32_t ldo, void *stream);

/* dZ[s,n,:] = dL/dlogits_s for the chosen loss; `Psum` [N,ldp] is sum_s P[s] over ALL samples of
 * the job (after the all-reduce when samples are sharded), `inv_S` = 1/(total samples),
 * `labels` int32[N] (argmax of the one-hot, lossGradients.py:23, adversarialAttacks.py:120),
 * `G_up` [N,ldp] only for RBNN_LOSS_UPSTREAM / RBNN_LOSS_UPSTREAM_LOGIT.
 * Replaces CrossEntropyLoss()(output,label) + the softmax part of loss.backward():
 * adversarialAttacks.py:76-78, :99-101; lossGradients.py:34-36 (closed form). */
int rbnn_loss_dlogits(int32_t mode, const float *P, const float *Psum, int32_t ldp, const float *G_up,
                      const int32_t *labels, int32_t n_samples, float inv_S, int32_t n_points,
                      int32_t n_classes, float *dZ, void *stream);

/* Partial expected input gradients: slabs[k,n,:] = sum_{s in chunk k} dA_s[n,:] . W1_s, with
 * dA_s = act'(A_s) * (dZ_s . W2_s) built on the fly from ws->dZ and the stash of rbnn_fc_forward.
 * Replaces the rest of loss.backward() down to image.grad (adversarialAttacks.py:78-79, :101;
 * lossGradients.py:36-40).  Writes *n_slabs_out slabs of [N, D_pad] into ws->slabs. [n_slabs_out: host] */
int rbnn_fc_input_grad(const rbnn_posterior *net, const int32_t *sample_idx, int32_t n_samples,
                       int32_t n_points, int32_t chunk, const rbnn_workspace *ws,
                       int32_t *n_slabs_out, void *stream);

/* out[n,d] = scale * sum_k slabs[k,n,d]  (d < D_pad; out row stride ldo).
 * Replaces torch.stack(loss_gradients,0).mean(0), lossGradients.py:40 (1/S is already in dZ). */
int rbnn_sum_slabs(const float *slabs, int32_t n_slabs, int32_t n_points, int32_t d_pad, float scale,
                   float *out, int32_t ldo, void *stream);

/* rbnn_sum_slabs with the norms of every point's result fused into the same pass (no second read of the gradients):
 * linf[n] = max_{d<D} |out[n,d]|, l2[n] = sqrt(sum_{d<D} out[n,d]^2)  (fp32, fixed summation order: reproducible).
 * Replaces np.max(np.abs(g)) / np.linalg.norm(g) per image in compute_vanishing_norms_idxs, lossGradients.py:91-94,102-105
 * (and plot_gradients_components.py:73-79).  `out` is bit-identical to rbnn_sum_slabs'. */
int rbnn_sum_slabs_norms(const float *slabs, int32_t n_slabs, int32_t n_points, int32_t d_pad, int32_t in_features, float scale,
                         float *out, int32_t ldo, float *linf, float *l2, void *stream);

/* alpha[n] = 2 / max_d X0[n,d]   — adversarialAttacks.py:89 (per image, from the clean image). */
int rbnn_pgd_alpha(const float *X0, int32_t ldx, int32_t n_points, int32_t in_features,
                   float *alpha, void *stream);

/* One attack step on all points, g = sum_k G[k] (n_slabs partial gradients, slab stride in floats):
 *   project == 0 (FGSM):  X = clamp(X + step*sign(g), 0, 1)                      adversarialAttacks.py:81-82
 *   project == 1 (PGD):   X = clamp(X0 + clamp(X + step*sign(g) - X0, -eps, eps), 0, 1)   :103-105
 * step = alpha[n] if alpha != NULL else alpha_scalar.  X is updated in place (X0 untouched);
 * only columns d < D are touched. */
int rbnn_attack_step(float *X, const float *X0, int32_t ldx, const float *G, int32_t n_slabs,
                     size_t slab_stride, int32_t ldg, const float *alpha, float alpha_scalar,
                     float eps, int32_t project, int32_t n_points, int32_t in_features, void *stream);

/* attack_evaluation's reductions, adversarialAttacks.py:179,186 (argmax == label counts) and
 * :30-51,60 (softmax applied again to the outputs, 1 - Linf difference).
 * out_orig/out_adv: [N,ldp] forward outputs (mean probs, or logits for NN/ensemble).
 * counts: int32[2] = {#correct original, #correct adversarial}, zeroed by this call; rob: [N]. */
int rbnn_eval_metrics(const float *out_orig, const float *out_adv, int32_t ldp, const int32_t *labels,
                      int32_t n_points, int32_t n_classes, int32_t *counts, float *rob, void *stream);

/* out[(r/4), c, r%4] = W[r, c] for a row-major [rows, cols] matrix (rows % 4 == 0): the packed weight image that
 * rbnn_fc_input_grad reads (rows = S_total*H of W1 [cols = D_pad] or of Wm [cols = H]).  One-off layout
 * transform at posterior-load time; no counterpart in the reference. */
int rbnn_pack_rows4(const float *W, int64_t rows, int32_t cols, float *out, void *stream);

/* A power-of-two operand scale that lives in DEVICE memory, so that the split images of data-dependent operands (the
 * inputs, and the activations they bound) need no device->host round trip: value * scale = hi + lo, scale = 2^exp. */
typedef struct rbnn_dev_scale {
    uint32_t absmax_bits;          /* bit pattern of the bound the exponent was derived from                    */
    int32_t  exp;                  /* 14 - ceil(log2(bound)), clamped to [-100, 100]; 0 for a zero / non-finite bound */
    float    scale;                /* 2^exp                                                                     */
    float    inv_scale;            /* 2^-exp                                                                    */
} rbnn_dev_scale;

/* out[0] <- scale of the inputs:        bound0 = max(floor_abs, max_{r,c<cols} |X[r,c]|)
 * out[1] <- scale of what they bound:   bound1 = min(cap, mul * bound0 + add)
 * (fc2: the layer-1 activations, mul = max_h sum_d |W1[h,d]|, add = max|b1|, cap = 1 for sigmoid / tanh else +inf;
 * conv: the pooled conv1 activations).  floor_abs = 1 when later iterates are clamp(., 0, 1) of something
 * (adversarialAttacks.py:105), else 0.  No counterpart in the reference (it computes in fp32).  Asynchronous on `stream`. */
int rbnn_input_scales(const float *X, int64_t rows, int32_t cols, int32_t ld, float floor_abs, float mul, float add,
                      float cap, rbnn_dev_scale *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * `conv` architecture (model_nn.py:93-106): Conv2d(Cin,32,5) -> act -> MaxPool2d(2) -> Conv2d(32,Hc,5) -> act ->
 * MaxPool2d(2, stride 1) -> Flatten -> Linear(NP2*Hc, C), all four activations (:66-75).  state_dict keys: model.0.*
 * model.3.* model.7.* (SURVEY 8a row a1).  Two input geometries:
 *   1x28x28  mnist / fashion_mnist — the only inputs the reference's conv accepts (:95-96) and sizes its head for (:106:
 *            NP2 = 49); pinned by the reference-generated fixtures;
 *   3x32x32  CIFAR-shaped (BASELINE.json configs[4]): NP2 = 81, a BUILD-DEFINED head (Hc/16 * 3072 of :106 would be wrong,
 *            SURVEY 8a note) — parity unpinned, checked against the fp64 oracle only.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct rbnn_conv_posterior {
    int32_t activation;            /* rbnn_activation                                              */
    int32_t hidden;                /* Hc = conv2 output channels, multiple of 16                   */
    int32_t n_classes;             /* C <= 16                                                      */
    int32_t n_stored;              /* S_total                                                      */
    int32_t in_channels;           /* Cin: 1 or 3                                                  */
    int32_t in_width;              /* square input width: 28 (Cin = 1) or 32 (Cin = 3); X rows hold Cin*W*W floats */
    const float *K1w, *K1b;        /* model.0.weight [S_total,32,Cin,5,5], model.0.bias [S_total,32] */
    const float *K2w, *K2b;        /* model.3.weight [S_total,Hc,32,5,5], model.3.bias [S_total,Hc] */
    const float *Fw, *Fb;          /* model.7.weight [S_total,C,NP2*Hc], model.7.bias [S_total,C]  */
    const float *K2w_ci;           /* [S_total,32,Hc/16,25,16]: model.3.weight regrouped [ci][hc block][tap][hc%16] (backward only) */
} rbnn_conv_posterior;

typedef struct rbnn_conv_workspace {
    float   *P, *dZ;               /* [S,N,16]                                                     */
    float   *P1;                   /* pooled+activated conv1 output, dense fp32 [S,N,32,P1W,P1W] (P1W = 12 / 14), or the 24 KiB split image per
                                      (s,n); sized by rbnn_conv_workspace_query (whole 1-KiB pieces per point)               */
    uint8_t *st1;                  /* [S,N,32*P1W*P1W] pool-1 stash: argmax (bits 0-1) | pre-activation > 0 (bit 2) */
    float   *Q2;                   /* [S,N,Hc*NP2]    pooled+activated conv2 output (the Linear's input) */
    uint8_t *st2;                  /* [S,N,Hc*NP2]    pool-2 stash, same encoding                  */
    float   *G;                    /* [S,N,Cin*W*W]   per-sample input gradients (backward)        */
} rbnn_conv_workspace;

typedef struct rbnn_conv_workspace_sizes { size_t P, dZ, P1, st1, Q2, st2, G; } rbnn_conv_workspace_sizes;

int rbnn_conv_workspace_query(const rbnn_conv_posterior *net, int32_t n_points, int32_t n_samples,
                              rbnn_conv_workspace_sizes *out);

/* Stacked per-sample forward of the conv net: P[s] = softmax(NN_s(X)) (or logits) + the two pooling stashes.
 * Replaces the sample loop of BNN.forward (model_bnn.py:251-255) with NN.forward = model_nn.py:98-106,126-141. */
int rbnn_conv_forward(const rbnn_conv_posterior *net, const float *X, int32_t ldx, int32_t n_points,
                      const int32_t *sample_idx, int32_t n_samples, int32_t out_kind,
                      const rbnn_conv_workspace *ws, void *stream);

/* Per-sample input gradients of the conv net: G[s,n,:] = dL_s/dx_n for the dZ left in ws->dZ by rbnn_loss_dlogits
 * (sum them with rbnn_sum_slabs(G, S, N, Cin*W*W, ...)).  Replaces loss.backward() through model_nn.py:98-106. */
int rbnn_conv_input_grad(const rbnn_conv_posterior *net, const int32_t *sample_idx, int32_t n_samples,
                         int32_t n_points, const rbnn_conv_workspace *ws, void *stream);

/* rbnn_conv_forward in split-half precision (the technique of the fc split mode applied to conv2, 98 % of the MACs; 1x28x28
 * inputs, relu / leaky only — RBNN_ERR_UNSUPPORTED otherwise):
 * K2_rows = rbnn_split_rows image of model.3.weight regrouped [S_total*Hc, 25 taps * 32 ci] (K tap-major) holding W * 2^k2_exp;
 * the pooled conv1 activations are carried as value * 2^p1_exp = hi + lo (the caller bounds them: |P1| <= max_c(sum|K1w_c| *
 * max|x| + |K1b_c|)).  ws->P1 holds the 24 KiB split image per (sample, point) (rbnn_conv_workspace_query sizes it);
 * same outputs as rbnn_conv_forward, and rbnn_conv_input_grad follows it unchanged.  p1_dev_scale != NULL: record [1] of
 * rbnn_input_scales, read on the device instead of p1_exp. */
int rbnn_conv_forward_split(const rbnn_conv_posterior *net, const void *K2_rows, int32_t k2_exp, int32_t p1_exp,
                            const rbnn_dev_scale *p1_dev_scale, const float *X, int32_t ldx,
                            int32_t n_points, const int32_t *sample_idx, int32_t n_samples, int32_t out_kind,
                            const rbnn_conv_workspace *ws, void *stream);

/* rbnn_conv_forward with conv2 in the triple-split ("f16x6") mode (both geometries, all four activations): full-width fp32 operands as three
 * fp16 pieces, six exact product terms per product on the f16 matrix pipe, fp32 accumulation (see the triple-split section below).
 * K2_triple = rbnn_triple_rows image of model.3.weight regrouped [S_total*Hc, 25 taps * 32 ci] (K tap-major) holding W * 2^k2_exp;
 * the pooled conv1 activations (computed in fp32 by the exact conv1 kernel into ws->P1) are split on the fly, scaled by 2^p1_exp or
 * by record [1] of rbnn_input_scales (p1_dev_scale != NULL).  Same outputs as rbnn_conv_forward; rbnn_conv_input_grad follows it unchanged. */
int rbnn_conv_forward_triple(const rbnn_conv_posterior *net, const void *K2_triple, int32_t k2_exp, int32_t p1_exp,
                             const rbnn_dev_scale *p1_dev_scale, const float *X, int32_t ldx, int32_t n_points,
                             const int32_t *sample_idx, int32_t n_samples, int32_t out_kind,
                             const rbnn_conv_workspace *ws, void *stream);

/* rbnn_conv_input_grad with conv2^T in the triple-split mode (both geometries, all four activations), DENSE form: conv2^T as one GEMM per tap
 * over the conv2 OUTPUT positions (64 at 1x28x28: one pass; 100 at 3x32x32: two passes over 64 + 36 positions whose col2im contributions
 * meet in wave-private partial images) — T[tap][ci][pos] = sum_hc W[hc][ci][tap] * dO2[hc][pos], every MFMA useful — + a col2im, instead of
 * a gather over a zero-padded gradient image (the fp32 and split kernels' form: 36-39 % of its MFMAs multiply padding; the triple kernel of
 * that form, rbnn_conv_input_grad_triple, was removed in ABI 9).  K2_dense = triple image of model.3.weight * 2^k2_exp regrouped
 * [S_total, ceil(Hc/32) K steps, 25 taps, 2 tiles of 16 ci][3 pieces][4 chunks of 8 hc][16 ci][8 hc] halves (hc zero-padded to a multiple of
 * 32; a piece = 1 KiB, fragment-major: the kernel loads it straight into the MFMA's A registers), as rbnn_conv_weight_images writes it; fw_l1 = max_f sum_c |model.7.weight[c, f]| bounds the routed gradients (a per-(sample, point) power-of-two scale is derived from
 * it and max|dZ|).  Same G as rbnn_conv_input_grad. */
int rbnn_conv_input_grad_dense(const rbnn_conv_posterior *net, const void *K2_dense, int32_t k2_exp, float fw_l1,
                               const int32_t *sample_idx, int32_t n_samples, int32_t n_points,
                               const rbnn_conv_workspace *ws, void *stream);

/* Both triple images of model.3.weight from the fp32 stack K2w [n_samples][hidden][32 ci x 25 taps] (nn.Conv2d's order) in one launch, at the
 * scale 2^k2_exp: K2_rows = what rbnn_conv_forward_triple reads (rbnn_triple_rows_grouped of the tap-major regrouping), K2_dense = what
 * rbnn_conv_input_grad_dense reads; either may be NULL.  Bit-identical to building them through rbnn_triple_rows and permuted copies; a
 * redrawable SVI stack calls it after every draw (robustbnns_amd/conv.py::ConvStackedPosterior.redraw). */
int rbnn_conv_weight_images(const float *K2w, int32_t n_samples, int32_t hidden, int32_t k2_exp, void *K2_rows, void *K2_dense, void *stream);

/* rbnn_conv_input_grad with conv2^T in split-half precision.  K2_bwd = rbnn_split_rows image of model.3.weight regrouped
 * [S_total*32 ci, (Hc/16 chunks) * 13 tap pairs * 4 * 8]: element (ci; chunk, t, lg, j) = W[hc = 16*chunk + 8*(lg&1) + j, ci,
 * tap = 2t + (lg>>1)] * 2^k2_exp (0 for the padded 26th tap); fw_l1 = max_f sum_c |model.7.weight[c, f]| bounds the routed
 * gradients (a per-(sample, point) power-of-two scale is derived from it and max|dZ|).  Same G as rbnn_conv_input_grad. */
int rbnn_conv_input_grad_split(const rbnn_conv_posterior *net, const void *K2_bwd, int32_t k2_exp, float fw_l1,
                               const int32_t *sample_idx, int32_t n_samples, int32_t n_points,
                               const rbnn_conv_workspace *ws, void *stream);

/* W[s,i] = loc[i] + softplus(scale_raw[i]) * eps[s,i]   — the SVI guide's draw, model_bnn.py:124-130
 * (Normal(loc, softplus(scale)).rsample()).  PARITY UNPINNED: pyro-ppl 1.3.0 is not available;
 * eps is supplied by the caller.  out row stride ld_out >= n_elem (lets W1 be written D_pad-strided
 * by calling once per row block). */
int rbnn_svi_materialize(const float *loc, const float *scale_raw, const float *eps, int64_t n_elem,
                         int32_t n_samples, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Split-half precision mode ("f16x3") of the fc forward / input-gradient contractions.  No counterpart in the
 * reference (it computes in fp32): every fp32 operand v is carried as v * 2^e = hi + lo with hi, lo fp16, products
 * are hi*hi' + hi*lo' + lo*hi' on the f16 matrix pipe with fp32 accumulation (2^-22 per product; the parity tests
 * hold this mode to the same 1e-5 bar as the exact mode).  hidden % 128 == 0, <= 10 classes; arch fc and fc2, all four activations.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct rbnn_split_images {
    const void *W1_rows;           /* rbnn_split_rows image of W1 viewed as [S_total*H, D] rows: [S_total,H,ld_rows/8,2,8] halves */
    const void *W1_cols;           /* rbnn_split_cols image of W1 (backward B operand): [S_total,H/32,4,2,ld_cols,8] halves        */
    const void *W2_gen;            /* rbnn_split_w2gen image of W2 (backward dA generator): [S_total,H/16,64,8] halves             */
    int32_t ld_rows;               /* columns per row of W1_rows, multiple of 32, >= D                                            */
    int32_t ld_cols;               /* columns of W1_cols = in_stride (D_pad)                                                      */
    int32_t w1_exp;                /* W1_rows and W1_cols hold W1 * 2^w1_exp                                                      */
    int32_t w2_exp;                /* W2_gen holds W2 * 2^w2_exp                                                                  */
    /* fc2: */
    const void *Wm_rows;           /* rbnn_split_rows image of Wm viewed as [S_total*H, H] rows, holding Wm * 2^wm_exp (forward)   */
    const void *Wm_cols;           /* rbnn_split_cols image of Wm [S_total,H/32,4,2,H,8] (backward step 1); W2_gen is then built    */
                                   /* from the OUTPUT layer (model.5), w2_exp its exponent                                        */
    int32_t wm_exp;
    int32_t h1_exp;                /* layer-1 activations are carried as h * 2^h1_exp = hi + lo in ws->hid1; set per call from the  */
                                   /* bound max_h sum_d |W1[h,d]| * max|x| + max|b1|                                              */
} rbnn_split_images;

/* per-problem scratch of the split mode */
typedef struct rbnn_split_workspace {
    void  *X_split;                /* [N, ld_rows] split-rows image of the current inputs (caller fills it with rbnn_split_rows) */
    void  *dZ_gen;                 /* [S, N_pad, 64 B] dA-generator image of dZ, written by rbnn_fc_input_grad_split              */
    float *g_scale;                /* [N_pad] per-point 2^-e(n) of that image                                                     */
} rbnn_split_workspace;
typedef struct rbnn_split_workspace_sizes { size_t X_split, dZ_gen, g_scale; } rbnn_split_workspace_sizes;

int rbnn_split_workspace_query(const rbnn_posterior *net, const rbnn_split_images *sp, int32_t n_points,
                               int32_t n_samples, rbnn_split_workspace_sizes *out);


/* dst[r, g, 0, :] = fp16(v), dst[r, g, 1, :] = fp16(v - hi) for v = src[r, 8g..8g+7] * 2^scale_exp (0 past `cols`):
 * the split-rows image [rows, ld_dst/8, 2, 8] halves of a row-major fp32 matrix.  ld_dst % 32 == 0.
 * The caller picks scale_exp so that max|v| * 2^scale_exp <= 2^14; with dev_scale != NULL the scale is read from that device
 * record instead (rbnn_input_scales) and scale_exp is ignored. */
int rbnn_split_rows(const float *src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                    const rbnn_dev_scale *dev_scale, void *dst, int32_t ld_dst, void *stream);

/* Split-cols image of n_mats row-major [rows, ld_src] matrices (rows % 32 == 0):
 * dst[m, hb, lg, p, d, j] = (p ? lo : hi) of W[m, 32*hb + 16*(j>>2) + 4*lg + (j&3), d] * 2^scale_exp, 8 halves j per
 * 16-byte unit, ld_dst columns (% 16 == 0, zero past `cols`).  One-off at posterior load. */
int rbnn_split_cols(const float *W, int64_t n_mats, int32_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                    void *dst, int32_t ld_dst, void *stream);

/* dA-generator image of the output layer W2 [n_mats, C, H] (C <= 10): per 16 hidden units one 1-KiB tile holding,
 * per MFMA lane, the W2 side (hi, lo, hi) of the 30 K slots 10*p + c.  One-off at posterior load. */
int rbnn_split_w2gen(const float *W2, int32_t n_mats, int32_t n_classes, int32_t hidden, int32_t scale_exp, void *dst,
                     void *stream);

/* rbnn_fc_forward in split precision: same outputs (ws->P, ws->mask1 / ws->dact1), inputs as split-rows images:
 * X_split = rbnn_split_rows(X, N, D, ldx_src, x_exp, ., ldx) with ldx == sp->ld_rows.  dev_scales != NULL: the two records
 * of rbnn_input_scales — [0] replaces x_exp, [1] replaces sp->h1_exp (fc2) — read on the device. */
int rbnn_fc_forward_split(const rbnn_posterior *net, const rbnn_split_images *sp, const void *X_split, int32_t ldx,
                          int32_t x_exp, const rbnn_dev_scale *dev_scales, int32_t n_points, const int32_t *sample_idx,
                          int32_t n_samples, int32_t out_kind, const rbnn_workspace *ws, void *stream);

/* rbnn_fc_input_grad in split precision: same slabs (ws->slabs, already un-scaled), from ws->dZ and ws->mask1.
 * n_classes <= 10 (fc2: two steps through ws->dhid1).  Re-scales dZ per point (2^e(n), so vanishing gradients keep their 22 bits). */
int rbnn_fc_input_grad_split(const rbnn_posterior *net, const rbnn_split_images *sp, const int32_t *sample_idx,
                             int32_t n_samples, int32_t n_points, int32_t chunk, const rbnn_workspace *ws,
                             const rbnn_split_workspace *sws, int32_t *n_slabs_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Triple-split mode ("f16x6") of the fc forward / input-gradient contractions: fp32 operands at FULL width on the f16
 * matrix pipe.  No counterpart in the reference (it computes in fp32).  v * 2^e = p0 + p1 + p2 with p_i fp16 — exact,
 * bit for bit, for |v| >= 2^-15 * max|tensor| (3 x 11 significand bits cover fp32's 24; within 2^-39 * max|tensor| below
 * that) — and a*b = a0*b2 + a2*b0 + a1*b1 + a1*b0 + a0*b1 + a0*b0 (six f16 MFMAs, exact terms; the dropped terms are
 * <= 2^-32 |a*b|), accumulated in fp32: the only rounding left is the fp32 accumulation, as in rbnn_fc_forward / rbnn_fc_input_grad.
 * Architectures fc and fc2, hidden % 128 == 0, all four activations, <= 10 classes in the input gradient.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct rbnn_triple_images {
    const void *W1_rows;           /* rbnn_triple_rows_grouped image of W1 viewed as [S_total*H, D] rows: [S_total*H/16,ld_rows/32,3,16,32] halves */
    const void *W1_cols;           /* rbnn_triple_cols image of W1 (backward B operand): [S_total,H/32,4,3,ld_cols,8] halves         */
    const void *W2_gen;            /* rbnn_triple_w2gen image of the OUTPUT layer (backward dA generator): [S_total,H/16,2,64,8] halves */
    int32_t ld_rows;               /* columns per row of W1_rows, multiple of 32, >= D                                              */
    int32_t ld_cols;               /* columns of W1_cols = in_stride (D_pad)                                                        */
    int32_t w1_exp;                /* W1_rows and W1_cols hold W1 * 2^w1_exp                                                        */
    int32_t w2_exp;                /* W2_gen holds the output layer * 2^w2_exp                                                      */
    /* fc2: */
    const void *Wm_rows;           /* rbnn_triple_rows_grouped image of Wm viewed as [S_total*H, H] rows, holding Wm * 2^wm_exp (forward) */
    const void *Wm_cols;           /* rbnn_triple_cols image of Wm [S_total,H/32,4,3,H,8] (backward step 1)                         */
    int32_t wm_exp;
    int32_t h1_exp;                /* layer-1 activations are carried as h * 2^h1_exp = p0 + p1 + p2 in tws->hid_triple; set from the */
                                   /* bound max_h sum_d |W1[h,d]| * max|x| + max|b1| (or record [1] of rbnn_input_scales)           */
} rbnn_triple_images;

/* per-problem scratch of the triple mode */
typedef struct rbnn_triple_workspace {
    void  *X_triple;               /* [ceil16(N), ld_rows] grouped triple-rows image of the current inputs (rbnn_triple_rows_grouped) */
    void  *dZ_gen;                 /* [S, N_pad, 64 B] dA-generator image of dZ, written by rbnn_fc_input_grad_triple               */
    float *g_scale;                /* [N_pad] per-point 2^-e(n) of that image                                                       */
    void  *hid_triple;             /* fc2: the hidden activations x 2^h1_exp as FP32, [S, H/32 stages, ceil(N/16) groups, 16 points, 32 units] (ABI 9; until then three fp16 pieces); layer 2 splits at its operand read */
} rbnn_triple_workspace;
typedef struct rbnn_triple_workspace_sizes { size_t X_triple, dZ_gen, g_scale, hid_triple; } rbnn_triple_workspace_sizes;

int rbnn_triple_workspace_query(const rbnn_posterior *net, const rbnn_triple_images *tp, int32_t n_points,
                                int32_t n_samples, rbnn_triple_workspace_sizes *out);

/* dst[r, k, p, :] = piece p (p0 = fp16(v), p1 = fp16(v - p0), p2 = fp16(v - p0 - p1)) of v = src[r, 32k..32k+31] * 2^scale_exp
 * (0 past `cols`): the triple-rows image [rows, ld_dst/32, 3, 32] halves.  ld_dst % 32 == 0.  scale_exp / dev_scale as rbnn_split_rows. */
int rbnn_triple_rows(const float *src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                     const rbnn_dev_scale *dev_scale, void *dst, int32_t ld_dst, void *stream);

/* The same pieces in the GROUPED order the fc forward kernel reads (W1_rows, Wm_rows, X_triple): 16 consecutive rows share one 3-KiB block
 * per K stage, dst[r / 16, k, p, r % 16, :] — [rows/16, ld_dst/32, 3, 16, 32] halves.  A 16-row group then sits in memory exactly as in the
 * kernel's stage tile, so its three LDS-DMA pieces (1 KiB apart on both sides) share one address and one M0 write.  Rows past `rows` in the
 * last group are left untouched: size dst for ceil(rows / 16) * 16 rows.  Same arguments as rbnn_triple_rows. */
int rbnn_triple_rows_grouped(const float *src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                             const rbnn_dev_scale *dev_scale, void *dst, int32_t ld_dst, void *stream);

/* Triple-cols image of n_mats row-major [rows, ld_src] matrices (rows % 32 == 0):
 * dst[m, hb, lg, p, d, j] = piece p of W[m, 32*hb + 16*(j>>2) + 4*lg + (j&3), d] * 2^scale_exp.  One-off at posterior load. */
int rbnn_triple_cols(const float *W, int64_t n_mats, int32_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                     void *dst, int32_t ld_dst, void *stream);

/* dA-generator image of the output layer W2 [n_mats, C, H] (C <= 10): per 16 hidden units one 2-KiB tile = the W2 side of the
 * two chained generator MFMAs (K-slot plan in rbnn_triple.hip).  One-off at posterior load. */
int rbnn_triple_w2gen(const float *W2, int32_t n_mats, int32_t n_classes, int32_t hidden, int32_t scale_exp, void *dst,
                      void *stream);

/* rbnn_fc_forward on triple images: same outputs (ws->P, ws->mask1 / ws->dact1; fc2: ws->mask2 / ws->dact2).  tws->X_triple =
 * rbnn_triple_rows_grouped(X, N, D, ., x_exp, ., tp->ld_rows); dev_scales != NULL: the two records of rbnn_input_scales — [0] replaces
 * x_exp, [1] replaces tp->h1_exp (fc2) — read on the device. */
int rbnn_fc_forward_triple(const rbnn_posterior *net, const rbnn_triple_images *tp, const rbnn_triple_workspace *tws,
                           int32_t x_exp, const rbnn_dev_scale *dev_scales, int32_t n_points, const int32_t *sample_idx,
                           int32_t n_samples, int32_t out_kind, const rbnn_workspace *ws, void *stream);

/* rbnn_fc_input_grad on triple images: same slabs (ws->slabs, already un-scaled), from ws->dZ and ws->mask1 (fc2: two steps
 * through ws->dhid1, with ws->mask2).  Re-scales dZ per point (2^e(n)) like rbnn_fc_input_grad_split. */
int rbnn_fc_input_grad_triple(const rbnn_posterior *net, const rbnn_triple_images *tp, const int32_t *sample_idx,
                              int32_t n_samples, int32_t n_points, int32_t chunk, const rbnn_workspace *ws,
                              const rbnn_triple_workspace *tws, int32_t *n_slabs_out, void *stream);

/* The tail of a step between the two GEMM calls, fused (round 4): what rbnn_reduce_samples (scale 1) + rbnn_loss_dlogits + the dZ re-scaling
 * inside rbnn_fc_input_grad_triple compute, in ONE launch, bit for bit — P [n_samples][n_points][16] in, tws->dZ_gen / tws->g_scale out;
 * the fp32 dZ buffer is not written at all.  mode: RBNN_LOSS_MEAN_PROB (adversarialAttacks.py:74-78), RBNN_LOSS_PER_SAMPLE
 * (lossGradients.py:29-40) or RBNN_LOSS_MEAN_LOGIT (ensemble / NN); inv_S as rbnn_loss_dlogits.  Psum_out (nullable): sum_s P [n_points][ldo].
 * rbnn_fc_input_grad_triple called afterwards with ws->dZ == NULL takes the image as given (it skips its own re-scaling launch).
 * Single-GPU only: the sample-sharded step needs its all-reduce between the sum and the loss (AttackEngine._step_sharded). */
int rbnn_step_tail_triple(int32_t mode, const float *P, const int32_t *labels, int32_t n_samples, float inv_S, int32_t n_points,
                          int32_t n_classes, float *Psum_out, int32_t ldo, const rbnn_triple_workspace *tws, void *stream);

/* rbnn_attack_step + rbnn_triple_rows_grouped of the NEW iterate in one pass (bit-identical X and image): inside a PGD loop
 * (adversarialAttacks.py:95-105) the next iteration's forward then needs no image-builder launch.  dev_scale: record [0] of
 * rbnn_input_scales (computed once per attack with floor_abs = 1); X_triple: ceil16(n_points) x ld_rows grouped image. */
int rbnn_attack_step_triple(float *X, const float *X0, int32_t ldx, const float *G, int32_t n_slabs, size_t slab_stride, int32_t ldg,
                            const float *alpha, float alpha_scalar, float eps, int32_t project, int32_t n_points, int32_t in_features,
                            const rbnn_dev_scale *dev_scale, void *X_triple, int32_t ld_rows, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * The SVI guide's draw written IN PLACE into a stacked posterior and all of its weight images, one launch, no eps tensor.
 * Replaces the per-forward pyro.random_module(basenet, Normal(loc, softplus(scale)))() of model_bnn.py:121-136 called S times per
 * prediction (:222-232): W[s] = loc + sigma * eps(s) for every tensor of the net (sigma = softplus(raw scale)), eps = Box-Muller of
 * Philox4x32-10(counter = (element quad, tensor id, s or 0, draw_id), key = sample_keys[s] or key).  PARITY UNPINNED against
 * pyro-ppl 1.3.0's RNG stream (the package is not available to check it against).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct rbnn_svi_guide {    /* device pointers to the variational parameters, UNPADDED row-major as the param store holds them;     */
                                   /* *_scale = the STANDARD DEVIATION softplus(raw `<key>_scale`) (model_bnn.py:127), taken once per guide */
    const float *W1_loc, *W1_scale;   /* model.1.weight_loc, softplus(model.1.weight_scale)  [hidden, in_features]                   */
    const float *b1_loc, *b1_scale;   /* model.1.bias                  [hidden]                                                      */
    const float *Wm_loc, *Wm_scale;   /* fc2: model.3.weight           [hidden, hidden]                                              */
    const float *bm_loc, *bm_scale;   /* fc2: model.3.bias             [hidden]                                                      */
    const float *W2_loc, *W2_scale;   /* output layer weight           [n_classes, hidden]                                           */
    const float *b2_loc, *b2_scale;   /* output layer bias             [n_classes]                                                   */
    int32_t hidden;                   /* the net's own hidden size (<= net->hidden: 16 is stored padded to 32, padding stays zero)   */
    int32_t reserved;
} rbnn_svi_guide;

/* Writes samples [0, n_samples) of `net`: W1 b1 (Wm bm) W2 b2 — the pointers are const in rbnn_posterior because every other entry
 * point only reads them; this one writes through them — plus W1_pack4 / Wm_pack4 when non-NULL, plus, when `tp` is non-NULL, the
 * triple images W1_rows, W1_cols, W2_gen (Wm_rows, Wm_cols) at the scales tp->w1_exp / w2_exp / wm_exp, which the caller fixes per
 * guide from the bound |w| <= |loc| + RBNN_SVI_EPS_MAX * sigma.  sample_keys: device array of n_samples 64-bit keys (one
 * seed per sample: model_bnn.py:222-226) or NULL (all samples under `key`, the sample index in the counter).  Asynchronous on `stream`. */
#define RBNN_SVI_EPS_MAX 6.77f
int rbnn_svi_draw(const rbnn_posterior *net, const rbnn_triple_images *tp, const rbnn_svi_guide *guide, int32_t n_samples,
                  const uint64_t *sample_keys, uint64_t key, uint32_t draw_id, void *stream);
/* The same draw for a caller that is about to run the TRIPLE kernels only: W1 (Wm) are written as their triple images (12 B per weight), not
 * as the fp32 stack + its pack_rows4 image (8 B per weight: 40 % of rbnn_svi_draw's writes, read by no triple kernel); b1 (bm) W2 b2 and the W2
 * generator image as always.  Whoever needs the fp32 W1 / Wm later calls rbnn_svi_draw with the same (key, draw_id): the same weights
 * (robustbnns_amd.posterior.StackedPosterior.materialize).  tp must not be NULL. */
int rbnn_svi_draw_images(const rbnn_posterior *net, const rbnn_triple_images *tp, const rbnn_svi_guide *guide, int32_t n_samples,
                         const uint64_t *sample_keys, uint64_t key, uint32_t draw_id, void *stream);
/* 1 when rbnn_svi_draw covers this posterior (fc / fc2; W2 [n_classes, hidden] fits the 160 KB of LDS it is staged in; with triple
 * images: n_classes <= 10, hidden % 128 == 0) — the host falls back to rbnn_svi_materialize + a new stack otherwise. */
int rbnn_svi_draw_supported(const rbnn_posterior *net, int32_t with_triple_images);

/* The same draw for tensors of any shape (the conv architecture: model.0 / .3 / .7 weights and biases), written IN PLACE into the fp32
 * stack: out[s, e] = loc[e] + sigma[e] * eps, eps of element e = component e % 4 of the Philox block with counter
 * (e / 4, tensor_id, s — or 0 when sample_keys is given —, draw_id).  One launch for up to 8 tensors and all samples; images derived
 * from the stack (the regrouped / triple images of the conv2 weights) are rebuilt by their builders (robustbnns_amd/conv.py). */
typedef struct rbnn_svi_flat_tensor {
    const float *loc, *sigma;         /* [n_elem] variational mean and standard deviation softplus(raw scale) */
    float *out;                       /* [n_samples, out_sample_stride] destination, first n_elem of each row  */
    int64_t n_elem, out_sample_stride;
    int32_t tensor_id, reserved;
} rbnn_svi_flat_tensor;
int rbnn_svi_draw_flat(const rbnn_svi_flat_tensor *tensors, int32_t n_tensors, int32_t n_samples, const uint64_t *sample_keys, uint64_t key,
                       uint32_t draw_id, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Low-dimensional nets (in_features <= 16, n_classes <= 10; half-moons: 2 -> H -> 2) — arch fc: the WHOLE hot path in one launch.
 * One call = what a sequence of rbnn_fc_forward, rbnn_reduce_samples, rbnn_loss_dlogits, rbnn_fc_input_grad, rbnn_sum_slabs(_norms),
 * rbnn_pgd_alpha and `iters` x rbnn_attack_step computes, i.e. for every point the loop nest adversarialAttacks.py:118 -> :95 ->
 * model_bnn.py:251 (attack), lossGradients.py:20-40 (gradient) or model_bnn.py:243-258 (forward), in fp32 FMA arithmetic.
 *   op RBNN_LOWDIM_FORWARD   out[N, ldo]  = out_scale * sum_s (probabilities | logits per out_kind) of the samples
 *   op RBNN_LOWDIM_GRADIENT  out[N, ldo]  = out_scale * sum_s dL_s/dx for loss_mode (MEAN_PROB, PER_SAMPLE, MEAN_LOGIT; inv_S as
 *                            rbnn_loss_dlogits), linf / l2 (nullable): the per-point norms of that gradient (lossGradients.py:91-105)
 *   op RBNN_LOWDIM_ATTACK    out[N, ldo]  = the iterate after `iters` steps from X towards the eps-ball around X0 (project != 0) or one
 *                            FGSM step (project == 0); step size alpha[n], or 2 / max(X0[n]) when alpha_per_image
 *                            (adversarialAttacks.py:89), or alpha_scalar
 * P_scratch: [n_samples, n_points, 16] floats (required for MEAN_PROB gradients / attacks).  sample_idx as rbnn_fc_forward.
 * ------------------------------------------------------------------------------------------------------------ */
/* fc2 (round 4; the reference's half-moons grid, grid_search_halfMoons.py:159-169: 2 -> H -> H -> 2, 250 HMC samples): hidden in
 * {32, 64, 128, 256, 512}, in_stride == 16, Wm_pack4 required.  The H x H layer is a GEMM over the points of a sample (fp32 MFMA), so
 * samples spread over the CUs and the mean over samples couples all blocks between forward and backward: one call issues, back to
 * back on `stream`, per iteration: forward kernel, sum over samples, backward kernel, slab sum + step (4 launches; 2 for the
 * per-sample loss) — csrc/rbnn_lowdim.hip.  relu / leaky: the forward kernel leaves one sign bit per hidden unit and (sample, point)
 * and the backward kernel starts at dL/dlogits; sigmoid / tanh and the per-sample loss (which has no forward launch): the backward
 * kernel recomputes the forward.  Same arguments and results as for fc, except that P_scratch must hold rbnn_lowdim_scratch_bytes()
 * bytes (per-sample outputs, per-sample gradient slabs, the sum over samples, the sign bits), is always required, and for
 * RBNN_LOWDIM_ATTACK `out` must not alias X (the iterate is kept in `out`: iteration 0 reads X with row stride ldx, every later one reads
 * `out` with ITS row stride ldo — the two strides are independent, a compact out beside a padded X is legal; X0 always has X's stride). */
typedef enum rbnn_lowdim_op { RBNN_LOWDIM_FORWARD = 0, RBNN_LOWDIM_GRADIENT = 1, RBNN_LOWDIM_ATTACK = 2 } rbnn_lowdim_op;
int rbnn_lowdim_supported(const rbnn_posterior *net);      /* 1 when rbnn_lowdim_run covers this posterior */
size_t rbnn_lowdim_scratch_bytes(const rbnn_posterior *net, int32_t n_points, int32_t n_samples);   /* bytes of P_scratch (0: bad arguments) */
int rbnn_lowdim_run(const rbnn_posterior *net, int32_t op, int32_t loss_mode, int32_t out_kind, const float *X, const float *X0,
                    int32_t ldx, int32_t n_points, const int32_t *sample_idx, int32_t n_samples, const int32_t *labels, float inv_S,
                    float out_scale, float eps, const float *alpha, float alpha_scalar, int32_t alpha_per_image, int32_t project,
                    int32_t iters, float *P_scratch, float *out, int32_t ldo, float *linf, float *l2, void *stream);

/* rbnn_svi_draw + rbnn_lowdim_run in ONE launch (arch fc, small posteriors: every block generates the weights of all samples of the call into its
 * LDS cache from the guide — loc + sigma * eps(key or sample_keys[s], draw_id, tensor, sample, quad), rbnn_svi_draw's generator, the same
 * weights — instead of copying them from the stack).  The stack is NOT written: a caller that needs it afterwards runs rbnn_svi_draw with the
 * same (key, draw_id) (robustbnns_amd.posterior.StackedPosterior.materialize).  BASELINE config 1 (half-moons, SVI): a redraw + an FGSM pass
 * was two launches at the launch floor.  rbnn_lowdim_fused_draw_supported: 1 when the call's samples fit the kernel's 60 KB weight cache. */
int rbnn_lowdim_fused_draw_supported(const rbnn_posterior *net, int32_t n_points, int32_t n_samples);
int rbnn_lowdim_run_svi(const rbnn_posterior *net, const rbnn_svi_guide *guide, const uint64_t *sample_keys, uint64_t key, uint32_t draw_id,
                        int32_t op, int32_t loss_mode, int32_t out_kind, const float *X, const float *X0, int32_t ldx, int32_t n_points,
                        const int32_t *sample_idx, int32_t n_samples, const int32_t *labels, float inv_S, float out_scale, float eps,
                        const float *alpha, float alpha_scalar, int32_t alpha_per_image, int32_t project, int32_t iters, float *P_scratch,
                        float *out, int32_t ldo, float *linf, float *l2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROBUSTBNNS_HIP_H */
