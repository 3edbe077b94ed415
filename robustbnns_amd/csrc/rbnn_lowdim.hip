// rbnn_lowdim.hip — the whole hot path of a LOW-DIMENSIONAL fc net (half-moons: 2 -> H -> 2; in_features <= 16, classes <= 10) in ONE
// launch: forward over all samples -> mean over samples -> loss -> hand-rolled input gradient -> sum over samples -> sign / project /
// clamp, for ALL PGD iterations, with the iterate resident in registers.
//
// Why a second implementation: with D = 2 the "GEMMs" of rbnn_kernels.hip are 16-column MFMA tiles that are 7/8 zero padding, and a
// BASELINE C1 pass (N = 100, S = 10, H = 64: 0.5 MFLOP) is nothing but the launch floor of its 7 kernels (90 us per FGSM pass, 40 passes
// per PGD attack, one attack per cell of the reference's grid search, grid_search_halfMoons.py:133-169).  Here the nest
// adversarialAttacks.py:118 -> :95 -> model_bnn.py:251 becomes:
//
//   block  = PB points x SL sample lanes (PB * SL <= 256); thread (j, p) owns point p and samples j, j + SL, j + 2 SL, ...
//   pass 1   per owned sample: a_h = b1[h] + W1[h,:] . x, act, z_c += W2[c,h] act(a_h); softmax; P[s,n,:] -> scratch; partial sum
//            -> LDS -> every thread of the point adds the SL partials in lane order  (= rbnn_reduce_samples)
//   loss     dL/d(mean output) exactly as rbnn_loss_dlogits (double softmax for a BNN, adversarialAttacks.py:74-76)
//   pass 2   per owned sample: dZ_s from P[s,n,:]; a_h recomputed (cheaper than a stash: D + 1 FMAs); dA_h = act'(a_h) sum_c W2[c,h] dZ_c;
//            g += dA_h W1[h,:]                                                      (= rbnn_fc_input_grad, fp32 FMA instead of MFMA)
//            -> LDS -> summed in lane order by every thread of the point            (= rbnn_sum_slabs)
//   step     x <- clamp(x0 + clamp(x + alpha sign(g) - x0, -eps, eps), 0, 1)         (= rbnn_attack_step; every lane of a point keeps
//            an identical copy of x: same operands, same order)
//
// All samples of a point live in one block, so no grid-wide synchronisation exists and every block runs its T iterations on its own.
// Weights are read straight from the stacked posterior (lanes of a wave share the sample: one request per load); a whole posterior of
// this kind is tens of KB per sample and stays in L2.  Everything is fp32 FMA in a fixed order: results are deterministic.
#include "rbnn_common.hpp"
#include <algorithm>

namespace {

enum { OP_FORWARD = 0, OP_GRADIENT = 1, OP_ATTACK = 2 };

struct LowArgs {
    rbnn_posterior net;
    const float *X, *X0;           // start iterate and clean inputs [N, ldx] (X0 == X for a fresh attack)
    const int* sidx;
    const int* labels;
    const float* alpha;
    float *P, *out, *linf, *l2;
    int ldx, N, S, op, loss, out_kind, ldo, iters, project, alpha_per_image, SL, PB;
    int cache_stride;              // > 0: floats per sample of the LDS weight cache (all S samples of the call fit)
    float inv_S, out_scale, eps, alpha_scalar;
    // fused SVI draw (rbnn_lowdim_run_svi): the cache is GENERATED — loc + sigma * eps(key, draw, tensor, sample, quad), rbnn_svi_draw's generator —
    // instead of copied from the stack: a redraw + a pass is one launch (the stack itself is not written: the host materialises it on demand)
    rbnn_svi_guide g;
    const unsigned long long* sample_keys;
    unsigned long long key;
    uint32_t draw_id;
    int fused_draw;
};

template <int ACT> __device__ __forceinline__ float act_deriv(float a, float hv) {
    if (ACT == RBNN_ACT_RELU)  return a > 0.f ? 1.f : 0.f;
    if (ACT == RBNN_ACT_LEAKY) return a > 0.f ? 1.f : LEAKY_SLOPE;
    return act_grad_from_value<ACT>(hv);
}

// One sample's weights: straight from the stacked posterior, or from the block's LDS cache (compact: W1 [H][4 DQ], b1 [H], W2 [C][H], b2 [C]).
struct SampleW {
    const float *W1, *b1, *W2, *b2;
    int ldw;
};

template <int DQ> __device__ __forceinline__ SampleW sample_view(const LowArgs& a, const float* cache, int s) {
    const int H = a.net.hidden, C = a.net.n_classes;
    SampleW v;
    if (cache) {
        const float* const base = cache + (long long)s * a.cache_stride;
        v.W1 = base; v.ldw = 4 * DQ; v.b1 = base + H * 4 * DQ; v.W2 = v.b1 + H; v.b2 = v.W2 + C * H;
    } else {
        const int sw = a.sidx ? a.sidx[s] : s;
        v.W1 = a.net.W1 + (long long)sw * H * a.net.in_stride; v.ldw = a.net.in_stride; v.b1 = a.net.b1 + (long long)sw * H;
        v.W2 = a.net.W2 + (long long)sw * C * H; v.b2 = a.net.b2 + (long long)sw * C;
    }
    return v;
}

// logits of one sample at x
template <int ACT, int DQ, int CM>
__device__ __forceinline__ void forward_sample(const SampleW& w_, int H, int C, const float (&x)[4 * DQ], float (&z)[CM]) {
    const float* const W1 = w_.W1; const float* const b1 = w_.b1; const float* const W2 = w_.W2;
    const int ldw = w_.ldw;
#pragma unroll
    for (int c = 0; c < CM; ++c) z[c] = (c < C) ? w_.b2[c] : 0.f;
    for (int h = 0; h < H; h += 4) {
        const f32x4 bq = *(const f32x4*)(b1 + h);
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = bq[r];
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const f32x4 w = *(const f32x4*)(W1 + (long long)(h + r) * ldw + 4 * q);
                a = fmaf(w[0], x[4 * q], a); a = fmaf(w[1], x[4 * q + 1], a); a = fmaf(w[2], x[4 * q + 2], a); a = fmaf(w[3], x[4 * q + 3], a);
            }
            hv[r] = act_fwd<ACT>(a);
        }
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                const f32x4 w2 = *(const f32x4*)(W2 + (long long)c * H + h);
                z[c] = fmaf(w2[0], hv[0], z[c]); z[c] = fmaf(w2[1], hv[1], z[c]); z[c] = fmaf(w2[2], hv[2], z[c]); z[c] = fmaf(w2[3], hv[3], z[c]);
            }
        }
    }
}

// gx += sum_h act'(a_h) (sum_c W2[c,h] dz_c) W1[h,:]   (a_h recomputed)
template <int ACT, int DQ, int CM>
__device__ __forceinline__ void backward_sample(const SampleW& w_, int H, int C, const float (&x)[4 * DQ], const float (&dz)[CM],
                                                float (&gx)[4 * DQ]) {
    const float* const W1 = w_.W1; const float* const b1 = w_.b1; const float* const W2 = w_.W2;
    const int ldw = w_.ldw;
    for (int h = 0; h < H; h += 4) {
        const f32x4 bq = *(const f32x4*)(b1 + h);
        float da[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                const f32x4 w2 = *(const f32x4*)(W2 + (long long)c * H + h);
#pragma unroll
                for (int r = 0; r < 4; ++r) da[r] = fmaf(w2[r], dz[c], da[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 w[DQ];
            float a = bq[r];
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                w[q] = *(const f32x4*)(W1 + (long long)(h + r) * ldw + 4 * q);
                a = fmaf(w[q][0], x[4 * q], a); a = fmaf(w[q][1], x[4 * q + 1], a); a = fmaf(w[q][2], x[4 * q + 2], a); a = fmaf(w[q][3], x[4 * q + 3], a);
            }
            const float d = da[r] * act_deriv<ACT>(a, act_fwd<ACT>(a));
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                gx[4 * q] = fmaf(d, w[q][0], gx[4 * q]); gx[4 * q + 1] = fmaf(d, w[q][1], gx[4 * q + 1]);
                gx[4 * q + 2] = fmaf(d, w[q][2], gx[4 * q + 2]); gx[4 * q + 3] = fmaf(d, w[q][3], gx[4 * q + 3]);
            }
        }
    }
}

template <int CM> __device__ __forceinline__ void softmax_inplace(float (&v)[CM], int C) {
    float m = -INFINITY, den = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) m = fmaxf(m, v[c]);
#pragma unroll
    for (int c = 0; c < CM; ++c) { v[c] = (c < C) ? expf(v[c] - m) : 0.f; den += v[c]; }
#pragma unroll
    for (int c = 0; c < CM; ++c) v[c] = v[c] / den;
}

// dL/d(what the loss saw), as rbnn_loss_dlogits: softmax(t) - onehot(y), times inv_S
template <int CM> __device__ __forceinline__ void loss_grad(const float (&t)[CM], int C, int y, float inv_S, float (&g)[CM]) {
    ce_softmax_grad<CM>(t, C, y, inv_S, g);                          // the label class without its cancellation (rbnn_common.hpp)
}

template <int ACT, int DQ, int CM>
__global__ void __launch_bounds__(256) lowdim_kernel(const LowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [SL][PB][RW] reduction slots, then (cache_stride > 0) the weight cache
    constexpr int DW = 4 * DQ, RW = DW > CM ? DW : CM;
    const int t = threadIdx.x, PB = a.PB, SL = a.SL;
    const int H = a.net.hidden;
    const float* cache = nullptr;
    if (a.cache_stride > 0) {
        // Small posteriors (C1: 10 samples x 1.8 KB): every block first copies ALL the call's samples into LDS, compact, in one round of
        // independent loads — the per-hidden-unit loop then waits on ds_read latency, not on a chain of dependent L2 round trips (which
        // is all a 4-block launch has to hide them with).
        float* const cw = red + ((a.SL * a.PB * RW + 3) & ~3);
        const int Cn = a.net.n_classes, SS = a.cache_stride;
        // in 16-byte units: per sample H * DQ (W1 rows, their first 4 DQ columns), H / 4 (b1), C * H / 4 (W2), ceil(C / 4) (b2); eight
        // independent loads in flight per thread before the first LDS store
        const int u1 = H * DQ, u2 = u1 + H / 4, u3 = u2 + Cn * H / 4, US = SS / 4, total = a.S * US;
        for (int base = 0; base < total; base += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 256 + t;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (e < total) {
                    const int sl = e / US, r = e - sl * US;
                    const int sw = a.sidx ? a.sidx[sl] : sl;
                    if (a.fused_draw) {                                   // the same (tensor, quad) counters as svi_draw_kernel: identical weights
                        const unsigned long long key = a.sample_keys ? a.sample_keys[sw] : a.key;
                        const Rng rng = {(uint32_t)key, (uint32_t)(key >> 32), a.sample_keys ? 0u : (uint32_t)sw, a.draw_id};
                        const int H0 = a.g.hidden, D = a.net.in_features;
                        float w[4] = {0.f, 0.f, 0.f, 0.f};
                        if (r < u1) { if (r / DQ < H0 && 4 * (r % DQ) < D) draw_quad(rng, T_W1, a.g.W1_loc, a.g.W1_scale, r / DQ, r % DQ, D, w); }
                        else if (r < u2) { if (4 * (r - u1) < H0) draw_quad(rng, T_B1, a.g.b1_loc, a.g.b1_scale, 0, r - u1, H0, w); }
                        else if (r < u3) { const int i2 = r - u2, c = i2 / (H / 4), q = i2 - c * (H / 4); if (4 * q < H0) draw_quad(rng, T_W2, a.g.W2_loc, a.g.W2_scale, c, q, H0, w); }
                        else draw_quad(rng, T_B2, a.g.b2_loc, a.g.b2_scale, 0, r - u3, Cn, w);
                        v[u] = (f32x4){w[0], w[1], w[2], w[3]};
                    } else
                    if (r < u1) v[u] = *(const f32x4*)(a.net.W1 + ((long long)sw * H + r / DQ) * a.net.in_stride + 4 * (r % DQ));
                    else if (r < u2) v[u] = *(const f32x4*)(a.net.b1 + (long long)sw * H + 4 * (r - u1));
                    else if (r < u3) v[u] = *(const f32x4*)(a.net.W2 + (long long)sw * Cn * H + 4 * (r - u2));
                    else {
                        const int c0 = 4 * (r - u3);
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (c0 + k < Cn) v[u][k] = a.net.b2[(long long)sw * Cn + c0 + k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 256 + t;
                if (e < total) *(f32x4*)(cw + 4 * e) = v[u];
            }
        }
        cache = cw;
        __syncthreads();
    }
    const int j = t / PB, p = t - j * PB;
    const int n = blockIdx.x * PB + p;
    const bool live = j < SL && n < a.N;
    const int C = a.net.n_classes, D = a.net.in_features, S = a.S, N = a.N;
    float x[DW], x0[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) {
        x[d] = (live && d < D) ? a.X[(long long)n * a.ldx + d] : 0.f;
        x0[d] = (live && d < D && a.op == OP_ATTACK) ? a.X0[(long long)n * a.ldx + d] : x[d];
    }
    const int y = (live && a.labels) ? a.labels[n] : 0;
    float step = a.alpha_scalar;
    if (a.op == OP_ATTACK && live) {
        if (a.alpha) step = a.alpha[n];
        else if (a.alpha_per_image) {                                 // adversarialAttacks.py:89: alpha = 2 / image.max(), from the CLEAN image
            float m = -INFINITY;
#pragma unroll
            for (int d = 0; d < DW; ++d) if (d < D) m = fmaxf(m, x0[d]);
            step = 2.f / m;
        }
    }
    float* const slot = red + ((long long)j * PB + p) * RW;
    const float* const col = red + (long long)p * RW;                 // lane jj of this point: col + jj * PB * RW
    const int iters = a.op == OP_ATTACK ? a.iters : 1;
    const bool need_mean = a.op == OP_FORWARD || a.loss != RBNN_LOSS_PER_SAMPLE;
    const bool probs = a.op == OP_FORWARD ? a.out_kind == RBNN_OUT_PROBS : a.loss != RBNN_LOSS_MEAN_LOGIT;
    const bool single = S <= SL;                                      // every thread owns at most ONE sample: its output stays in registers
    float keep[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) keep[c] = 0.f;
    for (int it = 0; it < iters; ++it) {
        float tot[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) tot[c] = 0.f;
        if (need_mean) {                                              // ---- pass 1: every sample's output at x, and their sum
            float acc[CM];
#pragma unroll
            for (int c = 0; c < CM; ++c) acc[c] = 0.f;
            if (live)
                for (int s = j; s < S; s += SL) {
                    float z[CM];
                    forward_sample<ACT, DQ, CM>(sample_view<DQ>(a, cache, s), H, C, x, z);
                    if (probs) softmax_inplace<CM>(z, C);
                    if (a.op != OP_FORWARD && a.loss == RBNN_LOSS_MEAN_PROB) {
                        if (single) {
#pragma unroll
                            for (int c = 0; c < CM; ++c) keep[c] = z[c];
                        } else {
                            float* const ps = a.P + ((long long)s * N + n) * RBNN_CPAD;
#pragma unroll
                            for (int c = 0; c < CM; ++c) if (c < C) ps[c] = z[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CM; ++c) acc[c] += z[c];
                }
            if (j < SL) {
#pragma unroll
                for (int c = 0; c < CM; ++c) slot[c] = acc[c];
            }
            __syncthreads();
            for (int jj = 0; jj < SL; ++jj) {
#pragma unroll
                for (int c = 0; c < CM; ++c) tot[c] += col[(long long)jj * PB * RW + c];
            }
            __syncthreads();
        }
        if (a.op == OP_FORWARD) {
            if (live && j == 0) {
#pragma unroll
                for (int c = 0; c < CM; ++c) if (c < C) a.out[(long long)n * a.ldo + c] = tot[c] * a.out_scale;
            }
            return;
        }
        float g[CM];                                                  // ---- loss (mean-prob / mean-logit: one gradient for all samples)
        if (need_mean) {
            float tm[CM];
#pragma unroll
            for (int c = 0; c < CM; ++c) tm[c] = tot[c] * a.inv_S;
            loss_grad<CM>(tm, C, y, a.inv_S, g);
        }
        float gx[DW];                                                 // ---- pass 2: input gradient of every owned sample
#pragma unroll
        for (int d = 0; d < DW; ++d) gx[d] = 0.f;
        if (live)
            for (int s = j; s < S; s += SL) {
                const SampleW sv = sample_view<DQ>(a, cache, s);
                float dz[CM];
                if (a.loss == RBNN_LOSS_MEAN_LOGIT) {
#pragma unroll
                    for (int c = 0; c < CM; ++c) dz[c] = g[c];
                } else {
                    float ps[CM];
                    if (a.loss == RBNN_LOSS_PER_SAMPLE) {             // lossGradients.py:29-40: CE of THIS sample's probabilities
                        forward_sample<ACT, DQ, CM>(sv, H, C, x, ps);
                        softmax_inplace<CM>(ps, C);
                        loss_grad<CM>(ps, C, y, a.inv_S, g);
                    } else if (single) {
#pragma unroll
                        for (int c = 0; c < CM; ++c) ps[c] = keep[c];
                    } else {
                        const float* const pp = a.P + ((long long)s * N + n) * RBNN_CPAD;
#pragma unroll
                        for (int c = 0; c < CM; ++c) ps[c] = (c < C) ? pp[c] : 0.f;
                    }
                    softmax_backward<CM>(g, ps, C, dz);               // (g - <g,p>) p, cancellation-free (rbnn_common.hpp)
                }
                backward_sample<ACT, DQ, CM>(sv, H, C, x, dz, gx);
            }
        if (j < SL) {
#pragma unroll
            for (int d = 0; d < DW; ++d) slot[d] = gx[d];
        }
        __syncthreads();
        float G[DW];
#pragma unroll
        for (int d = 0; d < DW; ++d) G[d] = 0.f;
        for (int jj = 0; jj < SL; ++jj) {
#pragma unroll
            for (int d = 0; d < DW; ++d) G[d] += col[(long long)jj * PB * RW + d];
        }
        __syncthreads();
        if (a.op == OP_GRADIENT) {
            if (live && j == 0) {
                float m = 0.f, ss = 0.f;
#pragma unroll
                for (int d = 0; d < DW; ++d) {
                    if (d < D) {
                        const float v = G[d] * a.out_scale;
                        a.out[(long long)n * a.ldo + d] = v;
                        m = fmaxf(m, fabsf(v)); ss = fmaf(v, v, ss);
                    }
                }
                if (a.linf) a.linf[n] = m;
                if (a.l2) a.l2[n] = sqrtf(ss);
            }
            return;
        }
#pragma unroll
        for (int d = 0; d < DW; ++d) {                                // ---- step: rbnn_attack_step's operation order
            const float sgn = (G[d] > 0.f) ? 1.f : ((G[d] < 0.f) ? -1.f : 0.f);
            float pert = x[d] + step * sgn;
            if (a.project) pert = x0[d] + fminf(fmaxf(pert - x0[d], -a.eps), a.eps);
            x[d] = (d < D) ? fminf(fmaxf(pert, 0.f), 1.f) : 0.f;
        }
    }
    if (live && j == 0) {
#pragma unroll
        for (int d = 0; d < DW; ++d) if (d < D) a.out[(long long)n * a.ldo + d] = x[d];
    }
}

template <int ACT, int DQ, int CM> int launch_low(const LowArgs& a, hipStream_t st) {
    constexpr int RW = (4 * DQ > CM) ? 4 * DQ : CM;
    LowArgs b = a;
    const size_t red_floats = ((size_t)a.SL * a.PB * RW + 3) & ~(size_t)3;
    const int H = a.net.hidden, C = a.net.n_classes;
    const size_t ss = (size_t)H * (4 * DQ + 1 + C) + 4 * (size_t)((C + 3) / 4);          // compact floats per sample (whole 16-byte units)
    // the cache pays when a block is short of work to hide L2 latency with (few owned samples per thread) and fits the default 64 KB of
    // dynamic LDS; big posteriors keep reading through L1 / L2, where many resident waves hide the latency
    b.cache_stride = ((red_floats + ss * a.S) * sizeof(float) <= 60 * 1024) ? (int)ss : 0;
    if (a.fused_draw && !b.cache_stride) return RBNN_ERR_UNSUPPORTED;      // (rbnn_lowdim_fused_draw_supported says so beforehand)
    const size_t lds = (red_floats + (b.cache_stride ? ss * a.S : 0)) * sizeof(float);
    hipLaunchKernelGGL((lowdim_kernel<ACT, DQ, CM>), dim3((unsigned)((a.N + a.PB - 1) / a.PB)), dim3(256), lds, st, b);
    return launch_status();
}

template <int ACT> int launch_low_act(const LowArgs& a, hipStream_t st) {
    const bool small_d = a.net.in_features <= 4, small_c = a.net.n_classes <= 2;
    if (small_d) return small_c ? launch_low<ACT, 1, 2>(a, st) : launch_low<ACT, 1, 10>(a, st);
    return small_c ? launch_low<ACT, 4, 2>(a, st) : launch_low<ACT, 4, 10>(a, st);
}


// =====================================================================================================
// fc2 on low-dimensional inputs (the reference's half-moons grid: 2 -> H -> H -> 2, H in {32, 128, 256, 512}, 250 HMC samples,
// grid_search_halfMoons.py:159-169).  The middle layer is H x H per sample — a real GEMM over the POINTS of a sample — so, unlike the fc
// nets above, a (point, sample) pair cannot live in one thread, and samples must spread over the CUs: one block = one sample x a group of
// <= 112 points.  The mean-probability loss couples all samples of a point (adversarialAttacks.py:74-76), i.e. all blocks, between
// the forward and the backward: with no grid-wide barrier inside a kernel, a pass is FOUR launches issued back to back by one C call
// (the generic path: 8 from Python) —
//   low2_kernel<BWD = false>   layer 1 (D <= 16: VALU) -> LDS; layer 2 on v_mfma_f32_16x16x4_f32 (A = the sample's Wm rows from L2, B = the
//                              hidden activations in LDS), activation on the accumulators; layer 3 as MFMAs that take those accumulators
//                              as their B operand; softmax                                                       -> P[s, n, :]
//   low2_reduce_kernel         sum over samples                                                                  -> Psum[n, :]  (or `out`)
//   low2_kernel<BWD = true>    the forward again (cheaper than stashing two hidden layers in HBM), loss, dZ; dA2 = act' . (W2^T dZ) as
//                              MFMAs -> LDS; dH1 = Wm^T dA2 on the MFMA from the pack_rows4 image of Wm (the same operand walk as
//                              layer 2); dA1 = act' . dH1; g = W1^T dA1 as MFMAs fed from the accumulators       -> slab[s, n, :]
//   low2_finish_kernel         sum of the S slabs in a fixed order, then the gradient + its norms, or the sign / project / clamp step
// — three (two kernels) for the per-sample loss of lossGradients.py:29-40.  Every sum has a fixed order: results are deterministic.
// =====================================================================================================
#ifndef RBNN_LOW2_STASH
#define RBNN_LOW2_STASH 1                                                  // 0: the backward launch always recomputes the forward (the form until round 4's second half)
#endif
struct Low2Args {
    rbnn_posterior net;
    const float* X;                // current iterate [N, ldx]
    const int* sidx;
    const int* labels;
    float* P;                      // [S][N][16] per-sample probabilities (logits for the mean-logit loss / OUT_LOGITS)
    const float* Psum;             // [N][16] sum over samples (BWD, mean-probability / mean-logit loss)
    float* slabs;                  // [S][N][16] per-sample input gradients (BWD)
    uint4* mask;                   // relu / leaky, a pass with a forward launch: [blocks][threads] sign bits of both hidden layers (low2_kernel<.., STASH>); else NULL
    int ldx, N, S, NG, loss, probs, dq;
    float inv_S;
};

template <int ACT> __device__ __forceinline__ float act_deriv_from_value(float hv) {
    if (ACT == RBNN_ACT_RELU)  return hv > 0.f ? 1.f : 0.f;            // h > 0 <=> a > 0 (relu: h = max(a, 0); leaky: h = a or slope * a)
    if (ACT == RBNN_ACT_LEAKY) return hv > 0.f ? 1.f : LEAKY_SLOPE;
    return act_grad_from_value<ACT>(hv);
}

template <int NW, int KTW, int NPTB, bool BWD> struct Low2Lds {
    static constexpr int H = 16 * NW * KTW, HS = H + 4, PT = 16 * NPTB;
    static constexpr int FLOATS = PT * HS * (BWD ? 2 : 1) + PT * 16 + NW * PT * 16 + (BWD ? PT * 16 : 0);
    static_assert(FLOATS * 4 <= 160 * 1024, "LDS");
};

// STASH (relu / leaky; round 4, second half): act' of both hidden layers is one BIT per (unit, point).  The forward launch of a mean-probability /
// mean-logit pass writes them (a.mask != NULL: 16 bytes per thread — bit (i NPTB + p) 4 + r of a lane = unit row0 + 16 i + 4 lg + r, point 16 p + li,
// the accumulator layout both backward products read act' in), and low2_kernel<BWD, STASH> starts at dZ: no layer 1, no H x H forward product, no
// softmax — the probabilities come from P.  (The per-sample loss has no forward launch, sigmoid / tanh need the activation values: they keep the
// recomputing backward.)
template <int ACT, int NW, int KTW, int NPTB, bool BWD, bool STASH = false>
__global__ void __launch_bounds__(64 * NW) low2_kernel(const Low2Args a) {
    static_assert(!STASH || (BWD && (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY) && KTW * NPTB * 4 <= 64), "the stashed backward: relu / leaky, 64 bits per layer and lane");
    using L = Low2Lds<NW, KTW, NPTB, BWD>;
    constexpr int H = L::H, HS = L::HS, PT = L::PT, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float sm2[];
    float* const bufA = sm2;                                              // h1 [PT][HS]: hidden activations of layer 1, point-major
    float* const bufB = bufA + PT * HS;                                   // BWD: dA2 [PT][HS]
    float* const xs = bufA + PT * HS * (BWD ? 2 : 1);                     // inputs [PT][16]
    float* const red = xs + PT * 16;                                      // cross-wave partials [NW][PT][16]
    float* const dzb = red + NW * PT * 16;                                // BWD: dZ [PT][16]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (sample, point group): blocks b, b + 8, b + 16, ... share an XCD (round-robin dispatch, rbnn_common.hpp), so the NG point
    // groups of a sample are spaced 8 apart: they run side by side on ONE XCD and share the sample's Wm (1 MB at hidden 512) in its L2
    // (consecutive block numbers put them on four different XCDs, each fetching its own copy: 4.4 TB/s from beyond the L2)
    const int bx = blockIdx.x & 7, bq = blockIdx.x >> 3;
    const int s = 8 * (bq / a.NG) + bx, n0 = (bq % a.NG) * PT;
    if (s >= a.S) return;                                                 // (whole block: the grid covers ceil(S / 8) * 8 samples)
    const int sw = a.sidx ? a.sidx[s] : s;
    const int C = a.net.n_classes, D = a.net.in_features, N = a.N;
    const float* const W1 = a.net.W1 + (long long)sw * H * 16;            // [H][16] (in_stride = 16: zero columns beyond D)
    const float* const b1 = a.net.b1 + (long long)sw * H;
    const float* const Wm = a.net.Wm + (long long)sw * H * H;
    const float* const bm = a.net.bm + (long long)sw * H;
    const float* const W2 = a.net.W2 + (long long)sw * C * H;
    const float* const b2 = a.net.b2 + (long long)sw * C;

    // Software pipeline: THREE register sets of A, each reloaded for K step j + 3 right behind the MFMAs of step j that read it (no copies;
    // an L2 / Infinity-Cache round trip is longer than the 4 KTW NPTB MFMAs of one step when one wave per SIMD is all the CU holds), B one
    // step ahead (LDS).  gemm_load issues the first three steps' A loads — callers place it BEFORE the barriers / LDS phases in front of
    // the product, so that those round trips overlap them.
    constexpr int J = H / 16;
    auto gemm_load = [&](auto&& ak, f32x4 (&A)[3][KTW]) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < KTW; ++i) A[u][i] = *(const f32x4*)ak(i, u < J ? u : 0);
    };
    auto gemm = [&](auto&& ak, const float* Bsm, f32x4 (&A)[3][KTW], f32x4 (&acc)[KTW][NPTB]) {
#pragma unroll
        for (int i = 0; i < KTW; ++i)
#pragma unroll
            for (int p = 0; p < NPTB; ++p) acc[i][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 b[NPTB], bn[NPTB];
#pragma unroll
        for (int p = 0; p < NPTB; ++p) b[p] = *(const f32x4*)(Bsm + (16 * p + li) * HS + 4 * lg);
        auto step = [&](f32x4 (&a)[KTW], int j) {
            if (j + 1 < J) {
#pragma unroll
                for (int p = 0; p < NPTB; ++p) bn[p] = *(const f32x4*)(Bsm + (16 * p + li) * HS + 16 * (j + 1) + 4 * lg);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < KTW; ++i)
#pragma unroll
                    for (int p = 0; p < NPTB; ++p) acc[i][p] = MFMA16(a[i][r], b[p][r], acc[i][p]);
            if (j + 3 < J) {
#pragma unroll
                for (int i = 0; i < KTW; ++i) a[i] = *(const f32x4*)ak(i, j + 3);
            }
#pragma unroll
            for (int p = 0; p < NPTB; ++p) b[p] = bn[p];
        };
        for (int j = 0; j < J; j += 3) {
            step(A[0], j);
            if (j + 1 < J) step(A[1], j + 1);
            if (j + 2 < J) step(A[2], j + 2);
        }
    };
    const int row0 = 16 * wave * KTW;                                     // this wave's first row (hidden unit) of either GEMM
    auto ak_fwd = [&](int i, int j) { return Wm + (long long)(row0 + 16 * i + li) * H + 16 * j + 4 * lg; };     // Wm rows, row-major
    const float* const Wmp = a.net.Wm_pack4 + (long long)sw * H * H;                                           // pack_rows4 image: Wm^T rows
    auto ak_bwd = [&](int i, int j) { return Wmp + ((long long)(4 * j + lg) * H + row0 + 16 * i + li) * 4; };
    f32x4 Areg[3][KTW];
    f32x4 acc[KTW][NPTB];
    const long long mslot = ((long long)blockIdx.x * NT + tid);           // this thread's 16 bytes of the sign-bit stash
    unsigned m1lo = 0, m1hi = 0, m2lo = 0, m2hi = 0;                       // layer-1 / layer-2 sign bits of this lane's (i, p, r) elements
    if constexpr (STASH) {
        const uint4 mk = a.mask[mslot];
        m1lo = mk.x; m1hi = mk.y; m2lo = mk.z; m2hi = mk.w;
        gemm_load(ak_bwd, Areg);                                          // the Wm^T product's first A tiles: under the dZ phase
        if (tid < PT) {
            const int n = n0 + tid;
            float z[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = (n < N) ? *(const f32x4*)(a.P + ((long long)s * N + n) * RBNN_CPAD + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
                z[4 * q] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
            }
            float g[16], e[16], dz[16];                                   // (as the recomputing backward below: rbnn_loss_dlogits on what the forward launch left in P)
            const int y = (n < N) ? a.labels[n] : 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) e[c] = (n < N && c < C) ? a.Psum[(long long)n * RBNN_CPAD + c] * a.inv_S : 0.f;
            ce_softmax_grad<16>(e, C, y, a.inv_S, g);
            if (a.loss != RBNN_LOSS_MEAN_LOGIT) softmax_backward<16>(g, z, C, dz);      // (g - <g,p>) p, cancellation-free (rbnn_common.hpp)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (a.loss == RBNN_LOSS_MEAN_LOGIT) dz[c] = g[c];
                if (!(n < N && c < C)) dz[c] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dzb + tid * 16 + 4 * q) = (f32x4){dz[4 * q], dz[4 * q + 1], dz[4 * q + 2], dz[4 * q + 3]};
        }
    } else {
    for (int i = tid; i < PT * 16; i += NT) {
        const int n = n0 + (i >> 4), d = i & 15;
        xs[i] = (n < N && d < D) ? a.X[(long long)n * a.ldx + d] : 0.f;
    }
    gemm_load(ak_fwd, Areg);                                              // lands under the input tile's round trip and layer 1
    __syncthreads();
    // ---- layer 1: h1[pt][h] = act(b1[h] + W1[h, :] . x[pt, :]).  A thread owns four hidden units — their W1 rows and biases are loaded ONCE
    // into registers — and walks over points (x from LDS, a broadcast): no global load inside the loop (first version: one (point, unit
    // quad) item per trip with its W1 rows re-read from L2 every trip — 14 dependent round trips, 14 of the kernel's 22 us at hidden 128) ----
    {
        constexpr int HQ = H / 4, PSTEP = NT / HQ;
        static_assert(NT % HQ == 0, "threads per unit quad");
        const int h = 4 * (tid % HQ);
        const f32x4 bq = *(const f32x4*)(b1 + h);
        f32x4 w[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[r][q] = (q < a.dq) ? *(const f32x4*)(W1 + (h + r) * 16 + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int pt = tid / HQ; pt < PT; pt += PSTEP) {
            f32x4 av = bq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < a.dq) {
                    const f32x4 xv = *(const f32x4*)(xs + pt * 16 + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        av[r] = fmaf(w[r][q][0], xv[0], av[r]); av[r] = fmaf(w[r][q][1], xv[1], av[r]);
                        av[r] = fmaf(w[r][q][2], xv[2], av[r]); av[r] = fmaf(w[r][q][3], xv[3], av[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = act_fwd<ACT>(av[r]);
            *(f32x4*)(bufA + pt * HS + h) = av;
        }
    }
    __syncthreads();
    // ---- H x H GEMM on the fp32 MFMA: acc[i][p] (rows 16 (wave KTW + i) .., points 16 p ..) = sum_k A[row][k] B[point][k].  A lane holds four
    // consecutive k of its row (one 16-byte load = four MFMA K steps), a B lane the same four k of its point (one ds_read_b128);
    // `ak` maps (row tile, 16-wide k step) to the lane's address: row-major Wm for the forward, the pack_rows4 image for Wm^T ----
    gemm(ak_fwd, bufA, Areg, acc);
    // acc[i][p][r] = a2[unit row0 + 16 i + 4 lg + r][point 16 p + li]: bias, activation (the value also carries act')
    f32x4 zacc[NPTB];
#pragma unroll
    for (int p = 0; p < NPTB; ++p) zacc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
        const int hrow = row0 + 16 * i + 4 * lg;
        const f32x4 bias = *(const f32x4*)(bm + hrow);
        f32x4 w2f = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (li < C) w2f = *(const f32x4*)(W2 + li * H + hrow);
#pragma unroll
        for (int p = 0; p < NPTB; ++p) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][p][r] = act_fwd<ACT>(acc[i][p][r] + bias[r]);
                zacc[p] = MFMA16(w2f[r], acc[i][p][r], zacc[p]);          // layer 3: the accumulator IS the B operand (K index = its row)
                if constexpr (!BWD && (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY) && KTW * NPTB * 4 <= 64) {
                    const int bit = (i * NPTB + p) * 4 + r;               // (a constant after unrolling)
                    if (acc[i][p][r] > 0.f) { if (bit < 32) m2lo |= 1u << (bit & 31); else m2hi |= 1u << (bit & 31); }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NPTB; ++p) *(f32x4*)(red + (wave * PT + 16 * p + li) * 16 + 4 * lg) = zacc[p];      // Z^T[c = 4 lg + r][point] -> [point][c]
    __syncthreads();
    if (tid < PT) {
        const int n = n0 + tid;
        float z[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = 0.f;
            if (c < C) {
                v = b2[c];
                for (int w = 0; w < NW; ++w) v += red[(w * PT + tid) * 16 + c];
            }
            z[c] = v;
        }
        if (a.probs) {
            float m = -INFINITY, den = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < C) m = fmaxf(m, z[c]);
#pragma unroll
            for (int c = 0; c < 16; ++c) { z[c] = (c < C) ? expf(z[c] - m) : 0.f; den += z[c]; }
#pragma unroll
            for (int c = 0; c < 16; ++c) z[c] = z[c] / den;
        }
        if (!BWD) {
            if (n < N) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(a.P + ((long long)s * N + n) * RBNN_CPAD + 4 * q) = (f32x4){z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
            }
        } else {
            // dL/dlogits of this sample, exactly as rbnn_loss_dlogits: the loss gradient g on what the loss saw, through this sample's softmax
            float g[16], e[16], dz[16];
            const int y = (n < N) ? a.labels[n] : 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) e[c] = (a.loss == RBNN_LOSS_PER_SAMPLE) ? z[c] : ((n < N && c < C) ? a.Psum[(long long)n * RBNN_CPAD + c] * a.inv_S : 0.f);
            ce_softmax_grad<16>(e, C, y, a.inv_S, g);
            if (a.loss != RBNN_LOSS_MEAN_LOGIT) softmax_backward<16>(g, z, C, dz);      // (g - <g,p>) p, cancellation-free (rbnn_common.hpp)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (a.loss == RBNN_LOSS_MEAN_LOGIT) dz[c] = g[c];
                if (!(n < N && c < C)) dz[c] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dzb + tid * 16 + 4 * q) = (f32x4){dz[4 * q], dz[4 * q + 1], dz[4 * q + 2], dz[4 * q + 3]};
        }
    }
    if constexpr (!BWD && (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY) && KTW * NPTB * 4 <= 64) {
        if (a.mask) {                                                     // the sign bits of h1 in the same (i, p, r) order, and both layers' words out
#pragma unroll
            for (int i = 0; i < KTW; ++i)
#pragma unroll
                for (int p = 0; p < NPTB; ++p) {
                    const f32x4 hv = *(const f32x4*)(bufA + (16 * p + li) * HS + row0 + 16 * i + 4 * lg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int bit = (i * NPTB + p) * 4 + r;
                        if (hv[r] > 0.f) { if (bit < 32) m1lo |= 1u << (bit & 31); else m1hi |= 1u << (bit & 31); }
                    }
                }
            a.mask[mslot] = make_uint4(m1lo, m1hi, m2lo, m2hi);
        }
    }
    }                                                                     // (the forward: everything between the stash read and here)
    auto stash_deriv = [&](unsigned lo, unsigned hi, int bit) -> float {  // act' from a sign bit (relu: 1 / 0, leaky: 1 / slope)
        const bool pos = ((bit < 32 ? lo : hi) >> (bit & 31)) & 1u;
        return pos ? 1.f : (ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE);
    };
    if constexpr (BWD) {
        if constexpr (!STASH) gemm_load(ak_bwd, Areg);                    // the second product's first A tiles: under the dZ / dA2 phases
        __syncthreads();
        // ---- dA2[unit][point] = act'(h2) * sum_c W2[c][unit] dZ[point][c]: K = classes (<= 10: three K steps), -> LDS point-major ----
        const int QC = (C + 3) / 4;
        float bz[NPTB][3];
#pragma unroll
        for (int p = 0; p < NPTB; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) bz[p][q] = dzb[(16 * p + li) * 16 + 4 * q + lg];
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
            float aw[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) aw[q] = (4 * q + lg < C) ? W2[(4 * q + lg) * H + row0 + 16 * i + li] : 0.f;
#pragma unroll
            for (int p = 0; p < NPTB; ++p) {
                f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; ++q) if (q < QC) t = MFMA16(aw[q], bz[p][q], t);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] *= STASH ? stash_deriv(m2lo, m2hi, (i * NPTB + p) * 4 + r) : act_deriv_from_value<ACT>(acc[i][p][r]);
                *(f32x4*)(bufB + (16 * p + li) * HS + row0 + 16 * i + 4 * lg) = t;
            }
        }
        __syncthreads();
        // ---- dH1 = Wm^T dA2: the same walk with A from the pack_rows4 image [H/4][H][4] (four consecutive K per 16-byte load) ----
        gemm(ak_bwd, bufB, Areg, acc);
        // ---- dA1 = act'(h1) * dH1;  g[d][point] = sum_h W1[h][d] dA1[h][point], again fed from the accumulators ----
        f32x4 gacc[NPTB];
#pragma unroll
        for (int p = 0; p < NPTB; ++p) gacc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
            const int hrow = row0 + 16 * i + 4 * lg;
            float w1f[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w1f[r] = W1[(hrow + r) * 16 + li];
#pragma unroll
            for (int p = 0; p < NPTB; ++p) {
                f32x4 hv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (!STASH) hv = *(const f32x4*)(bufA + (16 * p + li) * HS + hrow);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gacc[p] = MFMA16(w1f[r], acc[i][p][r] * (STASH ? stash_deriv(m1lo, m1hi, (i * NPTB + p) * 4 + r) : act_deriv_from_value<ACT>(hv[r])), gacc[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < NPTB; ++p) *(f32x4*)(red + (wave * PT + 16 * p + li) * 16 + 4 * lg) = gacc[p];  // (red's Z partials were consumed before the last barrier)
        __syncthreads();
        for (int i = tid; i < PT * 4; i += NT) {
            const int pt = i >> 2, q = i & 3, n = n0 + pt;
            if (n < N) {
                f32x4 v = *(const f32x4*)(red + pt * 16 + 4 * q);
                for (int w = 1; w < NW; ++w) v += *(const f32x4*)(red + (w * PT + pt) * 16 + 4 * q);
                *(f32x4*)(a.slabs + ((long long)s * N + n) * 16 + 4 * q) = v;
            }
        }
    }
}

// Sum over samples of a [S][N][16] buffer, one block per point: thread (column c, lane g of 16) adds the samples g, g + 16, ... (eight
// independent loads in flight), then the 16 partials of a column are added in lane order by one thread — a fixed order: deterministic.
// (First version: one thread per (point, column) walking all S samples — 63 dependent trips: 21 us of a 100-us pass at S = 250.)
__device__ __forceinline__ float sum_over_samples(const float* __restrict__ buf, int S, int N, int n, float* sh) {
    const int t = threadIdx.x, c = t & 15, g = t >> 4;
    const long long st = (long long)N * 16;
    const float* const p = buf + (long long)n * 16 + c;
    float acc = 0.f;
    int s = g;
    for (; s + 7 * 16 < S; s += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(s + 16 * u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < S; s += 16) acc += p[s * st];
    sh[g * 16 + c] = acc;
    __syncthreads();
    float tot = 0.f;
    if (t < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += sh[k * 16 + t];
    }
    return tot;                                                           // valid in threads 0..15 (column = thread)
}

__global__ void __launch_bounds__(256) low2_reduce_kernel(const float* __restrict__ P, int S, int N, int C, float scale, float* __restrict__ out, int ldo) {
    __shared__ float sh[256];
    const int n = blockIdx.x, c = threadIdx.x;
    const float tot = sum_over_samples(P, S, N, n, sh);
    if (c < 16 && c < ldo) out[(long long)n * ldo + c] = (c < C) ? tot * scale : 0.f;
}

struct Low2Finish {
    const float* slabs; const float* Xcur; const float* X0; const float* alpha;
    float *out, *linf, *l2;
    int S, N, D, ldx, ldcur, ldo, op, project, alpha_per_image;      // ldx: X0's row stride; ldcur: the current iterate's (X's at iteration 0, out's afterwards)
    float out_scale, eps, alpha_scalar;
};

// one block per point: sum of the S per-sample slabs (as above), then — threads 0..15, one per column — the gradient (+ norms by a 16-lane
// butterfly) or the sign / project / clamp step
__global__ void __launch_bounds__(256) low2_finish_kernel(const Low2Finish a) {
    __shared__ float sh[256];
    const int n = blockIdx.x, d = threadIdx.x;
    const float G = sum_over_samples(a.slabs, a.S, a.N, n, sh);
    if (d >= 16) return;
    if (a.op == OP_GRADIENT) {
        const float v = (d < a.D) ? G * a.out_scale : 0.f;
        if (d < a.D) a.out[(long long)n * a.ldo + d] = v;
        float m = fabsf(v), ss = v * v;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); ss += __shfl_xor(ss, o); }
        if (d == 0) {
            if (a.linf) a.linf[n] = m;
            if (a.l2) a.l2[n] = sqrtf(ss);
        }
        return;
    }
    // the step, in rbnn_attack_step's operation order (adversarialAttacks.py:81-82, :103-105)
    const float x0 = (d < a.D) ? a.X0[(long long)n * a.ldx + d] : -INFINITY;
    float step = a.alpha_scalar;
    if (a.alpha) step = a.alpha[n];
    else if (a.alpha_per_image) {                                         // 2 / max of the CLEAN image (:89)
        float m = x0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        step = 2.f / m;
    }
    if (d < a.D) {
        const float x = a.Xcur[(long long)n * a.ldcur + d];
        const float sgn = (G > 0.f) ? 1.f : ((G < 0.f) ? -1.f : 0.f);
        float pert = x + step * sgn;
        if (a.project) pert = x0 + fminf(fmaxf(pert - x0, -a.eps), a.eps);
        a.out[(long long)n * a.ldo + d] = fminf(fmaxf(pert, 0.f), 1.f);
    }
}

template <int ACT, int NW, int KTW, int NPTB, bool BWD> int launch_low2_cfg(Low2Args a, hipStream_t st) {
    using L = Low2Lds<NW, KTW, NPTB, BWD>;
    a.NG = (a.N + L::PT - 1) / L::PT;
    const dim3 grid((unsigned)(8LL * a.NG * ((a.S + 7) / 8)));
    constexpr bool CAN_STASH = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY) && KTW * NPTB * 4 <= 64;
    if constexpr (!CAN_STASH) a.mask = nullptr;
    if constexpr (BWD && CAN_STASH) {
        if (a.mask) {                                                     // the forward launch of this pass left the sign bits: start at dZ
            static unsigned long long attr_s = 0;
            if (L::FLOATS * 4 > 64 * 1024 && !ensure_dynamic_lds((const void*)low2_kernel<ACT, NW, KTW, NPTB, true, true>, L::FLOATS * 4, attr_s)) return RBNN_ERR_LAUNCH;
            hipLaunchKernelGGL((low2_kernel<ACT, NW, KTW, NPTB, true, true>), grid, dim3(64 * NW), L::FLOATS * 4, st, a);
            return launch_status();
        }
    }
    static unsigned long long attr = 0;
    if (L::FLOATS * 4 > 64 * 1024 && !ensure_dynamic_lds((const void*)low2_kernel<ACT, NW, KTW, NPTB, BWD>, L::FLOATS * 4, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL((low2_kernel<ACT, NW, KTW, NPTB, BWD>), grid, dim3(64 * NW), L::FLOATS * 4, st, a);
    return launch_status();
}

// tile plan per hidden size: waves x unit tiles per wave x point tiles per block (accumulators <= 16 tiles, the two hidden images <= 133 KB)
template <int ACT, bool BWD> int launch_low2_h(const Low2Args& a, hipStream_t st) {
    switch (a.net.hidden) {
        case 32:  return launch_low2_cfg<ACT, 2, 1, 7, BWD>(a, st);
        case 64:  return launch_low2_cfg<ACT, 4, 1, 7, BWD>(a, st);
        case 128: return launch_low2_cfg<ACT, 4, 2, 7, BWD>(a, st);
        case 256: return launch_low2_cfg<ACT, 4, 4, 4, BWD>(a, st);
        case 512: return launch_low2_cfg<ACT, 4, 8, 2, BWD>(a, st);
    }
    return RBNN_ERR_UNSUPPORTED;
}

template <bool BWD> int launch_low2(const Low2Args& a, hipStream_t st) {
    switch (a.net.activation) {
        case RBNN_ACT_RELU:  return launch_low2_h<RBNN_ACT_RELU, BWD>(a, st);
        case RBNN_ACT_LEAKY: return launch_low2_h<RBNN_ACT_LEAKY, BWD>(a, st);
        case RBNN_ACT_SIGM:  return launch_low2_h<RBNN_ACT_SIGM, BWD>(a, st);
        case RBNN_ACT_TANH:  return launch_low2_h<RBNN_ACT_TANH, BWD>(a, st);
    }
    return RBNN_ERR_UNSUPPORTED;
}

bool low2_hidden_ok(int H) { return H == 32 || H == 64 || H == 128 || H == 256 || H == 512; }

// the fc2 sequence of one rbnn_lowdim_run call (see the header of this section); scratch = [P | slabs | Psum]
int run_low2(const rbnn_posterior* net, int op, int loss, int out_kind, const float* X, const float* X0, int ldx, int N, const int* sidx, int S,
             const int* labels, float inv_S, float out_scale, float eps, const float* alpha, float alpha_scalar, int alpha_per_image, int project,
             int iters, float* scratch, float* out, int ldo, float* linf, float* l2, hipStream_t st) {
    if (!net->Wm || !net->bm || !net->Wm_pack4 || !scratch) return RBNN_ERR_NULL;
    if (!aligned16(net->Wm) || !aligned16(net->bm) || !aligned16(net->Wm_pack4) || !aligned16(scratch)) return RBNN_ERR_ALIGN;
    const long long SN = (long long)S * N * RBNN_CPAD;
    float* const P = scratch;
    float* const slabs = scratch + SN;
    float* const Psum = scratch + 2 * SN;
    uint4* const mask = (uint4*)(scratch + 2 * SN + (long long)N * RBNN_CPAD);      // (16-byte aligned: every part is a multiple of 16 floats)
    Low2Args a = {};
    a.net = *net; a.sidx = sidx; a.labels = labels; a.P = P; a.Psum = Psum; a.slabs = slabs; a.ldx = ldx; a.N = N; a.S = S; a.loss = loss;
    a.dq = net->in_features <= 4 ? 1 : (net->in_features + 3) / 4; a.inv_S = inv_S;
    const unsigned rgrid = (unsigned)N;                                   // low2_reduce_kernel / low2_finish_kernel: one block per point
    int rc;
    if (op == OP_FORWARD) {
        a.X = X; a.probs = out_kind == RBNN_OUT_PROBS;
        if ((rc = launch_low2<false>(a, st))) return rc;
        hipLaunchKernelGGL(low2_reduce_kernel, dim3(rgrid), dim3(256), 0, st, P, S, N, net->n_classes, out_scale, out, ldo);
        return launch_status();
    }
    a.probs = loss != RBNN_LOSS_MEAN_LOGIT;
    // relu / leaky with a forward launch in the pass: the forward leaves the sign bits of both hidden layers, the backward starts at dZ
    a.mask = (RBNN_LOW2_STASH && loss != RBNN_LOSS_PER_SAMPLE && (net->activation == RBNN_ACT_RELU || net->activation == RBNN_ACT_LEAKY)) ? mask : nullptr;
    const int T = op == OP_ATTACK ? iters : 1;
    for (int it = 0; it < T; ++it) {
        a.X = (it == 0) ? X : out;                                         // the iterate lives in `out` from the first step on,
        a.ldx = (it == 0) ? ldx : ldo;                                     // with out's row stride (a compact out beside a padded X is legal)
        if (loss != RBNN_LOSS_PER_SAMPLE) {
            if ((rc = launch_low2<false>(a, st))) return rc;
            hipLaunchKernelGGL(low2_reduce_kernel, dim3(rgrid), dim3(256), 0, st, P, S, N, net->n_classes, 1.f, Psum, RBNN_CPAD);
            if ((rc = launch_status())) return rc;
        }
        if ((rc = launch_low2<true>(a, st))) return rc;
        Low2Finish f = {};
        f.slabs = slabs; f.Xcur = a.X; f.X0 = X0 ? X0 : X; f.alpha = alpha; f.out = out; f.linf = linf; f.l2 = l2; f.S = S; f.N = N; f.D = net->in_features;
        f.ldx = ldx; f.ldcur = a.ldx; f.ldo = ldo; f.op = op; f.project = project; f.alpha_per_image = alpha_per_image; f.out_scale = out_scale; f.eps = eps;
        f.alpha_scalar = alpha_scalar;
        hipLaunchKernelGGL(low2_finish_kernel, dim3(rgrid), dim3(256), 0, st, f);
        if ((rc = launch_status())) return rc;
    }
    return RBNN_OK;
}

}  // namespace

extern "C" {

int rbnn_lowdim_supported(const rbnn_posterior* net) {
    if (!net || (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2)) return 0;
    if (!(net->in_features >= 1 && net->in_features <= 16 && net->n_classes >= 1 && net->n_classes <= 10 && net->hidden >= 32 &&
          (net->hidden & 31) == 0 && net->in_stride >= 16 && (net->in_stride & 15) == 0)) return 0;
    if (net->arch == RBNN_ARCH_FC2) return net->in_stride == 16 && low2_hidden_ok(net->hidden);      // the fc2 tile plans (launch_low2_h)
    return 1;
}

size_t rbnn_lowdim_scratch_bytes(const rbnn_posterior* net, int32_t n_points, int32_t n_samples) {
    if (!net || n_points < 1 || n_samples < 1) return 0;
    const size_t SN = (size_t)n_samples * n_points * RBNN_CPAD * sizeof(float);
    if (net->arch != RBNN_ARCH_FC2) return SN;
    // fc2: [P | slabs | Psum | sign-bit stash: 16 bytes per thread of every low2_kernel block; <= 256 threads per block of >= 32 points (launch_low2_h)]
    const size_t groups = ((size_t)n_points + 31) / 32, blocks = 8 * groups * (((size_t)n_samples + 7) / 8);
    return 2 * SN + (size_t)n_points * RBNN_CPAD * sizeof(float) + blocks * 256 * 16;
}

// sample lanes per point: enough threads to fill the chip (256 CUs x 2 x 256) when N is small, one lane per point when N is large
static void lowdim_plan(int n_points, int n_samples, int& SL, int& PB) {
    const long long want = (131072 + (long long)n_points - 1) / n_points;
    SL = (int)std::max(1LL, std::min<long long>(std::min<long long>(n_samples, 256), want));
    PB = 256 / SL;
}

int rbnn_lowdim_fused_draw_supported(const rbnn_posterior* net, int32_t n_points, int32_t n_samples) {
    if (!net || net->arch != RBNN_ARCH_FC || !rbnn_lowdim_supported(net) || n_points < 1 || n_samples < 1) return 0;
    int SL, PB;
    lowdim_plan(n_points, n_samples, SL, PB);
    const int DQ = net->in_features <= 4 ? 1 : 4, CM = net->n_classes <= 2 ? 2 : 10, RW = std::max(4 * DQ, CM);
    const size_t red_floats = ((size_t)SL * PB * RW + 3) & ~(size_t)3;
    const size_t ss = (size_t)net->hidden * (4 * DQ + 1 + net->n_classes) + 4 * (size_t)((net->n_classes + 3) / 4);
    return (red_floats + ss * n_samples) * sizeof(float) <= 60 * 1024;      // launch_low's weight cache holds every sample of the call
}

static int lowdim_run_impl(const rbnn_posterior* net, const rbnn_svi_guide* guide, const uint64_t* sample_keys, uint64_t key, uint32_t draw_id,
                           int32_t op, int32_t loss_mode, int32_t out_kind, const float* X, const float* X0, int32_t ldx,
                           int32_t n_points, const int32_t* sample_idx, int32_t n_samples, const int32_t* labels, float inv_S, float out_scale,
                           float eps, const float* alpha, float alpha_scalar, int32_t alpha_per_image, int32_t project, int32_t iters,
                           float* P_scratch, float* out, int32_t ldo, float* linf, float* l2, void* stream) {
    if (!net || !X || !out || !net->W1 || !net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (!rbnn_lowdim_supported(net)) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    if (op < OP_FORWARD || op > OP_ATTACK || n_points < 1 || n_samples < 1 || ldx < net->in_features) return RBNN_ERR_SHAPE;
    if (op == OP_FORWARD ? ldo < net->n_classes : ldo < net->in_features) return RBNN_ERR_SHAPE;
    if (op != OP_FORWARD) {
        if (!labels) return RBNN_ERR_NULL;
        if (loss_mode != RBNN_LOSS_MEAN_PROB && loss_mode != RBNN_LOSS_PER_SAMPLE && loss_mode != RBNN_LOSS_MEAN_LOGIT) return RBNN_ERR_UNSUPPORTED;
        if (loss_mode == RBNN_LOSS_MEAN_PROB && !P_scratch) return RBNN_ERR_NULL;
    }
    if (net->arch == RBNN_ARCH_FC2 && !P_scratch) return RBNN_ERR_NULL;
    if (op == OP_ATTACK && (iters < 1 || (project && !X0))) return RBNN_ERR_SHAPE;
    if (!aligned16(net->W1) || !aligned16(net->b1) || !aligned16(net->W2)) return RBNN_ERR_ALIGN;
    if (net->arch == RBNN_ARCH_FC2) {
        if (guide) return RBNN_ERR_UNSUPPORTED;
        if (op != OP_FORWARD && loss_mode != RBNN_LOSS_PER_SAMPLE && !P_scratch) return RBNN_ERR_NULL;
        if (op == OP_ATTACK && out == X) return RBNN_ERR_SHAPE;             // the iterate is updated in `out` while X is still read
        return run_low2(net, op, loss_mode, out_kind, X, X0, ldx, n_points, sample_idx, n_samples, labels, inv_S, out_scale, eps, alpha, alpha_scalar,
                        alpha_per_image, project, iters, P_scratch, out, ldo, linf, l2, (hipStream_t)stream);
    }
    LowArgs a = {};
    a.net = *net; a.X = X; a.X0 = X0 ? X0 : X; a.sidx = sample_idx; a.labels = labels; a.alpha = alpha; a.P = P_scratch; a.out = out;
    a.linf = linf; a.l2 = l2; a.ldx = ldx; a.N = n_points; a.S = n_samples; a.op = op; a.loss = loss_mode; a.out_kind = out_kind; a.ldo = ldo;
    a.iters = iters; a.project = project; a.alpha_per_image = alpha_per_image; a.inv_S = inv_S; a.out_scale = out_scale; a.eps = eps;
    a.alpha_scalar = alpha_scalar;
    lowdim_plan(n_points, n_samples, a.SL, a.PB);
    if (guide) {
        if (!guide->W1_loc || !guide->W1_scale || !guide->b1_loc || !guide->b1_scale || !guide->W2_loc || !guide->W2_scale || !guide->b2_loc || !guide->b2_scale)
            return RBNN_ERR_NULL;
        if (guide->hidden < 1 || guide->hidden > net->hidden || !rbnn_lowdim_fused_draw_supported(net, n_points, n_samples)) return RBNN_ERR_UNSUPPORTED;
        a.g = *guide; a.sample_keys = (const unsigned long long*)sample_keys; a.key = key; a.draw_id = draw_id; a.fused_draw = 1;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (net->activation) {
        case RBNN_ACT_RELU:  return launch_low_act<RBNN_ACT_RELU>(a, st);
        case RBNN_ACT_LEAKY: return launch_low_act<RBNN_ACT_LEAKY>(a, st);
        case RBNN_ACT_SIGM:  return launch_low_act<RBNN_ACT_SIGM>(a, st);
        case RBNN_ACT_TANH:  return launch_low_act<RBNN_ACT_TANH>(a, st);
    }
    return RBNN_ERR_UNSUPPORTED;
}

int rbnn_lowdim_run(const rbnn_posterior* net, int32_t op, int32_t loss_mode, int32_t out_kind, const float* X, const float* X0, int32_t ldx,
                    int32_t n_points, const int32_t* sample_idx, int32_t n_samples, const int32_t* labels, float inv_S, float out_scale,
                    float eps, const float* alpha, float alpha_scalar, int32_t alpha_per_image, int32_t project, int32_t iters,
                    float* P_scratch, float* out, int32_t ldo, float* linf, float* l2, void* stream) {
    return lowdim_run_impl(net, nullptr, nullptr, 0, 0, op, loss_mode, out_kind, X, X0, ldx, n_points, sample_idx, n_samples, labels, inv_S, out_scale, eps,
                           alpha, alpha_scalar, alpha_per_image, project, iters, P_scratch, out, ldo, linf, l2, stream);
}

int rbnn_lowdim_run_svi(const rbnn_posterior* net, const rbnn_svi_guide* guide, const uint64_t* sample_keys, uint64_t key, uint32_t draw_id,
                        int32_t op, int32_t loss_mode, int32_t out_kind, const float* X, const float* X0, int32_t ldx,
                        int32_t n_points, const int32_t* sample_idx, int32_t n_samples, const int32_t* labels, float inv_S, float out_scale,
                        float eps, const float* alpha, float alpha_scalar, int32_t alpha_per_image, int32_t project, int32_t iters,
                        float* P_scratch, float* out, int32_t ldo, float* linf, float* l2, void* stream) {
    if (!guide) return RBNN_ERR_NULL;
    return lowdim_run_impl(net, guide, sample_keys, key, draw_id, op, loss_mode, out_kind, X, X0, ldx, n_points, sample_idx, n_samples, labels, inv_S,
                           out_scale, eps, alpha, alpha_scalar, alpha_per_image, project, iters, P_scratch, out, ldo, linf, l2, stream);
}

}  // extern "C"
