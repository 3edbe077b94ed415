// rbnn_split.hip — the "f16x3" precision mode of the two big contractions (gfx950 / MI355X).
//
// rbnn_kernels.hip runs both GEMMs of the hot path on v_mfma_f32_16x16x4_f32: exact fp32 products, 157 TFLOP/s peak.
// The f16 matrix pipe is 16x faster (v_mfma_f32_16x16x32_f16: 16 cycles for 8x the MACs).  This file carries every fp32
// operand v as an error-compensated PAIR of halves,
//
//      v * 2^e  =  hi + lo,     hi = fp16(v * 2^e),   lo = fp16(v * 2^e - hi)            (22 significant bits)
//
// and forms each product as  hi*hi' + hi*lo' + lo*hi'  (three f16 MFMAs, products exact, fp32 accumulation; the
// dropped lo*lo' term is 2^-22 relative).  Per-product error 2^-22 = 2.4e-7 against fp32's 6e-8 — inside the 1e-5
// parity bar of the path (tests/test_hip_parity.py checks this mode against the same oracle as the exact mode) — at a
// 5.3x higher arithmetic ceiling.  The power-of-two scale e (chosen by the host from the operand's max magnitude, so
// that |v * 2^e| <= 2^14) keeps hi AND lo in fp16's normal range over ~5 decades of |v| and is divided out exactly
// in the epilogue.
//
// Operand images (same footprint as the fp32 matrices they replace):
//   "split rows"   [R][ld/8][2][8] halves: per row, per group of 8 consecutive columns, 16 B of hi then 16 B of lo —
//                  one lane's MFMA operand (8 K values) is one ds_read_b128 of hi and one of lo.
// Lane maps (v_mfma_f32_16x16x32_f16, wave64, li = lane & 15, lg = lane >> 4):
//   A operand  a[j] = A[i = li][k = 8*lg + j]    B operand  b[j] = B[k = 8*lg + j][j' = li]    acc[r] = D[4*lg + r][li]
#include "rbnn_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {

// ===================================================================================================
// fp32 rows -> split-rows image.  One thread per (row, group of 8 columns): 32-B store.
// ===================================================================================================
__global__ void split_rows_kernel(const float* __restrict__ src, long long rows, int cols, int ld_src, float scale,
                                  uint4* __restrict__ dst, int groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * groups) return;
    const long long r = i / groups;
    const int g = (int)(i % groups);
    const float* const p = src + r * ld_src + 8 * g;
    union { f16x8 v; uint4 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (8 * g + j < cols) ? p[j] * scale : 0.f;
        const _Float16 h = (_Float16)v;
        hi.v[j] = h;
        lo.v[j] = (_Float16)(v - (float)h);
    }
    dst[2 * i] = hi.u;
    dst[2 * i + 1] = lo.u;
}

// ===================================================================================================
// S1: stacked forward, split precision.  One block = one (256-point tile, sample) item, 8 waves:
//   acc[h][n] = sum_d W[s][h][d] * X[n][d]        (A operand = W rows, B operand = X rows; D = [h][n])
// K runs in stages of 32 columns: a stage tile is (BH + BN) rows of 128 B (hi/lo chunks of the 4 column groups),
// brought in by LDS-DMA in 1-KiB pieces of 8 rows into a linear LDS image.  Physical 16-B chunk of logical chunk c
// in row r is c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group (rows li = 0..15, same chunk) then cover
// 16 distinct 16-B slots of a 256-B bank row.  The swizzle is applied on the SOURCE address of the DMA.
// The epilogue (bias, activation, 1-bit stash, skinny H->C layer on the fp32 MFMA taking the accumulators as its
// B operand, softmax) is the exact-mode kernel's (rbnn_kernels.hip, fc_forward_kernel).
// ===================================================================================================
struct FwdSplitArgs {
    const char* X;  int ldx;  int N;                           // split-rows image of the inputs [N][ldx] (ldx elements, % 32 == 0)
    const char* W;  long long w_sample_bytes;  int ldw;  int KT;   // split-rows image of W1 [S_total][H][ldw]; KT = ldw / 32
    const float* b;  const float* W2;  const float* b2;  int C;  int H;
    const int* sidx;  int S;  int NT;  float out_scale;        // out_scale = 2^-(e_x + e_w)
    float* P;  uint32_t* mask;  float* dact;  int out_kind;
};

template <int ACT, int WH, int HTW, int WN, int NTW>
__global__ void __launch_bounds__(64 * WH * WN, 2) fc_forward_split_kernel(const FwdSplitArgs a) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int ROWB = 128;                                  // bytes per tile row: 32 columns x (hi + lo)
    constexpr int TILEB = (BH + BN) * ROWB;                    // bytes per LDS stage buffer
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);
    constexpr int NW = WH * WN;
    static_assert(NW % 2 == 0, "the DMA source swizzle assumes a wave's pieces all have the same parity");
    static_assert((BH / 8) % NW == 0 && (BN / 8) % NW == 0, "whole pieces per wave");
    static_assert(HTW % 2 == 0 && HTW <= 8, "a wave's h range is whole 32-bit mask words, at most 4");
    static_assert(WH * BN * 64 <= 2 * TILEB, "the Z^T reduction scratch aliases the tile buffers");
    constexpr int WP = BH / 8 / NW, XP = BN / 8 / NW;         // DMA pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * TILEB bytes (dynamic: above the 64-KB static limit)
    char* const ldsb = (char*)lds;
    float* const zred = lds;

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.S, id)) return;
    int ntile, s;                                              // 2-D blocked item order, see fc_forward_kernel
    {
        const int full = a.S / 8, per = 8 * a.NT;
        if (id < full * per) { ntile = (id % per) / 8; s = (id / per) * 8 + id % 8; }
        else { const int rem = id - full * per, cnt = a.S - full * 8; ntile = rem / cnt; s = full * 8 + rem % cnt; }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int wave_h = wave % WH, wave_n = wave / WH;
    const int sw = a.sidx ? a.sidx[s] : s;
    const char* const Ws = a.W + (long long)sw * a.w_sample_bytes;
    const int n0 = ntile * BN;
    const int HW = a.H >> 5;
    // DMA piece q = tile rows 8q..8q+7; lane p lands at row 8q + (p >> 3), physical chunk p & 7, so it fetches
    // logical chunk (p & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = (4*(q & 1) + (p >> 4)) & 7 and q = wave (mod NW, even).
    const int prow = lane >> 3;
    const int src_off = (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 16;
    // fragment read of row li (any 16-row tile): hi = logical chunk 2*lg, lo = 2*lg + 1 (= physical chunk ^ 1)
    const int foff = li * ROWB + (((2 * lg) ^ ((li >> 1) & 7)) * 16), foff_lo = foff ^ 16;

    f32x4 zacc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) zacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int hc0 = 0; hc0 < a.H; hc0 += BH) {
        f32x4 acc[HTW][NTW];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        auto stage = [&](int kt, int buf) {
            char* const Wt = ldsb + buf * TILEB;
            char* const Xt = Wt + BH * ROWB;
            const long long koff = (long long)kt * ROWB + src_off;
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int q = wave + NW * i;
                glds16((const float*)(Ws + (long long)(hc0 + 8 * q + prow) * a.ldw * 4 + koff), (float*)(Wt + q * 1024));
            }
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = wave + NW * i;
                const int n = min(n0 + 8 * q + prow, a.N - 1);  // rows past N repeat the last point; never stored
                glds16((const float*)(a.X + (long long)n * a.ldx * 4 + koff), (float*)(Xt + q * 1024));
            }
        };
        stage(0, 0);
        ring_wait_barrier<0>();
        for (int kt = 0; kt < a.KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < a.KT) stage(kt + 1, buf ^ 1);         // lands while this stage is multiplied
            const char* const Wt = ldsb + buf * TILEB + (wave_h * HTW) * 16 * ROWB;
            const char* const Xt = ldsb + buf * TILEB + BH * ROWB + (wave_n * NTW) * 16 * ROWB;
            f16x8 bh[NTW], bl[NTW], ah, al, ah_n, al_n;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                bh[nt] = *(const f16x8*)(Xt + nt * 16 * ROWB + foff);
                bl[nt] = *(const f16x8*)(Xt + nt * 16 * ROWB + foff_lo);
            }
            ah = *(const f16x8*)(Wt + foff);
            al = *(const f16x8*)(Wt + foff_lo);
            ah_n = ah; al_n = al;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                if (ht + 1 < HTW) {
                    ah_n = *(const f16x8*)(Wt + (ht + 1) * 16 * ROWB + foff);
                    al_n = *(const f16x8*)(Wt + (ht + 1) * 16 * ROWB + foff_lo);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = MFMA_H(al, bh[nt], acc[ht][nt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = MFMA_H(ah, bl[nt], acc[ht][nt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = MFMA_H(ah, bh[nt], acc[ht][nt]);
                ah = ah_n; al = al_n;
            }
            // pin the order: B fragments + A(0) first, then per h tile half its MFMAs, the next tile's two reads, the rest
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NTW + 2, 0);
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                __builtin_amdgcn_sched_group_barrier(0x008, NTW + NTW / 2, 0);
                if (ht + 1 < HTW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTW - (NTW + NTW / 2), 0);
            }
            ring_wait_barrier<0>();                            // stage kt+1 landed; everyone is done with stage kt
        }

        // ---- epilogue of this h chunk: scale, bias, activation, derivative stash, skinny output layer ----
        const int hw0 = hc0 + (wave_h * HTW) * 16;
        unsigned mine[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) mine[nt] = 0u;
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht) {
            const int hrow = hw0 + ht * 16 + 4 * lg;           // acc[ht][nt][r] is hidden unit hrow + r
            const f32x4 bias = *(const f32x4*)(a.b + (long long)sw * a.H + hrow);
            f32x4 w2f = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (li < a.C) w2f = *(const f32x4*)(a.W2 + ((long long)sw * a.C + li) * a.H + hrow);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = n0 + (wave_n * NTW + nt) * 16 + li;
                f32x4 v = acc[ht][nt] * a.out_scale + bias, hv;
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bits |= (v[r] > 0.f ? 1u : 0u) << r;
                    hv[r] = act_fwd<ACT>(v[r]);
                }
                if (BITMASK) {
                    unsigned part = bits << (16 * (ht & 1) + 4 * lg);
                    part |= __shfl_xor(part, 16);
                    part |= __shfl_xor(part, 32);
                    if (lg == (ht >> 1)) mine[nt] |= part;
                }
                if (!BITMASK && a.dact && n < a.N) {
                    f32x4 dv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv[r] = act_grad_from_value<ACT>(hv[r]);
                    *(f32x4*)(a.dact + ((long long)s * a.N + n) * a.H + hrow) = dv;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) zacc[nt] = MFMA16(w2f[r], hv[r], zacc[nt]);
            }
        }
        if (BITMASK && a.mask) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = n0 + (wave_n * NTW + nt) * 16 + li;
                if (lg < HTW / 2 && n < a.N) a.mask[((long long)s * HW + (hw0 >> 5) + lg) * mask_ld(a.N) + n] = mine[nt];
            }
        }
    }

    // Z^T partials of the WH waves that split h -> LDS -> one thread per point finishes the softmax.
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
        *(f32x4*)(zred + (wave_h * BN + (wave_n * NTW + nt) * 16 + li) * 16 + 4 * lg) = zacc[nt];
    __syncthreads();
    if (tid < BN) {
        const int n = n0 + tid;
        float z[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 sum = *(const f32x4*)(zred + tid * 16 + 4 * q);
#pragma unroll
            for (int w = 1; w < WH; ++w) sum += *(const f32x4*)(zred + (w * BN + tid) * 16 + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[4 * q + r] = sum[r];
        }
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < a.C) { z[c] += a.b2[(long long)sw * a.C + c]; m = fmaxf(m, z[c]); }
            else z[c] = 0.f;
        }
        if (a.out_kind == RBNN_OUT_PROBS) {
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < a.C) { z[c] = expf(z[c] - m); den += z[c]; }
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < a.C) z[c] = z[c] / den;
        }
        if (n < a.N) {
            float* const dst = a.P + ((long long)s * a.N + n) * RBNN_CPAD;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 4 * q) = (f32x4){z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
        }
    }
}

template <int ACT, int WH, int HTW, int WN, int NTW>
int launch_forward_split_cfg(FwdSplitArgs a, hipStream_t st) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int LDSB = 2 * (BH + BN) * 128;
    a.NT = (a.N + BN - 1) / BN;
    auto kern = fc_forward_split_kernel<ACT, WH, HTW, WN, NTW>;
    static bool attr_done = false;                              // per instantiation; idempotent, so a race is harmless
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess) return RBNN_ERR_LAUNCH;
        attr_done = true;
    }
    const int grid = grid_for_items((long long)a.NT * a.S);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WH * WN), LDSB, st, a);
    return launch_status();
}

template <int ACT>
int launch_forward_split_act(const FwdSplitArgs& a, hipStream_t st) {
    if (a.H % 256 == 0) return launch_forward_split_cfg<ACT, 2, 8, 4, 4>(a, st);   // 256 h x 256 n, 8 waves of 128 h x 64 n
    if (a.H % 128 == 0) return launch_forward_split_cfg<ACT, 1, 8, 4, 4>(a, st);   // 128 h x 256 n, 4 waves
    return RBNN_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

int rbnn_split_rows(const float* src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp, void* dst,
                    int32_t ld_dst, void* stream) {
    if (!src || !dst) return RBNN_ERR_NULL;
    if (rows < 1 || cols < 1 || ld_src < cols || ld_dst < cols || (ld_dst & 31)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const int groups = ld_dst / 8;
    const long long total = (long long)rows * groups;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (long long)rows, cols, ld_src, ldexpf(1.f, scale_exp), (uint4*)dst, groups);
    return launch_status();
}

int rbnn_fc_forward_split(const rbnn_posterior* net, const rbnn_split_images* sp, const void* X_split, int32_t ldx,
                          int32_t x_exp, int32_t N, const int32_t* sidx, int32_t S, int32_t out_kind,
                          const rbnn_workspace* ws, void* stream) {
    if (!net || !sp || !X_split || !ws || !ws->P || !sp->W1_rows) return RBNN_ERR_NULL;
    if (!net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    const int H = net->hidden, ld = sp->ld_rows;
    if (H < 128 || (H % 128) || ld < net->in_features || (ld & 31) || ldx != ld) return RBNN_ERR_SHAPE;
    if (net->n_classes < 1 || net->n_classes > RBNN_CPAD || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(X_split) || !aligned16(sp->W1_rows) || !aligned16(ws->P) || !aligned16(net->b1) || !aligned16(net->W2)) return RBNN_ERR_ALIGN;
    FwdSplitArgs a = {};
    a.X = (const char*)X_split; a.ldx = ldx; a.N = N;
    a.W = (const char*)sp->W1_rows; a.w_sample_bytes = (long long)H * ld * 4; a.ldw = ld; a.KT = ld / 32;
    a.b = net->b1; a.W2 = net->W2; a.b2 = net->b2; a.C = net->n_classes; a.H = H;
    a.sidx = sidx; a.S = S; a.out_scale = ldexpf(1.f, -(x_exp + sp->w1_exp));
    a.P = ws->P; a.mask = ws->mask1; a.dact = ws->dact1; a.out_kind = out_kind;
    hipStream_t st = (hipStream_t)stream;
    switch (net->activation) {
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_RELU:  return launch_forward_split_act<RBNN_ACT_RELU>(a, st);
#endif
        case RBNN_ACT_LEAKY: return launch_forward_split_act<RBNN_ACT_LEAKY>(a, st);
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_SIGM:  return launch_forward_split_act<RBNN_ACT_SIGM>(a, st);
        case RBNN_ACT_TANH:  return launch_forward_split_act<RBNN_ACT_TANH>(a, st);
#endif
    }
    return RBNN_ERR_UNSUPPORTED;
}

}  // extern "C"
