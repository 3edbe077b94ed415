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
#include <algorithm>


namespace {

// ===================================================================================================
// Device-resident operand scales (rbnn_input_scales): max |X| by a grid-stride pass (wave max by DPP shuffles, one
// atomicMax on the bit pattern per wave — non-negative floats order like unsigned ints), then one thread turns the
// bound(s) into exponents.  No host round trip: the consumers (split_rows_kernel, the forward kernels) read the record.
// ===================================================================================================
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ X, long long rows, int cols, int ld,
                                                      unsigned* __restrict__ out_bits) {
    const int c4 = cols >> 2;                                   // whole float4s per row (ld % 4 == 0 checked by the host)
    const long long total = rows * c4;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4;
        const int c = (int)(i % c4) * 4;
        const f32x4 v = *(const f32x4*)(X + r * ld + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) m = INFINITY;    // NaN inputs: non-finite bound -> exponent 0
    }
    const int tail = cols & 3;
    if (tail)
        for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x)
            for (int c = cols - tail; c < cols; ++c) {
                const float v = X[r * ld + c];
                m = (v != v) ? INFINITY : fmaxf(m, fabsf(v));
            }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];                                   // one atomic per BLOCK: thousands of same-address atomics serialise (73 us at 1.6 M floats)
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(out_bits, __float_as_uint(m));
    }
}

__device__ __forceinline__ void scale_record(rbnn_dev_scale* r, float bound) {
    int e = 0;
    if (bound > 0.f && bound < INFINITY) {
        int k;
        const float mant = frexpf(bound, &k);                   // bound = mant * 2^k, mant in [0.5, 1): ceil(log2 bound) = k, or k-1 for a power of two
        e = 14 - (mant == 0.5f ? k - 1 : k);
        e = max(-100, min(100, e));
    }
    r->absmax_bits = __float_as_uint(bound);
    r->exp = e;
    r->scale = ldexpf(1.f, e);
    r->inv_scale = ldexpf(1.f, -e);
}

__global__ void scale_finalize_kernel(rbnn_dev_scale* out, float floor_abs, float mul, float add, float cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float m = fmaxf(floor_abs, __uint_as_float(out[0].absmax_bits));
    scale_record(out, m);
    scale_record(out + 1, fminf(cap, mul * m + add));
}

// ===================================================================================================
// fp32 rows -> split-rows image.  One thread per (row, group of 8 columns): 32-B store.
// ===================================================================================================
__global__ void split_rows_kernel(const float* __restrict__ src, long long rows, int cols, int ld_src, float scale,
                                  const rbnn_dev_scale* __restrict__ ds, uint4* __restrict__ dst, int groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * groups) return;
    if (ds) scale = ds->scale;
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
// in row r is c ^ row_swz(r).  A ds_read_b128 is served in four 16-lane groups that are NOT contiguous —
// {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32 — i.e. rows {0-3,12-15} of chunk pair lg with rows {4-11} of
// chunk pair lg+1; row_swz makes each group cover 16 distinct 16-B slots of the 256-B bank row (measured: the plain
// (r >> 1) & 7 swizzle left 48 % conflict cycles).  The swizzle is applied on the SOURCE address of the DMA.
// The epilogue (bias, activation, 1-bit stash, skinny H->C layer on the fp32 MFMA taking the accumulators as its
// B operand, softmax) is the exact-mode kernel's (rbnn_kernels.hip, fc_forward_kernel).
// ===================================================================================================

struct FwdSplitArgs {
    const char* X;  int ldx;  int N;                           // split-rows image of the inputs [N][ldx] (ldx elements, % 32 == 0)
    long long x_sample_bytes;                                  // 0: shared inputs; fc2 layer 2: the per-sample hidden image [S][N][ldx]
    const char* W;  long long w_sample_bytes;  int ldw;  int KT;   // split-rows image of W1 [S_total][H][ldw]; KT = ldw / 32
    const float* b;  const float* W2;  const float* b2;  int C;  int H;
    const int* sidx;  int S;  int NT;  float out_scale;        // out_scale = 2^-(e_x + e_w)
    float* P;  uint32_t* mask;  float* dact;  int out_kind;
    char* hid;  float hid_scale;                               // !LAYER2 (fc2 layer 1): activations out as a split-rows image [S][N][H], value * hid_scale = hi + lo
    const rbnn_dev_scale* x_ds;                                // != NULL: out_scale *= x_ds->inv_scale (the X operand's scale lives on the device)
    const rbnn_dev_scale* hid_ds;                              // != NULL: hid_scale = hid_ds->scale
};

template <int ACT, int WH, int HTW, int WN, int NTW, bool LAYER2>
__global__ void __launch_bounds__(64 * WH * WN, WH * WN / 4) fc_forward_split_kernel(const FwdSplitArgs a) {   // one block per CU (LDS): WH*WN/4 waves per SIMD
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
    const float out_scale = a.x_ds ? a.out_scale * a.x_ds->inv_scale : a.out_scale;
    const float hid_scale = (!LAYER2 && a.hid_ds) ? a.hid_ds->scale : a.hid_scale;
    const char* const Ws = a.W + (long long)sw * a.w_sample_bytes;
    const char* const Xs = a.X + (long long)s * a.x_sample_bytes;
    const int n0 = ntile * BN;
    const int HW = a.H >> 5;
    // DMA piece q = tile rows 8q..8q+7; lane p lands at row 8q + (p >> 3), physical chunk p & 7, so it fetches
    // logical chunk (p & 7) ^ row_swz(row); row mod 16 = 8*(q & 1) + (p >> 3) and q = wave (mod NW, even).
    const int prow = lane >> 3;
    const int src_off = ((lane & 7) ^ row_swz(8 * (wave & 1) + prow)) * 16;
    // fragment read of row li (any 16-row tile): hi = logical chunk 2*lg, lo = 2*lg + 1 (= physical chunk ^ 1)
    const int foff = li * ROWB + (((2 * lg) ^ row_swz(li)) * 16), foff_lo = foff ^ 16;

    f32x4 zacc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) zacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int hc0 = 0; hc0 < a.H; hc0 += BH) {
        f32x4 acc[HTW][NTW];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // per-lane byte offsets from the block-uniform bases Ws / Xs fit 32 bits (the host checks H*ldw*4 and N*ldx*4 < 2^32):
        // a uniform 64-bit base + one 32-bit VGPR offset per piece instead of 64-bit per-lane addresses (which spilled)
        auto stage = [&](int kt, int buf) {
            char* const Wt = ldsb + buf * TILEB;
            char* const Xt = Wt + BH * ROWB;
            const unsigned koff = (unsigned)kt * ROWB + (unsigned)src_off;
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int q = wave + NW * i;
                glds16((const float*)(Ws + ((unsigned)(hc0 + 8 * q + prow) * (unsigned)a.ldw * 4u + koff)), (float*)(Wt + q * 1024));
            }
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = wave + NW * i;
                const int n = min(n0 + 8 * q + prow, a.N - 1);  // rows past N repeat the last point; never stored
                glds16((const float*)(Xs + ((unsigned)n * (unsigned)a.ldx * 4u + koff)), (float*)(Xt + q * 1024));
            }
        };
        stage(0, 0);
        ring_wait_barrier<0>();
        f16x8 bh[NTW], bl[NTW], ah, al, ah_n, al_n;
        for (int kt = 0; kt < a.KT; ++kt) {
            const int buf = kt & 1;
            if (!(RBNN_ABL & 1) && kt + 1 < a.KT) stage(kt + 1, buf ^ 1);   // lands while this stage is multiplied
            const char* const Wt = ldsb + buf * TILEB + (wave_h * HTW) * 16 * ROWB;
            const char* const Xt = ldsb + buf * TILEB + BH * ROWB + (wave_n * NTW) * 16 * ROWB;
            if (!(RBNN_ABL & 4) || kt == 0) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    bh[nt] = *(const f16x8*)(Xt + nt * 16 * ROWB + foff);
                    bl[nt] = *(const f16x8*)(Xt + nt * 16 * ROWB + foff_lo);
                }
                ah = *(const f16x8*)(Wt + foff);
                al = *(const f16x8*)(Wt + foff_lo);
                ah_n = ah; al_n = al;
            }
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                if (!(RBNN_ABL & 4) && ht + 1 < HTW) {
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
            if (!(RBNN_ABL & 4)) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NTW + 2, 0);
#pragma unroll
                for (int ht = 0; ht < HTW; ++ht) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NTW + NTW / 2, 0);
                    if (ht + 1 < HTW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTW - (NTW + NTW / 2), 0);
                }
            }
            if (!(RBNN_ABL & 2)) ring_wait_barrier<0>();       // stage kt+1 landed; everyone is done with stage kt
        }
        if (RBNN_ABL & 2) ring_wait_barrier<0>();

        if (RBNN_ABL & 8) {                                    // diagnostic: keep the accumulators live, skip the epilogue
            float t = 0.f;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) t += acc[ht][nt][0] + acc[ht][nt][1] + acc[ht][nt][2] + acc[ht][nt][3];
            if (t == 12345.678f) a.P[tid] = t;
            continue;
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
            if (LAYER2 && li < a.C) w2f = *(const f32x4*)(a.W2 + ((long long)sw * a.C + li) * a.H + hrow);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = n0 + (wave_n * NTW + nt) * 16 + li;
                f32x4 v = acc[ht][nt] * out_scale + bias, hv;
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
                if (LAYER2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) zacc[nt] = MFMA16(w2f[r], hv[r], zacc[nt]);
                }
                if (!LAYER2) {
                    // lanes lg (even) and lg+1 hold the two halves of a group of 8 units: swap so that the even lane owns the whole
                    // 16-byte hi block and the odd lane the whole lo block — one 16-byte store per lane instead of two 8-byte ones
                    union { _Float16 h[4]; unsigned long long u; unsigned w[2]; } hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = hv[r] * hid_scale;
                        hi.h[r] = (_Float16)x;
                        lo.h[r] = (_Float16)(x - (float)hi.h[r]);
                    }
                    const bool odd = lg & 1;
                    const unsigned s0 = odd ? hi.w[0] : lo.w[0], s1 = odd ? hi.w[1] : lo.w[1];
                    const unsigned r0 = __shfl_xor(s0, 16), r1 = __shfl_xor(s1, 16);
                    const uint4 blk = odd ? make_uint4(r0, r1, lo.w[0], lo.w[1]) : make_uint4(hi.w[0], hi.w[1], r0, r1);
                    if (n < a.N)
                        *(uint4*)(a.hid + (((long long)s * a.N + n) * a.H + (hrow & ~7)) * 4 + (odd ? 16 : 0)) = blk;
                }
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

    if (!LAYER2) return;
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

template <int ACT, int WH, int HTW, int WN, int NTW, bool LAYER2>
int launch_forward_split_cfg(FwdSplitArgs a, hipStream_t st) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int LDSB = 2 * (BH + BN) * 128;
    a.NT = (a.N + BN - 1) / BN;
    auto kern = fc_forward_split_kernel<ACT, WH, HTW, WN, NTW, LAYER2>;
    static unsigned long long attr_done = 0;                    // per instantiation, one bit per device
    if (!ensure_dynamic_lds((const void*)kern, LDSB, attr_done)) return RBNN_ERR_LAUNCH;
    const int grid = grid_for_items((long long)a.NT * a.S);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WH * WN), LDSB, st, a);
    return launch_status();
}

template <int ACT, bool LAYER2>
int launch_forward_split_act(const FwdSplitArgs& a, hipStream_t st) {
    if (a.H % 256 == 0) return launch_forward_split_cfg<ACT, 2, 8, 4, 4, LAYER2>(a, st);   // 256 h x 256 n, 8 waves of 128 h x 64 n
    if (a.H % 128 == 0) return launch_forward_split_cfg<ACT, 1, 8, 4, 4, LAYER2>(a, st);   // 128 h x 256 n, 4 waves
    return RBNN_ERR_UNSUPPORTED;
}

template <bool LAYER2>
int launch_forward_split(int act, const FwdSplitArgs& a, hipStream_t st) {
    switch (act) {
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_RELU:  return launch_forward_split_act<RBNN_ACT_RELU, LAYER2>(a, st);
#endif
        case RBNN_ACT_LEAKY: return launch_forward_split_act<RBNN_ACT_LEAKY, LAYER2>(a, st);
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_SIGM:  return launch_forward_split_act<RBNN_ACT_SIGM, LAYER2>(a, st);
        case RBNN_ACT_TANH:  return launch_forward_split_act<RBNN_ACT_TANH, LAYER2>(a, st);
#endif
    }
    return RBNN_ERR_UNSUPPORTED;
}


// ===================================================================================================
// Image builders of the backward.
// ===================================================================================================
// W1 "split cols" image, the backward's B operand: out[m][hb][lg][p][d][j] (p = 0 hi, 1 lo; 8 halves j) =
// split(W[m][32*hb + 16*(j>>2) + 4*lg + (j&3)][d] * scale): lane (d = li, lg) of the main MFMA reads its 8 K values
// (K slot 8*lg + j  <->  hidden unit 32*hb + 16*(j>>2) + 4*lg + (j&3), the order the dA generator leaves them in)
// as one 16-byte hi and one 16-byte lo read.  One thread per (m, hb, lg, d).
__global__ void split_cols_kernel(const float* __restrict__ W, long long n_mats, int rows, int cols, int ld_src, float scale,
                                  uint4* __restrict__ dst, int ld_dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HB = rows / 32;
    if (i >= n_mats * HB * 4 * ld_dst) return;
    const int d = (int)(i % ld_dst);
    const int lg = (int)((i / ld_dst) % 4);
    const long long mh = i / (4LL * ld_dst);                   // m * HB + hb
    const long long m = mh / HB;
    const int hb = (int)(mh % HB);
    union { f16x8 v; uint4 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int h = 32 * hb + 16 * (j >> 2) + 4 * lg + (j & 3);
        const float v = (d < cols) ? W[(m * rows + h) * ld_src + d] * scale : 0.f;
        const _Float16 x = (_Float16)v;
        hi.v[j] = x;
        lo.v[j] = (_Float16)(v - (float)x);
    }
    const long long base = ((mh * 4 + lg) * 2) * ld_dst;       // 16-byte units
    dst[base + d] = hi.u;
    dst[base + ld_dst + d] = lo.u;
}

// K-slot plan of the dA generator MFMA (one v_mfma_f32_16x16x32_f16 forms all three products of the C <= 10 classes):
//   slot sigma = 10*p + c (p = 0,1,2; c < 10):   W2 side  hi, lo, hi      dZ side  hi, hi, lo      slots 30, 31 zero.
// W2 generator image: out[m][t][lane][j] = W2 side of slot 8*lg + j for hidden unit 16*t + li (li = lane & 15,
// lg = lane >> 4): 1 KiB per 16-unit tile, each lane's A operand at lane*16.  One thread per (m, t, lane).
__global__ void split_w2gen_kernel(const float* __restrict__ W2, int n_mats, int C, int H, float scale, uint4* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_mats * (H / 16) * 64) return;
    const int lane = (int)(i & 63), li = lane & 15, lg = lane >> 4;
    const int t = (int)((i >> 6) % (H / 16));
    const long long m = (i >> 6) / (H / 16);
    union { f16x8 v; uint4 u; } o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int sg = 8 * lg + j, p = sg / 10, c = sg % 10;
        _Float16 r = (_Float16)0.f;
        if (sg < 30 && c < C) {
            const float v = W2[(m * C + c) * H + 16 * t + li] * scale;
            const _Float16 x = (_Float16)v;
            r = (p == 1) ? (_Float16)(v - (float)x) : x;
        }
        o.v[j] = r;
    }
    dst[i] = o.u;
}

// dZ generator image + per-point scale.  16 threads per point n < N_pad (sample-strided, LDS max-reduce): e(n) = 13 - ilogb(max_{s,c} |dZ[s][n][c]|),
// so max |dZ| * 2^e(n) lies in [2^13, 2^14); out[s][n][chunk ^ dz_swz(n)][j] = dZ side of slot 8*chunk + j of
// dZ[s][n][:] * 2^e(n) (a point is 64 B; the chunk swizzle makes the generator's ds_read_b128 lane groups — points
// {0-3,12-15} of chunk lg with points {4-11} of chunk lg+1 — conflict-free); gscale[n] = 2^-e(n).  Points n >= N get zeros.
__host__ __device__ __forceinline__ int dz_swz(long long n) { return (int)((0 - (n >> 2)) & 3); }

__global__ void __launch_bounds__(256) split_dz_kernel(const float* __restrict__ dZ, int S, int N, long long N_pad, int C,
                                                       uint4* __restrict__ dst, float* __restrict__ gscale) {
    // block = 16 consecutive points x 16 sample lanes: thread (p = tid & 15, q = tid >> 4) handles samples q, q+16, ...
    __shared__ float red[16][17];
    const int p = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long n = (long long)blockIdx.x * 16 + p;
    float m = 0.f;
    if (n < N)
        for (int s = q; s < S; s += 16) {
            const float* const src = dZ + ((long long)s * N + n) * RBNN_CPAD;
#pragma unroll
            for (int k = 0; k < 3; ++k) {                       // classes 0..11 cover C <= 10
                const f32x4 v = *(const f32x4*)(src + 4 * k);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * k + r < C) m = fmaxf(m, fabsf(v[r]));
            }
        }
    red[q][p] = m;
    __syncthreads();
    m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i][p]);
    int e = 0;
    if (m > 0.f && m < INFINITY) e = min(13 - ilogbf(m), 120);
    if (q == 0) gscale[n] = ldexpf(1.f, -e);
    const int sw = dz_swz(n);
    for (int s = q; s < S; s += 16) {
        _Float16 hi[10], lo[10];
        float v[12];
        if (n < N) {
            const float* const src = dZ + ((long long)s * N + n) * RBNN_CPAD;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f32x4 t = *(const f32x4*)(src + 4 * k);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * k + r] = t[r];
            }
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const float x = (n < N && c < C) ? ldexpf(v[c], e) : 0.f;
            hi[c] = (_Float16)x;
            lo[c] = (_Float16)(x - (float)hi[c]);
        }
        uint4* const o = dst + ((long long)s * N_pad + n) * 4;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            union { f16x8 v; uint4 u; } w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sg = 8 * ch + j, pp = sg / 10, c = sg % 10;
                w.v[j] = (sg >= 30) ? (_Float16)0.f : (pp == 2 ? lo[c] : hi[c]);
            }
            o[ch ^ sw] = w.u;
        }
    }
}

// ===================================================================================================
// S2: input gradient, split precision.  One block = one (256-point tile, TD*16-column group, chunk of samples) item,
// 4 waves of 64 points (NTW = 4 point tiles each):
//   acc[n][d] += sum_h dA[n][h] * W1[s][h][d],   dA[n][h] = act'(A_s[n][h]) * sum_c dZ[s][n][c] * W2[s][c][h]
// A stage is 32 hidden units of one sample = ONE K step of the f16 MFMA:
//   generator  (dA^T)[h][n] for the stage's two 16-unit tiles: one f16 MFMA per (tile, point tile) whose 32 K slots
//              hold the three split products of the <= 10 classes (slot plan above), fp32 result;
//   split      x act' (1-bit stash) x 2^GEN_Q on the VALU, then hi/lo halves: the accumulator layout
//              (register r of tile t on lane (n = li, lg) is unit 16*t + 4*lg + r) IS the A-operand layout of the
//              main MFMA with K slot 8*lg + 4*t + r — the map the split-cols image of W1 is built for;
//   main       3 f16 MFMAs per (point tile, column tile), B operand (W1 hi / lo) from the stage's LDS tile.
// Everything a stage needs arrives by LDS-DMA one stage ahead (W1 tile, W2 generator tiles, stash words); the next
// sample's dZ generator image (16 KiB per block) rides along, one 1-KiB piece per stage.
// ===================================================================================================
#define GEN_Q (-17)                                           // |generator| <= 16 * 2^14 * 2^14 = 2^32  ->  |dA| <= 2^15 < fp16 max

struct GradSplitArgs {
    const char* dzg;  long long n_pad;  const float* gscale;  const uint32_t* mask;
    const char* W1c;  int ldc;                                  // split-cols image, ldc columns
    const char* W2g;                                            // generator image [S_total][H/16][1 KiB]
    int H;  int HW;  const int* sidx;  int S;  int chunk;  int nchunks;
    int N;  int NT;  int ND;  int Dt;
    float* out;  int ldo;  float out_scale;                     // slabs [nchunks][N][ldo]; out_scale = 2^-(e_w2 + GEN_Q + e_w1)
    // fc2.  MODE 1 (step 1, one sample per block): out = [S][N][H] = act'(A1) * (dA2 . Wm), KEPT SCALED (x out_scale, no per-point
    // un-scaling): it is the fp32 source of step 2's A operand.  MODE 2 (step 2): A operand read from `amem` and split in registers.
    const uint32_t* omask;  int OHW;                            // MODE 1: stash of the layer below [S][H/32][N_pad]
    const float* amem;                                          // MODE 2: [S][N][H]
    const float* dact;  const float* odact;                     // sigmoid / tanh: act' as fp32 [S][N][H] (this layer / the layer below)
};
enum { GRAD_FC = 0, GRAD_FC2_STEP1 = 1, GRAD_FC2_STEP2 = 2 };

template <int ACT, int TD, int NTW, int NW, int MODE>
__global__ void __launch_bounds__(64 * NW, (TD * NTW > 32 ? 1 : 2)) fc_grad_split_kernel(const GradSplitArgs a) {   // wide column groups: one wave per SIMD, up to 512 registers
    constexpr bool GEN = MODE != GRAD_FC2_STEP2;               // dA generated from dZ, or read from memory
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);   // act' from the 1-bit stash, or an fp32 stream
    constexpr bool STREAM = !GEN || !BITMASK;                  // a per-lane fp32 operand (A itself, or act') is prefetched from memory
    constexpr int BM = NW * NTW * 16, LD = TD * 16;            // NW waves x NTW point tiles = 256 points per block
    static_assert(BM == 256, "the dZ image and the stash rows are laid out for 256-point blocks");
    constexpr int W1B = 8 * LD * 16;                           // bytes: [4 lg][2 hi/lo][LD columns][16 B]
    constexpr int NPIECE = W1B / 1024, PPW = (NPIECE + NW - 1) / NW;
    constexpr int BUFB = W1B + 2048 + 1024;                    // + 2 generator tiles + 256 stash words
    constexpr int DZB = BM * 64;                               // dZ generator image of the block's points, one sample
    static_assert(W1B % 1024 == 0, "whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * BUFB + 2 * DZB bytes
    char* const ldsb = (char*)lds;
    char* const dzl = ldsb + 2 * BUFB;

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.ND * a.nchunks, id)) return;
    int ntile, dg, ch;
    grad_item(id, a.NT, a.ND, STREAM, ntile, dg, ch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int nb = ntile * BM + wave * (NTW * 16);
    const int dc0 = dg * LD;
    const int Dp = a.Dt * 16;
    const int ntd = min(TD, a.Dt - dg * TD);                   // valid column tiles of this group (the last group may be partial)
    const int s_begin = ch * a.chunk, s_end = min(a.S, s_begin + a.chunk);
    const int HS = a.H / 32, nst = (s_end - s_begin) * HS;
    // dZ-image pieces of the NEXT sample ride on the stages hb >= 1 of a sample (16 pieces per sample).  Stage hb = 0 is
    // issued while the previous sample's last stage still reads the buffer they would land in.
    const int DZPS = (16 + HS - 2) / (HS - 1);

    // per-lane source offsets (bytes, relative to the stage's image base) of this wave's W1 pieces
    int goff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int f = (wave + NW * i) * 1024 + lane * 16, seg = f / (LD * 16), d = (f % (LD * 16)) >> 4;
        goff[i] = (seg * a.ldc + min(dc0 + d, a.ldc - 1)) * 16;   // columns past the image: any valid address, never stored
    }

    f32x4 acc[NTW][TD];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) acc[nt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto dz_issue = [&](int s, int piece, int dzbuf) {          // 16 points x 64 B of sample s -> dzl[dzbuf]
        glds16((const float*)(a.dzg + (((long long)s * a.n_pad + ntile * BM + piece * 16) * 64) + lane * 16),
               (float*)(dzl + dzbuf * DZB + piece * 1024));
    };
    auto stage_issue = [&](int st, int buf) {
        const int si = st / HS, hb = st % HS, s = s_begin + si;
        const int sw = a.sidx ? a.sidx[s] : s;
        const char* const Wb = a.W1c + ((long long)sw * HS + hb) * 8 * a.ldc * 16;
        char* const B = ldsb + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (wave + NW * i < NPIECE) glds16((const float*)(Wb + goff[i]), (float*)(B + (wave + NW * i) * 1024));
        if (GEN && wave >= NW - 2)                              // generator tiles 2*hb, 2*hb + 1 of this sample
            glds16((const float*)(a.W2g + (((long long)sw * (a.H / 16) + 2 * hb + (wave - (NW - 2))) * 1024) + lane * 16),
                   (float*)(B + W1B + (wave - (NW - 2)) * 1024));
        if (GEN && BITMASK && wave == NW - 3)                   // stash words [S][H/32][N_pad]: the block's 256 points = 1 KiB
            glds16((const float*)(a.mask + ((long long)s * a.HW + hb) * a.n_pad + ntile * BM + 4 * lane), (float*)(B + W1B + 2048));
        if (GEN && s + 1 < s_end) {                             // next sample's dZ image, spread over this sample's stages
            for (int j = 0; j < DZPS && hb >= 1; ++j) {
                const int piece = (hb - 1) * DZPS + j;
                if (piece < 16 && piece % NW == wave) dz_issue(s + 1, piece, (si + 1) & 1);
            }
        }
    };

    if (GEN) for (int piece = wave; piece < 16; piece += NW) dz_issue(s_begin, piece, 0);
    stage_issue(0, 0);
    // MODE 2: the A operand of stage st (this lane: point li of each tile, units 16t + 4lg + r of the stage's 32) is loaded from
    // memory one stage ahead, AFTER the next stage's LDS-DMA has been issued, so the barrier's vmcnt(0) covers both
    // (sigmoid / tanh in the generator modes: the same prefetch carries act' of the stage's units instead)
    f32x4 am[STREAM ? NTW : 1][2];
    auto load_a = [&](int st) {
        const int s = s_begin + st / HS, h0 = (st % HS) * 32;
        const float* const base = GEN ? a.dact : a.amem;
#pragma unroll
        for (int nt = 0; nt < (STREAM ? NTW : 0); ++nt) {
            const int n = min(nb + nt * 16 + li, a.N - 1);     // rows past N: any valid row, never stored
            const float* const src = base + ((long long)s * a.N + n) * a.H + h0 + 4 * lg;
            am[nt][0] = *(const f32x4*)src;
            am[nt][1] = *(const f32x4*)(src + 16);
        }
    };
    if (STREAM) load_a(0);
    ring_wait_barrier<0>();
    const float c_pos = ldexpf(1.f, GEN_Q), c_neg = (ACT == RBNN_ACT_RELU) ? 0.f : LEAKY_SLOPE * ldexpf(1.f, GEN_Q);
    f16x8 da_hi[NTW], da_lo[NTW];                              // A operand of the main MFMA: this wave's 4 point tiles, one stage
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1, dzbuf = (st / HS) & 1;
        f32x4 dm[(GEN && STREAM) ? NTW : 1][2];                // sigmoid / tanh: this stage's act' (copied before the next prefetch)
        if (GEN && STREAM) {
#pragma unroll
            for (int nt = 0; nt < ((GEN && STREAM) ? NTW : 0); ++nt) { dm[nt][0] = am[nt][0]; dm[nt][1] = am[nt][1]; }
        }
        if (!GEN) {                                            // split this stage's A (already in registers), then fetch the next one
#pragma unroll
            for (int nt = 0; nt < (GEN ? 0 : NTW); ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = am[nt][0][r], v1 = am[nt][1][r];
                    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                    da_hi[nt][r] = h0;      da_lo[nt][r] = (_Float16)(v0 - (float)h0);
                    da_hi[nt][4 + r] = h1;  da_lo[nt][4 + r] = (_Float16)(v1 - (float)h1);
                }
        }
        if (!(RBNN_ABL & 1) && st + 1 < nst) stage_issue(st + 1, buf ^ 1);
        if (STREAM && st + 1 < nst) load_a(st + 1);
        const char* const B = ldsb + buf * BUFB;

        // ---- generator + split: da_hi / da_lo[nt] = A operand of the main MFMA for this wave's 4 point tiles ----
        if (GEN && (!(RBNN_ABL & 16) || st == 0)) {
            const f16x8 w2g0 = *(const f16x8*)(B + W1B + lane * 16);
            const f16x8 w2g1 = *(const f16x8*)(B + W1B + 1024 + lane * 16);
            const unsigned* const Mk = (const unsigned*)(B + W1B + 2048) + wave * (NTW * 16) + li;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const f16x8 dz = *(const f16x8*)(dzl + dzbuf * DZB + (wave * (NTW * 16) + nt * 16 + li) * 64 + ((lg ^ dz_swz(li)) * 16));
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 g0 = MFMA_H(w2g0, dz, z), g1 = MFMA_H(w2g1, dz, z);
                const unsigned mw = BITMASK ? Mk[nt * 16] >> (4 * lg) : 0u;    // bit r: unit 4*lg + r; bit 16 + r: unit 16 + 4*lg + r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = g0[r] * (BITMASK ? (((mw >> r) & 1u) ? c_pos : c_neg) : dm[BITMASK ? 0 : nt][0][r] * c_pos);
                    const float v1 = g1[r] * (BITMASK ? (((mw >> (16 + r)) & 1u) ? c_pos : c_neg) : dm[BITMASK ? 0 : nt][1][r] * c_pos);
                    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                    da_hi[nt][r] = h0;      da_lo[nt][r] = (_Float16)(v0 - (float)h0);
                    da_hi[nt][4 + r] = h1;  da_lo[nt][4 + r] = (_Float16)(v1 - (float)h1);
                }
            }
        }
        // ---- main: column-tile major, B operand double-buffered one column tile ahead ----
        const char* const Bw = B + (lg * 2 * LD + li) * 16;
        f16x8 bh = *(const f16x8*)(Bw), bl = *(const f16x8*)(Bw + LD * 16), bh_n = bh, bl_n = bl;
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) {
            if (dt < ntd) {                                     // block-uniform; the loop stays fully unrolled (acc in registers)
                if (dt + 1 < TD) {
                    bh_n = *(const f16x8*)(Bw + (dt + 1) * 256);
                    bl_n = *(const f16x8*)(Bw + LD * 16 + (dt + 1) * 256);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da_lo[nt], bh, acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da_hi[nt], bl, acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da_hi[nt], bh, acc[nt][dt]);
                bh = bh_n; bl = bl_n;
            }
        }
        if (!(RBNN_ABL & 2)) ring_wait_barrier<0>();           // next stage landed; everyone is done with this one
    }
    if (RBNN_ABL & 2) ring_wait_barrier<0>();

    // ---- epilogue: acc[nt][dt][r] = D[n = nb + nt*16 + 4*lg + r][d = dc0 + dt*16 + li], un-scaled per point ----
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nb + nt * 16 + 4 * lg + r;
            if (n >= a.N) continue;
            const float gs = (MODE == GRAD_FC2_STEP1) ? 1.f : a.gscale[n];
            float* const dst = a.out + ((long long)ch * a.N + n) * a.ldo;
#pragma unroll
            for (int dt = 0; dt < TD; ++dt) {
                const int d = dc0 + dt * 16 + li;
                if (d >= Dp) continue;
                float v = acc[nt][dt][r] * a.out_scale * gs;
                if (MODE == GRAD_FC2_STEP1) {                   // derivative of the layer below: unit d of point n, sample ch
                    if (BITMASK) {
                        const unsigned w = a.omask[((long long)ch * a.OHW + (d >> 5)) * a.n_pad + n];
                        v = ((w >> (d & 31)) & 1u) ? v : (ACT == RBNN_ACT_RELU ? 0.f : v * LEAKY_SLOPE);
                    } else {
                        v *= a.odact[((long long)ch * a.N + n) * a.ldo + d];
                    }
                }
                dst[d] = v;
            }
        }
}

template <int ACT, int TD, int NTW, int NW, int MODE>
int launch_grad_split_cfg(GradSplitArgs a, hipStream_t st) {
    constexpr int LDSB = 2 * (8 * TD * 16 * 16 + 3072) + 2 * 256 * 64;
    a.NT = (a.N + 255) / 256;
    a.ND = (a.Dt + TD - 1) / TD;
    auto kern = fc_grad_split_kernel<ACT, TD, NTW, NW, MODE>;
    static unsigned long long attr_done = 0;
    if (!ensure_dynamic_lds((const void*)kern, LDSB, attr_done)) return RBNN_ERR_LAUNCH;
    const int grid = grid_for_items((long long)a.NT * a.ND * a.nchunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDSB, st, a);
    return launch_status();
}

// Block shape.  Every block re-derives the split dA tiles of its points, so fewer, wider column groups would save
// generator work (ablation at C2: the generator + split is 0.6 of 2.4 ms, the LDS-DMA instructions 0.37 ms) — but the
// variants that do it need 8-wave blocks (one per CU) and measured SLOWER than two independent 4-wave blocks per CU:
// 8 waves x 2 point tiles x 14 column tiles 2.74 ms vs 2.50 ms; producer/consumer wave specialisation 3.08 vs 2.43 ms.
// Building stage st+1's dA tiles inside stage st's MFMA stream (in-wave software pipelining, dA registers double-buffered,
// slices fenced between groups of 12 MFMAs) was also slower, 2.73 vs 2.49 ms: a wave issues in order, and the dependent
// VALU chains stall the independent MFMAs queued behind them.  (profiles/r01f/ablation_split.txt)
// RBNN_GRAD_SPLIT_TD14 keeps the first of those selectable for experiments.
template <int ACT, int MODE>
int launch_grad_split(const GradSplitArgs& a, hipStream_t st) {
#ifdef RBNN_GRAD_SPLIT_TD14
    if (MODE == GRAD_FC && a.Dt > 7) return launch_grad_split_cfg<ACT, 14, 2, 8, MODE>(a, st);
#endif
#ifdef RBNN_GRAD_SPLIT_TDW                                      // experiment: 4 waves x 4 point tiles x TDW column tiles, one block per CU
    if (MODE == GRAD_FC && a.Dt > 7) return launch_grad_split_cfg<ACT, RBNN_GRAD_SPLIT_TDW, 4, 4, MODE>(a, st);
#endif
    // 7 or 4 column tiles per block.  A partial last group skips its missing tiles' MFMAs, so padding costs little; every
    // group pays the dA generator (or the A-operand reads) again, so FEWER groups win: 7 wherever that saves a group
    // (fc2 step 1 at H = 512: 32 tiles = 5 groups of 7 instead of 8 of 4 — 2.72 -> see profiles)
    if ((a.Dt + 6) / 7 < (a.Dt + 3) / 4) return launch_grad_split_cfg<ACT, 7, 4, 4, MODE>(a, st);
    return launch_grad_split_cfg<ACT, 4, 4, 4, MODE>(a, st);
}

template <int MODE>
int launch_grad_split_act(int act, const GradSplitArgs& a, hipStream_t st) {
#ifndef RBNN_FAST_BUILD
    if (act == RBNN_ACT_RELU) return launch_grad_split<RBNN_ACT_RELU, MODE>(a, st);
    if (act == RBNN_ACT_SIGM || act == RBNN_ACT_TANH)          // both read act' from the stream: one instantiation serves them
        return launch_grad_split<RBNN_ACT_SIGM, (MODE == GRAD_FC2_STEP2 ? GRAD_FC2_STEP2 : MODE)>(a, st);
#endif
    return launch_grad_split<RBNN_ACT_LEAKY, MODE>(a, st);
}

}  // namespace

extern "C" {

int rbnn_input_scales(const float* X, int64_t rows, int32_t cols, int32_t ld, float floor_abs, float mul, float add, float cap,
                      rbnn_dev_scale* out, void* stream) {
    if (!X || !out) return RBNN_ERR_NULL;
    if (rows < 1 || cols < 1 || ld < cols || (ld & 3)) return RBNN_ERR_SHAPE;
    if (!(floor_abs >= 0.f) || !(mul >= 0.f) || !(add >= 0.f) || !(cap > 0.f)) return RBNN_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(out)) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 2 * sizeof(rbnn_dev_scale), st) != hipSuccess) return RBNN_ERR_LAUNCH;
    const long long work = (long long)rows * ((cols + 3) / 4);
    const unsigned grid = (unsigned)std::min<long long>(512, (work + 1023) / 1024);      // grid-stride: ~4+ float4 per thread, <= 512 atomics
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, X, (long long)rows, cols, ld, &out->absmax_bits);
    hipLaunchKernelGGL(scale_finalize_kernel, dim3(1), dim3(64), 0, st, out, floor_abs, mul, add, cap);
    return launch_status();
}

int rbnn_split_rows(const float* src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                    const rbnn_dev_scale* dev_scale, void* dst, int32_t ld_dst, void* stream) {
    if (!src || !dst) return RBNN_ERR_NULL;
    if (rows < 1 || cols < 1 || ld_src < cols || ld_dst < cols || (ld_dst & 31)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const int groups = ld_dst / 8;
    const long long total = (long long)rows * groups;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (long long)rows, cols, ld_src, ldexpf(1.f, scale_exp), dev_scale, (uint4*)dst, groups);
    return launch_status();
}

int rbnn_fc_forward_split(const rbnn_posterior* net, const rbnn_split_images* sp, const void* X_split, int32_t ldx,
                          int32_t x_exp, const rbnn_dev_scale* dev_scales, int32_t N, const int32_t* sidx, int32_t S,
                          int32_t out_kind, const rbnn_workspace* ws, void* stream) {
    if (!net || !sp || !X_split || !ws || !ws->P || !sp->W1_rows) return RBNN_ERR_NULL;
    if (!net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    const bool fc2 = net->arch == RBNN_ARCH_FC2, bm = net->activation == RBNN_ACT_RELU || net->activation == RBNN_ACT_LEAKY;
    const int H = net->hidden, ld = sp->ld_rows;
    if (H < 128 || (H % 128) || ld < net->in_features || (ld & 31) || ldx != ld) return RBNN_ERR_SHAPE;
    if (net->n_classes < 1 || net->n_classes > RBNN_CPAD || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    // the kernels address a sample's weight image and the input image with 32-bit byte offsets from a 64-bit base
    if ((long long)H * ld * 4 >= (1LL << 32) || (long long)N * ld * 4 >= (1LL << 32) || (long long)N * H * 4 >= (1LL << 32)) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(X_split) || !aligned16(sp->W1_rows) || !aligned16(ws->P) || !aligned16(net->b1) || !aligned16(net->W2)) return RBNN_ERR_ALIGN;
    if (fc2 && (!sp->Wm_rows || !net->bm || !ws->hid1 || (bm ? !ws->mask2 : !ws->dact2))) return RBNN_ERR_NULL;
    if (fc2 && (!aligned16(sp->Wm_rows) || !aligned16(ws->hid1) || !aligned16(net->bm))) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    FwdSplitArgs a = {};
    a.X = (const char*)X_split; a.ldx = ldx; a.N = N; a.x_sample_bytes = 0;
    a.W = (const char*)sp->W1_rows; a.w_sample_bytes = (long long)H * ld * 4; a.ldw = ld; a.KT = ld / 32;
    a.b = net->b1; a.W2 = net->W2; a.b2 = net->b2; a.C = net->n_classes; a.H = H;
    a.sidx = sidx; a.S = S; a.out_scale = ldexpf(1.f, -((dev_scales ? 0 : x_exp) + sp->w1_exp)); a.x_ds = dev_scales;
    a.P = ws->P; a.mask = ws->mask1; a.dact = ws->dact1; a.out_kind = out_kind;
    if (!fc2) return launch_forward_split<true>(net->activation, a, st);
    // fc2: layer 1 -> hidden activations as a split-rows image in ws->hid1 (same bytes as the fp32 [S,N,H] buffer), scaled by
    // 2^h1_exp (the caller bounds |h|: max_h sum_d |W1[h,d]| * max|x| + max|b1|); layer 2 reads it per sample
    a.hid = (char*)ws->hid1; a.hid_scale = ldexpf(1.f, sp->h1_exp); a.hid_ds = dev_scales ? dev_scales + 1 : nullptr;
    int rc = launch_forward_split<false>(net->activation, a, st);
    if (rc) return rc;
    FwdSplitArgs b = a;
    b.X = (const char*)ws->hid1; b.ldx = H; b.x_sample_bytes = (long long)N * H * 4;
    b.W = (const char*)sp->Wm_rows; b.w_sample_bytes = (long long)H * H * 4; b.ldw = H; b.KT = H / 32;
    b.b = net->bm; b.out_scale = ldexpf(1.f, -((dev_scales ? 0 : sp->h1_exp) + sp->wm_exp));
    b.x_ds = dev_scales ? dev_scales + 1 : nullptr; b.hid_ds = nullptr;
    b.mask = ws->mask2; b.dact = ws->dact2; b.hid = nullptr;
    return launch_forward_split<true>(net->activation, b, st);
}

int rbnn_split_cols(const float* W, int64_t n_mats, int32_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                    void* dst, int32_t ld_dst, void* stream) {
    if (!W || !dst) return RBNN_ERR_NULL;
    if (n_mats < 1 || rows < 32 || (rows & 31) || cols < 1 || ld_src < cols || ld_dst < cols || (ld_dst & 15)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const long long total = (long long)n_mats * (rows / 32) * 4 * ld_dst;
    hipLaunchKernelGGL(split_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W, (long long)n_mats, rows, cols, ld_src, ldexpf(1.f, scale_exp), (uint4*)dst, ld_dst);
    return launch_status();
}

int rbnn_split_w2gen(const float* W2, int32_t n_mats, int32_t C, int32_t H, int32_t scale_exp, void* dst, void* stream) {
    if (!W2 || !dst) return RBNN_ERR_NULL;
    if (n_mats < 1 || C < 1 || C > 10 || H < 16 || (H & 15)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const long long total = (long long)n_mats * (H / 16) * 64;
    hipLaunchKernelGGL(split_w2gen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W2, n_mats, C, H, ldexpf(1.f, scale_exp), (uint4*)dst);
    return launch_status();
}

int rbnn_split_workspace_query(const rbnn_posterior* net, const rbnn_split_images* sp, int32_t N, int32_t S,
                               rbnn_split_workspace_sizes* out) {
    if (!net || !sp || !out) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || sp->ld_rows < net->in_features || (sp->ld_rows & 31)) return RBNN_ERR_SHAPE;
    rbnn_split_workspace_sizes z = {};
    z.X_split = (size_t)N * sp->ld_rows * 4;
    z.dZ_gen = (size_t)S * mask_ld(N) * 64;
    z.g_scale = (size_t)mask_ld(N) * sizeof(float);
    *out = z;
    return RBNN_OK;
}

int rbnn_fc_input_grad_split(const rbnn_posterior* net, const rbnn_split_images* sp, const int32_t* sidx, int32_t S,
                             int32_t N, int32_t chunk, const rbnn_workspace* ws, const rbnn_split_workspace* sws,
                             int32_t* n_slabs_out, void* stream) {
    if (!net || !sp || !ws || !sws || !ws->dZ || !ws->slabs) return RBNN_ERR_NULL;
    if (!sp->W1_cols || !sp->W2_gen || !sws->dZ_gen || !sws->g_scale) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    const bool fc2 = net->arch == RBNN_ARCH_FC2, bm = net->activation == RBNN_ACT_RELU || net->activation == RBNN_ACT_LEAKY;
    if (bm ? !ws->mask1 : !ws->dact1) return RBNN_ERR_NULL;
    if (fc2 && (bm ? !ws->mask2 : !ws->dact2)) return RBNN_ERR_NULL;
    const int H = net->hidden, Dp = net->in_stride, C = net->n_classes;
    if (H < 128 || (H % 128) || C < 1 || C > 10 || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    if (sp->ld_cols != Dp || (Dp & 15)) return RBNN_ERR_SHAPE;
    if (!aligned16(sp->W1_cols) || !aligned16(sp->W2_gen) || !aligned16(sws->dZ_gen) || !aligned16(ws->dZ)) return RBNN_ERR_ALIGN;
    if (fc2 && (!sp->Wm_cols || !ws->dhid1)) return RBNN_ERR_NULL;
    if (fc2 && (!aligned16(sp->Wm_cols) || !aligned16(ws->dhid1))) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (chunk <= 0) {                                           // the exact mode's slab plan (same workspace)
        rbnn_workspace_sizes q;
        const int rc = rbnn_workspace_query(net, N, S, 0, &q);
        if (rc) return rc;
        chunk = q.chunk;
    }
    if (chunk > S) chunk = S;
    const int nchunks = (S + chunk - 1) / chunk;
    if (n_slabs_out) *n_slabs_out = nchunks;
    const long long n_pad = mask_ld(N);
    hipLaunchKernelGGL(split_dz_kernel, dim3((unsigned)(n_pad / 16)), dim3(256), 0, st,
                       ws->dZ, S, N, n_pad, C, (uint4*)sws->dZ_gen, sws->g_scale);
    if (hipGetLastError() != hipSuccess) return RBNN_ERR_LAUNCH;
    GradSplitArgs g = {};
    g.dzg = (const char*)sws->dZ_gen; g.n_pad = n_pad; g.gscale = sws->g_scale;
    g.W2g = (const char*)sp->W2_gen;
    g.H = H; g.HW = H / 32; g.sidx = sidx; g.S = S; g.N = N;
    if (!fc2) {
        g.mask = ws->mask1; g.dact = ws->dact1; g.W1c = (const char*)sp->W1_cols; g.ldc = sp->ld_cols; g.Dt = Dp / 16;
        g.chunk = chunk; g.nchunks = nchunks; g.out = ws->slabs; g.ldo = Dp;
        g.out_scale = ldexpf(1.f, -(sp->w2_exp + GEN_Q + sp->w1_exp));
        return launch_grad_split_act<GRAD_FC>(net->activation, g, st);
    }
    // fc2 step 1, one sample per block: dhid1[s] = act'(A1_s) * ((act'(A2_s) * (dZ_s . W3_s)) . Wm_s), kept scaled:
    //   stored = dhid1 * 2^(e(n) + e_w3 + GEN_Q + e_wm - Q2),  Q2 = 14 + ceil(log2 H): |dA2 scaled| <= 2^15, |Wm scaled| <= 2^14, K = H
    //   => |stored| <= 2^15, fp16 range, ready to be split as step 2's A operand
    int q2 = 14;
    while ((1 << (q2 - 14)) < H) ++q2;
    g.mask = ws->mask2; g.dact = ws->dact2; g.odact = ws->dact1; g.W1c = (const char*)sp->Wm_cols; g.ldc = H; g.Dt = H / 16;
    g.chunk = 1; g.nchunks = S; g.out = ws->dhid1; g.ldo = H; g.out_scale = ldexpf(1.f, -q2);
    g.omask = ws->mask1; g.OHW = H / 32;
    int rc = launch_grad_split_act<GRAD_FC2_STEP1>(net->activation, g, st);
    if (rc) return rc;
    // fc2 step 2: slabs[k] = sum_{s in chunk k} dhid1[s] . W1_s; acc = g * 2^(e(n) + e_w3 + GEN_Q + e_wm - Q2 + e_w1)
    GradSplitArgs h = g;
    h.amem = ws->dhid1; h.W1c = (const char*)sp->W1_cols; h.ldc = sp->ld_cols; h.Dt = Dp / 16;
    h.chunk = chunk; h.nchunks = nchunks; h.out = ws->slabs; h.ldo = Dp;
    h.out_scale = ldexpf(1.f, -(sp->w2_exp + GEN_Q + sp->wm_exp - q2 + sp->w1_exp));
    return launch_grad_split_act<GRAD_FC2_STEP2>(net->activation, h, st);
}

}  // extern "C"
