// rbnn_kernels.hip — hand-written gfx950 (CDNA4 / MI355X) kernels + the C-ABI of include/robustbnns_hip.h.
//
// The reference's hot path (model_bnn.py:243-258 -> lossGradients.py:20-50 / adversarialAttacks.py:69-108)
// is a Python nest point -> PGD iteration -> posterior sample with batch-1 GEMVs at the leaves.  Here the nest
// is inverted: all N points are batched against each sample's weights, so the two contractions become
//
//   forward   A[n, (s,h)] = sum_d X[n,d] * W1[s,h,d]          one (N) x (S*H) x (D)   fp32 GEMM
//   backward  g[n, d]     = sum_(s,h) dA[n,(s,h)] * W1[s,h,d]  one (N) x (D)  x (S*H)  fp32 GEMM
//
// on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain — needed for the 1e-5 parity bar; gfx950 has no
// xf32).  Everything between them is fused around the MFMA accumulators:
//
//   fc_forward_kernel   bias + activation + the skinny H->C layer (an MFMA that takes the accumulator tile as
//                       its B operand, no LDS round trip) + softmax; stashes 1 bit per hidden unit
//                       (pre-activation > 0) — or act' in fp32 for sigmoid/tanh — for the backward.
//   fc_grad_kernel      builds dA = act'(A) * (dZ . W2) on the VALU straight into the MFMA A-operand layout
//                       while the matrix pipe runs, accumulates over a chunk of samples, writes one slab.
//
// Lane maps used throughout (16x16x4 f32 MFMA, wave64, li = lane & 15, lg = lane >> 4):
//   A operand  a = A[i = li][k = lg]      B operand  b = B[k = lg][j = li]
//   C/D        acc[r] = D[i = 4*lg + r][j = li]
// The k <-> memory-index map of a K step is free as long as A and B agree; both kernels use
// "step r of a 16-wide k block: k = lg  <->  index 4*lg + r", so one 16-byte load feeds four K steps.
#include "rbnn_common.hpp"
#ifndef RBNN_FWD_SB
#define RBNN_FWD_SB 8                                          // forward: samples per XCD-resident panel of blocks
#endif
#include <stdlib.h>

namespace {

// ===================================================================================================
// K1: stacked forward.  One block = one (point tile, sample) item:
//   acc[h][n] = sum_d W[s][h][d] * X[n][d]      (A operand = W rows, B operand = X rows; D = [h][n])
// so the accumulator's ROW index is h: the skinny next layer Z^T[c][n] = sum_h W2[c][h] * act(acc)[h][n]
// sums over the accumulator's row index and takes each accumulator register directly as its B operand.
// ===================================================================================================
struct FwdArgs {
    const float* X;   long long x_sample_stride;  int ldx;  int N;      // input rows (shared, or per sample for fc2 layer 2)
    const float* W;   long long w_sample_stride;  int ldw;  int KT;     // [S_total][H][ldw], KT = ldw/16
    const float* b;                                                      // [S_total][H]
    const float* W2;  const float* b2;            int C;    int H;      // output layer (LAYER2 only)
    const int* sidx;  int S;   int NT;
    float* P;  uint32_t* mask;  float* dact;  float* hid;  int out_kind;
};

template <int ACT, int WH, int HTW, int WN, int NTW, bool LAYER2>
__global__ void __launch_bounds__(64 * WH * WN, (HTW * NTW > 16 ? 2 : (WH * WN) / 2)) fc_forward_kernel(const FwdArgs a) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int TILE = (BH + BN) * 16;                       // floats per LDS buffer: BH W-rows then BN X-rows of 16 floats
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);
    constexpr int NW = WH * WN;                                // waves per block (4 or 8)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
    static_assert(HTW % 2 == 0 && HTW <= 8, "a wave's h range is whole 32-bit mask words, at most 4");
    static_assert(WH * BN * 16 <= 2 * TILE, "the Z^T reduction scratch aliases the tile buffers");
    // K tiles go through a ring of LDS buffers filled by LDS-DMA.  When every wave issues the same number of DMA
    // pieces per tile (OPS) the ring is 3 deep and tile kt+2 is in flight while tile kt is multiplied: the wait
    // before the barrier is a COUNTED vmcnt(OPS) (tile kt+1 landed, kt+2 may still fly) and the barrier is a raw
    // s_barrier — __syncthreads() would drain vmcnt(0) and expose one full L2/HBM latency per tile.
    // (Measured at C2: the 3-deep ring is NOT faster than the 2-deep one — 6.34 vs 6.23 ms — the forward kernel is
    // bound by the CU's memory-pipe throughput for the DMA pieces, not by their latency.  Kept behind RBNN_RING3.)
#ifdef RBNN_RING3
    constexpr bool RING3 = (BH / 16) % NW == 0 && (BN / 16) % NW == 0;
#else
    constexpr bool RING3 = false;
#endif
    constexpr int NBUF = RING3 ? 3 : 2, OPS = BH / 16 / NW + BN / 16 / NW;
    // ONE LDS array (tile ring; the softmax scratch aliases it after the K loop): global_load_lds
    // staging beside a second __shared__ object makes hipcc drain vmcnt before every ds_read.
    __shared__ __attribute__((aligned(16))) float lds[NBUF * TILE];
    float* const zred = lds;

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.S, id)) return;
    // item -> (point tile, sample), 2-D blocked: panels of 8 samples, inside a panel the sample index runs fastest, so
    // the ~64 blocks resident on an XCD are 8 point tiles x 8 samples: each X tile and each W1 slice is fetched into that
    // XCD's L2 once per 8 users instead of once per user.
#ifndef RBNN_FWD_ITEM_1D
    int ntile, s;
    {
        constexpr int SB = RBNN_FWD_SB;
        const int full = a.S / SB, per = SB * a.NT;
        if (id < full * per) { ntile = (id % per) / SB; s = (id / per) * SB + id % SB; }
        else { const int rem = id - full * per, cnt = a.S - full * SB; ntile = rem / cnt; s = full * SB + rem % cnt; }
    }
#else
    const int ntile = id % a.NT, s = id / a.NT;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int wave_h = wave % WH, wave_n = wave / WH;
    const int sw = a.sidx ? a.sidx[s] : s;
    const float* const Ws = a.W + (long long)sw * a.w_sample_stride;
    const float* const Xs = a.X + (long long)s * a.x_sample_stride;
    const int n0 = ntile * BN;
    const int HW = a.H >> 5;
    // LDS-DMA piece = one wave instruction = 1 KiB = 16 tile rows; lane p lands at row p>>2, physical 16-B chunk p&3,
    // so it FETCHES logical chunk (p&3)^swz(row): the swizzle lives on the source address, the LDS image is linear.
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ swz(prow);
    const int pch = 4 * (lg ^ swz(li));                        // fragment read: row li, logical chunk lg

    f32x4 zacc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) zacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int hc0 = 0; hc0 < a.H; hc0 += BH) {
        f32x4 acc[HTW][NTW];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        auto stage = [&](int kt, int buf) {                    // K tile kt -> LDS buffer buf, asynchronously, no VGPRs
            float* const Wt = lds + buf * TILE;
            float* const Xt = Wt + BH * 16;
            const int k0 = kt * 16 + 4 * lchunk;
            for (int q = wave; q < BH / 16; q += NW)
                glds16(Ws + (long long)(hc0 + q * 16 + prow) * a.ldw + k0, Wt + q * 256);
            for (int q = wave; q < BN / 16; q += NW) {
                const int n = min(n0 + q * 16 + prow, a.N - 1);    // rows past N repeat the last point; never stored
                glds16(Xs + (long long)n * a.ldx + k0, Xt + q * 256);
            }
        };
        stage(0, 0);
        if (RING3 && a.KT > 1) stage(1, 1);
        if (RING3 && a.KT > 1) ring_wait_barrier<OPS>(); else ring_wait_barrier<0>();   // tile 0 landed for every wave
        f32x4 bf[NTW], a_cur, a_nxt;
        int buf = 0, nbuf = RING3 ? 2 : 1;                     // buffer of tile kt / of the tile staged in this iteration
        for (int kt = 0; kt < a.KT; ++kt) {
            const bool more = kt + (RING3 ? 2 : 1) < a.KT;
            if (!(RBNN_ABL & 1) && more) stage(kt + (RING3 ? 2 : 1), nbuf);   // lands while this (and the next) tile is multiplied
            const float* const Wt = lds + buf * TILE;
            const float* const Xt = Wt + BH * 16;
            // Fragment reads are software-pipelined one h-tile ahead of the MFMAs that consume them, and the
            // order is pinned (sched_group_barrier): left alone, hipcc waits lgkmcnt(0) right after each read.
            if (!(RBNN_ABL & 4) || kt == 0) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bf[nt] = *(const f32x4*)(Xt + ((wave_n * NTW + nt) * 16 + li) * 16 + pch);
                a_cur = *(const f32x4*)(Wt + ((wave_h * HTW) * 16 + li) * 16 + pch);
                a_nxt = a_cur;
            }
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                if (!(RBNN_ABL & 4) && ht + 1 < HTW) a_nxt = *(const f32x4*)(Wt + ((wave_h * HTW + ht + 1) * 16 + li) * 16 + pch);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = MFMA16(a_cur[j], bf[nt][j], acc[ht][nt]);
                a_cur = a_nxt;
            }
            // hipcc's waits are lgkmcnt(0): keep every prefetch read half a tile (8 MFMAs = 256 cycles) ahead of
            // the wait that follows it, so that draining it is free.
            if (!(RBNN_ABL & 4)) {
                __builtin_amdgcn_sched_group_barrier(0x100, NTW + 1, 0);           // DS reads: bf[*], a(0)
#pragma unroll
                for (int ht = 0; ht < HTW; ++ht) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);                // 8 MFMAs on a(ht)
                    if (ht + 1 < HTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read a(ht+1)
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);                // 8 MFMAs on a(ht)
                }
            }
            if (!(RBNN_ABL & 2)) {
                // tile kt+1 landed (this wave's share; the barrier covers the others); everyone is done with tile kt
                if (RING3 && more) ring_wait_barrier<OPS>(); else ring_wait_barrier<0>();
            }
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == NBUF) ? 0 : nbuf + 1;
        }
        if (RBNN_ABL & 2) __syncthreads();
        if (RBNN_ABL & 8) {                                    // diagnostic: keep the accumulators live, skip the epilogue
            float t = 0.f;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) t += acc[ht][nt][0] + acc[ht][nt][1] + acc[ht][nt][2] + acc[ht][nt][3];
            if (t == 12345.678f) a.P[tid] = t;
            continue;
        }

        // ---- epilogue of this h chunk: bias, activation, derivative stash, skinny output layer ----
        const int hw0 = hc0 + (wave_h * HTW) * 16;            // this wave's first hidden unit
        unsigned mine[NTW];                                    // lane lg keeps mask word lg of the wave's h range
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
                f32x4 v = acc[ht][nt] + bias, hv;
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bits |= (v[r] > 0.f ? 1u : 0u) << r;
                    hv[r] = act_fwd<ACT>(v[r]);
                }
                if (BITMASK) {
                    // bit (16*(ht&1) + 4*lg + r) of word ht/2; OR over the four lg lanes of this point
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
                if (!LAYER2) {
                    if (n < a.N) *(f32x4*)(a.hid + ((long long)s * a.N + n) * a.H + hrow) = hv;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) zacc[nt] = MFMA16(w2f[r], hv[r], zacc[nt]);
                }
            }
        }
        if (BITMASK && a.mask) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = n0 + (wave_n * NTW + nt) * 16 + li;
                if (lg < HTW / 2 && n < a.N) a.mask[((long long)s * HW + (hw0 >> 5) + lg) * mask_ld(a.N) + n] = mine[nt];   // [S][H/32][N_pad]: 16 lanes = 64 B
            }
        }
    }

    if (LAYER2) {
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
}

// ===================================================================================================
// K2: input gradient.  One block = one (256-point tile, TD*16-column group, chunk of samples) item:
//   acc[n][d] += sum_h dA[n][h] * W1[s][h][d],   dA[n][h] = act'(A_s[n][h]) * sum_c dZ[s][n][c] * W2[s][c][h]
// A operand (lane: n = li, k = lg) is produced on the VALU from dZ registers, a W2 row in LDS and the
// 1-bit stash; B operand (lane: k = lg, d = li) is read from an LDS tile of W1.  Accumulators stay in
// registers across the whole chunk (K = chunk * H) and leave as one slab tile.
// With A_MEM the A operand is read from memory instead (fc2: dL/d(pre-activation 1), [S][N][H]).
// With PER_SAMPLE the block handles ONE sample, multiplies the result by that sample's layer-1
// activation derivative and writes it to out[s] (fc2: dA1 = act'(A1) * (dA2 . Wm)).
// ===================================================================================================
struct GradArgs {
    const float* dZ;  const uint32_t* mask;  const float* dact;  const float* amem;   // A-operand sources
    const float* W1;  long long w1_sample_stride;  int ldw;                           // B operand [S_total][H][ldw]
    const float* W2;  int C;  int H;  int HW;
    const int* sidx;  int S;  int chunk;  int nchunks;
    int N;  int NT;  int ND;  int Dt;                                                  // Dt = valid 16-col tiles
    float* out;  int ldo;                                                              // slabs [nchunks][N][ldo] / per-sample [S][N][ldo]
    const uint32_t* omask;  const float* odact;  int OHW;                              // PER_SAMPLE epilogue: derivative of the layer below
};

template <int ACT, int TD, int CQ, bool A_MEM, bool PER_SAMPLE, int HSTG>
__global__ void __launch_bounds__(256, 2) fc_grad_kernel(const GradArgs a) {
    constexpr int NTW = 4, BM = 256;                           // 4 waves x 64 points; HSTG (32 or 64) hidden units per LDS stage
    constexpr int LD = TD * 16;                                // W1 stage tile: [HSTG/4 h-quads][TD*16 d][4], linear LDS-DMA image
    constexpr int W2RPP = 256 / HSTG;                          // W2 tile rows (classes) per 1-KiB piece
    constexpr int W2ROWS = (4 * CQ + W2RPP - 1) / W2RPP * W2RPP;   // W2 stage tile: [classes, whole pieces][HSTG h]
    constexpr int W1SZ = HSTG * LD, W2SZ = W2ROWS * HSTG;
    constexpr int MKSZ = (HSTG / 32) * BM;                      // mask words of the stage for the block's 256 points
    constexpr int BUF = W1SZ + W2SZ + MKSZ;
    constexpr int NPIECE = HSTG * TD / 16, PPW = (NPIECE + 3) / 4;   // 1-KiB LDS-DMA pieces per W1 tile / per wave
    constexpr int NT2 = HSTG / 16, MWS = HSTG / 32;            // h tiles / mask words per stage
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];   // one array: see fc_forward_kernel

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.ND * a.nchunks, id)) return;
    int ntile, dg, ch;
    grad_item(id, a.NT, a.ND, A_MEM || !BITMASK, ntile, dg, ch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int nb = ntile * BM + wave * (NTW * 16);            // this wave's first point
    const int dc0 = dg * TD * 16;                              // this block's first column
    const int Dp = a.Dt * 16;
    const int s_begin = ch * a.chunk, s_end = min(a.S, s_begin + a.chunk);
    const int HS = a.H / HSTG, nst = (s_end - s_begin) * HS;

    // W1 comes from its rbnn_pack_rows4 image [H/4][ldw][4]: a stage tile is [8 h-quads][TD*16 d][4] floats, and the
    // B operand of the 4 K steps of h tile t2 — W1[16*t2 + 4*lg + r][d], r = 0..3 — is ONE 16-byte LDS read at
    // [(4*t2 + lg)][dt*16 + li][0..3]; a 16-lane ds_read_b128 group then covers 16 distinct 16-B slots (the quad
    // stride TD*256 B is a multiple of 256 B), so the linear LDS-DMA image needs no swizzle.
    // goff: per-lane source offset (floats) of this wave's 1-KiB pieces inside a stage.
    int goff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int f = (wave + 4 * i) * 256 + 4 * lane, hq = f / (4 * LD), d = (f % (4 * LD)) >> 2;
        goff[i] = (hq * a.ldw + min(dc0 + d, Dp - 1)) * 4;    // columns past D_pad: any valid address, never stored
    }
    // W2 tile: piece q = classes q*W2RPP .., lane p -> class row p / (HSTG/4), 16-B chunk p % (HSTG/4) of its HSTG hidden
    // units; odd classes are stored rotated by 16 units (a half-wave reads classes c, c+1: opposite bank halves).
    const int w2row = lane / (HSTG / 4);
    const int w2goff_col = 4 * (((lane % (HSTG / 4)) + (HSTG / 4) - 4 * (w2row & 1)) % (HSTG / 4));

    f32x4 acc[NTW][TD];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) acc[nt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float dzb[NTW][CQ];                                        // B operand of the dA product: dZ[s][n = li][c = 4j + lg]
    // stage st -> LDS buffer buf, all by LDS-DMA (asynchronous, no VGPRs); mask words to registers
    auto stage_issue = [&](int st, int buf) {
        const int s = s_begin + st / HS, h0 = (st % HS) * HSTG;
        const int sw = a.sidx ? a.sidx[s] : s;
        const float* const Ws = a.W1 + (long long)sw * a.w1_sample_stride + (long long)h0 * a.ldw;   // quad h0/4 of the packed image
        const bool dma = !(RBNN_ABL & 64) || st == 0;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (dma && wave + 4 * i < NPIECE) glds16(Ws + goff[i], lds + buf * BUF + (wave + 4 * i) * 256);
        if (!A_MEM) {
            if (dma && wave * W2RPP < W2ROWS) {                // W2 rows past C repeat row C-1: they meet dZ columns that are 0
                const int c = min(wave * W2RPP + w2row, a.C - 1);
                glds16(a.W2 + ((long long)sw * a.C + c) * a.H + h0 + w2goff_col, lds + buf * BUF + W1SZ + wave * 256);
            }
            // mask words [S][H/32][N_pad] (N_pad % 256 == 0): the block's 256 points of one word row are 1 KiB = one piece
            if (BITMASK && dma && wave >= 4 - MWS) {
                const int w = wave - (4 - MWS);
                glds16((const float*)(a.mask + ((long long)s * a.HW + (h0 >> 5) + w) * mask_ld(a.N) + ntile * BM + 4 * lane),
                       lds + buf * BUF + W1SZ + W2SZ + w * BM);
            }
        }
    };

    stage_issue(0, 0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int s = s_begin + st / HS, h0 = (st % HS) * HSTG, buf = st & 1;
        if (!A_MEM && st % HS == 0) {                          // new sample: its dL/dlogits into B-operand registers
#pragma unroll                                                 // (before this stage's LDS-DMA is issued: waiting on
            for (int nt = 0; nt < NTW; ++nt) {                 //  these loads would drain it)
                const int n = nb + nt * 16 + li;
#pragma unroll
                for (int j = 0; j < CQ; ++j)
                    dzb[nt][j] = (n < a.N) ? a.dZ[((long long)s * a.N + n) * RBNN_CPAD + 4 * j + lg] : 0.f;
            }
            // Retire these ordinary loads NOW (s_waitcnt vmcnt(0), visible to hipcc's wait insertion): with an
            // LDS-DMA in flight hipcc waits vmcnt(0) at their first use, which would drain the DMA every stage.
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        if (!(RBNN_ABL & 1) && st + 1 < nst) stage_issue(st + 1, buf ^ 1);   // lands while this stage's MFMAs run

        const float* const W1t = lds + buf * BUF;
        const float* const W2t = W1t + W1SZ;
        const unsigned* const Mk = (const unsigned*)(W2t + W2SZ) + wave * (NTW * 16) + li;   // + w*BM + nt*16
        // dA tile of h tile t2 for this wave's 64 points: (dA^T)[h][n] = sum_c W2[c][h] * dZ[n][c] on the matrix pipe
        // (A = W2 fragment, i = h; B = dzb, j = n).  Its accumulator layout, register r <-> h = 16*t2 + 4*lg + r on
        // lane (n = li, lg), IS the main product's A-operand layout for K step (t2, r): no data movement.
        f32x4 da[2][NTW];
        auto make_da = [&](int t2, int slot) {
            if (A_MEM || !BITMASK) {                           // A from memory, or smooth act': 16-B loads, 4 K steps each
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n = nb + nt * 16 + li;
                    const float* const src = (A_MEM ? a.amem : a.dact) + ((long long)s * a.N + n) * a.H + h0 + t2 * 16 + 4 * lg;
                    da[slot][nt] = (n < a.N) ? *(const f32x4*)src : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (A_MEM) return;
            }
            float w2a[CQ];
#pragma unroll
            for (int j = 0; j < CQ; ++j) w2a[j] = W2t[(4 * j + lg) * HSTG + (16 * t2 + li + 16 * (lg & 1)) % HSTG];   // class 4j+lg, unit 16*t2+li
            f32x4 g[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) g[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < CQ; ++j)                       // class quad outer: the 4 accumulation chains interleave
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) g[nt] = MFMA16(w2a[j], dzb[nt][j], g[nt]);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                unsigned mwv = 0;                               // bit r = unit 16*t2 + 4*lg + r of this lane's point
                if (BITMASK) mwv = Mk[(t2 >> 1) * BM + nt * 16] >> (16 * (t2 & 1) + 4 * lg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (BITMASK) {
                        const bool pos = (mwv >> r) & 1u;
                        g[nt][r] = pos ? g[nt][r] : (ACT == RBNN_ACT_RELU ? 0.f : g[nt][r] * LEAKY_SLOPE);
                    } else {
                        g[nt][r] *= da[slot][nt][r];
                    }
                }
                da[slot][nt] = g[nt];
            }
        };
        // B operand: one ds_read_b128 per (h tile, column tile) = 4 K steps.  The MFMAs of an h tile run column-tile
        // major (dt outer; r, nt inner: 16 MFMAs per register quad), so b4[dt] is dead as soon as its 16 MFMAs have
        // issued and is re-read for the NEXT h tile right there — the read hides under the other column tiles'
        // MFMAs with no second register set.  Per accumulator the K order (t2, r ascending) is unchanged.
        f32x4 b4[TD];
        auto read_b = [&](int t2, int dt) {
            b4[dt] = *(const f32x4*)(W1t + ((4 * t2 + lg) * LD + dt * 16 + li) * 4);
        };
        make_da(0, 0);
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) read_b(0, dt);
        static_for<0, NT2>([&](auto tc) {
            constexpr int t2 = decltype(tc)::value;
            if (!(RBNN_ABL & 20) && t2 + 1 < NT2) make_da(t2 + 1, (t2 + 1) & 1);   // next h tile's dA ahead of this tile's MFMAs
#pragma unroll
            for (int dt = 0; dt < TD; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[nt][dt] = MFMA16(da[(RBNN_ABL & 20) ? 0 : (t2 & 1)][nt][r], b4[dt][r], acc[nt][dt]);
                if (!(RBNN_ABL & 4) && t2 + 1 < NT2) read_b(t2 + 1, dt);
#ifndef RBNN_NO_PIN
                __builtin_amdgcn_sched_barrier(0);             // keep the re-read HERE: hipcc would sink it to its consumer
#endif
            }
        });
        if (!(RBNN_ABL & 2)) __syncthreads();                  // vmcnt(0): next tiles landed; everyone is done with these
    }

    if (RBNN_ABL & 2) __syncthreads();
    // ---- epilogue: acc[nt][dt][r] = D[n = nb + nt*16 + 4*lg + r][d = dc0 + dt*16 + li]
    const int sidx_out = PER_SAMPLE ? s_begin : ch;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nb + nt * 16 + 4 * lg + r;
            if (n >= a.N) continue;
            float* const dst = a.out + ((long long)sidx_out * a.N + n) * a.ldo;
#pragma unroll
            for (int dt = 0; dt < TD; ++dt) {
                const int d = dc0 + dt * 16 + li;
                if (d >= Dp) continue;
                float v = acc[nt][dt][r];
                if (PER_SAMPLE) {
                    if (BITMASK) {
                        const unsigned w = a.omask[((long long)s_begin * a.OHW + (d >> 5)) * mask_ld(a.N) + n];
                        v = ((w >> (d & 31)) & 1u) ? v : (ACT == RBNN_ACT_RELU ? 0.f : v * LEAKY_SLOPE);
                    } else {
                        v *= a.odact[((long long)s_begin * a.N + n) * a.ldo + d];
                    }
                }
                dst[d] = v;
            }
        }
}

// ===================================================================================================
// Small streaming kernels (HBM-bound, trivially cheap next to K1/K2).
// ===================================================================================================
__global__ void reduce_samples_kernel(const float* __restrict__ P, int S, int N, int C, float scale,
                                      float* __restrict__ out, int ldo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (n, 4-class quad)
    if (i >= N * 4) return;
    const int n = i >> 2, q = i & 3;
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) sum += *(const f32x4*)(P + ((long long)s * N + n) * RBNN_CPAD + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * q + r < C) out[(long long)n * ldo + 4 * q + r] = sum[r] * scale;
}

__global__ void loss_dlogits_kernel(int mode, const float* __restrict__ P, const float* __restrict__ Psum, int ldp,
                                    const float* __restrict__ Gup, const int* __restrict__ labels, int S, float inv_S,
                                    int N, int C, float* __restrict__ dZ) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (s, n)
    if (i >= (long long)S * N) return;
    const int n = (int)(i % N);
    float p[16], g[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(P + i * RBNN_CPAD + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * q + r] = v[r];
    }
    const bool upstream = mode == RBNN_LOSS_UPSTREAM || mode == RBNN_LOSS_UPSTREAM_LOGIT;
    const int y = upstream ? -1 : labels[n];
    if (upstream) {
#pragma unroll
        for (int c = 0; c < 16; ++c) g[c] = (c < C) ? Gup[(long long)n * ldp + c] * inv_S : 0.f;
    } else {
        // softmax of what the loss saw: mean probs (double softmax), this sample's probs, or mean logits
        float t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = (c < C) ? ((mode == RBNN_LOSS_PER_SAMPLE) ? p[c] : Psum[(long long)n * ldp + c] * inv_S) : 0.f;
        ce_softmax_grad<16>(t, C, y, inv_S, g);                // softmax(t) - e_y, the label class without its cancellation (rbnn_common.hpp)
    }
    float out[16];
    if (mode == RBNN_LOSS_MEAN_LOGIT || mode == RBNN_LOSS_UPSTREAM_LOGIT) {        // P holds logits: dZ_s = dL/d(mean logits) / S, no softmax backward
#pragma unroll
        for (int c = 0; c < 16; ++c) out[c] = g[c];
    } else {
        softmax_backward<16>(g, p, C, out);                    // (g - <g,p>) * p in its cancellation-free form (rbnn_common.hpp)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *(f32x4*)(dZ + i * RBNN_CPAD + 4 * q) = (f32x4){out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
}

__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int K, long long slab_stride, long long total4,
                                 int d4, float scale, float* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 4 columns
    if (i >= total4) return;
    f32x4 sum = *(const f32x4*)(slabs + 4 * i);
    for (int k = 1; k < K; ++k) sum += *(const f32x4*)(slabs + k * slab_stride + 4 * i);
    const long long n = i / d4, c4 = i % d4;
    *(f32x4*)(out + n * ldo + 4 * c4) = sum * scale;
}

// sum_slabs with the per-point norms of the result fused in: one wave per point row, lanes stride over the row's float4s,
// slabs summed in index order (the same fixed order as sum_slabs_kernel: bit-identical `out`), then max |g| and sum g^2 over the
// row's d < D by a fixed butterfly (deterministic).  lossGradients.py:78-127 takes these norms of every expected gradient.
__global__ void __launch_bounds__(256) sum_slabs_norms_kernel(const float* __restrict__ slabs, int K, long long slab_stride, int N,
                                                               int d_pad, int D, float scale, float* __restrict__ out, int ldo,
                                                               float* __restrict__ linf, float* __restrict__ l2) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float m = 0.f, ss = 0.f;
    for (int c = 4 * lane; c < d_pad; c += 256) {
        const long long off = (long long)n * d_pad + c;
        f32x4 sum = *(const f32x4*)(slabs + off);
        for (int k = 1; k < K; ++k) sum += *(const f32x4*)(slabs + k * slab_stride + off);
        sum = sum * scale;
        *(f32x4*)(out + (long long)n * ldo + c) = sum;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c + r < D) { m = fmaxf(m, fabsf(sum[r])); ss = fmaf(sum[r], sum[r], ss); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); ss += __shfl_xor(ss, o); }
    if (lane == 0) { linf[n] = m; l2[n] = sqrtf(ss); }
}

__global__ void pgd_alpha_kernel(const float* __restrict__ X0, int ldx, int N, int D, float* __restrict__ alpha) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per image
    if (n >= N) return;
    float m = -INFINITY;
    for (int d = lane; d < D; d += 64) m = fmaxf(m, X0[(long long)n * ldx + d]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) alpha[n] = 2.f / m;
}

__global__ void attack_step_kernel(float* __restrict__ X, const float* __restrict__ X0, int ldx,
                                   const float* __restrict__ G, int K, long long slab_stride, int ldg,
                                   const float* __restrict__ alpha, float alpha_scalar, float eps, int project,
                                   int N, int D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pixel
    if (i >= (long long)N * D) return;
    const long long n = i / D;
    const int d = (int)(i % D);
    float g = G[n * ldg + d];
    for (int k = 1; k < K; ++k) g += G[k * slab_stride + n * ldg + d];
    const float sgn = (g > 0.f) ? 1.f : ((g < 0.f) ? -1.f : 0.f);          // torch.sign: sign(0) = 0
    const float step = alpha ? alpha[n] : alpha_scalar;
    const float x = X[n * ldx + d];
    float pert = x + step * sgn;
    if (project) {
        const float x0 = X0[n * ldx + d];
        const float eta = fminf(fmaxf(pert - x0, -eps), eps);
        pert = x0 + eta;
    }
    X[n * ldx + d] = fminf(fmaxf(pert, 0.f), 1.f);
}

// four pixels per thread (16-byte loads of every slab, of x and of x0): D, ldx, ldg multiples of 4 and 16-byte aligned bases
__global__ void attack_step4_kernel(float* __restrict__ X, const float* __restrict__ X0, int ldx,
                                    const float* __restrict__ G, int K, long long slab_stride, int ldg,
                                    const float* __restrict__ alpha, float alpha_scalar, float eps, int project,
                                    int N, int D4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * D4) return;
    const long long n = i / D4;
    const int d = 4 * (int)(i % D4);
    f32x4 g = *(const f32x4*)(G + n * ldg + d);
    for (int k = 1; k < K; ++k) g += *(const f32x4*)(G + k * slab_stride + n * ldg + d);
    const float step = alpha ? alpha[n] : alpha_scalar;
    const f32x4 x = *(const f32x4*)(X + n * ldx + d);
    f32x4 x0 = x, out;
    if (project) x0 = *(const f32x4*)(X0 + n * ldx + d);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float sgn = (g[r] > 0.f) ? 1.f : ((g[r] < 0.f) ? -1.f : 0.f);
        float pert = x[r] + step * sgn;
        if (project) pert = x0[r] + fminf(fmaxf(pert - x0[r], -eps), eps);
        out[r] = fminf(fmaxf(pert, 0.f), 1.f);
    }
    *(f32x4*)(X + n * ldx + d) = out;
}

__global__ void eval_metrics_kernel(const float* __restrict__ A, const float* __restrict__ B, int ldp,
                                    const int* __restrict__ labels, int N, int C, int* __restrict__ counts,
                                    float* __restrict__ rob) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    int ok_a = 0, ok_b = 0;
    if (n < N) {
        float ma = -INFINITY, mb = -INFINITY;
        int ia = 0, ib = 0;
        for (int c = 0; c < C; ++c) {
            const float a = A[(long long)n * ldp + c], b = B[(long long)n * ldp + c];
            if (a > ma) { ma = a; ia = c; }
            if (b > mb) { mb = b; ib = c; }
        }
        float da = 0.f, db = 0.f;
        for (int c = 0; c < C; ++c) { da += expf(A[(long long)n * ldp + c] - ma); db += expf(B[(long long)n * ldp + c] - mb); }
        float diff = 0.f;
        for (int c = 0; c < C; ++c)
            diff = fmaxf(diff, fabsf(expf(A[(long long)n * ldp + c] - ma) / da - expf(B[(long long)n * ldp + c] - mb) / db));
        rob[n] = 1.f - diff;
        ok_a = (ia == labels[n]);
        ok_b = (ib == labels[n]);
    }
    // wave-level count, one atomic per wave
    const unsigned long long ba = __ballot(ok_a), bb = __ballot(ok_b);
    if ((threadIdx.x & 63) == 0) {
        if (ba) atomicAdd(&counts[0], __popcll(ba));
        if (bb) atomicAdd(&counts[1], __popcll(bb));
    }
}

__global__ void pack_rows4_kernel(const float* __restrict__ W, long long quads, int cols, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (row quad, column): 16-B store
    if (i >= quads * cols) return;
    const long long q = i / cols, c = i % cols;
    const float* const src = W + (4 * q) * cols + c;
    *(f32x4*)(out + 4 * i) = (f32x4){src[0], src[cols], src[2LL * cols], src[3LL * cols]};
}

__global__ void svi_materialize_kernel(const float* __restrict__ loc, const float* __restrict__ scale_raw,
                                       const float* __restrict__ eps, long long n_elem, int S, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elem * S) return;
    const long long e = i % n_elem;
    const float sr = scale_raw[e];
    const float sp = sr > 20.f ? sr : log1pf(expf(sr));        // torch.nn.Softplus(beta=1, threshold=20), model_bnn.py:18
    out[i] = loc[e] + sp * eps[i];                             // Normal.rsample(): loc + eps * scale
}

// ---------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------
int validate_net(const rbnn_posterior* net) {
    if (!net || !net->W1 || !net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    if (net->arch == RBNN_ARCH_FC2 && (!net->Wm || !net->bm)) return RBNN_ERR_NULL;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    if (net->in_features < 1 || net->in_stride < net->in_features || (net->in_stride & 15)) return RBNN_ERR_SHAPE;
    const int H = net->hidden;
    if (H < 32 || (H & 31)) return RBNN_ERR_SHAPE;
    if (H < 512 ? (H & (H - 1)) != 0 : (H % 512) != 0) return RBNN_ERR_SHAPE;   // 32..256 powers of two, or k*512
    if (net->n_classes < 1 || net->n_classes > RBNN_CPAD || net->n_stored < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(net->W1) || !aligned16(net->b1) || !aligned16(net->W2)) return RBNN_ERR_ALIGN;
    if (net->arch == RBNN_ARCH_FC2 && (!aligned16(net->Wm) || !aligned16(net->bm))) return RBNN_ERR_ALIGN;
    return RBNN_OK;
}

inline bool is_bitmask(int act) { return act == RBNN_ACT_RELU || act == RBNN_ACT_LEAKY; }

// d-tile grouping of the gradient kernel for a K-padded width of Dt 16-column tiles
inline int pick_td(int Dt) { return (Dt % 7 == 0) ? 7 : (Dt < 4 ? 1 : 4); }

// samples per slab: fill the 2-blocks-per-CU slots evenly, keep fp32 chains short, count slab traffic
int pick_chunk(int N, int Dt, int S) {
    if (const char* e = getenv("RBNN_CHUNK")) {                 // experiments only
        const int c = atoi(e);
        if (c >= 1) return c < S ? c : S;
    }
    const int NT = (N + 255) / 256, TD = pick_td(Dt), ND = (Dt + TD - 1) / TD;
    const long long base = (long long)NT * ND;
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const long long slots = 2LL * cus;
    double best = 1e300;
    int best_chunk = 1;
    for (int chunk = 1; chunk <= 8 && chunk <= S; ++chunk) {
        const int nsl = (S + chunk - 1) / chunk;
        const long long rounds = (base * nsl + slots - 1) / slots;
        const double cost = (double)rounds * chunk + 0.13 * nsl * ((double)NT * ND / 280.0) * (512.0 / slots);
        if (cost < best - 1e-9) { best = cost; best_chunk = chunk; }
    }
    return best_chunk;
}

template <int ACT, int WH, int HTW, int WN, int NTW>
int launch_forward_cfg(const FwdArgs& a, bool layer2, hipStream_t st) {
    constexpr int BN = WN * NTW * 16;
    FwdArgs b = a;
    b.NT = (a.N + BN - 1) / BN;
    const int grid = grid_for_items((long long)b.NT * a.S);
    if (layer2) hipLaunchKernelGGL((fc_forward_kernel<ACT, WH, HTW, WN, NTW, true>), dim3(grid), dim3(64 * WH * WN), 0, st, b);
    else        hipLaunchKernelGGL((fc_forward_kernel<ACT, WH, HTW, WN, NTW, false>), dim3(grid), dim3(64 * WH * WN), 0, st, b);
    return launch_status();
}

template <int ACT>
int launch_forward_act(const FwdArgs& a, bool layer2, hipStream_t st) {
    const int H = a.H;
    // 256 h x 128 n per block, H/256 chunks per item: 33% fewer L2->LDS bytes per MAC than 512 h x 64 n (measured
    // 83.8% vs 81.0% of the fp32 MFMA peak at C2; the Z^T accumulators persist across the h chunks; the 8-wave 512 x 64 and 256 x 256 tilings
    // lost their round-1 A/Bs too: HISTORY.md)
    if (H % 512 == 0) return launch_forward_cfg<ACT, 2, 8, 2, 4>(a, layer2, st);
    if (H == 256)     return launch_forward_cfg<ACT, 2, 8, 2, 4>(a, layer2, st);   // 256 h x 128 n
    if (H == 128)     return launch_forward_cfg<ACT, 1, 8, 4, 4>(a, layer2, st);   // 128 h x 256 n
    if (H == 64)      return launch_forward_cfg<ACT, 1, 4, 4, 4>(a, layer2, st);   //  64 h x 256 n
    if (H == 32)      return launch_forward_cfg<ACT, 1, 2, 4, 4>(a, layer2, st);   //  32 h x 256 n
    return RBNN_ERR_SHAPE;
}

int launch_forward(int act, const FwdArgs& a, bool layer2, hipStream_t st) {
    switch (act) {
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_RELU:  return launch_forward_act<RBNN_ACT_RELU>(a, layer2, st);
#endif
        case RBNN_ACT_LEAKY: return launch_forward_act<RBNN_ACT_LEAKY>(a, layer2, st);
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_SIGM:  return launch_forward_act<RBNN_ACT_SIGM>(a, layer2, st);
        case RBNN_ACT_TANH:  return launch_forward_act<RBNN_ACT_TANH>(a, layer2, st);
#endif
    }
    return RBNN_ERR_UNSUPPORTED;
}

template <int ACT, int TD, bool A_MEM, bool PER_SAMPLE>
int launch_grad_c(const GradArgs& a, hipStream_t st) {
    const int grid = grid_for_items((long long)a.NT * a.ND * a.nchunks);
    if constexpr (A_MEM) {
        hipLaunchKernelGGL((fc_grad_kernel<ACT, TD, 1, true, PER_SAMPLE, 32>), dim3(grid), dim3(256), 0, st, a);
    } else {                          // CQ = MFMA K steps (4 classes each) of the dA product
        // (64-unit stages — template parameter HSTG — halve the per-stage overhead but spill: 256 VGPR + 124 B scratch,
        //  7.31 ms vs 6.79 ms at C2, so only the 32-unit stage is dispatched.)
#ifdef RBNN_GRAD_STAGE64
        if (!PER_SAMPLE && TD == 7 && a.C > 4 && a.C <= 12 && a.H % 64 == 0) {
            hipLaunchKernelGGL((fc_grad_kernel<ACT, TD, 3, false, PER_SAMPLE, 64>), dim3(grid), dim3(256), 0, st, a);
            return launch_status();
        }
#endif
        if (a.C <= 4)       hipLaunchKernelGGL((fc_grad_kernel<ACT, TD, 1, false, PER_SAMPLE, 32>), dim3(grid), dim3(256), 0, st, a);
        else if (a.C <= 12) hipLaunchKernelGGL((fc_grad_kernel<ACT, TD, 3, false, PER_SAMPLE, 32>), dim3(grid), dim3(256), 0, st, a);
        else                hipLaunchKernelGGL((fc_grad_kernel<ACT, TD, 4, false, PER_SAMPLE, 32>), dim3(grid), dim3(256), 0, st, a);
    }
    return launch_status();
}

template <int ACT, bool A_MEM, bool PER_SAMPLE>
int launch_grad_td(GradArgs a, hipStream_t st) {
    const int TD = pick_td(a.Dt);
    a.NT = (a.N + 255) / 256;
    a.ND = (a.Dt + TD - 1) / TD;
    if (TD == 4) return launch_grad_c<ACT, 4, A_MEM, PER_SAMPLE>(a, st);
    if (TD == 7) return launch_grad_c<ACT, 7, A_MEM, PER_SAMPLE>(a, st);
    return launch_grad_c<ACT, 1, A_MEM, PER_SAMPLE>(a, st);
}

template <bool A_MEM, bool PER_SAMPLE>
int launch_grad(int act, const GradArgs& a, hipStream_t st) {
    switch (act) {
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_RELU:  return launch_grad_td<RBNN_ACT_RELU, A_MEM, PER_SAMPLE>(a, st);
#endif
        case RBNN_ACT_LEAKY: return launch_grad_td<RBNN_ACT_LEAKY, A_MEM, PER_SAMPLE>(a, st);
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_SIGM:  return launch_grad_td<RBNN_ACT_SIGM, A_MEM, PER_SAMPLE>(a, st);
        case RBNN_ACT_TANH:  return launch_grad_td<RBNN_ACT_TANH, A_MEM, PER_SAMPLE>(a, st);
#endif
    }
    return RBNN_ERR_UNSUPPORTED;
}

}  // namespace

// ===================================================================================================
// C-ABI
// ===================================================================================================
extern "C" {

int rbnn_abi_version(void) { return RBNN_ABI_VERSION; }

// 0 for a product build; bit 0: some translation unit of this library was compiled with a timing-only ablation switch (rbnn_common.hpp)
extern __attribute__((weak)) int rbnn_ablation_build_marker;
int rbnn_build_flags(void) { return (&rbnn_ablation_build_marker != nullptr) ? 1 : 0; }

const char* rbnn_strerror(int status) {
    switch (status) {
        case RBNN_OK: return "ok";
        case RBNN_ERR_NULL: return "required pointer is NULL";
        case RBNN_ERR_SHAPE: return "shape violates the padding contract (D_pad %16, hidden %32 and 2^k or k*512, classes <= 16)";
        case RBNN_ERR_UNSUPPORTED: return "unsupported architecture / activation / mode";
        case RBNN_ERR_LAUNCH: return "HIP kernel launch failed";
        case RBNN_ERR_ALIGN: return "pointer is not 16-byte aligned";
    }
    return "unknown status";
}

int rbnn_workspace_query(const rbnn_posterior* net, int32_t N, int32_t S, int32_t chunk, rbnn_workspace_sizes* out) {
    if (!net || !out) return RBNN_ERR_NULL;
    // weight pointers are not needed to size a workspace
    if (net->in_features < 1 || (net->in_stride & 15) || net->in_stride < net->in_features) return RBNN_ERR_SHAPE;
    if (net->hidden < 32 || (net->hidden & 31) || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    const size_t SN = (size_t)S * N, H = net->hidden, Dp = net->in_stride;
    const bool bm = is_bitmask(net->activation), fc2 = net->arch == RBNN_ARCH_FC2;
    if (chunk <= 0) chunk = pick_chunk(N, (int)(Dp / 16), S);
    if (chunk > S) chunk = S;
    rbnn_workspace_sizes z = {};
    z.P = z.dZ = SN * RBNN_CPAD * sizeof(float);
    z.mask1 = bm ? (size_t)S * (H / 32) * mask_ld(N) * sizeof(uint32_t) : 0;
    z.dact1 = bm ? 0 : SN * H * sizeof(float);
    if (fc2) {
        z.hid1 = z.dhid1 = SN * H * sizeof(float);
        z.mask2 = z.mask1;
        z.dact2 = z.dact1;
    }
    z.chunk = chunk;
    z.n_slabs = (S + chunk - 1) / chunk;
    z.slabs = (size_t)z.n_slabs * N * Dp * sizeof(float);
    *out = z;
    return RBNN_OK;
}

int rbnn_fc_forward(const rbnn_posterior* net, const float* X, int32_t ldx, int32_t N, const int32_t* sidx,
                    int32_t S, int32_t out_kind, const rbnn_workspace* ws, void* stream) {
    int rc = validate_net(net);
    if (rc) return rc;
    if (!X || !ws || !ws->P) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || ldx < net->in_stride || (ldx & 3)) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(X) || !aligned16(ws->P)) return RBNN_ERR_ALIGN;
    const bool bm = is_bitmask(net->activation), fc2 = net->arch == RBNN_ARCH_FC2;
    hipStream_t st = (hipStream_t)stream;
    const int H = net->hidden;

    FwdArgs a = {};
    a.X = X; a.x_sample_stride = 0; a.ldx = ldx; a.N = N;
    a.W = net->W1; a.w_sample_stride = (long long)H * net->in_stride; a.ldw = net->in_stride; a.KT = net->in_stride / 16;
    a.b = net->b1; a.W2 = net->W2; a.b2 = net->b2; a.C = net->n_classes; a.H = H;
    a.sidx = sidx; a.S = S; a.P = ws->P; a.mask = ws->mask1; a.dact = ws->dact1; a.hid = nullptr; a.out_kind = out_kind;
    if (!fc2) return launch_forward(net->activation, a, true, st);

    // fc2: layer 1 -> hid1 (+ stash 1), then layer 2 over the per-sample hidden rows -> P (+ stash 2)
    if (!ws->hid1 || (bm ? !ws->mask2 : !ws->dact2)) return RBNN_ERR_NULL;
    a.hid = ws->hid1;
    rc = launch_forward(net->activation, a, false, st);
    if (rc) return rc;
    FwdArgs b = a;
    b.X = ws->hid1; b.x_sample_stride = (long long)N * H; b.ldx = H;
    b.W = net->Wm; b.w_sample_stride = (long long)H * H; b.ldw = H; b.KT = H / 16;
    b.b = net->bm; b.mask = ws->mask2; b.dact = ws->dact2; b.hid = nullptr;
    return launch_forward(net->activation, b, true, st);
}

int rbnn_reduce_samples(const float* P, int32_t S, int32_t N, int32_t C, float scale, float* out, int32_t ldo, void* stream) {
    if (!P || !out) return RBNN_ERR_NULL;
    if (S < 1 || N < 1 || C < 1 || C > RBNN_CPAD || ldo < C) return RBNN_ERR_SHAPE;
    hipLaunchKernelGGL(reduce_samples_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, S, N, C, scale, out, ldo);
    return launch_status();
}

int rbnn_loss_dlogits(int32_t mode, const float* P, const float* Psum, int32_t ldp, const float* G_up, const int32_t* labels,
                      int32_t S, float inv_S, int32_t N, int32_t C, float* dZ, void* stream) {
    if (!P || !dZ) return RBNN_ERR_NULL;
    if (mode < RBNN_LOSS_MEAN_PROB || mode > RBNN_LOSS_UPSTREAM_LOGIT) return RBNN_ERR_UNSUPPORTED;
    if ((mode == RBNN_LOSS_UPSTREAM || mode == RBNN_LOSS_UPSTREAM_LOGIT) ? !G_up : !labels) return RBNN_ERR_NULL;
    if ((mode == RBNN_LOSS_MEAN_PROB || mode == RBNN_LOSS_MEAN_LOGIT) && !Psum) return RBNN_ERR_NULL;
    if (S < 1 || N < 1 || C < 1 || C > RBNN_CPAD || ldp < C) return RBNN_ERR_SHAPE;
    const long long total = (long long)S * N;
    hipLaunchKernelGGL(loss_dlogits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mode, P, Psum, ldp, G_up, labels, S, inv_S, N, C, dZ);
    return launch_status();
}

int rbnn_fc_input_grad(const rbnn_posterior* net, const int32_t* sidx, int32_t S, int32_t N, int32_t chunk,
                       const rbnn_workspace* ws, int32_t* n_slabs_out, void* stream) {
    int rc = validate_net(net);
    if (rc) return rc;
    if (!ws || !ws->dZ || !ws->slabs) return RBNN_ERR_NULL;
    if (N < 1 || S < 1) return RBNN_ERR_SHAPE;
    const bool bm = is_bitmask(net->activation), fc2 = net->arch == RBNN_ARCH_FC2;
    if (bm ? !ws->mask1 : !ws->dact1) return RBNN_ERR_NULL;
    if (!net->W1_pack4 || (fc2 && !net->Wm_pack4)) return RBNN_ERR_NULL;
    if (!aligned16(net->W1_pack4) || (fc2 && !aligned16(net->Wm_pack4))) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int H = net->hidden, Dp = net->in_stride;
    if (chunk <= 0) chunk = pick_chunk(N, Dp / 16, S);
    if (chunk > S) chunk = S;
    const int nchunks = (S + chunk - 1) / chunk;
    if (n_slabs_out) *n_slabs_out = nchunks;

    GradArgs g = {};
    g.W2 = net->W2; g.C = net->n_classes; g.H = H; g.HW = H / 32; g.sidx = sidx; g.S = S; g.N = N;
    if (!fc2) {
        g.dZ = ws->dZ; g.mask = ws->mask1; g.dact = ws->dact1;
        g.W1 = net->W1_pack4; g.w1_sample_stride = (long long)H * Dp; g.ldw = Dp; g.Dt = Dp / 16;
        g.chunk = chunk; g.nchunks = nchunks; g.out = ws->slabs; g.ldo = Dp;
        return launch_grad<false, false>(net->activation, g, st);
    }
    // fc2 step 1, per sample: dhid1[s] = act'(A1_s) * ((act'(A2_s) * (dZ_s . W2_s)) . Wm_s)
    if (!ws->dhid1 || (bm ? !ws->mask2 : !ws->dact2)) return RBNN_ERR_NULL;
    g.dZ = ws->dZ; g.mask = ws->mask2; g.dact = ws->dact2;
    g.W1 = net->Wm_pack4; g.w1_sample_stride = (long long)H * H; g.ldw = H; g.Dt = H / 16;
    g.chunk = 1; g.nchunks = S; g.out = ws->dhid1; g.ldo = H;
    g.omask = ws->mask1; g.odact = ws->dact1; g.OHW = H / 32;
    rc = launch_grad<false, true>(net->activation, g, st);
    if (rc) return rc;
    // fc2 step 2: slabs[k] = sum_{s in chunk k} dhid1[s] . W1_s
    GradArgs h = {};
    h.amem = ws->dhid1; h.C = net->n_classes; h.H = H; h.HW = H / 32; h.sidx = sidx; h.S = S; h.N = N;
    h.W1 = net->W1_pack4; h.w1_sample_stride = (long long)H * Dp; h.ldw = Dp; h.Dt = Dp / 16;
    h.chunk = chunk; h.nchunks = nchunks; h.out = ws->slabs; h.ldo = Dp;
    return launch_grad<true, false>(net->activation, h, st);
}

int rbnn_sum_slabs(const float* slabs, int32_t K, int32_t N, int32_t d_pad, float scale, float* out, int32_t ldo, void* stream) {
    if (!slabs || !out) return RBNN_ERR_NULL;
    if (K < 1 || N < 1 || d_pad < 4 || (d_pad & 3) || ldo < d_pad || (ldo & 3)) return RBNN_ERR_SHAPE;
    if (!aligned16(slabs) || !aligned16(out)) return RBNN_ERR_ALIGN;
    const long long total4 = (long long)N * d_pad / 4;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       slabs, K, (long long)N * d_pad, total4, d_pad / 4, scale, out, ldo);
    return launch_status();
}

int rbnn_sum_slabs_norms(const float* slabs, int32_t K, int32_t N, int32_t d_pad, int32_t D, float scale, float* out, int32_t ldo,
                         float* linf, float* l2, void* stream) {
    if (!slabs || !out || !linf || !l2) return RBNN_ERR_NULL;
    if (K < 1 || N < 1 || d_pad < 4 || (d_pad & 3) || ldo < d_pad || (ldo & 3) || D < 1 || D > d_pad) return RBNN_ERR_SHAPE;
    if (!aligned16(slabs) || !aligned16(out)) return RBNN_ERR_ALIGN;
    hipLaunchKernelGGL(sum_slabs_norms_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       slabs, K, (long long)N * d_pad, N, d_pad, D, scale, out, ldo, linf, l2);
    return launch_status();
}

int rbnn_pgd_alpha(const float* X0, int32_t ldx, int32_t N, int32_t D, float* alpha, void* stream) {
    if (!X0 || !alpha) return RBNN_ERR_NULL;
    if (N < 1 || D < 1 || ldx < D) return RBNN_ERR_SHAPE;
    hipLaunchKernelGGL(pgd_alpha_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, X0, ldx, N, D, alpha);
    return launch_status();
}

int rbnn_attack_step(float* X, const float* X0, int32_t ldx, const float* G, int32_t K, size_t slab_stride, int32_t ldg,
                     const float* alpha, float alpha_scalar, float eps, int32_t project, int32_t N, int32_t D, void* stream) {
    if (!X || !G || (project && !X0)) return RBNN_ERR_NULL;
    if (N < 1 || D < 1 || ldx < D || ldg < D || K < 1) return RBNN_ERR_SHAPE;
    if (!(D & 3) && !(ldx & 3) && !(ldg & 3) && !(slab_stride & 3) && aligned16(X) && aligned16(G) && (!project || aligned16(X0))) {
        const long long total4 = (long long)N * (D / 4);
        hipLaunchKernelGGL(attack_step4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           X, X0, ldx, G, K, (long long)slab_stride, ldg, alpha, alpha_scalar, eps, project, N, D / 4);
        return launch_status();
    }
    const long long total = (long long)N * D;
    hipLaunchKernelGGL(attack_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       X, X0, ldx, G, K, (long long)slab_stride, ldg, alpha, alpha_scalar, eps, project, N, D);
    return launch_status();
}

int rbnn_eval_metrics(const float* out_orig, const float* out_adv, int32_t ldp, const int32_t* labels, int32_t N, int32_t C,
                      int32_t* counts, float* rob, void* stream) {
    if (!out_orig || !out_adv || !labels || !counts || !rob) return RBNN_ERR_NULL;
    if (N < 1 || C < 1 || ldp < C) return RBNN_ERR_SHAPE;
    if (hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL(eval_metrics_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       out_orig, out_adv, ldp, labels, N, C, counts, rob);
    return launch_status();
}

int rbnn_pack_rows4(const float* W, int64_t rows, int32_t cols, float* out, void* stream) {
    if (!W || !out) return RBNN_ERR_NULL;
    if (rows < 4 || (rows & 3) || cols < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(out)) return RBNN_ERR_ALIGN;
    const long long total = (rows / 4) * (long long)cols;
    hipLaunchKernelGGL(pack_rows4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, (long long)(rows / 4), cols, out);
    return launch_status();
}

int rbnn_svi_materialize(const float* loc, const float* scale_raw, const float* eps, int64_t n_elem, int32_t S, float* out, void* stream) {
    if (!loc || !scale_raw || !eps || !out) return RBNN_ERR_NULL;
    if (n_elem < 1 || S < 1) return RBNN_ERR_SHAPE;
    const long long total = (long long)n_elem * S;
    hipLaunchKernelGGL(svi_materialize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       loc, scale_raw, eps, (long long)n_elem, S, out);
    return launch_status();
}

}  // extern "C"
