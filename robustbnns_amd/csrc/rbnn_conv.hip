// rbnn_conv.hip — the reference's `conv` architecture (model_nn.py:93-106) on gfx950:
//   Conv2d(Cin,32,5) -> act -> MaxPool2d(2) -> Conv2d(32,Hc,5) -> act -> MaxPool2d(2, stride 1) -> Flatten -> Linear(P2W^2*Hc, C)
// Two input geometries are instantiated (struct Geo): 1x28x28 — the only one the reference's head is correct for
// (model_nn.py:95-96,106: pinned by the reference-generated fixtures) — and 3x32x32, BASELINE.json's CIFAR-shaped config 5,
// whose head size 81*Hc is BUILD-DEFINED (the reference cannot express it: parity unpinned, checked against the fp64 oracle).
//
// Per posterior sample the weights differ, so every layer is batched over (sample, point):
//   conv1_pool_kernel   25*Cin-tap conv + 2x2 max-pool + activation on the VALU (~2 % of the MACs)  -> P1 [S][N][32][P1W][P1W]
//   conv2_pool_kernel   the 98 % of the work: implicit GEMM on v_mfma_f32_16x16x4_f32,
//                       O2^T[hc][pos] = sum_k W2[hc][k] * P1[ci(k)][y(pos)+ky(k)][x(pos)+kx(k)],  k = (ci,ky,kx) in 0..799;
//                       A = W2 rows through an LDS-DMA tile ring, B gathered from the point's P1 image resident in LDS
//                       with a k -> offset table; epilogue: bias, 2x2/stride-1 max-pool, activation, 3-bit stash (argmax,
//                       sign) for the backward                                                       -> Q2 [S][N][Hc*NP2]
//   conv_fc_kernel      the skinny Linear(NP2*Hc -> C) as an MFMA with both operands K-contiguous in memory, + softmax
// relu / leaky: max-pool commutes with the monotone activation, so the pre-activation is pooled and activated once (exact).
// sigmoid / tanh: the pooling compares the ACTIVATED values, as torch does (two distinct pre-activations can round to the same
// activation, and then the first one wins); their backward takes act' from the stored activation value, no sign bit.
#include "rbnn_common.hpp"

// build switches of the conv2^T dense kernel / conv1^T that more than one section reads (the rest sit next to their kernels)
#ifndef RBNN_DENSE_COL2IM_RMW
#define RBNN_DENSE_COL2IM_RMW 1
#endif
#ifndef RBNN_FCBWD_NT
#define RBNN_FCBWD_NT 1                                                    // conv_fc_bwd writes dQ2 (5 GB per C5 pass, re-read from HBM by the dense kernel in any case) with non-temporal stores:
#endif                                                                     // they no longer push the sample's Fw out of the L2 — 1.54 -> 1.25 ms (same box, alternating: backward call 16.3 -> 16.0 ms)
#ifndef RBNN_X3FWD_NT
#define RBNN_X3FWD_NT 0                                                    // the same for conv2_pool_x3_kernel's Q2 / stash stores
#endif
#ifndef RBNN_CONV1_BWD_X3_MINCIN
#define RBNN_CONV1_BWD_X3_MINCIN 2                                        // input channels from which conv1^T runs on the f16 pipe (1x28x28 keeps the fp32 kernel: see launch_conv1_backward)
#endif
#ifndef RBNN_CONV1_BWD_X3
#define RBNN_CONV1_BWD_X3 RBNN_DENSE_COL2IM_RMW                            // conv1^T behind the dense kernel on the f16 matrix pipe (triple-split), scaled by the dense kernel's max |dP1|
#endif

namespace {

constexpr int C1 = 32;                                                 // conv1 output channels (model_nn.py:99)
template <int CIN_, int IW_> struct Geo {
    static constexpr int CIN = CIN_, IW = IW_, DIN = CIN_ * IW_ * IW_;    // input channels, (square) width, flattened size
    static constexpr int O1 = IW_ - 4, P1W = O1 / 2, P1SZ = C1 * P1W * P1W;   // conv1 output width, pooled width, floats per point
    static constexpr int O2W = P1W - 4, P2W = O2W - 1, NPOS = O2W * O2W, NP2 = P2W * P2W;   // conv2 output, stride-1 pooled
    static constexpr int NPT2 = (NPOS + 15) / 16, NPT1 = (P1W * P1W + 15) / 16;   // 16-position MFMA tiles of conv2's output / of dP1
    static constexpr int K1 = CIN_ * 25;
    static constexpr int P1STRIDE = (P1SZ * 4 > 24576 ? (P1SZ + 255) / 256 * 1024 : 24576);   // bytes of ws->P1 ALLOCATED per (s, n): the fp32 image rounded up to whole 1-KiB DMA pieces, >= the 24 KiB split image
    // zero-padded conv2-gradient image of the backward: every row is [4 zero columns][O2W gradients], so the pitch equals P1W and
    // a row's 4-column RIGHT border is the next row's left border (rows follow each other without a gap: a run of 16 output
    // positions reads 16 consecutive floats); a channel is O2W such rows + 4 zero rows, shared as the next channel's top border,
    // + CHPAD floats that make the channel stride = 4 (mod 8): lanes lg and lg+1 of a ds_read_b32 half (channels 4 apart) then
    // sit 16 banks apart — the gather is bank-conflict-free (measured before: 49 % conflict cycles)
    static constexpr int PITCH = O2W + 4;
    static constexpr int CHPAD = (4 - (PITCH * PITCH) % 8 + 8) % 8;
    static constexpr int CHS = PITCH * PITCH + CHPAD;
};
using GeoMnist = Geo<1, 28>;      // O1 24, P1W 12, O2W 8,  P2W 7, NPOS 64,  NP2 49, NPT2 4, NPT1 9
using GeoCifar = Geo<3, 32>;      // O1 28, P1W 14, O2W 10, P2W 9, NPOS 100, NP2 81, NPT2 7, NPT1 13
// the split-half kernels further down are built for the 1x28x28 geometry only
constexpr int P1W = GeoMnist::P1W, P1SZ = GeoMnist::P1SZ, K2 = C1 * 25, O2W = GeoMnist::O2W, P2W = GeoMnist::P2W,
              NPOS = GeoMnist::NPOS, NP2 = GeoMnist::NP2;

template <int ACT> constexpr bool smooth_act() { return ACT == RBNN_ACT_SIGM || ACT == RBNN_ACT_TANH; }

struct ConvArgs {
    const float* X; int ldx; int N;
    const float* K1w; const float* K1b;            // [S_total][32][Cin*25], [S_total][32]
    const float* K2w; const float* K2b;            // [S_total][Hc][800], [S_total][Hc]
    const float* Fw;  const float* Fb;             // [S_total][C][NP2*Hc], [S_total][C]
    int Hc; int C; const int* sidx; int S;
    float* P1; uint8_t* st1;                       // [S][N][P1SZ] (dense fp32 image; the buffer is allocated P1STRIDE bytes per point)
    float* Q2; uint8_t* st2;                       // [S][N][Hc*NP2]
    float* P; int out_kind;
    int NP2;                                       // pooled conv2 positions per channel (conv_fc / conv_fc_bwd are geometry-agnostic)
};

// ---------------------------------------------------------------------------------------------------
// One block = one (sample, point): the sample's 32 x Cin x 25 weights and the point's image go to LDS once, then one thread per
// pooled POSITION keeps its Cin x 6 x 6 input patch in registers and runs all 32 output channels over it (weights are LDS
// broadcasts: ~16 FMAs per LDS read).  The first version (one thread per pooled output, patch and weights re-read from memory
// by every thread: 1.6 FMAs per load) took 7.9 ms per pass at the CIFAR-shaped c5 bench.  Same accumulation order
// (input channel, ky, kx), so the results are bit-identical to it.
// Two threads per pooled position (16 channels each): 144 / 196 positions alone left 44 % / 23 % of a 256-thread block's lanes idle in
// the FMA loop (0.52 -> see profiles/r03a/conv_small_kernels.txt).  Each output is still produced by one thread in the same order.
// (3x32x32: 196 positions fill a 256-thread block well enough, and the two-halves form measured slower there: 2.45 -> 2.66 ms at c5)
template <class G> constexpr int conv1_halves() { return G::CIN == 1 ? 2 : 1; }
template <class G> constexpr int conv1_threads() { return (conv1_halves<G>() * G::P1W * G::P1W + 63) / 64 * 64; }
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef RBNN_CONV1_W128
#define RBNN_CONV1_W128 1
#endif
#ifndef RBNN_CONV1_WAVES
#define RBNN_CONV1_WAVES 1
#endif
template <int ACT, class G>
__global__ void __launch_bounds__(conv1_threads<G>(), RBNN_CONV1_WAVES) conv1_pool_kernel(const ConvArgs a) {
    constexpr int NPP = G::P1W * G::P1W, IW = G::IW, NTH = conv1_threads<G>();
    // Weights in LDS as CHANNEL PAIRS: [c / 2][ci][tap][2] (52 floats per (pair, ci): thirteen aligned float4).  The FMA loop then runs on
    // v_pk_fma_f32 — one instruction = the same tap of two output channels against one broadcast patch value — i.e. at the packed fp32 rate
    // (round 4; the un-packed loop ran at 0.73 of the plain FMA rate: 2.09 ms per C5 pass).  Every output is still one fp32 FMA chain in
    // the same (input channel, ky, kx) order: results are bit-identical.
    constexpr int WROW = 52;
    __shared__ __attribute__((aligned(16))) float wsh[(C1 / 2) * G::CIN * WROW];
    __shared__ __attribute__((aligned(8))) float xsh[G::DIN];
    static_assert(IW % 2 == 0, "patch pairs are 8-byte aligned");
    const long long sn = blockIdx.x;
    const int n = (int)(sn % a.N), s = (int)(sn / a.N), tid = threadIdx.x;
    const int sw = a.sidx ? a.sidx[s] : s;
    for (int e = tid; e < C1 * G::K1; e += NTH) {                        // e = (c, ci, tap) as nn.Conv2d stores it
        const int c = e / G::K1, k = e - c * G::K1;
        wsh[((c >> 1) * G::CIN + k / 25) * WROW + 2 * (k % 25) + (c & 1)] = a.K1w[(long long)sw * C1 * G::K1 + e];
    }
    for (int e = tid; e < G::DIN; e += NTH) xsh[e] = a.X[(long long)n * a.ldx + e];
    __syncthreads();
    constexpr int NH = conv1_halves<G>(), CPT = C1 / NH;                  // channels per thread
    if (tid >= NH * NPP) return;
    const int pos = tid % NPP, c0 = (tid / NPP) * CPT;
    const int py = pos / G::P1W, px = pos % G::P1W;
    // the 6 x 6 patch as 18 aligned PAIRS per input channel: v_pk_fma_f32 broadcasts either half of a pair to both lanes (op_sel), so a patch
    // value costs half a register pair — spelled as a scalar broadcast the compiler kept every value twice (216 registers, two waves per SIMD)
    f32x2 patch[G::CIN][6][3];
#pragma unroll
    for (int ci = 0; ci < G::CIN; ++ci)
#pragma unroll
        for (int y = 0; y < 6; ++y)
#pragma unroll
            for (int j = 0; j < 3; ++j) patch[ci][y][j] = *(const f32x2*)(xsh + ci * (IW * IW) + (2 * py + y) * IW + 2 * px + 2 * j);
    float* const p1 = a.P1 + sn * G::P1SZ + pos;                         // dense [S][N][32][P1W][P1W]
    uint8_t* const st = a.st1 + sn * G::P1SZ + pos;
#pragma unroll 1
    for (int cp = c0 / 2; cp < (c0 + CPT) / 2; ++cp) {
        f32x2 v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v4[q] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < G::CIN; ++ci) {                             // channel-major accumulation, taps in (ky, kx) order
            asm volatile("" ::: "memory");                               // keep the three input channels' weight reads from being hoisted together (156 registers)
            const float* const wrow = wsh + (cp * G::CIN + ci) * WROW;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int tap = ky * 5 + kx;
#if RBNN_CONV1_W128
                    const f32x4 w4 = *(const f32x4*)(wrow + 4 * (tap >> 1));
                    const f32x2 wv = (tap & 1) ? (f32x2){w4[2], w4[3]} : (f32x2){w4[0], w4[1]};
#else
                    const f32x2 wv = *(const f32x2*)(wrow + 2 * tap);     // (a 16-byte read per tap pair keeps 52 more registers live: 252 VGPRs, two waves per SIMD)
#endif
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            // acc.{lo,hi} += w.{lo,hi} * pair.{half}: the broadcast is the instruction's op_sel (written as a vector splat,
                            // the compiler hoisted 108 materialised (p, p) pairs out of the channel loop: 252 registers)
                            const f32x2 pp = patch[ci][dy + ky][(dx + kx) >> 1];
                            if ((dx + kx) & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(v4[dy * 2 + dx]) : "v"(wv), "v"(pp));
                            else               asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(v4[dy * 2 + dx]) : "v"(wv), "v"(pp));
                        }
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * cp + h;
            const float b = a.K1b[(long long)sw * C1 + c];
            float best = 0.f, best_pre = 0.f;
            int arg = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pre = v4[q][h] + b, v = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                if (q == 0 || v > best) { best = v; best_pre = pre; arg = q; }   // first maximum wins (torch max_pool2d)
            }
            p1[c * NPP] = smooth_act<ACT>() ? best : act_fwd<ACT>(best);
            st[c * NPP] = (uint8_t)(arg | (best_pre > 0.f ? 4 : 0));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// conv2: one block = one (sample, point); 4 waves, each 64 output channels x the point's NPOS output positions
// (4 x NPT2 accumulator tiles), output channels in chunks of 256.  Dynamic LDS: the point's P1 image, the k -> offset table
// and two weight stage tiles; the per-wave pooling tiles of the epilogue alias the (then idle) weight tiles.
// 1x28x28: 53 KiB, 3x32x32: 60 KiB — two blocks per CU either way.
template <class G> constexpr int conv2_lds_floats() { return (G::P1SZ + 255) / 256 * 256 + 800 + 2 * 256 * 16; }

template <int ACT, class G>
__global__ void __launch_bounds__(256, 2) conv2_pool_kernel(const ConvArgs a) {
    constexpr int WROWS = 256, TILE = WROWS * 16, NPT = G::NPT2;
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NPOS_ = G::NPOS, NP2_ = G::NP2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P1LDS = (G::P1SZ + 255) / 256 * 256;
    static_assert(4 * 16 * G::NPOS <= 2 * TILE, "the four waves' pooling tiles fit in the weight stage buffers they alias");
    float* const P1s = lds;
    int* const koff = (int*)(lds + P1LDS);
    float* const Wt = lds + P1LDS + 800;
    float* const scr = Wt;                                                // epilogue only: every wave is past the K loop's last barrier

    int id;
    if (!item_of_block(blockIdx.x, a.N * a.S, id)) return;
    const int n = id % a.N, s = id / a.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    const float* const Ws = a.K2w + (long long)sw * a.Hc * K2;
    const int F = a.Hc * NP2_;

    // the point's pooled conv1 image (dense [S][N][P1SZ] floats) -> LDS in whole 1-KiB pieces, and the k -> image offset table.
    // A ragged last piece (3x32x32: 24.5 pieces) reads up to 512 B past the point's image — into the next point's, or into the
    // slack of the allocation (ws->P1 is sized P1STRIDE >= P1SZ*4 + 512 bytes per point); those floats are never addressed.
    for (int q = wave; q < (G::P1SZ + 255) / 256; q += 4) glds16(a.P1 + sn * G::P1SZ + q * 256 + 4 * lane, P1s + q * 256);
    for (int k = tid; k < K2; k += 256) koff[k] = (k / 25) * (P1W_ * P1W_) + ((k % 25) / 5) * P1W_ + (k % 5);
    const int prow = lane >> 2, lchunk = (lane & 3) ^ swz(prow), pch = 4 * (lg ^ swz(li));
    int poff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {                                    // position pt*16+li = (y, x); positions past NPOS read (0,0), never stored
        const int pos = pt * 16 + li;
        poff[pt] = pos < NPOS_ ? (pos / O2W_) * P1W_ + pos % O2W_ : 0;
    }

    for (int hc0 = 0; hc0 < a.Hc; hc0 += WROWS) {
        f32x4 acc[4][NPT];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto stage = [&](int kt, int buf) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int q = wave + 4 * q4, row = min(hc0 + q * 16 + prow, a.Hc - 1);   // rows past Hc repeat the last channel; never stored
                glds16(Ws + (long long)row * K2 + kt * 16 + 4 * lchunk, Wt + buf * TILE + q * 256);
            }
        };
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < K2 / 16; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < K2 / 16) stage(kt + 1, buf ^ 1);
            const float* const W = Wt + buf * TILE;
            const int kq0 = koff[kt * 16 + 4 * lg], kq1 = koff[kt * 16 + 4 * lg + 1], kq2 = koff[kt * 16 + 4 * lg + 2], kq3 = koff[kt * 16 + 4 * lg + 3];
            f32x4 b[NPT], af[4];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
                b[pt] = (f32x4){P1s[kq0 + poff[pt]], P1s[kq1 + poff[pt]], P1s[kq2 + poff[pt]], P1s[kq3 + poff[pt]]};
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) af[ht] = *(const f32x4*)(W + ((wave * 4 + ht) * 16 + li) * 16 + pch);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA16(af[ht][r], b[pt][r], acc[ht][pt]);
            __syncthreads();
        }
        // epilogue: bias, 2x2 / stride-1 max-pool through a per-wave LDS tile, activation, stash
        float* const my = scr + wave * 16 * NPOS_;
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const int hcb = hc0 + (wave * 4 + ht) * 16;                    // wave-uniform
            if (hcb >= a.Hc) break;
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pt * 16 + li < NPOS_) {
                        const float pre = acc[ht][pt][r] + bias[r];
                        // smooth activations are pooled on their VALUES (sign bit 30 of the stored float is free for them: the
                        // pre-activation's sign rides there only for relu / leaky, which keep the pre-activation itself)
                        my[(4 * lg + r) * NPOS_ + pt * 16 + li] = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                    }
            for (int idx = lane; idx < 16 * NP2_; idx += 64) {
                const int hl = idx / NP2_, p = idx % NP2_, base = hl * NPOS_ + (p / P2W_) * O2W_ + (p % P2W_);
                float best = my[base];
                int arg = 0;
                if (my[base + 1] > best) { best = my[base + 1]; arg = 1; }
                if (my[base + O2W_] > best) { best = my[base + O2W_]; arg = 2; }
                if (my[base + O2W_ + 1] > best) { best = my[base + O2W_ + 1]; arg = 3; }
                const long long o = sn * F + (long long)(hcb + hl) * NP2_ + p;
                a.Q2[o] = smooth_act<ACT>() ? best : act_fwd<ACT>(best);
                a.st2[o] = (uint8_t)(arg | (best > 0.f ? 4 : 0));
            }
        }
        if (hc0 + WROWS < a.Hc) __syncthreads();                           // the pooling tiles alias the next chunk's first weight tile
    }
}

// ---------------------------------------------------------------------------------------------------
// Linear(NP2*Hc -> C) + softmax: one wave = 16 points of one sample.  D[i = class][j = point] = sum_f Fw[c][f] * Q2[n][f];
// both operands are read straight from memory, 16 bytes (4 K steps) per lane per load.
__global__ void __launch_bounds__(256) conv_fc_kernel(const ConvArgs a) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int NT = (a.N + 15) / 16;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= NT * a.S) return;
    const int s = item / NT, n0 = (item % NT) * 16;
    const int sw = a.sidx ? a.sidx[s] : s;
    const int F = a.Hc * a.NP2;
    const int n = min(n0 + li, a.N - 1), c = min(li, a.C - 1);
    const float* const fw = a.Fw + ((long long)sw * a.C + c) * F + 4 * lg;
    const float* const q2 = a.Q2 + ((long long)s * a.N + n) * F + 4 * lg;
    const float cmask = li < a.C ? 1.f : 0.f;
    // F = NP2*Hc is a multiple of 16 (Hc is).  Four interleaved accumulation chains: back-to-back MFMAs on ONE accumulator would
    // serialise on its latency, and a single fp32 chain over K = 25 088 .. 82 944 terms also carries 2x the rounding error.
    f32x4 zq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int f = 0;
    for (; f + 64 <= F; f += 64) {
        f32x4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = *(const f32x4*)(fw + f + 16 * q) * cmask; bv[q] = *(const f32x4*)(q2 + f + 16 * q); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = MFMA16(av[q][r], bv[q][r], zq[q]);
    }
    for (; f < F; f += 16) {
        const f32x4 av = *(const f32x4*)(fw + f) * cmask, bv = *(const f32x4*)(q2 + f);
#pragma unroll
        for (int r = 0; r < 4; ++r) zq[0] = MFMA16(av[r], bv[r], zq[0]);
    }
    f32x4 z = (zq[0] + zq[1]) + (zq[2] + zq[3]);
    // lane (point li, lg) holds classes 4*lg + r
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cc = 4 * lg + r;
        z[r] = cc < a.C ? z[r] + a.Fb[(long long)sw * a.C + cc] : -INFINITY;
        m = fmaxf(m, z[r]);
    }
    if (a.out_kind == RBNN_OUT_PROBS) {
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { z[r] = 4 * lg + r < a.C ? expf(z[r] - m) : 0.f; den += z[r]; }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = z[r] / den;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * lg + r >= a.C) z[r] = 0.f;
    }
    if (n0 + li < a.N) *(f32x4*)(a.P + ((long long)s * a.N + n0 + li) * RBNN_CPAD + 4 * lg) = z;
}

// =====================================================================================================
// Split-half precision forward (the technique of rbnn_split.hip applied to conv2, 98 % of the MACs).
//   conv1_pool_split_kernel  as conv1_pool_kernel, but one thread per pooled POSITION computing all 32 channels, so that it
//                            can write the point's image channel-last as fp16 hi / lo:
//                              P1s[s][n][plane hi|lo][y 0..11][x pitch 16][ci 32]   (2 x 12 KiB; value * 2^p1_exp = hi + lo)
//                            with 16-byte channel octet o stored at o ^ (((16y + x) >> 2 & 1) << 1)  (bank swizzle, below).
//   conv2_pool_split_kernel  K runs tap-major: k = tap*32 + ci, so ONE v_mfma_f32_16x16x32_f16 K step is one tap over the 32
//                            input channels and the gathered B operand of a lane (position li, channel octet lg) is ONE
//                            ds_read_b128 of the image (hi) and one of the lo plane — no k -> offset table, no scalar gathers.
//                            A = model.3.weight regrouped [hc][tap][ci] as a split-rows image (rbnn_split_rows), staged per
//                            tap through the same swizzled 128-B-row LDS-DMA ring as fc_forward_split_kernel.
//                            Block = 8 waves = 4 channel quarters (64 hc) x 2 points of one sample; channels in chunks of
//                            256; both points' images stay resident in LDS (48 KiB) and share every weight tile.
//                            Bank rule: a ds_read_b128 group is lanes {0-3,12-15} of octet lg with lanes {4-11} of octet
//                            lg^1; with the x pitch of 16 those two lane sets sit on 8 distinct positions mod 8 each, and
//                            the octet swizzle above separates (p mod 4) twins: conflict-free for every tap.
// =====================================================================================================
constexpr int P1PITCH = 16, P1PLANE = P1W * P1PITCH * C1 * 2;            // bytes of one (hi or lo) plane: 12 x 16 x 32 halves
constexpr int P1SPLIT = 2 * P1PLANE;                                     // 24 KiB per (sample, point)
static_assert(P1SPLIT == 24576, "split image size");

struct ConvSplitArgs {
    const char* K2r; int k2_exp; int p1_exp;                             // split-rows image of [S_total*Hc][25*32]
    const rbnn_dev_scale* p1_ds;                                         // != NULL: the P1 scale lives on the device (rbnn_input_scales record [1])
    char* P1s;                                                           // [S][N][P1SPLIT]
};

template <int ACT>
__global__ void __launch_bounds__(256) conv1_pool_split_kernel(const ConvArgs a, const ConvSplitArgs sp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one thread per pooled position (s, n, py, px)
    if (i >= (long long)a.S * a.N * (P1W * P1W)) return;
    const int pp = (int)(i % (P1W * P1W)), py = pp / P1W, px = pp % P1W;
    const long long sn = i / (P1W * P1W);
    const int n = (int)(sn % a.N), s = (int)(sn / a.N);
    const int sw = a.sidx ? a.sidx[s] : s;
    const float* const x = a.X + (long long)n * a.ldx + (2 * py) * 28 + 2 * px;
    float patch[6][6];
#pragma unroll
    for (int y = 0; y < 6; ++y)
#pragma unroll
        for (int xx = 0; xx < 6; ++xx) patch[y][xx] = x[y * 28 + xx];
    const float scale = sp.p1_ds ? sp.p1_ds->scale : ldexpf(1.f, sp.p1_exp);
    const int p = py * P1PITCH + px, osw = ((p >> 2) & 1) << 1;
    char* const dst = sp.P1s + sn * P1SPLIT + (long long)p * 64;
    for (int o = 0; o < 4; ++o) {                                         // 8 channels -> one 16-byte octet of hi and of lo
        union { f16x8 v; uint4 u; } hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * o + j;
            const float* const w = a.K1w + ((long long)sw * C1 + c) * 25;
            const float b = a.K1b[(long long)sw * C1 + c];
            float best = 0.f;
            int arg = 0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) v = fmaf(w[ky * 5 + kx], patch[dy + ky][dx + kx], v);
                    v += b;
                    if ((dy == 0 && dx == 0) || v > best) { best = v; arg = dy * 2 + dx; }
                }
            a.st1[sn * P1SZ + c * (P1W * P1W) + pp] = (uint8_t)(arg | (best > 0.f ? 4 : 0));
            const float v = act_fwd<ACT>(best) * scale;
            const _Float16 h = (_Float16)v;
            hi.v[j] = h;
            lo.v[j] = (_Float16)(v - (float)h);
        }
        *(uint4*)(dst + ((o ^ osw) * 16)) = hi.u;
        *(uint4*)(dst + P1PLANE + ((o ^ osw) * 16)) = lo.u;
    }
}

template <int ACT>
__global__ void __launch_bounds__(512, 2) conv2_pool_split_kernel(const ConvArgs a, const ConvSplitArgs sp) {
    constexpr int WROWS = 256, ROWB = 128, TILEB = WROWS * ROWB;          // weight stage tile: 256 channels x one tap (32 ci, hi + lo)
    constexpr int NW = 8, WP = WROWS / 8 / NW;                            // DMA pieces (8 rows each) per wave per stage
    constexpr int OFF_IMG = 2 * TILEB;                                    // two weight buffers, then the two points' images
    extern __shared__ __attribute__((aligned(16))) float lds[];           // OFF_IMG + 2 * P1SPLIT bytes
    char* const ldsb = (char*)lds;

    const int NB = (a.N + 1) / 2;                                         // blocks per sample
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, nb0 = (id % NB) * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int wq = wave & 3, wp = wave >> 2;                              // channel quarter, point of the pair
    const int sw = a.sidx ? a.sidx[s] : s;
    const int n = min(nb0 + wp, a.N - 1);                                 // a ragged last block computes its first point twice, stores once
    const bool live = nb0 + wp < a.N;
    const long long sn = (long long)s * a.N + n;
    const char* const Ws = sp.K2r + (long long)sw * a.Hc * (K2 * 4);
    const int F = a.Hc * NP2;
    const float out_scale = sp.p1_ds ? ldexpf(1.f, -sp.k2_exp) * sp.p1_ds->inv_scale : ldexpf(1.f, -(sp.k2_exp + sp.p1_exp));

    // both points' images -> LDS: 2 x 24 pieces of 1 KiB, linear
    for (int q = wave; q < 2 * (P1SPLIT / 1024); q += NW) {
        const int pt2 = q / (P1SPLIT / 1024), piece = q % (P1SPLIT / 1024);
        const long long src_sn = (long long)s * a.N + min(nb0 + pt2, a.N - 1);
        glds16((const float*)(sp.P1s + src_sn * P1SPLIT + piece * 1024 + lane * 16), (float*)(ldsb + OFF_IMG + pt2 * P1SPLIT + piece * 1024));
    }
    const int prow = lane >> 3;
    const int src_off = ((lane & 7) ^ row_swz(8 * (wave & 1) + prow)) * 16;
    const int foff = li * ROWB + (((2 * lg) ^ row_swz(li)) * 16), foff_lo = foff ^ 16;
    // image position of lane li in position tile pt: (y, x) = (2pt + li/8, li%8); per tap add ky*16 + kx
    int pbase[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) pbase[pt] = (2 * pt + (li >> 3)) * P1PITCH + (li & 7);
    const char* const img = ldsb + OFF_IMG + wp * P1SPLIT;

    for (int hc0 = 0; hc0 < a.Hc; hc0 += WROWS) {
        f32x4 acc[4][4];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto stage = [&](int tap, int buf) {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int q = wave + NW * i, row = min(hc0 + 8 * q + prow, a.Hc - 1);   // rows past Hc repeat the last channel; never stored
                glds16((const float*)(Ws + (long long)row * (K2 * 4) + tap * ROWB + src_off), (float*)(ldsb + buf * TILEB + q * 1024));
            }
        };
        stage(0, 0);
        ring_wait_barrier<0>();
        for (int tap = 0; tap < 25; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 25) stage(tap + 1, buf ^ 1);
            const char* const Wt = ldsb + buf * TILEB + (wq * 4) * 16 * ROWB;
            const int toff = (tap / 5) * P1PITCH + (tap % 5);
            f16x8 bh[4], bl[4], ah[4], al[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int p = pbase[pt] + toff;
                const int off = p * 64 + ((lg ^ (((p >> 2) & 1) << 1)) * 16);
                bh[pt] = *(const f16x8*)(img + off);
                bl[pt] = *(const f16x8*)(img + P1PLANE + off);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                ah[ht] = *(const f16x8*)(Wt + ht * 16 * ROWB + foff);
                al[ht] = *(const f16x8*)(Wt + ht * 16 * ROWB + foff_lo);
            }
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = MFMA_H(al[ht], bh[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = MFMA_H(ah[ht], bl[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = MFMA_H(ah[ht], bh[pt], acc[ht][pt]);
            }
            ring_wait_barrier<0>();                                      // tap+1's weights landed; everyone is done with this tile
        }
        // epilogue (as conv2_pool_kernel): scale, bias, 2x2 / stride-1 max-pool of the pre-activations through a per-wave LDS
        // tile (aliases weight buffer 0: every wave passed the barrier above), activation, stash
        float* const my = (float*)ldsb + wave * 16 * NPOS;
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const int hcb = hc0 + (wq * 4 + ht) * 16;                      // wave-uniform
            if (hcb >= a.Hc) break;
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r) my[(4 * lg + r) * NPOS + pt * 16 + li] = acc[ht][pt][r] * out_scale + bias[r];
            // four consecutive pooled cells per lane: one 16-byte store of Q2 and one 4-byte store of the stash (the tile's 16 x 49
            // cells are contiguous in both); 196 lane-items per tile
            for (int i4 = lane; i4 < 16 * NP2 / 4 && live; i4 += 64) {
                f32x4 q;
                unsigned stw = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = 4 * i4 + j, hl = idx / NP2, p = idx % NP2, base = hl * NPOS + (p / P2W) * O2W + (p % P2W);
                    float best = my[base];
                    int arg = 0;
                    if (my[base + 1] > best) { best = my[base + 1]; arg = 1; }
                    if (my[base + O2W] > best) { best = my[base + O2W]; arg = 2; }
                    if (my[base + O2W + 1] > best) { best = my[base + O2W + 1]; arg = 3; }
                    q[j] = act_fwd<ACT>(best);
                    stw |= (unsigned)(arg | (best > 0.f ? 4 : 0)) << (8 * j);
                }
                const long long o = sn * F + (long long)hcb * NP2 + 4 * i4;           // a multiple of 4
                *(f32x4*)(a.Q2 + o) = q;
                *(unsigned*)(a.st2 + o) = stw;
            }
        }
        __syncthreads();                                                 // the scratch aliases weight buffer 0 of the next chunk
    }
}

// =====================================================================================================
// Triple-split ("f16x6") conv2 forward: the technique of rbnn_triple.hip — every fp32 operand carried at FULL width as three
// fp16 pieces, six exact product terms per fp32 product on v_mfma_f32_16x16x32_f16, fp32 accumulation — on the layer that holds
// 98 % of the MACs, for BOTH geometries.  Structure = conv2_pool_split_kernel's (K tap-major: one K step = one tap over the 32
// input channels; the B operand of a lane is one ds_read_b128 per piece from the point's channel-last image resident in LDS):
//   * conv1 is the exact conv1_pool_kernel (fp32 P1 image, ~2 % of the MACs); each block splits its two points' images into
//     the three piece planes while loading them: img[point][piece][pos = y*P1W + x][32 ci] halves, 16-byte channel octet o at
//     o ^ (((pos >> 2) & 1) << 1), scaled by the device record of rbnn_input_scales (|P1| <= sum|K1w| * max|x| + max|K1b|);
//   * A = model.3.weight regrouped [hc][tap][ci] as a triple-rows image (rbnn_triple_rows, 25 K stages of 192 B per channel),
//     staged per tap through three plane tiles of 64-B rows (fc_forward_x3_kernel's ring and `swz` chunk swizzle);
//   * block = 8 waves = 4 channel groups (HTW tiles of 16 channels each) x 2 points; WROWS = 64 * HTW channels per chunk.
//     1x28x28: WROWS 256 (96 KB of weight tiles + 2 x 27 KB images); 3x32x32: WROWS 128 (48 KB + 2 x 37 KB).
// =====================================================================================================
struct ConvX3Args {
    const char* K2t; int k2_exp; int p1_exp;                             // triple-rows image of [S_total*Hc][25*32]
    const rbnn_dev_scale* p1_ds;                                         // != NULL: the P1 scale lives on the device (rbnn_input_scales record [1])
};

__device__ __forceinline__ void conv_split3(float v, _Float16& p0, _Float16& p1, _Float16& p2) {
    p0 = (_Float16)v;
    float r = v - (float)p0;
    p1 = (_Float16)r;
    r -= (float)p1;
    p2 = (_Float16)r;
}

#ifndef RBNN_X3FWD_BPREFETCH
#define RBNN_X3FWD_BPREFETCH 0                                           // 1: the B fragments of tap t + 1 read under tap t's MFMAs (round 4) — measured SLOWER: 1x28x28 8.42 -> 8.58 ms per forward call,
                                                                          // 3x32x32 15.1 -> 16.5 (spills beside 56 accumulators): with two waves per SIMD the other wave fills a tap's post-barrier round trip
#endif
#ifndef RBNN_CONVX3_OLD_IMG
#define RBNN_CONVX3_OLD_IMG 0                                            // 1: the round-2 image layout (pitch = P1W, chunk swizzle by (pos >> 2) & 1 only)
#endif
// Image layout of conv2_pool_x3_kernel: position (y, x) of the pooled conv1 image is a 64-B record (32 channels) at index y * IPITCH + x;
// 16-B channel octet o is stored at o ^ x3_img_swz(index, y).  A ds_read_b128 is served in four 16-lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, +32 (rbnn_common.hpp); lane (li, lg) gathers octet lg of the position of output li + a tap offset.  Enumerating
// every (tap, position tile, group) (tools/conv_x3_swizzle_search.py): with IPITCH = P1W and the swizzle ((idx >> 2) & 1) << 1 alone
// EVERY gather is a 2-way conflict at 1x28x28 (2.0 LDS passes per read; PMC round 2: 36 % of the LDS cycles) and 2.9 passes at
// 3x32x32 (43 %, the 12 idle lanes of the last tile all re-reading position 0 included).  1x28x28: XOR-ing the row parity into octet
// bit 0 makes all of them conflict-free (1.0).  3x32x32: rows of 10 outputs do not tile the 4-position period; a pitch of 18 (48 KB
// per point instead of 37) with the idle lanes spread over positions 0..11 brings it to 1.14.
template <class G> struct ConvX3Img {
    static constexpr bool MNIST = G::P1W == 12;
    static constexpr int IPITCH = RBNN_CONVX3_OLD_IMG ? G::P1W : (MNIST ? 12 : 18);
    static constexpr int ROWX = (!RBNN_CONVX3_OLD_IMG && MNIST) ? 1 : 0;
};
template <class G> __device__ __forceinline__ int x3_img_swz(int idx, int y) {
    return ((((idx >> 2) & 1) << 1) ^ (ConvX3Img<G>::ROWX * (y & 1)));
}

template <class G, int WROWS> struct ConvX3Lds {
    static constexpr int PLANEW = WROWS * 64, TILEW = 3 * PLANEW;        // one tap's weight tile: 3 planes of WROWS 64-B rows
    static constexpr int IPOS = G::P1W * ConvX3Img<G>::IPITCH, IMGP = IPOS * 64, IMGB = 3 * IMGP;   // one point's image: 3 planes of IPOS 64-B position records
    static constexpr int SCR = 8 * 16 * (G::NPOS + 4) * 4;               // the eight waves' pooling tiles (epilogue; alias the weight buffers)
    static constexpr int WBUF = (2 * TILEW > SCR ? 2 * TILEW : SCR);
    static constexpr int BYTES = WBUF + 2 * IMGB;
};

template <int ACT, class G, int WROWS>
__global__ void __launch_bounds__(512, 2) conv2_pool_x3_kernel(const ConvArgs a, const ConvX3Args x) {
    using L = ConvX3Lds<G, WROWS>;
    constexpr int HTW = WROWS / 64, NPT = G::NPT2, NW = 8;
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NPOS_ = G::NPOS, NP2_ = G::NP2;
    constexpr int WPP = WROWS / 16 / NW;                                  // DMA pieces (16 rows of one plane) per wave per plane
    static_assert(WROWS % 128 == 0 && L::BYTES <= 160 * 1024, "whole pieces per wave; LDS");
    static_assert((16 * NP2_) % 4 == 0, "four pooled cells per lane");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const ldsb = (char*)lds;
    char* const imgs = ldsb + L::WBUF;

    const int NB = (a.N + 1) / 2;                                         // blocks per sample
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, nb0 = (id % NB) * 2;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, wp = wave >> 2;                              // channel group, point of the pair
    const int sw = a.sidx ? a.sidx[s] : s;
    const int n = min(nb0 + wp, a.N - 1);                                 // a ragged last block computes its first point twice, stores once
    const bool live = nb0 + wp < a.N;
    const long long sn = (long long)s * a.N + n;
    const char* const Ws = x.K2t + (long long)sw * a.Hc * (K2 * 6);
    const int F = a.Hc * NP2_;
    const float p1_scale = x.p1_ds ? x.p1_ds->scale : ldexpf(1.f, x.p1_exp);
    const float out_scale = x.p1_ds ? ldexpf(1.f, -x.k2_exp) * x.p1_ds->inv_scale : ldexpf(1.f, -(x.k2_exp + x.p1_exp));

    // both points' fp32 images [32 ci][IPOS] -> three piece planes, channel-last, in LDS: one (position, channel octet) per thread-item
    constexpr int IPITCH = ConvX3Img<G>::IPITCH, NSRC = P1W_ * P1W_;
#ifdef RBNN_X3FWD_ABL_NOFILL
    for (int i = tid; i < 2 * NSRC * 4 && a.Hc == 12345; i += 512) {     // ablation (timing only): the images are never filled
#else
    for (int i = tid; i < 2 * NSRC * 4; i += 512) {
#endif
        const int pt2 = i / (NSRC * 4), rem = i % (NSRC * 4), pos = rem >> 2, o = rem & 3;
        const float* const src = a.P1 + ((long long)s * a.N + min(nb0 + pt2, a.N - 1)) * G::P1SZ + (8 * o) * NSRC + pos;
        union { f16x8 v; uint4 u; unsigned w[4]; } q0, q1, q2;
#pragma unroll
        for (int j = 0; j < 8; j += 2)
            split3_plain_pair(src[j * NSRC] * p1_scale, src[(j + 1) * NSRC] * p1_scale, 1.f, q0.w[j >> 1], q1.w[j >> 1], q2.w[j >> 1]);
        const int iy = pos / P1W_, idx = iy * IPITCH + pos % P1W_;
        char* const dst = imgs + pt2 * L::IMGB + idx * 64 + ((o ^ x3_img_swz<G>(idx, iy)) * 16);
        *(uint4*)dst = q0.u;
        *(uint4*)(dst + L::IMGP) = q1.u;
        *(uint4*)(dst + 2 * L::IMGP) = q2.u;
    }
    const int prow = lane >> 2;
    const unsigned src_off = (unsigned)(prow * 64 + ((lane & 3) ^ swz(prow)) * 16);   // inside a 1-KiB piece: row prow, logical chunk (lane & 3) ^ swz(row)
    const int foff = li * 64 + ((lg ^ swz(li)) * 16);
    // image position of output position 16pt + li (tap 0,0); positions past NPOS read (0,0), never stored
    int pbase[NPT], ybase[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        int pos = pt * 16 + li;
        if (pos >= NPOS_) pos = RBNN_CONVX3_OLD_IMG ? 0 : pos - NPOS_;  // idle lanes of the last tile: distinct valid positions (never stored)
        ybase[pt] = pos / O2W_;
        pbase[pt] = ybase[pt] * IPITCH + pos % O2W_;
    }
    const char* const img = imgs + wp * L::IMGB;
    // epilogue roles: lane handles the four consecutive pooled cells 4 * (lane + 64 it) .. of a 16-channel tile; their offsets in the wave's tile
    constexpr int CPITCH = NPOS_ + 4, EIT = (16 * NP2_ / 4 + 63) / 64;
    int pbase_e[EIT][4];
#pragma unroll
    for (int it = 0; it < EIT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = min(4 * (lane + 64 * it) + j, 16 * NP2_ - 1), hl = idx / NP2_, p = idx % NP2_;
            pbase_e[it][j] = hl * CPITCH + (p / P2W_) * O2W_ + (p % P2W_);
        }
    for (int hc0 = 0; hc0 < a.Hc; hc0 += WROWS) {
        f32x4 acc[HTW][NPT];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto stage = [&](int tap, int buf) {
            // the weight image is grouped [16 channels][tap][3 pieces][16 rows][64 B] (conv.py::_build_triple) and a 16-channel group sits in the
            // stage tile the same way — [group][3 pieces][1 KiB]: the immediate offset of global_load_lds applies to the global AND the LDS
            // address, so a group's three pieces share one address register and one M0 write
            char* const T = ldsb + buf * L::TILEW;
#pragma unroll
            for (int i = 0; i < WPP; ++i) {
                const unsigned grp = min((unsigned)(hc0 >> 4) + (unsigned)(wave + NW * i), (unsigned)(a.Hc >> 4) - 1u);   // groups past Hc repeat the last one; never stored
                const auto gsrc = (const __attribute__((address_space(1))) void*)(Ws + ((grp * 25u + (unsigned)tap) * 3072u + src_off));
                const auto ldst = (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)(T + (wave + NW * i) * 3072);
                __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 2048, 0);
            }
        };
        stage(0, 0);
        __syncthreads();                                                 // also orders the image fill (first chunk) / the previous chunk's pooling tiles
        // RBNN_X3FWD_BPREFETCH (off: measured slower, see its definition): B fragments (gathered from the point's image, which no barrier guards
        // after the first) read ONE TAP AHEAD into a second register set, under the current tap's MFMAs (the two sets alternate: no copies)
        // (3x32x32: seven position tiles x three planes x two sets do not fit beside the accumulators — only the plane of the FIRST product group,
        // b2, is read ahead there; b0 / b1 follow behind the barrier and land under that group's MFMAs)
        constexpr bool PF_ALL = RBNN_X3FWD_BPREFETCH && NPT <= 4;
        auto load_b = [&](int tap, f16x8 (&b0)[NPT], f16x8 (&b1)[NPT], f16x8 (&b2)[NPT], bool lo, bool hi) {
            const int ky = tap / 5, toff = ky * IPITCH + (tap % 5);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int p = pbase[pt] + toff;
                const char* const src = img + p * 64 + ((lg ^ x3_img_swz<G>(p, ybase[pt] + ky)) * 16);
                if (lo) { b0[pt] = *(const f16x8*)src; b1[pt] = *(const f16x8*)(src + L::IMGP); }
                if (hi) b2[pt] = *(const f16x8*)(src + 2 * L::IMGP);
            }
        };
        f16x8 bA0[NPT], bA1[NPT], bA2[NPT], bB0[PF_ALL ? NPT : 1], bB1[PF_ALL ? NPT : 1], bB2[NPT];
        auto tap_body = [&](int tap, f16x8 (&b0)[NPT], f16x8 (&b1)[NPT], f16x8 (&b2)[NPT], auto& n0, auto& n1, f16x8 (&n2)[NPT]) {
            const int buf = tap & 1;
            if (tap + 1 < 25) {
                stage(tap + 1, buf ^ 1);
#if RBNN_X3FWD_BPREFETCH
                if constexpr (PF_ALL) load_b(tap + 1, n0, n1, n2, true, true);
                else load_b(tap + 1, b0, b1, n2, false, true);            // (b0 / b1 unused by this call)
#endif
            }
#if RBNN_X3FWD_BPREFETCH
            if constexpr (!PF_ALL) load_b(tap, b0, b1, b2, true, false);
#else
            load_b(tap, b0, b1, b2, true, true);
#endif
            const char* const Wt = ldsb + buf * L::TILEW + (wq * HTW) * 3072 + foff;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                const f16x8 a0 = *(const f16x8*)(Wt + ht * 3072), a1 = *(const f16x8*)(Wt + ht * 3072 + 1024),
                            a2 = *(const f16x8*)(Wt + ht * 3072 + 2048);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b2[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a2, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a1, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a1, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b0[pt], acc[ht][pt]);
            }
            ring_wait_barrier<0>();                                      // tap+1's weights landed; everyone is done with this tile
        };
#if RBNN_X3FWD_BPREFETCH
        load_b(0, bA0, bA1, bA2, PF_ALL, true);
#endif
        for (int tap = 0; tap < 25; tap += 2) {
            if constexpr (PF_ALL) {
                tap_body(tap, bA0, bA1, bA2, bB0, bB1, bB2);
                if (tap + 1 < 25) tap_body(tap + 1, bB0, bB1, bB2, bA0, bA1, bA2);
            } else {                                                     // one b0 / b1 set (read behind the barrier each tap), two b2 sets
                tap_body(tap, bA0, bA1, bA2, bA0, bA1, bB2);
                if (tap + 1 < 25) tap_body(tap + 1, bA0, bA1, bB2, bA0, bA1, bA2);
            }
        }
        // epilogue (as conv2_pool_kernel): scale, bias, 2x2 / stride-1 max-pool of the pre-activations through a per-wave LDS tile
        // (aliases the weight buffers: every wave passed the barrier above), activation, stash.  The tile's channel pitch CPITCH = NPOS + 4
        // spreads the four channel groups of a store over the banks, and the cell -> tile offsets of a lane (pbase_e) are computed once per
        // kernel: the divisions by NP2 / P2W per cell made this epilogue 1.1 of the kernel's 7.9 ms (profiles/r03a/conv_dense_ablations.txt)
        float* const my = (float*)ldsb + wave * 16 * CPITCH;
#ifdef RBNN_X3FWD_ABL_NOEPI
        {                                                                  // ablation (timing only)
            float sink = 0.f;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) sink += acc[ht][pt][0] + acc[ht][pt][1] + acc[ht][pt][2] + acc[ht][pt][3];
            if (sink == 1.2345e-30f) a.Q2[sn * F] = sink;
            __syncthreads();
            continue;
        }
#endif
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht) {
            const int hcb = hc0 + (wq * HTW + ht) * 16;                    // wave-uniform
            if (hcb >= a.Hc) break;
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pt * 16 + li < NPOS_) {
                        const float pre = acc[ht][pt][r] * out_scale + bias[r];   // sigmoid / tanh are pooled on their VALUES, as torch does
                        my[(4 * lg + r) * CPITCH + pt * 16 + li] = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                    }
            // four consecutive pooled cells per lane: one 16-byte store of Q2 and one 4-byte store of the stash (the tile's 16 x NP2
            // cells are contiguous in both)
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int i4 = lane + 64 * it;
                if (i4 < 16 * NP2_ / 4 && live) {
                    f32x4 q;
                    unsigned stw = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int base = pbase_e[it][j];
                        float best = my[base];
                        int arg = 0;
                        if (my[base + 1] > best) { best = my[base + 1]; arg = 1; }
                        if (my[base + O2W_] > best) { best = my[base + O2W_]; arg = 2; }
                        if (my[base + O2W_ + 1] > best) { best = my[base + O2W_ + 1]; arg = 3; }
                        q[j] = smooth_act<ACT>() ? best : act_fwd<ACT>(best);
                        stw |= (unsigned)(arg | (best > 0.f ? 4 : 0)) << (8 * j);
                    }
                    const long long o = sn * F + (long long)hcb * NP2_ + 4 * i4;       // a multiple of 4
#if RBNN_X3FWD_NT
                    __builtin_nontemporal_store(q, (f32x4*)(a.Q2 + o));
                    __builtin_nontemporal_store(stw, (unsigned*)(a.st2 + o));
#else
                    *(f32x4*)(a.Q2 + o) = q;
                    *(unsigned*)(a.st2 + o) = stw;
#endif
                }
            }
        }
        __syncthreads();                                                 // the pooling tiles alias the weight buffers of the next chunk
    }
}

template <int ACT, class G>
int launch_conv_forward_x3(const ConvArgs& a, const ConvX3Args& x, hipStream_t st) {
    constexpr int WROWS = (G::CIN == 1 ? 256 : 128);
    hipLaunchKernelGGL((conv1_pool_kernel<ACT, G>), dim3((unsigned)((long long)a.S * a.N)), dim3(conv1_threads<G>()), 0, st, a);
    int rc = launch_status();
    if (rc) return rc;
    constexpr int LDSB = ConvX3Lds<G, WROWS>::BYTES;
    static unsigned long long attr = 0;                                   // per instantiation, one bit per device
    if (!ensure_dynamic_lds((const void*)conv2_pool_x3_kernel<ACT, G, WROWS>, LDSB, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL((conv2_pool_x3_kernel<ACT, G, WROWS>), dim3(grid_for_items((long long)((a.N + 1) / 2) * a.S)), dim3(512), LDSB, st, a, x);
    if ((rc = launch_status())) return rc;
    const int items = ((a.N + 15) / 16) * a.S;
    hipLaunchKernelGGL(conv_fc_kernel, dim3((items + 3) / 4), dim3(256), 0, st, a);
    return launch_status();
}

int validate_conv(const rbnn_conv_posterior* net) {
    if (!net || !net->K1w || !net->K1b || !net->K2w || !net->K2b || !net->Fw || !net->Fb) return RBNN_ERR_NULL;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    if (!((net->in_channels == 1 && net->in_width == 28) || (net->in_channels == 3 && net->in_width == 32))) return RBNN_ERR_UNSUPPORTED;
    if (net->hidden < 16 || (net->hidden & 15) || net->n_classes < 1 || net->n_classes > RBNN_CPAD || net->n_stored < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(net->K2w) || !aligned16(net->K2b) || !aligned16(net->Fw)) return RBNN_ERR_ALIGN;
    return RBNN_OK;
}
// the split-half conv kernels: 1x28x28, relu / leaky
int validate_conv_split(const rbnn_conv_posterior* net) {
    const int rc = validate_conv(net);
    if (rc) return rc;
    if (net->in_channels != 1 || net->in_width != 28) return RBNN_ERR_UNSUPPORTED;
    if (net->activation != RBNN_ACT_RELU && net->activation != RBNN_ACT_LEAKY) return RBNN_ERR_UNSUPPORTED;
    return RBNN_OK;
}

// geometry / activation dispatch of the exact kernels: f(Geo{}) / f(integral_constant<int, ACT>{})
template <class F> int for_geometry(const rbnn_conv_posterior* net, F&& f) {
    if (net->in_channels == 1 && net->in_width == 28) return f(GeoMnist{});
    if (net->in_channels == 3 && net->in_width == 32) return f(GeoCifar{});
    return RBNN_ERR_UNSUPPORTED;
}
template <class F> int for_activation(int act, F&& f) {
    switch (act) {
        case RBNN_ACT_RELU:  return f(std::integral_constant<int, RBNN_ACT_RELU>{});
        case RBNN_ACT_LEAKY: return f(std::integral_constant<int, RBNN_ACT_LEAKY>{});
        case RBNN_ACT_SIGM:  return f(std::integral_constant<int, RBNN_ACT_SIGM>{});
        case RBNN_ACT_TANH:  return f(std::integral_constant<int, RBNN_ACT_TANH>{});
    }
    return RBNN_ERR_UNSUPPORTED;
}

template <int ACT, class G>
int launch_conv_forward(const ConvArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((conv1_pool_kernel<ACT, G>), dim3((unsigned)((long long)a.S * a.N)), dim3(conv1_threads<G>()), 0, st, a);
    int rc = launch_status();
    if (rc) return rc;
    constexpr int LDSB = conv2_lds_floats<G>() * 4;
    static unsigned long long attr = 0;                                   // per instantiation, one bit per device
    if (!ensure_dynamic_lds((const void*)conv2_pool_kernel<ACT, G>, LDSB, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL((conv2_pool_kernel<ACT, G>), dim3(grid_for_items((long long)a.N * a.S)), dim3(256), LDSB, st, a);
    if ((rc = launch_status())) return rc;
    const int items = ((a.N + 15) / 16) * a.S;
    hipLaunchKernelGGL(conv_fc_kernel, dim3((items + 3) / 4), dim3(256), 0, st, a);
    return launch_status();
}

}  // namespace

extern "C" {

int rbnn_conv_workspace_query(const rbnn_conv_posterior* net, int32_t N, int32_t S, rbnn_conv_workspace_sizes* out) {
    if (!net || !out) return RBNN_ERR_NULL;
    if (net->hidden < 16 || (net->hidden & 15) || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        const size_t SN = (size_t)S * N, F = (size_t)net->hidden * G::NP2;
        rbnn_conv_workspace_sizes z = {};
        z.P = z.dZ = SN * RBNN_CPAD * sizeof(float);
        z.P1 = SN * (size_t)G::P1STRIDE;                          // the fp32 image [32][P1W][P1W] in whole 1-KiB pieces, or the 24 KiB split image; the backward reuses it for dP1
        z.st1 = SN * G::P1SZ;
        z.Q2 = SN * F * sizeof(float);
        z.st2 = SN * F;
        z.G = SN * G::DIN * sizeof(float);
        *out = z;
        return (int)RBNN_OK;
    });
}

int rbnn_conv_forward(const rbnn_conv_posterior* net, const float* X, int32_t ldx, int32_t N, const int32_t* sidx, int32_t S,
                      int32_t out_kind, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!X || !ws || !ws->P || !ws->P1 || !ws->st1 || !ws->Q2 || !ws->st2) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || ldx < net->in_channels * net->in_width * net->in_width) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(ws->P) || !aligned16(ws->P1) || !aligned16(ws->Q2)) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a = {};
    a.X = X; a.ldx = ldx; a.N = N;
    a.K1w = net->K1w; a.K1b = net->K1b; a.K2w = net->K2w; a.K2b = net->K2b; a.Fw = net->Fw; a.Fb = net->Fb;
    a.Hc = net->hidden; a.C = net->n_classes; a.sidx = sidx; a.S = S;
    a.P1 = ws->P1; a.st1 = ws->st1; a.Q2 = ws->Q2; a.st2 = ws->st2; a.P = ws->P; a.out_kind = out_kind;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        a.NP2 = G::NP2;
        return for_activation(net->activation, [&](auto act) { return launch_conv_forward<decltype(act)::value, G>(a, st); });
    });
}

int rbnn_conv_forward_split(const rbnn_conv_posterior* net, const void* K2_rows, int32_t k2_exp, int32_t p1_exp,
                            const rbnn_dev_scale* p1_dev_scale, const float* X,
                            int32_t ldx, int32_t N, const int32_t* sidx, int32_t S, int32_t out_kind,
                            const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv_split(net);
    if (rc) return rc;
    if (!K2_rows || !X || !ws || !ws->P || !ws->P1 || !ws->st1 || !ws->Q2 || !ws->st2) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || ldx < 784 || k2_exp < -100 || k2_exp > 100 || p1_exp < -100 || p1_exp > 100) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(K2_rows) || !aligned16(ws->P) || !aligned16(ws->P1) || !aligned16(ws->Q2)) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a = {};
    a.X = X; a.ldx = ldx; a.N = N;
    a.K1w = net->K1w; a.K1b = net->K1b; a.K2w = net->K2w; a.K2b = net->K2b; a.Fw = net->Fw; a.Fb = net->Fb;
    a.Hc = net->hidden; a.C = net->n_classes; a.sidx = sidx; a.S = S;
    a.P1 = ws->P1; a.st1 = ws->st1; a.Q2 = ws->Q2; a.st2 = ws->st2; a.P = ws->P; a.out_kind = out_kind; a.NP2 = NP2;
    ConvSplitArgs sp = {};
    sp.K2r = (const char*)K2_rows; sp.k2_exp = k2_exp; sp.p1_exp = p1_exp; sp.p1_ds = p1_dev_scale; sp.P1s = (char*)ws->P1;   // ws->P1 holds 24 KiB per (s, n)
    const long long t1 = (long long)S * N * (P1W * P1W);
    const bool leaky = net->activation == RBNN_ACT_LEAKY;
    if (leaky) hipLaunchKernelGGL(conv1_pool_split_kernel<RBNN_ACT_LEAKY>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, a, sp);
    else       hipLaunchKernelGGL(conv1_pool_split_kernel<RBNN_ACT_RELU>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, a, sp);
    if ((rc = launch_status())) return rc;
    constexpr int LDSB = 2 * 256 * 128 + 2 * P1SPLIT;
    static unsigned long long attr_leaky = 0, attr_relu = 0;              // one bit per device
    if (!ensure_dynamic_lds((const void*)conv2_pool_split_kernel<RBNN_ACT_LEAKY>, LDSB, attr_leaky) ||
        !ensure_dynamic_lds((const void*)conv2_pool_split_kernel<RBNN_ACT_RELU>, LDSB, attr_relu)) return RBNN_ERR_LAUNCH;
    const int grid = grid_for_items((long long)((N + 1) / 2) * S);
    if (leaky) hipLaunchKernelGGL(conv2_pool_split_kernel<RBNN_ACT_LEAKY>, dim3(grid), dim3(512), LDSB, st, a, sp);
    else       hipLaunchKernelGGL(conv2_pool_split_kernel<RBNN_ACT_RELU>, dim3(grid), dim3(512), LDSB, st, a, sp);
    if ((rc = launch_status())) return rc;
    const int items = ((N + 15) / 16) * S;
    hipLaunchKernelGGL(conv_fc_kernel, dim3((items + 3) / 4), dim3(256), 0, st, a);
    return launch_status();
}

int rbnn_conv_forward_triple(const rbnn_conv_posterior* net, const void* K2_triple, int32_t k2_exp, int32_t p1_exp,
                             const rbnn_dev_scale* p1_dev_scale, const float* X, int32_t ldx, int32_t N, const int32_t* sidx,
                             int32_t S, int32_t out_kind, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!K2_triple || !X || !ws || !ws->P || !ws->P1 || !ws->st1 || !ws->Q2 || !ws->st2) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || ldx < net->in_channels * net->in_width * net->in_width) return RBNN_ERR_SHAPE;
    if (k2_exp < -100 || k2_exp > 100 || p1_exp < -100 || p1_exp > 100) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(K2_triple) || !aligned16(ws->P) || !aligned16(ws->P1) || !aligned16(ws->Q2)) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a = {};
    a.X = X; a.ldx = ldx; a.N = N;
    a.K1w = net->K1w; a.K1b = net->K1b; a.K2w = net->K2w; a.K2b = net->K2b; a.Fw = net->Fw; a.Fb = net->Fb;
    a.Hc = net->hidden; a.C = net->n_classes; a.sidx = sidx; a.S = S;
    a.P1 = ws->P1; a.st1 = ws->st1; a.Q2 = ws->Q2; a.st2 = ws->st2; a.P = ws->P; a.out_kind = out_kind;
    ConvX3Args x = {};
    x.K2t = (const char*)K2_triple; x.k2_exp = k2_exp; x.p1_exp = p1_exp; x.p1_ds = p1_dev_scale;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        a.NP2 = G::NP2;
        return for_activation(net->activation, [&](auto act) { return launch_conv_forward_x3<decltype(act)::value, G>(a, x, st); });
    });
}

}  // extern "C"

// =====================================================================================================
// Backward to the input.  One WAVE = one (sample, point); a block is 4 points of one sample and never synchronises:
// the four waves only share the weight stream through L1.  Everything between dZ and dX stays in registers / LDS.
//   per chunk of 16 conv2 channels:
//     1. the wave zeroes its own zero-padded image dO2pad[16 hc][16][16] (8x8 gradients, border 4) in LDS and
//        fills it: dQ2[f] = sum_c dZ[c] * Fw[c][f] (Linear^T, coalesced over f), routed through the pool-2 argmax
//        and the activation derivative (LDS float atomics: the stride-1 windows overlap);
//     2. conv2^T as an implicit GEMM on the matrix pipe, gather form:
//          dP1^T[ci][(Y,X)] += sum_{k=(hc,ky,kx)} W2[hc][ci][ky][kx] * dO2pad[hc][Y - ky + 4][X - kx + 4]
//        A = model.3.weight with its channel axes swapped, [ci][hc*25 + tap] (K-contiguous, read straight from
//        memory, 16 B = 4 K steps per lane); B gathered from the image with a k -> offset table, exactly as the
//        forward kernel gathers from P1.  The whole output, 2 ci tiles x 9 position tiles, lives in 18 accumulator
//        tiles for the entire K = Hc*25 loop.  (56 % of these MFMAs multiply padding zeros; in exchange there are
//        ~1800 MFMAs between any two waits.)
//   3. pool-1 routing + activation derivative from the accumulators, conv1^T as a 25-tap scatter into dX[28][28].
// =====================================================================================================
namespace {

struct ConvBwdArgs {
    const float* dZ; const uint8_t* st1; const uint8_t* st2;
    const float* K1w; const float* K2cb; const float* Fw;
    int Hc; int C; int N; int S; const int* sidx;
    float* dQ2;                                                          // [S][N][Hc*NP2] dL/d(pooled conv2 output) = dZ . Fw: aliases the forward's Q2
    float* dP1;                                                          // [S][N][P1SZ] dL/d(pooled conv1 output): aliases the forward's P1
    float* G;                                                            // [S][N][DIN]
    int NP2;                                                             // pooled conv2 positions per channel (conv_fc_bwd is geometry-agnostic)
};

// Linear^T on the matrix pipe: dQ2[n][f] = sum_c dZ[n][c] * Fw[c][f].  One wave = 16 points x 64 features (4 MFMA tiles,
// K = 16 padded classes); Fw is read once per 16 points.  SMOOTH (sigmoid / tanh): the result overwrites the forward's Q2 in
// place, element by element, so act'(Q2) is taken from the value about to be overwritten and folded in here.
template <bool SMOOTH, int ACT>
__global__ void __launch_bounds__(256) conv_fc_bwd_kernel(const ConvBwdArgs a) {
    // The accumulator tile holds dQ2[point 4lg + r][feature li]: stored straight from the registers, a store instruction covers four point
    // rows x 64 BYTES each (first version: 6.4 GB of such half-line writes at 3.1 TB/s, 2.08 ms per C5 pass).  The wave's 16 x 64 tile goes
    // through a private LDS tile instead and leaves as 16-byte stores: one instruction = four rows x 256 contiguous bytes.
    __shared__ __attribute__((aligned(16))) float tile[4][16 * 68];
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4, wv = threadIdx.x >> 6;
    const int F = a.Hc * a.NP2, FT = (F + 63) / 64, NT = (a.N + 15) / 16;
    // (32-bit item arithmetic: the launch checks S * NT * FT < 2^31; 64-bit division by run-time values was ~400 of this kernel's 900 instructions)
    const unsigned item = blockIdx.x * 4u + (unsigned)wv;
    if (item >= (unsigned)a.S * (unsigned)NT * (unsigned)FT) return;     // (a whole wave: the tile is wave-private, no block barrier anywhere)
    const unsigned it2 = item / (unsigned)FT;
    const int ft = (int)(item - it2 * (unsigned)FT), s = (int)(it2 / (unsigned)NT), nt = (int)(it2 - (unsigned)s * (unsigned)NT);
    const int sw = a.sidx ? a.sidx[s] : s;
    const int n = min(nt * 16 + li, a.N - 1);
    const f32x4 av = *(const f32x4*)(a.dZ + ((long long)s * a.N + n) * RBNN_CPAD + 4 * lg);     // A[i = n][k = lg] for K step r: class 4lg + r
    float* const T = tile[wv];
    // B[k = class][j = feature]: one wave-uniform 64-bit base (the sample's Fw) + 32-bit offsets, classes past C read class C - 1 and are selected to 0
    // (spelled `c < C ? Fw[64-bit index] : 0`, each of the 16 loads was an exec-masked block behind a 64-bit multiply-add: 909 instructions per 4 KB of output)
    const float* const Fws = a.Fw + (long long)sw * a.C * F;
    unsigned coff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) coff[r] = (unsigned)min(4 * lg + r, a.C - 1) * (unsigned)F + (unsigned)(ft * 64 + li);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (ft * 64 + q * 16 >= F) break;                                                     // F = NP2*Hc is a multiple of 16
        f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bl = Fws[coff[r] + 16u * q];
            d = MFMA16(av[r], 4 * lg + r < a.C ? bl : 0.f, d);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(4 * lg + r) * 68 + q * 16 + li] = d[r];                 // d[r] = dQ2[n = 16nt + 4lg + r][f]
    }
    // (same wave: the LDS unit serves a wave's requests in order — the reads below see the stores above)
    const int c4 = 4 * (lane & 15), fcol = ft * 64 + c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 4 * k + (lane >> 4), nn = nt * 16 + row;
        if (nn < a.N && fcol < F) {
            f32x4 v = *(const f32x4*)(T + row * 68 + c4);
            float* const dst = a.dQ2 + ((long long)s * a.N + nn) * F + fcol;
            if (SMOOTH) {
                const f32x4 h = *(const f32x4*)dst;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= act_grad_from_value<ACT>(h[r]);
            }
#if RBNN_FCBWD_NT
            __builtin_nontemporal_store(v, (f32x4*)dst);
#else
            *(f32x4*)dst = v;
#endif
        }
    }
}

// zero-padded gradient image of one wave (layout: struct Geo): 4 zero rows, then 16 channels of CHS floats, then 8 floats of
// slack (the last row's right border).  1x28x28: 9.5 KiB per wave (38 KiB per block), 3x32x32: 12.5 KiB (50 KiB per block).
template <class G> constexpr int conv_bwd_img_floats() { return 4 * G::PITCH + 16 * G::CHS + 8; }
// + the staging area of one chunk's pooled gradients (16 x NP2 floats) and stash bytes (16 x NP2), brought in by LDS-DMA
template <class G> constexpr int conv_bwd_stage_floats() { return (16 * G::NP2 * 5 + 15) / 16 * 4; }
template <class G> constexpr int conv_bwd_lds_floats() { return 4 * (conv_bwd_img_floats<G>() + conv_bwd_stage_floats<G>()); }

template <int ACT, class G>
__global__ void __launch_bounds__(256, 2) conv_bwd_kernel(const ConvBwdArgs a) {
    constexpr int HCH = 16, PITCH = G::PITCH, CHS = G::CHS, IMG = conv_bwd_img_floats<G>(), KCH = HCH * 25, NPT = G::NPT1;
    static_assert(PITCH == G::P1W && CHS % 8 == 4, "contiguous position runs; channel stride 4 (mod 8)");
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NP2_ = G::NP2, NPOS_ = G::NPOS;
    // channels per chunk; image pitch (gradient map + border 4); floats per channel (data rows + one shared border); floats per
    // wave image; K per chunk; position tiles of dP1
    static_assert(IMG % 4 == 0, "the image is cleared with 16-byte stores");
    constexpr int NFL = HCH * NP2_, STG = conv_bwd_stage_floats<G>(), WLDS = IMG + STG;
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 * (IMG + STG) floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    float* const img = lds + wave * WLDS + 4 * PITCH;                    // row 0 of channel 0 (4 zero rows above it); channel c at + c*CHS
    float* const sdq = lds + wave * WLDS + IMG;                          // [16 hc][NP2] pooled gradients of the chunk
    unsigned char* const sst = (unsigned char*)(sdq + NFL);              // [16 hc][NP2] stash bytes

    const int NB = (a.N + 3) / 4;                                        // blocks per sample
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, n = (id % NB) * 4 + wave;
    const int sw = a.sidx ? a.sidx[s] : s;
    if (n >= a.N) return;                                                // whole wave idle (ragged last block); no block barrier anywhere
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2_, KW = a.Hc * 25;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;

    // K order inside a chunk is TAP-major: k = t*16 + hl, so one K tile = one tap (ky,kx) x 16 channels, and
    // B[k = lg][j = li] of step r is img[(4lg + r)*CHS + (Y - ky)*PITCH + (X - kx + 4)]  (Y - ky in -4 .. O2W+3); with
    // PITCH == P1W that is img[(4lg + r)*CHS + pos + 4 - (ky*PITCH + kx)].
    int poff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)                                     // positions past P1W^2 (ragged last tile): any valid offset, never stored
        poff[pt] = (4 * lg) * CHS + min(pt * 16 + li, P1W_ * P1W_ - 1) + 4;
    const float* const Wr0 = a.K2cb + ((long long)sw * C1 + li) * KW + 4 * lg;          // ci = li; [ci][chunk][tap][16 hl]
    const float* const Wr1 = Wr0 + (long long)16 * KW;                                   // ci = 16 + li

    // Blocked accumulation: `acc` runs over FLUSH chunks (400 products each) and is then folded into `tot`.  One fp32 chain over
    // all K = 25*Hc products (12 800 at Hc = 512) carries ~sqrt(K) roundings of the running sum; blocks of 800 carry
    // ~sqrt(800) + sqrt(K/800) — measured against fp64 the per-point median error of the whole path drops accordingly.
    // (The second register set fits two blocks per CU only while 2 * 2 * NPT tiles do: 1x28x28.  3x32x32 keeps one chain.)
#ifndef RBNN_CONV_BWD_UNBLOCKED
    constexpr bool BLOCKED = NPT <= 9;
#else
    constexpr bool BLOCKED = false;
#endif
    constexpr int FLUSH = 2, NTOT = BLOCKED ? NPT : 1;
    f32x4 acc[2][NPT], tot[2][NTOT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < NTOT; ++pt) tot[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int i = lane; i < IMG / 4; i += 64) *(f32x4*)(lds + wave * WLDS + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};   // the borders stay zero
    // this lane's positions (y, x) of the O2W x O2W gradient map: lane, lane + 64, ...
    constexpr int NGP = (NPOS_ + 63) / 64;
    // the <= 4 stride-1 pooling windows (py,px) in {y-1,y} x {x-1,x} that contain (y,x); window q = 2dy+dx has (y,x) as its
    // element q, so it routes here iff its stashed argmax == q
    int woff[NGP][4], goff[NGP];
    bool wok[NGP][4], gok[NGP];
#pragma unroll
    for (int g = 0; g < NGP; ++g) {
        const int gp = lane + 64 * g, gy = gp / O2W_, gx = gp % O2W_;
        gok[g] = gp < NPOS_;
        goff[g] = gok[g] ? gy * PITCH + gx + 4 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int py = gy - (q >> 1), px = gx - (q & 1);
            wok[g][q] = gok[g] && py >= 0 && py < P2W_ && px >= 0 && px < P2W_;
            woff[g][q] = wok[g][q] ? py * P2W_ + px : 0;
        }
    }
    // The chunk's 16 x NP2 pooled gradients and stash bytes are contiguous in memory: they come in by 4-byte-per-lane LDS-DMA
    // (coalesced, no registers), issued right after the previous chunk's routing has read the staging area, so they land under
    // that chunk's MFMAs; the routing then gathers from LDS instead of from memory (the scattered global loads of the routing —
    // 32 per lane per 4 channels — were ~30 % of this kernel: the matrix pipe sat at 69 % busy).
    auto dma4 = [&](const void* g, void* l) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 4, 0, 0);
    };
    auto stage_chunk = [&](int hc0) {                                    // lane p of instruction q lands at byte 256q + 4p of its region
        const long long fb = sn * F + (long long)hc0 * NP2_;             // a multiple of 4: the stash dwords are aligned
#pragma unroll
        for (int q = 0; q < (NFL + 63) / 64; ++q)
            if (q * 64 + lane < NFL) dma4(a.dQ2 + fb + q * 64 + lane, sdq + q * 64);
#pragma unroll
        for (int q = 0; q < (NFL + 255) / 256; ++q)
            if (q * 256 + 4 * lane < NFL) dma4(a.st2 + fb + q * 256 + 4 * lane, sst + q * 256);
    };
    stage_chunk(0);
    for (int hc0 = 0; hc0 < a.Hc; hc0 += HCH) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this chunk's staging has landed (wave-local, no barrier)
        asm volatile("" ::: "memory");
        // 1. interior of the padded image for channels hc0 .. hc0+15: pool-2 routing + activation derivative, gather form
#pragma unroll
        for (int g = 0; g < NGP; ++g) {
#pragma unroll
            for (int h4 = 0; h4 < HCH; h4 += 4) {
                int st[4][4];
                float dq[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int fb = (h4 + j) * NP2_;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st[j][q] = sst[fb + woff[g][q]]; dq[j][q] = sdq[fb + woff[g][q]]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (wok[g][q] && (st[j][q] & 3) == q)             // smooth activations: act' is already folded into dQ2
                            v += (smooth_act<ACT>() || (st[j][q] & 4)) ? dq[j][q] : dq[j][q] * slope;
                    if (gok[g]) img[(h4 + j) * CHS + goff[g]] = v;        // Hc is a multiple of 16: every channel of the chunk exists
                }
            }
        }
        asm volatile("" ::: "memory");
        if (hc0 + HCH < a.Hc) stage_chunk(hc0 + HCH);                    // the staging area is free again: the next chunk's rows fly under the MFMAs
        // 2. 25 K tiles = 25 taps x 16 channels.  A position tile spans rows Ya..Yb of the P1W x P1W output; tap row ky reaches it
        //    only if some Y - ky lies in 0..O2W-1: the others multiply pure padding and are skipped (27 % of the MFMAs at 1x28x28).
        const float* const w0 = Wr0 + (long long)(hc0 / HCH) * KCH;
        const float* const w1 = Wr1 + (long long)(hc0 / HCH) * KCH;
        f32x4 a0 = *(const f32x4*)w0, a1 = *(const f32x4*)w1;
#pragma unroll 1
        for (int t = 0; t < 25; ++t) {
            const f32x4 c0 = a0, c1 = a1;
            if (t + 1 < 25) { a0 = *(const f32x4*)(w0 + 16 * (t + 1)); a1 = *(const f32x4*)(w1 + 16 * (t + 1)); }
            const int ky = t / 5, toff = ky * PITCH + t % 5;             // image offset is poff - toff (scalar)
            const float* const src = img - toff;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                constexpr int last = P1W_ * P1W_ - 1;
                const int Ya = (16 * pt) / P1W_, Yb = min(16 * pt + 15, last) / P1W_;
                if (ky < Ya - (O2W_ - 1) || ky > Yb) continue;           // wave-uniform
                const f32x4 b = (f32x4){src[poff[pt]], src[poff[pt] + CHS], src[poff[pt] + 2 * CHS], src[poff[pt] + 3 * CHS]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0][pt] = MFMA16(c0[r], b[r], acc[0][pt]);
                    acc[1][pt] = MFMA16(c1[r], b[r], acc[1][pt]);
                }
            }
        }
        if (BLOCKED && ((hc0 / HCH) % FLUSH == FLUSH - 1 || hc0 + HCH >= a.Hc)) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int pt = 0; pt < (BLOCKED ? NPT : 0); ++pt) { tot[ct][pt] += acc[ct][pt]; acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
    // 3. tot[ct][pt][r] = dL/dP1[ci = 16ct + 4lg + r][pos = 16pt + li] -> memory, over the forward's P1 (dead after this read:
    //    for sigmoid / tanh the activation value at the same index gives act', folded in here); conv1_bwd finishes the path.
    //    (An in-kernel scatter of the 25 conv1 taps needs LDS float atomics, which cost ~250 cycles per wave instruction:
    //    measured, they made the LDS the bottleneck of this kernel.)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * pt + li >= P1W_ * P1W_) continue;
                float* const dst = a.dP1 + sn * G::P1SZ + (16 * ct + 4 * lg + r) * (P1W_ * P1W_) + 16 * pt + li;
                const float v = BLOCKED ? tot[ct][BLOCKED ? pt : 0][r] : acc[ct][pt][r];
                *dst = smooth_act<ACT>() ? v * act_grad_from_value<ACT>(*dst) : v;
            }
}

// conv2^T in split-half precision: conv_bwd_kernel's structure (one wave = one (sample, point), no block barriers, the whole
// dP1^T[32 ci][144 pos] in 18 accumulator tiles), with the chunk's zero-padded gradient image stored CHANNEL-LAST as fp16
// hi / lo — img[plane][16 x 16 positions][16 hc] — so that the gathered B operand of v_mfma_f32_16x16x32_f16 is one
// ds_read_b128 per plane: a K step is TWO taps x 16 channels (lanes lg = 0,1: tap 2t, channel octets 0,1; lg = 2,3: tap 2t+1),
// 13 steps per chunk (the 26th tap is zero weight).  A = model.3.weight regrouped [ci][chunk][step][lg][8] as a split-rows
// image (rbnn_split_rows), read straight from memory.  The gradients are scaled per (sample, point) by a power of two taken
// from max|dZ| * max_f sum_c |Fw[c][f]| (x4 for the overlapping pool windows), divided out in the epilogue.
template <int ACT>
__global__ void __launch_bounds__(256, 2) conv_bwd_split_kernel(const ConvBwdArgs a, const char* __restrict__ K2b, int k2_exp, float fw_l1) {
    constexpr int HCH = 16, NPT = 9, PLANE = 256 * 32, NSTEP = 13;        // bytes of one image plane: 256 positions x 16 halves
    constexpr int NFL = HCH * NP2;                                        // 784 pooled cells per chunk
    constexpr int STG = NFL * 4 + NFL;                                    // staging: 784 dQ2 floats + 784 stash bytes, packed (2 blocks per CU need <= 80 KiB each)
    static_assert(STG % 16 == 0 && 4 * (2 * PLANE + STG) <= 81920, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[4 * (2 * PLANE + STG)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    char* const img = lds + wave * (2 * PLANE + STG);
    float* const sdq = (float*)(img + 2 * PLANE);                          // [16 hc][49] pooled gradients of the chunk
    unsigned char* const sst = (unsigned char*)(img + 2 * PLANE + NFL * 4);   // [16 hc][49] stash bytes

    const int NB = (a.N + 3) / 4;
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, n = (id % NB) * 4 + wave;
    const int sw = a.sidx ? a.sidx[s] : s;
    if (n >= a.N) return;
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2, NCH = a.Hc / HCH;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;

    // per-(sample, point) scale: |dO2| <= 4 * max_c |dZ_c| * fw_l1
    float dzmax = fabsf(a.dZ[sn * RBNN_CPAD + li]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, o));
    const float bound = 4.f * dzmax * fw_l1;
    int e = 0;
    if (bound > 0.f && bound < INFINITY) e = max(-100, min(100, 11 - ilogbf(bound)));
    const float in_scale = ldexpf(1.f, e), out_scale = ldexpf(1.f, -(e + k2_exp));

    int poff[NPT];                                                        // image position index of output position 16pt + li (tap 0,0)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) { const int pos = pt * 16 + li; poff[pt] = (pos / P1W + 4) * 16 + pos % P1W + 4; }
    // weight rows: row = sw*32 + ci, one 32-byte group (hi8 | lo8) per (chunk, step, lg)
    const long long rowb = (long long)NCH * NSTEP * 4 * 32;
    const char* const Wr0 = K2b + ((long long)sw * C1 + li) * rowb + lg * 32;
    const char* const Wr1 = Wr0 + 16 * rowb;

    f32x4 acc[2][NPT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int i = lane; i < 2 * PLANE / 16; i += 64) *(uint4*)(img + 16 * i) = make_uint4(0, 0, 0, 0);   // the border stays zero
    const int gy = lane >> 3, gx = lane & 7;
    int woff[4];
    bool wok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int py = gy - (q >> 1), px = gx - (q & 1);
        wok[q] = py >= 0 && py < P2W && px >= 0 && px < P2W;
        woff[q] = wok[q] ? py * P2W + px : 0;
    }
    char* const mine = img + ((gy + 4) * 16 + gx + 4) * 32;               // this lane's interior position record
    // The chunk's 16 x 49 pooled gradients and stash bytes are contiguous in memory: they come in by 4-byte-per-lane LDS-DMA
    // (13 + 4 wave instructions, coalesced, no registers), issued right after the previous chunk's routing has read the
    // staging area, so they land under that chunk's MFMAs; the routing then gathers from LDS instead of from memory
    // (128 scattered global loads per lane per chunk made the routing 3/4 of this kernel's time).
    auto dma4 = [&](const void* g, void* l) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 4, 0, 0);
    };
    auto stage_chunk = [&](int ch) {                                       // lane p of instruction q lands at byte 256q + 4p of its region
        const long long fb = sn * F + (long long)ch * NFL;                 // a multiple of 4: the stash dwords are aligned
#pragma unroll
        for (int q = 0; q < 12; ++q) dma4(a.dQ2 + fb + q * 64 + lane, sdq + q * 64);
        if (lane < NFL - 12 * 64) dma4(a.dQ2 + fb + 12 * 64 + lane, sdq + 12 * 64);          // the last 16 floats
#pragma unroll
        for (int q = 0; q < 3; ++q) dma4(a.st2 + fb + q * 256 + 4 * lane, sst + q * 256);
        if (lane < (NFL - 3 * 256) / 4) dma4(a.st2 + fb + 3 * 256 + 4 * lane, sst + 3 * 256);    // the last 16 bytes
    };
    stage_chunk(0);
    for (int ch = 0; ch < NCH; ++ch) {
        const int hc0 = ch * HCH;
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): this chunk's staging has landed (wave-local, no barrier)
        asm volatile("" ::: "memory");
        // 1. interior of the image for channels hc0 .. hc0+15: pool-2 routing + activation derivative (gather form), scaled, split
        f16x8 hv[2], lv[2];
#pragma unroll
        for (int h4 = 0; h4 < HCH; h4 += 4) {
            int st[4][4];
            float dq[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fb = (h4 + j) * NP2;
#pragma unroll
                for (int q = 0; q < 4; ++q) { st[j][q] = sst[fb + woff[q]]; dq[j][q] = sdq[fb + woff[q]]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (wok[q] && (st[j][q] & 3) == q) v += (st[j][q] & 4) ? dq[j][q] : dq[j][q] * slope;
                v *= in_scale;
                const _Float16 h = (_Float16)v;
                hv[h4 >> 3][(h4 & 4) + j] = h;
                lv[h4 >> 3][(h4 & 4) + j] = (_Float16)(v - (float)h);
            }
        }
        *(f16x8*)(mine) = hv[0];
        *(f16x8*)(mine + 16) = hv[1];
        *(f16x8*)(mine + PLANE) = lv[0];
        *(f16x8*)(mine + PLANE + 16) = lv[1];
        asm volatile("" ::: "memory");
        if (ch + 1 < NCH) stage_chunk(ch + 1);                             // the staging area is free again: next chunk's rows fly under the MFMAs
        // 2. 13 K steps = tap pairs x 16 channels; a position tile whose rows no tap row of the pair can reach multiplies pure
        //    padding and is skipped
        const char* const w0 = Wr0 + (long long)ch * NSTEP * 128;
        const char* const w1 = Wr1 + (long long)ch * NSTEP * 128;
        f16x8 a0h = *(const f16x8*)w0, a0l = *(const f16x8*)(w0 + 16), a1h = *(const f16x8*)w1, a1l = *(const f16x8*)(w1 + 16);
#pragma unroll 1
        for (int t = 0; t < NSTEP; ++t) {
            const f16x8 c0h = a0h, c0l = a0l, c1h = a1h, c1l = a1l;
            if (t + 1 < NSTEP) {
                a0h = *(const f16x8*)(w0 + 128 * (t + 1)); a0l = *(const f16x8*)(w0 + 128 * (t + 1) + 16);
                a1h = *(const f16x8*)(w1 + 128 * (t + 1)); a1l = *(const f16x8*)(w1 + 128 * (t + 1) + 16);
            }
            const int tA = 2 * t, tB = min(2 * t + 1, 24);               // the padded 26th tap has zero weights: any image offset will do
            const int kyA = tA / 5, kyB = tB / 5;
            const int tap = (lg >> 1) ? tB : tA;                          // this lane's tap
            const char* const src = img + (lg & 1) * 16 - ((tap / 5) * 16 + tap % 5) * 32;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int Ya = (16 * pt) / P1W, Yb = (16 * pt + 15) / P1W;
                if ((kyA < Ya - 7 || kyA > Yb) && (kyB < Ya - 7 || kyB > Yb)) continue;   // wave-uniform
                const f16x8 bh = *(const f16x8*)(src + poff[pt] * 32), bl = *(const f16x8*)(src + PLANE + poff[pt] * 32);
                acc[0][pt] = MFMA_H(c0l, bh, acc[0][pt]);
                acc[1][pt] = MFMA_H(c1l, bh, acc[1][pt]);
                acc[0][pt] = MFMA_H(c0h, bl, acc[0][pt]);
                acc[1][pt] = MFMA_H(c1h, bl, acc[1][pt]);
                acc[0][pt] = MFMA_H(c0h, bh, acc[0][pt]);
                acc[1][pt] = MFMA_H(c1h, bh, acc[1][pt]);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) a.dP1[sn * P1SZ + (16 * ct + 4 * lg + r) * (P1W * P1W) + 16 * pt + li] = acc[ct][pt][r] * out_scale;
}

// conv2^T in the triple-split mode, BOTH geometries: conv_bwd_split_kernel's structure (one wave = one (sample, point), no block
// barriers, the whole dP1^T[32 ci][P1W^2 pos] in 2 x NPT1 accumulator tiles) with three piece planes and six product terms.  The
// chunk's zero-padded gradient image is stored channel-last in conv_bwd_kernel's shared-border geometry (pitch = P1W: a row's
// right border is the next row's left border): img[piece][(O2W + 8) * PITCH + 8 positions][16 hc] halves.  A K step is TWO taps x
// 16 channels (lanes lg = 0,1: tap 2t, channel octets 0,1; lg = 2,3: tap 2t + 1), 13 steps per chunk (the 26th tap has zero
// weights).  A = model.3.weight regrouped [ci][chunk][step][lg][8] as a triple-rows image (one 192-B stage per step), read straight
// from memory and shared by the block's waves through L1.  The gradients are scaled per (sample, point) by a power of two from
// max|dZ| * max_f sum_c |Fw[c][f]| (x4 for the overlapping pool windows), divided out in the epilogue.  NWB waves per block, one block
// per CU (23 / 31 KB of LDS per wave).
#ifndef RBNN_CONVBWD_X3_UNROLL
#define RBNN_CONVBWD_X3_UNROLL 0      // 1: the 13-step loop fully unrolled (one basic block per chunk) — measured SLOWER: 18.1 -> 19.6 ms (1x28x28), 31.8 -> 38.2 (3x32x32)
#endif
template <class G, bool SMOOTH = false> struct ConvBwdX3Lds {
    static constexpr int HCH = 16;
    static constexpr int IPB = (G::O2W + 8) * G::PITCH + 8;              // positions of a wave's padded image
    static constexpr int PLANE = IPB * HCH * 2;                           // bytes of one piece plane
    static constexpr int NFL = HCH * G::NP2;                              // pooled cells per chunk
    static constexpr int STG = (NFL * 5 + 15) / 16 * 16;                  // staging: NFL dQ2 floats + NFL stash bytes
    static constexpr int WAVE = 3 * PLANE + STG;
    // waves per block (one block per CU).  The kernel keeps TWO accumulator sets (blocked accumulation, below): 2 x 2 x NPT1 tiles =
    // 144 registers at 1x28x28 — six waves, two of the SIMDs hold two (256 registers each) — and 208 at 3x32x32: four waves, one per
    // SIMD (512 registers); LDS would allow 6 / 5
#ifndef RBNN_CONVBWD_X3_NWB_SMALL
#define RBNN_CONVBWD_X3_NWB_SMALL 6
#endif
    static constexpr int NWB = (G::NPT1 <= 9 && !SMOOTH) ? RBNN_CONVBWD_X3_NWB_SMALL : 4;   // (sigmoid / tanh: their read-modify-write epilogue spills at 256 registers)
    static_assert(NWB * WAVE <= 160 * 1024, "LDS");
};

template <int ACT, class G>
__global__ void __launch_bounds__((64 * ConvBwdX3Lds<G, smooth_act<ACT>()>::NWB), ((ConvBwdX3Lds<G, smooth_act<ACT>()>::NWB + 3) / 4))
conv_bwd_x3_kernel(const ConvBwdArgs a, const char* __restrict__ K2b, int k2_exp, float fw_l1) {
    using L = ConvBwdX3Lds<G, smooth_act<ACT>()>;
    constexpr int HCH = L::HCH, NPT = G::NPT1, PLANE = L::PLANE, NSTEP = 13, NFL = L::NFL, NWB = L::NWB;
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NP2_ = G::NP2, NPOS_ = G::NPOS, PITCH = G::PITCH;
    static_assert(PITCH == P1W_ && PLANE % 16 == 0 && L::WAVE % 16 == 0, "shared-border image; 16-byte clears");
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* const lds = (char*)lds_f;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const img = lds + wave * L::WAVE;
    float* const sdq = (float*)(img + 3 * PLANE);                          // [16 hc][NP2] pooled gradients of the chunk
    unsigned char* const sst = (unsigned char*)(img + 3 * PLANE + NFL * 4);   // [16 hc][NP2] stash bytes

    const int NB = (a.N + NWB - 1) / NWB;
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, n = (id % NB) * NWB + wave;
    const int sw = a.sidx ? a.sidx[s] : s;
    if (n >= a.N) return;                                                // whole wave idle (ragged last block); no block barrier anywhere
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2_, NCH = a.Hc / HCH;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;

    // per-(sample, point) scale: |dO2| <= 4 * max_c |dZ_c| * fw_l1
    float dzmax = fabsf(a.dZ[sn * RBNN_CPAD + li]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, o));
    const float bound = 4.f * dzmax * fw_l1;
    int e = 0;
    if (bound > 0.f && bound < INFINITY) e = max(-100, min(100, 13 - ilogbf(bound)));
    const float in_scale = ldexpf(1.f, e), out_scale = ldexpf(1.f, -(e + k2_exp));

    // image position of output position 16pt + li for tap (0,0): row 4 is the first gradient row, column 4 the first gradient column
    int poff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) poff[pt] = 4 * PITCH + min(pt * 16 + li, P1W_ * P1W_ - 1) + 4;
    // weight rows: row = sw*32 + ci; one 192-byte stage per (chunk, step): piece p at 64p, lane group lg at 16lg
    const long long rowb = (long long)NCH * NSTEP * 192;
    const char* const Wr0 = K2b + ((long long)sw * C1 + li) * rowb + lg * 16;
    const char* const Wr1 = Wr0 + 16 * rowb;

    // Blocked accumulation, as conv_bwd_kernel: `acc` runs over FLUSH chunks (2 x 13 steps x 6 terms = 156 accumulations) and is then
    // folded into `tot` — one fp32 chain over all 78 * Hc / 16 accumulations carried 4-7x the rounding error of the fp32-MFMA kernel
    // (measured against fp64 at Hc = 1024: median 3.0e-6 unblocked)
    constexpr int FLUSH = 2;
    f32x4 acc[2][NPT], tot[2][NPT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) { acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int i = lane; i < 3 * PLANE / 16; i += 64) *(uint4*)(img + 16 * i) = make_uint4(0, 0, 0, 0);   // the borders stay zero
    // this lane's positions (gy, gx) of the O2W x O2W gradient map: lane, lane + 64, ...; window q = 2dy + dx of the <= 4 stride-1
    // pooling windows containing it routes here iff its stashed argmax == q
    constexpr int NGP = (NPOS_ + 63) / 64;
    // (wcode: the stash code that routes window q here — argmax == q and, ReLU, bit 2 set; 15 for a window off the map: never matches)
    int woff[NGP][4], goff[NGP], wcode[NGP][4];
    bool gok[NGP];
#pragma unroll
    for (int g = 0; g < NGP; ++g) {
        const int gp = lane + 64 * g, gy = gp / O2W_, gx = gp % O2W_;
        gok[g] = gp < NPOS_;
        goff[g] = gok[g] ? ((gy + 4) * PITCH + gx + 4) * (HCH * 2) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int py = gy - (q >> 1), px = gx - (q & 1);
            const bool ok = gok[g] && py >= 0 && py < P2W_ && px >= 0 && px < P2W_;
            woff[g][q] = ok ? py * P2W_ + px : 0;
            wcode[g][q] = ok ? (ACT == RBNN_ACT_RELU ? (q | 4) : q) : 15;
        }
    }
    auto dma4 = [&](const void* g, void* l) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 4, 0, 0);
    };
    // lane p of instruction q lands at byte 256q + 4p of its region — and reads byte 256q + 4p of the chunk's rows: the immediate offset of
    // global_load_lds applies to both addresses, so up to 16 instructions (offsets 0 .. 3840) share one address register and one M0 write
    // (one wave per SIMD here: every set-up is time the matrix pipe idles)
    auto stage_chunk = [&](int ch) {
        const long long fb = sn * F + (long long)ch * NFL;                 // a multiple of 4: the stash dwords are aligned
        static_for<0, (NFL + 63) / 64>([&](auto Q) {
            constexpr int q = decltype(Q)::value, q0 = q & ~15;
            if (q * 64 + lane < NFL)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.dQ2 + fb + q0 * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)(sdq + q0 * 64), 4, (q - q0) * 256, 0);
        });
        static_for<0, (NFL + 255) / 256>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if (q * 256 + 4 * lane < NFL)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.st2 + fb + 4 * lane),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)sst, 4, q * 256, 0);
        });
    };
    stage_chunk(0);
    for (int ch = 0; ch < NCH; ++ch) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): this chunk's staging has landed (wave-local, no barrier)
        asm volatile("" ::: "memory");
        // 1. interior of the image for channels 16ch .. 16ch+15: pool-2 routing + activation derivative (gather form), scaled, split
#pragma unroll
        for (int g = 0; g < NGP; ++g) {
            union { f16x8 v; uint4 u; unsigned w[4]; } q0[2], q1[2], q2[2];
#pragma unroll
            for (int h4 = 0; h4 < HCH; h4 += 4) {
                int st[4][4];
                float dq[4][4], vr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int fb = (h4 + j) * NP2_;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st[j][q] = sst[fb + woff[g][q]]; dq[j][q] = sdq[fb + woff[g][q]]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                         // (as conv_bwd_dense_x3_kernel's route_one)
                        if (L::NWB > 4) {                                  // two waves on a SIMD (256 registers): the stash codes would spill
                            if (wcode[g][q] != 15 && (st[j][q] & 3) == q) v += (st[j][q] & 4) ? dq[j][q] : dq[j][q] * slope;
                        } else if (ACT == RBNN_ACT_RELU) {
                            v += (st[j][q] & 7) == wcode[g][q] ? dq[j][q] : 0.f;
                        } else if (ACT == RBNN_ACT_LEAKY) {
                            const int m = st[j][q] & 7;
                            v = fmaf(dq[j][q], m == (wcode[g][q] | 4) ? 1.f : (m == wcode[g][q] ? slope : 0.f), v);
                        } else {
                            v += (st[j][q] & 3) == wcode[g][q] ? dq[j][q] : 0.f;   // smooth activations: act' is already folded into dQ2
                        }
                    }
                    vr[j] = v * in_scale;
                }
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int w = ((h4 & 4) + j) >> 1;
                    split3_plain_pair(vr[j], vr[j + 1], 1.f, q0[h4 >> 3].w[w], q1[h4 >> 3].w[w], q2[h4 >> 3].w[w]);
                }
            }
            if (gok[g]) {
                char* const mine = img + goff[g];
                *(uint4*)(mine) = q0[0].u;             *(uint4*)(mine + 16) = q0[1].u;
                *(uint4*)(mine + PLANE) = q1[0].u;     *(uint4*)(mine + PLANE + 16) = q1[1].u;
                *(uint4*)(mine + 2 * PLANE) = q2[0].u; *(uint4*)(mine + 2 * PLANE + 16) = q2[1].u;
            }
        }
        asm volatile("" ::: "memory");
        if (ch + 1 < NCH) stage_chunk(ch + 1);                             // the staging area is free again: next chunk's rows fly under the MFMAs
        // 2. 13 K steps = tap pairs x 16 channels; a position tile whose rows no tap row of the pair can reach multiplies pure
        //    padding and is skipped
        const char* const w0 = Wr0 + (long long)ch * NSTEP * 192;
        const char* const w1 = Wr1 + (long long)ch * NSTEP * 192;
        f16x8 a00 = *(const f16x8*)w0, a01 = *(const f16x8*)(w0 + 64), a02 = *(const f16x8*)(w0 + 128);
        f16x8 a10 = *(const f16x8*)w1, a11 = *(const f16x8*)(w1 + 64), a12 = *(const f16x8*)(w1 + 128);
#if RBNN_CONVBWD_X3_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int t = 0; t < NSTEP; ++t) {
            const f16x8 c00 = a00, c01 = a01, c02 = a02, c10 = a10, c11 = a11, c12 = a12;
            if (t + 1 < NSTEP) {
                a00 = *(const f16x8*)(w0 + 192 * (t + 1)); a01 = *(const f16x8*)(w0 + 192 * (t + 1) + 64); a02 = *(const f16x8*)(w0 + 192 * (t + 1) + 128);
                a10 = *(const f16x8*)(w1 + 192 * (t + 1)); a11 = *(const f16x8*)(w1 + 192 * (t + 1) + 64); a12 = *(const f16x8*)(w1 + 192 * (t + 1) + 128);
            }
            const int tA = 2 * t, tB = min(2 * t + 1, 24);               // the padded 26th tap has zero weights: any image offset will do
            const int kyA = tA / 5, kyB = tB / 5;
            const int tap = (lg >> 1) ? tB : tA;                          // this lane's tap
            const char* const src = img + (lg & 1) * 16 - ((tap / 5) * PITCH + tap % 5) * (HCH * 2);
            // Every tile is its own basic block (the skip test is a wave-uniform branch), so the scheduler cannot move a tile's three B
            // fragment reads above the previous tile's MFMAs: with ONE wave per SIMD (3x32x32) the LDS latency would be exposed once per
            // 12 MFMAs.  PREF: read the NEXT tile's fragments unconditionally before this tile's branch (3 wasted reads per skipped tile).
            constexpr bool PREF = L::NWB <= 4;
            f16x8 n0, n1, n2;
            if (PREF) { const char* const bp = src + poff[0] * (HCH * 2); n0 = *(const f16x8*)bp; n1 = *(const f16x8*)(bp + PLANE); n2 = *(const f16x8*)(bp + 2 * PLANE); }
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                constexpr int last = P1W_ * P1W_ - 1;
                const int Ya = (16 * pt) / P1W_, Yb = min(16 * pt + 15, last) / P1W_;
                f16x8 b0, b1, b2;
                if (PREF) {
                    b0 = n0; b1 = n1; b2 = n2;
                    if (pt + 1 < NPT) { const char* const bp = src + poff[pt + 1 < NPT ? pt + 1 : pt] * (HCH * 2); n0 = *(const f16x8*)bp; n1 = *(const f16x8*)(bp + PLANE); n2 = *(const f16x8*)(bp + 2 * PLANE); }
                }
                if ((kyA < Ya - (O2W_ - 1) || kyA > Yb) && (kyB < Ya - (O2W_ - 1) || kyB > Yb)) continue;   // wave-uniform
                if (!PREF) { const char* const bp = src + poff[pt] * (HCH * 2); b0 = *(const f16x8*)bp; b1 = *(const f16x8*)(bp + PLANE); b2 = *(const f16x8*)(bp + 2 * PLANE); }
                acc[0][pt] = MFMA_H(c00, b2, acc[0][pt]);
                acc[1][pt] = MFMA_H(c10, b2, acc[1][pt]);
                acc[0][pt] = MFMA_H(c02, b0, acc[0][pt]);
                acc[1][pt] = MFMA_H(c12, b0, acc[1][pt]);
                acc[0][pt] = MFMA_H(c01, b1, acc[0][pt]);
                acc[1][pt] = MFMA_H(c11, b1, acc[1][pt]);
                acc[0][pt] = MFMA_H(c01, b0, acc[0][pt]);
                acc[1][pt] = MFMA_H(c11, b0, acc[1][pt]);
                acc[0][pt] = MFMA_H(c00, b1, acc[0][pt]);
                acc[1][pt] = MFMA_H(c10, b1, acc[1][pt]);
                acc[0][pt] = MFMA_H(c00, b0, acc[0][pt]);
                acc[1][pt] = MFMA_H(c10, b0, acc[1][pt]);
            }
        }
        if (ch % FLUSH == FLUSH - 1 || ch + 1 == NCH) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) { tot[ct][pt] += acc[ct][pt]; acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * pt + li >= P1W_ * P1W_) continue;
                float* const dst = a.dP1 + sn * G::P1SZ + (16 * ct + 4 * lg + r) * (P1W_ * P1W_) + 16 * pt + li;
                const float v = tot[ct][pt][r] * out_scale;              // over the forward's P1 (dead after this read): sigmoid / tanh take act' from it
                *dst = smooth_act<ACT>() ? v * act_grad_from_value<ACT>(*dst) : v;
            }
}

// pool-1 routing + activation derivative + conv1^T (in_channels = 1), gather form: one thread per input pixel (Yo, Xo).
// A pooled cell (c, py, px) routes its gradient to ONE conv1 output (Ya, Xa) = (2py + arg/2, 2px + arg%2); that output
// touches pixel (Yo, Xo) through tap (Yo - Ya, Xo - Xa) if it lies in the 5x5 kernel.  Only cells with
// py in [(Yo-4)/2, Yo/2], px likewise can qualify: at most 3 x 3 per channel.
template <int ACT, class G>
__global__ void __launch_bounds__(256) conv1_bwd_kernel(const ConvBwdArgs a) {
    // one block = one (sample, point): its P1SZ pooled gradients (activation derivative and argmax position folded in)
    // and the sample's 32 x Cin x 25 conv1 weights go to LDS once; each thread then gathers some of the DIN input pixels
    constexpr int P1W_ = G::P1W, IW = G::IW;
    __shared__ float gs[G::P1SZ];
    __shared__ unsigned char as[G::P1SZ];
    __shared__ float ws[C1 * G::K1];
    const long long sn = blockIdx.x;
    const int s = (int)(sn / a.N), sw = a.sidx ? a.sidx[s] : s, tid = threadIdx.x;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;
    for (int e = tid; e < G::P1SZ; e += 256) {
        const int st = a.st1[sn * G::P1SZ + e];
        const float d = a.dP1[sn * G::P1SZ + e];
        gs[e] = (smooth_act<ACT>() || (st & 4)) ? d : d * slope;          // smooth activations: act' was folded in by conv_bwd
        as[e] = (unsigned char)(st & 3);
    }
    for (int e = tid; e < C1 * G::K1; e += 256) ws[e] = a.K1w[(long long)sw * C1 * G::K1 + e];
    __syncthreads();
    for (int pix = tid; pix < G::DIN; pix += 256) {
        const int ci = pix / (IW * IW), Yo = (pix / IW) % IW, Xo = pix % IW;
        const int py0 = max(0, (Yo - 4) >> 1), py1 = min(P1W_ - 1, Yo >> 1), px0 = max(0, (Xo - 4) >> 1), px1 = min(P1W_ - 1, Xo >> 1);
        float g = 0.f;
        for (int c = 0; c < C1; ++c)
            for (int py = py0; py <= py1; ++py)
                for (int px = px0; px <= px1; ++px) {
                    const int e = c * (P1W_ * P1W_) + py * P1W_ + px, arg = as[e];
                    const int ky = Yo - (2 * py + (arg >> 1)), kx = Xo - (2 * px + (arg & 1));
                    if (ky >= 0 && ky < 5 && kx >= 0 && kx < 5) g = fmaf(gs[e], ws[(c * G::CIN + ci) * 25 + ky * 5 + kx], g);
                }
        a.G[sn * G::DIN + pix] = g;
    }
}

// pool-1 routing + conv1^T on the matrix pipe (the dispatched version; conv1_bwd_kernel above is its VALU gather form:
// 3.32 ms at the conv-512 bench against 1.24 ms for this one; 15.4 -> see profiles at the CIFAR-shaped c5 bench).
// One WAVE = one (sample, point), no barriers.  Per input channel ci:
//   dX[ci][Y][X] = sum_{c, ky, kx} w[c][ci][ky][kx] * R[c][Y - ky][X - kx],   R = the pooled gradient routed to its argmax position
//   (O1 x O1, one non-zero per 2x2 cell) times the activation derivative.
// Per position row Ya the wave forms  T[tap][Xa] = sum_c w[c][ci][tap] * R[c][Ya][Xa]  — a 32(taps, 25 used) x 32(c) x 32(Xa, O1 used)
// fp32 MFMA product whose B operand is built in registers from the stash bytes and pooled gradients of pooled row Ya/2 — keeps the
// last five rows of T in an LDS ring, and emits output row Y = Ya as a 25-term gather  dX[Y][X] = sum_tap T[tap][Y - ky][X - kx].
// With Cin > 1 the pass is repeated per input channel (the routed B operand is rebuilt from L2-resident data; the ring stays 62.5 KiB).
template <class G> constexpr int conv1_bwd_ts() { return G::O1 % 8 == 0 ? G::O1 + 2 : G::O1; }
template <int ACT, class G>
__global__ void __launch_bounds__(256, (G::CIN == 1 ? 3 : 2)) conv1_bwd_mfma_kernel(const ConvBwdArgs a) {
    // T row buffer of ONE position row: [25 taps][TS Xa]; TS >= O1 with 4 * TS = 8 or 16 mod 32 (26 for O1 = 24, 28 for O1 = 28) keeps the
    // accumulator stores at the 2-way minimum of a 64-lane ds_write_b32 (the four tap groups of a store land on different banks).
    // Round 4: the last five rows of T used to sit in an LDS ring (14 KB per wave and input channel: two blocks per CU at 3x32x32, and the pass
    // was repeated per input channel with the routed B operand rebuilt each time).  Now a lane keeps the five PARTIAL output rows that a T row
    // contributes to in registers (ring[ci][ky]: column X = lane, output row Ya + ky), T lives in LDS for one row only (2.8 KB per wave),
    // and the input-channel loop is INSIDE the row loop: stash bytes / pooled gradients are fetched and routed once per row, not Cin times.
    constexpr int O1 = G::O1, P1W_ = G::P1W, IW = G::IW, CIN = G::CIN;
    constexpr int TS = conv1_bwd_ts<G>(), TROW = 25 * TS;
    static_assert(O1 <= TS && (4 * TS) % 32 != 0 && O1 <= 32 && IW <= 64, "a conv1 output row fits two 16-wide MFMA tiles; an input row fits one wave");
    __shared__ float lds[4 * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int NB = (a.N + 3) / 4;
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, n = (id % NB) * 4 + wave;
    if (n >= a.N) return;                                                // whole wave idle; no block barrier anywhere
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    float* const T = lds + wave * TROW;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;
    int xcl[5];                                                           // the gather's column X - kx, clamped into the T row, and whether it lies inside
    bool xok[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) { xok[kx] = lane - kx >= 0 && lane - kx <= G::O1 - 1; xcl[kx] = min(max(lane - kx, 0), G::O1 - 1); }

    // this lane's 16 (channel, position) elements of a pooled row: pt = position tile (Xa = 16pt + li), channel c = 16kb + 4lg + r
    int eoff[2][2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) eoff[pt][kb][r] = (16 * kb + 4 * lg + r) * (P1W_ * P1W_) + min((16 * pt + li) >> 1, P1W_ - 1);
    const uint8_t* const st_sn = a.st1 + sn * G::P1SZ;
    const float* const d_sn = a.dP1 + sn * G::P1SZ;

    // A operand per input channel: A[i = tap][k = c], K step r of channel block kb is c = 16kb + 4lg + r (one f32x4 per (tap tile, channel block))
    f32x4 aw[CIN][2][2];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tap = 16 * mt + li, c = 16 * kb + 4 * lg + r;
                    aw[ci][mt][kb][r] = tap < 25 ? a.K1w[((long long)sw * C1 + c) * G::K1 + ci * 25 + tap] : 0.f;
                }
    float ring[CIN][5];                                                  // partial sums of output rows Ya .. Ya + 4, column X = lane
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int k = 0; k < 5; ++k) ring[ci][k] = 0.f;
    float* const Gout = a.G + sn * G::DIN;
    // the stash bytes and pooled gradients of pooled row py + 1 are fetched under the matrix work of row py (the wave has nothing
    // else in flight: without this every row paid a full memory round trip, 1.24 ms at the conv-512 bench)
    int stn[2][2][4];
    float dn[2][2][4];
    auto fetch = [&](int py) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    stn[pt][kb][r] = st_sn[eoff[pt][kb][r] + py * P1W_];
                    dn[pt][kb][r] = d_sn[eoff[pt][kb][r] + py * P1W_];
                }
    };
    fetch(0);
    for (int py = 0; py < P1W_; ++py) {
        float gv[2][2][4];
        int ar[2][2][4];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int Xa = 16 * pt + li, st = stn[pt][kb][r];
                    const float d = dn[pt][kb][r];
                    gv[pt][kb][r] = (Xa < O1) ? ((smooth_act<ACT>() || (st & 4)) ? d : d * slope) : 0.f;
                    ar[pt][kb][r] = (st & 3) ^ (Xa & 1);                 // == 2*half for the row half that owns the argmax, with the right column parity
                }
        if (py + 1 < P1W_) fetch(py + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int Ya = 2 * py + half;
            float b[2][2][4];                                            // the routed gradient row: B[k = channel][j = Xa], built ONCE for all input channels
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) b[pt][kb][r] = (ar[pt][kb][r] == 2 * half) ? gv[pt][kb][r] : 0.f;   // arg = 2*dy + dx
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                f32x4 acc[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) acc[mt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) acc[mt][pt] = MFMA16(aw[ci][mt][kb][r], b[pt][kb][r], acc[mt][pt]);
                // (same wave: the LDS unit serves its requests in order — these stores follow the previous channel's reads, the reads below follow them)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int tap = 16 * mt + 4 * lg + r;        // acc[mt][pt][r] = T[tap][Xa = 16pt + li]
                            if (tap < 25 && 16 * pt + li < TS) T[tap * TS + 16 * pt + li] = acc[mt][pt][r];
                        }
                if (lane < IW) {                                         // this T row's share of output rows Ya + ky: dX[Ya + ky][X] += T[(ky, kx)][X - kx]
                    // all 25 reads first, UNCONDITIONAL at a clamped column, then the adds with the terms outside the row selected to +0: written as
                    // `if (inside) ring += T[..]` every term became its own exec-masked block with its own wait — 25 dependent LDS round trips per
                    // (row, input channel), ~90 % of this kernel's time (rocprofv3 + the instruction mix, round 4 third part)
                    // (two tap rows = ten reads at a time, fenced: all 25 at once cost 24 registers and a wave per SIMD)
#pragma unroll
                    for (int ky0 = 0; ky0 < 5; ky0 += 2) {
                        float tv[2][5];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int kx = 0; kx < 5; ++kx) if (ky0 + k2 < 5) tv[k2][kx] = T[((ky0 + k2) * 5 + kx) * TS + xcl[kx]];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int kx = 0; kx < 5; ++kx) if (ky0 + k2 < 5) ring[ci][ky0 + k2] += xok[kx] ? tv[k2][kx] : 0.f;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // output row Ya has now received its last contribution (T rows Ya - 4 .. Ya): emit it, rotate the partial rows
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                if (lane < IW) Gout[ci * (IW * IW) + Ya * IW + lane] = ring[ci][0];
#pragma unroll
                for (int k = 0; k < 4; ++k) ring[ci][k] = ring[ci][k + 1];
                ring[ci][4] = 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)                                          // the last four output rows: O1 .. IW - 1
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            if (lane < IW) Gout[ci * (IW * IW) + (O1 + k) * IW + lane] = ring[ci][k];
}

// conv1^T on the F16 matrix pipe (triple-split arithmetic; round 4, second half): conv1_bwd_mfma_kernel's structure — one wave = one (sample,
// point), a row of T per conv1 output row, five partial output rows per lane in registers — with the 32-channel contraction as ONE K step of
// v_mfma_f32_16x16x32_f16 (six exact product terms per fp32 product; the fp32 form needs eight K steps of 16x16x4 at twice the cycles each:
// 1,024 matrix-pipe cycles per (row, input channel) against 384 here) and the rows of T packed over (input channel, tap): 75 rows = five
// tiles at 3x32x32 (six before), one pass per position row for all input channels.  K index k = 8 lg + e of a lane <-> channel
// c = 16 (e >> 2) + 4 lg + (e & 3): exactly the (channel block, quad) elements the routing already holds per lane.  Scales: the weights by
// the wave's own max |w| (its A operand is the sample's whole conv1 weight tensor); the routed gradients by max |dP1| of the (sample,
// point), which the dense conv2^T kernel leaves in G[sn][0] (this kernel reads it before it writes G) — so only rbnn_conv_input_grad_dense
// launches this kernel, and only for more than one input channel (launch_conv1_backward); the other conv2^T forms keep the fp32 one.
template <int ACT, class G>
__global__ void __launch_bounds__(256, 2) conv1_bwd_x3_kernel(const ConvBwdArgs a) {
    constexpr int O1 = G::O1, P1W_ = G::P1W, IW = G::IW, CIN = G::CIN;
    constexpr int MROWS = CIN * 25, MT = (MROWS + 15) / 16;
    constexpr int TS = conv1_bwd_ts<G>(), TROW = MT * 16 * TS;            // all MT * 16 rows exist: the accumulator stores need no row test
    static_assert(O1 <= TS && (4 * TS) % 32 != 0 && O1 <= 32 && IW <= 64, "a conv1 output row fits two 16-wide MFMA tiles; an input row fits one wave");
    __shared__ float lds[4 * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int NB = (a.N + 3) / 4;
    int id;
    if (!item_of_block(blockIdx.x, NB * a.S, id)) return;
    const int s = id / NB, n = (id % NB) * 4 + wave;
    if (n >= a.N) return;                                                // whole wave idle; no block barrier anywhere
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    float* const T = lds + wave * TROW;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;
    int xcl[5];                                                           // the gather's column X - kx, clamped into the T row, and whether it lies inside
    bool xok[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) { xok[kx] = lane - kx >= 0 && lane - kx <= G::O1 - 1; xcl[kx] = min(max(lane - kx, 0), G::O1 - 1); }
    float* const Gout = a.G + sn * G::DIN;
    const float gmax = Gout[0];                                          // max |dP1| of this (sample, point), from conv_bwd_dense_x3_kernel

    int eoff[2][2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) eoff[pt][kb][r] = (16 * kb + 4 * lg + r) * (P1W_ * P1W_) + min((16 * pt + li) >> 1, P1W_ - 1);
    const uint8_t* const st_sn = a.st1 + sn * G::P1SZ;
    const float* const d_sn = a.dP1 + sn * G::P1SZ;

    // A operand: row m = ci * 25 + tap of tile mt (a channel's weights are contiguous over m), K element e of this lane = channel 16 (e >> 2) + 4 lg + (e & 3)
    union F8 { f16x8 v; unsigned w[4]; };
    F8 aw0[MT], aw1[MT], aw2[MT];
    float w_inv;
    {
        float wv[MT][8], wmax = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = 16 * mt + li, c = 16 * (e >> 2) + 4 * lg + (e & 3);
                wv[mt][e] = m < MROWS ? a.K1w[((long long)sw * C1 + c) * G::K1 + m] : 0.f;
                wmax = fmaxf(wmax, fabsf(wv[mt][e]));
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        int ew = 0;
        if (wmax > 0.f && wmax < INFINITY) ew = max(-100, min(100, 13 - ilogbf(wmax)));
        const float wsc = ldexpf(1.f, ew);
        w_inv = ldexpf(1.f, -ew);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                split3_plain_pair(wv[mt][e] * wsc, wv[mt][e + 1] * wsc, 1.f, aw0[mt].w[e >> 1], aw1[mt].w[e >> 1], aw2[mt].w[e >> 1]);
    }
    int eg = 0;
    if (gmax > 0.f && gmax < INFINITY) eg = max(-100, min(100, 13 - ilogbf(gmax)));
    const float g_scale = ldexpf(1.f, eg), t_scale = ldexpf(1.f, -eg) * w_inv;

    float ring[CIN][5];                                                  // partial sums of output rows Ya .. Ya + 4, column X = lane
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int k = 0; k < 5; ++k) ring[ci][k] = 0.f;
    int stn[2][2][4];
    float dn[2][2][4];
    auto fetch = [&](int py) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    stn[pt][kb][r] = st_sn[eoff[pt][kb][r] + py * P1W_];
                    dn[pt][kb][r] = d_sn[eoff[pt][kb][r] + py * P1W_];
                }
    };
    fetch(0);
    for (int py = 0; py < P1W_; ++py) {
        float gv[2][2][4];
        unsigned arp = 0;                                                // 2 bits per element (16 registers as ints: the difference between one and two waves per SIMD here)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int Xa = 16 * pt + li, st = stn[pt][kb][r];
                    const float d = dn[pt][kb][r] * g_scale;
                    gv[pt][kb][r] = (Xa < O1) ? ((smooth_act<ACT>() || (st & 4)) ? d : d * slope) : 0.f;
                    arp |= (unsigned)((st & 3) ^ (Xa & 1)) << (2 * ((pt * 2 + kb) * 4 + r));   // == 2*half for the row half that owns the argmax, with the right column parity
                }
        auto ar = [&](int pt, int kb, int r) { return (int)((arp >> (2 * ((pt * 2 + kb) * 4 + r))) & 3u); };
        if (py + 1 < P1W_) fetch(py + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int Ya = 2 * py + half;
            F8 b0[2], b1[2], b2[2];                                      // the routed gradient row, three piece planes: B[k][j = Xa], built ONCE for all input channels
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float ve = (ar(pt, e >> 2, e & 3) == 2 * half) ? gv[pt][e >> 2][e & 3] : 0.f;               // arg = 2*dy + dx
                    const float vo = (ar(pt, (e + 1) >> 2, (e + 1) & 3) == 2 * half) ? gv[pt][(e + 1) >> 2][(e + 1) & 3] : 0.f;
                    split3_plain_pair(ve, vo, 1.f, b0[pt].w[e >> 1], b1[pt].w[e >> 1], b2[pt].w[e >> 1]);
                }
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};                      // smallest terms first (as everywhere in the triple-split kernels)
                    c = MFMA_H(aw2[mt].v, b0[pt].v, c);
                    c = MFMA_H(aw1[mt].v, b1[pt].v, c);
                    c = MFMA_H(aw0[mt].v, b2[pt].v, c);
                    c = MFMA_H(aw1[mt].v, b0[pt].v, c);
                    c = MFMA_H(aw0[mt].v, b1[pt].v, c);
                    acc[mt][pt] = MFMA_H(aw0[mt].v, b0[pt].v, c);
                }
            // (same wave: the LDS unit serves its requests in order — these stores follow the previous row's reads, the reads below follow them)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * pt + li < TS) T[(16 * mt + 4 * lg + r) * TS + 16 * pt + li] = acc[mt][pt][r];   // T[m][Xa]
            if (lane < IW) {                                             // this T row's share of output rows Ya + ky: dX[ci][Ya + ky][X] += T[ci * 25 + (ky, kx)][X - kx]
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {                       // (unconditional reads at clamped columns, then selected adds: see conv1_bwd_mfma_kernel)
#pragma unroll
                    for (int ky0 = 0; ky0 < 5; ky0 += 2) {               // (two tap rows = ten reads at a time, fenced: registers)
                        float tv[2][5];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int kx = 0; kx < 5; ++kx) if (ky0 + k2 < 5) tv[k2][kx] = T[(ci * 25 + (ky0 + k2) * 5 + kx) * TS + xcl[kx]];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int kx = 0; kx < 5; ++kx) if (ky0 + k2 < 5) ring[ci][ky0 + k2] += xok[kx] ? tv[k2][kx] : 0.f;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // output row Ya has now received its last contribution (T rows Ya - 4 .. Ya): emit it, rotate the partial rows
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                if (lane < IW) Gout[ci * (IW * IW) + Ya * IW + lane] = ring[ci][0] * t_scale;
#pragma unroll
                for (int k = 0; k < 4; ++k) ring[ci][k] = ring[ci][k + 1];
                ring[ci][4] = 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)                                          // the last four output rows: O1 .. IW - 1
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            if (lane < IW) Gout[ci * (IW * IW) + (O1 + k) * IW + lane] = ring[ci][k] * t_scale;
}

// conv1^T: the matrix-pipe kernel; RBNN_CONV1_BWD_VALU keeps its VALU gather form selectable (same results up to summation order)
template <int ACT, class G>
int launch_conv1_backward(const ConvBwdArgs& a, hipStream_t st, bool dp1_max_in_g = false) {
#if RBNN_CONV1_BWD_X3
    if (dp1_max_in_g && G::CIN >= RBNN_CONV1_BWD_X3_MINCIN) {                                    // one input channel (1x28x28): two row tiles either way, and the fp32 kernel runs three waves per SIMD — measured 0.73 against 0.79 ms
        hipLaunchKernelGGL((conv1_bwd_x3_kernel<ACT, G>), dim3(grid_for_items((long long)((a.N + 3) / 4) * a.S)), dim3(256), 0, st, a);
        return launch_status();
    }
#endif
#ifndef RBNN_CONV1_BWD_VALU
    hipLaunchKernelGGL((conv1_bwd_mfma_kernel<ACT, G>), dim3(grid_for_items((long long)((a.N + 3) / 4) * a.S)), dim3(256), 0, st, a);
#else
    hipLaunchKernelGGL((conv1_bwd_kernel<ACT, G>), dim3((unsigned)((long long)a.S * a.N)), dim3(256), 0, st, a);
#endif
    return launch_status();
}

}  // namespace

namespace {
template <int ACT, class G>
int launch_conv_backward(const ConvBwdArgs& a, hipStream_t st) {
    int rc;
    {
        const long long F = (long long)a.Hc * G::NP2, items = (long long)a.S * ((a.N + 15) / 16) * ((F + 63) / 64);
        if (items >= (1LL << 31)) return (int)RBNN_ERR_SHAPE;                     // conv_fc_bwd_kernel decodes its item number in 32 bits
        hipLaunchKernelGGL((conv_fc_bwd_kernel<smooth_act<ACT>(), ACT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
        if ((rc = launch_status())) return rc;
    }
    const int grid = grid_for_items((long long)((a.N + 3) / 4) * a.S);
    constexpr int LDSB = conv_bwd_lds_floats<G>() * 4;
    static unsigned long long attr = 0;
    if (!ensure_dynamic_lds((const void*)conv_bwd_kernel<ACT, G>, LDSB, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_bwd_kernel<ACT, G>), dim3(grid), dim3(256), LDSB, st, a);
    if ((rc = launch_status())) return rc;
    return launch_conv1_backward<ACT, G>(a, st);
}
}  // namespace

extern "C" int rbnn_conv_input_grad(const rbnn_conv_posterior* net, const int32_t* sidx, int32_t S, int32_t N,
                                    const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!net->K2w_ci || !ws || !ws->dZ || !ws->P1 || !ws->Q2 || !ws->st1 || !ws->st2 || !ws->G) return RBNN_ERR_NULL;
    if (N < 1 || S < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(net->K2w_ci) || !aligned16(ws->G)) return RBNN_ERR_ALIGN;
    ConvBwdArgs a = {};
    a.dZ = ws->dZ; a.st1 = ws->st1; a.st2 = ws->st2; a.K1w = net->K1w; a.K2cb = net->K2w_ci; a.Fw = net->Fw;
    a.Hc = net->hidden; a.C = net->n_classes; a.N = N; a.S = S; a.sidx = sidx; a.dQ2 = ws->Q2; a.dP1 = ws->P1; a.G = ws->G;
    hipStream_t st = (hipStream_t)stream;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        a.NP2 = G::NP2;
        return for_activation(net->activation, [&](auto act) { return launch_conv_backward<decltype(act)::value, G>(a, st); });
    });
}

extern "C" int rbnn_conv_input_grad_split(const rbnn_conv_posterior* net, const void* K2_bwd, int32_t k2_exp, float fw_l1,
                                          const int32_t* sidx, int32_t S, int32_t N, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv_split(net);
    if (rc) return rc;
    if (!K2_bwd || !ws || !ws->dZ || !ws->P1 || !ws->Q2 || !ws->st1 || !ws->st2 || !ws->G) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || k2_exp < -100 || k2_exp > 100 || !(fw_l1 >= 0.f)) return RBNN_ERR_SHAPE;
    if (!aligned16(K2_bwd) || !aligned16(ws->G)) return RBNN_ERR_ALIGN;
    ConvBwdArgs a = {};
    a.dZ = ws->dZ; a.st1 = ws->st1; a.st2 = ws->st2; a.K1w = net->K1w; a.K2cb = nullptr; a.Fw = net->Fw;
    a.Hc = net->hidden; a.C = net->n_classes; a.N = N; a.S = S; a.sidx = sidx; a.dQ2 = ws->Q2; a.dP1 = ws->P1; a.G = ws->G; a.NP2 = NP2;
    const int grid = grid_for_items((long long)((N + 3) / 4) * S);
    const bool leaky = net->activation == RBNN_ACT_LEAKY;
    hipStream_t st = (hipStream_t)stream;
    {
        const long long F = (long long)net->hidden * NP2, items = (long long)S * ((N + 15) / 16) * ((F + 63) / 64);
        if (items >= (1LL << 31)) return (int)RBNN_ERR_SHAPE;                     // conv_fc_bwd_kernel decodes its item number in 32 bits
        hipLaunchKernelGGL((conv_fc_bwd_kernel<false, RBNN_ACT_LEAKY>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
        if ((rc = launch_status())) return rc;
    }
    if (leaky) hipLaunchKernelGGL(conv_bwd_split_kernel<RBNN_ACT_LEAKY>, dim3(grid), dim3(256), 0, st, a, (const char*)K2_bwd, k2_exp, fw_l1);
    else       hipLaunchKernelGGL(conv_bwd_split_kernel<RBNN_ACT_RELU>, dim3(grid), dim3(256), 0, st, a, (const char*)K2_bwd, k2_exp, fw_l1);
    if ((rc = launch_status())) return rc;
    if (leaky) return launch_conv1_backward<RBNN_ACT_LEAKY, GeoMnist>(a, st);
    return launch_conv1_backward<RBNN_ACT_RELU, GeoMnist>(a, st);
}

// =====================================================================================================
// conv2^T, DENSE form (triple-split arithmetic), 1x28x28: a GEMM per tap over the conv2 OUTPUT positions + col2im, instead of the
// gather form of conv_bwd_x3_kernel, whose zero-padded gradient image makes 36-39 % of its MFMAs multiply padding:
//
//     T[tap][ci][pos2] = sum_hc W[hc][ci][tap] * dO2[hc][pos2]          pos2 over the 8 x 8 conv2 outputs: M = 32 ci, N = 64, K = Hc
//     dP1[ci][y + ky][x + kx] += T[(ky, kx)][ci][(y, x)]                 col2im, once per (sample, point)
//
// Every MFMA is useful (25 taps x 2 ci tiles x 4 position tiles x 6 product terms per 32 channels: 1200, against ~1970 issued by the
// gather form).  One block = one (sample, point), 8 waves = 2 input-channel tiles x 4 tap groups (7 + 6 + 6 + 6 taps; the 7-tap groups
// of the two channel tiles sit on different SIMDs): a wave holds T for its taps in <= 28 accumulator tiles over the WHOLE Hc loop.  Per
// K step of 32 channels the block (a) routes the pooled gradients of those channels through the pool-2 argmax / activation derivative
// into a dense channel-last image [64 positions][32 hc] of three fp16 piece planes (12 KB; the staging of dQ2 / stash rows by 4-byte
// LDS-DMA and the image are double-buffered: one barrier per K step), (b) reads its B fragments ONCE (12 ds_read_b128: a fragment
// feeds 7 taps = 42 MFMAs, against 12 in the gather form, which was LDS-bandwidth bound) and its A fragments — model.3.weight regrouped
// [K step][tap][ci][32 hc] as a triple-rows image — straight from memory (L2: 2.4 MB per block, as the gather form).
// Epilogue: the T tiles of one channel tile go to LDS ([25 taps][16 ci][64 pos] floats, aliasing the loop buffers) and every thread gathers
// its dP1 outputs as a fixed-order sum of <= 25 terms (deterministic, no atomics); twice (two channel tiles).
// =====================================================================================================
#ifndef RBNN_CONV_BWD_DENSE
#define RBNN_CONV_BWD_DENSE 1
#endif
// RBNN_DENSE_STAMPS (diagnostic build, fenced like the ablation switches; the results stay right): s_memtime stamps of waves 0 and 3 of every
// block, summed per segment into rbnn_dense_stamp_acc and read back by rbnn_debug_dense_stamps (tools/dense_stamps.py).  1: per-pass prologue /
// K loop / col2im; 2: also the time inside the K loop's barrier (each stamp drains lgkmcnt: level 2 perturbs the loop it measures).
#ifdef RBNN_DENSE_STAMPS
__device__ unsigned long long rbnn_dense_stamp_acc[64];
#define DSTAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (lane == 0 && (wave == 0 || wave == 3)) atomicAdd(&rbnn_dense_stamp_acc[(wave ? 32 : 0) + (slot)], t_ - tprev); tprev = t_; } while (0)
#define DSTAMP_ADD(slot, v) do { if (lane == 0 && (wave == 0 || wave == 3)) atomicAdd(&rbnn_dense_stamp_acc[(wave ? 32 : 0) + (slot)], (unsigned long long)(v)); } while (0)
#else
#define DSTAMP(slot) do { } while (0)
#define DSTAMP_ADD(slot, v) do { } while (0)
#endif
#ifndef RBNN_DENSE_STAGGER
#define RBNN_DENSE_STAGGER 0                                               // 1: the two waves of a SIMD route one tap group apart (the 6-tap loop has room for one)
#endif
template <class G> struct ConvBwdDenseLds {
    // The conv2 output positions are covered in PASSES of <= 4 position tiles (64 positions): 7 taps x 4 tiles is what a wave's accumulators
    // hold (112 registers).  1x28x28: 64 positions, one pass.  3x32x32: 100 positions = pass 0 (positions 0..63) + pass 1 (64..99, three
    // tiles); a pass streams the weights once more (L2) but routes only its own positions, and the two passes' col2im partial sums meet in
    // registers (16 per gathering thread) — splitting the block over input-channel tiles or taps instead would route everything twice.
    static constexpr int NPASS = (G::NPT2 + 3) / 4;
    static constexpr int NPOSP = 64;                                      // positions of a pass, padded to whole MFMA tiles
    static constexpr int PLANE = NPOSP * 64, IMG = 3 * PLANE;             // one piece plane: NPOSP records of 32 hc halves
    static constexpr int NFL = 32 * G::NP2;                               // pooled cells of one K step
    static constexpr int STG = (NFL * 5 + 4 * G::NP2 + 15) / 16 * 16;     // staging: NFL dQ2 floats + NFL stash bytes + 4 NP2 bytes that hold the code 8 (no window's: the
                                                                          // windows of a position that lie off the pooled map read their stash byte here)
    static constexpr int RING = 4, SLOT = 3 * 1024;                       // per wave: RING weight tiles (one tap x 16 ci x 32 hc: three 1-KiB plane tiles)
    static constexpr int AOFF = 2 * IMG;                                  // the eight waves' rings follow the images;
    static constexpr int SOFF = AOFF + 8 * RING * SLOT;                   // the staging buffers come LAST, above the col2im images (EPI): a pass's first two K steps
    static constexpr int LOOP = SOFF + 2 * STG;                           //   are staged before the previous pass's col2im and land under it
#if RBNN_DENSE_COL2IM_RMW
    static constexpr int EIMG = G::P1W * G::P1W * 64;                     // col2im: one wave's partial gradient image [P1W x P1W output positions][16 ci] floats
    static constexpr int EPI = 8 * EIMG;
#else
    static constexpr int EPI = 25 * NPOSP * 16 * 4;                       // T of one channel tile and pass: [25 taps][pos][16 ci] floats
#endif
    static constexpr int BYTES = LOOP > EPI ? LOOP : EPI;
    static_assert(EPI <= SOFF, "the staging buffers must survive the col2im");
    static_assert(EPI + 64 + 8 * 1024 <= SOFF, "room for the eight wave maxima and the waves' dummy records behind the col2im images");
    static_assert(BYTES <= 160 * 1024, "LDS");
    static_assert(NPASS * 64 >= G::NPOS && (NPASS - 1) * 64 < G::NPOS, "passes of 64 positions");
};

template <int ACT, class G>
__global__ void __launch_bounds__(512, 2) conv_bwd_dense_x3_kernel(const ConvBwdArgs a, const char* __restrict__ K2d, int k2_exp, float fw_l1) {
    using L = ConvBwdDenseLds<G>;
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NP2_ = G::NP2, NPOS_ = G::NPOS, NFL = L::NFL, NPASS = L::NPASS;
    constexpr int SMIN = NFL / 512 + (NFL / 4) / 512;                      // staging pieces every wave issues for a whole K step (stage_issue: its rounds of 512 lanes that lie inside the step entirely)
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* const lds = (char*)lds_f;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave >> 2, q = wave & 3;
    // taps of this wave: 7 for q == ct (waves 0 and 5: SIMDs 0 and 1), 6 for the others, in tap order
    const int ntap = 6 + (q == ct ? 1 : 0);
    const int tap0 = 6 * q + (q > ct ? 1 : 0);

    int id;
    if (!item_of_block(blockIdx.x, a.N * a.S, id)) return;
    const int s = id / a.N, n = id % a.N;
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2_, KS = (a.Hc + 31) / 32;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;
#ifdef RBNN_DENSE_STAMPS
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
    unsigned long long barw = 0;
#endif

    // per-(sample, point) scale: |dO2| <= 4 * max_c |dZ_c| * fw_l1 (as conv_bwd_x3_kernel).  The load is issued here; the scales are formed in the first
    // pass's prologue, behind the staging / tile DMA issue (forming them here put the load's round trip in front of the DMA's)
    const float dz_lane = a.dZ[sn * RBNN_CPAD + li];
    float in_scale = 1.f, out_scale = 1.f;
    auto set_scales = [&]() {
        float dzmax = fabsf(dz_lane);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, o));
        const float bound = 4.f * dzmax * fw_l1;
        int e = 0;
        if (bound > 0.f && bound < INFINITY) e = max(-100, min(100, 13 - ilogbf(bound)));
        e = __builtin_amdgcn_readfirstlane(e);                             // wave-uniform: the two scales live in scalar registers
        in_scale = ldexpf(1.f, e);
        out_scale = ldexpf(1.f, -(e + k2_exp));
    };

    auto dma4 = [&](const void* g, void* l) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 4, 0, 0);
    };
    // (wholec = std::true_type: the step is known to be a whole one — Hc % 32 == 0 —, so the pieces that lie inside a whole step entirely need no
    // per-lane test: the K loop's calls; each tested piece is an exec-masked block of five instructions around its DMA)
    auto stage_issue = [&](int ks, int buf, auto wholec) {                  // rows of channels 32ks .. of dQ2 and of the stash: contiguous in memory
        constexpr bool WH = decltype(wholec)::value;
        char* const S = lds + L::SOFF + buf * L::STG;
        const int nvalid = min(32, a.Hc - 32 * ks) * NP2_;                  // a multiple of 4 (Hc % 16 == 0)
        const long long fb = sn * F + (long long)ks * NFL;
        // sources = a wave-uniform 64-bit base (the step's rows) + ONE 32-bit per-lane offset: the SGPR-base addressing form, no 64-bit vector add per piece
        const char* const qrow = (const char*)(a.dQ2 + fb);
        const char* const srow = (const char*)(a.st2 + fb);
        const unsigned l4 = 4u * (unsigned)lane;
        static_for<0, (NFL + 511) / 512>([&](auto I) {                       // pieces 2 KiB apart on both sides: pairs share address and M0
            constexpr int i = decltype(I)::value, i0 = i & ~1;
            const int b = 512 * i0 + 64 * wave;                             // wave-uniform destination base
            if ((WH && 512 * i + 512 <= NFL) || b + 512 * (i - i0) + lane < nvalid)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qrow + (4u * (unsigned)b + l4)),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)((float*)S + b), 4, (i - i0) * 2048, 0);
        });
        static_for<0, (NFL / 4 + 511) / 512>([&](auto I) {                   // stash: one dword (4 cells) per lane, 512 lanes per round
            const int d = 512 * decltype(I)::value + 64 * wave;
            if ((WH && 512 * decltype(I)::value + 512 <= NFL / 4) || 4 * (d + lane) < nvalid) dma4(srow + (4u * (unsigned)d + l4), S + NFL * 4 + 4 * d);
        });
    };
    // A operand: the wave's weight tiles (tap, 16 ci of its channel tile, 32 hc of the K step: 16 rows x 192 B) come in by LDS-DMA into a
    // private ring of RING slots, three 1-KiB pieces per tile (one per plane: lane p lands at row p >> 2, physical chunk p & 3 and
    // fetches logical chunk (p & 3) ^ swz(row)), issued THREE taps ahead of their use: an L2 round trip (500+ cycles) is far longer than
    // the 24 MFMAs of a tap, and registers for a deeper prefetch do not exist (112 accumulators + 48 B-fragment registers).  The
    // wave's running tap index g = ks * ntap + i names the slot g & 3; nothing but this wave touches its ring (no barrier involved).
    // (weight tiles are PLANE-major in memory — [tile][3 pieces][16 ci][64 B], conv.py::_build_dense — exactly as they sit in a ring slot: the
    // immediate offset of global_load_lds applies to the global AND the LDS address, so a tile's three pieces share one address and one M0)
    const int prow = lane >> 2;
    const unsigned a_lane = (unsigned)(prow * 64 + (((lane & 3) ^ swz(prow)) * 16));    // per-lane part of a piece's source address
    const char* const Awave = K2d + (((long long)sw * KS * 25) * 32 + 16 * ct) * 192;   // wave-uniform part (SGPR pair): the DMA needs no vector address arithmetic
    char* const ring = lds + L::AOFF + wave * (L::RING * L::SLOT);
    const int foff = li * 64 + ((lg ^ swz(li)) * 16);                       // fragment of row li (position / input channel), K chunk lg
    constexpr int NPP = P1W_ * P1W_;
    static_assert(2 * NPP <= 512, "one thread per output position and pair of channel quads");
    f32x4 part[2][2];                                                      // NPASS > 1: col2im partial sums of the earlier passes ([channel tile][quad of the pair])
#pragma unroll
    for (int i = 0; i < 4; ++i) part[i >> 1][i & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef RBNN_DENSE_ABL_NOMFMA
#define DENSE_MFMA(A, B, C) (C)
#else
#define DENSE_MFMA(A, B, C) MFMA_H(A, B, C)
#endif

    static_for<0, NPASS>([&](auto PASS) {
    constexpr int pass = decltype(PASS)::value;
    constexpr int NPT = (G::NPT2 - 4 * pass) < 4 ? (G::NPT2 - 4 * pass) : 4;   // position tiles of this pass
    if (pass) __syncthreads();                                             // the previous pass's T (LDS) has been gathered
    // routing role of this thread: position gp = 64 * pass + lane (gy, gx) of the O2W x O2W gradient map, channel quad qd = wave of the K step's 32
    const int gp = 64 * pass + lane, gy = gp / O2W_, gx = gp % O2W_, qd = wave;
    // byte offsets of this thread's four windows' cells inside a staging buffer (channel j of its quad: + j * NP2 cells).  The stash code
    // that routes window w here is w (argmax) with bit 2 = the pre-activation was positive; stash bytes are <= 7 by construction, so they
    // are compared WHOLE (no masking), and a window off the map (or a lane past the last position: its image row is zeros) reads the byte 8
    int adq[4], ast[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {                                          // window w = 2dy + dx of the <= 4 stride-1 pooling windows containing (gy, gx)
        const int py = gy - (w >> 1), px = gx - (w & 1);
        const bool ok = gp < NPOS_ && py >= 0 && py < P2W_ && px >= 0 && px < P2W_;
        const int cell = 4 * qd * NP2_ + (ok ? py * P2W_ + px : 0);
        adq[w] = 4 * cell;
        ast[w] = ok ? NFL * 4 + cell : NFL * 5;                            // off the map (or a lane past the last position): the never-matching byte (+ j NP2 <= 4 NP2 of them)
    }
    const int rec = lane * 64 + (((qd >> 1) ^ swz(lane)) * 16) + (qd & 1) * 8;   // this thread's 8 bytes of a piece plane
    // routing of ONE channel (j of this thread's quad) of K step ks from staging buffer sbuf: pool-2 argmax + activation derivative (gather
    // form), scaled, split into the three pieces.  Called between the MFMA groups of the previous K step so that its LDS reads and
    // vector work issue under the matrix pipe (one basic block with the MFMAs: no branch in between).
    union Q { unsigned w[2]; uint2 u; };                                    // pieces of the thread's four channels: [j0 | j1 << 16], [j2 | j3 << 16]
    float vpend = 0.f;                                                     // the even channel of a pair waits for the odd one (split3_plain_pair)
    // (two halves: the eight LDS reads of a channel are issued one tap group AHEAD of the selects that consume them — in one piece the selects
    // waited for the reads right in front of the tap's MFMAs, an LDS round trip per routing tap with nothing issued behind it)
    struct RouteIn { int st[4]; float dq[4]; };
    auto route_load = [&](int sbuf, int j, RouteIn& in) {
        const char* const sb = lds + L::SOFF + sbuf * L::STG;
#pragma unroll
        for (int w = 0; w < 4; ++w) {                                      // eight independent LDS reads
            in.st[w] = *(const unsigned char*)(sb + ast[w] + j * NP2_);
            in.dq[w] = *(const float*)(sb + adq[w] + 4 * j * NP2_);
        }
    };
    auto route_calc = [&](int ks, int j, const RouteIn& in, Q& p0, Q& p1, Q& p2) {
        const bool live = 32 * ks + 4 * qd + j < a.Hc;                     // wave-uniform; channels past Hc (and the step past the last): zeros
        const int (&st)[4] = in.st;
        const float (&dq)[4] = in.dq;
        // window w routes here iff its stashed argmax is w; act' = 1 or the slope by bit 2 of the stash (folding act' into dQ2 in
        // conv_fc_bwd_kernel instead was measured: that kernel went from 1.1 to 2.35 ms on its byte loads of the stash).  ReLU: both
        // tests are one compare of the stash's low three bits (a cell whose pre-activation was <= 0 passes nothing on).
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {                                      // (w: a constant after unrolling — compares against immediates)
            if (ACT == RBNN_ACT_RELU) {
                v += st[w] == (w | 4) ? dq[w] : 0.f;
            } else if (ACT == RBNN_ACT_LEAKY) {                            // factor 1 / slope / 0
                v = fmaf(dq[w], st[w] == (w | 4) ? 1.f : (st[w] == w ? slope : 0.f), v);
            } else {
                v += (st[w] & 11) == w ? dq[w] : 0.f;                      // sigmoid / tanh: act' is already folded into dQ2; bit 2 of the stash is ignored, bit 3 marks the dummy
            }
        }
        const float vs = (live ? v : 0.f) * in_scale;
        if (!(j & 1)) vpend = vs;
        else split3_plain_pair(vpend, vs, 1.f, p0.w[j >> 1], p1.w[j >> 1], p2.w[j >> 1]);
    };
    auto route_one = [&](int ks, int sbuf, int j, Q& p0, Q& p1, Q& p2) {
        RouteIn in;
        route_load(sbuf, j, in);
        route_calc(ks, j, in, p0, p1, p2);
    };
    auto route_store = [&](int ibuf, const Q& p0, const Q& p1, const Q& p2) {
        char* const I = lds + ibuf * L::IMG;
        *(uint2*)(I + rec) = p0.u;
        *(uint2*)(I + L::PLANE + rec) = p1.u;
        *(uint2*)(I + 2 * L::PLANE + rec) = p2.u;
    };

    f32x4 acc[7][NPT];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Weight tile (K step iks_, tap tapi of this wave) -> ring slot.  ALWAYS issued — past the pass's last tile the callers name a tile of the
    // last K step again (it lands in a consumed slot and is never read): with no "is there a tile left" test the K loop below has no branch
    // between its MFMA groups and every counted wait is one immediate.
    auto tile_issue = [&](int iks_, int tapi, int slot) {
        // source = a block-uniform 64-bit base (the sample's image) + ONE 32-bit per-lane offset (tile offset + lane part; a sample's image is
        // KS * 25 * 6 KiB): hipcc then uses the SGPR-base addressing form instead of a 64-bit vector add per piece
        const unsigned off = (unsigned)((iks_ * 25 + tap0 + tapi) * (32 * 192)) + a_lane;
        char* const dst = ring + slot * L::SLOT;
        const auto gsrc = (const __attribute__((address_space(1))) void*)(Awave + off);
        const auto ldst = (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)dst;
#ifdef RBNN_DENSE_ABL_PARTA
        if (lane < RBNN_DENSE_ABL_PARTA)                                   // ablation (timing only): a fraction of every weight tile is fetched (same instruction and wait counts)
#endif
        {
            __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 2048, 0);
        }
    };

    if (pass == 0)                                                         // the never-matching stash bytes of both staging buffers (ordered before their first read by the prologue's barrier)
        for (int i = tid; i < 2 * ((L::STG - NFL * 5) / 4); i += 512) {
            constexpr int ND = (L::STG - NFL * 5) / 4;
            *(unsigned*)(lds + L::SOFF + (i / ND) * L::STG + NFL * 5 + 4 * (i % ND)) = 0x08080808u;
        }
    // K steps 0 and 1 are staged together (one HBM round trip, not two) — by the first pass here, for a later pass by the pass before it, ahead
    // of its col2im (the rows staged do not depend on the pass)
    if (pass == 0) {
        stage_issue(0, 0, std::false_type{});
        if (KS > 1) stage_issue(1, 1, std::false_type{});
    }
    tile_issue(0, 0, 0); tile_issue(0, 1, 1); tile_issue(0, 2, 2);         // (a wave has >= 6 taps)
    if (pass == 0) set_scales();
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0)
    __syncthreads();
    {
        Q p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) route_one(0, 0, j, p0, p1, p2);
        route_store(0, p0, p1, p2);
    }
    f16x8 a0 = *(const f16x8*)(ring + foff), a1 = *(const f16x8*)(ring + 1024 + foff), a2 = *(const f16x8*)(ring + 2048 + foff);
#ifdef RBNN_DENSE_ABL_NOB
    f16x8 b0[NPT], b1[NPT], b2[NPT];
#endif
    // The K loop, one instantiation per tap count NT of the wave (6 or 7: wave-uniform, chosen once) — so that which tile is issued at tap t
    // (tile g + 3 = tap (t + 3) % NT of step ks + (t + 3) / NT), its ring slot and every wait count are compile-time facts and a K step is
    // straight-line code.  (Round 3 kept running (K step, tap, index) counters with an "any tile left" test, a 7th-tap test and a choice of wait
    // per tap: four scalar branches and ~20 scalar instructions between two taps' MFMA groups; adding three more branches per tap — one counted
    // wait per weight plane — cost 6.6 % of the kernel, which is what pointed here.)
    // The staging pieces of K step ks + 2 (dQ2 / stash rows: first touch, they come from HBM) are issued BEHIND tap 0's tile, and the first three
    // taps' waits leave them outstanding: vmcnt counts in issue order, so a wait for a tile issued after them is a wait for them (round 3 issued
    // them first and waited vmcnt(6) at tap 0: an HBM round trip per K step and wave).  They too are always issued (the last two steps re-stage
    // the last step's rows into the free buffer) so that the count behind a tile is the same in every step; SMIN = pieces EVERY wave issues
    // for a whole step (assuming fewer than were issued only waits longer) — 0 when the last step is half a step (Hc % 32 == 16).
    auto kloop = [&](auto NTC, auto WHOLEC, auto R0C) {
    constexpr int NT = decltype(NTC)::value, SB = decltype(WHOLEC)::value ? SMIN : 0, R0 = decltype(R0C)::value;
    for (int ks = 0; ks < KS; ++ks) {
        // this wave's staging DMA of K step ks + 1 (issued a whole step ago, or in the prologue) has landed: at most the 9 youngest
        // vector-memory operations — the ring tiles issued since — may still be in flight
        // (a raw s_barrier behind the counted wait: __syncthreads() makes hipcc drain vmcnt to 0 first — the three weight tiles in flight
        // for the coming taps included)
#ifdef RBNN_DENSE_ABL_NOBAR
        if (ks == 0)
#endif
        {
#if defined(RBNN_DENSE_STAMPS) && RBNN_DENSE_STAMPS >= 2
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
#endif
            if (ks == 0) ring_wait_barrier<0>();                           // (nothing has been issued behind the prologue's staging piece yet)
            else ring_wait_barrier<9>();                                   // image ks complete; staging ks + 1 complete; image / staging ks - 1 free
#if defined(RBNN_DENSE_STAMPS) && RBNN_DENSE_STAMPS >= 2
            barw += __builtin_amdgcn_s_memtime() - tb_;
#endif
        }
        const int ksn = min(ks + 1, KS - 1), gk = ks * NT;                 // gk = running index of the step's first tile (slot = index & 3)
        auto issue = [&](auto TC) {                                        // tap t's tile: tile g + 3 -> the slot of tile g - 1 (consumed)
            constexpr int t = decltype(TC)::value;
#ifdef RBNN_DENSE_ABL_NOA
            if (false)                                                     // ablation (timing only): no weight-tile traffic after the prologue
#endif
            tile_issue((t + 3) / NT ? ksn : ks, (t + 3) % NT, (gk + t + 3) & (L::RING - 1));
        };
        issue(std::integral_constant<int, 0>{});                           // tap 0's tile, then the staging pieces — both before the B fragments are
        stage_issue(min(ks + 2, KS - 1), ks & 1, WHOLEC);                  // live: the pieces' per-lane addresses need registers of their own
        const char* const I = lds + (ks & 1) * L::IMG + foff;
#ifdef RBNN_DENSE_ABL_NOB
        if (ks == 0)                                                       // ablation (timing only): the B fragments of the first K step serve all
#else
        f16x8 b0[NPT], b1[NPT], b2[NPT];
#endif
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            b0[pt] = *(const f16x8*)(I + pt * 1024);
            b1[pt] = *(const f16x8*)(I + L::PLANE + pt * 1024);
            b2[pt] = *(const f16x8*)(I + 2 * L::PLANE + pt * 1024);
        }
        Q p0, p1, p2;
        RouteIn rin;
        static_for<0, NT>([&](auto TC) {
            constexpr int t = decltype(TC)::value;
            if constexpr (t > 0) issue(TC);
            // tile g + 1 has landed once all but the 6 youngest operations (tiles g + 2, g + 3) are done — plus, in a step's first three
            // taps, the staging pieces issued behind tile g + 3 of tap 0
            asm volatile("" ::: "memory");
#ifndef RBNN_DENSE_ABL_NOA
            __builtin_amdgcn_s_waitcnt(VMCNT(6 + (t < 3 ? SB : 0)));
#endif
            asm volatile("" ::: "memory");
            const char* const nx = ring + ((gk + t + 1) & (L::RING - 1)) * L::SLOT + foff;
            // The six product groups, ordered by the piece of A they read — a2 | a1 a1 | a0 a0 a0 — so that each piece of the NEXT tile
            // is loaded IN PLACE right behind the last MFMA that reads the current one (an MFMA reads its operands when it issues):
            // no copies, and every reload has >= 12 MFMAs before its first use.  (a2*b0 and a1*b1 are the 2^-22 terms, a1*b0 and
            // a0*b1 the 2^-11 ones: apart from a0*b2, still small terms first.)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a2, b0[pt], acc[t][pt]);
#ifndef RBNN_DENSE_ABL_NOAREAD
            a2 = *(const f16x8*)(nx + 2048);
#endif
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a1, b1[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a1, b0[pt], acc[t][pt]);
#ifndef RBNN_DENSE_ABL_NOAREAD
            a1 = *(const f16x8*)(nx + 1024);
#endif
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a0, b2[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a0, b1[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(a0, b0[pt], acc[t][pt]);
#ifndef RBNN_DENSE_ABL_NOAREAD
            a0 = *(const f16x8*)nx;
#endif
#ifndef RBNN_DENSE_ABL_NOROUTE
            // the next K step's image: channel j of the thread's quad is read in tap group j and routed in tap group j + 1
            // (R0: the channel-tile-1 wave of a SIMD routes RBNN_DENSE_STAGGER tap groups later than its channel-tile-0 partner)
            if constexpr (t >= R0 + 1 && t < R0 + 5) route_calc(ks + 1, t - R0 - 1, rin, p0, p1, p2);
            if constexpr (t >= R0 && t < R0 + 4) route_load((ks + 1) & 1, t - R0, rin);
            if constexpr (t == R0 + 4) route_store((ks + 1) & 1, p0, p1, p2);
#endif
            __builtin_amdgcn_sched_barrier(0);                             // a tap is one scheduling region (the whole step as one region: routing reads hoisted across taps, 256 registers and scratch)
        });
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): the tiles / staging pieces issued past the end have landed (the col2im tile aliases their slots)
    };
    DSTAMP(8 * pass + 0);                                                  // prologue (pass 0: from the block's start; later passes: from the end of the previous col2im)
    {
        const bool whole = (a.Hc & 31) == 0;                               // block-uniform
        using Z = std::integral_constant<int, 0>;
        using R = std::integral_constant<int, RBNN_DENSE_STAGGER>;
        if (!whole) { if (ntap == 7) kloop(std::integral_constant<int, 7>{}, std::false_type{}, Z{}); else kloop(std::integral_constant<int, 6>{}, std::false_type{}, Z{}); }
        else if (RBNN_DENSE_STAGGER && ct) { if (ntap == 7) kloop(std::integral_constant<int, 7>{}, std::true_type{}, R{}); else kloop(std::integral_constant<int, 6>{}, std::true_type{}, R{}); }
        else { if (ntap == 7) kloop(std::integral_constant<int, 7>{}, std::true_type{}, Z{}); else kloop(std::integral_constant<int, 6>{}, std::true_type{}, Z{}); }
    }
    if (pass + 1 < NPASS) {                                                // the next pass's first two K steps: they land under this pass's col2im (whose barriers do not wait for them)
        stage_issue(0, 0, std::false_type{});
        if (KS > 1) stage_issue(1, 1, std::false_type{});
    }
    DSTAMP(8 * pass + 1);                                                  // K loop
    // ---- col2im: two rounds (input-channel tiles).  T of a round sits in LDS as [25 taps][64 pos][16 ci] floats: an accumulator tile's four
    // registers are four consecutive channels (one ds_write_b128 per tile; the four channel quads of a position are XOR-swizzled by
    // (pos >> 1) & 3, which spreads eight consecutive positions over all 32 banks), and a gathering thread = one output position (Y, X) x two
    // channel quads adds up its <= 25 terms in (ky, kx) order with eight independent sums.  (First version: [tap][ci][pos] floats, 112
    // 4-way-conflicting ds_write_b32 per lane and one serial chain of ~60 dependent LDS reads per thread: 2.0 of the kernel's 11.0 ms,
    // profiles/r03a/conv_dense_ablations.txt.)  With several passes a term belongs to the pass that holds its position; the sums of the
    // earlier passes wait in `part` and the LAST pass adds them (fixed order: deterministic) and writes. ----
    float* const T = lds_f;
#ifdef RBNN_DENSE_ABL_NOEPI
    {                                                                      // ablation (timing only): the accumulators stay live, nothing is gathered
        float sink = 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) sink += acc[t][pt][0] + acc[t][pt][1] + acc[t][pt][2] + acc[t][pt][3];
        if (sink == 1.2345e-30f) a.dP1[sn * G::P1SZ] = sink;
        return;
    }
#endif
#if RBNN_DENSE_COL2IM_RMW
    // (round 4, second half) col2im through WAVE-PRIVATE partial images instead of the T tile above: every wave adds the T tiles of its own
    // taps into its own [P1W x P1W][16 ci] fp32 image in LDS (read - add - write of one ds_*_b128 per accumulator tile; within one tap the
    // positions of a tile land on distinct outputs, taps follow each other in program order, nobody else touches the image: no atomics, a
    // fixed order), then one output position x two channel quads per thread adds the four images of a channel tile in wave order.  Both channel
    // tiles at once: two barriers per pass instead of four, 0.55 MB through LDS per pass instead of 0.86 (the T tile was written by half the
    // waves and gathered with 2/3 of the reads masked off: 16.7k of a block's 135k cycles per pass, tools/dense_stamps.sh).  The channel quads of
    // an output position are XOR-swizzled by (position >> 2) & 3: the 16 lanes of a ds_*_b128 service group cover positions p .. p + 3 and
    // p + 12 .. p + 15 of a tile, whose 64-byte records would otherwise share banks four positions apart.
    {
        char* const img = lds + wave * L::EIMG;
        ring_wait_barrier<63>();                                           // the loop buffers are free: every wave is out of its K loop with its ring DMA drained (raw barriers here:
                                                                           // __syncthreads() would wait for the next pass's staging pieces just issued)
        for (int i = lane; i < L::EIMG / 16; i += 64) *(f32x4*)(img + 16 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
        int o0[NPT];
        bool val[NPT];
        const int dummy = (L::EPI + 64 + lane * 16) - wave * L::EIMG + wave * 1024;   // (relative to img) behind the images and the wave maxima: 1 KiB per wave
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int gpos = 64 * pass + 16 * pt + li;                     // acc[t][pt][r] = T[tap0 + t][ci = 4lg + r][gpos]
            val[pt] = gpos < NPOS_;
            o0[pt] = (gpos / O2W_) * P1W_ + gpos % O2W_;                   // output position of tap (0, 0)
        }
#pragma unroll
        for (int t = 0; t < 7; ++t)
            if (t < ntap) {                                                // wave-uniform
                const int tap = tap0 + t, shift = (tap / 5) * P1W_ + tap % 5;
                f32x4 cur[NPT];
                int ad[NPT];
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {                         // the tile's reads together, then its writes
                    const int o = o0[pt] + shift;
                    // lanes past the last position (the last pass's last tile) go through a private dummy record instead of an exec-masked
                    // block per access (a masked LDS read is waited for inside its block: one round trip each, in series)
                    ad[pt] = val[pt] ? o * 64 + ((lg ^ ((o >> 2) & 3)) << 4) : dummy;
                    cur[pt] = *(const f32x4*)(img + ad[pt]);
                }
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) *(f32x4*)(img + ad[pt]) = cur[pt] + acc[t][pt];
            }
        ring_wait_barrier<63>();
        float omax = 0.f;
        if (tid < 2 * NPP) {
            const int qp = tid / NPP, pp = tid % NPP, sw = (pp >> 2) & 3;
            const char* const rec = lds + pp * 64;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {                               // channel tile c2 = waves 4 c2 .. 4 c2 + 3
                f32x4 u0[4], u1[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u0[q4] = *(const f32x4*)(rec + (4 * c2 + q4) * L::EIMG + (((2 * qp) ^ sw) << 4));
                    u1[q4] = *(const f32x4*)(rec + (4 * c2 + q4) * L::EIMG + (((2 * qp + 1) ^ sw) << 4));
                }
                const f32x4 s0 = part[c2][0] + (((u0[0] + u0[1]) + u0[2]) + u0[3]);   // earlier passes + this pass's four tap groups, in that order
                const f32x4 s1 = part[c2][1] + (((u1[0] + u1[1]) + u1[2]) + u1[3]);
                if (pass + 1 < NPASS) {
                    part[c2][0] = s0;
                    part[c2][1] = s1;
                } else {
                    float* const dst0 = a.dP1 + sn * G::P1SZ + (16 * c2 + 8 * qp) * NPP + pp;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float* const dst = dst0 + r * NPP;
                        const float v = (r < 4 ? s0[r & 3] : s1[r & 3]) * out_scale;   // over the forward's P1 (dead after this read): sigmoid / tanh take act' from it
                        const float o = smooth_act<ACT>() ? v * act_grad_from_value<ACT>(*dst) : v;
                        *dst = o;
                        omax = fmaxf(omax, fabsf(o));
                    }
                }
            }
        }
#if RBNN_CONV1_BWD_X3
        if (G::CIN >= RBNN_CONV1_BWD_X3_MINCIN && pass + 1 == NPASS) {                                           // max |dP1| of this (sample, point) -> G[sn][0]: the scale of conv1_bwd_x3_kernel (which reads it before it writes G)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
            float* const wm = (float*)(lds + L::EPI);                      // (the rings' area: free)
            if (lane == 0) wm[wave] = omax;
            ring_wait_barrier<63>();
            if (tid == 0) {
                float m = wm[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) m = fmaxf(m, wm[w]);
                a.G[sn * G::DIN] = m;
            }
        }
#endif
    }
#else
    static_for<0, 2>([&](auto ROUND) {
        constexpr int round = decltype(ROUND)::value;
        __syncthreads();                                                   // the loop buffers / the previous round's T are free
        if (ct == round) {
#pragma unroll
            for (int t = 0; t < 7; ++t)
                if (t < ntap) {
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        const int pos = 16 * pt + li;                      // acc[t][pt][r] = T[tap0 + t][ci = 4lg + r][pos], pos local to the pass
                        *(f32x4*)(T + ((tap0 + t) * 64 + pos) * 16 + 4 * (lg ^ ((pos >> 1) & 3))) = acc[t][pt];
                    }
                }
        }
        __syncthreads();
        if (tid < 2 * NPP) {
            const int qp = tid / NPP, pp = tid % NPP, Y = pp / P1W_, X = pp % P1W_;
            const int ky0 = max(0, Y - (O2W_ - 1)), ky1 = min(4, Y), kx0 = max(0, X - (O2W_ - 1)), kx1 = min(4, X);
            // the five kx terms of a row are read together (ten independent ds_read_b128; terms outside [kx0, kx1] — or, with several passes,
            // at a position of another pass — read a clamped position and add +0): a wave runs as long as its busiest lane, 25 terms for the
            // interior positions, and one read pair per trip made that 25 dependent LDS round trips
            f32x4 s0 = part[round][0], s1 = part[round][1];
            for (int ky = ky0; ky <= ky1; ++ky) {
                f32x4 u0[5], u1[5];
                bool ok[5];
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    int pos = (Y - ky) * O2W_ + min(max(X - kx, 0), O2W_ - 1) - 64 * pass;
                    ok[kx] = kx >= kx0 && kx <= kx1;
                    if (NPASS > 1) {
                        ok[kx] = ok[kx] && pos >= 0 && pos < 64;
                        pos = min(max(pos, 0), 63);
                    }
                    const int sw = (pos >> 1) & 3;
                    const float* const rowp = T + ((ky * 5 + kx) * 64 + pos) * 16;
                    u0[kx] = *(const f32x4*)(rowp + 4 * ((2 * qp) ^ sw));
                    u1[kx] = *(const f32x4*)(rowp + 4 * ((2 * qp + 1) ^ sw));
                }
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    s0 += ok[kx] ? u0[kx] : z;
                    s1 += ok[kx] ? u1[kx] : z;
                }
            }
            if (pass + 1 < NPASS) {
                part[round][0] = s0;
                part[round][1] = s1;
            } else {
                float* const dst0 = a.dP1 + sn * G::P1SZ + (16 * round + 8 * qp) * NPP + pp;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float* const dst = dst0 + r * NPP;
                    const float v = (r < 4 ? s0[r & 3] : s1[r & 3]) * out_scale;   // over the forward's P1 (dead after this read): sigmoid / tanh take act' from it
                    *dst = smooth_act<ACT>() ? v * act_grad_from_value<ACT>(*dst) : v;
                }
            }
        }
    });
#endif
    DSTAMP(8 * pass + 2);                                                  // col2im
    });
#ifdef RBNN_DENSE_STAMPS
    DSTAMP_ADD(24, __builtin_amdgcn_s_memtime() - tstart);
    DSTAMP_ADD(25, 1);
    DSTAMP_ADD(26, barw);
#endif
}

// =====================================================================================================
// Both triple images of model.3.weight — the forward's tap-major grouped rows image and the dense conv2^T image — from the fp32 stack in ONE
// launch.  A redrawable SVI stack rebuilds them after every draw (BASELINE config 5: every PGD iteration); through their stand-alone builders
// that was two permuted fp32 copies, two rbnn_triple_rows launches and two more permuted copies per draw (0.45 of the 0.6 ms a C5 draw took).
// One block = one (sample, 32 output channels): its [32 hc][32 ci x 25 taps] cube is 100 KB of CONTIGUOUS fp32 — read once, coalesced, into
// LDS — and both images leave as 1-KiB runs of 16-byte stores.  Same split3 of the same scaled values: the images are bit-identical to the
// stand-alone builders' (tests/test_hip_round4.py).
// =====================================================================================================
namespace {
constexpr int K2IMG_PITCH = 801;                                          // floats per hc row of the cube in LDS (odd: reads along hc spread over the banks)
__global__ void __launch_bounds__(256) conv_k2_images_kernel(const float* __restrict__ K2w, int Hc, float scale, uint4* __restrict__ rows_img,
                                                             uint4* __restrict__ dense_img) {
    extern __shared__ __attribute__((aligned(16))) float cube[];          // [32 hc][K2IMG_PITCH], k = ci * 25 + tap (nn.Conv2d's order)
    const int tid = threadIdx.x, KS = (Hc + 31) / 32;
    const int s = blockIdx.x / KS, ks = blockIdx.x - s * KS;
    const float* const src = K2w + ((long long)s * Hc + 32 * ks) * 800;
    for (int i = tid; i < 32 * 200; i += 256) {                           // 16-byte loads: row i / 200, columns 4 (i % 200) ..
        const int hcl = i / 200, c4 = 4 * (i - hcl * 200);
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (32 * ks + hcl < Hc) v = *(const f32x4*)(src + hcl * 800 + c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) cube[hcl * K2IMG_PITCH + c4 + r] = v[r] * scale;
    }
    __syncthreads();
    union U { f16x8 v; uint4 u; };
    if (rows_img) {
        // forward image: rows = output channels, K = tap * 32 + ci, grouped [16-channel group][tap][3 pieces][16 channels][32 ci] halves;
        // item = (tap, channel, unit of 8 ci): 64 consecutive items = one 1-KiB (group, tap, piece) run
        for (int it = tid; it < 25 * 32 * 4; it += 256) {
            const int u = it & 3, hcl = (it >> 2) & 31, tap = it >> 7;
            const int hc = 32 * ks + hcl;
            if (hc < Hc) {
                U o[3];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    _Float16 p0, p1, p2;
                    conv_split3(cube[hcl * K2IMG_PITCH + (8 * u + j) * 25 + tap], p0, p1, p2);
                    o[0].v[j] = p0; o[1].v[j] = p1; o[2].v[j] = p2;
                }
                const long long G = ((long long)s * Hc + hc) >> 4;
                uint4* const out = rows_img + ((G * 25 + tap) * 3) * 64 + (hc & 15) * 4 + u;
                out[0] = o[0].u; out[64] = o[1].u; out[128] = o[2].u;
            }
        }
    }
    if (dense_img) {
        // dense image: [sample][K step = these 32 hc][tap][input-channel half][3 pieces][16 ci][32 hc] halves; item = (tap, ci, unit of 8 hc)
        for (int it = tid; it < 25 * 32 * 4; it += 256) {
            const int u = it & 3, ci = (it >> 2) & 31, tap = it >> 7;
            U o[3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 p0, p1, p2;
                conv_split3(cube[(8 * u + j) * K2IMG_PITCH + ci * 25 + tap], p0, p1, p2);
                o[0].v[j] = p0; o[1].v[j] = p1; o[2].v[j] = p2;
            }
            const long long T = (((long long)s * KS + ks) * 25 + tap) * 2 + (ci >> 4);
            uint4* const out = dense_img + (T * 3) * 64 + (ci & 15) * 4 + u;
            out[0] = o[0].u; out[64] = o[1].u; out[128] = o[2].u;
        }
    }
}
}  // namespace

extern "C" int rbnn_conv_weight_images(const float* K2w, int32_t n_samples, int32_t hidden, int32_t k2_exp, void* K2_rows, void* K2_dense, void* stream) {
    if (!K2w || (!K2_rows && !K2_dense)) return RBNN_ERR_NULL;
    if (n_samples < 1 || hidden < 16 || (hidden & 15) || k2_exp < -100 || k2_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(K2w) || (K2_rows && !aligned16(K2_rows)) || (K2_dense && !aligned16(K2_dense))) return RBNN_ERR_ALIGN;
    const int KS = (hidden + 31) / 32;
    constexpr int LDSB = 32 * K2IMG_PITCH * 4;
    static unsigned long long attr = 0;
    if (!ensure_dynamic_lds((const void*)conv_k2_images_kernel, LDSB, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_k2_images_kernel, dim3((unsigned)((long long)n_samples * KS)), dim3(256), LDSB, (hipStream_t)stream,
                       K2w, hidden, ldexpf(1.f, k2_exp), (uint4*)K2_rows, (uint4*)K2_dense);
    return launch_status();
}

extern "C" int rbnn_conv_input_grad_dense(const rbnn_conv_posterior* net, const void* K2_dense, int32_t k2_exp, float fw_l1,
                                          const int32_t* sidx, int32_t S, int32_t N, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!K2_dense || !ws || !ws->dZ || !ws->P1 || !ws->Q2 || !ws->st1 || !ws->st2 || !ws->G) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || k2_exp < -100 || k2_exp > 100 || !(fw_l1 >= 0.f)) return RBNN_ERR_SHAPE;
    if (!aligned16(K2_dense) || !aligned16(ws->G)) return RBNN_ERR_ALIGN;
    ConvBwdArgs a = {};
    a.dZ = ws->dZ; a.st1 = ws->st1; a.st2 = ws->st2; a.K1w = net->K1w; a.K2cb = nullptr; a.Fw = net->Fw;
    a.Hc = net->hidden; a.C = net->n_classes; a.N = N; a.S = S; a.sidx = sidx; a.dQ2 = ws->Q2; a.dP1 = ws->P1; a.G = ws->G;
    hipStream_t st = (hipStream_t)stream;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        a.NP2 = G::NP2;
        return for_activation(net->activation, [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;
            int rc2;
            {
                const long long F = (long long)a.Hc * G::NP2, items = (long long)S * ((N + 15) / 16) * ((F + 63) / 64);
                if (items >= (1LL << 31)) return (int)RBNN_ERR_SHAPE;                     // conv_fc_bwd_kernel decodes its item number in 32 bits
                hipLaunchKernelGGL((conv_fc_bwd_kernel<smooth_act<ACT>(), ACT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
                if ((rc2 = launch_status())) return rc2;
            }
            constexpr int LDSB = ConvBwdDenseLds<G>::BYTES;
            static unsigned long long attr = 0;
            if (!ensure_dynamic_lds((const void*)conv_bwd_dense_x3_kernel<ACT, G>, LDSB, attr)) return (int)RBNN_ERR_LAUNCH;
            hipLaunchKernelGGL((conv_bwd_dense_x3_kernel<ACT, G>), dim3(grid_for_items((long long)N * S)), dim3(512), LDSB, st, a, (const char*)K2_dense, k2_exp, fw_l1);
            if ((rc2 = launch_status())) return rc2;
            return launch_conv1_backward<ACT, G>(a, st, true);
        });
    });
}

#ifdef RBNN_DENSE_STAMPS
// diagnostic builds only: copy (and optionally clear) the stamp sums; out = 64 x u64
extern "C" __attribute__((visibility("default"))) int rbnn_debug_dense_stamps(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return RBNN_ERR_LAUNCH;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rbnn_dense_stamp_acc), 64 * sizeof(unsigned long long)) != hipSuccess) return RBNN_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[64] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rbnn_dense_stamp_acc), z, sizeof z) != hipSuccess) return RBNN_ERR_LAUNCH;
    }
    return RBNN_OK;
}
#endif

extern "C" int rbnn_conv_input_grad_triple(const rbnn_conv_posterior* net, const void* K2_bwd, int32_t k2_exp, float fw_l1,
                                           const int32_t* sidx, int32_t S, int32_t N, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!K2_bwd || !ws || !ws->dZ || !ws->P1 || !ws->Q2 || !ws->st1 || !ws->st2 || !ws->G) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || k2_exp < -100 || k2_exp > 100 || !(fw_l1 >= 0.f)) return RBNN_ERR_SHAPE;
    if (!aligned16(K2_bwd) || !aligned16(ws->G)) return RBNN_ERR_ALIGN;
    ConvBwdArgs a = {};
    a.dZ = ws->dZ; a.st1 = ws->st1; a.st2 = ws->st2; a.K1w = net->K1w; a.K2cb = nullptr; a.Fw = net->Fw;
    a.Hc = net->hidden; a.C = net->n_classes; a.N = N; a.S = S; a.sidx = sidx; a.dQ2 = ws->Q2; a.dP1 = ws->P1; a.G = ws->G;
    hipStream_t st = (hipStream_t)stream;
    return for_geometry(net, [&](auto g) {
        using G = decltype(g);
        a.NP2 = G::NP2;
        return for_activation(net->activation, [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;
            using L = ConvBwdX3Lds<G, smooth_act<ACT>()>;
            int rc2;
            {
                const long long F = (long long)a.Hc * G::NP2, items = (long long)S * ((N + 15) / 16) * ((F + 63) / 64);
                if (items >= (1LL << 31)) return (int)RBNN_ERR_SHAPE;                     // conv_fc_bwd_kernel decodes its item number in 32 bits
                hipLaunchKernelGGL((conv_fc_bwd_kernel<smooth_act<ACT>(), ACT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
                if ((rc2 = launch_status())) return rc2;
            }
            constexpr int LDSB = L::NWB * L::WAVE;
            static unsigned long long attr = 0;
            if (!ensure_dynamic_lds((const void*)conv_bwd_x3_kernel<ACT, G>, LDSB, attr)) return (int)RBNN_ERR_LAUNCH;
            const int grid = grid_for_items((long long)((N + L::NWB - 1) / L::NWB) * S);
            hipLaunchKernelGGL((conv_bwd_x3_kernel<ACT, G>), dim3(grid), dim3(64 * L::NWB), LDSB, st, a, (const char*)K2_bwd, k2_exp, fw_l1);
            if ((rc2 = launch_status())) return rc2;
            return launch_conv1_backward<ACT, G>(a, st);
        });
    });
}
