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
#include "rbnn_conv_common.hpp"

#ifndef RBNN_FCBWD_NT
#define RBNN_FCBWD_NT 1                                                    // conv_fc_bwd writes dQ2 (5 GB per C5 pass, re-read from HBM by the dense kernel in any case) with non-temporal stores:
#endif                                                                     // they no longer push the sample's Fw out of the L2 — 1.54 -> 1.25 ms (same box, alternating: backward call 16.3 -> 16.0 ms)

using namespace rbnn_conv_shared;

namespace {

// ---------------------------------------------------------------------------------------------------
// One block = one (sample, point): the sample's 32 x Cin x 25 weights and the point's image go to LDS once, then one thread per
// pooled POSITION keeps its Cin x 6 x 6 input patch in registers and runs all 32 output channels over it (weights are LDS
// broadcasts: ~16 FMAs per LDS read).  The first version (one thread per pooled output, patch and weights re-read from memory
// by every thread: 1.6 FMAs per load) took 7.9 ms per pass at the CIFAR-shaped c5 bench.  Same accumulation order
// (input channel, ky, kx), so the results are bit-identical to it.
// Two threads per pooled position (16 channels each): 144 / 196 positions alone left 44 % / 23 % of a 256-thread block's lanes idle in
// the FMA loop (0.52 -> see profiles/r03a/conv_small_kernels.txt).  Each output is still produced by one thread in the same order.
// (3x32x32: 196 positions fill a 256-thread block well enough, and the two-halves form measured slower there: 2.45 -> 2.66 ms at c5)
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
                            // The BROADCAST operand (the patch pair, one half of it for both result lanes) is src0: op_sel / op_sel_hi bit 0.  Rounds 3-5 had it as
                            // src1 (op_sel:[0,1,0]) — a form hipcc itself never emits (it canonicalises a packed-fp32 broadcast onto src0) and that is NOT reliable
                            // on gfx950 when waves of another kernel share the SIMD: with two processes on one GPU the low result lane intermittently took the
                            // other half of the pair (46 of 320 forward calls differed from their own reference; 0 of 320 in this form; profiles/r05w).  The
                            // product is commutative: results are bit-identical to the old form's (one process).
                            if ((dx + kx) & 1) asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(v4[dy * 2 + dx]) : "v"(wv), "v"(pp));
                            else               asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(v4[dy * 2 + dx]) : "v"(wv), "v"(pp));
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

// pool-1 routing + conv1^T on the matrix pipe (the dispatched version; conv1_bwd_kernel above is its VALU gather form:
// 3.32 ms at the conv-512 bench against 1.24 ms for this one; 15.4 -> see profiles at the CIFAR-shaped c5 bench).
// One WAVE = one (sample, point), no barriers.  Per input channel ci:
//   dX[ci][Y][X] = sum_{c, ky, kx} w[c][ci][ky][kx] * R[c][Y - ky][X - kx],   R = the pooled gradient routed to its argmax position
//   (O1 x O1, one non-zero per 2x2 cell) times the activation derivative.
// Per position row Ya the wave forms  T[tap][Xa] = sum_c w[c][ci][tap] * R[c][Ya][Xa]  — a 32(taps, 25 used) x 32(c) x 32(Xa, O1 used)
// fp32 MFMA product whose B operand is built in registers from the stash bytes and pooled gradients of pooled row Ya/2 — keeps the
// last five rows of T in an LDS ring, and emits output row Y = Ya as a 25-term gather  dX[Y][X] = sum_tap T[tap][Y - ky][X - kx].
// With Cin > 1 the pass is repeated per input channel (the routed B operand is rebuilt from L2-resident data; the ring stays 62.5 KiB).
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

// pool-1 routing + conv1^T on the fp32 matrix pipe (the f16-pipe form, conv1_bwd_x3_kernel, runs behind the dense conv2^T kernel: rbnn_conv_x3.hip)
template <int ACT, class G>
int launch_conv1_backward(const ConvBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((conv1_bwd_mfma_kernel<ACT, G>), dim3(grid_for_items((long long)((a.N + 3) / 4) * a.S)), dim3(256), 0, st, a);
    return launch_status();
}

template <int ACT>
int launch_conv_fc_bwd_t(const ConvBwdArgs& a, hipStream_t st) {
    const long long F = (long long)a.Hc * a.NP2, items = (long long)a.S * ((a.N + 15) / 16) * ((F + 63) / 64);
    if (items >= (1LL << 31)) return (int)RBNN_ERR_SHAPE;                     // conv_fc_bwd_kernel decodes its item number in 32 bits
    hipLaunchKernelGGL((conv_fc_bwd_kernel<smooth_act<ACT>(), ACT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
    return launch_status();
}

}  // namespace

namespace {
template <int ACT, class G>
int launch_conv_backward(const ConvBwdArgs& a, hipStream_t st) {
    int rc = launch_conv_fc_bwd_t<ACT>(a, st);
    if (rc) return rc;
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
    if ((rc = launch_conv_fc_bwd_t<RBNN_ACT_LEAKY>(a, st))) return rc;          // (relu / leaky: the same instantiation — act' is not folded in)
    if (leaky) hipLaunchKernelGGL(conv_bwd_split_kernel<RBNN_ACT_LEAKY>, dim3(grid), dim3(256), 0, st, a, (const char*)K2_bwd, k2_exp, fw_l1);
    else       hipLaunchKernelGGL(conv_bwd_split_kernel<RBNN_ACT_RELU>, dim3(grid), dim3(256), 0, st, a, (const char*)K2_bwd, k2_exp, fw_l1);
    if ((rc = launch_status())) return rc;
    if (leaky) return launch_conv1_backward<RBNN_ACT_LEAKY, GeoMnist>(a, st);
    return launch_conv1_backward<RBNN_ACT_RELU, GeoMnist>(a, st);
}

// ---- the launchers rbnn_conv_x3.hip shares (rbnn_conv_common.hpp)
namespace rbnn_conv_shared {
int launch_conv1_pool(int act, int in_channels, const ConvArgs& a, hipStream_t st) {
    auto go = [&](auto g) {
        using G = decltype(g);
        return for_activation(act, [&](auto actc) {
            hipLaunchKernelGGL((conv1_pool_kernel<decltype(actc)::value, G>), dim3((unsigned)((long long)a.S * a.N)), dim3(conv1_threads<G>()), 0, st, a);
            return launch_status();
        });
    };
    return in_channels == 1 ? go(GeoMnist{}) : go(GeoCifar{});
}
int launch_conv_fc(const ConvArgs& a, hipStream_t st) {
    const int items = ((a.N + 15) / 16) * a.S;
    hipLaunchKernelGGL(conv_fc_kernel, dim3((items + 3) / 4), dim3(256), 0, st, a);
    return launch_status();
}
int launch_conv_fc_bwd(int act, const ConvBwdArgs& a, hipStream_t st) {
    return for_activation(act, [&](auto actc) { return launch_conv_fc_bwd_t<decltype(actc)::value>(a, st); });
}
int launch_conv1_bwd_fp32(int act, int in_channels, const ConvBwdArgs& a, hipStream_t st) {
    auto go = [&](auto g) {
        return for_activation(act, [&](auto actc) { return launch_conv1_backward<decltype(actc)::value, decltype(g)>(a, st); });
    };
    return in_channels == 1 ? go(GeoMnist{}) : go(GeoCifar{});
}
}  // namespace rbnn_conv_shared
