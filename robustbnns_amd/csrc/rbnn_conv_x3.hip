// rbnn_conv_x3.hip — the conv architecture (model_nn.py:93-106) in the TRIPLE-SPLIT mode (what precision="auto" runs on it, both geometries):
// conv2 forward and conv2^T / conv1^T backward with every fp32 operand carried at full width as three fp16 pieces, six exact product terms per
// fp32 product on v_mfma_f32_16x16x32_f16, fp32 accumulation (rbnn_triple.hip's arithmetic).  conv1 + pool, the Linear head and its transpose
// are the fp32 kernels of rbnn_conv.hip (2 % of the MACs; rbnn_conv_common.hpp declares their launchers).
//   conv2_pool_x3_kernel        forward conv2 + pool + activation + stash
//   conv_bwd_dense_x3_kernel    conv2^T as a GEMM per tap over the conv2 OUTPUT positions + col2im (no padding MFMAs)
//   conv1_bwd_x3_kernel         pool-1 routing + conv1^T (more than one input channel)
//   conv_k2_images_kernel       both weight images of model.3.weight from the fp32 stack in one launch
#include "rbnn_conv_common.hpp"
#include <algorithm>
#include <cstdlib>

#ifndef RBNN_X3FWD_NT
#define RBNN_X3FWD_NT 0                                                    // non-temporal Q2 / stash stores of conv2_pool_x3_kernel: measured equal / slower
#endif
#ifndef RBNN_CONV1_BWD_X3_MINCIN
#define RBNN_CONV1_BWD_X3_MINCIN 2                                        // input channels from which conv1^T runs on the f16 pipe (1x28x28 keeps the fp32 kernel)
#endif

using namespace rbnn_conv_shared;

namespace {

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

#define RBNN_X3FWD_BPREFETCH 0                                           // 1: the B fragments of tap t + 1 read under tap t's MFMAs (round 4) — measured SLOWER: 1x28x28 8.42 -> 8.58 ms per forward call,
                                                                          // 3x32x32 15.1 -> 16.5 (spills beside 56 accumulators): with two waves per SIMD the other wave fills a tap's post-barrier round trip
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
    static constexpr int IPITCH = MNIST ? 12 : 18;
    static constexpr int ROWX = MNIST ? 1 : 0;
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
    // RBNN_X3FWD_PAIR13 (round 5; geometries whose NPOS is not a multiple of 16, i.e. 3x32x32): the two points of a block are tiled TOGETHER —
    // combined position c = 100 * point + position, 200 positions = 13 tiles of 16 instead of 2 x 7 (7.1 % fewer MFMAs: eight idle columns per
    // pair instead of twenty-four).  The wave of the pair's first point takes tiles 0 .. 6 (its last tile: 4 positions of point 0, 12 of point 1,
    // which it writes into its partner's pooling tile), the other wave tiles 7 .. 12; the two waves of a channel group share a SIMD (wave & 3),
    // so every SIMD issues 13 tiles per tap instead of 14.  Same-box: 14.37 -> 13.97 ms per C5 forward call (profiles/r05w).
#ifndef RBNN_X3FWD_PAIR13
#define RBNN_X3FWD_PAIR13 1
#endif
    constexpr bool PAIR = RBNN_X3FWD_PAIR13 && (NPOS_ % 16 != 0);
    static_assert(!PAIR || (2 * NPOS_ + 15) / 16 == 2 * NPT - 1, "13 = 7 + 6 tiles");
    const char* const img = imgs + wp * L::IMGB;
    constexpr int CPITCH = NPOS_ + 4, EIT = (16 * NP2_ / 4 + 63) / 64;
    auto body = [&](auto NTC) {
    constexpr int NT = decltype(NTC)::value;                              // position tiles of this wave
    const int c0 = PAIR && wp ? 16 * NPT : 0;                              // (PAIR) first combined position of the wave
    // image position of output position 16pt + li (tap 0,0); idle lanes of the last tile read distinct valid positions, never stored
    int pbase[NT], ybase[NT];
    int ioff_s = 0;                                                        // (PAIR, first point's wave, last tile) the lane's image relative to `img`: 0 / IMGB
#pragma unroll
    for (int pt = 0; pt < NT; ++pt) {
        int pos = pt * 16 + li;
        if (PAIR) {
            int c = c0 + pos;
            if (c >= 2 * NPOS_) c -= NPOS_;                               // idle lanes of the pair's last tile: valid positions of point 1
            const int point = c >= NPOS_ ? 1 : 0;
            pos = c - point * NPOS_;
            if (pt == NT - 1 && NT == NPT) ioff_s = (point - wp) * L::IMGB;
        } else if (pos >= NPOS_) pos = pos - NPOS_;
        ybase[pt] = pos / O2W_;
        pbase[pt] = ybase[pt] * IPITCH + pos % O2W_;
    }
    int pbase_e[EIT][4];
#pragma unroll
    for (int it = 0; it < EIT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = min(4 * (lane + 64 * it) + j, 16 * NP2_ - 1), hl = idx / NP2_, p = idx % NP2_;
            pbase_e[it][j] = hl * CPITCH + (p / P2W_) * O2W_ + (p % P2W_);
        }
    for (int hc0 = 0; hc0 < a.Hc; hc0 += WROWS) {
        f32x4 acc[HTW][NT];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
        constexpr bool PF_ALL = RBNN_X3FWD_BPREFETCH && NT <= 4;
        auto load_b = [&](int tap, f16x8 (&b0)[NT], f16x8 (&b1)[NT], f16x8 (&b2)[NT], bool lo, bool hi) {
            const int ky = tap / 5, toff = ky * IPITCH + (tap % 5);
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
                const int p = pbase[pt] + toff;
                const char* const src = img + ((PAIR && NT == NPT && pt == NT - 1) ? ioff_s : 0) + p * 64 + ((lg ^ x3_img_swz<G>(p, ybase[pt] + ky)) * 16);
                if (lo) { b0[pt] = *(const f16x8*)src; b1[pt] = *(const f16x8*)(src + L::IMGP); }
                if (hi) b2[pt] = *(const f16x8*)(src + 2 * L::IMGP);
            }
        };
        f16x8 bA0[NT], bA1[NT], bA2[NT], bB0[PF_ALL ? NT : 1], bB1[PF_ALL ? NT : 1], bB2[NT];
        auto tap_body = [&](int tap, f16x8 (&b0)[NT], f16x8 (&b1)[NT], f16x8 (&b2)[NT], auto& n0, auto& n1, f16x8 (&n2)[NT]) {
            const int buf = tap & 1;
            if (tap + 1 < 25) {
                stage(tap + 1, buf ^ 1);
            }
            load_b(tap, b0, b1, b2, true, true);
            const char* const Wt = ldsb + buf * L::TILEW + (wq * HTW) * 3072 + foff;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                const f16x8 a0 = *(const f16x8*)(Wt + ht * 3072), a1 = *(const f16x8*)(Wt + ht * 3072 + 1024),
                            a2 = *(const f16x8*)(Wt + ht * 3072 + 2048);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a0, b2[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a2, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a1, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a1, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a0, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = MFMA_H(a0, b0[pt], acc[ht][pt]);
            }
            ring_wait_barrier<0>();                                      // tap+1's weights landed; everyone is done with this tile
        };
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
                for (int pt = 0; pt < NT; ++pt) sink += acc[ht][pt][0] + acc[ht][pt][1] + acc[ht][pt][2] + acc[ht][pt][3];
            if (sink == 1.2345e-30f) a.Q2[sn * F] = sink;
            __syncthreads();
            continue;
        }
#endif
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht) {
            const int hcb = hc0 + (wq * HTW + ht) * 16;                    // wave-uniform
            // (PAIR: the block meets at two barriers per channel tile — a channel group past Hc skips the work, not the barriers)
            const bool valid = hcb < a.Hc;
            if (!PAIR && !valid) break;
            if (valid) {
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
                // PAIR: combined position c of this lane's column -> the pooling tile of ITS point (wave wq + 4 * point)
                const int c = c0 + pt * 16 + li, point = (PAIR && c >= NPOS_) ? 1 : 0, pos = c - point * NPOS_;
                float* const T = PAIR ? (float*)ldsb + (wq + 4 * point) * 16 * CPITCH : my;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (PAIR ? c < 2 * NPOS_ : pos < NPOS_) {
                        const float pre = acc[ht][pt][r] * out_scale + bias[r];   // sigmoid / tanh are pooled on their VALUES, as torch does
                        T[(4 * lg + r) * CPITCH + pos] = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                    }
            }
            }
            if (PAIR) __syncthreads();                                     // both waves of a channel group have written the point's tile
            // four consecutive pooled cells per lane: one 16-byte store of Q2 and one 4-byte store of the stash (the tile's 16 x NP2
            // cells are contiguous in both)
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int i4 = lane + 64 * it;
                if (i4 < 16 * NP2_ / 4 && live && valid) {
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
            if (PAIR && ht + 1 < HTW) __syncthreads();                     // the tiles are rewritten by the next channel tile (the partner's wave writes into this one's)
        }
        __syncthreads();                                                 // the pooling tiles alias the weight buffers of the next chunk
    }
    };
    if (PAIR && wp) body(std::integral_constant<int, PAIR ? NPT - 1 : NPT>{});
    else body(std::integral_constant<int, NPT>{});
}

// =====================================================================================================
// conv1 + pool + activation + stash on the F16 matrix pipe (triple-split arithmetic; round 6).  conv1_pool_kernel (rbnn_conv.hip) is a packed-FMA VALU
// kernel: 1.66 ms per C5 pass of 512 points x 62 samples at 0.46 of the fp32 VECTOR peak — the last forward kernel of the conv path that was not on
// the matrix pipe (VERDICT r5 missing #3).  Here
//
//     O1[c][y][x] = sum_{ci, ky, kx} w[c][ci][ky][kx] . X[ci][y + ky][x + kx]        M = 32 channels (2 tiles), K = (ci, ky, kx), N = output positions
//
// with the K index laid out so that NO operand is gathered element by element:
//   * kx is padded from 5 to 6: k = (ci, ky, kx6), pairs P = (ci, ky, kx6 / 2): Cin * 15 pairs = 45 -> 48 (three K steps of 32 = 16 pairs; one
//     input channel: 15 -> 16, ONE K step).  A pair of the B operand is X[ci][y + ky][x + 2 kxp], [.. + 1]: two adjacent halves of the image =
//     one aligned ds_read_b32 — IF x is even.
//   * x = 2 px + dx: the two columns of a pooling window read the SAME pairs X[..][2 px + 2 kxp (+ 1)], kxp = 0 .. 2, when the odd column's
//     weights are shifted by one instead of its image: A_dx[c][(ci, ky, kx6)] = w[c][ci][ky][kx6 - dx] (zero outside 0 .. 4) — six padded taps are
//     exactly what both shifts need.  Two A register sets (dx = 0, 1), ONE B fragment for both: 12 ds_read_b32 feed 24 MFMAs.
//   * N tile = 16 POOLED cells; the four candidates of a cell's 2x2 window are four accumulator tiles of the SAME lane and register (dy: two B
//     fragments, rows 2 py + dy + ky; dx: the two A sets): max-pool, argmax (first maximum wins, as torch), activation and stash are in-lane VALU.
// One wave = one point at a time (its image — three fp16 piece planes of X * 2^e, e from the point's own max |x| — in a wave-private LDS region:
// no block barrier anywhere), 4 waves per block, each running `pw` points of ONE sample against that sample's weights, which a wave holds as
// register-resident triple pieces (scaled by the wave's own max |w|) for its whole life.  P1 / st1 are written in conv1_pool_kernel's layout.
// Arithmetic: rbnn_triple.hip's (operands exact as three pieces, six exact product terms smallest first, fp32 accumulation; K = 75: chains of <= 18).
// =====================================================================================================
template <class G> struct Conv1X3 {
    static constexpr int CIN = G::CIN, IW = G::IW, NPP = G::P1W * G::P1W;
    static constexpr int NPAIR = CIN * 15, KT = (NPAIR + 15) / 16;        // K steps of 32 halves = 16 pairs
    static constexpr int NCT = (NPP + 15) / 16;                           // tiles of 16 pooled cells
    static constexpr int PLB = (G::DIN * 2 + 15) / 16 * 16;               // bytes of one piece plane of the image
    static constexpr int IMGB = 3 * PLB, LDSB = 4 * IMGB;                 // per wave, per block
};

template <int ACT, class G>
__global__ void __launch_bounds__(256, 2) conv1_pool_x3_kernel(const ConvArgs a, int pw) {
    using T = Conv1X3<G>;
    constexpr int CIN = G::CIN, IW = G::IW, NPP = T::NPP, KT = T::KT, P1W_ = G::P1W;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int groups = (a.N + 4 * pw - 1) / (4 * pw);
    int id;
    if (!item_of_block(blockIdx.x, groups * a.S, id)) return;
    const int s = id / groups, n_first = (id % groups) * 4 * pw + wave * pw;
    if (n_first >= a.N) return;                                          // whole wave idle; there is no block barrier
    const int sw = a.sidx ? a.sidx[s] : s;
    char* const img = (char*)lds + wave * T::IMGB;
    union F8 { f16x8 v; unsigned w[4]; };

    // ---- A: the sample's conv1 weights, both column shifts, as register-resident triple pieces.  Lane (li, lg): row = channel 16 mt + li,
    // K elements 32 t + 8 lg + j  <->  pair P = 16 t + 4 lg + (j >> 1) = (ci, ky, kxp), kx6 = 2 kxp + (j & 1)
    F8 A0[2][2][KT], A1[2][2][KT], A2[2][2][KT];                          // [dx][mt][t]
    float w_inv;
    {
        auto weight = [&](int mt, int t, int j, int dx) {
            const int P = 16 * t + 4 * lg + (j >> 1), ci = P / 15, ky = (P % 15) / 3, kx = 2 * (P % 3) + (j & 1) - dx;
            const float* const row = a.K1w + ((long long)sw * C1 + 16 * mt + li) * G::K1 + min(ci, CIN - 1) * 25 + ky * 5;
            return (P < T::NPAIR && kx >= 0 && kx < 5) ? row[min(max(kx, 0), 4)] : 0.f;
        };
        float wmax = 0.f;                                                 // first sweep: the scale (the dx = 0 set holds every weight once) ...
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) wmax = fmaxf(wmax, fabsf(weight(mt, t, j, 0)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        int ew = 0;
        if (wmax > 0.f && wmax < INFINITY) ew = max(-100, min(100, 13 - ilogbf(wmax)));
        const float wsc = ldexpf(1.f, ew);
        w_inv = ldexpf(1.f, -ew);
        asm volatile("" ::: "memory");
        // ... second sweep (L1 hits): eight values at a time become their fragment's pieces — nothing but the pieces stays live
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int t = 0; t < KT; ++t) {
#pragma unroll
                    for (int j = 0; j < 8; j += 2)
                        split3_plain_pair(weight(mt, t, j, dx) * wsc, weight(mt, t, j + 1, dx) * wsc, 1.f, A0[dx][mt][t].w[j >> 1], A1[dx][mt][t].w[j >> 1],
                                          A2[dx][mt][t].w[j >> 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
    }
    float bias[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mt][r] = a.K1b[(long long)sw * C1 + 16 * mt + 4 * lg + r];
    // byte offset of pair (t, q) of this lane inside a plane, relative to the window's top-left pixel; pairs past Cin * 15 (their weights are zero) read pair 0
    int koff[KT][4];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int P = 16 * t + 4 * lg + q, Pc = P < T::NPAIR ? P : 0;
            koff[t][q] = 2 * ((Pc / 15) * (IW * IW) + ((Pc % 15) / 3) * IW + 2 * (Pc % 3));
        }

    for (int n = n_first; n < min(n_first + pw, a.N); ++n) {
        const long long sn = (long long)s * a.N + n;
        // ---- the point's image: fp32 [ci][y][x] -> three fp16 piece planes of x * 2^ex in the wave's LDS region (pairs of pixels: 4-byte units)
        float out_scale;
        {
            constexpr int NV4 = (G::DIN / 4 + 63) / 64;
            static_assert(G::DIN % 4 == 0, "float4 rows");
            const f32x4* const src = (const f32x4*)(a.X + (long long)n * a.ldx);
            float xmax = 0.f;
#pragma unroll
            for (int i = 0; i < NV4; ++i) {
                const int e4 = lane + 64 * i;
                const f32x4 v = e4 < G::DIN / 4 ? src[e4] : (f32x4){0.f, 0.f, 0.f, 0.f};
                xmax = fmaxf(xmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
            int ex = 0;
            if (xmax > 0.f && xmax < INFINITY) ex = max(-100, min(100, 13 - ilogbf(xmax)));
            const float xsc = ldexpf(1.f, ex);
            out_scale = ldexpf(1.f, -ex) * w_inv;
            asm volatile("" ::: "memory");                               // (the previous point's fragment reads stay in front of these stores: one wave, LDS in order)
            // second sweep over the point (its 3-12 KB sit in L1 / L2 now): the values are not held across the reduction — beside 144 registers of
            // weight pieces that spilled
#pragma unroll 2
            for (int i = 0; i < NV4; ++i) {
                const int e4 = lane + 64 * i;
                const f32x4 v = e4 < G::DIN / 4 ? src[e4] : (f32x4){0.f, 0.f, 0.f, 0.f};
                unsigned d0[2], d1[2], d2[2];
                split3_plain_pair(v[0] * xsc, v[1] * xsc, 1.f, d0[0], d1[0], d2[0]);
                split3_plain_pair(v[2] * xsc, v[3] * xsc, 1.f, d0[1], d1[1], d2[1]);
                if (e4 < G::DIN / 4) {
                    *(uint2*)(img + 8 * e4) = make_uint2(d0[0], d0[1]);
                    *(uint2*)(img + T::PLB + 8 * e4) = make_uint2(d1[0], d1[1]);
                    *(uint2*)(img + 2 * T::PLB + 8 * e4) = make_uint2(d2[0], d2[1]);
                }
            }
            asm volatile("" ::: "memory");                               // same wave: the LDS unit serves these stores before the reads below
        }
        float* const p1 = a.P1 + sn * G::P1SZ;
        uint8_t* const st = a.st1 + sn * G::P1SZ;
#pragma unroll 1
        for (int ct = 0; ct < T::NCT; ++ct) {
            const int cell = 16 * ct + li, cc = min(cell, NPP - 1);      // lanes past the last cell compute a valid cell again, never stored
            const int py = cc / P1W_, px = cc - py * P1W_;
            const char* const base = img + 2 * (2 * py * IW + 2 * px);
            f32x4 acc[2][2][2];                                          // [dy][dx][mt]
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[dy][dx][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    F8 b0, b1, b2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const char* const src = base + koff[t][q] + dy * (2 * IW);
                        b0.w[q] = *(const unsigned*)src;
                        b1.w[q] = *(const unsigned*)(src + T::PLB);
                        b2.w[q] = *(const unsigned*)(src + 2 * T::PLB);
                    }
                    // six exact product terms, smallest first; the four accumulators of a (dy) are independent chains
#define RBNN_C1X3_TERM(AP, BP) \
                    _Pragma("unroll") for (int dx = 0; dx < 2; ++dx) \
                        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) acc[dy][dx][mt] = MFMA_H(AP[dx][mt][t].v, BP.v, acc[dy][dx][mt]);
                    RBNN_C1X3_TERM(A0, b2) RBNN_C1X3_TERM(A2, b0) RBNN_C1X3_TERM(A1, b1) RBNN_C1X3_TERM(A1, b0) RBNN_C1X3_TERM(A0, b1) RBNN_C1X3_TERM(A0, b0)
#undef RBNN_C1X3_TERM
                    __builtin_amdgcn_sched_barrier(0);                   // one fragment set in flight: hoisted, the reads of all six steps spilled beside the 144 weight registers
                }
            // ---- bias, 2x2 max-pool over (dy, dx) with the first maximum winning (torch max_pool2d; candidates in conv1_pool_kernel's order
            // q = 2 dy + dx), activation, stash = argmax | sign bit
            if (cell < NPP) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * mt + 4 * lg + r;
                        float best = 0.f, best_pre = 0.f;
                        int arg = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float pre = acc[q >> 1][q & 1][mt][r] * out_scale + bias[mt][r];
                            const float v = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                            if (q == 0 || v > best) { best = v; best_pre = pre; arg = q; }
                        }
                        p1[c * NPP + cell] = smooth_act<ACT>() ? best : act_fwd<ACT>(best);
                        st[c * NPP + cell] = (uint8_t)(arg | (best_pre > 0.f ? 4 : 0));
                    }
            }
        }
    }
}

// pw = points per wave: enough blocks to fill the chip's 512 slots several times over, few enough that a wave's weight set-up (~100 loads + splits) is amortised
template <int ACT, class G>
int launch_conv1_pool_x3(const ConvArgs& a, hipStream_t st) {
    using T = Conv1X3<G>;
    const char* const pe = getenv("RBNN_CONV1_X3_PW");                     // (read per launch: the tests set it per case)
    const int pw_env = pe ? atoi(pe) : 0;
    const long long pairs = (long long)a.N * a.S;
    // (measured at N = 512, S = 62 on 3x32x32, profiles/r06c: pw 1 / 2 / 4 / 8 -> 0.93 / 0.86 / 0.80 / 0.77 ms; the fp32 VALU kernel: 1.71.)  A block
    // runs pw points per wave after a set-up worth ~0.25 of a point, and the grid runs in ROUNDS of 2 blocks per CU: pw = the value in 1 .. 8 that
    // minimises rounds x (pw + 0.25) — a plain "as large as possible" picked 7 at that size, 2.3 rounds = three of 7 instead of two of 8 (0.90 ms).
    int pw = pw_env;
    if (pw <= 0) {
        const long long slots = 2LL * device_cus();
        double best = 1e300;
        for (int c = 1; c <= 8; ++c) {
            const long long blocks = ((a.N + 4LL * c - 1) / (4LL * c)) * a.S;
            const double cost = (double)((blocks + slots - 1) / slots) * (c + 0.25);
            if (cost < best - 1e-9) { best = cost; pw = c; }
        }
    }
    (void)pairs;
    static unsigned long long attr = 0;
    if (!ensure_dynamic_lds((const void*)conv1_pool_x3_kernel<ACT, G>, T::LDSB, attr)) return RBNN_ERR_LAUNCH;
    const long long groups = (a.N + 4LL * pw - 1) / (4LL * pw);
    hipLaunchKernelGGL((conv1_pool_x3_kernel<ACT, G>), dim3(grid_for_items(groups * a.S)), dim3(256), T::LDSB, st, a, pw);
    return launch_status();
}

template <int ACT, class G>
int launch_conv_forward_x3(const ConvArgs& a, const ConvX3Args& x, hipStream_t st) {
    constexpr int WROWS = (G::CIN == 1 ? 256 : 128);
    // conv1 + pool: on the f16 pipe too (conv1_pool_x3_kernel, round 6); RBNN_CONV1_X3=0 keeps the fp32 VALU kernel of rbnn_conv.hip (the A/B of tests and profiles)
    const char* const ce = getenv("RBNN_CONV1_X3");
    const bool c1x3 = !ce || ce[0] != '0';
    int rc = c1x3 ? launch_conv1_pool_x3<ACT, G>(a, st) : launch_conv1_pool(ACT, G::CIN, a, st);
    if (rc) return rc;
    constexpr int LDSB = ConvX3Lds<G, WROWS>::BYTES;
    static unsigned long long attr = 0;                                   // per instantiation, one bit per device
    if (!ensure_dynamic_lds((const void*)conv2_pool_x3_kernel<ACT, G, WROWS>, LDSB, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL((conv2_pool_x3_kernel<ACT, G, WROWS>), dim3(grid_for_items((long long)((a.N + 1) / 2) * a.S)), dim3(512), LDSB, st, a, x);
    if ((rc = launch_status())) return rc;
    return launch_conv_fc(a, st);
}

}  // namespace

extern "C" int rbnn_conv_forward_triple(const rbnn_conv_posterior* net, const void* K2_triple, int32_t k2_exp, int32_t p1_exp,
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

namespace {

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

}  // namespace

// =====================================================================================================
// conv2^T, DENSE form (triple-split arithmetic), both geometries: a GEMM per tap over the conv2 OUTPUT positions + col2im
// (a gather over a zero-padded gradient image — the fp32 / split kernels' form — spends 36-39 % of its MFMAs on padding):
//
//     T[tap][ci][pos2] = sum_hc W[hc][ci][tap] * dO2[hc][pos2]          pos2 over the O2W x O2W conv2 outputs: M = 32 ci, N = 64 per pass, K = Hc
//     dP1[ci][y + ky][x + kx] += T[(ky, kx)][ci][(y, x)]                 col2im, once per (sample, point)
//
// One block = one (sample, point), 8 waves = 2 input-channel tiles x 4 tap groups (7 + 6 + 6 + 6 taps; the 7-tap groups of the two channel
// tiles sit on different SIMDs): a wave holds T of its taps in <= 28 accumulator tiles over the WHOLE Hc loop.  The conv2 output positions are
// covered in PASSES of <= 4 position tiles (1x28x28: one pass of 64; 3x32x32: 64 + 36).  Per K step of 32 channels the block
//   (a) routes the pooled gradients of those channels through the pool-2 argmax / activation derivative into a dense channel-last image
//       [64 positions][32 hc] of three fp16 piece planes (12 KB; the rows of dQ2 / of the stash arrive by 4-byte LDS-DMA into a staging buffer;
//       staging and image are double-buffered: ONE barrier per K step),
//   (b) reads its B fragments ONCE (12 ds_read_b128 feed 7 taps = 168 MFMAs),
//   (c) takes its A fragments — model.3.weight regrouped [K step][tap][ci][32 hc] as a triple image — STRAIGHT FROM MEMORY INTO REGISTERS
//       (round 5).  Rounds 3-4 brought every weight tile in by LDS-DMA into a private ring of four 3-KiB slots per wave and read it back with
//       ds_read_b128: 21 DMA pieces per wave and K step at 60-180 cycles of issue each (in-kernel stamps: 20k of a block's 208k K-loop cycles),
//       336 KB of LDS traffic per K step beside the 96 KB of B-fragment reads, and 96 KB of LDS that forced the col2im images to alias the loop
//       buffers (two block-wide barriers, a zero fill and a partial gather per pass, 16 registers of partial sums across the passes).  The image
//       is now laid out fragment-major — a tile's piece is [4 K chunks][16 ci][16 B], lane l of a wave owns bytes 16 l .. 16 l + 15 — so a
//       piece is ONE global_load_dwordx4 of 1 KiB contiguous, the compiler counts the waits, and two register sets alternate: tap t + 1's
//       three pieces are requested at the top of tap t (an L2 round trip is ~200-500 cycles, a tap is 24 MFMAs = 384 cycles of this wave and
//       as many of its SIMD partner).
//   vmcnt counts in issue order, so a weight load issued behind a staging piece (first touch: HBM) cannot be consumed before that piece has
//   landed — an HBM round trip under load is several thousand cycles, a tap 770.  Only TWO waves issue staging pieces (16-byte LDS-DMA, 1 KiB
//   per piece, half of a K step's rows each, at the top of a K step behind tap 1's loads): waves 2 and 7, 6-tap waves on the two SIMDs whose
//   other wave is a 6-tap wave too.  While such a wave waits for its pieces its SIMD partner has the matrix pipe to itself, and the two of them
//   have a tap of slack per K step against the SIMDs that hold a 7-tap wave — where nobody ever waits for HBM.  (All six 6-tap waves staging:
//   214k cycles per block in the K loops, a 7-tap wave 75k of them at the K-step barrier, profiles/r05e.)
// col2im: every wave adds the T tiles of its own taps into its OWN [P1W x P1W][16 ci] fp32 image in LDS (read - add - write of one ds_*_b128
// per accumulator tile, fixed order, nobody else touches it: no atomics, no barrier) right after a pass's K loop; the images persist over the
// passes (zeroed once).  After the last pass one barrier, then one output position x two channel quads per thread adds the four images of a
// channel tile in wave order, scales, folds act' in for sigmoid / tanh and writes dP1 — and leaves max |dP1| in G[sn][0] for conv1_bwd_x3_kernel.
// =====================================================================================================
// RBNN_DENSE_STAMPS (diagnostic build, fenced like the ablation switches; the results stay right): s_memtime stamps of waves 0 and 3 of every
// block, summed per segment into rbnn_dense_stamp_acc and read back by rbnn_debug_dense_stamps (tools/dense_stamps.py): per pass the prologue,
// the K loop and the col2im; slot 24 the whole block, 25 the block count, 26 the time inside the K loop's barriers (level 2 only).
#ifdef RBNN_DENSE_STAMPS
#ifndef RBNN_DENSE_STAMP_WA
#define RBNN_DENSE_STAMP_WA 0                                              // the two stamped waves (default: the 7-tap wave 0 and the plain 6-tap wave 3; 2 and 7 are the staging waves)
#define RBNN_DENSE_STAMP_WB 3
#endif
__device__ unsigned long long rbnn_dense_stamp_acc[64];
#define DSTAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (lane == 0 && (wave == RBNN_DENSE_STAMP_WA || wave == RBNN_DENSE_STAMP_WB)) atomicAdd(&rbnn_dense_stamp_acc[(wave == RBNN_DENSE_STAMP_WB ? 32 : 0) + (slot)], t_ - tprev); tprev = t_; } while (0)
#define DSTAMP_ADD(slot, v) do { if (lane == 0 && (wave == RBNN_DENSE_STAMP_WA || wave == RBNN_DENSE_STAMP_WB)) atomicAdd(&rbnn_dense_stamp_acc[(wave == RBNN_DENSE_STAMP_WB ? 32 : 0) + (slot)], (unsigned long long)(v)); } while (0)
#else
#define DSTAMP(slot) do { } while (0)
#define DSTAMP_ADD(slot, v) do { } while (0)
#endif
namespace {
template <class G> struct ConvBwdDenseLds {
    static constexpr int NPASS = (G::NPT2 + 3) / 4;                       // passes of <= 4 position tiles: 7 taps x 4 tiles is what a wave's accumulators hold (112 registers)
    static constexpr int PLANE = 64 * 64, IMG = 3 * PLANE;                // one piece plane of the routed image: 64 position records of 32 hc halves
    static constexpr int NFL = 32 * G::NP2;                               // pooled cells of one K step
    static constexpr int STG = (NFL * 5 + 4 * G::NP2 + 15) / 16 * 16;     // staging: NFL dQ2 floats + NFL stash bytes + 4 NP2 bytes that hold the code 8 (no window's: the
                                                                          // windows of a position that lie off the pooled map read their stash byte here)
    static constexpr int SOFF = 2 * IMG;                                  // two images, two staging buffers,
    static constexpr int EOFF = SOFF + 2 * STG;                           // the eight waves' col2im images [P1W x P1W output positions][16 ci] floats,
    static constexpr int EIMG = G::P1W * G::P1W * 64;
    static constexpr int MOFF = EOFF + 8 * EIMG;                          // the eight wave maxima (64 B),
    static constexpr int DOFF = MOFF + 64;                                // the waves' dummy records (1 KiB each: lanes past the last position read / write these)
    static constexpr int BYTES = DOFF + 8 * 1024;
    static_assert(BYTES <= 160 * 1024, "LDS");
    static_assert(NPASS * 64 >= G::NPOS && (NPASS - 1) * 64 < G::NPOS, "passes of 64 positions");
};

template <int ACT, class G>
__global__ void __launch_bounds__(512, 2) conv_bwd_dense_x3_kernel(const ConvBwdArgs a, const char* __restrict__ K2d, int k2_exp, float fw_l1) {
    using L = ConvBwdDenseLds<G>;
    constexpr int P1W_ = G::P1W, O2W_ = G::O2W, P2W_ = G::P2W, NP2_ = G::NP2, NPOS_ = G::NPOS, NFL = L::NFL, NPASS = L::NPASS;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* const lds = (char*)lds_f;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave >> 2, q = wave & 3;
    // taps of this wave: 7 for q == ct (waves 0 and 5: SIMDs 0 and 1), 6 for the others, in tap order
    const int ntap = 6 + (q == ct ? 1 : 0);
    const int tap0 = 6 * q + (q > ct ? 1 : 0);
    // staging role (see the header): waves 2 and 7 — 6-tap waves on the two SIMDs that hold two 6-tap waves — bring in the first / second half of a
    // K step's rows; 0: this wave issues no staging piece
    const int stg_role = wave == 2 ? 1 : (wave == 7 ? 2 : 0);

    int id;
    if (!item_of_block(blockIdx.x, a.N * a.S, id)) return;
    const int s = id / a.N, n = id % a.N;
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2_, KS = (a.Hc + 31) / 32;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;
#ifndef RBNN_DENSE_PRIO
#define RBNN_DENSE_PRIO 1
#endif
    // A SIMD issues from its OLDEST ready wave: of the two waves that share one, the lower-numbered ran ahead every K step and its partner did the
    // rest of its taps alone, its stalls uncovered (stamps: a 7-tap wave 47k of 189k K-loop cycles at the barrier, its partner 15k).  The wave of a
    // SIMD that must not be the one left alone — the 7-tap waves (0, 5), the staging waves (2, 7) — takes the higher issue priority.
    // (RBNN_DENSE_PRIO: 1 = those four waves, always — the default; 2 = the 7-tap waves only; 3 = the four, but only through tap 3 of every K step;
    //  4 = the OTHER four waves; measured in profiles/r05q/dense_prio.txt)
    const bool prio_wave = RBNN_DENSE_PRIO == 4 ? !(wave == 0 || wave == 5 || wave == 2 || wave == 7)
                         : (wave == 0 || wave == 5 || (RBNN_DENSE_PRIO != 2 && (wave == 2 || wave == 7)));
    if (RBNN_DENSE_PRIO && prio_wave) __builtin_amdgcn_s_setprio(1);
#ifdef RBNN_DENSE_STAMPS
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
    unsigned long long barw = 0;
#endif

    // per-(sample, point) scale: |dO2| <= 4 * max_c |dZ_c| * fw_l1.  The load is issued here; the scales are formed in the prologue, behind the
    // staging DMA / weight load issue (forming them here put the load's round trip in front of theirs)
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

    // Staging of K step ks into buffer buf: the step's rows of dQ2 (NFL floats) and of the stash (NFL bytes) are contiguous in memory, 16-byte aligned
    // (Hc % 16 == 0) and copied verbatim by 16-byte LDS-DMA, 1 KiB per piece (rounds 3-4: 4-byte pieces of 256 B, four times as many).  The pieces are
    // numbered dQ2 first, then stash; staging wave `half` (1 / 2) issues the first / second half of them.  Source = a wave-uniform 64-bit base + ONE
    // 32-bit per-lane offset; consecutive pieces share the address and M0 (the immediate offset of global_load_lds applies to both sides).  wholec:
    // the step is known to be a whole one (Hc % 32 == 0) — a piece that lies inside a whole step's rows needs no per-lane test.
    constexpr int NPQ = (NFL * 4 + 1023) / 1024, NPS = (NFL + 1023) / 1024, NPH = (NPQ + NPS + 1) / 2;   // pieces: dQ2 rows, stash rows, first half
    auto stage_issue = [&](int ks, int buf, auto wholec, auto halfc) {
        constexpr bool WH = decltype(wholec)::value;
        constexpr int HALF = decltype(halfc)::value;
        char* const S = lds + L::SOFF + buf * L::STG;
        const int nvalid = min(32, a.Hc - 32 * ks) * NP2_;                  // cells of the step (a multiple of 16)
        const long long fb = sn * F + (long long)ks * NFL;
        const char* const qrow = (const char*)(a.dQ2 + fb);
        const char* const srow = (const char*)(a.st2 + fb);
        const unsigned l16 = 16u * (unsigned)lane;
        static_for<(HALF == 1 ? 0 : NPH), (HALF == 1 ? NPH : NPQ + NPS)>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr bool ISQ = i < NPQ;
            constexpr int j = ISQ ? i : i - NPQ;                           // piece j of its region
            constexpr int bytes = ISQ ? NFL * 4 : NFL;
            const char* const src = ISQ ? qrow : srow;
            char* const dst = S + (ISQ ? 0 : NFL * 4);
            // (the instruction's immediate offset is a signed 13-bit field: groups of four pieces share an address and an M0, offsets 0 .. 3072)
            constexpr int j0 = j & ~3;
            if ((WH && 1024 * (j + 1) <= bytes) || 1024u * j + l16 < (unsigned)(ISQ ? 4 * nvalid : nvalid))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (1024u * j0 + l16)),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)(dst + 1024 * j0), 16, 1024 * (j - j0), 0);
        });
    };
    auto stage_by_role = [&](int ks, int buf) {                            // outside the K loop: wave-uniform branches
        if (stg_role == 1) stage_issue(ks, buf, std::false_type{}, std::integral_constant<int, 1>{});
        else if (stg_role == 2) stage_issue(ks, buf, std::false_type{}, std::integral_constant<int, 2>{});
    };
    // A operand: tile (K step, tap) of this wave's channel tile = three 1-KiB pieces [piece][4 K chunks][16 ci][16 B] (conv.py::_build_dense,
    // conv_k2_images_kernel): lane (li, lg) owns row li, chunk lg = bytes 16 * lane of every piece.  A sample's image is KS * 25 * 6 KiB < 4 GB:
    // source = a block-uniform 64-bit base (SGPR pair) + ONE 32-bit per-lane offset, the pieces by the instruction's immediate offset.
    const char* const Awave = K2d + ((long long)sw * KS * 25 * 2 + ct) * 3072 + 16 * lane;
    struct ATile { f16x8 p0, p1, p2; };
    auto tile_load = [&](int iks_, int tapi, ATile& A) {
        const char* const src = Awave + (unsigned)((iks_ * 25 + tap0 + tapi) * 6144);
        A.p0 = *(const f16x8*)src;
        A.p1 = *(const f16x8*)(src + 1024);
        A.p2 = *(const f16x8*)(src + 2048);
    };
    const int foff = li * 64 + ((lg ^ swz(li)) * 16);                       // B fragment of position li (of a tile), K chunk lg
    constexpr int NPP = P1W_ * P1W_;
    static_assert(2 * NPP <= 512, "one thread per output position and pair of channel quads");
    char* const eimg = lds + L::EOFF + wave * L::EIMG;                      // this wave's col2im image
#ifdef RBNN_DENSE_ABL_NOMFMA
#define DENSE_MFMA(A, B, C) (C)
#else
#define DENSE_MFMA(A, B, C) MFMA_H(A, B, C)
#endif

    // ---- once per block: the never-matching stash bytes of both staging buffers, the wave's col2im image zeroed, K steps 0 and 1 staged together
    // (one HBM round trip, not two), the first weight tile requested ----
    for (int i = tid; i < 2 * ((L::STG - NFL * 5) / 4); i += 512) {
        constexpr int ND = (L::STG - NFL * 5) / 4;
        *(unsigned*)(lds + L::SOFF + (i / ND) * L::STG + NFL * 5 + 4 * (i % ND)) = 0x08080808u;
    }
    stage_by_role(0, 0);
    if (KS > 1) stage_by_role(1, 1);
    for (int i = lane; i < L::EIMG / 16; i += 64) *(f32x4*)(eimg + 16 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    static_for<0, NPASS>([&](auto PASS) {
    constexpr int pass = decltype(PASS)::value;
    constexpr int NPT = (G::NPT2 - 4 * pass) < 4 ? (G::NPT2 - 4 * pass) : 4;   // position tiles of this pass
    // routing role of this thread: position gp = 64 * pass + lane (gy, gx) of the O2W x O2W gradient map, channel quad qd = wave of the K step's 32
    const int gp = 64 * pass + lane, gy = gp / O2W_, gx = gp % O2W_, qd = wave;
    // byte offsets of this thread's four windows' cells inside a staging buffer (channel j of its quad: + j * NP2 cells).  The stash code
    // that routes window w here is w (argmax) with bit 2 = the pre-activation was positive; stash bytes are <= 7 by construction, so they
    // are compared WHOLE (no masking), and a window off the map (or a lane past the last position: its image row is zeros) reads the byte 8
    // ONE register per window: the cell index c — its dQ2 float sits at 4 c, its stash byte at NFL * 4 + c.  A window off the map (or a lane past
    // the last position) names cell NFL: its "stash byte" is the first never-matching byte (NFL * 5; + j NP2 <= 4 NP2 of them) and its "float" four
    // stash bytes (values <= 8 each: a finite, tiny number, multiplied by a zero factor)
    int cw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {                                          // window w = 2dy + dx of the <= 4 stride-1 pooling windows containing (gy, gx)
        const int py = gy - (w >> 1), px = gx - (w & 1);
        const bool ok = gp < NPOS_ && py >= 0 && py < P2W_ && px >= 0 && px < P2W_;
        cw[w] = ok ? 4 * qd * NP2_ + py * P2W_ + px : NFL;
    }
    const int rec = lane * 64 + (((qd >> 1) ^ swz(lane)) * 16) + (qd & 1) * 8;   // this thread's 8 bytes of a piece plane
    // routing of ONE channel (j of this thread's quad) of K step ks from staging buffer sbuf: pool-2 argmax + activation derivative (gather
    // form), scaled, split into the three pieces.  Called between the MFMA groups of the previous K step so that its LDS reads and
    // vector work issue under the matrix pipe (one basic block with the MFMAs: no branch in between).
    union Q { unsigned w[2]; uint2 u; };                                    // pieces of the thread's four channels: [j0 | j1 << 16], [j2 | j3 << 16]
    float vpend = 0.f;                                                     // the even channel of a pair waits for the odd one (split3_plain_pair)
    // (two halves: the eight LDS reads of a channel are issued one tap AHEAD of the selects that consume them — in one piece the selects
    // waited for the reads right in front of the tap's MFMAs, an LDS round trip per routing tap with nothing issued behind it)
    struct RouteIn { int st[4]; float dq[4]; };
    auto route_load = [&](int sbuf, int j, RouteIn& in) {
        const char* const sb = lds + L::SOFF + sbuf * L::STG;
#pragma unroll
        for (int w = 0; w < 4; ++w) {                                      // eight independent LDS reads
            in.st[w] = *(const unsigned char*)(sb + NFL * 4 + j * NP2_ + cw[w]);
            in.dq[w] = *(const float*)(sb + 4 * j * NP2_ + 4 * cw[w]);
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
    auto route_store = [&](int ibuf, const Q& p0, const Q& p1, const Q& p2) {
        char* const I = lds + ibuf * L::IMG;
        *(uint2*)(I + rec) = p0.u;
        *(uint2*)(I + L::PLANE + rec) = p1.u;
        *(uint2*)(I + 2 * L::PLANE + rec) = p2.u;
    };

    f32x4 acc[7][NPT];

    // ---- prologue of the pass: staging of K steps 0 / 1 has been issued (by the block's prologue, or ahead of the previous pass's col2im); the first
    // weight tile; the image of K step 0 for this pass's positions ----
    // Register sets of the A operand: tap g (running index over the pass) of a wave WITHOUT a staging role reads set g & 1 and requests tap g + 1;
    // a staging wave (6 taps, 16 accumulator registers to spare) runs THREE sets — tap t reads set t % 3 and requests tap t + 2 — so that the first
    // wait that covers its staging pieces (issued at tap 0 behind tap 2's request) is that of tap 3
    ATile A0, A1, A2;
    tile_load(0, 0, A0);
    if (stg_role) tile_load(0, 1, A1);
    if (pass == 0) set_scales();
    ring_wait_barrier<0>();                                                // every wave's staging pieces have landed (and, pass > 0: every wave is out of the previous K loop)
    {
        Q p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            RouteIn in;
            route_load(0, j, in);
            route_calc(0, j, in, p0, p1, p2);
        }
        route_store(0, p0, p1, p2);
    }
    // One K step, PAR = parity of the running index of its first tap (the register set tap 0 reads): a wave with an even tap count (6) always
    // runs PAR = 0; a 7-tap wave alternates, so its loop below is unrolled over two K steps.  A step is straight-line code: a tap is one
    // scheduling region (the whole step as one region: the scheduler hoists the routing reads across taps into 256 registers and scratch), tile
    // and staging requests are ALWAYS issued (past the end: the last step's tile / rows again, never read).
    // (a volatile asm: as a builtin the lane id is loop-invariant — hoisted, held over the loop, spilled again)
    auto fresh_foff = [&]() {
        int l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
        return (l_ & 15) * 64 + (((l_ >> 4) ^ swz(l_ & 15)) * 16);
    };
    int foff_next = fresh_foff();
    auto kstep = [&](int ks, auto NTC, auto PARC, auto WHOLEC, auto STGC) {
        constexpr int NT = decltype(NTC)::value, PAR = decltype(PARC)::value, STGR = decltype(STGC)::value;
        // image ks complete; every wave's staging pieces of step ks + 1 landed (issued a whole step ago: older than the three weight loads in
        // flight for tap 0, which stay outstanding); image / staging ks - 1 free.  (A raw s_barrier behind the counted wait: __syncthreads()
        // makes hipcc drain vmcnt to 0 first.)
#if defined(RBNN_DENSE_STAMPS) && RBNN_DENSE_STAMPS >= 2
        const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef RBNN_DENSE_ABL_HALFBAR
        if (ks & 1) {                                                      // ablation (timing only, wrong results): a barrier every OTHER K step — what a 64-channel step could gain at most
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(VMCNT_LGKM0((STGR ? 6 : 3)));
            asm volatile("" ::: "memory");
        } else
#endif
        ring_wait_barrier<(STGR ? 6 : 3)>();
#if defined(RBNN_DENSE_STAMPS) && RBNN_DENSE_STAMPS >= 2
        barw += __builtin_amdgcn_s_memtime() - tb_;
#endif
        const int ksn = min(ks + 1, KS - 1);
        // (the fragment offset: held in a register over the whole K loop it was the one value the <3,32> instantiations reloaded from scratch at
        // the top of every step, behind s_waitcnt vmcnt(0).  It is re-derived from a fresh lane id under the LAST tap's MFMAs of the previous
        // step — fresh_foff() below — and so lives only from there to the twelve reads here: no spill, and nothing in front of the reads)
        const int foff_s = foff_next;
        const char* const I = lds + (ks & 1) * L::IMG + foff_s;
        f16x8 b0[NPT], b1[NPT], b2[NPT];
        // (plane by plane, in the order the first tap's product groups want them: the first group starts after four reads, not ten)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) b0[pt] = *(const f16x8*)(I + pt * 1024);
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) b1[pt] = *(const f16x8*)(I + L::PLANE + pt * 1024);
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) b2[pt] = *(const f16x8*)(I + 2 * L::PLANE + pt * 1024);
        Q p0, p1, p2;
        RouteIn rin;
        static_for<0, NT>([&](auto TC) {
            constexpr int t = decltype(TC)::value;
            constexpr int LA = STGR ? 2 : 1;                                // taps of lookahead
            static_assert(!STGR || (NT % 3 == 0 && PAR == 0), "three sets: the set of a tap is t % 3 in every K step");
            constexpr int ic = STGR ? t % 3 : (PAR + t) & 1, in = STGR ? (t + 2) % 3 : (PAR + t + 1) & 1;
            ATile& cur = ic == 0 ? A0 : (ic == 1 ? A1 : A2);
            ATile& nxt = in == 0 ? A0 : (in == 1 ? A1 : A2);
            // a later tap's tile into a free register set (the compiler places the counted wait in front of the first MFMA that reads `cur`)
            if constexpr (RBNN_DENSE_PRIO == 3) {                          // (wave-uniform scalar branches: two per K step)
                if (t == 0 && prio_wave) __builtin_amdgcn_s_setprio(1);
                if (t == 4 && prio_wave) __builtin_amdgcn_s_setprio(0);
            }
            tile_load((t + LA) / NT ? ksn : ks, (t + LA) % NT, nxt);
            // (the staging role is a compile-time fact here, so that the counted waits of the other waves know of no staging piece)
            if constexpr (t == 0 && STGR != 0) stage_issue(min(ks + 2, KS - 1), ks & 1, WHOLEC, STGC);   // behind tap 2's request: first covered by the wait of tap 3
            __builtin_amdgcn_sched_barrier(0);                             // the requests stay at the top of the tap (the scheduler sank them below the MFMAs: no time left to land)
            // the six product groups, small terms first (a2*b0 and a1*b1 are the 2^-22 terms, a1*b0 and a0*b1 the 2^-11 ones)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p2, b0[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p1, b1[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p1, b0[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p0, b2[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p0, b1[pt], acc[t][pt]);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = DENSE_MFMA(cur.p0, b0[pt], acc[t][pt]);
#ifndef RBNN_DENSE_ABL_NOROUTE
            // the next K step's image: channel j of the thread's quad is read in tap j and routed in tap j + 1
            if constexpr (t >= 1 && t < 5) route_calc(ks + 1, t - 1, rin, p0, p1, p2);
            if constexpr (t < 4) route_load((ks + 1) & 1, t, rin);
            if constexpr (t == 4) route_store((ks + 1) & 1, p0, p1, p2);
#endif
            if constexpr (t == NT - 1) foff_next = fresh_foff();            // for the next step's fragment reads (see the top of the step)
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    DSTAMP(8 * pass + 0);                                                  // prologue (pass 0: from the block's start; later passes: from the end of the previous col2im)
    // The K loop and the col2im of the pass, instantiated per (tap count, staging role) of the wave and chosen once: a 6-tap wave's seventh
    // accumulator row does not exist in its instantiation (16 registers: the staging waves' third A set)
    auto run_pass = [&](auto NTC, auto STGC) {
    constexpr int NT = decltype(NTC)::value;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[t][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        using Z = std::integral_constant<int, 0>;
        using O = std::integral_constant<int, 1>;
        auto kloop = [&](auto WHOLEC) {
            if constexpr (NT == 7) {
                for (int ks = 0; ks < KS; ks += 2) {
                    kstep(ks, NTC, Z{}, WHOLEC, STGC);
                    if (ks + 1 < KS) kstep(ks + 1, NTC, O{}, WHOLEC, STGC);
                }
            } else {
                for (int ks = 0; ks < KS; ++ks) kstep(ks, NTC, Z{}, WHOLEC, STGC);
            }
        };
        if ((a.Hc & 31) == 0) kloop(std::true_type{}); else kloop(std::false_type{});      // block-uniform
    }
    if (pass + 1 < NPASS) {                                                // the next pass's first two K steps: they land under this pass's col2im
        stage_by_role(0, 0);
        if (KS > 1) stage_by_role(1, 1);
    }
    DSTAMP(8 * pass + 1);                                                  // K loop
#ifdef RBNN_DENSE_ABL_NOEPI
    {                                                                      // ablation (timing only): the accumulators stay live, nothing is gathered
        float sink = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) sink += acc[t][pt][0] + acc[t][pt][1] + acc[t][pt][2] + acc[t][pt][3];
        if (sink == 1.2345e-30f) a.dP1[sn * G::P1SZ] = sink;
    }
#else
    // ---- col2im of this pass into the wave's own image (no barrier: nobody else touches it).  The channel quads of an output position are
    // XOR-swizzled by (position >> 2) & 3: the 16 lanes of a ds_*_b128 service group cover positions p .. p + 3 and p + 12 .. p + 15 of a
    // tile, whose 64-byte records would otherwise share banks four positions apart. ----
    {
        int o0[NPT];
        bool val[NPT];
        const int dummy = (L::DOFF + wave * 1024 + lane * 16) - (L::EOFF + wave * L::EIMG);   // (relative to eimg)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int gpos = 64 * pass + 16 * pt + li;                     // acc[t][pt][r] = T[tap0 + t][ci = 4lg + r][gpos]
            val[pt] = gpos < NPOS_;
            o0[pt] = (gpos / O2W_) * P1W_ + gpos % O2W_;                   // output position of tap (0, 0)
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = tap0 + t, shift = (tap / 5) * P1W_ + tap % 5;
            f32x4 cur[NPT];
            int ad[NPT];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {                             // the tile's reads together, then its writes
                const int o = o0[pt] + shift;
                // lanes past the last position (the last pass's last tile) go through a private dummy record instead of an exec-masked
                // block per access (a masked LDS read is waited for inside its block: one round trip each, in series)
                ad[pt] = val[pt] ? o * 64 + ((lg ^ ((o >> 2) & 3)) << 4) : dummy;
                cur[pt] = *(const f32x4*)(eimg + ad[pt]);
            }
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) *(f32x4*)(eimg + ad[pt]) = cur[pt] + acc[t][pt];
        }
    }
#endif
    };
    {
        using N6 = std::integral_constant<int, 6>;
        if (ntap == 7) run_pass(std::integral_constant<int, 7>{}, std::integral_constant<int, 0>{});
        else if (stg_role == 1) run_pass(N6{}, std::integral_constant<int, 1>{});
        else if (stg_role == 2) run_pass(N6{}, std::integral_constant<int, 2>{});
        else run_pass(N6{}, std::integral_constant<int, 0>{});
    }
    DSTAMP(8 * pass + 2);                                                  // col2im
    });

    // ---- the output: one thread = one output position x two channel quads; the four images of a channel tile in wave order ----
    ring_wait_barrier<0>();                                                // every wave's col2im is in LDS
    float omax = 0.f;
    if (tid < 2 * NPP) {
        const int qp = tid / NPP, pp = tid % NPP, xs = (pp >> 2) & 3;
        const char* const rec = lds + L::EOFF + pp * 64;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {                                   // channel tile c2 = waves 4 c2 .. 4 c2 + 3
            f32x4 u0[4], u1[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                u0[q4] = *(const f32x4*)(rec + (4 * c2 + q4) * L::EIMG + (((2 * qp) ^ xs) << 4));
                u1[q4] = *(const f32x4*)(rec + (4 * c2 + q4) * L::EIMG + (((2 * qp + 1) ^ xs) << 4));
            }
            const f32x4 s0 = ((u0[0] + u0[1]) + u0[2]) + u0[3];
            const f32x4 s1 = ((u1[0] + u1[1]) + u1[2]) + u1[3];
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
    if (G::CIN >= RBNN_CONV1_BWD_X3_MINCIN) {                              // max |dP1| of this (sample, point) -> G[sn][0]: the scale of conv1_bwd_x3_kernel (which reads it before it writes G)
        // (the butterfly's lane indices from a fresh lane id: those of set_scales() at the top of the block, kept for reuse here, were five registers
        // spilled over the K loops)
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __int_as_float(__builtin_amdgcn_ds_bpermute((lane_e ^ o) << 2, __float_as_int(omax))));
        float* const wm = (float*)(lds + L::MOFF);
        if (lane == 0) wm[wave] = omax;
        ring_wait_barrier<0>();
        if (tid == 0) {
            float m = wm[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, wm[w]);
            a.G[sn * G::DIN] = m;
        }
    }
#ifdef RBNN_DENSE_STAMPS
    DSTAMP_ADD(24, __builtin_amdgcn_s_memtime() - tstart);
    DSTAMP_ADD(25, 1);
    DSTAMP_ADD(26, barw);
#endif
}
}  // namespace

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
        // dense image: [sample][K step = these 32 hc][tap][input-channel half][3 pieces][4 units of 8 hc][16 ci][8 hc] halves; item = (tap, ci, unit of 8 hc)
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
            uint4* const out = dense_img + (T * 3) * 64 + u * 16 + (ci & 15);      // a piece is [4 K chunks][16 ci][16 B]: lane (li, lg) of the dense kernel owns unit 16 lg + li
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
            int rc2 = launch_conv_fc_bwd(ACT, a, st);                      // dQ2 = dZ . Fw (the fp32 kernel of rbnn_conv.hip)
            if (rc2) return rc2;
            constexpr int LDSB = ConvBwdDenseLds<G>::BYTES;
            static unsigned long long attr = 0;
            if (!ensure_dynamic_lds((const void*)conv_bwd_dense_x3_kernel<ACT, G>, LDSB, attr)) return (int)RBNN_ERR_LAUNCH;
            hipLaunchKernelGGL((conv_bwd_dense_x3_kernel<ACT, G>), dim3(grid_for_items((long long)N * S)), dim3(512), LDSB, st, a, (const char*)K2_dense, k2_exp, fw_l1);
            if ((rc2 = launch_status())) return rc2;
            // conv1^T: on the f16 pipe for more than one input channel (scaled by the max |dP1| the dense kernel left in G[sn][0]); one input
            // channel (1x28x28): two row tiles either way, and the fp32 kernel runs three waves per SIMD — measured 0.73 against 0.79 ms
            if (G::CIN >= RBNN_CONV1_BWD_X3_MINCIN) {
                hipLaunchKernelGGL((conv1_bwd_x3_kernel<ACT, G>), dim3(grid_for_items((long long)((a.N + 3) / 4) * a.S)), dim3(256), 0, st, a);
                return launch_status();
            }
            return launch_conv1_bwd_fp32(ACT, G::CIN, a, st);
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
