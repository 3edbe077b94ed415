// rbnn_triple.hip — the "f16x6" mode of the two big contractions (gfx950 / MI355X): FULL-WIDTH fp32 operands on the f16 matrix pipe.
//
// rbnn_kernels.hip multiplies on v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak); rbnn_split.hip carries an operand as TWO halves
// (22-23 significant bits: narrower than fp32, hence opt-in).  Here every fp32 operand v is carried as THREE halves
//
//      v * 2^e  =  p0 + p1 + p2,     p0 = fp16(v*2^e),  p1 = fp16(v*2^e - p0),  p2 = fp16(v*2^e - p0 - p1)
//
// 3 x 11 significand bits + 2 sign bits cover fp32's 24 exactly: the sum is EXACT (bit for bit the fp32 value) for every
// |v| >= 2^-15 * max|tensor| — e is chosen so that max|v * 2^e| <= 2^14, and fp16's last subnormal bit is 2^-24 — and within
// 2^-39 * max|tensor| absolutely below that.  A product is formed as
//
//      a*b  ~=  a0*b2 + a2*b0 + a1*b1 + a1*b0 + a0*b1 + a0*b0          (six f16 MFMAs, smallest terms first)
//
// each term exact in the fp32 accumulator's multiplier; the three dropped terms (a1*b2, a2*b1, a2*b2) are <= 2^-32 |a*b|.
// So the ONLY rounding is the fp32 accumulation — the same as in the fp32 MFMA kernels (and in any fp32 GEMM), whose
// summation order differs between implementations anyway.  Arithmetic ceiling: 2516.6 / 6 = 419 TFLOP/s fp32-equivalent,
// 2.7x the fp32 MFMA peak.
//
// Images:
//   "triple rows"  [R][ld/32][3][32] halves: per row and per K stage of 32 columns, 64 B of p0, 64 B of p1, 64 B of p2
//                  (6 bytes per element).  A stage tile in LDS is three PLANE tiles of 64-B rows; a lane's MFMA operand
//                  (8 K values) is one ds_read_b128 per plane; chunk swizzle `swz` (rbnn_common.hpp) as for fp32 64-B rows.
//   "triple cols"  [S][H/32][4 lg][3][ld][8] halves: the backward's B operand, K-slot order = the dA generator's output order.
// Lane maps as in rbnn_split.hip (v_mfma_f32_16x16x32_f16): a[j] = A[li][8*lg + j], b[j] = B[8*lg + j][li], acc[r] = D[4*lg + r][li].
#include "rbnn_common.hpp"
#include <algorithm>
#include <cstdlib>

#ifndef RBNN_X3_FWD_SB
#define RBNN_X3_FWD_SB 4                                      // forward: samples per XCD-resident group of blocks
#endif
#ifndef RBNN_X3_BARRIER_END
#define RBNN_X3_BARRIER_END 0
#endif

namespace {

__device__ __forceinline__ void split3(float v, _Float16& p0, _Float16& p1, _Float16& p2) {
    p0 = (_Float16)v;
    float r = v - (float)p0;                                    // exact (Sterbenz / the remainder of a rounding is representable)
    p1 = (_Float16)r;
    r -= (float)p1;
    p2 = (_Float16)r;
}

// The backward's dA pieces for two adjacent hidden units (bits BIT, BIT + 1 of the stash word): act' multiplier m = bit ? cp : cn
// (sign-extended bit-field + bit-select), then the three fp16 pieces of g * m, each by one fused v_fma_mix* that rounds to f16
// and writes its half of the packed result — 11 vector instructions per pair.  (Plain C++ compiles to ~18: hipcc re-derives
// the pieces through f32 round trips and SLP-packs them into v_pk_* ops, and the kernel is bound by vector-instruction ISSUE.)
// FIRST: these are the first reads of a generator MFMA's result — the 7 wait states an XDL write needs before a VALU read
// (the compiler cannot see into the asm to insert them).
// d0 / d1 / d2 are MFMA A operands and the block ends in a vector write (v_fma_mixhi_f16): a VGPR written by a vector instruction needs two wait
// states before an MFMA reads it, and hipcc cannot see into the string.  In this kernel the sixteen blocks of a wave-stage are followed by the
// generator's next point tile or by LDS fragment reads — the first main MFMA that takes da* sits dozens of instructions behind the last block —
// and THE BUILT LIBRARY IS SCANNED for any MFMA closer than two wait states to the vector write of one of its operands
// (tools/kernel_resources.py::mfma_operand_hazards, tests/test_host_cpu.py::test_no_mfma_reads_a_vgpr_inside_the_valu_write_window): a build in
// which the scheduler ever placed one there fails that test.  RBNN_X3_PAIR_NOP=1 pads every block with `s_nop 1` instead (ADVICE r5's other
// option; same-box A/B, profiles/r06k: sixteen pads per wave-stage cost the gradient kernel 0.5-1.7 %, which is why the scan is the fence here;
// split3_plain_pair, rbnn_common.hpp, whose users DID hit the hazard, carries its pad).
#ifndef RBNN_X3_PAIR_NOP
#define RBNN_X3_PAIR_NOP 0
#endif
#if RBNN_X3_PAIR_NOP
#define RBNN_X3_PAIR_PAD "\n\ts_nop 1"
#else
#define RBNN_X3_PAIR_PAD ""
#endif
#define RBNN_X3_PAIR_BODY \
    "v_bfe_i32 %[me], %[mw], %[b0], 1\n\t" \
    "v_bfe_i32 %[mo], %[mw], %[b1], 1\n\t" \
    "v_bfi_b32 %[me], %[me], %[cp], %[cn]\n\t" \
    "v_bfi_b32 %[mo], %[mo], %[cp], %[cn]\n\t" \
    "v_fma_mixlo_f16 %[d0], %[ge], %[me], 0\n\t" \
    "v_fma_mixhi_f16 %[d0], %[go], %[mo], 0\n\t" \
    "v_fma_mix_f32 %[re], %[ge], %[me], -%[d0] op_sel_hi:[0,0,1]\n\t" \
    "v_fma_mix_f32 %[ro], %[go], %[mo], -%[d0] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
    "v_cvt_pk_f16_f32 %[d1], %[re], %[ro]\n\t" \
    "v_fma_mixlo_f16 %[d2], -%[d1], %[one], %[re] op_sel_hi:[1,0,0]\n\t" \
    "v_fma_mixhi_f16 %[d2], -%[d1], %[one], %[ro] op_sel:[1,0,0] op_sel_hi:[1,0,0]" RBNN_X3_PAIR_PAD
#define RBNN_X3_PAIR_OPS \
    : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [me] "=&v"(me), [mo] "=&v"(mo), [re] "=&v"(re), [ro] "=&v"(ro) \
    : [ge] "v"(ge), [go] "v"(go), [mw] "v"(mw), [cp] "v"(cp), [cn] "v"(cn), [one] "v"(one), [b0] "n"(BIT), [b1] "n"(BIT + 1)
template <int BIT, bool FIRST>
__device__ __forceinline__ void split3_pair(float ge, float go, unsigned mw, unsigned cp, unsigned cn, float one,
                                            unsigned& d0, unsigned& d1, unsigned& d2) {
    unsigned me, mo;
    float re, ro;
    // (FIRST: eight wait states behind v_mfma_f32_16x16x32_f16 before its result is read — hipcc's own padding for that pair; the built library is
    // scanned for it: tools/kernel_resources.py::mfma_result_hazards.  Counting the block's four mask instructions towards them, `s_nop 3`, measured
    // equal — 3.66 vs 3.69 ms, profiles/r06k/first_nop_ab.txt: the result is not there earlier either way — so the full pad stays)
    if constexpr (FIRST) asm volatile("s_nop 7\n\t" RBNN_X3_PAIR_BODY RBNN_X3_PAIR_OPS);
    else asm volatile(RBNN_X3_PAIR_BODY RBNN_X3_PAIR_OPS);   // volatile: the pairs of one MFMA result stay behind the FIRST one
}

// The same with the two multipliers given (sigmoid / tanh: act' comes from an fp32 stream, already times 2^GEN_Q3): 7 per pair.
template <bool FIRST>
__device__ __forceinline__ void split3_pair_m(float ge, float go, float me, float mo, float one, unsigned& d0, unsigned& d1, unsigned& d2) {
    float re, ro;
#define RBNN_X3_PAIRM_BODY \
    "v_fma_mixlo_f16 %[d0], %[ge], %[me], 0\n\t" \
    "v_fma_mixhi_f16 %[d0], %[go], %[mo], 0\n\t" \
    "v_fma_mix_f32 %[re], %[ge], %[me], -%[d0] op_sel_hi:[0,0,1]\n\t" \
    "v_fma_mix_f32 %[ro], %[go], %[mo], -%[d0] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
    "v_cvt_pk_f16_f32 %[d1], %[re], %[ro]\n\t" \
    "v_fma_mixlo_f16 %[d2], -%[d1], %[one], %[re] op_sel_hi:[1,0,0]\n\t" \
    "v_fma_mixhi_f16 %[d2], -%[d1], %[one], %[ro] op_sel:[1,0,0] op_sel_hi:[1,0,0]" RBNN_X3_PAIR_PAD
#define RBNN_X3_PAIRM_OPS \
    : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [re] "=&v"(re), [ro] "=&v"(ro) \
    : [ge] "v"(ge), [go] "v"(go), [me] "v"(me), [mo] "v"(mo), [one] "v"(one)
    if constexpr (FIRST) asm volatile("s_nop 7\n\t" RBNN_X3_PAIRM_BODY RBNN_X3_PAIRM_OPS);
    else asm volatile(RBNN_X3_PAIRM_BODY RBNN_X3_PAIRM_OPS);
}

// (split3_plain_pair — two plain fp32 values, 6 vector instructions per pair — lives in rbnn_common.hpp: the conv kernels use it too)

// ===================================================================================================
// fp32 rows -> triple-rows image.  One thread per (row, group of 8 columns): three 16-B stores.
// ===================================================================================================
// grouped (rbnn_triple_rows_grouped, the fc forward's operands): 16 consecutive rows share a 3-KiB block per K stage, [3 pieces][16 rows][64 B] —
// the order a 16-row group has in the forward kernel's stage tile, so that its three LDS-DMA pieces differ by 1 KiB on BOTH sides and share
// one address register and one M0 write (the immediate offset of global_load_lds applies to the global and the LDS address).
__global__ void triple_rows_kernel(const float* __restrict__ src, long long rows, int cols, int ld_src, float scale,
                                   const rbnn_dev_scale* __restrict__ ds, uint4* __restrict__ dst, int groups, int grouped) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * groups) return;
    if (ds) scale = ds->scale;
    const long long r = i / groups;
    const int g = (int)(i % groups);
    const float* const p = src + r * ld_src + 8 * g;
    union { f16x8 v; uint4 u; } o[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (8 * g + j < cols) ? p[j] * scale : 0.f;
        _Float16 a, b, c;
        split3(v, a, b, c);
        o[0].v[j] = a; o[1].v[j] = b; o[2].v[j] = c;
    }
    if (grouped) {                                             // 16-B units: block of (row group, stage) = 192 units: [plane][16 rows][4 chunks]
        uint4* const out = dst + (((r >> 4) * (groups >> 2) + (g >> 2)) * 192 + (r & 15) * 4 + (g & 3));
        out[0] = o[0].u; out[64] = o[1].u; out[128] = o[2].u;
        return;
    }
    // 16-B units: row r has (groups / 4) stages of 12 units: [plane][4 chunks]
    uint4* const out = dst + (r * (groups >> 2) + (g >> 2)) * 12 + (g & 3);
    out[0] = o[0].u; out[4] = o[1].u; out[8] = o[2].u;
}

// ===================================================================================================
// T1: stacked forward.  One block = one (BN-point tile, sample) item, WH x WN waves, each HTW x NTW accumulator tiles:
//   acc[h][n] = sum_d W[s][h][d] * X[n][d]        (A operand = W rows, B operand = X rows; D = [h][n])
// K runs in stages of 32 columns = one MFMA K step.  A stage tile is 3 planes x (BH + BN) rows of 64 B, brought in by LDS-DMA
// in 1-KiB pieces of 16 rows; physical 16-B chunk of logical chunk c in row r is c ^ swz(r), applied on the SOURCE address.
// Epilogue = the exact kernel's (bias, activation, 1-bit stash, skinny H->C layer on the fp32 MFMA, softmax).
// ===================================================================================================
struct FwdX3Args {
    const char* X;  int ldx;  int N;                           // grouped triple-rows image of the inputs [ceil16(N)][ldx] (ldx elements, % 32 == 0)
    const char* W;  long long w_sample_bytes;  int ldw;  int KT;   // grouped triple-rows image of W1 [S_total][H][ldw]; KT = ldw / 32
    const float* b;  const float* W2;  const float* b2;  int C;  int H;
    const int* sidx;  int S;  int NT;  float out_scale;        // out_scale = 2^-(e_x + e_w)
    float* P;  uint32_t* mask;  float* dact;  int out_kind;
    const rbnn_dev_scale* x_ds;                                // != NULL: out_scale *= x_ds->inv_scale
    // fc2: layer 1 (!LAYER2) writes the hidden activations as a per-sample STAGE-major triple image [S][H/32][N][3 pieces][32 units]
    // (value * hid_scale = p0 + p1 + p2); layer 2 reads it as its X operand (x_sample_bytes = N * H * 6)
    long long x_sample_bytes;  char* hid;  float hid_scale;  const rbnn_dev_scale* hid_ds;
    // X image geometry (grouped rows: 3-KiB blocks [3 pieces][16 rows][64 B] per (16-row group, stage)): bytes between consecutive row groups
    // of one stage / between consecutive stages of one group.  rbnn_triple_rows_grouped images: KT * 3072 and 3072.  The hidden image of fc2
    // is stage-major ([S][H/32 stages][N/16 groups][3 KiB]): 3072 and ceil(N/16) * 3072
    unsigned x_group_bytes, x_stage_bytes;
};

template <int ACT, int WH, int HTW, int WN, int NTW, bool LAYER2, bool XF32 = false>      // XF32: the X operand is fc2's fp32 hidden image (layer 2 of fc2 only)
__global__ void __launch_bounds__(64 * WH * WN, WH * WN / 4) fc_forward_x3_kernel(const FwdX3Args a) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int PLANEB = (BH + BN) * 64;                     // one plane of a stage tile
    constexpr int TILEB = 3 * PLANEB;
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);
    constexpr int NW = WH * WN;
    constexpr int WPP = BH / 16 / NW, XPP = BN / 16 / NW;     // DMA pieces (16 rows of one plane) per wave per plane
    static_assert((BH / 16) % NW == 0 && (BN / 16) % NW == 0, "whole pieces per wave");
    static_assert(HTW % 2 == 0 && HTW <= 8, "a wave's h range is whole 32-bit mask words, at most 4");
    static_assert(WH * BN * 64 <= 2 * TILEB, "the Z^T reduction scratch aliases the tile buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * TILEB bytes
    char* const ldsb = (char*)lds;
    float* const zred = lds;

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.S, id)) return;
    // 2-D blocked item order: the ~32 blocks resident on an XCD together are SB samples x (32 / SB) point tiles.  A block streams its
    // sample's W1 image (H * ldw * 6 B) and its X tile (BN * ldx * 6 B), both shared through that XCD's L2 with the co-resident blocks
    // of the same sample / tile: bytes from beyond the L2 per 32 blocks ~ SB * |W1| + (32 / SB) * |X tile|, minimal at SB = 2..4 here
    // (2.46 MB vs 0.61 MB); the exact kernel's SB = 8 cost 33 % more (profiles/r02t: FETCH_SIZE)
    constexpr int SB = RBNN_X3_FWD_SB;
    int ntile, s;
    {
        const int full = a.S / SB, per = SB * a.NT;
        if (id < full * per) { ntile = (id % per) / SB; s = (id / per) * SB + id % SB; }
        else { const int rem = id - full * per, cnt = a.S - full * SB; ntile = rem / cnt; s = full * SB + rem % cnt; }
    }
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations (M0) become SALU work
    const int wave_h = wave % WH, wave_n = wave / WH;
    const int sw = a.sidx ? a.sidx[s] : s;
    const float out_scale = a.x_ds ? a.out_scale * a.x_ds->inv_scale : a.out_scale;
    const float hid_scale = (!LAYER2 && a.hid_ds) ? a.hid_ds->scale : a.hid_scale;
    const char* const Ws = a.W + (long long)sw * a.w_sample_bytes;
    const char* const Xs = a.X + (long long)s * a.x_sample_bytes;
    const int n0 = ntile * BN;
    const int HW = a.H >> 5;
    // DMA piece = 16 rows x 64 B of one plane: lane p lands at row (p >> 2), physical chunk p & 3, so it fetches logical chunk
    // (p & 3) ^ swz(row); pieces start at multiples of 16 rows, so swz(row) = swz(p >> 2)
    const int prow = lane >> 2;
    const unsigned src_off = (unsigned)(prow * 64 + ((lane & 3) ^ swz(prow)) * 16);   // inside a 1-KiB piece of a grouped image: row prow, logical chunk
    // fragment read of row li (any 16-row tile), K chunk lg
    const int foff = li * 64 + ((lg ^ swz(li)) * 16);
    // XF32: the X operand is the fp32 hidden image — a 16-point group of a stage is [16 points][32 units] floats = two 1-KiB DMA pieces (rows 0-7,
    // 8-15; lane p lands at row p >> 3, physical 16-byte chunk p & 7 and fetches logical chunk (p & 7) ^ sw8(row)); lane (li, lg) of a fragment
    // reads logical chunks 2 lg and 2 lg + 1 of row li.  sw8(r) = bit 1 of r | bit 2 of r << 2 makes both reads conflict-free over all four 16-lane
    // service groups of ds_read_b128 (enumerated: 192 of the 512 GF(2)-linear maps of r & 7 do; it must not depend on bit 3, the piece).
    auto sw8 = [](int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); };
    const unsigned src_off_f = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ sw8(lane >> 3)) * 16));
    const int foff_f0 = li * 128 + (((2 * lg) ^ sw8(li)) * 16), foff_f1 = li * 128 + (((2 * lg + 1) ^ sw8(li)) * 16);

    f32x4 zacc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) zacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Static issue priority for the younger half of an 8-wave block: a SIMD issues from its older wave first, so waves 4-7 lose every arbitration
    // against their SIMD partners 0-3 (MI355X_MICROARCH.md, two waves per SIMD, item 4).  One s_setprio for the whole kernel; same-box A/B
    // (profiles/r06k/fwd_prio_ab.txt): forward 3.276 / 3.293 -> 3.260 / 3.276 ms at C2; priority for waves 0-3 instead: no change.
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // per-lane offsets (bytes) of this wave's 16-row groups; 32-bit (the host checks H*ldw*6 and N*ldx*6 < 2^32).  W groups are relative to
    // the h chunk: the chunk's base goes into the uniform part of the address.  Point groups past N repeat the last one (never stored).
    unsigned xrow[XPP], wrow[WPP];
#pragma unroll
    for (int i = 0; i < XPP; ++i) xrow[i] = (unsigned)min((n0 >> 4) + wave + NW * i, ((a.N + 15) >> 4) - 1) * a.x_group_bytes + (XF32 ? src_off_f : src_off);
#pragma unroll
    for (int i = 0; i < WPP; ++i) wrow[i] = (unsigned)(wave + NW * i) * (unsigned)a.KT * 3072u + src_off;

    // The K stages of all h chunks form ONE software pipeline: stage g = (chunk g / KT, columns 32 * (g % KT)), buffer g & 1; the
    // first stage of the next chunk is in flight while a chunk's epilogue runs.  The pieces of stage g + 1 (PPS per wave) are
    // issued BETWEEN stage g's MFMA groups, not at its top next to the LDS reads (an LDS-DMA instruction costs ~60 cycles among
    // MFMAs against 100-185 there), all within the first two h tiles so that they land before the stage ends.
    // A "piece" below is one 16-row GROUP of one stage: three 1-KiB LDS-DMA instructions (the three fp16 pieces) that share their address
    // register and M0 — a group is [3 pieces][1 KiB] both in the image and in the stage tile.
    constexpr int PPS = WPP + XPP;
    const int G = (a.H / BH) * a.KT;
    auto piece = [&](int c, int kt, int buf, int j) {           // chunk c, columns 32 * kt -> buffer buf; j: compile-time constant
        char* const T = ldsb + buf * TILEB;
        const char* src;
        char* dst;
        if (j < WPP) {
            src = Ws + (long long)c * BH * a.ldw * 6 + (wrow[j < WPP ? j : 0] + (unsigned)kt * 3072u);
            dst = T + (wave + NW * j) * 3072;
        } else {
            src = Xs + (xrow[j >= WPP ? j - WPP : 0] + (unsigned)kt * a.x_stage_bytes);
            dst = T + (BH / 16 + wave + NW * (j - WPP)) * 3072;
        }
        const auto gsrc = (const __attribute__((address_space(1))) void*)src;
        const auto ldst = (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)dst;
        __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 1024, 0);
        if (!(XF32 && j >= WPP)) __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 2048, 0);      // (an fp32 X group is 2 KiB: two pieces)
    };
#pragma unroll
    for (int i = 0; i < PPS; ++i) piece(0, 0, 0, i);
    ring_wait_barrier<0>();

    f32x4 acc[HTW][NTW];
#pragma unroll
    for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kt = 0, ch = 0;                                        // stage g = (chunk ch, K stage kt)
    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        // next stage; the very last one re-fetches itself into the idle buffer (1 stage in G: keeps the body branch-free)
        const bool wrap = kt + 1 == a.KT, last = g + 1 == G;
        const int ktn = last ? kt : (wrap ? 0 : kt + 1), chn = (wrap && !last) ? ch + 1 : ch;
        const char* const Wt = ldsb + buf * TILEB + (wave_h * HTW) * 3072 + foff;
        const char* const Xt = ldsb + buf * TILEB + (BH / 16 + wave_n * NTW) * 3072 + foff;
        f16x8 b0[NTW], b1[NTW], b2[NTW], a0, a1, a2, a0n, a1n, a2n;
        if (!(RBNN_ABL & 1)) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) piece(chn, ktn, buf ^ 1, i);
        }
        if constexpr (XF32) {
            // fp32 hidden activations (already x 2^h1_exp) -> the three pieces of the B fragment, in registers: 8 values, four split3 pairs
            const char* const Xf = ldsb + buf * TILEB + (BH / 16 + wave_n * NTW) * 3072;
            f32x4 x0[NTW], x1[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                x0[nt] = *(const f32x4*)(Xf + nt * 3072 + foff_f0);
                x1[nt] = *(const f32x4*)(Xf + nt * 3072 + foff_f1);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                union { f16x8 v; unsigned w[4]; } q0, q1, q2;
                split3_plain_pair(x0[nt][0], x0[nt][1], 1.f, q0.w[0], q1.w[0], q2.w[0]);
                split3_plain_pair(x0[nt][2], x0[nt][3], 1.f, q0.w[1], q1.w[1], q2.w[1]);
                split3_plain_pair(x1[nt][0], x1[nt][1], 1.f, q0.w[2], q1.w[2], q2.w[2]);
                split3_plain_pair(x1[nt][2], x1[nt][3], 1.f, q0.w[3], q1.w[3], q2.w[3]);
                b0[nt] = q0.v; b1[nt] = q1.v; b2[nt] = q2.v;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                b0[nt] = *(const f16x8*)(Xt + nt * 3072);
                b1[nt] = *(const f16x8*)(Xt + nt * 3072 + 1024);
                b2[nt] = *(const f16x8*)(Xt + nt * 3072 + 2048);
            }
        }
        a0 = *(const f16x8*)(Wt);
        a1 = *(const f16x8*)(Wt + 1024);
        a2 = *(const f16x8*)(Wt + 2048);
        a0n = a0; a1n = a1; a2n = a2;
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht) {
            // six product groups of NTW MFMAs, smallest terms first; between them (fenced: the scheduler would otherwise cluster all
            // memory instructions at the top) one DMA piece at a time, and the next h tile's three fragment reads after the third
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const f16x8 av = (k == 1) ? a2 : ((k == 2 || k == 3) ? a1 : a0);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const f16x8 bv = (k == 0) ? b2[nt] : ((k == 2 || k == 4) ? b1[nt] : b0[nt]);
                    acc[ht][nt] = MFMA_H(av, bv, acc[ht][nt]);
                }
                if (k == 0 && ht + 1 < HTW) {
                    a0n = *(const f16x8*)(Wt + (ht + 1) * 3072);
                    a1n = *(const f16x8*)(Wt + (ht + 1) * 3072 + 1024);
                    a2n = *(const f16x8*)(Wt + (ht + 1) * 3072 + 2048);
                }
            }
            a0 = a0n; a1 = a1n; a2 = a2n;
        }
        {
            // pin the order: B fragments + A(0) first, then per h tile half its MFMAs, the next tile's three reads, the rest
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * NTW + 3, 0);
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTW, 0);
                if (ht + 1 < HTW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTW, 0);
            }
        }
        if (RBNN_X3_BARRIER_END) __builtin_amdgcn_sched_barrier(0);   // keep the hand-off behind the stage's last MFMA
        if (!(RBNN_ABL & 2)) ring_wait_barrier<0>();           // stage g+1 landed; everyone is done with stage g
        const int hc0 = ch * BH;
        kt = wrap ? 0 : kt + 1;
        ch = wrap ? ch + 1 : ch;
        if (!wrap) continue;

        // ---- epilogue of this h chunk: scale, bias, activation, derivative stash, skinny output layer ----
        if (RBNN_ABL & 8) {                                    // diagnostic: keep the accumulators live, skip the epilogue
            float t = 0.f;
#pragma unroll
            for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) { t += acc[ht][nt][0] + acc[ht][nt][1] + acc[ht][nt][2] + acc[ht][nt][3]; acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if (t == 12345.678f) a.P[tid] = t;
            continue;
        }
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
                acc[ht][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
                } else {
                    // stage-major fp32 image [S][H/32 stages][N/16 groups][16 points][32 units]: this lane's four units of tile ht are 16 bytes at
                    // byte ((hrow >> 4) & 1) * 64 + lg * 16 of its point's 128-byte row; a tile's 16 points are one 2-KiB block, and the two h tiles of a
                    // stage fill its lines between them — no LDS staging, no split here (layer 2 splits at its operand read)
#ifdef RBNN_X3_L1_ABL_NOSTORE
                    if (n < a.N && hv[0] == 1.2345e-30f) {                 // ablation (timing only): never stored
#else
                    if (n < a.N) {
#endif
                        const long long blk = ((long long)s * HW + (hrow >> 5)) * ((a.N + 15) >> 4) + (n >> 4);
#ifdef RBNN_X3_L1_ABL_SMALL
                        char* const row = a.hid + ((blk * 2048) & 0xFFC00) + (n & 15) * 128 + ((hrow >> 4) & 1) * 64 + lg * 16;   // ablation: all stores into 1 MB
#else
                        char* const row = a.hid + blk * 2048 + (n & 15) * 128 + ((hrow >> 4) & 1) * 64 + lg * 16;
#endif
                        *(f32x4*)row = hv * hid_scale;
                    }
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
    if (RBNN_ABL & 2) ring_wait_barrier<0>();

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

template <int ACT, int WH, int HTW, int WN, int NTW, bool LAYER2, bool XF32>
int launch_forward_x3_cfg(FwdX3Args a, hipStream_t st) {
    constexpr int BH = WH * HTW * 16, BN = WN * NTW * 16;
    constexpr int LDSB = 2 * 3 * (BH + BN) * 64;
    static_assert(LDSB <= 160 * 1024, "LDS");
    a.NT = (a.N + BN - 1) / BN;
    auto kern = fc_forward_x3_kernel<ACT, WH, HTW, WN, NTW, LAYER2, XF32>;
    static unsigned long long attr_done = 0;                    // per instantiation, one bit per device
    if (!ensure_dynamic_lds((const void*)kern, LDSB, attr_done)) return RBNN_ERR_LAUNCH;
    const int grid = grid_for_items((long long)a.NT * a.S);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WH * WN), LDSB, st, a);
    return launch_status();
}

#ifndef RBNN_X3_FWD_CFG
#define RBNN_X3_FWD_CFG 4, 4, 2, 4                              // 256 h x 128 n, 8 waves of 64 h x 64 n, 144 KB of LDS
#endif

template <int ACT, bool LAYER2, bool XF32>
int launch_forward_x3_act(const FwdX3Args& a, hipStream_t st) {
    if (a.H % 256 == 0) return launch_forward_x3_cfg<ACT, RBNN_X3_FWD_CFG, LAYER2, XF32>(a, st);
    if (a.H % 128 == 0) return launch_forward_x3_cfg<ACT, 2, 4, 2, 4, LAYER2, XF32>(a, st);   // 128 h x 128 n, 4 waves
    return RBNN_ERR_UNSUPPORTED;
}

template <bool LAYER2, bool XF32 = false>
int launch_forward_x3(int act, const FwdX3Args& a, hipStream_t st) {
    switch (act) {
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_RELU:  return launch_forward_x3_act<RBNN_ACT_RELU, LAYER2, XF32>(a, st);
#endif
        case RBNN_ACT_LEAKY: return launch_forward_x3_act<RBNN_ACT_LEAKY, LAYER2, XF32>(a, st);
#ifndef RBNN_FAST_BUILD
        case RBNN_ACT_SIGM:  return launch_forward_x3_act<RBNN_ACT_SIGM, LAYER2, XF32>(a, st);
        case RBNN_ACT_TANH:  return launch_forward_x3_act<RBNN_ACT_TANH, LAYER2, XF32>(a, st);
#endif
    }
    return RBNN_ERR_UNSUPPORTED;
}

// ===================================================================================================
// Image builders of the backward.
// ===================================================================================================
// W1 "triple cols" image: out[m][hb][lg][p][d][j] (p = piece 0..2; 8 halves j) = piece p of W[m][32*hb + 16*(j>>2) + 4*lg + (j&3)][d] * scale.
__global__ void triple_cols_kernel(const float* __restrict__ W, long long n_mats, int rows, int cols, int ld_src, float scale,
                                   uint4* __restrict__ dst, int ld_dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HB = rows / 32;
    if (i >= n_mats * HB * 4 * ld_dst) return;
    const int d = (int)(i % ld_dst);
    const int lg = (int)((i / ld_dst) % 4);
    const long long mh = i / (4LL * ld_dst);                   // m * HB + hb
    const long long m = mh / HB;
    const int hb = (int)(mh % HB);
    union { f16x8 v; uint4 u; } o[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int h = 32 * hb + 16 * (j >> 2) + 4 * lg + (j & 3);
        const float v = (d < cols) ? W[(m * rows + h) * ld_src + d] * scale : 0.f;
        _Float16 a, b, c;
        split3(v, a, b, c);
        o[0].v[j] = a; o[1].v[j] = b; o[2].v[j] = c;
    }
    const long long base = ((mh * 4 + lg) * 3) * ld_dst;       // 16-byte units
    dst[base + d] = o[0].u;
    dst[base + ld_dst + d] = o[1].u;
    dst[base + 2 * ld_dst + d] = o[2].u;
}

// K-slot plan of the dA generator: TWO f16 MFMAs (64 K slots) chained on one accumulator form all six products of the C <= 10
// classes.  Classes 0..7 fill whole 8-slot chunks, classes 8, 9 share a "tail" chunk.  With w_p / d_p = piece p of W2[c][h] / dZ[n][c]:
//   dZ row of a point (64 B):  chunk 0 = d0[c0..7]   chunk 1 = d1[c0..7]   chunk 2 = d2[c0..7]   chunk 3 = T = [d0c8 d0c9 d1c8 d1c9 d2c8 d2c9 d0c8 d0c9]
//   MFMA 1, lane group lg:     dZ chunks {0, 1, 0, 3}      W2 side { w0[c0..7], w0[c0..7], w1[c0..7], [w0c8 w0c9 w0c8 w0c9 w0c8 w0c9 w1c8 w1c9] }
//   MFMA 2, lane group lg:     dZ chunks {1, 2, 0, 3}      W2 side { w1[c0..7], w0[c0..7], w2[c0..7], [w2c8 w2c9 w1c8 w1c9 0 0 0 0] }
// => (w0 d0) (w0 d1) (w1 d0) | (w1 d1) (w0 d2) (w2 d0) for c < 8, and the same six for c = 8, 9 out of the tail chunks.
// W2 generator image: out[m][t][k][lane][8] = W2 side of MFMA k+1 for hidden unit 16*t + li: 2 KiB per 16-unit tile.
__global__ void triple_w2gen_kernel(const float* __restrict__ W2, int n_mats, int C, int H, float scale, uint4* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_mats * (H / 16) * 128) return;
    const int lane = (int)(i & 63), li = lane & 15, lg = lane >> 4, k = (int)((i >> 6) & 1);
    const int t = (int)((i >> 7) % (H / 16));
    const long long m = (i >> 7) / (H / 16);
    _Float16 w[3][10];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const float v = (c < C) ? W2[(m * C + c) * H + 16 * t + li] * scale : 0.f;
        split3(v, w[0][c], w[1][c], w[2][c]);
    }
    union { f16x8 v; uint4 u; } o;
    const _Float16 z = (_Float16)0.f;
    if (lg < 3) {
        const int piece = (k == 0) ? (lg == 2 ? 1 : 0) : (lg == 0 ? 1 : (lg == 1 ? 0 : 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = piece == 0 ? w[0][j] : (piece == 1 ? w[1][j] : w[2][j]);
    } else if (k == 0) {
        o.v[0] = w[0][8]; o.v[1] = w[0][9]; o.v[2] = w[0][8]; o.v[3] = w[0][9];
        o.v[4] = w[0][8]; o.v[5] = w[0][9]; o.v[6] = w[1][8]; o.v[7] = w[1][9];
    } else {
        o.v[0] = w[2][8]; o.v[1] = w[2][9]; o.v[2] = w[1][8]; o.v[3] = w[1][9];
        o.v[4] = z; o.v[5] = z; o.v[6] = z; o.v[7] = z;
    }
    dst[i] = o.u;
}

// dZ generator image + per-point scale: e(n) = 13 - ilogb(max_{s,c} |dZ[s][n][c]|); out[s][n][chunk ^ dz_swz3(n)] per the plan above;
// gscale[n] = 2^-e(n).  Points n >= N get zeros.  (Block = 16 points x 16 sample lanes, as split_dz_kernel.)
__host__ __device__ __forceinline__ int dz_swz3(long long n) { return (int)((0 - (n >> 2)) & 3); }

__global__ void __launch_bounds__(256) triple_dz_kernel(const float* __restrict__ dZ, int S, int N, long long N_pad, int C,
                                                        uint4* __restrict__ dst, float* __restrict__ gscale) {
    __shared__ float red[16][17];
    const int p = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long n = (long long)blockIdx.x * 16 + p;
    float m = 0.f;
    if (n < N)
        for (int s = q; s < S; s += 16) {
            const float* const src = dZ + ((long long)s * N + n) * RBNN_CPAD;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
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
    const int sw = dz_swz3(n);
    for (int s = q; s < S; s += 16) {
        _Float16 d[3][10];
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
            split3(x, d[0][c], d[1][c], d[2][c]);
        }
        uint4* const o = dst + ((long long)s * N_pad + n) * 4;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            union { f16x8 v; uint4 u; } w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w.v[j] = d[ch][j];
            o[ch ^ sw] = w.u;
        }
        union { f16x8 v; uint4 u; } t;
        t.v[0] = d[0][8]; t.v[1] = d[0][9]; t.v[2] = d[1][8]; t.v[3] = d[1][9];
        t.v[4] = d[2][8]; t.v[5] = d[2][9]; t.v[6] = d[0][8]; t.v[7] = d[0][9];
        o[3 ^ sw] = t.u;
    }
}

// ===================================================================================================
// The tail of a step between the two GEMM kernels, fused (round 4): rbnn_reduce_samples + rbnn_loss_dlogits + triple_dz_kernel in ONE launch.
// P [S][N][16] is read by one block per 16 points (its second and third pass hit the L2), the fp32 dZ [S][N][16] is never written:
//   A  Psum[n][:] = sum_s P[s][n][:]             — sequentially in s by one thread per (point, class quad): the order of reduce_samples_kernel
//   B  g = dL/d(what the loss saw)               — loss_dlogits_kernel's formulas, once per thread (not per sample) for the mean losses
//   C  e(n) = 13 - ilogb(max_{s,c} |dZ[s][n][c]|) — dZ recomputed per sample from P (a dozen flops), as in triple_dz_kernel
//   D  dZ again, x 2^e(n), three pieces, the generator image; gscale[n] = 2^-e(n)
// Every operation and its order are those of the three kernels it replaces: the image is BIT-IDENTICAL (tests/test_hip_round4.py).
// ===================================================================================================
template <int MODE>
__device__ __forceinline__ void tail_dz(const float* __restrict__ prow, const float (&gmean)[16], int y, int C, float inv_S, float (&out)[16]) {
    float p[16], g[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(prow + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * q + r] = v[r];
    }
    if (MODE == RBNN_LOSS_PER_SAMPLE) {
        ce_softmax_grad<16>(p, C, y, inv_S, g);                // loss_dlogits_kernel's, bit for bit
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) g[c] = gmean[c];
    }
    if (MODE == RBNN_LOSS_MEAN_LOGIT) {
#pragma unroll
        for (int c = 0; c < 16; ++c) out[c] = g[c];
    } else {
        softmax_backward<16>(g, p, C, out);                    // loss_dlogits_kernel's, bit for bit
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) step_tail_x3_kernel(const float* __restrict__ P, const int* __restrict__ labels, int S, float inv_S, int N,
                                                           long long N_pad, int C, float* __restrict__ Psum_out, int ldo,
                                                           uint4* __restrict__ dst, float* __restrict__ gscale) {
    __shared__ float red[16][17];
    __shared__ float psum[16][16];
    const int p = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long n = (long long)blockIdx.x * 16 + p;
    if (MODE != RBNN_LOSS_PER_SAMPLE) {
        if (q < 4) {
            f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (n < N)
                for (int s = 0; s < S; ++s) sum += *(const f32x4*)(P + ((long long)s * N + n) * RBNN_CPAD + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                psum[p][4 * q + r] = sum[r];
                if (Psum_out && n < N && 4 * q + r < C) Psum_out[n * ldo + 4 * q + r] = sum[r];
            }
        }
        __syncthreads();
    }
    const int y = (n < N) ? labels[n] : 0;
    float gmean[16];
    if (MODE != RBNN_LOSS_PER_SAMPLE) {
        float t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = (c < C) ? psum[p][c] * inv_S : 0.f;
        ce_softmax_grad<16>(t, C, y, inv_S, gmean);
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) gmean[c] = 0.f;
    }
    float m = 0.f;
    if (n < N)
        for (int s = q; s < S; s += 16) {
            float dz[16];
            tail_dz<MODE>(P + ((long long)s * N + n) * RBNN_CPAD, gmean, y, C, inv_S, dz);
#pragma unroll
            for (int c = 0; c < 12; ++c) if (c < C) m = fmaxf(m, fabsf(dz[c]));
        }
    red[q][p] = m;
    __syncthreads();
    m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i][p]);
    int e = 0;
    if (m > 0.f && m < INFINITY) e = min(13 - ilogbf(m), 120);
    if (q == 0) gscale[n] = ldexpf(1.f, -e);
    const int sw = dz_swz3(n);
    for (int s = q; s < S; s += 16) {
        _Float16 d[3][10];
        float dz[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) dz[c] = 0.f;
        if (n < N) tail_dz<MODE>(P + ((long long)s * N + n) * RBNN_CPAD, gmean, y, C, inv_S, dz);
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const float x = (n < N && c < C) ? ldexpf(dz[c], e) : 0.f;
            split3(x, d[0][c], d[1][c], d[2][c]);
        }
        uint4* const o = dst + ((long long)s * N_pad + n) * 4;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            union { f16x8 v; uint4 u; } w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w.v[j] = d[ch][j];
            o[ch ^ sw] = w.u;
        }
        union { f16x8 v; uint4 u; } t;
        t.v[0] = d[0][8]; t.v[1] = d[0][9]; t.v[2] = d[1][8]; t.v[3] = d[1][9];
        t.v[4] = d[2][8]; t.v[5] = d[2][9]; t.v[6] = d[0][8]; t.v[7] = d[0][9];
        o[3 ^ sw] = t.u;
    }
}

// rbnn_attack_step + rbnn_triple_rows_grouped of the NEW iterate in one pass (a PGD loop then needs no image-builder launch per iteration):
// one thread per (row, group of 8 columns) — the step in attack_step4_kernel's operation order (bit-identical X), then the three pieces of
// the eight new values (bit-identical image).  Groups beyond D hold the image's zero padding.
__global__ void __launch_bounds__(256) attack_step_x3_kernel(float* __restrict__ X, const float* __restrict__ X0, int ldx, const float* __restrict__ G,
                                                             int K, long long slab_stride, int ldg, const float* __restrict__ alpha, float alpha_scalar,
                                                             float eps, int project, long long N, int D, const rbnn_dev_scale* __restrict__ ds,
                                                             uint4* __restrict__ dst, int groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * groups) return;
    const long long n = i / groups;
    const int gq = (int)(i % groups), d0 = 8 * gq;
    const float scale = ds->scale;
    float xn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xn[j] = 0.f;
    if (d0 < D) {                                                          // D, ldx, ldg are multiples of 4: whole float4s
        const float step = alpha ? alpha[n] : alpha_scalar;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = d0 + 4 * h;
            if (d < D) {
                f32x4 g = *(const f32x4*)(G + n * ldg + d);
                for (int k = 1; k < K; ++k) g += *(const f32x4*)(G + k * slab_stride + n * ldg + d);
                const f32x4 x = *(const f32x4*)(X + n * ldx + d);
                f32x4 x0 = x, out;
                if (project) x0 = *(const f32x4*)(X0 + n * ldx + d);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sgn = (g[r] > 0.f) ? 1.f : ((g[r] < 0.f) ? -1.f : 0.f);
                    float pert = x[r] + step * sgn;
                    if (project) pert = x0[r] + fminf(fmaxf(pert - x0[r], -eps), eps);
                    out[r] = fminf(fmaxf(pert, 0.f), 1.f);
                    xn[4 * h + r] = out[r];
                }
                *(f32x4*)(X + n * ldx + d) = out;
            }
        }
    }
    union { f16x8 v; uint4 u; } o[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (d0 + j < D) ? xn[j] * scale : 0.f;
        _Float16 a, b, c;
        split3(v, a, b, c);
        o[0].v[j] = a; o[1].v[j] = b; o[2].v[j] = c;
    }
    uint4* const out = dst + (((n >> 4) * (groups >> 2) + (gq >> 2)) * 192 + (n & 15) * 4 + (gq & 3));
    out[0] = o[0].u; out[64] = o[1].u; out[128] = o[2].u;
}

// ===================================================================================================
// T2: input gradient.  One block = one (256-point tile, TD*16-column group, chunk of samples) item, 4 waves of 64 points:
//   acc[n][d] += sum_h dA[n][h] * W1[s][h][d],   dA[n][h] = act'(A_s[n][h]) * sum_c dZ[s][n][c] * W2[s][c][h]
// A stage is 32 hidden units of one sample = ONE K step of the f16 MFMA:
//   generator  (dA^T)[h][n] for the stage's two 16-unit tiles: two chained f16 MFMAs per (tile, point tile), fp32 result;
//   split      x act' (1-bit stash) x 2^GEN_Q on the VALU, then the three halves: the accumulator layout IS the A-operand layout
//              of the main MFMA with K slot 8*lg + 4*t + r (the order of the triple-cols image);
//   main       6 f16 MFMAs per (point tile, column tile), B operand (W1 pieces) from the stage's LDS tile.
// The stage's W1 tile, W2 generator tiles and stash words arrive by LDS-DMA one stage ahead.  The dZ generator image of a
// wave's own 64 points (4 KiB per sample) lives in a SINGLE buffer: the wave itself re-fills it for the next sample during
// the last stage of the current one, after its generator reads (two blocks per CU need <= 80 KB each).
// ===================================================================================================
#define GEN_Q3 (-17)                                           // |generator| <= 16 * 2^14 * 2^14 = 2^32  ->  |dA| <= 2^15 < fp16 max

struct GradX3Args {
    const char* dzg;  long long n_pad;  const float* gscale;  const uint32_t* mask;
    const char* W1c;  int ldc;                                  // triple-cols image, ldc columns
    const char* W2g;                                            // generator image [S_total][H/16][2 KiB]
    int H;  int HW;  const int* sidx;  int S;  int chunk;  int nchunks;
    int N;  int NT;  int ND;  int Dt;
    float* out;  int ldo;  float out_scale;                     // slabs [nchunks][N][ldo]; out_scale = 2^-(e_w2 + GEN_Q3 + e_w1)
    // fc2.  X3_FC2_STEP1 (one sample per block): out = dhid1 [S][N][H] = act'(A1) * (dA2 . Wm), KEPT SCALED (x out_scale, no per-point
    // un-scaling): it is the fp32 source of step 2's A operand.  X3_FC2_STEP2: A operand read from `amem` and split in registers.
    const uint32_t* omask;  int OHW;                            // step 1: stash of the layer below [S][H/32][N_pad]
    const float* amem;                                          // step 2: [S][N][H]
    const float* dact;  const float* odact;                     // sigmoid / tanh: act' as fp32 [S][N][H] (this layer / the layer below)
};
enum { X3_FC = 0, X3_FC2_STEP1 = 1, X3_FC2_STEP2 = 2 };

template <int ACT, int TD, int MODE>
__global__ void __launch_bounds__(256, 2) fc_grad_x3_kernel(const GradX3Args a) {
    constexpr int NW = 4;                                      // 4 waves x (64 points x TD*16 columns), two blocks per CU
    constexpr bool GEN = MODE != X3_FC2_STEP2;                 // dA generated from dZ, or read from memory
    constexpr bool BITMASK = (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY);   // act' from the 1-bit stash, or an fp32 stream (sigmoid / tanh)
    constexpr bool STREAM = !GEN || !BITMASK;                  // a per-lane fp32 operand (A itself, or act') is prefetched from memory
    constexpr int NTW = 16 / NW, BM = 256, LD = TD * 16;
    constexpr int W1B = 12 * LD * 16;                          // bytes: [4 lg][3 pieces][LD columns][16 B]
    constexpr int NPIECE = W1B / 1024, PPW = (NPIECE + NW - 1) / NW;
    constexpr int BUFB = W1B + 4096 + 1024;                    // + 2 generator tiles of 2 KiB + 256 stash words
    constexpr int DZB = BM * 64;
    static_assert(W1B % 1024 == 0, "whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * BUFB + DZB bytes
    char* const ldsb = (char*)lds;
    char* const dzl = ldsb + 2 * BUFB;

    int id;
    if (!item_of_block(blockIdx.x, a.NT * a.ND * a.nchunks, id)) return;
    int ntile, dg, ch;
    grad_item(id, a.NT, a.ND, STREAM, ntile, dg, ch);          // (a per-lane fp32 operand — step 2's A, or sigmoid / tanh's act' — is read from memory by every column group: 2-D blocked order, rbnn_common.hpp)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations (M0) and piece selection become SALU work
    const int nb = ntile * BM + wave * (NTW * 16);
    const int dc0 = dg * LD;
    const int Dp = a.Dt * 16;
    const int ntd = min(TD, a.Dt - dg * TD);                   // valid column tiles of this group (the last group may be partial)
    const int s_begin = ch * a.chunk, s_end = min(a.S, s_begin + a.chunk);
    const int HS = a.H / 32, nst = (s_end - s_begin) * HS;

    // LDS-DMA sources are a block-uniform 64-bit base (SGPR pair) + one 32-bit per-lane offset: no vector address arithmetic per piece
    unsigned goff[PPW];                                        // this wave's W1 pieces: byte offsets from the stage's image base
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int f = (wave + NW * i) * 1024 + lane * 16, seg = f / (LD * 16), d = (f % (LD * 16)) >> 4;
        goff[i] = (unsigned)(seg * a.ldc + min(dc0 + d, a.ldc - 1)) * 16u;   // columns past the image: any valid address, never stored
    }
    const unsigned loff = (unsigned)lane * 16u;

    f32x4 acc[NTW][TD];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) acc[nt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto dz_issue = [&](int s) {                                // this wave's 64 points x 64 B of sample s -> its own region of dzl
        const char* const src = a.dzg + ((long long)s * a.n_pad + nb) * 64;
#pragma unroll
        for (int j = 0; j < NTW; ++j) glds16((const float*)(src + j * 1024 + loff), (float*)(dzl + wave * (NTW * 1024) + j * 1024));
    };
    // The DMA pieces of a stage, per wave: 0 .. PPW-1 its W1 pieces, PPW its generator-tile piece (tiles 2*hb, 2*hb + 1 of the sample
    // are 4 KiB contiguous: one piece per wave), PPW + 1 the stash words (the block's 256 points = 1 KiB; last wave only).
    constexpr int NDMA = GEN ? (BITMASK ? PPW + 2 : PPW + 1) : PPW;
    struct StageSrc { const char* W; const char* G; const char* M; char* B; };
    auto stage_src = [&](int st, int buf) {
        const int si = st / HS, hb = st % HS, s = s_begin + si;
        const int sw = a.sidx ? a.sidx[s] : s;
        StageSrc r;
        r.W = a.W1c + ((long long)sw * HS + hb) * 12 * a.ldc * 16;
        r.G = a.W2g + ((long long)sw * (a.H / 16) + 2 * hb) * 2048 + wave * 1024;
        r.M = (const char*)(a.mask + ((long long)s * a.HW + hb) * a.n_pad + ntile * BM);
        r.B = ldsb + buf * BUFB;
        return r;
    };
    auto issue_piece = [&](const StageSrc& q, int i) {          // i is a compile-time constant at every call site
        if (i < PPW) {
            if (wave + NW * i < NPIECE) glds16((const float*)(q.W + goff[i < PPW ? i : 0]), (float*)(q.B + (wave + NW * i) * 1024));
        } else if (i == PPW) {
            glds16((const float*)(q.G + loff), (float*)(q.B + W1B + wave * 1024));   // the two generator tiles are 4 pieces: one per wave
        } else if (BITMASK && i == PPW + 1) {
            if (wave == NW - 1) glds16((const float*)(q.M + loff), (float*)(q.B + W1B + 4096));
        }
    };

    if (GEN) dz_issue(s_begin);
    {
        const StageSrc q = stage_src(0, 0);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_piece(q, i);
    }
    // step 2: the A operand of stage st (this lane: point li of each tile, units 16t + 4lg + r of the stage's 32) is loaded from memory
    // one stage ahead, AFTER the next stage's LDS-DMA has been issued, so that the barrier's vmcnt(0) covers both
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
    const float c_pos = ldexpf(1.f, GEN_Q3);
    const unsigned cp_bits = __float_as_uint(c_pos);
    const unsigned cn_bits = (ACT == RBNN_ACT_RELU) ? 0u : __float_as_uint(LEAKY_SLOPE * ldexpf(1.f, GEN_Q3));
    const int dzc1 = (lg == 2 ? 0 : lg), dzc2 = (lg == 0 ? 1 : (lg == 1 ? 2 : (lg == 2 ? 0 : 3)));   // dZ chunk of MFMA 1 / 2 for this lane group
    const int sz = dz_swz3(li);
    const char* const dzw = dzl + wave * (NTW * 1024) + li * 64;
    f16x8 da0[NTW], da1[NTW], da2[NTW];                        // A operand of the main MFMA: this wave's 4 point tiles, one stage
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1, hb = st % HS;
        // next stage's pieces: issued between the generator's point tiles, where the wave is in a vector-only phase (an LDS-DMA
        // instruction costs 25-60 cycles there against 100-185 at the top of the stage next to the LDS reads) and early enough to land
        const bool more = !(RBNN_ABL & 1) && st + 1 < nst;
        const StageSrc q = stage_src(more ? st + 1 : st, buf ^ 1);
        const char* const B = ldsb + buf * BUFB;
        const bool gen_on = GEN && (!(RBNN_ABL & 16) || st == 0);
        if (!GEN) {                                            // split this stage's A (already in registers), then fetch the next one
#pragma unroll
            for (int nt = 0; nt < (GEN ? 0 : NTW); ++nt) {
                union { f16x8 v; unsigned u[4]; } o0, o1, o2;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    split3_plain_pair(am[nt][t][0], am[nt][t][1], 1.f, o0.u[2 * t], o1.u[2 * t], o2.u[2 * t]);
                    split3_plain_pair(am[nt][t][2], am[nt][t][3], 1.f, o0.u[2 * t + 1], o1.u[2 * t + 1], o2.u[2 * t + 1]);
                }
                da0[nt] = o0.v; da1[nt] = o1.v; da2[nt] = o2.v;
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) issue_piece(q, i);
        }
        f32x4 dm[(GEN && STREAM) ? NTW : 1][2];                // sigmoid / tanh: this stage's act' x 2^GEN_Q3 (copied before the next prefetch)
        if (GEN && STREAM) {
#pragma unroll
            for (int nt = 0; nt < ((GEN && STREAM) ? NTW : 0); ++nt) { dm[nt][0] = am[nt][0] * c_pos; dm[nt][1] = am[nt][1] * c_pos; }
        }
        if (STREAM && st + 1 < nst) load_a(st + 1);

        // ---- generator + split ----
        if (gen_on) {
            const f16x8 w00 = *(const f16x8*)(B + W1B + lane * 16);            // tile 0: MFMA 1, MFMA 2
            const f16x8 w01 = *(const f16x8*)(B + W1B + 1024 + lane * 16);
            const f16x8 w10 = *(const f16x8*)(B + W1B + 2048 + lane * 16);     // tile 1
            const f16x8 w11 = *(const f16x8*)(B + W1B + 3072 + lane * 16);
            const unsigned* const Mk = (const unsigned*)(B + W1B + 4096) + wave * (NTW * 16) + li;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const f16x8 dz1 = *(const f16x8*)(dzw + nt * 1024 + ((dzc1 ^ sz) * 16));
                const f16x8 dz2 = *(const f16x8*)(dzw + nt * 1024 + ((dzc2 ^ sz) * 16));
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 g0 = MFMA_H(w01, dz2, MFMA_H(w00, dz1, z)), g1 = MFMA_H(w11, dz2, MFMA_H(w10, dz1, z));
                // the kernel is bound by vector-instruction ISSUE (profiles/r02t), so the split is hand-scheduled: split3_pair
                union { f16x8 v; unsigned u[4]; } o0, o1, o2;
                if constexpr (BITMASK) {
                    const unsigned mw = Mk[nt * 16] >> (4 * lg);   // bit r: unit 4*lg + r; bit 16 + r: unit 16 + 4*lg + r
                    split3_pair<0, true>(g0[0], g0[1], mw, cp_bits, cn_bits, 1.f, o0.u[0], o1.u[0], o2.u[0]);
                    split3_pair<2, false>(g0[2], g0[3], mw, cp_bits, cn_bits, 1.f, o0.u[1], o1.u[1], o2.u[1]);
                    split3_pair<16, true>(g1[0], g1[1], mw, cp_bits, cn_bits, 1.f, o0.u[2], o1.u[2], o2.u[2]);
                    split3_pair<18, false>(g1[2], g1[3], mw, cp_bits, cn_bits, 1.f, o0.u[3], o1.u[3], o2.u[3]);
                } else {
                    constexpr int q = (GEN && STREAM) ? 1 : 0;     // (dm has one dummy entry in the other instantiations)
                    split3_pair_m<true>(g0[0], g0[1], dm[q * nt][0][0], dm[q * nt][0][1], 1.f, o0.u[0], o1.u[0], o2.u[0]);
                    split3_pair_m<false>(g0[2], g0[3], dm[q * nt][0][2], dm[q * nt][0][3], 1.f, o0.u[1], o1.u[1], o2.u[1]);
                    split3_pair_m<true>(g1[0], g1[1], dm[q * nt][1][0], dm[q * nt][1][1], 1.f, o0.u[2], o1.u[2], o2.u[2]);
                    split3_pair_m<false>(g1[2], g1[3], dm[q * nt][1][2], dm[q * nt][1][3], 1.f, o0.u[3], o1.u[3], o2.u[3]);
                }
                da0[nt] = o0.v; da1[nt] = o1.v; da2[nt] = o2.v;
            }
        }
        if (GEN && !(RBNN_ABL & 1) && hb == HS - 1 && st + 1 < nst) {  // last stage of a sample: the dZ reads above are this wave's last of it
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): those reads have returned
            asm volatile("" ::: "memory");
            dz_issue(s_begin + st / HS + 1);                   // lands under the main MFMAs; the stage barrier's vmcnt(0) covers it
        }
        // ---- main: column-tile major ----
        // the B fragments of column tile dt + 1 are read into the OTHER of two register sets while tile dt's MFMAs run: the set index is a
        // compile-time constant of the unrolled loop (a "next" set copied into the current one cost 36 v_mov_b64 per wave-stage, and vector
        // instructions do not overlap the matrix pipe's time here: profiles/r03a/conv_dense_ablations.txt)
        const char* const Bw = B + (lg * 3 * LD + li) * 16;
        f16x8 bb[2][3];
        bb[0][0] = *(const f16x8*)(Bw); bb[0][1] = *(const f16x8*)(Bw + LD * 16); bb[0][2] = *(const f16x8*)(Bw + 2 * LD * 16);
#pragma unroll
        for (int dt = 0; dt < TD; ++dt) {
            if (dt < ntd) {                                     // block-uniform; the loop stays fully unrolled (acc in registers)
                f16x8 (&bc)[3] = bb[dt & 1];
                if (dt + 1 < TD) {
                    f16x8 (&bn)[3] = bb[(dt + 1) & 1];
                    bn[0] = *(const f16x8*)(Bw + (dt + 1) * 256);
                    bn[1] = *(const f16x8*)(Bw + LD * 16 + (dt + 1) * 256);
                    bn[2] = *(const f16x8*)(Bw + 2 * LD * 16 + (dt + 1) * 256);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da0[nt], bc[2], acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da2[nt], bc[0], acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da1[nt], bc[1], acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da1[nt], bc[0], acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da0[nt], bc[1], acc[nt][dt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt][dt] = MFMA_H(da0[nt], bc[0], acc[nt][dt]);
            }
        }
        if (!(RBNN_ABL & 2)) ring_wait_barrier<0>();           // next stage landed; everyone is done with this one
    }
    if (RBNN_ABL & 2) ring_wait_barrier<0>();

    if (RBNN_ABL & 8) {                                        // diagnostic: keep the accumulators live, skip the epilogue
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int dt = 0; dt < TD; ++dt) t += acc[nt][dt][0] + acc[nt][dt][1] + acc[nt][dt][2] + acc[nt][dt][3];
        if (t == 12345.678f) a.out[tid] = t;
        return;
    }
    // ---- epilogue: acc[nt][dt][r] = D[n = nb + nt*16 + 4*lg + r][d = dc0 + dt*16 + li], un-scaled per point ----
#ifndef RBNN_X3_GRAD_EPI_LDS
#define RBNN_X3_GRAD_EPI_LDS 1
#endif
    if constexpr (RBNN_X3_GRAD_EPI_LDS) {
        // (round 5) Straight from the registers a store instruction covered four point rows x 64 BYTES — 128 of them per wave, half-line writes
        // (and, fc2 step 1, a mask-word load per element).  The timing ablation without epilogues priced them at 1.1 of the fc2 backward's
        // 6.2 ms (profiles/r05s).  Now each 16-point tile of a wave goes through a wave-private LDS tile [16 points][LD + 4] (row pitch = 4 mod 16
        // floats: the four row groups of a ds_write_b32 land 16 banks apart) and leaves as 16-BYTE stores of whole rows: a store instruction is
        // 64 x 16 B of at most 2 rows' contiguous runs; the stash word of the layer below is loaded once per 16 bytes.  The loop buffers are free:
        // every wave has passed the last stage's barrier.  Same values, same order of operations per element: bit-identical results.
        constexpr int LDW = LD + 4, C4 = LD / 4, E4 = 16 * C4;
        static_assert(NW * 16 * LDW * 4 <= 2 * BUFB + DZB, "the waves' transposition tiles fit the loop buffers");
        float* const tw = (float*)ldsb + wave * (16 * LDW);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + nt * 16 + 4 * lg + r;
                const float gs = (MODE == X3_FC2_STEP1) ? a.out_scale : (n < a.N ? a.gscale[n] : 0.f) * a.out_scale;
#pragma unroll
                for (int dt = 0; dt < TD; ++dt) tw[(4 * lg + r) * LDW + dt * 16 + li] = acc[nt][dt][r] * gs;
            }
#pragma unroll
            for (int i = 0; i < (E4 + 63) / 64; ++i) {
                const int e4 = i * 64 + lane, row = e4 / C4, c4 = e4 - row * C4;
                const int n = nb + nt * 16 + row, d = dc0 + 4 * c4;
                if (e4 < E4 && n < a.N && d < Dp) {                       // (Dp is a multiple of 16: a quad is in or out as a whole)
                    f32x4 v = *(const f32x4*)(tw + row * LDW + 4 * c4);
                    if (MODE == X3_FC2_STEP1) {                           // derivative of the layer below: units d .. d + 3 of point n, sample ch
                        if (BITMASK) {
                            const unsigned w = a.omask[((long long)ch * a.OHW + (d >> 5)) * a.n_pad + n] >> (d & 31);
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = ((w >> k) & 1u) ? v[k] : (ACT == RBNN_ACT_RELU ? 0.f : v[k] * LEAKY_SLOPE);
                        } else {
                            v *= *(const f32x4*)(a.odact + ((long long)ch * a.N + n) * a.ldo + d);
                        }
                    }
                    *(f32x4*)(a.out + ((long long)ch * a.N + n) * a.ldo + d) = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nb + nt * 16 + 4 * lg + r;
            if (n >= a.N) continue;
            const float gs = (MODE == X3_FC2_STEP1) ? a.out_scale : a.gscale[n] * a.out_scale;
            float* const dst = a.out + ((long long)ch * a.N + n) * a.ldo;
#pragma unroll
            for (int dt = 0; dt < TD; ++dt) {
                const int d = dc0 + dt * 16 + li;
                if (d >= Dp) continue;
                float v = acc[nt][dt][r] * gs;
                if (MODE == X3_FC2_STEP1) {                     // derivative of the layer below: unit d of point n, sample ch
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

template <int ACT, int TD, int MODE>
int launch_grad_x3_cfg(GradX3Args a, hipStream_t st) {
    constexpr int LDSB = 2 * (12 * TD * 16 * 16 + 5120) + 256 * 64;
    static_assert(2 * LDSB <= 160 * 1024, "two blocks per CU");
    a.NT = (a.N + 255) / 256;
    a.ND = (a.Dt + TD - 1) / TD;
    auto kern = fc_grad_x3_kernel<ACT, TD, MODE>;
    static unsigned long long attr_done = 0;
    if (!ensure_dynamic_lds((const void*)kern, LDSB, attr_done)) return RBNN_ERR_LAUNCH;
    const int grid = grid_for_items((long long)a.NT * a.ND * a.nchunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDSB, st, a);
    return launch_status();
}

#ifndef RBNN_X3_GRAD_TD9
#define RBNN_X3_GRAD_TD9 1
#endif
#ifndef RBNN_X3_GRAD_TD8
#define RBNN_X3_GRAD_TD8 1
#endif

inline bool x3_grad_td9() {
    static const bool on = [] { const char* e = getenv("RBNN_X3_GRAD_TD9"); return !e || e[0] != '0'; }();
    return on;
}

// The wider column groups only pay while TWO blocks still share a CU (9 tiles: 2 x 80 KB = every byte of the 160 KB): ask the runtime once
// per (configuration, device) and fall back to 7 tiles where it says one.
template <int ACT, int TD, int MODE>
bool x3_grad_two_blocks() {
    static int ok[64] = {0};                                   // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!ok[dev]) {
        constexpr int LDSB = 2 * (12 * TD * 16 * 16 + 5120) + 256 * 64;
        auto kern = fc_grad_x3_kernel<ACT, TD, MODE>;
        static unsigned long long attr_done = 0;
        int nb = 0;
        const bool fine = ensure_dynamic_lds((const void*)kern, LDSB, attr_done) &&
                          hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, LDSB) == hipSuccess && nb >= 2;
        ok[dev] = fine ? 1 : 2;
    }
    return ok[dev] == 1;
}

// 7 or 4 column tiles per block: every group pays the dA generator (or the A-operand reads) again, so fewer groups win — 7 wherever
// that saves a group (a partial last group skips its missing tiles' MFMAs)
template <int ACT, int MODE>
int launch_grad_x3(const GradX3Args& a, hipStream_t st) {
#if RBNN_X3_GRAD_TD9
    // 9 column tiles per block (two blocks' LDS = exactly 160 KB): 6 column groups instead of 7 at D = 784, i.e. one generator + split
    // pass in seven less: 3.71 / 3.68 -> 3.59 / 3.59 ms at C2 (alternating builds, same box).  Environment RBNN_X3_GRAD_TD9=0 switches
    // back.  (Re-planning the slab size for the 6-group grid — 6 samples per slab instead of 5 — measured slower, 3.75 ms: kept as planned.)
    if constexpr (MODE == X3_FC && (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY)) {   // (fc2 step 2 spills at 9 tiles)
        if (x3_grad_td9() && (a.Dt + 8) / 9 < (a.Dt + 6) / 7 && x3_grad_two_blocks<ACT, 9, MODE>()) return launch_grad_x3_cfg<ACT, 9, MODE>(a, st);
    }
#endif
#if RBNN_X3_GRAD_TD8
    // 8 column tiles per block where that saves a group over 7 (hidden = 512: fc2 step 1 runs 4 groups instead of 5)
    if constexpr (MODE != X3_FC2_STEP2 && (ACT == RBNN_ACT_RELU || ACT == RBNN_ACT_LEAKY)) {   // (the streamed-operand forms spill at 8 tiles)
        if (x3_grad_td9() && (a.Dt + 7) / 8 < (a.Dt + 6) / 7 && x3_grad_two_blocks<ACT, 8, MODE>()) return launch_grad_x3_cfg<ACT, 8, MODE>(a, st);
    }
#endif
    if ((a.Dt + 6) / 7 < (a.Dt + 3) / 4) return launch_grad_x3_cfg<ACT, 7, MODE>(a, st);
    return launch_grad_x3_cfg<ACT, 4, MODE>(a, st);
}

template <int MODE>
int launch_grad_x3_act(int act, const GradX3Args& a, hipStream_t st) {
#ifndef RBNN_FAST_BUILD
    if (act == RBNN_ACT_RELU) return launch_grad_x3<RBNN_ACT_RELU, MODE>(a, st);
    if (act == RBNN_ACT_SIGM || act == RBNN_ACT_TANH) return launch_grad_x3<RBNN_ACT_SIGM, MODE>(a, st);   // both read act' from the stream
#endif
    return launch_grad_x3<RBNN_ACT_LEAKY, MODE>(a, st);
}

}  // namespace

extern "C" {

static int triple_rows_launch(const float* src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                              const rbnn_dev_scale* dev_scale, void* dst, int32_t ld_dst, void* stream, int grouped) {
    if (!src || !dst) return RBNN_ERR_NULL;
    if (rows < 1 || cols < 1 || ld_src < cols || ld_dst < cols || (ld_dst & 31)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const int groups = ld_dst / 8;
    const long long total = (long long)rows * groups;
    hipLaunchKernelGGL(triple_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (long long)rows, cols, ld_src, ldexpf(1.f, scale_exp), dev_scale, (uint4*)dst, groups, grouped);
    return launch_status();
}

int rbnn_triple_rows(const float* src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                     const rbnn_dev_scale* dev_scale, void* dst, int32_t ld_dst, void* stream) {
    return triple_rows_launch(src, rows, cols, ld_src, scale_exp, dev_scale, dst, ld_dst, stream, 0);
}

int rbnn_triple_rows_grouped(const float* src, int64_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                             const rbnn_dev_scale* dev_scale, void* dst, int32_t ld_dst, void* stream) {
    return triple_rows_launch(src, rows, cols, ld_src, scale_exp, dev_scale, dst, ld_dst, stream, 1);
}

int rbnn_triple_cols(const float* W, int64_t n_mats, int32_t rows, int32_t cols, int32_t ld_src, int32_t scale_exp,
                     void* dst, int32_t ld_dst, void* stream) {
    if (!W || !dst) return RBNN_ERR_NULL;
    if (n_mats < 1 || rows < 32 || (rows & 31) || cols < 1 || ld_src < cols || ld_dst < cols || (ld_dst & 15)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const long long total = (long long)n_mats * (rows / 32) * 4 * ld_dst;
    hipLaunchKernelGGL(triple_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W, (long long)n_mats, rows, cols, ld_src, ldexpf(1.f, scale_exp), (uint4*)dst, ld_dst);
    return launch_status();
}

int rbnn_triple_w2gen(const float* W2, int32_t n_mats, int32_t C, int32_t H, int32_t scale_exp, void* dst, void* stream) {
    if (!W2 || !dst) return RBNN_ERR_NULL;
    if (n_mats < 1 || C < 1 || C > 10 || H < 16 || (H & 15)) return RBNN_ERR_SHAPE;
    if (scale_exp < -100 || scale_exp > 100) return RBNN_ERR_SHAPE;
    if (!aligned16(dst)) return RBNN_ERR_ALIGN;
    const long long total = (long long)n_mats * (H / 16) * 128;
    hipLaunchKernelGGL(triple_w2gen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W2, n_mats, C, H, ldexpf(1.f, scale_exp), (uint4*)dst);
    return launch_status();
}

int rbnn_triple_workspace_query(const rbnn_posterior* net, const rbnn_triple_images* tp, int32_t N, int32_t S,
                                rbnn_triple_workspace_sizes* out) {
    if (!net || !tp || !out) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || tp->ld_rows < net->in_features || (tp->ld_rows & 31)) return RBNN_ERR_SHAPE;
    rbnn_triple_workspace_sizes z = {};
    z.X_triple = (size_t)((N + 15) / 16 * 16) * tp->ld_rows * 6;          // grouped image: whole 16-row groups
    z.dZ_gen = (size_t)S * mask_ld(N) * 64;
    z.g_scale = (size_t)mask_ld(N) * sizeof(float);
    z.hid_triple = net->arch == RBNN_ARCH_FC2 ? (size_t)S * ((N + 15) / 16 * 16) * net->hidden * 4 : 0;   // fp32 x 2^h1_exp (round 5; the name stays: ABI)
    *out = z;
    return RBNN_OK;
}

int rbnn_fc_forward_triple(const rbnn_posterior* net, const rbnn_triple_images* tp, const rbnn_triple_workspace* tws,
                           int32_t x_exp, const rbnn_dev_scale* dev_scales, int32_t N, const int32_t* sidx, int32_t S,
                           int32_t out_kind, const rbnn_workspace* ws, void* stream) {
    if (!net || !tp || !tws || !tws->X_triple || !ws || !ws->P || !tp->W1_rows) return RBNN_ERR_NULL;
    if (!net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    const bool fc2 = net->arch == RBNN_ARCH_FC2, bm = net->activation == RBNN_ACT_RELU || net->activation == RBNN_ACT_LEAKY;
    const int H = net->hidden, ld = tp->ld_rows;
    if (H < 128 || (H % 128) || ld < net->in_features || (ld & 31)) return RBNN_ERR_SHAPE;
    if (net->n_classes < 1 || net->n_classes > RBNN_CPAD || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    // the kernel addresses a sample's weight image and the input image with 32-bit byte offsets from a 64-bit base
    if ((long long)H * ld * 6 >= (1LL << 32) || ((long long)N + 15) * ld * 6 >= (1LL << 32) || ((long long)N + 15) * H * 6 >= (1LL << 32)) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(tws->X_triple) || !aligned16(tp->W1_rows) || !aligned16(ws->P) || !aligned16(net->b1) || !aligned16(net->W2)) return RBNN_ERR_ALIGN;
    if (fc2 && (!tp->Wm_rows || !net->bm || !tws->hid_triple || (bm ? !ws->mask2 : !ws->dact2))) return RBNN_ERR_NULL;
    if (fc2 && (!aligned16(tp->Wm_rows) || !aligned16(tws->hid_triple) || !aligned16(net->bm))) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    FwdX3Args a = {};
    a.X = (const char*)tws->X_triple; a.ldx = ld; a.N = N; a.x_sample_bytes = 0; a.x_group_bytes = (unsigned)(ld / 32) * 3072u; a.x_stage_bytes = 3072u;
    a.W = (const char*)tp->W1_rows; a.w_sample_bytes = (long long)H * ld * 6; a.ldw = ld; a.KT = ld / 32;
    a.b = net->b1; a.W2 = net->W2; a.b2 = net->b2; a.C = net->n_classes; a.H = H;
    a.sidx = sidx; a.S = S; a.out_scale = ldexpf(1.f, -((dev_scales ? 0 : x_exp) + tp->w1_exp)); a.x_ds = dev_scales;
    a.P = ws->P; a.mask = ws->mask1; a.dact = ws->dact1; a.out_kind = out_kind;
    if (!fc2) return launch_forward_x3<true>(net->activation, a, st);
    // fc2: layer 1 -> hidden activations as a triple-rows image in tws->hid_triple, scaled by 2^h1_exp (the caller bounds |h|:
    // max_h sum_d |W1[h,d]| * max|x| + max|b1|; record [1] of rbnn_input_scales on the device); layer 2 reads it per sample
    a.hid = (char*)tws->hid_triple; a.hid_scale = ldexpf(1.f, tp->h1_exp); a.hid_ds = dev_scales ? dev_scales + 1 : nullptr;
    int rc = launch_forward_x3<false>(net->activation, a, st);
    if (rc) return rc;
    FwdX3Args b = a;
    const int NG = (N + 15) / 16;
    b.X = (const char*)tws->hid_triple; b.ldx = H; b.x_sample_bytes = (long long)NG * 16 * H * 4; b.x_group_bytes = 2048u; b.x_stage_bytes = (unsigned)NG * 2048u;
    b.W = (const char*)tp->Wm_rows; b.w_sample_bytes = (long long)H * H * 6; b.ldw = H; b.KT = H / 32;
    b.b = net->bm; b.out_scale = ldexpf(1.f, -((dev_scales ? 0 : tp->h1_exp) + tp->wm_exp));
    b.x_ds = dev_scales ? dev_scales + 1 : nullptr; b.hid_ds = nullptr;
    b.mask = ws->mask2; b.dact = ws->dact2; b.hid = nullptr;
    return launch_forward_x3<true, true>(net->activation, b, st);      // layer 2: its X operand is the fp32 hidden image
}

int rbnn_step_tail_triple(int32_t mode, const float* P, const int32_t* labels, int32_t S, float inv_S, int32_t N, int32_t C, float* Psum_out,
                          int32_t ldo, const rbnn_triple_workspace* tws, void* stream) {
    if (!P || !labels || !tws || !tws->dZ_gen || !tws->g_scale) return RBNN_ERR_NULL;
    if (mode != RBNN_LOSS_MEAN_PROB && mode != RBNN_LOSS_PER_SAMPLE && mode != RBNN_LOSS_MEAN_LOGIT) return RBNN_ERR_UNSUPPORTED;
    if (S < 1 || N < 1 || C < 1 || C > 10 || (Psum_out && ldo < C)) return RBNN_ERR_SHAPE;
    if (!aligned16(P) || !aligned16(tws->dZ_gen)) return RBNN_ERR_ALIGN;
    const long long n_pad = mask_ld(N);
    const dim3 grid((unsigned)(n_pad / 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (mode == RBNN_LOSS_MEAN_PROB)
        hipLaunchKernelGGL(step_tail_x3_kernel<RBNN_LOSS_MEAN_PROB>, grid, block, 0, st, P, labels, S, inv_S, N, n_pad, C, Psum_out, ldo, (uint4*)tws->dZ_gen, tws->g_scale);
    else if (mode == RBNN_LOSS_PER_SAMPLE)
        hipLaunchKernelGGL(step_tail_x3_kernel<RBNN_LOSS_PER_SAMPLE>, grid, block, 0, st, P, labels, S, inv_S, N, n_pad, C, Psum_out, ldo, (uint4*)tws->dZ_gen, tws->g_scale);
    else
        hipLaunchKernelGGL(step_tail_x3_kernel<RBNN_LOSS_MEAN_LOGIT>, grid, block, 0, st, P, labels, S, inv_S, N, n_pad, C, Psum_out, ldo, (uint4*)tws->dZ_gen, tws->g_scale);
    return launch_status();
}

int rbnn_attack_step_triple(float* X, const float* X0, int32_t ldx, const float* G, int32_t K, size_t slab_stride, int32_t ldg, const float* alpha,
                            float alpha_scalar, float eps, int32_t project, int32_t N, int32_t D, const rbnn_dev_scale* dev_scale, void* X_triple,
                            int32_t ld_rows, void* stream) {
    if (!X || !G || (project && !X0) || !dev_scale || !X_triple) return RBNN_ERR_NULL;
    if (N < 1 || D < 1 || ldx < D || ldg < D || K < 1 || ld_rows < D || (ld_rows & 31)) return RBNN_ERR_SHAPE;
    if ((D & 3) || (ldx & 3) || (ldg & 3) || (slab_stride & 3)) return RBNN_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(G) || (project && !aligned16(X0)) || !aligned16(X_triple)) return RBNN_ERR_ALIGN;
    const int groups = ld_rows / 8;
    const long long total = (long long)N * groups;
    hipLaunchKernelGGL(attack_step_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       X, X0, ldx, G, K, (long long)slab_stride, ldg, alpha, alpha_scalar, eps, project, (long long)N, D, dev_scale, (uint4*)X_triple, groups);
    return launch_status();
}

int rbnn_fc_input_grad_triple(const rbnn_posterior* net, const rbnn_triple_images* tp, const int32_t* sidx, int32_t S,
                              int32_t N, int32_t chunk, const rbnn_workspace* ws, const rbnn_triple_workspace* tws,
                              int32_t* n_slabs_out, void* stream) {
    if (!net || !tp || !ws || !tws || !ws->slabs) return RBNN_ERR_NULL;       // ws->dZ == NULL: tws->dZ_gen / g_scale are already built (rbnn_step_tail_triple)
    if (!tp->W1_cols || !tp->W2_gen || !tws->dZ_gen || !tws->g_scale) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    const bool fc2 = net->arch == RBNN_ARCH_FC2, bm = net->activation == RBNN_ACT_RELU || net->activation == RBNN_ACT_LEAKY;
    if (bm ? !ws->mask1 : !ws->dact1) return RBNN_ERR_NULL;
    if (fc2 && (bm ? !ws->mask2 : !ws->dact2)) return RBNN_ERR_NULL;
    const int H = net->hidden, Dp = net->in_stride, C = net->n_classes;
    if (H < 128 || (H % 128) || C < 1 || C > 10 || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    if (tp->ld_cols != Dp || (Dp & 15)) return RBNN_ERR_SHAPE;
    if (!aligned16(tp->W1_cols) || !aligned16(tp->W2_gen) || !aligned16(tws->dZ_gen) || (ws->dZ && !aligned16(ws->dZ))) return RBNN_ERR_ALIGN;
    if (fc2 && (!tp->Wm_cols || !ws->dhid1)) return RBNN_ERR_NULL;
    if (fc2 && (!aligned16(tp->Wm_cols) || !aligned16(ws->dhid1))) return RBNN_ERR_ALIGN;
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
    if (ws->dZ) {
        hipLaunchKernelGGL(triple_dz_kernel, dim3((unsigned)(n_pad / 16)), dim3(256), 0, st,
                           ws->dZ, S, N, n_pad, C, (uint4*)tws->dZ_gen, tws->g_scale);
        if (hipGetLastError() != hipSuccess) return RBNN_ERR_LAUNCH;
    }
    GradX3Args g = {};
    g.dzg = (const char*)tws->dZ_gen; g.n_pad = n_pad; g.gscale = tws->g_scale;
    g.W2g = (const char*)tp->W2_gen;
    g.H = H; g.HW = H / 32; g.sidx = sidx; g.S = S; g.N = N;
    if (!fc2) {
        g.mask = ws->mask1; g.dact = ws->dact1; g.W1c = (const char*)tp->W1_cols; g.ldc = tp->ld_cols; g.Dt = Dp / 16;
        g.chunk = chunk; g.nchunks = nchunks; g.out = ws->slabs; g.ldo = Dp;
        g.out_scale = ldexpf(1.f, -(tp->w2_exp + GEN_Q3 + tp->w1_exp));
        return launch_grad_x3_act<X3_FC>(net->activation, g, st);
    }
    // fc2 step 1, one sample per block: dhid1[s] = act'(A1_s) * ((act'(A2_s) * (dZ_s . W3_s)) . Wm_s), kept scaled:
    //   stored = dhid1 * 2^(e(n) + e_w3 + GEN_Q3 + e_wm - Q2),  Q2 = 14 + ceil(log2 H): |dA2 scaled| <= 2^15, |Wm scaled| <= 2^14, K = H
    //   => |stored| <= 2^15: in fp16 range, ready to be split as step 2's A operand
    int q2 = 14;
    while ((1 << (q2 - 14)) < H) ++q2;
    g.mask = ws->mask2; g.dact = ws->dact2; g.odact = ws->dact1; g.W1c = (const char*)tp->Wm_cols; g.ldc = H; g.Dt = H / 16;
    g.chunk = 1; g.nchunks = S; g.out = ws->dhid1; g.ldo = H; g.out_scale = ldexpf(1.f, -q2);
    g.omask = ws->mask1; g.OHW = H / 32;
    int rc = launch_grad_x3_act<X3_FC2_STEP1>(net->activation, g, st);
    if (rc) return rc;
    // fc2 step 2: slabs[k] = sum_{s in chunk k} dhid1[s] . W1_s; acc = g * 2^(e(n) + e_w3 + GEN_Q3 + e_wm - Q2 + e_w1)
    GradX3Args h = g;
    h.amem = ws->dhid1; h.W1c = (const char*)tp->W1_cols; h.ldc = tp->ld_cols; h.Dt = Dp / 16;
    h.chunk = chunk; h.nchunks = nchunks; h.out = ws->slabs; h.ldo = Dp;
    h.out_scale = ldexpf(1.f, -(tp->w2_exp + GEN_Q3 + tp->wm_exp - q2 + tp->w1_exp));
    return launch_grad_x3_act<X3_FC2_STEP2>(net->activation, h, st);
}

}  // extern "C"
