// rbnn_common.hpp — shared device/host helpers of the gfx950 kernels (rbnn_kernels.hip, rbnn_conv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "../../include/robustbnns_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)   // split-half mode (rbnn_split.hip, rbnn_conv.hip)
// Diagnostic ablation bits (tools/ablate.hip builds this file with RBNN_ABL != 0 to price each part of the K loops;
// results are then wrong by construction).  1: no LDS-DMA in the loop  2: no barrier in the loop
// 4: operands not re-read from LDS  8: skip the epilogue  16: grad: no dA generation in the loop
// 32: grad: no mask-word loads in the loop  64: grad: no W1/W2 LDS-DMA in the loop (mask loads kept)
#ifndef RBNN_ABL
#define RBNN_ABL 0
#endif
// FENCE: every timing-only ablation switch (results wrong by design) needs -DRBNN_ALLOW_ABLATION beside it, and a translation unit built
// with one plants the (weak) marker rbnn_ablation_build_marker, which rbnn_build_flags() reports: robustbnns_amd._hip.load() refuses such
// a library unless RBNN_ALLOW_ABLATION=1 is in the environment (the variant scripts under tools/ set it for their own runs only).
#if (RBNN_ABL != 0) || defined(RBNN_X3FWD_ABL_NOFILL) || defined(RBNN_X3FWD_ABL_NOEPI) || defined(RBNN_DENSE_ABL_NOA) || \
    defined(RBNN_DENSE_ABL_NOMFMA) || defined(RBNN_DENSE_ABL_NOB) || defined(RBNN_DENSE_ABL_NOBAR) || defined(RBNN_DENSE_ABL_NOAREAD) || \
    defined(RBNN_DENSE_ABL_NOROUTE) || defined(RBNN_DENSE_ABL_NOEPI) || defined(RBNN_X3_L1_ABL_NOSTORE) || defined(RBNN_X3_L1_ABL_SMALL) || \
    defined(RBNN_FAST_BUILD) || defined(RBNN_DENSE_STAMPS) || defined(RBNN_DENSE_ABL_PARTA) || defined(RBNN_DENSE_ABL_HALFBAR)
#ifndef RBNN_ALLOW_ABLATION
#error "a timing-only ablation switch (RBNN_ABL / RBNN_*_ABL_* / RBNN_FAST_BUILD) is set: such a build computes wrong results; pass -DRBNN_ALLOW_ABLATION to build it on purpose"
#endif
extern "C" __attribute__((weak, visibility("default"))) int rbnn_ablation_build_marker = 1;
#endif
// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
#define VMCNT(n) ((((n) & 15) | (((n) >> 4) << 14)) | 0x0F70)
#define VMCNT_LGKM0(n) ((((n) & 15) | (((n) >> 4) << 14)) | 0x0070)
#define LEAKY_SLOPE 0.01f                       // torch.nn.LeakyReLU() default (model_nn.py:68-69)

namespace {

// ---------------------------------------------------------------------------------------------------
// Work-item <-> block map.  Blocks b and b+8 share an XCD (observed round-robin dispatch; speed only):
// give every XCD one contiguous run of the item space so that blocks resident together on an XCD work on
// the same sample / the same W1 column slice and share it in that XCD's L2.  Bijective for any M.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool item_of_block(int b, int M, int& id) {
    const int q = M >> 3, r = M & 7, x = b & 7, j = b >> 3;
    if (j >= q + (x < r ? 1 : 0)) return false;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    return true;
}

// Softmax backward dZ_c = p_c (g_c - <g, p>) for p = softmax(z), evaluated WITHOUT the cancellation of that form.  A trained net's softmax saturates
// (p_max = 1 - 1e-6): for the WINNING class m the difference g_m - <g, p> = g_m (1 - p_m) - sum_{k != m} g_k p_k cancels to O(1 - p_max) with an absolute
// error of an ulp of g — several per cent of the result (torch's autograd evaluates exactly that form: the reference's own fp32 gradients sit up to
// 3e-3 from an fp64 evaluation on the trained fixtures, tests/conftest.py).  Since sum_k p_k = 1,   g_m - <g, p> = sum_k p_k (g_m - g_k)   (the k = m
// term is exactly 0): every term is a product of accurately known factors (the small p_k come out of exp(z_k - z_max) with full relative
// precision), nothing cancels.  The other classes need nothing special: g_c - <g, p> is then a difference of O(1) numbers with an O(1) result.
// ~5 C operations per (sample, point) (the all-pairs form, C^2, made step_tail_x3_kernel 76 -> 132 us at C2: it evaluates this twice per pair).
// C (<= CM <= 16) live classes.  The one definition all kernels use (loss_dlogits_kernel, step_tail_x3_kernel, the lowdim kernels), so the fused
// and the separate forms stay bit-identical.
template <int CM> __device__ __forceinline__ void softmax_backward(const float (&g)[CM], const float (&p)[CM], int C, float (&out)[CM]) {
    float dot = 0.f, pm = -INFINITY, gm = 0.f;
    int m = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c)
        if (c < C) {
            dot += g[c] * p[c];
            if (p[c] > pm) { pm = p[c]; gm = g[c]; m = c; }             // the first maximum
        }
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < CM; ++k)
        if (k < C) sm = fmaf(p[k], gm - g[k], sm);
#pragma unroll
    for (int c = 0; c < CM; ++c) out[c] = (c < C) ? (c == m ? sm : g[c] - dot) * p[c] : 0.f;
}

// dL/dt of L = CE(softmax(t), y) (times inv_S): softmax(t) - e_y.  For the LABEL class the textbook form t_y / den - 1 cancels to O(1 - p_y): on a trained
// net (p_y = 1 - 1e-6) that is an absolute error of an ulp of 1 on a result of 1e-6 — several per cent; torch's cross-entropy backward (the reference)
// evaluates exactly that.  Since sum_k softmax_k = 1 the label class is -(sum_{k != y} e_k) / den: a sum of accurately known positive terms, nothing
// cancels.  The other classes are e_c / den as they were.  One definition for every kernel that forms a loss gradient (loss_dlogits_kernel,
// step_tail_x3_kernel, the lowdim kernels): the fused and the separate forms stay bit-identical.  t[c] is read for c < C only.
template <int CM> __device__ __forceinline__ void ce_softmax_grad(const float (&t)[CM], int C, int y, float inv_S, float (&g)[CM]) {
    float e[CM], m = -INFINITY, den = 0.f, rest = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) m = fmaxf(m, t[c]);
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        e[c] = (c < C) ? expf(t[c] - m) : 0.f;
        den += e[c];
        if (c != y) rest += e[c];
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) g[c] = (c < C) ? ((c == y ? -rest : e[c]) / den) * inv_S : 0.f;
}

// Item -> (point tile, column group, chunk) of the input-gradient GEMM kernels.  Kernels that GENERATE their A operand read no A from memory:
// point tiles run fastest, so that the blocks resident on an XCD together share one (chunk, column group)'s W1 slices in its L2.  fc2's step
// through W1 READS its A operand — dL/d(pre-activation 1) of (point tile, chunk), 32 KB per stage — and every column group reads it again: with
// the point tiles fastest the seven readers of a tile ran 40 items apart and each fetched it from beyond the L2 (PMC, profiles/r05p/fc2: 18.5 GB
// per backward call, 4 TB/s under a 3.9-ms kernel).  a_from_memory: the order is blocked 2-D — TB point tiles x all column groups, column
// group fastest — so that the ~56 blocks an XCD holds at a time read 8 A tiles and 7 W1 slices per stage between them.  Same work per item:
// results bit-identical.  fc2-512 backward 6.61 -> 6.17 ms, fc2-1024 17.74 -> 16.73 (same box, profiles/r05s).
#ifndef RBNN_GRAD_STEP2_TB
#define RBNN_GRAD_STEP2_TB 8
#endif
__device__ __forceinline__ void grad_item(int id, int NT, int ND, bool a_from_memory, int& ntile, int& dg, int& ch) {
    if (a_from_memory && RBNN_GRAD_STEP2_TB > 0) {
        constexpr int TB = RBNN_GRAD_STEP2_TB > 0 ? RBNN_GRAD_STEP2_TB : 1;
        const int per = NT * ND, blk = TB * ND, full = NT / TB;
        ch = id / per;
        const int il = id - ch * per;
        const int tb = min(il / blk, full), r = il - tb * blk;   // (the last block of point tiles may hold fewer than TB)
        dg = r % ND;
        ntile = tb * TB + r / ND;
    } else {
        ntile = id % NT; dg = (id / NT) % ND; ch = id / (NT * ND);
    }
}

// compile-time loop: the body sees its index as a constant (sched_group_barrier sizes must be constant expressions)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// The three fp16 pieces (rbnn_triple.hip split3: p0 = f16(v), p1 = f16(v - p0), p2 = f16(v - p0 - p1), round-to-nearest-even) of TWO fp32 values,
// packed [even | odd << 16] per piece: 6 vector instructions per pair (plain C++ compiles to ~12 per value).  `one` must hold 1.0f.
// A VGPR written by a vector instruction needs TWO wait states before an MFMA takes it as an operand; hipcc's hazard recognizer cannot see into an asm
// block and pads ONE state behind it — an MFMA scheduled straight behind the block (any block: these are plain asm statements, the scheduler orders
// them freely) read a stale d2: fc2's layer 2 on the fp32 hidden image was off by the low
// pieces, 7e-4, on exactly the last point tile of the 8-wave configuration; with the pad only behind the LAST of a fragment's four pairs it failed again
// (DESIGN §3z).  `s_nop 1` inside the string closes the window for every consumer, whatever the scheduler puts there; it is part of this
// block for every user.  The gradient kernel's own pair splits (split3_pair / split3_pair_m, rbnn_triple.hip) are fenced by a scan of the BUILT
// library's disassembly instead — tools/kernel_resources.py::mfma_operand_hazards, tests/test_host_cpu.py::
// test_no_mfma_reads_a_vgpr_inside_the_valu_write_window, which covers every kernel — because sixteen pads per wave-stage cost that kernel 0.5-1.7 %.
__device__ __forceinline__ void split3_plain_pair(float ve, float vo, float one, unsigned& d0, unsigned& d1, unsigned& d2) {
    float re, ro;
    asm("v_cvt_pk_f16_f32 %[d0], %[ve], %[vo]\n\t"
        "v_fma_mix_f32 %[re], %[ve], %[one], -%[d0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %[ro], %[vo], %[one], -%[d0] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %[d1], %[re], %[ro]\n\t"
        "v_fma_mixlo_f16 %[d2], -%[d1], %[one], %[re] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %[d2], -%[d1], %[one], %[ro] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [re] "=&v"(re), [ro] "=&v"(ro)
        : [ve] "v"(ve), [vo] "v"(vo), [one] "v"(one));
}

// row stride (in words) of the 1-bit activation stash [S][H/32][N_pad]: padded to the gradient kernel's 256-point block so that
// a block's words of one row are one aligned 1-KiB LDS-DMA piece
__host__ __device__ __forceinline__ long long mask_ld(int N) { return ((long long)N + 255) / 256 * 256; }

// 16-float (64 B) LDS rows read with ds_read_b128 by lane (row li, 16-B chunk lg): physical chunk =
// lg ^ swz(row) with swz = [0,2,3,1][(row>>2)&3] makes every 16-lane b128 group hit 16 distinct slots.
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// Split-half stage tiles: rows of 128 B = [hi8 | lo8] x 4 column groups, read with ds_read_b128 (row li, chunk 2*lg / 2*lg+1).
// ds_read_b128 is served in four NON-contiguous 16-lane groups — {0-3,12-15,20-27}, {4-11,16-19,28-31}, same +32 — i.e. rows
// {0-3,12-15} of chunk pair lg together with rows {4-11} of chunk pair lg+1; physical chunk = c ^ row_swz(row) makes every group
// cover 16 distinct 16-B slots of the 256-B bank row (the plain (r>>1)&7 swizzle measured 48 % conflict cycles).
__device__ __forceinline__ int row_swz(int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); }

// Asynchronous 16-B-per-lane global -> LDS copy (global_load_lds_dwordx4): per-lane source, LDS destination =
// wave-uniform base + lane*16.  Completion is tracked by vmcnt; __syncthreads() drains it before the barrier.
__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 16, 0, 0);
}

// Ring hand-off: this wave's DMA pieces except the N youngest have landed and its LDS reads are done (counted
// s_waitcnt), then a raw s_barrier (no vmcnt(0) drain).  The empty asm statements stop hipcc from moving LDS
// accesses across the pair (the raw barrier is not a memory fence to the compiler).
template <int N> __device__ __forceinline__ void ring_wait_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(VMCNT_LGKM0(N));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int ACT> __device__ __forceinline__ float act_fwd(float a) {
    if (ACT == RBNN_ACT_RELU)  return a > 0.f ? a : 0.f;
    if (ACT == RBNN_ACT_LEAKY) return a > 0.f ? a : a * LEAKY_SLOPE;
    if (ACT == RBNN_ACT_SIGM)  return 1.f / (1.f + expf(-a));
    return tanhf(a);
}
// derivative from the activation VALUE (sigmoid / tanh only)
template <int ACT> __device__ __forceinline__ float act_grad_from_value(float h) {
    if (ACT == RBNN_ACT_SIGM) return h * (1.f - h);
    return 1.f - h * h;
}

// ---------------------------------------------------------------------------------------------------
// The SVI draw's generator (rbnn_svi.hip: svi_draw_kernel; rbnn_lowdim.hip: the draw fused into the one-launch path): eps is a pure function of
// (key, draw id, tensor, sample, element quad) — Philox4x32-10 + Box-Muller on the hardware transcendentals.  Shared so that both kernels
// produce the SAME weights for the same (key, draw).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four standard normals from one Philox block: Box-Muller on (x0, x1) and (x2, x3); u = (x + 0.5) 2^-32 in (0, 1)
__device__ __forceinline__ void normal4(const uint32_t x[4], float n[4]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)x[2 * p] + 0.5f) * 2.3283064365386963e-10f;       // fp32: x + 0.5 rounds for x >= 2^24, still in (0, 1]
        const float u2 = ((float)x[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
        // the hardware transcendentals: v_log_f32 (log2, 1 ulp) and v_sin_f32 / v_cos_f32, whose argument is in REVOLUTIONS — sin(2 pi u2)
        // is one instruction.  (libm's logf + sincospif cost ~25 vector instructions per weight and made this kernel ALU-bound.)
        const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(fminf(u1, 0.99999994f)));   // sqrt(-2 ln u1), ln = log2 * ln 2
        n[2 * p] = r * __builtin_amdgcn_cosf(u2); n[2 * p + 1] = r * __builtin_amdgcn_sinf(u2);
    }
}

enum { T_W1 = 0, T_B1 = 1, T_WM = 2, T_BM = 3, T_W2 = 4, T_B2 = 5 };

struct Rng {
    uint32_t k0, k1, c2, c3;
    __device__ __forceinline__ void quad(int tensor, uint32_t q, float n[4]) const {
        uint32_t x[4];
        philox4x32_10(q, (uint32_t)tensor, c2, c3, k0, k1, x);
        normal4(x, n);
    }
};

// one weight of a [rows, cols] row-major guide tensor (quad index = r * ceil(cols/4) + c/4: eps does not depend on the padding)
__device__ __forceinline__ void draw_quad(const Rng& rng, int tensor, const float* loc, const float* scl, int r, int c4, int cols, float w[4]) {
    const int Q = (cols + 3) >> 2;
    float n[4];
    rng.quad(tensor, (uint32_t)(r * Q + c4), n);
    const long long base = (long long)r * cols + 4 * c4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = 4 * c4 + j < cols;
        w[j] = in ? fmaf(scl[base + j], n[j], loc[base + j]) : 0.f;              // Normal.rsample(): loc + eps * scale (model_bnn.py:127-130); scl = softplus(raw scale), applied ONCE per guide by the caller
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int launch_status() { return hipGetLastError() == hipSuccess ? RBNN_OK : RBNN_ERR_LAUNCH; }
inline int grid_for_items(long long M) { return (int)(8 * ((M + 7) / 8)); }
inline int device_cus() {                                      // compute units of the current device (cached per ordinal)
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once per (instantiation,
// device) — `done` is that instantiation's bitmask over device ordinals (a race only repeats an idempotent call).
inline bool ensure_dynamic_lds(const void* kern, int bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (dev >= 0 && dev < 64) done |= 1ull << dev;
    return true;
}

}  // namespace
