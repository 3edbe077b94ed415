// rbnn_svi.hip — the SVI guide's weight draw (model_bnn.py:121-136, :222-232) as ONE kernel that writes a whole stacked posterior
// IN PLACE: W = loc + sigma * eps (sigma = softplus(raw scale), taken once per guide by the caller) with eps generated in registers (Philox4x32-10 + Box-Muller — no eps tensor in HBM),
// stored as the fp32 stack AND, in the same pass, as every image the two GEMM modes read:
//
//      fp32 stack        W1 [S,H,D_pad]  b1  (Wm bm)  W2 [S,C,H]  b2        what rbnn_fc_forward reads
//      rbnn_pack_rows4   W1_pack4 / Wm_pack4 [S,H/4,cols,4]                  the fp32-MFMA backward's B operand
//      triple rows       W1_rows / Wm_rows  [S*H/16, ld/32, 3, 16, 32] halves  rbnn_fc_forward_triple's A operand (grouped rows)
//      triple cols       W1_cols / Wm_cols  [S,H/32,4,3,ld,8] halves         rbnn_fc_input_grad_triple's B operand
//      W2 generator      W2_gen [S,H/16,2,64,8] halves                       its dA generator
//
// The reference draws one net per forward call (pyro.random_module over Normal(loc, softplus(scale)), S calls per prediction); here a
// redraw of all S samples is one launch into buffers whose addresses never change, so the posterior descriptor, the engine and its
// workspaces are reused draw after draw (an SVI PGD attack redraws every iteration: adversarialAttacks.py:95-97 -> model_bnn.py:230-232).
//
// eps is a pure function of (key, draw id, tensor, sample, element): the same (key, draw) gives the same weights whatever the tiling.
// PARITY UNPINNED against pyro-ppl 1.3.0's RNG stream (absent here: SURVEY 8c) — the oracle restates THIS generator.
// The images' power-of-two scales are host integers fixed per guide from the a-priori bound |w| <= |loc| + 6.77 softplus(scale)
// (Box-Muller on a 32-bit uniform cannot exceed sqrt(-2 ln 2^-33) = 6.764): no abs-max pass, no device->host sync per draw.
#include "rbnn_common.hpp"
#include <algorithm>

namespace {

// p0 + p1 + p2 = v exactly (rbnn_triple.hip's split3, plain C: the image builders are not instruction-issue bound)
__device__ __forceinline__ void split3(float v, _Float16& a, _Float16& b, _Float16& c) {
    a = (_Float16)v;
    const float r1 = v - (float)a;
    b = (_Float16)r1;
    c = (_Float16)(r1 - (float)b);
}

struct DrawArgs {
    rbnn_posterior net;
    rbnn_triple_images tp;
    rbnn_svi_guide g;
    const unsigned long long* sample_keys;
    unsigned long long key;
    uint32_t draw_id;
    int S, has_tp, tiles_w1_d, tiles_wm_d, tiles_per_sample;
    int images_only;               // rbnn_svi_draw_images: W1 / Wm go into the triple images only (their fp32 stack + pack_rows4 copies are skipped)
    float w1_scale, wm_scale, w2_scale;
};

// ---------------------------------------------------------------------------------------------------
// A [H, cols] matrix tile of 32 hidden units x 64 columns of sample s: drawn into LDS, then written in every layout.
// ---------------------------------------------------------------------------------------------------
__device__ void draw_matrix_tile(const DrawArgs& a, const Rng& rng, int tensor, int s, int hb, int dt, const float* loc, const float* scl,
                                 int H0, int cols, float* W, int ldw, float* pack4, uint4* rows_img, int ld_rows, uint4* cols_img, int ld_cols,
                                 float img_scale, float (*tile)[68]) {
    const int H = a.net.hidden, t = threadIdx.x, d0 = 64 * dt;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = t + 256 * i, hl = q >> 4, dq = q & 15;
        const int h = 32 * hb + hl, c4 = (d0 >> 2) + dq;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (h < H0 && 4 * c4 < cols) draw_quad(rng, tensor, loc, scl, h, c4, cols, w);
        *(f32x4*)&tile[hl][4 * dq] = (f32x4){w[0], w[1], w[2], w[3]};
    }
    __syncthreads();
    // (1) fp32 stack [S][H][ldw] and (2) its pack_rows4 image [S][H/4][ldw][4]
    if (W) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = t + 256 * i, hl = q >> 4, dq = q & 15, d = d0 + 4 * dq;
            if (d < ldw) *(f32x4*)(W + ((long long)s * H + 32 * hb + hl) * ldw + d) = *(const f32x4*)&tile[hl][4 * dq];
        }
    }
    if (pack4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = t + 256 * i, hq = q >> 6, dl = q & 63, d = d0 + dl;
            if (d < ldw)
                *(f32x4*)(pack4 + (((long long)s * (H / 4) + 8 * hb + hq) * ldw + d) * 4) =
                    (f32x4){tile[4 * hq][dl], tile[4 * hq + 1][dl], tile[4 * hq + 2][dl], tile[4 * hq + 3][dl]};
        }
    }
    if (rows_img) {
        // (3) triple rows, GROUPED (rbnn_triple_rows_grouped): row r = s*H + h; block of (16-row group, stage of 32 columns) = 192 units of 16 B:
        //     [piece][16 rows][4 groups of 8 columns]
        const int hl = t >> 3, g = t & 7, d = d0 + 8 * g;
        if (d < ld_rows) {
            // (the 6-instruction pair split of rbnn_common.hpp — the same pieces as split3, bit for bit; plain C spends ~12 vector instructions per
            // VALUE, and with Philox + Box-Muller this kernel is bound by vector issue, not by its stores: 3.4 of ~6 TB/s)
            union { unsigned w[4]; uint4 u; } o[3];
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                split3_plain_pair(tile[hl][8 * g + j] * img_scale, tile[hl][8 * g + j + 1] * img_scale, 1.f, o[0].w[j >> 1], o[1].w[j >> 1], o[2].w[j >> 1]);
            const long long r = (long long)s * H + 32 * hb + hl;
            uint4* const out = rows_img + (((r >> 4) * (ld_rows >> 5) + (d >> 5)) * 192 + (r & 15) * 4 + ((d >> 3) & 3));
            out[0] = o[0].u; out[64] = o[1].u; out[128] = o[2].u;
        }
        // (4) triple cols: out[s][hb][lg][p][d][j] = piece p of W[32 hb + 16 (j>>2) + 4 lg + (j&3)][d]
        const int lg = t >> 6, dl = t & 63, dc = d0 + dl;
        if (dc < ld_cols) {
            union { unsigned w[4]; uint4 u; } o[3];
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                split3_plain_pair(tile[16 * (j >> 2) + 4 * lg + (j & 3)][dl] * img_scale, tile[16 * ((j + 1) >> 2) + 4 * lg + ((j + 1) & 3)][dl] * img_scale, 1.f,
                                  o[0].w[j >> 1], o[1].w[j >> 1], o[2].w[j >> 1]);
            const long long base = ((((long long)s * (H / 32) + hb) * 4 + lg) * 3) * ld_cols;
            cols_img[base + dc] = o[0].u;
            cols_img[base + ld_cols + dc] = o[1].u;
            cols_img[base + 2LL * ld_cols + dc] = o[2].u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The small tensors of sample s: b1, (bm), b2, and the output layer W2 [C, H] + its dA-generator image.
// ---------------------------------------------------------------------------------------------------
__device__ void draw_small(const DrawArgs& a, const Rng& rng, int s, float* w2l /* [C][H] dynamic LDS */) {
    const int H = a.net.hidden, H0 = a.g.hidden, C = a.net.n_classes, t = threadIdx.x;
    float* const b1 = const_cast<float*>(a.net.b1) + (long long)s * H;
    for (int q = t; 4 * q < H; q += 256) {
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * q < H0) draw_quad(rng, T_B1, a.g.b1_loc, a.g.b1_scale, 0, q, H0, w);
        *(f32x4*)(b1 + 4 * q) = (f32x4){w[0], w[1], w[2], w[3]};
    }
    if (a.net.arch == RBNN_ARCH_FC2) {
        float* const bm = const_cast<float*>(a.net.bm) + (long long)s * H;
        for (int q = t; 4 * q < H; q += 256) {
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (4 * q < H0) draw_quad(rng, T_BM, a.g.bm_loc, a.g.bm_scale, 0, q, H0, w);
            *(f32x4*)(bm + 4 * q) = (f32x4){w[0], w[1], w[2], w[3]};
        }
    }
    if (t < (C + 3) / 4) {
        float w[4];
        draw_quad(rng, T_B2, a.g.b2_loc, a.g.b2_scale, 0, t, C, w);
        float* const b2 = const_cast<float*>(a.net.b2) + (long long)s * C;
        for (int j = 0; j < 4; ++j)
            if (4 * t + j < C) b2[4 * t + j] = w[j];
    }
    float* const W2 = const_cast<float*>(a.net.W2) + (long long)s * C * H;
    const int Q = H / 4;
    for (int i = t; i < C * Q; i += 256) {
        const int c = i / Q, q = i % Q;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * q < H0) draw_quad(rng, T_W2, a.g.W2_loc, a.g.W2_scale, c, q, H0, w);
        *(f32x4*)(W2 + (long long)c * H + 4 * q) = (f32x4){w[0], w[1], w[2], w[3]};
        *(f32x4*)(w2l + c * H + 4 * q) = (f32x4){w[0], w[1], w[2], w[3]};
    }
    if (!a.has_tp) return;
    __syncthreads();
    // W2 generator image (rbnn_triple.hip triple_w2gen_kernel's slot plan): out[s][tile][k][lane][8]
    uint4* const gen = (uint4*)const_cast<void*>(a.tp.W2_gen) + (long long)s * (H / 16) * 128;
    for (int i = t; i < (H / 16) * 128; i += 256) {
        const int lane = i & 63, li = lane & 15, lg = lane >> 4, k = (i >> 6) & 1, tl = i >> 7;
        _Float16 w[3][10];
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const float v = (c < C) ? w2l[c * H + 16 * tl + li] * a.w2_scale : 0.f;
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
        gen[i] = o.u;
    }
}

__global__ void __launch_bounds__(256) svi_draw_kernel(const DrawArgs a) {
    extern __shared__ float dyn_lds[];
    const int s = blockIdx.x / a.tiles_per_sample, j = blockIdx.x % a.tiles_per_sample;
    const unsigned long long key = a.sample_keys ? a.sample_keys[s] : a.key;
    const Rng rng = {(uint32_t)key, (uint32_t)(key >> 32), a.sample_keys ? 0u : (uint32_t)s, a.draw_id};
    const int H = a.net.hidden;
    const int n_w1 = (H / 32) * a.tiles_w1_d, n_wm = (H / 32) * a.tiles_wm_d;
    float (*tile)[68] = (float (*)[68])dyn_lds;
    if (j < n_w1) {
        draw_matrix_tile(a, rng, T_W1, s, j / a.tiles_w1_d, j % a.tiles_w1_d, a.g.W1_loc, a.g.W1_scale, a.g.hidden, a.net.in_features,
                         a.images_only ? nullptr : const_cast<float*>(a.net.W1), a.net.in_stride, a.images_only ? nullptr : const_cast<float*>(a.net.W1_pack4),
                         a.has_tp ? (uint4*)const_cast<void*>(a.tp.W1_rows) : nullptr, a.tp.ld_rows,
                         a.has_tp ? (uint4*)const_cast<void*>(a.tp.W1_cols) : nullptr, a.tp.ld_cols, a.w1_scale, tile);
    } else if (j < n_w1 + n_wm) {
        const int jj = j - n_w1;
        draw_matrix_tile(a, rng, T_WM, s, jj / a.tiles_wm_d, jj % a.tiles_wm_d, a.g.Wm_loc, a.g.Wm_scale, a.g.hidden, a.g.hidden,
                         a.images_only ? nullptr : const_cast<float*>(a.net.Wm), H, a.images_only ? nullptr : const_cast<float*>(a.net.Wm_pack4),
                         a.has_tp ? (uint4*)const_cast<void*>(a.tp.Wm_rows) : nullptr, H,
                         a.has_tp ? (uint4*)const_cast<void*>(a.tp.Wm_cols) : nullptr, H, a.wm_scale, tile);
    } else {
        draw_small(a, rng, s, dyn_lds);
    }
}


// ---------------------------------------------------------------------------------------------------
// Flat draw: up to 8 tensors of any shape (the conv architecture's six), W[s][e] = loc[e] + sigma[e] * eps(s, e) (sigma = softplus(raw scale)), eps of
// element e = component e % 4 of the Philox block with counter (e / 4, tensor id, s or 0, draw id).  One launch for all tensors and
// samples; derived weight images (regrouped / triple images of model.3.weight) are rebuilt from the fp32 stack by their own builders.
// ---------------------------------------------------------------------------------------------------
struct FlatArgs {
    rbnn_svi_flat_tensor t[8];
    long long first_block[9];      // blocks of tensor i: [first_block[i], first_block[i + 1]) within one sample's range
    int n_tensors, S;
    const unsigned long long* sample_keys;
    unsigned long long key;
    uint32_t draw_id;
};

__global__ void __launch_bounds__(256) svi_draw_flat_kernel(const FlatArgs a) {
    const long long per_sample = a.first_block[a.n_tensors];
    const int s = (int)(blockIdx.x / per_sample);
    const long long b = blockIdx.x % per_sample;
    int ti = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < a.n_tensors && b >= a.first_block[i]) ti = i;
    const rbnn_svi_flat_tensor t = a.t[ti];
    const long long q = (b - a.first_block[ti]) * 256 + threadIdx.x, e0 = 4 * q;
    if (e0 >= t.n_elem) return;
    const unsigned long long key = a.sample_keys ? a.sample_keys[s] : a.key;
    const Rng rng = {(uint32_t)key, (uint32_t)(key >> 32), a.sample_keys ? 0u : (uint32_t)s, a.draw_id};
    float n[4], w[4];
    rng.quad(t.tensor_id, (uint32_t)q, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (e0 + j < t.n_elem) ? fmaf(t.sigma[e0 + j], n[j], t.loc[e0 + j]) : 0.f;
    float* const out = t.out + (long long)s * t.out_sample_stride + e0;
    if (e0 + 3 < t.n_elem && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) *(f32x4*)out = (f32x4){w[0], w[1], w[2], w[3]};
    else
        for (int j = 0; j < 4; ++j) if (e0 + j < t.n_elem) out[j] = w[j];
}

}  // namespace

extern "C" {

int rbnn_svi_draw_flat(const rbnn_svi_flat_tensor* tensors, int32_t n_tensors, int32_t n_samples, const uint64_t* sample_keys, uint64_t key,
                       uint32_t draw_id, void* stream) {
    if (!tensors) return RBNN_ERR_NULL;
    if (n_tensors < 1 || n_tensors > 8 || n_samples < 1) return RBNN_ERR_SHAPE;
    FlatArgs a = {};
    a.n_tensors = n_tensors; a.S = n_samples; a.sample_keys = (const unsigned long long*)sample_keys; a.key = key; a.draw_id = draw_id;
    long long blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const rbnn_svi_flat_tensor& t = tensors[i];
        if (!t.loc || !t.sigma || !t.out) return RBNN_ERR_NULL;
        if (t.n_elem < 1 || t.n_elem > 0x3FFFFFFFFLL || t.out_sample_stride < t.n_elem) return RBNN_ERR_SHAPE;
        a.t[i] = t;
        a.first_block[i] = blocks;
        blocks += ((t.n_elem + 3) / 4 + 255) / 256;
    }
    for (int i = n_tensors; i <= 8; ++i) a.first_block[i] = blocks;
    if (blocks * n_samples > 0x7FFFFFFFLL) return RBNN_ERR_SHAPE;
    hipLaunchKernelGGL(svi_draw_flat_kernel, dim3((unsigned)(blocks * n_samples)), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}

// the small-tensor block stages W2 [C, H] in dynamic LDS: C * H * 4 bytes of the CU's 160 KB (hidden 4096 at 10 classes)
static constexpr size_t SVI_DRAW_LDS_MAX = 160 * 1024;

int rbnn_svi_draw_supported(const rbnn_posterior* net, int32_t with_triple_images) {
    if (!net || (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2)) return 0;
    const int H = net->hidden, C = net->n_classes;
    if (H < 32 || (H & 31) || C < 1 || C > 16 || net->in_features < 1 || net->in_stride < net->in_features || (net->in_stride & 15)) return 0;
    if ((size_t)C * H * sizeof(float) > SVI_DRAW_LDS_MAX) return 0;
    if (with_triple_images && (C > 10 || (H & 127))) return 0;
    return 1;
}

static int svi_draw_impl(const rbnn_posterior* net, const rbnn_triple_images* tp, const rbnn_svi_guide* g, int32_t n_samples,
                         const uint64_t* sample_keys, uint64_t key, uint32_t draw_id, int images_only, void* stream) {
    if (!net || !g || !net->W1 || !net->b1 || !net->W2 || !net->b2) return RBNN_ERR_NULL;
    if (images_only && !tp) return RBNN_ERR_NULL;
    if (!g->W1_loc || !g->W1_scale || !g->b1_loc || !g->b1_scale || !g->W2_loc || !g->W2_scale || !g->b2_loc || !g->b2_scale) return RBNN_ERR_NULL;
    if (net->arch != RBNN_ARCH_FC && net->arch != RBNN_ARCH_FC2) return RBNN_ERR_UNSUPPORTED;
    const bool fc2 = net->arch == RBNN_ARCH_FC2;
    if (fc2 && (!net->Wm || !net->bm || !g->Wm_loc || !g->Wm_scale || !g->bm_loc || !g->bm_scale)) return RBNN_ERR_NULL;
    const int H = net->hidden, C = net->n_classes;
    if (n_samples < 1 || n_samples > net->n_stored || H < 32 || (H & 31) || g->hidden < 1 || g->hidden > H || C < 1 || C > 16) return RBNN_ERR_SHAPE;
    if (net->in_features < 1 || net->in_stride < net->in_features || (net->in_stride & 15)) return RBNN_ERR_SHAPE;
    DrawArgs a = {};
    a.net = *net; a.g = *g; a.sample_keys = (const unsigned long long*)sample_keys; a.key = key; a.draw_id = draw_id; a.S = n_samples;
    a.images_only = images_only;
    int extent = net->in_stride;
    if (tp) {
        if (!tp->W1_rows || !tp->W1_cols || !tp->W2_gen || (fc2 && (!tp->Wm_rows || !tp->Wm_cols))) return RBNN_ERR_NULL;
        if (tp->ld_rows < net->in_features || (tp->ld_rows & 31) || tp->ld_cols != net->in_stride || C > 10 || (H & 127)) return RBNN_ERR_SHAPE;
        if (tp->w1_exp < -100 || tp->w1_exp > 100 || tp->w2_exp < -100 || tp->w2_exp > 100 || tp->wm_exp < -100 || tp->wm_exp > 100) return RBNN_ERR_SHAPE;
        a.tp = *tp; a.has_tp = 1;
        a.w1_scale = ldexpf(1.f, tp->w1_exp); a.w2_scale = ldexpf(1.f, tp->w2_exp); a.wm_scale = ldexpf(1.f, tp->wm_exp);
        extent = std::max(extent, (int)tp->ld_rows);
    }
    a.tiles_w1_d = (extent + 63) / 64;
    a.tiles_wm_d = fc2 ? (H + 63) / 64 : 0;
    a.tiles_per_sample = (H / 32) * (a.tiles_w1_d + a.tiles_wm_d) + 1;
    const size_t lds = std::max((size_t)32 * 68 * sizeof(float), (size_t)C * H * sizeof(float));
    if (lds > SVI_DRAW_LDS_MAX) return RBNN_ERR_SHAPE;                     // rbnn_svi_draw_supported() says so beforehand
    static unsigned long long attr = 0;
    if (lds > 64 * 1024 && !ensure_dynamic_lds((const void*)svi_draw_kernel, (int)SVI_DRAW_LDS_MAX, attr)) return RBNN_ERR_LAUNCH;
    hipLaunchKernelGGL(svi_draw_kernel, dim3((unsigned)((long long)n_samples * a.tiles_per_sample)), dim3(256), lds, (hipStream_t)stream, a);
    return launch_status();
}

int rbnn_svi_draw(const rbnn_posterior* net, const rbnn_triple_images* tp, const rbnn_svi_guide* g, int32_t n_samples,
                  const uint64_t* sample_keys, uint64_t key, uint32_t draw_id, void* stream) {
    return svi_draw_impl(net, tp, g, n_samples, sample_keys, key, draw_id, 0, stream);
}

int rbnn_svi_draw_images(const rbnn_posterior* net, const rbnn_triple_images* tp, const rbnn_svi_guide* g, int32_t n_samples,
                         const uint64_t* sample_keys, uint64_t key, uint32_t draw_id, void* stream) {
    return svi_draw_impl(net, tp, g, n_samples, sample_keys, key, draw_id, 1, stream);
}

}  // extern "C"
