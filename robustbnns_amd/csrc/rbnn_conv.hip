// rbnn_conv.hip — the reference's `conv` architecture (model_nn.py:93-106) on gfx950:
//   Conv2d(1,32,5) -> act -> MaxPool2d(2) -> Conv2d(32,Hc,5) -> act -> MaxPool2d(2, stride 1) -> Flatten -> Linear(49*Hc, C)
// on 1x28x28 inputs (the only input size the reference's head is correct for: model_nn.py:95-96,106).
//
// Per posterior sample the weights differ, so every layer is batched over (sample, point):
//   conv1_pool_kernel   25-tap conv + 2x2 max-pool + activation on the VALU (1.7 % of the MACs)   -> P1 [S][N][32][12][12]
//   conv2_pool_kernel   the 98 % of the work: implicit GEMM on v_mfma_f32_16x16x4_f32,
//                       O2^T[hc][pos] = sum_k W2[hc][k] * P1[ci(k)][y(pos)+ky(k)][x(pos)+kx(k)],  k = (ci,ky,kx) in 0..799;
//                       A = W2 rows through an LDS-DMA tile ring, B gathered from the point's P1 image resident in LDS
//                       with a k -> offset table; epilogue: bias, 2x2/stride-1 max-pool on the pre-activations,
//                       activation, 3-bit stash (argmax, sign) for the backward                    -> Q2 [S][N][Hc*49]
//   conv_fc_kernel      the skinny Linear(49*Hc -> C) as an MFMA with both operands K-contiguous in memory, + softmax
// max-pool commutes with the (monotone) relu / leaky-relu, so pooling the pre-activation and activating once is exact.
#include "rbnn_common.hpp"

namespace {

constexpr int C1 = 32, O1 = 24, P1W = 12, P1SZ = C1 * P1W * P1W;      // conv1 channels, its output width, pooled width, floats per point
constexpr int K2 = C1 * 25, O2W = 8, P2W = 7, NPOS = O2W * O2W, NP2 = P2W * P2W;

struct ConvArgs {
    const float* X; int ldx; int N;
    const float* K1w; const float* K1b;            // [S_total][32][25], [S_total][32]
    const float* K2w; const float* K2b;            // [S_total][Hc][800], [S_total][Hc]
    const float* Fw;  const float* Fb;             // [S_total][C][49*Hc], [S_total][C]
    int Hc; int C; const int* sidx; int S;
    float* P1; uint8_t* st1;                       // [S][N][4608]
    float* Q2; uint8_t* st2;                       // [S][N][Hc*49]
    float* P; int out_kind;
};

// ---------------------------------------------------------------------------------------------------
template <int ACT>
__global__ void __launch_bounds__(256) conv1_pool_kernel(const ConvArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one thread per pooled output (s, n, c, py, px)
    if (i >= (long long)a.S * a.N * P1SZ) return;
    const int e = (int)(i % P1SZ), c = e / (P1W * P1W), py = (e / P1W) % P1W, px = e % P1W;
    const long long sn = i / P1SZ;
    const int n = (int)(sn % a.N), s = (int)(sn / a.N);
    const int sw = a.sidx ? a.sidx[s] : s;
    const float* const x = a.X + (long long)n * a.ldx + (2 * py) * 28 + 2 * px;
    const float* const w = a.K1w + ((long long)sw * C1 + c) * 25;
    float patch[6][6];
#pragma unroll
    for (int y = 0; y < 6; ++y)
#pragma unroll
        for (int xx = 0; xx < 6; ++xx) patch[y][xx] = x[y * 28 + xx];
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
            if ((dy == 0 && dx == 0) || v > best) { best = v; arg = dy * 2 + dx; }   // first maximum wins (torch max_pool2d)
        }
    a.P1[i] = act_fwd<ACT>(best);
    a.st1[i] = (uint8_t)(arg | (best > 0.f ? 4 : 0));
}

// ---------------------------------------------------------------------------------------------------
// conv2: one block = one (sample, point); 4 waves, each 64 output channels x the point's 64 output positions
// (4 x 4 accumulator tiles), output channels in chunks of 256.
template <int ACT>
__global__ void __launch_bounds__(256, 2) conv2_pool_kernel(const ConvArgs a) {
    constexpr int WROWS = 256, TILE = WROWS * 16;
    __shared__ __attribute__((aligned(16))) float lds[P1SZ + 800 + 2 * TILE + 4 * 16 * NPOS];
    float* const P1s = lds;
    int* const koff = (int*)(lds + P1SZ);
    float* const Wt = lds + P1SZ + 800;
    float* const scr = Wt + 2 * TILE;

    int id;
    if (!item_of_block(blockIdx.x, a.N * a.S, id)) return;
    const int n = id % a.N, s = id / a.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    const float* const Ws = a.K2w + (long long)sw * a.Hc * K2;
    const int F = a.Hc * NP2;

    // the point's pooled conv1 image -> LDS (18 pieces of 1 KiB), and the k -> image offset table
    for (int q = wave; q < P1SZ / 256; q += 4) glds16(a.P1 + sn * P1SZ + q * 256 + 4 * lane, P1s + q * 256);
    for (int k = tid; k < K2; k += 256) koff[k] = (k / 25) * (P1W * P1W) + ((k % 25) / 5) * P1W + (k % 5);
    const int prow = lane >> 2, lchunk = (lane & 3) ^ swz(prow), pch = 4 * (lg ^ swz(li));
    int poff[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) poff[pt] = (2 * pt + (li >> 3)) * P1W + (li & 7);   // position pt*16+li = (y, x) = (2pt + li/8, li%8)

    for (int hc0 = 0; hc0 < a.Hc; hc0 += WROWS) {
        f32x4 acc[4][4];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
            f32x4 b[4], af[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
                b[pt] = (f32x4){P1s[kq0 + poff[pt]], P1s[kq1 + poff[pt]], P1s[kq2 + poff[pt]], P1s[kq3 + poff[pt]]};
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) af[ht] = *(const f32x4*)(W + ((wave * 4 + ht) * 16 + li) * 16 + pch);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) acc[ht][pt] = MFMA16(af[ht][r], b[pt][r], acc[ht][pt]);
            __syncthreads();
        }
        // epilogue: bias, 2x2 / stride-1 max-pool of the pre-activations through a per-wave LDS tile, activation, stash
        float* const my = scr + wave * 16 * NPOS;
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const int hcb = hc0 + (wave * 4 + ht) * 16;                    // wave-uniform
            if (hcb >= a.Hc) break;
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r) my[(4 * lg + r) * NPOS + pt * 16 + li] = acc[ht][pt][r] + bias[r];
            for (int idx = lane; idx < 16 * NP2; idx += 64) {
                const int hl = idx / NP2, p = idx % NP2, base = hl * NPOS + (p / P2W) * O2W + (p % P2W);
                float best = my[base];
                int arg = 0;
                if (my[base + 1] > best) { best = my[base + 1]; arg = 1; }
                if (my[base + O2W] > best) { best = my[base + O2W]; arg = 2; }
                if (my[base + O2W + 1] > best) { best = my[base + O2W + 1]; arg = 3; }
                const long long o = sn * F + (long long)(hcb + hl) * NP2 + p;
                a.Q2[o] = act_fwd<ACT>(best);
                a.st2[o] = (uint8_t)(arg | (best > 0.f ? 4 : 0));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Linear(49*Hc -> C) + softmax: one wave = 16 points of one sample.  D[i = class][j = point] = sum_f Fw[c][f] * Q2[n][f];
// both operands are read straight from memory, 16 bytes (4 K steps) per lane per load.
__global__ void __launch_bounds__(256) conv_fc_kernel(const ConvArgs a) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int NT = (a.N + 15) / 16;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= NT * a.S) return;
    const int s = item / NT, n0 = (item % NT) * 16;
    const int sw = a.sidx ? a.sidx[s] : s;
    const int F = a.Hc * NP2;
    const int n = min(n0 + li, a.N - 1), c = min(li, a.C - 1);
    const float* const fw = a.Fw + ((long long)sw * a.C + c) * F + 4 * lg;
    const float* const q2 = a.Q2 + ((long long)s * a.N + n) * F + 4 * lg;
    const float cmask = li < a.C ? 1.f : 0.f;
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < F; f += 16) {                                      // F = 49*Hc is a multiple of 16
        const f32x4 av = *(const f32x4*)(fw + f) * cmask, bv = *(const f32x4*)(q2 + f);
#pragma unroll
        for (int r = 0; r < 4; ++r) z = MFMA16(av[r], bv[r], z);
    }
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

int validate_conv(const rbnn_conv_posterior* net) {
    if (!net || !net->K1w || !net->K1b || !net->K2w || !net->K2b || !net->Fw || !net->Fb) return RBNN_ERR_NULL;
    if (net->activation != RBNN_ACT_RELU && net->activation != RBNN_ACT_LEAKY) return RBNN_ERR_UNSUPPORTED;
    if (net->hidden < 16 || (net->hidden & 15) || net->n_classes < 1 || net->n_classes > RBNN_CPAD || net->n_stored < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(net->K2w) || !aligned16(net->K2b) || !aligned16(net->Fw)) return RBNN_ERR_ALIGN;
    return RBNN_OK;
}

}  // namespace

extern "C" {

int rbnn_conv_workspace_query(const rbnn_conv_posterior* net, int32_t N, int32_t S, rbnn_conv_workspace_sizes* out) {
    if (!net || !out) return RBNN_ERR_NULL;
    if (net->hidden < 16 || (net->hidden & 15) || N < 1 || S < 1) return RBNN_ERR_SHAPE;
    const size_t SN = (size_t)S * N, F = (size_t)net->hidden * NP2;
    rbnn_conv_workspace_sizes z = {};
    z.P = z.dZ = SN * RBNN_CPAD * sizeof(float);
    z.P1 = SN * P1SZ * sizeof(float);
    z.st1 = SN * P1SZ;
    z.Q2 = SN * F * sizeof(float);
    z.st2 = SN * F;
    z.G = SN * 784 * sizeof(float);
    *out = z;
    return RBNN_OK;
}

int rbnn_conv_forward(const rbnn_conv_posterior* net, const float* X, int32_t ldx, int32_t N, const int32_t* sidx, int32_t S,
                      int32_t out_kind, const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!X || !ws || !ws->P || !ws->P1 || !ws->st1 || !ws->Q2 || !ws->st2) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || ldx < 784) return RBNN_ERR_SHAPE;
    if (out_kind != RBNN_OUT_PROBS && out_kind != RBNN_OUT_LOGITS) return RBNN_ERR_UNSUPPORTED;
    if (!aligned16(ws->P) || !aligned16(ws->P1) || !aligned16(ws->Q2)) return RBNN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a = {};
    a.X = X; a.ldx = ldx; a.N = N;
    a.K1w = net->K1w; a.K1b = net->K1b; a.K2w = net->K2w; a.K2b = net->K2b; a.Fw = net->Fw; a.Fb = net->Fb;
    a.Hc = net->hidden; a.C = net->n_classes; a.sidx = sidx; a.S = S;
    a.P1 = ws->P1; a.st1 = ws->st1; a.Q2 = ws->Q2; a.st2 = ws->st2; a.P = ws->P; a.out_kind = out_kind;
    const long long t1 = (long long)S * N * P1SZ;
    const bool leaky = net->activation == RBNN_ACT_LEAKY;
    if (leaky) hipLaunchKernelGGL(conv1_pool_kernel<RBNN_ACT_LEAKY>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, a);
    else       hipLaunchKernelGGL(conv1_pool_kernel<RBNN_ACT_RELU>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, a);
    if ((rc = launch_status())) return rc;
    const int grid = grid_for_items((long long)N * S);
    if (leaky) hipLaunchKernelGGL(conv2_pool_kernel<RBNN_ACT_LEAKY>, dim3(grid), dim3(256), 0, st, a);
    else       hipLaunchKernelGGL(conv2_pool_kernel<RBNN_ACT_RELU>, dim3(grid), dim3(256), 0, st, a);
    if ((rc = launch_status())) return rc;
    const int items = ((N + 15) / 16) * S;
    hipLaunchKernelGGL(conv_fc_kernel, dim3((items + 3) / 4), dim3(256), 0, st, a);
    return launch_status();
}

}  // extern "C"

// =====================================================================================================
// Backward to the input.  One block = one (sample, point); everything between dZ and dX stays in LDS:
//   1. per chunk of 64 output channels: dQ2[f] = sum_c dZ[c] * Fw[c][f]  (Linear^T), routed through the pool-2
//      argmax and the activation derivative into dO2t[64 hc][64 positions]  (LDS float atomics; 2x2 windows overlap);
//   2. conv2^T as 25 tap-GEMMs on the matrix pipe:  T_t[ci][pos] = sum_hc W2[hc][ci][tap t] * dO2t[hc][pos]
//      (A = the tap-major image K2w_tap[t][ci][hc], LDS-DMA double buffer; B = dO2t), scatter-added at the tap's
//      shift into dP1s[32][12][12];
//   3. pool-1 routing + activation derivative, and conv1^T as a scatter of 25 taps per pooled element into dXs[28][28].
// =====================================================================================================
namespace {

struct ConvBwdArgs {
    const float* dZ; const uint8_t* st1; const uint8_t* st2;
    const float* K1w; const float* K2tap; const float* Fw;
    int Hc; int C; int N; int S; const int* sidx; int act;
    float* G;                                                            // [S][N][784]
};

template <int ACT>
__global__ void __launch_bounds__(256, 2) conv_bwd_kernel(const ConvBwdArgs a) {
    constexpr int HCH = 64, DLD = 68;                                    // channels per chunk; dO2t row stride (4 rows apart = 16 banks apart)
    constexpr int WT = C1 * HCH;                                         // floats per tap tile [32 ci][64 hc]
    __shared__ __attribute__((aligned(16))) float lds[P1SZ + HCH * DLD + 2 * WT + 784];
    float* const dP1s = lds;
    float* const dO2t = lds + P1SZ;
    float* const Wt = dO2t + HCH * DLD;
    float* const dXs = Wt + 2 * WT;

    int id;
    if (!item_of_block(blockIdx.x, a.N * a.S, id)) return;
    const int n = id % a.N, s = id / a.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int sw = a.sidx ? a.sidx[s] : s;
    const long long sn = (long long)s * a.N + n;
    const int F = a.Hc * NP2;
    const float slope = ACT == RBNN_ACT_RELU ? 0.f : LEAKY_SLOPE;

    for (int i = tid; i < P1SZ; i += 256) dP1s[i] = 0.f;
    for (int i = tid; i < 784; i += 256) dXs[i] = 0.f;
    float dz[RBNN_CPAD];
#pragma unroll
    for (int c = 0; c < RBNN_CPAD; ++c) dz[c] = a.dZ[sn * RBNN_CPAD + c];  // zero beyond C
    // tap tile DMA: piece q = rows 4q..4q+3 (256 B each); lane p -> row 4q + (p>>4), physical 16-B chunk p&15 holds
    // logical chunk (p&15) ^ (row&15): a b128 fragment read (16 rows, one chunk) then hits 16 distinct slots.
    const int trow = lane >> 4, tchunk = lane & 15;
    const int y0 = 2 * wave + (li >> 3), x0 = li & 7;                     // this lane's output position in pos tile `wave`

    for (int hc0 = 0; hc0 < a.Hc; hc0 += HCH) {
        __syncthreads();                                                 // previous chunk's GEMMs are done with dO2t / Wt
        for (int i = tid; i < HCH * DLD; i += 256) dO2t[i] = 0.f;
        __syncthreads();
        // 1. Linear^T + pool-2 routing for channels hc0 .. hc0+63
        const int nch = min(HCH, a.Hc - hc0);
        for (int e = tid; e < nch * NP2; e += 256) {
            const long long f = (long long)hc0 * NP2 + e;
            float dq = 0.f;
#pragma unroll
            for (int c = 0; c < RBNN_CPAD; ++c)
                if (c < a.C) dq = fmaf(dz[c], a.Fw[((long long)sw * a.C + c) * F + f], dq);
            const int st = a.st2[sn * F + f], hl = e / NP2, p = e % NP2;
            const int pos = (p / P2W + ((st >> 1) & 1)) * O2W + (p % P2W) + (st & 1);
            atomicAdd(&dO2t[hl * DLD + pos], (st & 4) ? dq : dq * slope);
        }
        auto stage = [&](int t, int buf) {                               // tap t: K2tap[s][t][ci][hc0 .. hc0+63]
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = wave + 4 * q2, row = 4 * q + trow;         // ci
                const int lch = tchunk ^ (row & 15);
                const int hc = min(hc0 + 4 * lch, a.Hc - 4);             // columns past Hc: any valid address (their dO2t rows are zero)
                glds16(a.K2tap + (((long long)sw * 25 + t) * C1 + row) * a.Hc + hc, Wt + buf * WT + q * 256);
            }
        };
        stage(0, 0);
        __syncthreads();                                                 // dO2t complete, tap 0 landed
        for (int t = 0; t < 25; ++t) {
            const int buf = t & 1;
            if (t + 1 < 25) stage(t + 1, buf ^ 1);
            const float* const W = Wt + buf * WT;
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int kk = 0; kk < HCH / 16; ++kk) {
                const int ch = 4 * kk + lg;                              // logical 16-B chunk = hidden units 16kk + 4lg .. +3
                const f32x4 a0 = *(const f32x4*)(W + li * HCH + 4 * (ch ^ (li & 15)));
                const f32x4 a1 = *(const f32x4*)(W + (16 + li) * HCH + 4 * (ch ^ ((16 + li) & 15)));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float b = dO2t[(16 * kk + 4 * lg + r) * DLD + wave * 16 + li];
                    acc0 = MFMA16(a0[r], b, acc0);
                    acc1 = MFMA16(a1[r], b, acc1);
                }
            }
            // acc0[r] = T_t[ci = 4lg + r][pos], acc1[r] = T_t[16 + 4lg + r][pos]; conv2^T: dP1[ci][y + ky][x + kx] += T_t
            const int tgt = (y0 + t / 5) * P1W + x0 + t % 5;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                atomicAdd(&dP1s[(4 * lg + r) * (P1W * P1W) + tgt], acc0[r]);
                atomicAdd(&dP1s[(16 + 4 * lg + r) * (P1W * P1W) + tgt], acc1[r]);
            }
            __syncthreads();                                             // next tap landed (vmcnt(0)); this one is free
        }
    }
    __syncthreads();
    // 3. pool-1 routing + conv1^T (in_channels = 1): scatter 25 taps per pooled element
    for (int e = tid; e < P1SZ; e += 256) {
        const int st = a.st1[sn * P1SZ + e], c = e / (P1W * P1W), py = (e / P1W) % P1W, px = e % P1W;
        const float g = (st & 4) ? dP1s[e] : dP1s[e] * slope;
        if (g == 0.f) continue;
        const int Y = 2 * py + ((st >> 1) & 1), X = 2 * px + (st & 1);
        const float* const w = a.K1w + ((long long)sw * C1 + c) * 25;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) atomicAdd(&dXs[(Y + ky) * 28 + X + kx], g * w[ky * 5 + kx]);
    }
    __syncthreads();
    for (int i = tid; i < 784; i += 256) a.G[sn * 784 + i] = dXs[i];
}

}  // namespace

extern "C" int rbnn_conv_input_grad(const rbnn_conv_posterior* net, const int32_t* sidx, int32_t S, int32_t N,
                                    const rbnn_conv_workspace* ws, void* stream) {
    int rc = validate_conv(net);
    if (rc) return rc;
    if (!net->K2w_tap || !ws || !ws->dZ || !ws->st1 || !ws->st2 || !ws->G) return RBNN_ERR_NULL;
    if (N < 1 || S < 1 || (net->hidden & 3)) return RBNN_ERR_SHAPE;
    if (!aligned16(net->K2w_tap)) return RBNN_ERR_ALIGN;
    ConvBwdArgs a = {};
    a.dZ = ws->dZ; a.st1 = ws->st1; a.st2 = ws->st2; a.K1w = net->K1w; a.K2tap = net->K2w_tap; a.Fw = net->Fw;
    a.Hc = net->hidden; a.C = net->n_classes; a.N = N; a.S = S; a.sidx = sidx; a.G = ws->G;
    const int grid = grid_for_items((long long)N * S);
    if (net->activation == RBNN_ACT_LEAKY) hipLaunchKernelGGL(conv_bwd_kernel<RBNN_ACT_LEAKY>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else                                   hipLaunchKernelGGL(conv_bwd_kernel<RBNN_ACT_RELU>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}
