// rbnn_conv_common.hpp — what the translation units of the conv architecture share: the geometry descriptor, the kernels' argument records, the
// geometry / activation dispatch, and the host-side launchers of the fp32 kernels that EVERY precision mode runs (conv1, the skinny Linear and
// their transposes: rbnn_conv.hip defines them, rbnn_conv_x3.hip calls them).  Model: model_nn.py:93-106.
#pragma once
#include "rbnn_common.hpp"

namespace rbnn_conv_shared {

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

template <class G> constexpr int conv1_halves() { return G::CIN == 1 ? 2 : 1; }
template <class G> constexpr int conv1_threads() { return (conv1_halves<G>() * G::P1W * G::P1W + 63) / 64 * 64; }

struct ConvBwdArgs {
    const float* dZ; const uint8_t* st1; const uint8_t* st2;
    const float* K1w; const float* K2cb; const float* Fw;
    int Hc; int C; int N; int S; const int* sidx;
    float* dQ2;                                                          // [S][N][Hc*NP2] dL/d(pooled conv2 output) = dZ . Fw: aliases the forward's Q2
    float* dP1;                                                          // [S][N][P1SZ] dL/d(pooled conv1 output): aliases the forward's P1
    float* G;                                                            // [S][N][DIN]
    int NP2;                                                             // pooled conv2 positions per channel (conv_fc_bwd is geometry-agnostic)
};

inline int validate_conv(const rbnn_conv_posterior* net) {
    if (!net || !net->K1w || !net->K1b || !net->K2w || !net->K2b || !net->Fw || !net->Fb) return RBNN_ERR_NULL;
    if (net->activation < RBNN_ACT_RELU || net->activation > RBNN_ACT_TANH) return RBNN_ERR_UNSUPPORTED;
    if (!((net->in_channels == 1 && net->in_width == 28) || (net->in_channels == 3 && net->in_width == 32))) return RBNN_ERR_UNSUPPORTED;
    if (net->hidden < 16 || (net->hidden & 15) || net->n_classes < 1 || net->n_classes > RBNN_CPAD || net->n_stored < 1) return RBNN_ERR_SHAPE;
    if (!aligned16(net->K2w) || !aligned16(net->K2b) || !aligned16(net->Fw)) return RBNN_ERR_ALIGN;
    return RBNN_OK;
}
// the split-half conv kernels: 1x28x28, relu / leaky
inline int validate_conv_split(const rbnn_conv_posterior* net) {
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

// row pitch of the conv1^T kernels' T buffer: >= O1 with 4 * TS = 8 or 16 mod 32 (LDS banks)
template <class G> constexpr int conv1_bwd_ts() { return G::O1 % 8 == 0 ? G::O1 + 2 : G::O1; }

// Host-side launchers defined in rbnn_conv.hip (the kernels are instantiated there once): conv1 + pool (fp32 VALU), the Linear head + softmax,
// its transpose (dQ2 = dZ . Fw, act' folded in for sigmoid / tanh), and pool-1 routing + conv1^T on the fp32 matrix pipe.
#define RBNN_HIDDEN __attribute__((visibility("hidden")))
RBNN_HIDDEN int launch_conv1_pool(int act, int in_channels, const ConvArgs& a, hipStream_t st);
RBNN_HIDDEN int launch_conv_fc(const ConvArgs& a, hipStream_t st);
RBNN_HIDDEN int launch_conv_fc_bwd(int act, const ConvBwdArgs& a, hipStream_t st);
RBNN_HIDDEN int launch_conv1_bwd_fp32(int act, int in_channels, const ConvBwdArgs& a, hipStream_t st);

}  // namespace rbnn_conv_shared
