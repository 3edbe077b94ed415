// Diagnostic: time the split-precision GEMM kernels of the C2 workload (variants via -D flags; see tools/run_variants_split.sh).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD <-D...> -o /tmp/abls tools/ablate_split.hip
#include "../robustbnns_amd/csrc/rbnn_kernels.hip"
#include "../robustbnns_amd/csrc/rbnn_split.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

static float* dev_rand(size_t n, float scale, unsigned seed, bool positive = false) {
    std::vector<float> h(n);
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = scale * ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23); if (positive) h[i] = fabsf(h[i]); }
    float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice); return d;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000, S = argc > 2 ? atoi(argv[2]) : 100, reps = 6;
    const int D = 784, H = 512, C = 10, LD = 800;
    rbnn_posterior net = {};
    net.arch = RBNN_ARCH_FC; net.activation = RBNN_ACT_LEAKY; net.in_features = D; net.in_stride = D; net.hidden = H; net.n_classes = C; net.n_stored = S;
    net.W1 = dev_rand((size_t)S * H * D, 0.2f, 1); net.b1 = dev_rand((size_t)S * H, 0.05f, 2);
    net.W2 = dev_rand((size_t)S * C * H, 0.2f, 3); net.b2 = dev_rand((size_t)S * C, 0.05f, 4);
    float* X = dev_rand((size_t)N * D, 1.0f, 5, true);
    rbnn_split_images sp = {};
    void *w1r, *w1c, *w2g, *xs;
    hipMalloc(&w1r, (size_t)S * H * LD * 4); hipMalloc(&w1c, (size_t)S * H * D * 4); hipMalloc(&w2g, (size_t)S * (H / 16) * 1024); hipMalloc(&xs, (size_t)N * LD * 4);
    sp.w1_exp = 16; sp.w2_exp = 16; sp.ld_rows = LD; sp.ld_cols = D;
    rbnn_split_rows(net.W1, (int64_t)S * H, D, D, sp.w1_exp, nullptr, w1r, LD, nullptr);
    rbnn_split_cols(net.W1, S, H, D, D, sp.w1_exp, w1c, D, nullptr);
    rbnn_split_w2gen(net.W2, S, C, H, sp.w2_exp, w2g, nullptr);
    rbnn_split_rows(X, N, D, D, 14, nullptr, xs, LD, nullptr);
    sp.W1_rows = w1r; sp.W1_cols = w1c; sp.W2_gen = w2g;
    rbnn_workspace_sizes sz; rbnn_workspace_query(&net, N, S, 0, &sz);
    rbnn_workspace ws = {};
    hipMalloc(&ws.P, sz.P); hipMalloc(&ws.dZ, sz.dZ); hipMalloc(&ws.mask1, sz.mask1); hipMalloc(&ws.slabs, sz.slabs);
    { float* t = dev_rand((size_t)S * N * 16, 0.01f, 6); hipMemcpy(ws.dZ, t, sz.dZ, hipMemcpyDeviceToDevice); hipFree(t); }
    rbnn_split_workspace_sizes ssz; rbnn_split_workspace_query(&net, &sp, N, S, &ssz);
    rbnn_split_workspace sws = {}; sws.X_split = xs; hipMalloc(&sws.dZ_gen, ssz.dZ_gen); hipMalloc((void**)&sws.g_scale, ssz.g_scale);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * N * S * ((double)D * H + (double)H * C);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps + 1; ++r) {
            int ns = 0;
            hipEventRecord(e0, nullptr);
            const int rc = which == 0 ? rbnn_fc_forward_split(&net, &sp, xs, LD, 14, nullptr, N, nullptr, S, RBNN_OUT_PROBS, &ws, nullptr)
                                      : rbnn_fc_input_grad_split(&net, &sp, nullptr, S, N, 0, &ws, &sws, &ns, nullptr);
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            if (rc) { printf("rc=%d (%s)\n", rc, rbnn_strerror(rc)); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r) { best = fminf(best, ms); sum += ms; }
        }
        printf("%s avg %.3f ms  best %.3f ms  -> %.1f TFLOP/s fp32-equivalent (%.0f on the f16 pipe = %.1f%% of 2516.6)\n",
               which ? "fc_grad_split (+split_dz)" : "fc_forward_split        ", sum / reps, best, flop / best / 1e9, 3 * flop / best / 1e9, 300 * flop / best / 1e9 / 2516.6);
    }
    return 0;
}
