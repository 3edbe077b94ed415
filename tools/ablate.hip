// Diagnostic: time the two GEMM kernels of the C2 workload with parts of their K loops switched off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD -DRBNN_ABL=<bits> -o /tmp/ablate_<bits> tools/ablate.hip
// Results of RBNN_ABL != 0 builds are wrong by construction; only the timings mean anything.
#include "../robustbnns_amd/csrc/rbnn_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

static float* dev_rand(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = scale * ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23); }
    float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice); return d;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000, S = argc > 2 ? atoi(argv[2]) : 100, reps = 5;
    const int D = 784, H = 512, C = 10;
    rbnn_posterior net = {};
    net.arch = RBNN_ARCH_FC; net.activation = RBNN_ACT_LEAKY; net.in_features = D; net.in_stride = D; net.hidden = H; net.n_classes = C; net.n_stored = S;
    net.W1 = dev_rand((size_t)S * H * D, 0.05f, 1); net.b1 = dev_rand((size_t)S * H, 0.05f, 2);
    net.W2 = dev_rand((size_t)S * C * H, 0.05f, 3); net.b2 = dev_rand((size_t)S * C, 0.05f, 4);
    { float* p4; hipMalloc(&p4, (size_t)S * H * D * sizeof(float)); rbnn_pack_rows4(net.W1, (int64_t)S * H, D, p4, nullptr); net.W1_pack4 = p4; }
    float* X = dev_rand((size_t)N * D, 0.5f, 5);
    rbnn_workspace_sizes sz; rbnn_workspace_query(&net, N, S, 0, &sz);
    rbnn_workspace ws = {};
    hipMalloc(&ws.P, sz.P); hipMalloc(&ws.dZ, sz.dZ); hipMalloc(&ws.mask1, sz.mask1); hipMalloc(&ws.slabs, sz.slabs);
    hipMemset(ws.dZ, 0, sz.dZ);
    { float* t = dev_rand((size_t)S * N * 16, 0.01f, 6); hipMemcpy(ws.dZ, t, sz.dZ, hipMemcpyDeviceToDevice); hipFree(t); }
    {
        int nf = 0, ng = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, (const void*)fc_forward_kernel<RBNN_ACT_LEAKY, 4, 8, 1, 4, true>, 256, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&ng, (const void*)fc_grad_kernel<RBNN_ACT_LEAKY, 7, 3, false, false, 32>, 256, 0);
        printf("occupancy API: fwd %d blocks/CU, grad %d blocks/CU; chunk %d n_slabs %d\n", nf, ng, sz.chunk, sz.n_slabs);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * N * S * ((double)D * H + (double)H * C);
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps + 1; ++r) {
            hipEventRecord(e0, 0);
            int rc = which == 0 ? rbnn_fc_forward(&net, X, D, N, nullptr, S, RBNN_OUT_PROBS, &ws, nullptr)
                                : rbnn_fc_input_grad(&net, nullptr, S, N, sz.chunk, &ws, nullptr, nullptr);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            if (rc) { printf("rc=%d\n", rc); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r) { sum += ms; best = ms < best ? ms : best; }
        }
        printf("ABL=%d %-14s avg %.3f ms  best %.3f ms  -> %.1f TFLOP/s (algorithmic) %.1f%% of 157.3\n", RBNN_ABL,
               which == 0 ? "fc_forward" : "fc_input_grad", sum / reps, best, flop / (best * 1e-3) / 1e12, 100 * flop / (best * 1e-3) / 1e12 / 157.3);
    }
    return 0;
}
