// Diagnostic: sustained rate of the two f16 MFMA shapes on random operands (registers only, no LDS / memory in the loop).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/mfma_shape_bench.hip && /tmp/mfma_shape
// Question it answers: these kernels are power-limited (~2.0 GHz under the triple kernels) — does v_mfma_f32_32x32x16_f16 (half the
// operand-register reads per MAC) sustain a higher rate than v_mfma_f32_16x16x32_f16?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFF]; }
    float sum = 0.f;
    if (SHAPE == 16 || SHAPE == 17) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if (SHAPE == 16) {                                 // A operand stable over 4 consecutive MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {                                           // SHAPE 17: both operands change on every MFMA (diagonal order)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int i = t % 4, j = (t + t / 4) % 4;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // same MACs per iteration as the 16x16x32 arm: 2 x 2 tiles of 32 x 32, two K = 16 steps each
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * k + i], b[2 * k + j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    }
    if (sum == 12345.678f) out[tid] = sum;
}

int main() {
    const int n = 1 << 16;
    std::vector<_Float16> h(n * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) - (1 << 23)) / (float)(1 << 23) * 100.f); }
    f16x8* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 24);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 2;                 // two blocks (8 waves) per CU
    for (int shape : {16, 17, 32, 16, 17, 32}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            else if (shape == 17) hipLaunchKernelGGL(mfma_loop<17>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 16 * 8192 * (double)iters * blocks * 4;   // 16 MFMAs of 8192 MACs (or 8 of 16384) per wave-iteration
            if (rep == 2) printf("v_mfma_f32_%s_f16: %.2f ms  %.0f TFLOP/s  (%.1f %% of 2516.6)\n", shape == 16 ? "16x16x32 (A stable x4)" : shape == 17 ? "16x16x32 (A, B change every MFMA)" : "32x32x16", ms, flop / ms / 1e9, flop / ms / 1e9 / 25.166);
        }
    }
    return 0;
}
