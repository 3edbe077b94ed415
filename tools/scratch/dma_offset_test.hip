// does the immediate offset of global_load_lds apply to BOTH the global and the LDS address?  (prints yes/no)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const float* g = src + threadIdx.x * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)lds, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)lds, 16, 2048, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 2048 * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    // expected if the offset applies to both: lds[256 p + i] = src[256 p + i] for p = 0, 1, 2
    bool both = true, lds_only = true, glob_only = true;
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 256; ++i) {
        both &= r[256 * p + i] == (float)(256 * p + i);
        lds_only &= r[256 * p + i] == (float)i;
    }
    for (int i = 0; i < 256; ++i) glob_only &= r[i] == (float)(512 + i);   // the last write wins at LDS offset 0
    printf("offset applies to both: %s; LDS only: %s; global only: %s; r[0]=%g r[256]=%g r[512]=%g r[768]=%g\n", both ? "yes" : "no", lds_only ? "yes" : "no", glob_only ? "yes" : "no", r[0], r[256], r[512], r[768]);
    return 0;
}
