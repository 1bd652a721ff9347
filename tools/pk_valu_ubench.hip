// Micro-benchmark: cost of packed-f32 vector instructions (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32) next to
// v_mfma_f32_16x16x4_f32, in the shape wino32.hip runs: 32 MFMAs back to back, then a burst of NV vector instructions,
// two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/pk_valu_ubench.hip -o tools/bin/pk_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND>  // KIND 0 v_add_f32, 1 v_pk_add_f32, 2 v_fma_f32, 3 v_pk_fma_f32, 4 v_pk_mul_f32, 5 v_max_f32, 6 v_pk_mov_b32
__global__ void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 v[8], pa = {a, b};
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j % 8].x) : "v"(a));
            if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(pa));
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 8].x) : "v"(a), "v"(b));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[j % 8]) : "v"(pa));
            if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(pa));
            if (KIND == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j % 8].x) : "v"(a));
            if (KIND == 6) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(v[j % 8]) : "v"(pa));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int KIND>
void run(int threads, const char* name) {
    float* out; long long* cyc;
    const int grid = 256, iters = 500;
    hipMalloc(&out, grid * threads * 4); hipMalloc(&cyc, grid * 16 * 8);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    printf("%-8s waves/SIMD %d  NV %3d : %.0f cycles per (32 MFMA + burst) per wave\n", name, threads / 256, NV, s / h.size() / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512}) {
        run<0, 0>(th, "none");
        run<64, 0>(th, "add"); run<64, 1>(th, "pk_add"); run<64, 2>(th, "fma"); run<64, 3>(th, "pk_fma"); run<64, 4>(th, "pk_mul");
        run<64, 5>(th, "max"); run<64, 6>(th, "pk_mov"); run<128, 0>(th, "add"); run<128, 1>(th, "pk_add");
    }
    return 0;
}
