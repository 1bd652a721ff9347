// Micro-benchmark: how many vector instructions does one v_mfma_f32_16x16x4_f32 (32 cycles on the SIMD's matrix pipe) cover
// when they sit between two MFMAs of the same wave?  hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>  // NV vector instructions behind every MFMA; KIND 0: v_fma_f32, 1: v_add_f32, 2: v_mov_b32, 3: ds_read_b128 every 4th MFMA + v_add
__global__ void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    __shared__ float lds[4096];
    lds[threadIdx.x] = a;
    __syncthreads();
    f32x4 lv = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
            if (KIND == 3 && (m & 3) == 0) lv = *reinterpret_cast<volatile f32x4*>(&lds[(threadIdx.x & 63) * 4 + m * 256]);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 8]) : "v"(a), "v"(b));
                if (KIND == 1 || KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(a));
                if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j % 8]) : "v"(a));
            }
        }
    }
    const long long t1 = clock64();
    float s = lv.x;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int KIND>
void run(int threads, const char* name) {
    float* out; long long* cyc;
    const int grid = 256, iters = 2000;
    hipMalloc(&out, grid * threads * 4); hipMalloc(&cyc, grid * 16 * 8);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    printf("%-10s waves/SIMD %d  NV %d : %.1f cycles per MFMA per wave\n", name, threads / 256, NV, s / h.size() / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512}) {
        run<0, 0>(th, "none"); run<2, 0>(th, "fma"); run<4, 0>(th, "fma"); run<5, 0>(th, "fma"); run<6, 0>(th, "fma"); run<8, 0>(th, "fma");
        run<4, 1>(th, "add"); run<6, 1>(th, "add"); run<4, 2>(th, "mov"); run<6, 2>(th, "mov"); run<4, 3>(th, "ds+add"); run<6, 3>(th, "ds+add");
    }
    return 0;
}
