// Micro-benchmark: do the vector instructions of one wave run under the MFMAs of the OTHER wave on the same SIMD?  Two waves
// per SIMD, each alternating 32 v_mfma_f32_16x16x4_f32 and a burst of NV v_pk_add_f32.  Started together the two waves stay in
// phase (both in their MFMA run, then both in their vector burst) and nothing overlaps; with the second wave of every SIMD
// started `skew` cycles late they alternate.  hipcc --offload-arch=gfx950 -O3 tools/phase_ubench.hip -o tools/bin/phase_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>
__global__ void k(float* out, long long* cyc, int iters, int skew) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 v[8], pa = {a, b};
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    if (threadIdx.x >= 256) {  // waves 4..7: the second wave of each SIMD
        const long long s = clock64();
        while (clock64() - s < skew) {}
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(pa));
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV>
void run(int skew) {
    float* out; long long* cyc;
    const int grid = 256, iters = 500, threads = 512;
    (void)hipMalloc(&out, grid * threads * 4); (void)hipMalloc(&cyc, grid * 16 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, skew);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid * (threads / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    printf("NV %3d  skew %5d : %.0f cycles per (32 MFMA + burst) per wave\n", NV, skew, s / h.size() / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int skew : {0, 300, 600, 900, 1200, 1500}) { run<64>(skew); run<128>(skew); run<200>(skew); }
    return 0;
}
