// How long after a run of hand-issued v_mfma_f32_16x16x4_f32 (inline asm: hipcc inserts no wait states) may a vector
// instruction read an accumulator?  N MFMAs back to back (each on its own accumulator), W wait states (s_nop 0 each), then
// v_mov reads of ALL accumulators; compared with a run that waits 2048 states.  Also: cycles the wave needs to get past
// the N MFMAs (is MFMA issue blocking, or is there a queue?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W>
__global__ void k(float* out, long long* cyc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + (threadIdx.x % 7) * 0.125f, b = 0.5f + (threadIdx.x % 5) * 0.25f;
    float r[8];
    long long t0, t1;
    for (int it = 0; it < 4; ++it) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        t0 = clock64();
        asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n\tv_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n\tv_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %8, %9, %6\n\tv_mfma_f32_16x16x4_f32 %7, %8, %9, %7"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
            : "v"(a), "v"(b));
        t1 = clock64();
#pragma unroll
        for (int n = 0; n < W; ++n) asm volatile("s_nop 0");
        // read element 0 of every accumulator with explicit v_mov (the compiler sees only the asm outputs)
        asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\tv_mov_b32 %4, %12\n\t"
                     "v_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                     : "v"(acc[0].x), "v"(acc[1].x), "v"(acc[2].x), "v"(acc[3].x), "v"(acc[4].x), "v"(acc[5].x), "v"(acc[6].x), "v"(acc[7].x));
    }
    for (int i = 0; i < 8; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + i] = r[i];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int W>
std::vector<float> run(int th, double* cycles) {
    float* out; long long* cyc;
    const int grid = 256;
    hipMalloc(&out, grid * th * 8 * 4); hipMalloc(&cyc, grid * 16 * 8);
    hipLaunchKernelGGL((k<W>), dim3(grid), dim3(th), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<float> h(grid * th * 8); std::vector<long long> c(grid * (th / 64));
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : c) s += x; *cycles = s / c.size();
    hipFree(out); hipFree(cyc);
    return h;
}

template <int W>
void test(int th, const std::vector<float>& ref) {
    double cy; auto x = run<W>(th, &cy);
    size_t bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < x.size(); ++i) if (x[i] != ref[i]) ++bad[i % 8];
    printf("waves/SIMD %d  W %4d : stale reads per accumulator (first .. last MFMA):", th / 256, W);
    for (int i = 0; i < 8; ++i) printf(" %zu", bad[i]);
    printf("   | cycles to get past the 8 MFMAs %.0f\n", cy);
}

int main() {
    for (int th : {256, 512}) {
        double cy; auto ref = run<2048>(th, &cy);
        test<0>(th, ref); test<4>(th, ref); test<8>(th, ref); test<16>(th, ref); test<32>(th, ref); test<64>(th, ref);
        test<128>(th, ref); test<256>(th, ref); test<512>(th, ref);
    }
    return 0;
}
