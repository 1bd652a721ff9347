// Write-after-read on the operands of hand-issued MFMAs (inline asm: hipcc's hazard handling does not see them).
// NM v_mfma_f32_16x16x4_f32 back to back on operand register b (or a), then NOPS wait states, then `v_mov b, junk`;
// the reference never touches b.  Prints how many lanes differ.  Measured on MI355X: see wino32.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NOPS, int WHICH>  // NOPS < 0: reference
__global__ void k(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a0 = 1.0f + (threadIdx.x % 7) * 0.125f, b0 = 0.5f + (threadIdx.x % 5) * 0.25f;
    float a = a0, b = b0, junk = 1000.f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (NM == 8)
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n\t"
                "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n\tv_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n\t"
                "v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n\tv_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n\t"
                "v_mfma_f32_16x16x4_f32 %6, %8, %9, %6\n\tv_mfma_f32_16x16x4_f32 %7, %8, %9, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                : "v"(a), "v"(b));
        else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
        if (NOPS >= 0) {
#pragma unroll
            for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 0");
            if (WHICH == 0) asm volatile("v_mov_b32 %0, %1" : "+v"(b) : "v"(junk));
            else asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(junk));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NOPS, int WHICH>
std::vector<float> run(int th) {
    float* out;
    const int grid = 256, iters = 1;  // one pass: the overwritten operand is never needed again
    hipMalloc(&out, grid * th * 4);
    hipLaunchKernelGGL((k<NM, NOPS, WHICH>), dim3(grid), dim3(th), 0, 0, out, iters);
    hipDeviceSynchronize();
    std::vector<float> h(grid * th);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    hipFree(out);
    return h;
}

template <int NM, int NOPS, int WHICH>
void test(int th, const std::vector<float>& ref) {
    auto x = run<NM, NOPS, WHICH>(th);
    size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i) if (std::fabs((double)x[i] - ref[i]) > 1e-3 * std::fabs(ref[i])) ++bad;
    printf("waves/SIMD %d  %d MFMA(s), %s overwritten after %2d wait states: %zu of %zu lanes differ\n", th / 256, NM, WHICH ? "A" : "B", NOPS,
           bad, x.size());
}

template <int NM>
void sweep(int th) {
    auto ref = run<NM, -1, 0>(th);
    test<NM, 0, 0>(th, ref); test<NM, 1, 0>(th, ref); test<NM, 2, 0>(th, ref); test<NM, 4, 0>(th, ref); test<NM, 8, 0>(th, ref);
    test<NM, 16, 0>(th, ref); test<NM, 32, 0>(th, ref); test<NM, 64, 0>(th, ref);
    test<NM, 0, 1>(th, ref); test<NM, 4, 1>(th, ref); test<NM, 16, 1>(th, ref);
}

int main() {
    for (int th : {256, 512}) { sweep<1>(th); sweep<8>(th); }
    return 0;
}
