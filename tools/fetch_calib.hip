// fetch_calib.hip - calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the conv kernels
// use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Every kernel below moves a KNOWN number of bytes of a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   rd_dword_linear     buffer_load_dword, 4 B/lane, a wave reads 256 contiguous bytes
//   rd_dword_rows128    buffer_load_dword, 4 B/lane, 128-byte runs at a 2-KiB pitch (the halo-tile staging of wino.hip /
//                       conv.hip: 32-pixel row segments of a 512-wide plane)
//   rd_dwordx2_rows     buffer_load_dwordx2, 8 B/lane, 128-byte runs at a 2-KiB pitch (patch staging / shortcut centres)
//   rd_dwordx4_linear   global_load_dwordx4, 16 B/lane (the guide's calibrated case: expect 0.5)
//   rd_lds_dma          buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction (weight slabs, bf16 activations)
//   wr_dword_rows128    4 B/lane stores, 128-byte runs at a 2-KiB pitch (conv.hip epilogue)
//   wr_dwordx2_rows128  8 B/lane stores, 128-byte runs (wino.hip epilogue: float2 per lane, 16 lanes per row segment)
//   wr_dwordx4_linear   16 B/lane stores
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o x --output-format csv -- ./fetch_calib   (and WRITE_SIZE),
// then tools/fetch_calib_summary.py turns the two CSVs into profiles/rNN/fetch_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t N = (size_t)1 << 28;  // floats = 1 GiB
constexpr int W = 512;                 // plane width of the row kernels

__global__ __launch_bounds__(256) void rd_dword_linear(const float* __restrict__ p, float* __restrict__ sink) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + (size_t)blockIdx.x * 16384, 0, 65536, 0x00020000);
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(threadIdx.x * 4), i * 1024, 0));
    if (s == 123.456f) sink[0] = s;
}

// block = 8 rows x 32 columns of a W-wide plane per pass; 64 passes walk down 512 rows
__global__ __launch_bounds__(256) void rd_dword_rows128(const float* __restrict__ p, float* __restrict__ sink) {
    const size_t plane = (size_t)(blockIdx.x / (W / 32)) * 512 * W;
    const int x0 = (blockIdx.x % (W / 32)) * 32;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + plane, 0, 512 * W * 4, 0x00020000);
    const int row = threadIdx.x / 32, col = threadIdx.x % 32;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((row * W + x0 + col) * 4), i * 8 * W * 4, 0));
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256) void rd_dwordx2_rows(const float* __restrict__ p, float* __restrict__ sink) {
    const size_t plane = (size_t)(blockIdx.x / (W / 32)) * 512 * W;
    const int x0 = (blockIdx.x % (W / 32)) * 32;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + plane, 0, 512 * W * 4, 0x00020000);
    const int row = threadIdx.x / 16, col = (threadIdx.x % 16) * 2;  // 16 rows x 32 columns per pass
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const float2 v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)((row * W + x0 + col) * 4), i * 16 * W * 4, 0));
        s += v.x + v.y;
    }
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256) void rd_dwordx4_linear(const float4* __restrict__ p, float* __restrict__ sink) {
    const float4* q = p + (size_t)blockIdx.x * 4096;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 16; ++i) { const float4 v = q[i * 256 + threadIdx.x]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256) void rd_lds_dma(const float* __restrict__ p, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 256 * 4];  // 4 waves x 1 KiB x 4 pieces
    const unsigned long long a = (unsigned long long)(p + (size_t)blockIdx.x * 16384);
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
    r.z = 65536;
    r.w = 0x00020000;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    for (int i = 0; i < 16; ++i) {
        const unsigned soff = (unsigned)((i * 4 + wave) * 1024), la = lbase + (unsigned)(((i & 3) * 4 + wave) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"((unsigned)(lane * 16)), "s"(r), "s"(soff), "s"(la) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void wr_dword_rows128(float* __restrict__ p) {
    float* q = p + (size_t)(blockIdx.x / (W / 32)) * 512 * W + (blockIdx.x % (W / 32)) * 32;
    const int row = threadIdx.x / 32, col = threadIdx.x % 32;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) q[(size_t)(i * 8 + row) * W + col] = (float)i;
}

__global__ __launch_bounds__(256) void wr_dwordx2_rows128(float* __restrict__ p) {
    float* q = p + (size_t)(blockIdx.x / (W / 32)) * 512 * W + (blockIdx.x % (W / 32)) * 32;
    const int row = threadIdx.x / 16, col = (threadIdx.x % 16) * 2;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) *reinterpret_cast<float2*>(q + (size_t)(i * 16 + row) * W + col) = make_float2((float)i, 1.f);
}

__global__ __launch_bounds__(256) void wr_dwordx4_linear(float4* __restrict__ p) {
    float4* q = p + (size_t)blockIdx.x * 4096;
#pragma unroll 8
    for (int i = 0; i < 16; ++i) q[i * 256 + threadIdx.x] = make_float4((float)i, 1.f, 2.f, 3.f);
}

int main() {
    float *buf = nullptr, *sink = nullptr;
    CK(hipMalloc((void**)&buf, N * sizeof(float)));
    CK(hipMalloc((void**)&sink, 256));
    CK(hipMemset(buf, 0, N * sizeof(float)));
    CK(hipDeviceSynchronize());
    const unsigned blocks = (unsigned)(N / 16384);  // every kernel: one block per 64 KiB
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rd_dword_linear, dim3(blocks), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL(rd_dword_rows128, dim3(blocks), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL(rd_dwordx2_rows, dim3(blocks), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL(rd_dwordx4_linear, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, sink);
        hipLaunchKernelGGL(rd_lds_dma, dim3(blocks), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL(wr_dword_rows128, dim3(blocks), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(wr_dwordx2_rows128, dim3(blocks), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(wr_dwordx4_linear, dim3(blocks), dim3(256), 0, 0, (float4*)buf);
        CK(hipDeviceSynchronize());
    }
    CK(hipGetLastError());
    printf("fetch_calib: every kernel moved %zu bytes\n", N * sizeof(float));
    return 0;
}
