// wino32.hip - Winograd F(2x2,3x3) for the 32-output-channel, full-resolution layers (encoder_block1, decoder_block6's
// ConvBlockRes; models/resunet.py:147-165 at the shapes of :315-323,408-418): weights resident, waves decoupled.
//
// Same products as wino.hip (same transforms up to the order of the two passes of B^T d B, same k order of the f32 MFMA
// chain), another structure.  With 32 output channels the transform-domain weights of a whole layer are 2 KiB per input
// channel (64 KiB at Cin = 32, 128 KiB at Cin = 64): they are loaded into LDS ONCE per workgroup, and a persistent workgroup
// (one per CU, 8 waves) then walks over the image.  And because every wave of such a layer owns its tiles alone (32 couts x
// 16 tiles x all 16 xi, wino.hip's register-local output transform), the lane that TRANSFORMS a 4x4 patch is the lane that
// FEEDS it to the MFMA: lane (kq, l15) of v_mfma_f32_16x16x4_f32 supplies B[k = kq][col = l15], so it loads the patch of
// channel 4*ks + kq at tile l15 straight from global memory (8 unaligned 8-byte loads, two k-steps ahead), applies the
// BN+FiLM+leaky prologue and B^T d B in registers, and its 16 results ARE the B operands of the 16 xi of that k-step.
// No raw tile, no V image, no weight DMA per chunk, no workgroup barrier after start-up: the only LDS traffic of the K loop
// is one ds_read_b128 per 4 MFMAs (the A fragments), and the waves of a CU run free of each other.
//
// What the schedule is built on (tools/mfma_valu_ubench.hip, measured on MI355X): the f32 MFMA does NOT run beside vector
// instructions - one v_mfma_f32_16x16x4_f32 alone costs 32 cycles per wave, with 4 v_fma behind it 59, with 8 78, and with
// two waves on the SIMD the cost per vector instruction only falls from 4.5 to 3 cycles.  Every vector, LDS or memory
// instruction is therefore paid in full on top of the MFMA time, interleaved or not; so the kernel issues as few of them as
// it can (per k-step of 32 MFMAs: 32 adds for B^T d B, 48 for the prologue where there is one, 8 loads, 8 LDS reads), in ONE
// block between two runs of 32 back-to-back MFMAs, and nothing else (pins between single MFMAs cost a v_mov and an s_nop
// each when they were tried).  Also tried and not kept (profiles/r03/ubench/): packed-f32 arithmetic for
// the transforms, 16-byte patch loads, the next strip's first patches requested in front of the epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "kernels.h"
#include "pixel_ops.h"
#include "wino_common.h"
#include "wino_epilogue.h"

namespace {

constexpr int F_PRO = 1, F_PHASEB = 2, F_BIAS = 4, F_RES = 8, F_EPIACT = 16;
constexpr int F_MASK = 1024, F_PRECONV = 64, F_RESPRE = 128;  // as in wino.hip
constexpr int NW32 = 8;           // waves per workgroup: two per SIMD, one workgroup per CU
#ifndef W32_EXP
#define W32_EXP 0
#endif
// Every MFMA statement opens with two wait states.  Cause (round 4, established in the ISA): on gfx950 an MFMA must sit at
// least two wait states behind a vector instruction that wrote one of its source operands; hipcc pads that for its builtin
// MFMAs (tools/mfma_operand_hazard.hip: v_mul, s_waitcnt, s_nop 0, v_mfma) and pads nothing for an asm statement.  Without the
// s_nop the K loop's FIRST statement follows the input transform directly - `v_sub_f32 v0, v0, v12` then
// `v_mfma_f32_16x16x4_f32 v[108:111], v4, v0, 0` in decoder_block6's kernel - and reads a stale B operand into the xi = 0
// accumulators: round 3's run-to-run wrong results.  tools/audit_wino32_isa.py checks the two wait states for every MFMA of
// the shipped ISA and fails on the s_nop-less build (tests/test_host_cpu.py::test_wino32_isa_audit).
#ifndef W32_NOP
#define W32_NOP "s_nop 1\n\t"
#endif
constexpr int NTH32 = 64 * NW32;

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(3))) const float lds_cfloat;  // LDS pointers that stay LDS pointers (ds_read, not flat)
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;

// Eight MFMAs (xi = x0 .. x0+3, both cout tiles) as ONE asm volatile statement.  hipcc treats the MFMA builtins as pure
// arithmetic and, in a basic block as large as a whole strip, re-orders them at will across
// __builtin_amdgcn_sched_barrier (whole phases swapped places, the accumulators spilled in between); asm volatile statements
// keep their program order.  The compiler does not see inside: nothing it generates may write an operand right in front of
// a statement or touch an accumulator inside the K loop (tools/audit_wino32_isa.py checks both in the ISA); the accumulate
// chain itself needs no wait states (an accumulator is touched again 32 MFMAs later), and the first reader behind the K
// loop waits explicitly (mfma_drain).  a0 = {(x0,t0), (x0,t1), (x0+1,t0), (x0+1,t1)}, a1 the same for x0+2, x0+3.
template <bool FIRST>
__device__ __forceinline__ void mfma8(f32x4 (&acc)[16][2], int x0, const f32x4& a0, const f32x4& a1, const float* b) {
    if (FIRST)
        asm volatile(
            W32_NOP "v_mfma_f32_16x16x4_f32 %0, %8, %16, 0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %16, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %10, %17, 0\n\tv_mfma_f32_16x16x4_f32 %3, %11, %17, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %12, %18, 0\n\tv_mfma_f32_16x16x4_f32 %5, %13, %18, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %14, %19, 0\n\tv_mfma_f32_16x16x4_f32 %7, %15, %19, 0"
            : "=&v"(acc[x0][0]), "=&v"(acc[x0][1]), "=&v"(acc[x0 + 1][0]), "=&v"(acc[x0 + 1][1]), "=&v"(acc[x0 + 2][0]),
              "=&v"(acc[x0 + 2][1]), "=&v"(acc[x0 + 3][0]), "=&v"(acc[x0 + 3][1])
            : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(b[0]), "v"(b[1]),
              "v"(b[2]), "v"(b[3]));
    else
        asm volatile(
            W32_NOP "v_mfma_f32_16x16x4_f32 %0, %8, %16, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %16, %1\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %10, %17, %2\n\tv_mfma_f32_16x16x4_f32 %3, %11, %17, %3\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %12, %18, %4\n\tv_mfma_f32_16x16x4_f32 %5, %13, %18, %5\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %14, %19, %6\n\tv_mfma_f32_16x16x4_f32 %7, %15, %19, %7"
            : "+v"(acc[x0][0]), "+v"(acc[x0][1]), "+v"(acc[x0 + 1][0]), "+v"(acc[x0 + 1][1]), "+v"(acc[x0 + 2][0]),
              "+v"(acc[x0 + 2][1]), "+v"(acc[x0 + 3][0]), "+v"(acc[x0 + 3][1])
            : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(b[0]), "v"(b[1]),
              "v"(b[2]), "v"(b[3]));
}
// Four MFMAs of one xi pair (the shortcut phase): a = {(xi0,t0), (xi0,t1), (xi1,t0), (xi1,t1)}.
__device__ __forceinline__ void mfma4(f32x4 (&acc)[16][2], int xi0, int xi1, const f32x4& a, float b0, float b1) {
    asm volatile(
        W32_NOP "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\tv_mfma_f32_16x16x4_f32 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %6, %9, %2\n\tv_mfma_f32_16x16x4_f32 %3, %7, %9, %3"
        : "+v"(acc[xi0][0]), "+v"(acc[xi0][1]), "+v"(acc[xi1][0]), "+v"(acc[xi1][1])
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b0), "v"(b1));
}
// Behind the last MFMA statement, in front of the first instruction that reads an accumulator (8-pass MFMA: 18 wait states
// would do).  The statements name every accumulator, so no reader (the output transform is pure arithmetic, which hipcc
// hoists freely) can be scheduled above them.
__device__ __forceinline__ void mfma_drain(f32x4 (&acc)[16][2]) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int xi = 0; xi < 16; xi += 4)
        asm volatile("" : "+v"(acc[xi][0]), "+v"(acc[xi][1]), "+v"(acc[xi + 1][0]), "+v"(acc[xi + 1][1]), "+v"(acc[xi + 2][0]),
                          "+v"(acc[xi + 2][1]), "+v"(acc[xi + 3][0]), "+v"(acc[xi + 3][1]));
}

typedef unsigned v2u32 __attribute__((ext_vector_type(2)));

// Output transform Y = A^T M A and epilogue of one strip, written for instruction count (the vector instructions of the
// epilogue are paid on top of the MFMA time like those of the K loop; wino_epilogue.h spends ~900 per strip, two thirds of
// them on 64-bit addresses, selects for 16-byte stores and branches).  Every store is a buffer store: the lane part of the
// address (its kq * 4 channels, its 2x2 tile) is ONE register per tensor, the channel (t, r) is a scalar offset; a tile row
// is one 8-byte store (16 lanes = one 128-byte line).  Tables: (bias | epilogue scale, shift | pre_conv w, b) of channel n
// at tb[n] / tb2[n] as float2.  H is even, so both rows of a tile are inside the image together.
template <int FLAGS>
__device__ __forceinline__ void w32_epilogue(const ConvArgs& p, f32x4 (&acc)[16][2], int b, int n0, int oy, int ox, int kq,
                                             const float* lds_bias, const float2* lds_ep, const float2* lds_pre) {
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0, BIAS = (FLAGS & F_BIAS) != 0, RESPRE = (FLAGS & F_RESPRE) != 0;
    const int HW = p.H * p.W;
    // n0: first output channel of this workgroup's 32-cout slice (tables in LDS are the slice's own)
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.out_bs + (size_t)n0 * HW, 0, (int)(32u * (unsigned)HW * 4u), 0x00020000);
    const unsigned vo0 = (unsigned)(kq * 4 * HW + oy * p.W + ox) * 4u, vo1 = vo0 + (unsigned)p.W * 4u;
    const bool pool = p.pool_out != nullptr;  // wave-uniform
    const int Wo = p.W / 2, HWo = (p.H / 2) * Wo;
    const size_t pool_bs = p.pool_bs ? (size_t)p.pool_bs : (size_t)p.N * HWo;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pool ? p.pool_out + (size_t)b * pool_bs + (size_t)n0 * HWo : p.out, 0,
                                                                          pool ? (int)(32u * (unsigned)HWo * 4u) : 0, 0x00020000);
    const unsigned vp = (unsigned)(kq * 4 * HWo + (oy >> 1) * Wo + (ox >> 1)) * 4u;
    float2 x0r0 = make_float2(0.f, 0.f), x0r1 = x0r0;
    if (RESPRE) {  // the residual is pre_conv(x0) (resunet.py:555,165): one 2x2 patch of x0 serves all 32 channels
        const float* rp = p.res + (size_t)b * p.res_bs + (size_t)oy * p.W + ox;
        x0r0 = *reinterpret_cast<const float2*>(rp);
        x0r1 = *reinterpret_cast<const float2*>(rp + p.W);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = t * 16 + kq * 4 + r;
            float s0[4], s1[4];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                s0[jx] = acc[0 + jx][t][r] + acc[4 + jx][t][r] + acc[8 + jx][t][r];
                s1[jx] = acc[4 + jx][t][r] - acc[8 + jx][t][r] - acc[12 + jx][t][r];
            }
            float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
            float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
            if (BIAS) {
                const float bb = lds_bias[n];
                y00 += bb; y01 += bb; y10 += bb; y11 += bb;
            }
            if (RESPRE) {
                const float2 w = lds_pre[n];
                y00 += x0r0.x * w.x + w.y; y01 += x0r0.y * w.x + w.y;
                y10 += x0r1.x * w.x + w.y; y11 += x0r1.y * w.x + w.y;
            }
            if (EPI) {
                const float2 e = lds_ep[n];
                y00 = leaky(y00 * e.x + e.y); y01 = leaky(y01 * e.x + e.y);
                y10 = leaky(y10 * e.x + e.y); y11 = leaky(y11 * e.x + e.y);
            }
            const unsigned so = (unsigned)((t * 16 + r) * HW) * 4u;  // scalar
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, make_float2(y00, y01)), ors, (int)vo0, (int)so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, make_float2(y10, y11)), ors, (int)vo1, (int)so, 0);
            if (pool) {  // F.avg_pool2d (2, 2) of the block output (resunet.py:197), summed in the reference's row-major order
                float sum = y00 + y01;
                sum += y10;
                sum += y11;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum * 0.25f), prs, (int)vp,
                                                      (int)((unsigned)((t * 16 + r) * HWo) * 4u), 0);
            }
        }
}

template <int FLAGS, int CIN, int CIN2>
__global__ __launch_bounds__(NTH32, 2) void wino32_kernel(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr bool HASB = (FLAGS & F_PHASEB) != 0;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;
    constexpr bool BIAS = (FLAGS & F_BIAS) != 0;
    constexpr bool PRE = (FLAGS & F_PRECONV) != 0;
    constexpr bool RESPRE = (FLAGS & F_RESPRE) != 0;
    constexpr bool MASK = (FLAGS & F_MASK) != 0;
    constexpr int NKS = CIN / 4;                   // k-steps of the 3x3 phase
    constexpr int U_F = CIN * 512;                 // floats: [ks][xi pair 8][kq 4][l15 16][4]
    constexpr int U2_F = HASB ? CIN2 * 128 : 0;    // [ks][pair 2][kq][l15][4]
    constexpr int TAB_F = PRO ? CIN * 2 : 0;           // per wave: prologue (scale, shift) of every input channel, current clip
    constexpr int EPI_F = EPI ? 64 : 0;                // per wave: epilogue scale[32], shift[32] of the current clip
    constexpr int SH_F = 32 + 64 + 100;                // shared: bias[32], pre_w[32] + pre_b[32], mask_w[96] + mask_b[3]
    static_assert(!PRE || (CIN == 32 && PRO), "pre_conv has 32 output channels and is folded into the prologue");
    static_assert(CIN % 8 == 0 && CIN2 % 8 == 0, "k-steps come in pairs");

    __shared__ __attribute__((aligned(16))) float lds[U_F + U2_F + NW32 * (TAB_F + EPI_F) + SH_F];
    float* lu = lds;
    float* lu2 = lds + U_F;
    float* lds_bias = lds + U_F + U2_F + NW32 * (TAB_F + EPI_F);
    float* lds_pw = lds_bias + 32;
    float* lds_pb = lds_pw + 32;
    float* lds_mw = lds_pb + 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, l15 = lane & 15;
    float* wtab = lds + U_F + U2_F + wave * (TAB_F + EPI_F);  // this wave's private tables
    float* wes = wtab + TAB_F;  // epilogue (scale, shift) pairs of the 32 output channels, current clip
    float* weh = wes + 32;      // (shared epilogue of the MASK / plain-RES variants: separate arrays)
    const int HW = p.H * p.W;

    // ---- which blocks, which 32-cout slice -----------------------------------------------------------------------------
    // A block = 8 vertically adjacent strips of 2 rows x 32 columns.  A layer with N = 64 output channels runs as two
    // slices of 32: a workgroup serves ONE slice (its weights stay resident) and walks the blocks; the workgroups of the
    // two slices walk the same blocks side by side (the patch is transformed once per slice: that is this kernel's price).
    // Block order: workgroup g runs on XCD g % 8 (round-robin dispatch), and every XCD has its own L2.  Each XCD walks ONE
    // contiguous range of blocks, its workgroups side by side in it: the strips that share halo rows / the 128-byte lines
    // either side of a strip run at the same time under the same L2 (in plain round-robin order horizontal neighbours sit
    // on 8 different XCDs and every one of them fetches the shared lines from HBM: 5x the algorithmic read, measured).
    const int tiles_x = p.W / 32;
    const int rows_blk = (p.H / 2 + NW32 - 1) / NW32;
    const unsigned blocks_per_clip = (unsigned)(tiles_x * rows_blk);
    const unsigned nblk = blocks_per_clip * (unsigned)p.B;
    const unsigned ns = (unsigned)p.N >> 5;  // slices; the host makes gridDim.x a multiple of it
    unsigned blk, blk_end, blk_step, slice;
    if ((gridDim.x & 7u) == 0 && ((gridDim.x >> 3) % ns) == 0 && nblk * ns >= gridDim.x) {
        const unsigned j = blockIdx.x >> 3, chunk = (nblk + 7u) >> 3;
        slice = j % ns;
        blk = (blockIdx.x & 7u) * chunk + j / ns;
        blk_end = min(nblk, ((blockIdx.x & 7u) + 1u) * chunk);
        blk_step = (gridDim.x >> 3) / ns;
    } else {
        slice = blockIdx.x % ns;
        blk = blockIdx.x / ns;
        blk_end = nblk;
        blk_step = gridDim.x / ns;
    }
    const int n0 = (int)slice * 32;

    // ---- start-up: the slice's transform-domain weights -> LDS, once (linear LDS-DMA copy of the image lass_finalize
    // wrote, see wino32_weights_kernel), and the clip-independent tables --------------------------------------------------
    {
        const unsigned lu_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lu;
        const v4i32 urs = make_rsrc_words(p.w_wino32 + (size_t)slice * U_F, (unsigned)U_F * 4u);
        for (int piece = wave; piece < U_F / 256; piece += NW32)
            lds_dma_16B(urs, (unsigned)lane * 16u, (unsigned)piece * 1024u, lu_addr + (unsigned)piece * 1024u);
        if (HASB) {
            const v4i32 u2rs = make_rsrc_words(p.w2_wino32 + (size_t)slice * U2_F, (unsigned)U2_F * 4u);
            for (int piece = wave; piece < U2_F / 256; piece += NW32)
                lds_dma_16B(u2rs, (unsigned)lane * 16u, (unsigned)piece * 1024u, lu_addr + (unsigned)(U_F * 4 + piece * 1024));
        }
    }
    if (BIAS && tid < 32) lds_bias[tid] = p.bias[n0 + tid];
    constexpr bool OWN_EPI = !MASK && !((FLAGS & F_RES) != 0 && !RESPRE);  // w32_epilogue; else wino_epilogue.h
    if ((PRE || RESPRE) && tid < 32) {
        if (OWN_EPI) {
            *reinterpret_cast<float2*>(lds_pw + tid * 2) = make_float2(p.pre_w[tid], p.pre_b[tid]);
        } else {
            lds_pw[tid] = p.pre_w[tid];
            lds_pb[tid] = p.pre_b[tid];
        }
    }
    if (MASK && tid < 99) lds_mw[tid] = tid < 96 ? p.mask_w[tid] : p.mask_b[tid - 96];
    wait_vmcnt<0>();
    __syncthreads();  // the only workgroup barrier of the kernel

    lds_cfloat* afrag = (lds_cfloat*)lu + kq * 64 + l15 * 4;    // + ks * 2048 + pair * 256
    lds_cfloat* afrag2 = (lds_cfloat*)lu2 + kq * 64 + l15 * 4;  // + ks * 512 + pair * 256

    int cur_b = -1;
#ifdef LASS_CONV_DIAG
    // timing experiments, compile-time (-DW32_EXP=n; results are wrong when set; a run-time switch would put a branch around
    // every MFMA and wreck the schedule being measured): 1 no patch loads, 2 no epilogue, 8 no MFMA, 16 no patch transform,
    // 32 one wave per SIMD, 64 patch loads of every k-step from channel group 0 (same instructions, cache hits)
    constexpr int EXPF = W32_EXP;
    long long dg[4] = {0, 0, 0, 0};
    const long long dg_k0 = clock64(), dg_r0 = wall_clock64();
    int dg_n = 0;
#else
    constexpr int EXPF = 0;
#endif

    for (; blk < blk_end; blk += blk_step) {
        const int b = (int)(blk / blocks_per_clip);
        const unsigned rr = blk - (unsigned)b * blocks_per_clip;
        const int by = (int)(rr / (unsigned)tiles_x), bx = (int)(rr - (unsigned)by * (unsigned)tiles_x);
        const int y0 = (by * NW32 + wave) * 2, x0 = bx * 32;
        if (y0 >= p.H) continue;  // wave-uniform; no barrier inside the loop
        if ((EXPF & 32) && wave >= 4) continue;  // timing experiment: one wave per SIMD

        if (b != cur_b) {  // per-clip tables into this wave's private LDS region (FiLM shifts differ from clip to clip)
            cur_b = b;
            if (PRO && lane < CIN) {
                float2 t = make_float2(p.pro_scale[lane], p.pro_shift[(size_t)b * p.pro_shift_bs + lane]);
                // pre_conv (1x1, 1 -> 32: w x0 + b, resunet.py:555) folded into the prologue's affine: one fma per element
                if (PRE) t = make_float2(p.pre_w[lane] * t.x, p.pre_b[lane] * t.x + t.y);
                *reinterpret_cast<float2*>(wtab + lane * 2) = t;
            }
            if (EPI && lane < 32)
                *reinterpret_cast<float2*>(wes + lane * 2) = make_float2(p.epi_scale[n0 + lane], p.epi_shift[(size_t)b * p.epi_shift_bs + n0 + lane]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: in-order LDS, no barrier
        }

        // ---- this lane's patch geometry ---------------------------------------------------------------------------
        const int oy = y0, ox = x0 + 2 * l15;
        const int gy0 = oy - 1, gx0 = ox - 1;
        const bool left = gx0 < 0, right = gx0 + 3 >= p.W;
        const bool border = __builtin_amdgcn_readfirstlane((y0 == 0 || y0 + 2 >= p.H || x0 == 0 || x0 + 32 >= p.W) ? 1 : 0) != 0;
        // a patch row = two 8-byte loads: lane part (channel kq, column pair) in a VGPR, row + k-step part in an SGPR
        const unsigned vl = 4u * (unsigned)((PRE ? 0 : kq * HW) + (left ? 0 : gx0));            // pair (gx0, gx0+1); left edge (0, 1)
        const unsigned vr = 4u * (unsigned)((PRE ? 0 : kq * HW) + (right ? p.W - 2 : gx0 + 2));  // pair (gx0+2, gx0+3); right edge (W-2, W-1)
        unsigned srow[4], rowok = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy = gy0 + i;  // wave-uniform
            srow[i] = 4u * (unsigned)(min(max(gy, 0), p.H - 1) * p.W);
            rowok |= (gy >= 0 && gy < p.H ? 1u : 0u) << i;
        }
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in + (size_t)b * p.in_bs), 0, (int)((unsigned)(PRE ? 1 : CIN) * (unsigned)HW * 4u), 0x00020000);

#ifdef LASS_CONV_DIAG
        const long long dg_t0 = clock64();
        long long dg_t1 = dg_t0, dg_t2 = dg_t0;
#endif
        // The strip body exists twice: interior strips (the vast majority) carry no padding selects at all.
        auto run_strip = [&](auto border_c) __attribute__((always_inline)) {
        constexpr bool BORDER = decltype(border_c)::value;
        f2u pa[8], pb[8];        // raw patches of an even / odd k-step (PRE: pa = the x0 patch, for every k-step)
        float bq0[16], bq1[16];  // B operands (V = B^T d B) of an even / odd k-step
        f32x4 aa[8], ab[8];      // A fragments (one per xi pair) of an even / odd k-step
        float2 tab;              // (scale, shift) of this lane's channel of the k-step being prepared

        auto pload = [&](int ks, f2u (&buf)[8]) __attribute__((always_inline)) {
            const unsigned soff = (PRE || (EXPF & 64)) ? 0u : (unsigned)(ks * 4 * HW) * 4u;  // 64: every k-step re-reads channel group 0 (cache hits)
            if (EXPF & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) buf[i] = f2u{0.25f, 0.5f};
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                buf[2 * i] = __builtin_bit_cast(f2u, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)vl, (int)(soff + srow[i]), 0));
                buf[2 * i + 1] = __builtin_bit_cast(f2u, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)vr, (int)(soff + srow[i]), 0));
            }
        };
        auto load_tab = [&](int ks) __attribute__((always_inline)) {
            if (PRO) tab = *reinterpret_cast<const float2*>(wtab + (ks * 4 + kq) * 2);
        };
        // The A fragments of k-step ks.  The weights in LDS never change after start-up, so to the compiler these reads are
        // invariant in the persistent loop: left alone it hoists all of them out of it (256 registers, spilled to scratch).
        // An opaque address per k-step pins them where they are written.
        auto aload = [&](int ks, f32x4 (&a)[8]) __attribute__((always_inline)) {
            lds_cfloat* ap = afrag + ks * 2048;
            asm volatile("" : "+v"(ap));
#pragma unroll
            for (int g = 0; g < 8; ++g) a[g] = *reinterpret_cast<lds_cf32x4*>(ap + g * 256);
        };
        // Patch of one k-step -> its 16 B operands: prologue (+ zero padding) and V = B^T (d B), row pass first (the
        // activated patch itself is never kept: 4 live values per row).
        auto prep = [&](f2u (&buf)[8], float (&bq)[16]) __attribute__((always_inline)) {
            if (EXPF & 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) bq[i] = 0.5f + i;
                return;
            }
            float u[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2u a = buf[2 * i], c = buf[2 * i + 1];
                float v0 = a.x, v1 = a.y, v2 = c.x, v3 = c.y;
                if (BORDER) {
                    v1 = left ? a.x : a.y;
                    v2 = right ? c.y : c.x;
                }
                if (PRO) {
                    v0 = leaky(v0 * tab.x + tab.y); v1 = leaky(v1 * tab.x + tab.y);
                    v2 = leaky(v2 * tab.x + tab.y); v3 = leaky(v3 * tab.x + tab.y);
                }
                if (BORDER) {  // zero padding comes AFTER the activation
                    const bool rok = ((rowok >> i) & 1u) != 0;
                    v0 = (rok && !left) ? v0 : 0.f;
                    v1 = rok ? v1 : 0.f;
                    v2 = rok ? v2 : 0.f;
                    v3 = (rok && !right) ? v3 : 0.f;
                }
                u[i][0] = v0 - v2; u[i][1] = v1 + v2; u[i][2] = v2 - v1; u[i][3] = v1 - v3;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bq[c] = u[0][c] - u[2][c];
                bq[4 + c] = u[1][c] + u[2][c];
                bq[8 + c] = u[2][c] - u[1][c];
                bq[12 + c] = u[1][c] - u[3][c];
            }
        };

        f32x4 acc[16][2];
        // One k-step: the 32 MFMAs of k-step ks back to back (B operands `cur`, A fragments `ac`), then ONE block of everything
        // else: the A fragments of k-step ks+1 (`an`), the patch of k-step ks+1 (`nbuf`) -> its B operands (`nxt`) when NEXT,
        // and the loads of the patch of k-step ks+3 into the registers that just became free when LOAD.  The switches are
        // compile-time: the K loop has no branch but its own back edge.
        auto kstep = [&](auto first_c, auto next_c, auto load_c, int ks, const float (&cur)[16], float (&nxt)[16], f32x4 (&ac)[8],
                         f32x4 (&an)[8], f2u (&nbuf)[8]) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value, NEXT = decltype(next_c)::value;
            constexpr bool LOAD = decltype(load_c)::value && !PRE;
            if (!(EXPF & 8) || FIRST) {
                mfma8<FIRST>(acc, 0, ac[0], ac[1], cur);
                mfma8<FIRST>(acc, 4, ac[2], ac[3], cur + 4);
                mfma8<FIRST>(acc, 8, ac[4], ac[5], cur + 8);
                mfma8<FIRST>(acc, 12, ac[6], ac[7], cur + 12);
            }
            if (NEXT) {
                aload(ks + 1, an);
                prep(PRE ? pa : nbuf, nxt);
                load_tab(ks + 2 < NKS ? ks + 2 : NKS - 1);
            }
            if (LOAD) pload(ks + 3, nbuf);
        };
        constexpr std::true_type T{};
        constexpr std::false_type F{};

        // ---- 3x3 phase: patch ks+1 is transformed behind the MFMAs of k-step ks, patch ks+3 is loaded there too ------------
        pload(0, pa);
        if (!PRE) pload(1, pb);
        aload(0, aa);
        load_tab(0);
        prep(pa, bq0);
        load_tab(1);
#ifdef LASS_CONV_DIAG
        dg_t1 = clock64();
#endif
        if (!PRE) pload(2, pa);
        kstep(T, T, T, 0, bq0, bq1, aa, ab, pb);
        kstep(F, T, T, 1, bq1, bq0, ab, aa, pa);
        for (int ks = 2; ks + 4 < NKS; ks += 2) {
            kstep(F, T, T, ks, bq0, bq1, aa, ab, pb);
            kstep(F, T, T, ks + 1, bq1, bq0, ab, aa, pa);
        }
        kstep(F, T, T, NKS - 4, bq0, bq1, aa, ab, pb);  // loads the last patch
        kstep(F, T, F, NKS - 3, bq1, bq0, ab, aa, pa);
        kstep(F, T, F, NKS - 2, bq0, bq1, aa, ab, pb);
        kstep(F, F, F, NKS - 1, bq1, bq0, ab, aa, pa);

        // ---- shortcut phase: 1x1 over p.in2 in the transform domain (xi in {5, 6, 9, 10}) ------------------------------
        if constexpr (HASB) {
            constexpr int NKB = CIN2 / 4;
            constexpr int PF = 4;  // k-steps of load prefetch (a k-step is only 8 MFMAs)
            const __amdgpu_buffer_rsrc_t in2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in2 + (size_t)b * p.in2_bs), 0, (int)((unsigned)CIN2 * (unsigned)HW * 4u), 0x00020000);
            const unsigned vo1 = (unsigned)(kq * HW + oy * p.W + ox) * 4u, vo2 = vo1 + (unsigned)p.W * 4u;
            asm volatile("" ::: "memory");  // keep the loads below out of the 3x3 phase (hoisted there they are spilled)
            float2 rb[PF][2];
            auto loadB = [&](int ks, float2 (&r)[2]) __attribute__((always_inline)) {
                const unsigned soff = (unsigned)(ks * 4 * HW) * 4u;
                r[0] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(in2_rsrc, (int)vo1, (int)soff, 0));
                r[1] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(in2_rsrc, (int)vo2, (int)soff, 0));
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) loadB(i, rb[i]);
            for (int ks0 = 0; ks0 < NKB; ks0 += PF) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int ks = ks0 + i;
                    const float2 r1 = rb[i][0], r2 = rb[i][1];  // patch rows 1, 2 x cols 1, 2
                    if (ks + PF < NKB) loadB(ks + PF, rb[i]);
                    const float t1a = r1.x + r2.x, t1b = r1.y + r2.y;
                    const float t2a = r2.x - r1.x, t2b = r2.y - r1.y;
                    const float v5 = t1a + t1b, v6 = t1b - t1a, v9 = t2a + t2b, v10 = t2b - t2a;
                    lds_cfloat* ap2 = afrag2 + ks * 512;
                    asm volatile("" : "+v"(ap2));  // as in kstep: keep the reads of the resident weights in place
                    const f32x4 a0 = *reinterpret_cast<lds_cf32x4*>(ap2);
                    const f32x4 a1 = *reinterpret_cast<lds_cf32x4*>(ap2 + 256);
                    mfma4(acc, 5, 6, a0, v5, v6);
                    mfma4(acc, 9, 10, a1, v9, v10);
                }
            }
        }

        mfma_drain(acc);
#ifdef LASS_CONV_DIAG
        dg_t2 = clock64();
#endif
        if (!(EXPF & 2)) {
            if constexpr (OWN_EPI)
                w32_epilogue<FLAGS>(p, acc, b, n0, oy, ox, kq, lds_bias, reinterpret_cast<const float2*>(wes),
                                    reinterpret_cast<const float2*>(lds_pw));
            else
                wino_epilogue<FLAGS>(p, acc, b, 0, 0, oy, ox, lane, lds_bias, wes, weh, lds_pw, lds_pb, lds_mw);
        }
        };  // run_strip
        if (border)
            run_strip(std::true_type{});
        else
            run_strip(std::false_type{});
#ifdef LASS_CONV_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long dg_t3 = clock64();
        dg[0] += dg_t1 - dg_t0; dg[1] += dg_t2 - dg_t1; dg[2] += dg_t3 - dg_t2; ++dg_n;
#endif
    }
#ifdef LASS_CONV_DIAG
    if (p.dbg && lane == 0) {
        long long* d = p.dbg + 8 * ((size_t)blockIdx.x * NW32 + wave);
        d[0] = dg[0]; d[1] = dg[1]; d[2] = dg[2]; d[3] = dg_n;
        d[4] = clock64() - dg_k0; d[5] = wall_clock64() - dg_r0;
    }
#endif
}

// Transform-domain weights of a 32-cout layer as the LDS image wino32_kernel keeps resident: element
// [ks = cin / 4][xi pair = xi / 2][kq = cin % 4][l15 = cout % 16][(xi % 2) * 2 + cout / 16]; U as in wino_weights_kernel.
__global__ __launch_bounds__(256) void wino32_weights_kernel(const float* __restrict__ w, int Cin, float* __restrict__ U) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // (cin, cout), cout fastest
    if (i >= 32 * Cin) return;
    const int co = i & 31, ci = i >> 5;
    const float* g = w + ((size_t)co * Cin + ci) * 9;
    double gg[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) gg[a][c] = g[a * 3 + c];
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double t[4][3];
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 3; ++c) t[a][c] = G[a][0] * gg[0][c] + G[a][1] * gg[1][c] + G[a][2] * gg[2][c];
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 4; ++c) {
            const double u = t[a][0] * G[c][0] + t[a][1] * G[c][1] + t[a][2] * G[c][2];
            const int xi = a * 4 + c;
            U[((((ci >> 2) * 8 + (xi >> 1)) * 4 + (ci & 3)) * 16 + (co & 15)) * 4 + (xi & 1) * 2 + (co >> 4)] = (float)u;
        }
}

// Shortcut (1x1) weights in the transform domain, q = 0..3 <-> xi = 5, 6, 9, 10 (wino_shortcut_weights_kernel): element
// [ks][pair = q / 2][kq][l15][(q % 2) * 2 + cout / 16].
__global__ __launch_bounds__(256) void wino32_shortcut_weights_kernel(const float* __restrict__ w, int Cin, float* __restrict__ U) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 32 * Cin) return;
    const int co = i & 31, ci = i >> 5;
    const float v = w[(size_t)co * Cin + ci] * 0.25f;
    const float sgn[4] = {1.f, -1.f, -1.f, 1.f};
    for (int q = 0; q < 4; ++q)
        U[((((ci >> 2) * 2 + (q >> 1)) * 4 + (ci & 3)) * 16 + (co & 15)) * 4 + (q & 1) * 2 + (co >> 4)] = sgn[q] * v;
}

template <int FLAGS, int CIN, int CIN2>
hipError_t launch32(const ConvArgs& p, hipStream_t stream) {
    // CUs of the CURRENT device (lass_separate runs under hipSetDevice(ctx->device)); cached per device, not per process
    static int ncu_of[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cached = ncu_of[dev & 63];
    if (cached <= 0) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cached = n > 0 ? n : 256;
    }
    const int ncu = cached;
    const long ns = p.N / 32;  // 32-cout slices: a workgroup serves one of them
    const long nblk = (long)(p.W / 32) * ((p.H / 2 + NW32 - 1) / NW32) * p.B * ns;
    // persistent: one workgroup per CU, every wave's loop is bounded; a multiple of the slice count
    const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu / ns * ns);
#ifdef LASS_CONV_DIAG
    ConvArgs q = p;
    static long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc((void**)&dbuf, (size_t)1024 * NW32 * 64);
    (void)hipMemset(dbuf, 0, (size_t)1024 * NW32 * 64);
    q.dbg = dbuf;
    hipLaunchKernelGGL((wino32_kernel<FLAGS, CIN, CIN2>), dim3(grid), dim3(NTH32), 0, stream, q);
    {
        std::vector<long long> h((size_t)grid * NW32 * 8);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < (size_t)grid * NW32; ++i)
            for (int k = 0; k < 6; ++k) s[k] += (double)h[i * 8 + k];
        const double n = s[3] > 0 ? s[3] : 1;
        fprintf(stderr, "[wino32-diag] exp=%d flags=%d Cin=%d Cin2=%d %dx%d B=%d grid=%u | cycles per strip and wave: first patch %.0f  K loop %.0f  "
                "epilogue %.0f  (strips per wave %.1f) | wave total %.0f cycles, clock %.3f GHz\n", W32_EXP, FLAGS, CIN, CIN2, p.H, p.W, p.B, grid,
                s[0] / n, s[1] / n, s[2] / n, n / ((double)grid * NW32), s[4] / ((double)grid * NW32), s[4] / s[5] * 0.1);
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((wino32_kernel<FLAGS, CIN, CIN2>), dim3(grid), dim3(NTH32), 0, stream, p);
    return hipGetLastError();
#endif
}

}  // namespace

bool lass_wino32_supported(ConvKind kind, const ConvArgs& p) {
    static const int kind_mask = [] { const char* e = getenv("LASS_W32_KINDS"); return e ? atoi(e) : 0xff; }();  // debugging aid
    if (!((kind_mask >> (int)kind) & 1)) return false;
    if (p.pool_out && p.pool_h != 2) return false;
    if ((p.N != 32 && p.N != 64 && p.N != 128) || p.Nw != p.N || p.W < 32 || (p.W % 32) != 0 || (p.H % 2) != 0 || !p.w_wino32) return false;
    // 32-bit byte offsets inside one clip's tensors
    if ((unsigned long long)p.Cin * p.H * p.W * 4ull > 0xFFFF0000ull) return false;
    if (p.N > 32) {  // 32-cout slices (encoder_block2, encoder_block3.conv1): the variants whose resident weights fit LDS
        static const int nmax = [] { const char* e = getenv("LASS_W32_NMAX"); return e ? atoi(e) : 128; }();  // A/B: 32 = no slices
        if (p.N > nmax) return false;
        if (kind == CONV1_ACT) return p.Cin == 32 || p.Cin == 64;
        return kind == CONV2_SHORTCUT && p.N == 64 && p.Cin == 64 && p.Cin2 == 32 && p.w2_wino32 && !p.mask_re;
    }
    switch (kind) {
        case CONV1_ACT: return p.Cin == 32 || p.Cin == 64;
        case CONV1_ACT_PRE: return p.Cin == 32;
        case CONV2_IDENT: case CONV2_IDENT_PRE: return p.Cin == 32;
        case CONV2_SHORTCUT:
            return p.Cin == 32 && (p.Cin2 == 64 || p.Cin2 == 128) && p.w2_wino32 &&
                   (unsigned long long)p.Cin2 * p.H * p.W * 4ull <= 0xFFFF0000ull;
        default: return false;
    }
}

hipError_t lass_launch_wino32(ConvKind kind, const ConvArgs& p, hipStream_t stream) {
    if (!lass_wino32_supported(kind, p) || !p.in || (!p.out && !p.mask_re)) return hipErrorInvalidValue;
    switch (kind) {
        case CONV1_ACT:
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift) return hipErrorInvalidValue;
            return p.Cin == 64 ? launch32<F_PRO | F_EPIACT, 64, 0>(p, stream) : launch32<F_PRO | F_EPIACT, 32, 0>(p, stream);
        case CONV1_ACT_PRE:
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift || !p.pre_w || !p.pre_b) return hipErrorInvalidValue;
            return launch32<F_PRO | F_EPIACT | F_PRECONV, 32, 0>(p, stream);
        case CONV2_IDENT:
            if (!p.res) return hipErrorInvalidValue;
            return launch32<F_RES, 32, 0>(p, stream);
        case CONV2_IDENT_PRE:
            if (!p.res || !p.pre_w || !p.pre_b) return hipErrorInvalidValue;
            return launch32<F_RES | F_RESPRE, 32, 0>(p, stream);
        case CONV2_SHORTCUT:
            if (!p.in2 || !p.bias) return hipErrorInvalidValue;
            if (p.mask_re) {
                if (p.W + 1 != p.mask_nbins || !p.mask_w || !p.mask_b || !p.mask_mag || !p.mask_cos || !p.mask_sin || !p.mask_im ||
                    p.mask_T <= 0 || p.mask_T > p.H)
                    return hipErrorInvalidValue;
                return p.Cin2 == 64 ? launch32<F_PHASEB | F_BIAS | F_MASK, 32, 64>(p, stream)
                                    : launch32<F_PHASEB | F_BIAS | F_MASK, 32, 128>(p, stream);
            }
            if (p.Cin == 64) return launch32<F_PHASEB | F_BIAS, 64, 32>(p, stream);
            return p.Cin2 == 64 ? launch32<F_PHASEB | F_BIAS, 32, 64>(p, stream) : launch32<F_PHASEB | F_BIAS, 32, 128>(p, stream);
        default:
            return hipErrorInvalidValue;
    }
}

// w: (Cout, Cin, 3, 3) with Cout a multiple of 32 -> U: one resident image (512 * Cin floats) per 32-cout slice
hipError_t lass_launch_wino32_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream) {
    for (int s = 0; s < Cout / 32; ++s)
        hipLaunchKernelGGL(wino32_weights_kernel, dim3((unsigned)((32 * Cin + 255) / 256)), dim3(256), 0, stream,
                           w + (size_t)s * 32 * Cin * 9, Cin, U + (size_t)s * 512 * Cin);
    return hipGetLastError();
}

// w: (Cout, Cin) -> U: 128 * Cin floats per 32-cout slice
hipError_t lass_launch_wino32_shortcut_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream) {
    for (int s = 0; s < Cout / 32; ++s)
        hipLaunchKernelGGL(wino32_shortcut_weights_kernel, dim3((unsigned)((32 * Cin + 255) / 256)), dim3(256), 0, stream,
                           w + (size_t)s * 32 * Cin, Cin, U + (size_t)s * 128 * Cin);
    return hipGetLastError();
}
