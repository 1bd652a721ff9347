// wino4.hip - Winograd F(4x4,3x3) 3x3 convolutions on the gfx950 f32 MFMA (the >= 64-cout layers at W >= 32).
//
// Same reference semantics as wino.hip (models/resunet.py:101-119,147-165: 3x3 / stride 1 / pad 1 cross-correlation behind the
// BN+FiLM+leaky prologue, epilogue activation in front of conv2), one more step of the same algebra:
//      Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      with 6x6 transforms (Lavin & Gray 2016, F(4x4,3x3))
// i.e. 36 independent GEMMs  M_xi[cout][tile] += U_xi[cout][cin] * V_xi[cin][tile]  per 4x4 output tile: 36 multiplies for 16
// outputs = 2.25 per output and (cin, cout) pair where F(2x2,3x3) needs 4 and the direct form 9 - 1.78x fewer MFMAs than
// wino.hip on an MFMA-bound path.  The price is conditioning (transform constants 4, 5, 8 and 1/24): ~6x the rounding error
// of F(2x2,3x3) per layer (2-5e-6 relative at K = 128..768), against a parity bar of 1e-4 RMS on waveforms.
//
// Mapping: v_mfma_f32_16x16x4_f32; a wave owns 16 couts x 16 tiles and keeps ALL 36 xi accumulators (144 registers), so the
// output transform is register-local.  Workgroup = 2 x 2 waves = 32 couts x 32 tiles (512 output pixels), two per CU.
// Per chunk of 8 input channels:
//   1. the weight slab U[36][8][32] arrives by LDS-DMA (36 KiB, the exact LDS image lass_finalize wrote);
//   2. every thread owns ONE (tile, channel) item: its 6x6 input patch comes straight from global memory (an aligned
//      16-byte load + two 4-byte loads per row, requested one chunk ahead), gets the BN+FiLM+leaky prologue and the zero
//      padding and is transformed in registers (144 add / fma), then written to V[36][4][32][2] in LDS;
//   3. 72 MFMAs per wave (36 xi x 2 k-steps), A and B fragments one ds_read_b64 each.
// Kinds: conv1 of a ConvBlockRes (prologue + epilogue activation) and conv2 with the 1x1 shortcut (resunet.py:122-128,163),
// bias and the block's fused avg-pool (:197).  The shortcut is NOT taken through the transform domain: once the 36 xi are
// folded into this lane's 4 couts x 16 pixels, accumulator tile s = [16 couts][16 tiles] of sub-pixel s has exactly the MFMA
// D layout, so the 1x1 conv is 16 more MFMAs per 4 input channels with B operands straight from global memory (this lane's
// tile of channel 4 ks + kq: four 16-byte loads) and A = the shortcut weights - no transform, no LDS.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"
#include "pixel_ops.h"
#include "wino_common.h"

namespace {

constexpr int F_PRO = 1, F_PHASEB = 2, F_BIAS = 4, F_RES = 8, F_EPIACT = 16, F_PRECONV = 64, F_RESPRE = 128, F_MASK = 1024;  // as wino.hip
constexpr int NTHREADS = 256;
constexpr int KC = 8;
constexpr int NXI = 36;
constexpr int U_F = NXI * 256;            // floats of one (8-channel chunk, 32-cout group) weight slab
constexpr int V_F = NXI * 4 * 32 * 2;     // [xi][kq][tile][k-step]

typedef float f32x2v __attribute__((ext_vector_type(2)));

// 1-D input transform r = B^T d (6 -> 6), B^T of F(4,3): 12 instructions
__device__ __forceinline__ void bt6(const float (&d)[6], float (&r)[6]) {
    const float t0 = fmaf(-4.f, d[2], d[4]);   // d4 - 4 d2
    const float t1 = fmaf(-4.f, d[1], d[3]);   // d3 - 4 d1
    const float t2 = d[4] - d[2];
    const float u = d[3] - d[1];
    r[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    r[1] = t0 + t1;
    r[2] = t0 - t1;
    r[3] = fmaf(2.f, u, t2);
    r[4] = fmaf(-2.f, u, t2);
    r[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}

// 1-D output transform y = A^T m (6 -> 4)
__device__ __forceinline__ void at6(const float (&m)[6], float (&y)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = fmaf(2.f, d34, d12);
    y[2] = fmaf(4.f, s34, s12);
    y[3] = fmaf(8.f, d34, d12) + m[5];
}

// TC: tile columns of the block (32 tiles = (32 / TC) tile rows x TC tile columns; block = 4 * (32 / TC) rows x 4 * TC columns)
// NG: 32-cout groups per workgroup.  NG = 1: 4 waves, 32 couts x 32 tiles, two workgroups per CU (round 4).
// NG = 2 (round 5): 8 waves, 64 couts x 32 tiles, ONE workgroup per CU (U 72 KiB + V 36 KiB).  The transformed input V of a
// chunk is shared by 64 couts - every input tile is transformed once per 64 instead of once per 32 output channels.  Waves
// w and w + 4 sit on one SIMD; the halves {0-3} and {4-7} take turns: chunk c is transformed by half c & 1 (one wave per SIMD
// in the transform, 256 items as before) while the other half issues the chunk's weight DMA (72 pieces) - then all eight waves
// run the chunk's 72 MFMAs each.  Per SIMD and chunk: 144 MFMAs beside ONE input transform (NG = 1: beside two).
template <int TC, int FLAGS, int NG = 1>
__global__ __launch_bounds__(NTHREADS * NG, 2) void wino4_kernel(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0, EPI = (FLAGS & F_EPIACT) != 0, SC = (FLAGS & F_PHASEB) != 0;
    constexpr bool PRE = (FLAGS & F_PRECONV) != 0;   // the input is the 1-channel x0: channel c = pre_w[c] * x0 + pre_b[c] (resunet.py:555)
    constexpr bool RESPRE = (FLAGS & F_RESPRE) != 0; // identity residual = pre_conv(x0), never materialised (encoder_block1.conv2)
    constexpr bool MASK = (FLAGS & F_MASK) != 0;     // epilogue = after_conv + complex ratio mask; the block output is not written
    static_assert(!SC || (FLAGS & F_BIAS) != 0, "the shortcut conv has a bias");
    static_assert(!PRE || PRO, "pre_conv is folded into the prologue's affine");
    static_assert(!RESPRE || ((FLAGS & F_RES) != 0 && !SC && !EPI), "conv2 with the identity residual");
    static_assert(!MASK || SC, "the output head sits behind decoder_block6's conv2 + shortcut");
    constexpr int TR = 32 / TC;
    constexpr int OR_ = 4 * TR, OC = 4 * TC;
    static_assert(NG == 1 || NG == 2, "one or two 32-cout groups per workgroup");
    static_assert(NG == 1 || !MASK, "the fused output head is a 32-cout launch");
    static_assert(NG == 1 || (!PRE && !RESPRE), "encoder_block1's kinds are 32-cout launches");
    constexpr int NCO = 32 * NG;  // output channels of the workgroup
    __shared__ __attribute__((aligned(16))) float lds[NG * U_F + V_F + 2 * NCO + (MASK ? 100 : 0)];
    float* lu = lds;
    float* lv = lds + NG * U_F;
    float* lds_es = lv + V_F;
    float* lds_eh = lds_es + NCO;
    float* lds_mw = lds_eh + NCO;  // MASK: after_conv weight [3][32] + bias [3]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wwt = wave & 1;  // cout tile (16 couts, 0 .. 2 NG - 1) / tile group (16 tiles) of this wave
    const int half = NG == 2 ? (wave >> 2) : 0;  // NG = 2: which half of the workgroup (waves 0-3 / 4-7: SIMD partners w, w + 4)
    int bx_, by_, b;
    block_coords(p, bx_, by_, b);
    const int n0 = by_ * NCO;
    const int tiles_x = p.W / OC;
    const int y0 = (bx_ / tiles_x) * OR_, x0 = (bx_ % tiles_x) * OC;
    const int HW = p.H * p.W;
    const float* in_b = p.in + (size_t)b * p.in_bs;
    const float* sc = PRO ? p.pro_scale : nullptr;
    const float* sh = PRO ? p.pro_shift + (size_t)b * p.pro_shift_bs : nullptr;

    if (EPI && tid < NCO) {
        lds_es[tid] = p.epi_scale[n0 + tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + n0 + tid];
    }
    if (SC && tid < NCO) lds_es[tid] = p.bias[n0 + tid];  // (SC and EPI exclude each other: one table)
    if (RESPRE && tid < 32) {                               // residual affine of this block's 32 output channels
        lds_es[tid] = p.pre_w[n0 + tid];
        lds_eh[tid] = p.pre_b[n0 + tid];
    }
    if (MASK && tid < 99) lds_mw[tid] = tid < 96 ? p.mask_w[tid] : p.mask_b[tid - 96];

    f32x4 acc[NXI];
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[xi][r] = 0.f;

    const int kq = lane >> 4, l15 = lane & 15;
    // fragments: U image [xi][t][kq][l15][k]: this lane's {k-step 0, k-step 1} of cout tile wco; V image [xi][kq][tile ^ swz][k]
    const float* afrag = lu + (wco >> 1) * U_F + (wco & 1) * 128 + (kq * 16 + l15) * 2;  // slab of the 32-cout group, cout tile inside it
    const float* bfrag = lv + (kq * 32 + ((wwt * 16 + l15) ^ ((kq & 1) << 4))) * 2;
    const unsigned slab_pitch = (unsigned)(p.Nw / 32) * (unsigned)(U_F * 4);  // bytes between the slabs of consecutive chunks
    const unsigned slab_n0 = (unsigned)(n0 / 32) * (unsigned)(U_F * 4);
    const unsigned lu_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lu;
    const unsigned ulane = (unsigned)lane * 16u;
    const v4i32 uw = make_rsrc_words(p.w_wino4, (unsigned)(NXI * p.Cin * p.Nw) * 4u);

    // ---- this thread's item: tile pt (0..31) and channel c8 (0..7) of the chunk; its 6x6 patch, top-left (gy0, gx0) ------
    const int pt = tid & 31, c8 = (tid & 255) >> 5;  // (NG = 2: both halves map their 256 threads onto the 256 items of a chunk)
    const int pty = pt / TC, ptx = pt % TC;
    const int gy0 = y0 + 4 * pty - 1, gx0 = x0 + 4 * ptx - 1;
    const bool left = gx0 < 0, right = gx0 + 5 >= p.W;
    unsigned vo_c[6], vo_l[6], vo_r[6], rowok = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int gy = gy0 + i;
        const int row = (PRE ? 0 : c8 * HW) + min(max(gy, 0), p.H - 1) * p.W;
        vo_c[i] = 4u * (unsigned)(row + gx0 + 1);                 // columns gx0+1 .. gx0+4: 16-byte aligned, always inside
        vo_l[i] = 4u * (unsigned)(row + (left ? 0 : gx0));        // column gx0 (clamped at the left edge)
        vo_r[i] = 4u * (unsigned)(row + (right ? p.W - 1 : gx0 + 5));
        rowok |= (gy >= 0 && gy < p.H ? 1u : 0u) << i;
    }
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, (int)((unsigned)(PRE ? 1 : p.Cin) * (unsigned)HW * 4u), 0x00020000);
    const bool edge = left || right || rowok != 0x3fu;  // this item's patch reaches into the zero padding
    float4 pc[6];
    float pl[6], pr[6], ps = 1.f, ph = 0.f;
    const unsigned tvo = (unsigned)c8 * 4u;  // this item's entry of a per-channel table, within the chunk
    const auto tab_rsrc = [&](const float* t, int n) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t), 0, n * 4, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t sc_rsrc = tab_rsrc(PRO ? sc : p.in, p.Cin), sh_rsrc = tab_rsrc(PRO ? sh : p.in, p.Cin);
    const __amdgpu_buffer_rsrc_t pw_rsrc = tab_rsrc(PRE ? p.pre_w : p.in, 32), pb_rsrc = tab_rsrc(PRE ? p.pre_b : p.in, 32);
    auto pload = [&](int ch) {
        const unsigned soff = PRE ? 0u : (unsigned)(ch * KC * HW) * 4u;
        if (!PRE || ch == 0)  // PRE: every channel is an affine function of the one x0 patch, loaded once
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            pc[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)vo_c[i], (int)soff, 0));
            pl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)vo_l[i], (int)soff, 0));
            pr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)vo_r[i], (int)soff, 0));
        }
        // The table reads are buffer loads as well (one vector-memory instruction each, by construction): wait_vmcnt<NLOAD> below
        // counts them, and a plain C++ load could be merged, hoisted or scalarised by the compiler behind the count's back.
        const unsigned toff = (unsigned)(ch * KC) * 4u;
        if (PRO) {
            ps = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sc_rsrc, (int)tvo, (int)toff, 0));
            ph = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sh_rsrc, (int)tvo, (int)toff, 0));
        }
        if (PRE) {  // leaky(bn(pre_w x0 + pre_b) + beta) = leaky(x0 * (pre_w s) + (pre_b s + h))
            const float pw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pw_rsrc, (int)tvo, (int)toff, 0));
            const float pb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pb_rsrc, (int)tvo, (int)toff, 0));
            ph = fmaf(pb, ps, ph);
            ps = pw * ps;
        }
    };
    constexpr int NLOAD = (PRE ? 0 : 6 * 3) + (PRO ? 2 : 0) + (PRE ? 2 : 0);  // vector-memory operations of one pload (chunks >= 1)
    // V destination of this item: row (xi, kq = c8 % 4), column tile ^ swizzle, k-step c8 / 4; xi stride = 4 * 64 floats
    float* vdst = lv + ((c8 & 3) * 32 + (pt ^ ((c8 & 1) << 4))) * 2 + (c8 >> 2);
    auto pprocess = [&]() {
        float d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float v[6] = {pl[i], pc[i].x, pc[i].y, pc[i].z, pc[i].w, pr[i]};
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) d[i][jx] = PRO ? leaky(fmaf(v[jx], ps, ph)) : v[jx];
        }
        // zero padding comes AFTER the activation (resunet.py:150, conv padding) and touches only the outer ring of the patch
        // (row 0 / 5, column 0 / 5) of the items at the image border: wave-uniform branch, skipped by interior waves
        if (__builtin_amdgcn_ballot_w64(edge) != 0) {
            const bool r0 = (rowok & 1u) != 0, r5 = (rowok & 32u) != 0;
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) {
                d[0][jx] = r0 ? d[0][jx] : 0.f;
                d[5][jx] = r5 ? d[5][jx] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                d[i][0] = left ? 0.f : d[i][0];
                d[i][5] = right ? 0.f : d[i][5];
            }
        }
        float tt[6][6];  // B^T d: columns
#pragma unroll
        for (int jx = 0; jx < 6; ++jx) {
            const float col[6] = {d[0][jx], d[1][jx], d[2][jx], d[3][jx], d[4][jx], d[5][jx]};
            float r[6];
            bt6(col, r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][jx] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {  // (B^T d) B: rows
            float r[6];
            bt6(tt[i], r);
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) vdst[(i * 6 + jx) * 256] = r[jx];
        }
    };

    const int nch = p.Cin / KC;
#ifndef W4_EXP
#define W4_EXP 0  // timing experiments (wrong results): 1 weight DMA only for chunk 0, 2 no patch transform, 4 no MFMA, 8 no patch loads
#endif
    if (NG == 1 || half == 0) pload(0);
    else if (nch > 1) pload(1);  // NG = 2: half h transforms the chunks c = h mod 2 and fetches their patches
    lds_barrier();  // epilogue tables visible
    for (int ch = 0; ch < nch; ++ch) {
        lds_barrier();  // previous chunk's MFMAs have finished reading V / U
        __builtin_amdgcn_s_setprio(2);
        const bool duty = NG == 1 || half == (ch & 1);  // (wave-uniform) this wave transforms chunk ch
        if (NG == 1) {
        // The patch of this chunk was requested a whole MFMA phase ago.  Pin it as arrived HERE: hipcc counts only its own
        // loads, so a wait placed behind the LDS-DMA below would be vmcnt(0) and drain the weight slab before the transform.
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            asm volatile("" : "+v"(pc[i].x), "+v"(pc[i].y), "+v"(pc[i].z), "+v"(pc[i].w), "+v"(pl[i]), "+v"(pr[i]));
        }
        if (PRO) asm volatile("" : "+v"(ps), "+v"(ph));
        // weight slab of (chunk ch, cout group n0 / 32): 36 pieces of 1 KiB, 9 per wave
        if (!(W4_EXP & 1) || ch == 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const unsigned piece = (unsigned)(wave * 9 + i) * 1024u;
            lds_dma_16B(uw, ulane, (unsigned)ch * slab_pitch + slab_n0 + piece, lu_addr + piece);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(W4_EXP & 2) || ch == 0) pprocess();
        __builtin_amdgcn_sched_barrier(0);
        const bool pf = ch + 1 < nch;
        if (pf && !(W4_EXP & 8)) pload(ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (pf && !(W4_EXP & 8))
            wait_vmcnt<NLOAD>();  // this wave's pieces of U(ch) have landed; the patch of chunk ch+1 stays in flight
        else
            wait_vmcnt<0>();
        } else if (duty) {
            // transform half: the patch (requested two chunks ago) -> V; then request the patch of this half's next chunk, which
            // stays in flight through the MFMA phases (this half issues no LDS-DMA: hipcc's own load accounting is exact here)
            if (!(W4_EXP & 2) || ch == 0) pprocess();
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 2 < nch && !(W4_EXP & 8)) pload(ch + 2);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // DMA half: the two adjacent 32-cout slabs of (chunk ch, cout block n0 / 64) = 72 pieces of 1 KiB, 18 per wave.
            // (Its own patch loads for chunk ch + 1, requested a chunk ago, are older than these pieces: vmcnt(0) covers both.)
            if (!(W4_EXP & 1) || ch == 0)
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const unsigned piece = (unsigned)((wave & 3) * 18 + i) * 1024u;
                lds_dma_16B(uw, ulane, (unsigned)ch * slab_pitch + slab_n0 + piece, lu_addr + piece);
            }
            wait_vmcnt<0>();
        }
        lds_barrier();  // V visible, every wave's U pieces landed
        __builtin_amdgcn_s_setprio(0);
        // 36 GEMM steps x 2 k-steps; two xi in flight so that no MFMA depends on its predecessor (40-cycle dependent latency)
        constexpr int PFD = 2;  // pairs of fragment reads ahead
        f32x2v av[PFD + 1][2], bv[PFD + 1][2];
        auto rd = [&](int s) {  // step s = xi pair (2s, 2s+1)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                av[s % (PFD + 1)][q] = *reinterpret_cast<const f32x2v*>(afrag + (2 * s + q) * 256);
                bv[s % (PFD + 1)][q] = *reinterpret_cast<const f32x2v*>(bfrag + (2 * s + q) * 256);
            }
        };
#pragma unroll
        for (int s = 0; s < PFD; ++s) rd(s);
#pragma unroll
        for (int s = 0; s < ((W4_EXP & 4) ? 1 : NXI / 2); ++s) {
            if (s + PFD < NXI / 2) rd(s + PFD);
            __builtin_amdgcn_sched_barrier(0);
            const f32x2v a0 = av[s % (PFD + 1)][0], a1 = av[s % (PFD + 1)][1];
            const f32x2v b0 = bv[s % (PFD + 1)][0], b1 = bv[s % (PFD + 1)][1];
            acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[2 * s], 0, 0, 0);
            acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[2 * s + 1], 0, 0, 0);
            acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[2 * s], 0, 0, 0);
            acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[2 * s + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- output transform Y = A^T M A: this lane's 4 couts (D rows kq * 4 + r) x the 16 pixels of its tile ----------------
    const int ot = wwt * 16 + l15;  // this lane's tile
    const int oy = y0 + 4 * (ot / TC), ox = x0 + 4 * (ot % TC);
    f32x4 ysp[16];  // [sub-pixel a * 4 + c][r]: tile s of the MFMA D layout [16 couts][16 tiles]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nl = wco * 16 + kq * 4 + r;
        float tmp[4][6];
#pragma unroll
        for (int jx = 0; jx < 6; ++jx) {
            const float m[6] = {acc[0 + jx][r], acc[6 + jx][r], acc[12 + jx][r], acc[18 + jx][r], acc[24 + jx][r], acc[30 + jx][r]};
            float y[4];
            at6(m, y);
#pragma unroll
            for (int a = 0; a < 4; ++a) tmp[a][jx] = y[a];
        }
        const float es = (EPI || SC) ? lds_es[nl] : 0.f, eh = EPI ? lds_eh[nl] : 0.f;  // SC: es = the shortcut's bias
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float y[4];
            at6(tmp[a], y);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = y[c];
                if (EPI) v = leaky(fmaf(v, es, eh));  // bn2 + FiLM + leaky (resunet.py:151)
                if (SC) v += es;
                ysp[a * 4 + c][r] = v;
            }
        }
    }
    if constexpr (RESPRE) {  // + pre_conv(x0) at this lane's 16 pixels (resunet.py:555,165)
        const float* xr = p.res + (size_t)b * p.res_bs + (size_t)min(oy, p.H - 4) * p.W + ox;
        float4 xv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xv[a] = *reinterpret_cast<const float4*>(xr + (size_t)a * p.W);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pw = lds_es[wco * 16 + kq * 4 + r], pb = lds_eh[wco * 16 + kq * 4 + r];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ysp[a * 4 + 0][r] += fmaf(xv[a].x, pw, pb);
                ysp[a * 4 + 1][r] += fmaf(xv[a].y, pw, pb);
                ysp[a * 4 + 2][r] += fmaf(xv[a].z, pw, pb);
                ysp[a * 4 + 3][r] += fmaf(xv[a].w, pw, pb);
            }
        }
    }
    if constexpr (SC) {
        // ---- 1x1 shortcut over the raw block input (resunet.py:163), direct: per 4 input channels 16 MFMAs, one per sub-pixel;
        // B[k = kq][col = l15] = x[channel 4 ks + kq][this lane's tile, sub-pixel s] (four 16-byte row loads),
        // A[row = l15][k = kq] = Wsc[cout wco * 16 + l15][channel 4 ks + kq]; operands of k-step ks + 1 are requested first
        const float* x2 = p.in2 + (size_t)b * p.in2_bs + (size_t)min(oy, p.H - 4) * p.W + ox;
        const float* wsc = p.w2 + n0 + wco * 16 + l15;  // [Cin2][Nw]
        const int nks = p.Cin2 / 4;
        // operands of the next THREE k-steps are in flight behind the 16 MFMAs of the current one (512 cycles: less than one
        // L2 round trip under load); the accumulators of the main phase are dead here, registers are free
        constexpr int NSB = 4;
        float4 xb[NSB][4];
        float wa[NSB];
        auto ldk = [&](int ks, int buf) {
            const float* xp = x2 + (size_t)(4 * ks + kq) * HW;
#pragma unroll
            for (int a = 0; a < 4; ++a) xb[buf][a] = *reinterpret_cast<const float4*>(xp + (size_t)a * p.W);
            wa[buf] = wsc[(size_t)(4 * ks + kq) * p.Nw];
        };
        auto mmk = [&](int buf) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ysp[a * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[buf], xb[buf][a].x, ysp[a * 4 + 0], 0, 0, 0);
                ysp[a * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[buf], xb[buf][a].y, ysp[a * 4 + 1], 0, 0, 0);
                ysp[a * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[buf], xb[buf][a].z, ysp[a * 4 + 2], 0, 0, 0);
                ysp[a * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[buf], xb[buf][a].w, ysp[a * 4 + 3], 0, 0, 0);
            }
        };
        ldk(0, 0);
        ldk(1, 1);
        ldk(2, 2);
        for (int ks = 0; ks < nks; ks += NSB) {  // Cin2 % 16 == 0 (host-checked): whole groups of NSB k-steps
#pragma unroll
            for (int u = 0; u < NSB; ++u) {
                ldk(min(ks + u + 3, nks - 1), (u + 3) % NSB);  // (behind the end: the last k-step again, unused)
                __builtin_amdgcn_sched_barrier(0);
                mmk(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (MASK) {
        // ---- fused output head (resunet.py:570-574,436-519): after_conv needs all 32 channels of a pixel; they sit in the 4 kq
        // lane groups of the 2 cout waves.  Every lane forms its partial logits (3 x 16 pixels over its 4 channels) and leaves
        // them in the dead U / V region, [source = wco * 4 + kq][logit][pixel = tile * 16 + s]; then every thread finishes 2 pixels.
        float* part = lds;  // 8 * 3 * 512 floats = 48 KiB <= U_F + V_F
        lds_barrier();      // every wave is past its last MFMA phase
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float w4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w4[r] = lds_mw[q * 32 + wco * 16 + kq * 4 + r];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float4 o;
                float* op = &o.x;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 v = ysp[a * 4 + c];
                    op[c] = v[0] * w4[0] + v[1] * w4[1] + v[2] * w4[2] + v[3] * w4[3];
                }
                *reinterpret_cast<float4*>(part + ((wco * 4 + kq) * 3 + q) * 512 + ot * 16 + a * 4) = o;
            }
        }
        lds_barrier();
        const int px = tid * 2;  // pixels px, px + 1: same tile, same row
        const int mt = px >> 4, ms = px & 15;
        const int my = y0 + 4 * (mt / TC) + (ms >> 2), mx = x0 + 4 * (mt % TC) + (ms & 3);
        float lg[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float2 sum = make_float2(lds_mw[96 + q], lds_mw[96 + q]);
#pragma unroll
            for (int src = 0; src < 8; ++src) {
                const float2 v = *reinterpret_cast<const float2*>(part + (src * 3 + q) * 512 + px);
                sum.x += v.x;
                sum.y += v.y;
            }
            lg[q][0] = sum.x;
            lg[q][1] = sum.y;
        }
        if (my < p.mask_T) {
            mask_pixel(p, b, my, mx, lg[0][0], lg[1][0], lg[2][0]);
            mask_pixel(p, b, my, mx + 1, lg[0][1], lg[1][1], lg[2][1]);
        }
        return;
    }
    // ---- stores: 16-byte rows; the block's 2x2 avg-pool (resunet.py:197) from the same registers -----------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + wco * 16 + kq * 4 + r;
        float* dst = p.out + (size_t)b * p.out_bs + (size_t)n * HW + (size_t)oy * p.W + ox;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 o = make_float4(ysp[a * 4 + 0][r], ysp[a * 4 + 1][r], ysp[a * 4 + 2][r], ysp[a * 4 + 3][r]);
            if (oy + a < p.H) *reinterpret_cast<float4*>(dst + (size_t)a * p.W) = o;
        }
        if ((SC || RESPRE) && p.pool_out) {  // wave-uniform; pool_h == 2 (host-checked): row-major summation order of F.avg_pool2d
            const int Ho = p.H / 2, Wo = p.W / 2;
            float* pd = p.pool_out + (size_t)b * (p.pool_bs ? (size_t)p.pool_bs : (size_t)p.N * Ho * Wo) + (size_t)n * Ho * Wo +
                        (size_t)(oy >> 1) * Wo + (ox >> 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float2 o;
                float s0 = ysp[(2 * i) * 4 + 0][r] + ysp[(2 * i) * 4 + 1][r];
                s0 += ysp[(2 * i + 1) * 4 + 0][r];
                s0 += ysp[(2 * i + 1) * 4 + 1][r];
                float s1 = ysp[(2 * i) * 4 + 2][r] + ysp[(2 * i) * 4 + 3][r];
                s1 += ysp[(2 * i + 1) * 4 + 2][r];
                s1 += ysp[(2 * i + 1) * 4 + 3][r];
                o.x = s0 * 0.25f;
                o.y = s1 * 0.25f;
                if (oy + 2 * i + 1 < p.H) *reinterpret_cast<float2*>(pd + (size_t)i * Wo) = o;
            }
        }
    }
}

// Transform-domain weights U = G g G^T (6x6) of g = w[cout][cin][3][3], stored as the LDS images the kernel DMAs: slab
// (chunk = cin / 8, group = cout / 32) of 36 * 256 floats, element [xi][t = (cout % 32) / 16][kq = cin % 4][l15 = cout % 16]
// [k = (cin % 8) / 4].
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ U) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (cin, cout), cout fastest
    if (i >= (long)Cout * Cin) return;
    const int co = (int)(i % Cout), ci = (int)(i / Cout);
    const float* g = w + ((size_t)co * Cin + ci) * 9;
    const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    double t[6][3];
    for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 3; ++c) t[a][c] = G[a][0] * (double)g[0 * 3 + c] + G[a][1] * (double)g[1 * 3 + c] + G[a][2] * (double)g[2 * 3 + c];
    float* slab = U + ((size_t)(ci / 8) * (Cout / 32) + co / 32) * U_F;
    const int e = (((co & 31) >> 4) * 4 + (ci & 3)) * 32 + (co & 15) * 2 + ((ci & 7) >> 2);
    for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 6; ++c) {
            const double u = t[a][0] * G[c][0] + t[a][1] * G[c][1] + t[a][2] * G[c][2];
            slab[(a * 6 + c) * 256 + e] = (float)u;
        }
}

template <int TC, int FLAGS, int NG>
hipError_t launch_wino4_tc(const ConvArgs& p0, hipStream_t stream) {
    ConvArgs p = p0;
    constexpr int OR_ = 4 * (32 / TC), OC = 4 * TC;
    p.gx = (p.W / OC) * ((p.H + OR_ - 1) / OR_);
    p.gy = p.N / (32 * NG);
    static const int xcd = [] { const char* e = getenv("LASS_XCD_MAP"); return e ? atoi(e) : 2; }();
    p.xcd_map = (xcd && ((long)p.gx * p.B) % 8 == 0 && (p.gy > 1 || xcd == 2)) ? xcd : 0;
    hipLaunchKernelGGL((wino4_kernel<TC, FLAGS, NG>), dim3((unsigned)((long)p.gx * p.gy * p.B)), dim3(NTHREADS * NG), 0, stream, p);
    return hipGetLastError();
}

// LASS_WINO4_NG=2 runs the launches with N % 64 == 0 at 64 couts per workgroup (NG = 2).  Measured A/B on one box, round 5
// (gpurun_out/r5j): 983 vs 1 015 clips/s, conv3x3 class 15.3 vs 14.4 ms - SLOWER by 6 %: the transform count per MFMA halves
// as designed, but the single 8-wave workgroup of a CU meets at two barriers per chunk with no second workgroup to run in its
// waits, and U (72 KiB) + V (36 KiB) leave no room to double-buffer either operand.  Default stays NG = 1.
template <int FLAGS>
hipError_t launch_wino4(const ConvArgs& p, hipStream_t stream) {
    constexpr bool CAN2 = (FLAGS & (F_MASK | F_PRECONV | F_RESPRE)) == 0;
    const char* ng_env = getenv("LASS_WINO4_NG");  // (read per launch: the parity test switches it inside one process)
    const int ngmax = ng_env ? atoi(ng_env) : 1;
    if constexpr (CAN2) {
        if (ngmax >= 2 && p.N % 64 == 0) {
            if (p.W % 64 == 0 && p.H % 8 == 0) return launch_wino4_tc<16, FLAGS, 2>(p, stream);
            if (p.H % 16 == 0) return launch_wino4_tc<8, FLAGS, 2>(p, stream);
            return hipErrorInvalidValue;
        }
    }
    if (p.W % 64 == 0 && p.H % 8 == 0) return launch_wino4_tc<16, FLAGS, 1>(p, stream);  // 8 rows x 64 columns
    if (p.H % 16 == 0) return launch_wino4_tc<8, FLAGS, 1>(p, stream);                   // 16 rows x 32 columns
    return hipErrorInvalidValue;
}

}  // namespace

bool lass_wino4_supported(ConvKind kind, const ConvArgs& p) {
    if (!(p.w_wino4 && ((p.W % 64 == 0 && p.H % 8 == 0) || (p.W % 32 == 0 && p.H % 16 == 0)) && p.Cin % KC == 0 && p.N % 32 == 0 &&
          p.Nw % 32 == 0 && (unsigned long long)p.Cin * p.H * p.W * 4ull < 0xFFFF0000ull))
        return false;
    switch (kind) {
        case CONV1_ACT:
            return true;
        case CONV1_ACT_PRE:  // encoder_block1.conv1: the 32 input channels are formed from x0
            return p.pre_w && p.pre_b && p.Cin == 32;
        case CONV2_IDENT_PRE:  // encoder_block1.conv2: residual = pre_conv(x0), fused 2x2 avg-pool
            return p.res && p.pre_w && p.pre_b && p.N == 32 && p.Nw == 32 && (!p.pool_out || p.pool_h == 2);
        case CONV2_SHORTCUT:  // conv2 + 1x1 shortcut (+ fused 2x2 avg-pool, or decoder_block6's fused output head)
            if (!(p.in2 && p.w2 && p.bias && p.Cin2 % 16 == 0 && (!p.pool_out || p.pool_h == 2))) return false;
            if (p.mask_re)
                return p.N == 32 && p.Nw == 32 && p.W + 1 == p.mask_nbins && p.mask_w && p.mask_b && p.mask_mag && p.mask_cos && p.mask_sin &&
                       p.mask_im && p.mask_T > 0 && p.mask_T <= p.H && !p.pool_out;
            return true;
        default:
            return false;
    }
}

hipError_t lass_launch_wino4(ConvKind kind, const ConvArgs& p, hipStream_t stream) {
    if (!lass_wino4_supported(kind, p) || !p.in || (!p.out && !p.mask_re)) return hipErrorInvalidValue;
    switch (kind) {
        case CONV1_ACT:
        case CONV1_ACT_PRE:
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift) return hipErrorInvalidValue;
            return kind == CONV1_ACT ? launch_wino4<F_PRO | F_EPIACT>(p, stream) : launch_wino4<F_PRO | F_EPIACT | F_PRECONV>(p, stream);
        case CONV2_IDENT_PRE:
            return launch_wino4<F_RES | F_RESPRE>(p, stream);
        case CONV2_SHORTCUT:
            return p.mask_re ? launch_wino4<F_PHASEB | F_BIAS | F_MASK>(p, stream) : launch_wino4<F_PHASEB | F_BIAS>(p, stream);
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t lass_launch_wino4_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream) {
    const long n = (long)Cout * Cin;
    hipLaunchKernelGGL(wino4_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin, U);
    return hipGetLastError();
}
