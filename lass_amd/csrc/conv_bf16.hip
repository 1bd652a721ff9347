// conv_bf16.hip - BASELINE configs[2]: the 3x3 (+ fused 1x1 shortcut) convolutions on the bf16 MFMA.
//
// Same semantics, tensors (NCHW f32 in HBM), prologue / epilogue fusions and accumulator layout as conv.hip; only the
// contraction runs on v_mfma_f32_32x32x16_bf16 (f32 accumulate, 16x the f32-MFMA rate).  Activations are rounded to
// bf16 (RNE) while they are staged into LDS - after the f32 BN+FiLM+leaky prologue - and weights are converted once in
// lass_finalize.  This is reduced precision by design (8-bit mantissa operands): it is selected only with
// lass_finalize(ctx, LASS_COMPUTE_BF16) and has its own, looser parity tests.
//
// LDS images are K-contiguous so that BOTH operands of an MFMA are one conflict-free ds_read_b128:
//   input   [cin-octet (2)][row][col][8 x bf16]      lane (h, p): octet h, pixel p  (+ tap shift = whole 16-B units)
//   weights [tap][cin-octet (2)][cout][8 x bf16]     lane (h, n): octet h, cout n
// A chunk is 16 input channels = one MFMA K; per chunk a wave issues TAPS x NCO x NPX MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "conv_common.h"
#include "kernels.h"
#include "wino_common.h"  // make_rsrc_words, lds_dma_16B, wait_vmcnt

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KB = 16;  // input channels per chunk

// SPLIT = 1: operands rounded to bf16.  SPLIT = 2: every f32 operand is split x = hi + lo (two bf16) and the product is
// formed as hi*hi + hi*lo + lo*hi (three MFMAs, f32 accumulate): ~16 mantissa bits per operand, error ~2^-17 per product.
// PRE: the input is the single-channel x0; channel c = pre_w[c]*x0 + pre_b[c] (pre_conv, resunet.py:555) is formed while
// staging, so a chunk needs no activation loads at all (x0 at this thread's pixels is fetched once per tile).
// DMA: the input is the block's bf16 intermediate in the blocked layout [C/8][H][W][8 x bf16] (16-B units; written by
// conv1's epilogue, lo plane separately for SPLIT = 2): staging is a pure copy, done by LDS-DMA with per-lane unit offsets
// (out-of-image units get an out-of-range offset: the buffer bounds check returns the zero padding) - no VGPRs, no VALU.
template <int TAPS, int NCO, int NPX, int PW, bool PRO, int SPLIT, bool PRE = false, bool DMA = false>
struct Phase16 {
    static constexpr int PH = 32 / PW;
    static constexpr int WROWS = NPX * PH;
    static constexpr int PHT = 4 * WROWS;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int IR = PHT + 2 * HALO;
    static constexpr int IP = PW + 2 * HALO;
    static constexpr int NT = 32 * NCO;
    static constexpr int NPIX = IR * IP;
    static constexpr int NPP = (NPIX + NTHREADS - 1) / NTHREADS;  // pixel passes (each thread: one pixel, 8 channels)
    static constexpr int NPIECE = (2 * NPIX + 63) / 64;           // DMA: 1-KiB pieces (64 units) per plane
    static constexpr int NPC = (NPIECE + 3) / 4;                  // DMA: pieces per wave
    static constexpr int IN1_U4 = 2 * NPIX;                       // 16-byte units, one plane (hi or lo)
    static constexpr int W1_U4 = TAPS * 2 * NT;
    static constexpr int IN_U4 = SPLIT * IN1_U4;
    static constexpr int W_U4 = SPLIT * W1_U4;
    static constexpr int NWLD = (W1_U4 + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_U4 = IN_U4 + W_U4;

    unsigned goff[NPP];  // BYTE offset of this thread's pixel inside a channel plane (32-bit lane part of a buffer address)
    unsigned woff[NWLD]; // BYTE offset of this thread's 16-B weight units inside a chunk's slab
    unsigned okbits;
    unsigned dvo[DMA ? NPC : 1];  // DMA: byte offset of this lane's unit in piece wave + 4*i (0xC0000000 = zero fill)
    // DMA kernels stage the weight slab by LDS-DMA too: rows (tap, octet) of NT couts x 16 B; a 1-KiB piece is RPP rows
    // (bf16 mode only: with split operands the doubled image + slab, double-buffered, would leave one workgroup per CU -
    // measured 8 % slower - so bf16x3 keeps the register-staged single weight region)
    static constexpr bool WDMA = DMA && SPLIT == 1;
    static constexpr int RPP = 64 / NT;            // 1 (64-cout tiles) or 2 (32-cout tiles)
    static constexpr int NWPIECE = W1_U4 / 64;     // pieces per plane (hi or lo): 18 / 9 (3x3), 2 / 1 (1x1)
    static_assert(W1_U4 % 64 == 0, "whole pieces");
    static constexpr int NWPC = (NWPIECE + 3) / 4; // pieces per wave
    unsigned wvo;                                  // this lane's byte offset inside a weight piece's source rows
    float v[(PRE || DMA) ? 1 : 2][(PRE || DMA) ? 1 : NPP][(PRE || DMA) ? 1 : 8];  // prefetched f32 activations: [octet][pass][channel in octet]
    float x0v[PRE ? NPP : 1];        // PRE: x0 at this thread's pixels
    float pcw[PRE ? KB : 1], pcb[PRE ? KB : 1];  // PRE: pre_conv weight / bias of the prefetched chunk's channels
    uint4 wv[(DMA && SPLIT == 1) ? 1 : SPLIT][(DMA && SPLIT == 1) ? 1 : NWLD];  // prefetched bf16 weights (hi, lo); unused when the slab is DMA'd
    float psc[KB], psh[KB];

    __device__ __forceinline__ static int upos(int tid, int k) {
        const int u = tid + k * NTHREADS;
        return u < NPIX ? u : NPIX - 1;
    }
    __device__ __forceinline__ void init(int tid, int y0, int x0, int H, int W) {
        okbits = 0;
#pragma unroll
        for (int k = 0; k < NPP; ++k) {
            const int u = upos(tid, k);
            const int r = u / IP, x = u % IP;
            const int gy = y0 + r - HALO, gx = x0 + x - HALO;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            goff[k] = 4u * (unsigned)(min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1));
            okbits |= (ok ? 1u : 0u) << k;
        }
    }
    __device__ __forceinline__ void init_w(int tid, int Cout) {
#pragma unroll
        for (int i = 0; i < NWLD; ++i) {
            const int e0 = tid + i * NTHREADS;
            const int e = e0 < W1_U4 ? e0 : W1_U4 - 1;
            const int row = e / NT, col = e % NT;  // row = tap*2 + octet
            woff[i] = 16u * (unsigned)(row * Cout + col);
        }
    }
    __device__ __forceinline__ void init_dma(int lane, int wave, int y0, int x0, int H, int W) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int e = (wave + 4 * i) * 64 + lane;  // unit index in the [octet][NPIX] image
            const int o = e / NPIX, u = e % NPIX;
            const int r = u / IP, x = u % IP;
            const int gy = y0 + r - HALO, gx = x0 + x - HALO;
            const bool ok = e < 2 * NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            // 0xC0000000: beyond any descriptor -> zero fill (padding); 0xFFFFFFFF: lane past the image -> not issued
            dvo[i] = ok ? 16u * (unsigned)((o * H + gy) * W + gx) : (e < 2 * NPIX ? 0xC0000000u : 0xFFFFFFFFu);
        }
    }
    __device__ __forceinline__ void init_wdma(int lane, int Cout) {
        wvo = 16u * (unsigned)((lane / NT) * Cout + lane % NT);
    }
    // w_rs / wl_rs: weight matrix (hi / lo) from column n0 on; wb: byte offset of the chunk's slab [tap][octet][Cout];
    // wl_addr: LDS byte address of the weight buffer to fill
    // [P0, P1): the pieces to request (the whole slab by default; HALF_SLAB requests taps 0-4 and taps 5-8 separately)
    static constexpr int H0_PIECES = (TAPS == 9) ? 5 * 2 / RPP : NWPIECE;  // pieces of taps 0-4
    template <int P0 = 0, int P1 = NWPIECE>
    __device__ __forceinline__ void issue_wdma(v4i32 w_rs, v4i32 wl_rs, unsigned wb, int Cout, unsigned wl_addr, int wave) {
#pragma unroll
        for (int i = 0; i < NWPC; ++i) {
            const int piece = P0 + wave + 4 * i;  // wave-uniform
            if (piece < P1) {
                const unsigned soff = wb + (unsigned)(piece * RPP * Cout) * 16u;
                lds_dma_16B(w_rs, wvo, soff, wl_addr + (unsigned)piece * 1024u);
                if (SPLIT == 2) lds_dma_16B(wl_rs, wvo, soff, wl_addr + (unsigned)(W1_U4 * 16) + (unsigned)piece * 1024u);
            }
        }
    }
    // rs / rs_lo: descriptors of this clip's blocked bf16 planes; soff: byte offset of the chunk's first octet;
    // img: LDS byte address of the image buffer to fill
    __device__ __forceinline__ void issue_dma(v4i32 rs, v4i32 rs_lo, unsigned soff, unsigned img, int wave) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int piece = wave + 4 * i;
            // the last piece of a plane is partial: its surplus lanes are masked off (an LDS-DMA lane that is not
            // executed writes nothing), so the image needs no padding behind it
            if (piece < NPIECE && dvo[i] != 0xFFFFFFFFu) {
                lds_dma_16B(rs, dvo[i], soff, img + (unsigned)piece * 1024u);
                if (SPLIT == 2) lds_dma_16B(rs_lo, dvo[i], soff, img + (unsigned)(IN1_U4 * 16) + (unsigned)piece * 1024u);
            }
        }
    }
    // Buffer-addressed loads (descriptor + scalar byte offset + constant 32-bit lane offset: no VALU address arithmetic).
    // in_rs: this clip's input planes, c0b = byte offset of the chunk's first channel; w_rs / wl_rs: the weight matrix
    // (hi / lo) from column n0 on, wb = byte offset of the chunk's slab [tap][octet][Cout].
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t in_rs, unsigned c0b, int HW, __amdgpu_buffer_rsrc_t w_rs,
                                         __amdgpu_buffer_rsrc_t wl_rs, unsigned wb, const float* __restrict__ sc,
                                         const float* __restrict__ sh, const float* __restrict__ pw = nullptr,
                                         const float* __restrict__ pb = nullptr) {
        if (!PRE && !DMA) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int k = 0; k < NPP; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[o][k][j] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(in_rs, (int)goff[k],
                                                                        (int)(c0b + (unsigned)((o * 8 + j) * HW) * 4u), 0));
        } else if (PRE) {
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                pcw[c] = pw[c];
                pcb[c] = pb[c];
            }
        }
#pragma unroll
        for (int i = 0; i < (WDMA ? 0 : NWLD); ++i) {
            wv[0][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)woff[i], (int)wb, 0));
            if (SPLIT == 2)
                wv[1][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wl_rs, (int)woff[i], (int)wb, 0));
        }
        if (PRO) {
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                psc[c] = sc[c];
                psh[c] = sh[c];
            }
        }
    }
    __device__ __forceinline__ void load_x0(__amdgpu_buffer_rsrc_t in_rs) {  // PRE only: once per tile
#pragma unroll
        for (int k = 0; k < NPP; ++k)
            x0v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rs, (int)goff[k], 0, 0));
    }
    // lds: image buffer (non-DMA); wl: weight region
    __device__ __forceinline__ void store(uint4* lds, uint4* wl, int tid) {
#pragma unroll
        for (int o = 0; o < (DMA ? 0 : 2); ++o)
#pragma unroll
            for (int k = 0; k < NPP; ++k) {
                const int u = upos(tid, k);
                const bool ok = (okbits >> k) & 1u;
                bf16x8 pk, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = PRE ? x0v[k] * pcw[o * 8 + j] + pcb[o * 8 + j]
                                  : v[(PRE || DMA) ? 0 : o][(PRE || DMA) ? 0 : k][(PRE || DMA) ? 0 : j];
                    if (PRO) t = leaky(t * psc[o * 8 + j] + psh[o * 8 + j]);
                    t = ok ? t : 0.f;  // conv zero padding comes after the activation
                    pk[j] = (__bf16)t;
                    if (SPLIT == 2) pl[j] = (__bf16)(t - (float)pk[j]);
                }
                *reinterpret_cast<bf16x8*>(lds + o * NPIX + u) = pk;
                if (SPLIT == 2) *reinterpret_cast<bf16x8*>(lds + IN1_U4 + o * NPIX + u) = pl;
            }
#pragma unroll
        for (int i = 0; i < (WDMA ? 0 : NWLD); ++i) {
            const int e0 = tid + i * NTHREADS;
            const int e = e0 < W1_U4 ? e0 : W1_U4 - 1;
            wl[e] = wv[0][i];
            if (SPLIT == 2) wl[W1_U4 + e] = wv[1][i];
        }
    }
    template <int T0 = 0, int T1 = TAPS>
    __device__ __forceinline__ static void compute(const uint4* lds, const uint4* wl, f32x16 (&acc)[NCO][NPX], int lane,
                                                   int wave) {
        const int h = lane >> 5, j = lane & 31;
        const int ty = j / PW, tx = j % PW;
        const bf16x8* bbase = reinterpret_cast<const bf16x8*>(lds) + h * NPIX + (wave * WROWS + ty) * IP + tx;
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(wl) + h * NT + j;
        if (SPLIT == 2) {
#pragma unroll
            for (int tap = T0; tap < T1; ++tap) {
                bf16x8 ah[NCO], al[NCO], bh[NPX], bl[NPX];
#pragma unroll
                for (int co = 0; co < NCO; ++co) {
                    ah[co] = abase[tap * 2 * NT + co * 32];
                    al[co] = abase[W1_U4 + tap * 2 * NT + co * 32];
                }
#pragma unroll
                for (int px = 0; px < NPX; ++px) {
                    const int off = (px * PH + (TAPS == 9 ? tap / 3 : 0)) * IP + (TAPS == 9 ? tap % 3 : 0);
                    bh[px] = bbase[off];
                    bl[px] = bbase[IN1_U4 + off];
                }
#pragma unroll
                for (int co = 0; co < NCO; ++co)
#pragma unroll
                    for (int px = 0; px < NPX; ++px) {  // small terms first
                        acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[co], bh[px], acc[co][px], 0, 0, 0);
                        acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[co], bl[px], acc[co][px], 0, 0, 0);
                        acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[co], bh[px], acc[co][px], 0, 0, 0);
                    }
            }
            return;
        }
        // fragments of tap + PFD are read before the MFMAs of tap (pinned): PFD taps = PFD * NCO * NPX MFMAs of cover for
        // the LDS latency
        constexpr int PFD = TAPS > 2 ? 2 : 1;
        bf16x8 a[PFD + 1][NCO], b[PFD + 1][NPX];
        auto rd = [&](int tap, bf16x8 (&aa)[NCO], bf16x8 (&bb)[NPX]) {
#pragma unroll
            for (int co = 0; co < NCO; ++co) aa[co] = abase[tap * 2 * NT + co * 32];
#pragma unroll
            for (int px = 0; px < NPX; ++px)
                bb[px] = bbase[(px * PH + (TAPS == 9 ? tap / 3 : 0)) * IP + (TAPS == 9 ? tap % 3 : 0)];
        };
#pragma unroll
        for (int t = T0; t < T0 + PFD && t < T1; ++t) rd(t, a[t % (PFD + 1)], b[t % (PFD + 1)]);
#pragma unroll
        for (int tap = T0; tap < T1; ++tap) {
            if (tap + PFD < T1) rd(tap + PFD, a[(tap + PFD) % (PFD + 1)], b[(tap + PFD) % (PFD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int co = 0; co < NCO; ++co)
#pragma unroll
                for (int px = 0; px < NPX; ++px)
                    acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap % (PFD + 1)][co], b[tap % (PFD + 1)][px],
                                                                          acc[co][px], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

template <int A, int B>
struct MaxU {
    static constexpr int v = A > B ? A : B;
};

// Register-fed 1x1 contraction (round 5).  With the blocked bf16 layout [C/8][H][W][8] a lane's B fragment of
// v_mfma_f32_32x32x16_bf16 - pixel j, input-channel octet 2c + khalf - is ONE aligned 16-byte unit of the tensor, and its A
// fragment - cout j, the same octet - one 16-byte unit of the weight matrix [chunk][octet][Cout][8]: both go from global
// memory straight into the MFMA operands (half a wave reads 512 contiguous bytes).  A 1x1 chunk is FOUR MFMAs per wave; fed
// through LDS it cost a DMA round trip and a workgroup barrier per chunk (850-1 080 cycles: DESIGN.md section 8).  Two uses:
//  * the 1x1 shortcut of a ConvBlockRes (resunet.py:163-165) rides in the 3x3 chunk loop - the fragments of shortcut chunks
//    q*ch .. q*ch + q - 1 are requested while 3x3 chunk ch - 1 is contracted and consumed right behind chunk ch's barrier,
//    8 MFMAs beside its 36 - instead of running as its own barriered phase behind the main loop;
//  * the transposed convolutions (kernel = stride: a pointwise GEMM, resunet.py:216-224) run without LDS operands and
//    without a barrier in their K loop.
// Lane offsets: bvo[px] = byte offset of this lane's unit in octet khalf of a chunk (rows past the image: 0xC0000000 = the
// descriptor's bounds check returns zeros), avo[co] = the same inside a chunk's weight rows.
template <int NCO, int NPX, int PW>
struct Reg1x1 {
    static constexpr int PH = 32 / PW, WROWS = NPX * PH;
    unsigned bvo[NPX], avo[NCO];
    __device__ __forceinline__ void init(int lane, int wave, int y0, int x0, int H, int W, int Nw) {
        const int h = lane >> 5, j = lane & 31, ty = j / PW, tx = j % PW;
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = y0 + wave * WROWS + px * PH + ty;
            bvo[px] = y < H ? 16u * (unsigned)((h * H + y) * W + x0 + tx) : 0xC0000000u;
        }
#pragma unroll
        for (int co = 0; co < NCO; ++co) avo[co] = 16u * (unsigned)(h * Nw + co * 32 + j);
    }
    // chunk c of the tensor behind `in_rs` (one clip, HW pixels per octet plane) and of the weights behind `w_rs` (row pitch Nw)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t in_rs, __amdgpu_buffer_rsrc_t w_rs, int c, int HW, int Nw,
                                         bf16x8 (&a)[NCO], bf16x8 (&b)[NPX]) const {
#pragma unroll
        for (int co = 0; co < NCO; ++co)
            a[co] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)avo[co], (int)((unsigned)(c * 2 * Nw) * 16u), 0));
#pragma unroll
        for (int px = 0; px < NPX; ++px)
            b[px] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(in_rs, (int)bvo[px], (int)((unsigned)(c * 2 * HW) * 16u), 0));
    }
    __device__ __forceinline__ static void mfma(const bf16x8 (&a)[NCO], const bf16x8 (&b)[NPX], f32x16 (&acc)[NCO][NPX]) {
#pragma unroll
        for (int co = 0; co < NCO; ++co)
#pragma unroll
            for (int px = 0; px < NPX; ++px)
                acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[co], b[px], acc[co][px], 0, 0, 0);
    }
};

#ifndef LASS_HALF_SLAB
#define LASS_HALF_SLAB 1
#endif
// The instantiations that run the half-slab schedule (see HALF_SLAB below) are built for three waves per SIMD: the encoder
// blocks' conv2 + shortcut kernels of the blocked-bf16 pipeline.  Measured per launch against the two-slab build: encoder
// conv2 + shortcut (154 registers, no spill) 0.906; decoder conv2 + shortcut, whose activated output needs two more tables,
// 0.981 at 168 registers WITH 100 B of scratch per thread (0.2 GB of spill traffic per launch) - left at two slabs; the
// conv1 kernels, whose launch is nearly all main phase, 1.022 - left at two slabs and two workgroups per CU.
template <int TAPS, int NCO, int NPX, int FLAGS, int SPLIT>
constexpr bool half_slab_v = LASS_HALF_SLAB && (FLAGS & F_INBF16) != 0 && (FLAGS & F_PHASEB) != 0 && (FLAGS & F_EPIACT) == 0 && SPLIT == 1 &&
                             TAPS == 9 && NCO == 2 && NPX == 2;

template <int TAPS, int NCO, int NPX, int PW, int FLAGS, int SPLIT>
__global__ __launch_bounds__(NTHREADS, (half_slab_v<TAPS, NCO, NPX, FLAGS, SPLIT> ? 3 : 1)) void conv_bf16_kernel(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr bool HASB = (FLAGS & F_PHASEB) != 0;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;
    constexpr bool BIAS = (FLAGS & F_BIAS) != 0;
    constexpr bool RES = (FLAGS & F_RES) != 0;
    constexpr bool RES_PF = RES && NCO == 1;
    constexpr bool PRE = (FLAGS & F_PRECONV) != 0;
    constexpr bool RESPRE = (FLAGS & F_RESPRE) != 0;
    static_assert(!RESPRE || RES_PF, "x0-derived residual needs the register-prefetch path");
    constexpr bool INBF = (FLAGS & F_INBF16) != 0;    // phase A reads the blocked bf16 intermediate by LDS-DMA
    using PA = Phase16<TAPS, NCO, NPX, PW, PRO, SPLIT, PRE, INBF>;
    constexpr bool IN2BF = (FLAGS & F_IN2BF16) != 0;  // phase B reads the blocked bf16 raw copy by LDS-DMA
    using PB = Phase16<1, NCO, NPX, PW, false, SPLIT, false, IN2BF>;
    // DMA-fed phases: image double-buffered; the weight slab too when it is DMA'd (bf16), else one register-staged region
    // HALF_SLAB (bf16, 64-cout tiles): ONE weight slab, refilled half by half (taps 0-4 while taps 5-8 of the previous chunk
    // are contracted, taps 5-8 while taps 0-4 are) instead of two whole slabs: 40 KB instead of 59 KB of LDS per workgroup =
    // THREE workgroups per CU.  Each half has half a chunk of MFMAs as cover instead of a whole one; the third workgroup more
    // than pays for that (prologue, epilogue and shortcut of one workgroup now run beside TWO others' MFMA phases).
    constexpr bool HALF_SLAB = half_slab_v<TAPS, NCO, NPX, FLAGS, SPLIT>;
    static_assert(!HALF_SLAB || PA::WDMA, "half-slab schedule needs the DMA'd weight slab");
    // Round 5: 1x1 contractions whose operands are blocked bf16 go from global memory straight into the MFMA (Reg1x1):
    // RF_MAIN = the whole K loop of a 1x1 kernel (transposed convs), RF_SC = the 1x1 shortcut folded into the 3x3 chunk loop
    // (at most RF_Q shortcut chunks per 3x3 chunk; a launch with more falls back to the ring schedule behind the main loop).
#ifndef LASS_RF1X1
#define LASS_RF1X1 1
#endif
    constexpr bool RF_MAIN = LASS_RF1X1 && TAPS == 1 && INBF && PA::WDMA && !HASB;
    constexpr bool RF_SC = LASS_RF1X1 && TAPS == 9 && HASB && IN2BF && INBF && PA::WDMA && PB::WDMA;
    constexpr int RF_Q = HALF_SLAB ? 1 : 2;  // (the half-slab kernels live within 168 registers: three workgroups per CU)
    constexpr int PA_LDS = INBF ? 2 * PA::IN_U4 + (PA::WDMA && !HALF_SLAB ? 2 : 1) * PA::W_U4 : PA::LDS_U4;
    constexpr int PB_LDS = IN2BF ? (PB::WDMA ? 4 : 2) * PB::IN_U4 + (PB::WDMA ? 4 : 1) * PB::W_U4 : PB::LDS_U4;  // WDMA: >= 4 ring slots
    constexpr int LDS_U4 = HASB ? MaxU<PA_LDS, PB_LDS>::v : PA_LDS;
    constexpr int PH = PA::PH, WROWS = PA::WROWS, PHT = PA::PHT, NT = PA::NT;
    constexpr bool MASK = (FLAGS & F_MASK) != 0;
    constexpr bool OUTBF = (FLAGS & F_OUTBF16) != 0;
    constexpr bool TCV = (FLAGS & F_TCONV) != 0;
    constexpr int NTAB = (EPI ? 2 * NT : 0) + (BIAS ? NT : 0) + (MASK ? 100 : 0) + (OUTBF ? 4 * NT : 0) + (TCV ? 16 * NCO : 0);

    __shared__ uint4 lds4[LDS_U4 + (NTAB + 3) / 4];
    float* tabs = reinterpret_cast<float*>(lds4 + LDS_U4);
    float* lds_es = tabs;
    float* lds_eh = tabs + NT;
    float* lds_bias = tabs + (EPI ? 2 * NT : 0);
    float* lds_mw = tabs + (EPI ? 2 * NT : 0) + (BIAS ? NT : 0);  // MASK: after_conv weight [3][32] + bias [3]
    float* lds_act = tabs + NTAB - 4 * NT - (TCV ? 16 * NCO : 0);  // OUTBF: activation tables of the blocked copies (skip, pooled)
    float* lds_tact = tabs + NTAB - 16 * NCO;  // TCONV with blocked outputs: consumer prologue of this tile's 8*NCO channels

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx_, by_, b;
    block_coords(p, bx_, by_, b);
    const int n0 = by_ * NT;
    const int tiles_x = p.W / PW;
    const int y0 = (bx_ / tiles_x) * PHT, x0 = (bx_ % tiles_x) * PW;
    const int HW = p.H * p.W;
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;
    const int nA = p.Cin / KB;
    const int nB = HASB ? p.Cin2 / KB : 0;
    const float* in_b = INBF ? nullptr : p.in + (size_t)b * p.in_bs;
    const float* in2_b = (HASB && !IN2BF) ? p.in2 + (size_t)b * p.in2_bs : nullptr;
    const float* sc = PRO ? p.pro_scale : nullptr;
    const float* sh = PRO ? p.pro_shift + (size_t)b * p.pro_shift_bs : nullptr;
    const uint4* wa = reinterpret_cast<const uint4*>(p.w_bf16) + n0;    // [chunk][tap][octet][Cout] 16-B units
    const uint4* wb2 = HASB ? reinterpret_cast<const uint4*>(p.w2_bf16) + n0 : nullptr;
    const uint4* wa_lo = SPLIT == 2 ? reinterpret_cast<const uint4*>(p.w_bf16_lo) + n0 : nullptr;
    const uint4* wb2_lo = (HASB && SPLIT == 2) ? reinterpret_cast<const uint4*>(p.w2_bf16_lo) + n0 : nullptr;

    if (EPI && tid < NT) {
        lds_es[tid] = p.epi_scale[n0 + tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + n0 + tid];
    }
    if (BIAS && tid < NT) lds_bias[tid] = p.bias[n0 + tid];
    if (MASK && tid < 99) lds_mw[tid] = tid < 96 ? p.mask_w[tid] : p.mask_b[tid - 96];
    if (TCV && p.out_bf16 && tid < 8 * NCO) {
        lds_tact[tid] = p.act_scale[n0 / 4 + tid];
        lds_tact[8 * NCO + tid] = p.act_shift[(size_t)b * p.act_shift_bs + n0 / 4 + tid];
    }
    if (OUTBF && tid < NT) {
        if (p.out_bf16_act) {
            lds_act[tid] = p.act_scale[n0 + tid];
            lds_act[NT + tid] = p.act_shift[(size_t)b * p.act_shift_bs + n0 + tid];
        }
        if (p.pool_bf16) {
            lds_act[2 * NT + tid] = p.pool_act_scale[n0 + tid];
            lds_act[3 * NT + tid] = p.pool_act_shift[(size_t)b * p.act_shift_bs + n0 + tid];
        }
    }

#ifdef LASS_CONV_DIAG
    const int EXPF = p.exp;  // timing experiments (LASS_EXP; wrong results): 1 no weight DMA, 2 no image DMA, 8 no MFMA
    const long long dg_t0 = clock64();
    long long dg_t1 = dg_t0, dg_t2 = dg_t0, dg_t3 = dg_t0;
#else
    constexpr int EXPF = 0;
#endif
    PA pa;
    PB pb;
    const auto rs = [](const void* ptr, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t in_rs = rs(in_b, INBF ? 0 : (long)(PRE ? 1 : p.Cin) * HW * 4);
    const __amdgpu_buffer_rsrc_t wa_rs = rs(wa, ((long)(p.Cin / KB) * TAPS * 2 * p.Nw - n0) * 16);
    const __amdgpu_buffer_rsrc_t wal_rs = SPLIT == 2 ? rs(wa_lo, ((long)(p.Cin / KB) * TAPS * 2 * p.Nw - n0) * 16) : wa_rs;
    const __amdgpu_buffer_rsrc_t in2_rs = (HASB && !IN2BF) ? rs(in2_b, (long)p.Cin2 * HW * 4) : in_rs;
    const __amdgpu_buffer_rsrc_t wb_rs = HASB ? rs(wb2, ((long)(p.Cin2 / KB) * 2 * p.Nw - n0) * 16) : wa_rs;
    const __amdgpu_buffer_rsrc_t wbl_rs = (HASB && SPLIT == 2) ? rs(wb2_lo, ((long)(p.Cin2 / KB) * 2 * p.Nw - n0) * 16) : wb_rs;
    auto loadA = [&](int c) {
        pa.load(in_rs, (unsigned)(c * KB * HW) * 4u, HW, wa_rs, wal_rs, (unsigned)(c * TAPS * 2 * p.Nw) * 16u, sc + c * KB,
                sh + c * KB, PRE ? p.pre_w + c * KB : nullptr, PRE ? p.pre_b + c * KB : nullptr);
    };
    auto loadB = [&](int c) {
        pb.load(in2_rs, (unsigned)(c * KB * HW) * 4u, HW, wb_rs, wbl_rs, (unsigned)(c * 2 * p.Nw) * 16u, nullptr, nullptr);
    };

    uint4* wl_a = lds4 + PA::IN_U4;  // weight region of phase A (register-staged kernels)
    pa.init_w(tid, p.Nw);
    f32x16 acc[NCO][NPX];
    auto init_acc = [&]() {
#pragma unroll
        for (int co = 0; co < NCO; ++co)
#pragma unroll
            for (int px = 0; px < NPX; ++px)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[co][px][r] = BIAS ? lds_bias[co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf] : 0.f;
    };
    bool sc_folded = false;  // (wave-uniform) the 1x1 shortcut has been contracted inside the main loop
    if (INBF) {
        // Everything a chunk needs - the activation image and the weight slab - arrives by LDS-DMA into alternating
        // buffers while the previous chunk is contracted: no VGPRs, no VALU, no ds_write, ONE barrier per chunk.
        const long plane = (long)(p.Cin / 8) * HW * 16;  // bytes of one clip in the blocked layout
        const v4i32 a_rs = make_rsrc_words(reinterpret_cast<const char*>(p.in_bf16) + (size_t)b * plane, (unsigned)plane);
        const v4i32 al_rs = SPLIT == 2
                                ? make_rsrc_words(reinterpret_cast<const char*>(p.in_bf16_lo) + (size_t)b * plane, (unsigned)plane)
                                : a_rs;
        if constexpr (RF_MAIN) {
            // 1x1 main phase (the transposed convs) fed from registers: no LDS operands, no barrier in the K loop; a ring of
            // RF_D chunks of fragments (16 bytes x (NCO + NPX) per lane and chunk) is kept in flight
            Reg1x1<NCO, NPX, PW> rf;
            rf.init(lane, wave, y0, x0, p.H, p.W, p.Nw);
            const __amdgpu_buffer_rsrc_t x_rs = rs(reinterpret_cast<const char*>(p.in_bf16) + (size_t)b * plane, plane);
            __syncthreads();  // epilogue tables visible
            init_acc();
            constexpr int RF_D = 4;
            bf16x8 fa[RF_D][NCO], fb[RF_D][NPX];
#pragma unroll
            for (int s = 0; s < RF_D; ++s)
                if (s < nA) rf.load(x_rs, wa_rs, s, HW, p.Nw, fa[s], fb[s]);
            for (int ch = 0; ch < nA; ch += RF_D) {
#pragma unroll
                for (int s = 0; s < RF_D; ++s)
                    if (ch + s < nA) {
                        rf.mfma(fa[s], fb[s], acc);
                        if (ch + s + RF_D < nA) rf.load(x_rs, wa_rs, ch + s + RF_D, HW, p.Nw, fa[s], fb[s]);
                    }
            }
        } else
        if constexpr (PA::WDMA) {
        const unsigned wbytes = (unsigned)(((long)(p.Cin / KB) * TAPS * 2 * p.Nw - n0) * 16);
        const v4i32 wd_rs = make_rsrc_words(wa, wbytes);
        const v4i32 wdl_rs = SPLIT == 2 ? make_rsrc_words(wa_lo, wbytes) : wd_rs;
        const unsigned img0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
        const unsigned wl0 = img0 + (unsigned)(2 * PA::IN_U4 * 16);
        pa.init_dma(lane, wave, y0, x0, p.H, p.W);
        pa.init_wdma(lane, p.Nw);
        pa.issue_dma(a_rs, al_rs, 0u, img0, wave);
        pa.issue_wdma(wd_rs, wdl_rs, 0u, p.Nw, wl0, wave);
        __syncthreads();  // epilogue tables visible
        init_acc();
#ifdef LASS_CONV_DIAG
        dg_t1 = clock64();
#endif
        // RF_SC: shortcut chunks q*ch + s (s < q) ride with 3x3 chunk ch.  Their fragments are requested one chunk ahead
        // (ordinary loads the compiler tracks; the LDS-DMA is issued from asm and is invisible to it) and consumed right
        // behind the barrier, BEFORE the next chunk's DMA is issued: hipcc waits vmcnt(0) in front of their first use, which
        // at that point finds nothing outstanding - behind the DMA issue it would drain the next chunk's pieces.
        Reg1x1<NCO, NPX, PW> rf;
        bf16x8 sa[RF_SC ? RF_Q : 1][NCO], sb[RF_SC ? RF_Q : 1][NPX];
        const int scq = RF_SC ? (nB + nA - 1) / nA : 0;
        const bool fold = RF_SC && scq <= RF_Q;
        const __amdgpu_buffer_rsrc_t r2_rs = RF_SC ? rs(reinterpret_cast<const char*>(p.in2_bf16) + (size_t)b * ((long)(p.Cin2 / 8) * HW * 16),
                                                        (long)(p.Cin2 / 8) * HW * 16) : wa_rs;
        auto sc_load = [&](int ch) {
#pragma unroll
            for (int s = 0; s < (RF_SC ? RF_Q : 0); ++s)
                if (s < scq && ch * scq + s < nB) rf.load(r2_rs, wb_rs, ch * scq + s, HW, p.Nw, sa[s], sb[s]);
        };
        auto sc_mfma = [&](int ch) {
#pragma unroll
            for (int s = 0; s < (RF_SC ? RF_Q : 0); ++s)
                if (s < scq && ch * scq + s < nB) rf.mfma(sa[s], sb[s], acc);
        };
        if (fold) {
            rf.init(lane, wave, y0, x0, p.H, p.W, p.Nw);
            sc_load(0);
        }
        if constexpr (HALF_SLAB) {
            // (the prologue above has requested image 0 and the WHOLE slab of chunk 0)
            int img_ops = 0;  // image DMA instructions of this wave per chunk (wave-uniform)
#pragma unroll
            for (int i = 0; i < PA::NPC; ++i) img_ops += (wave + 4 * i < PA::NPIECE) ? 1 : 0;
            const uint4* slab = lds4 + 2 * PA::IN_U4;
            for (int ch = 0; ch < nA; ++ch) {
                const int cur = ch & 1;
                const bool more = ch + 1 < nA;
                wait_vmcnt<0>();  // image ch and taps 0-4 of chunk ch (chunk 0: the whole slab) have landed ...
                lds_barrier();    // ... for every wave, and every wave is past taps 5-8 of chunk ch-1
                if (fold) sc_mfma(ch);
                if (ch > 0) pa.template issue_wdma<PA::H0_PIECES, PA::NWPIECE>(wd_rs, wdl_rs, (unsigned)(ch * TAPS * 2 * p.Nw) * 16u, p.Nw, wl0, wave);
                if (more) pa.issue_dma(a_rs, al_rs, (unsigned)((ch + 1) * 2 * HW) * 16u, img0 + (unsigned)((cur ^ 1) * PA::IN_U4 * 16), wave);
                int sc_ops = 0;  // fragment loads of the next chunk's shortcut slots (younger than the pieces waited for below)
                if (fold && more) {
                    sc_load(ch + 1);
#pragma unroll
                    for (int s2 = 0; s2 < RF_Q; ++s2) sc_ops += (s2 < scq && (ch + 1) * scq + s2 < nB) ? NCO + NPX : 0;
                }
                PA::template compute<0, 5>(lds4 + cur * PA::IN_U4, slab, acc, lane, wave);
                wait_vmcnt_dyn((more ? img_ops : 0) + sc_ops);  // taps 5-8 have landed (the younger image pieces / fragments may still be under way)
                lds_barrier();                       // ... for every wave, and every wave is past taps 0-4
                if (more) pa.template issue_wdma<0, PA::H0_PIECES>(wd_rs, wdl_rs, (unsigned)((ch + 1) * TAPS * 2 * p.Nw) * 16u, p.Nw, wl0, wave);
                PA::template compute<5, 9>(lds4 + cur * PA::IN_U4, slab, acc, lane, wave);
            }
            sc_folded = fold;
        } else {
        for (int ch = 0; ch < nA; ++ch) {
            const int cur = ch & 1;
            wait_vmcnt<0>();   // this wave's pieces of chunk ch have landed
            __syncthreads();   // ... everyone's have, and everyone has finished contracting chunk ch-1
            if (fold) sc_mfma(ch);
            if (ch + 1 < nA) {
                if (!(EXPF & 2)) pa.issue_dma(a_rs, al_rs, (unsigned)((ch + 1) * 2 * HW) * 16u, img0 + (unsigned)((cur ^ 1) * PA::IN_U4 * 16), wave);
                if (!(EXPF & 1)) pa.issue_wdma(wd_rs, wdl_rs, (unsigned)((ch + 1) * TAPS * 2 * p.Nw) * 16u, p.Nw,
                              wl0 + (unsigned)((cur ^ 1) * PA::W_U4 * 16), wave);
                if (fold) sc_load(ch + 1);
            }
            if (!(EXPF & 8)) PA::compute(lds4 + cur * PA::IN_U4, lds4 + 2 * PA::IN_U4 + cur * PA::W_U4, acc, lane, wave);
        }
        sc_folded = fold;
        }
        } else {
            // split operands: image of chunk ch+1 by LDS-DMA into the other buffer while chunk ch is contracted; weights
            // through registers into the single weight region between the two barriers
            uint4* wl_d = lds4 + 2 * PA::IN_U4;
            const unsigned img0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
            const unsigned img1 = img0 + (unsigned)(PA::IN_U4 * 16);
            pa.init_dma(lane, wave, y0, x0, p.H, p.W);
            pa.issue_dma(a_rs, al_rs, 0u, img0, wave);
            loadA(0);
            __syncthreads();  // epilogue tables visible
            pa.store(lds4, wl_d, tid);
            wait_vmcnt<0>();
            __syncthreads();
            init_acc();
            for (int ch = 0; ch < nA; ++ch) {
                const bool more = ch + 1 < nA;
                if (more) {
                    pa.issue_dma(a_rs, al_rs, (unsigned)((ch + 1) * 2 * HW) * 16u, (ch & 1) ? img0 : img1, wave);
                    loadA(ch + 1);
                }
                PA::compute(lds4 + ((ch & 1) ? PA::IN_U4 : 0), wl_d, acc, lane, wave);
                __syncthreads();
                if (more) {
                    pa.store(lds4, wl_d, tid);
                    wait_vmcnt<0>();
                }
                __syncthreads();
            }
        }
    } else {
        pa.init(tid, y0, x0, p.H, p.W);
        if (PRE) pa.load_x0(in_rs);
        loadA(0);
        __syncthreads();
        pa.store(lds4, wl_a, tid);
        __syncthreads();
        init_acc();
        for (int ch = 0; ch + 1 < nA; ++ch) {
            loadA(ch + 1);
            PA::compute(lds4, wl_a, acc, lane, wave);
            __syncthreads();
            pa.store(lds4, wl_a, tid);
            __syncthreads();
        }
    }
#ifdef LASS_CONV_DIAG
    dg_t2 = clock64();
#endif
    float rtmp[RES_PF ? NPX : 1][16];
    if (HASB) {
        if (!IN2BF) pb.init(tid, y0, x0, p.H, p.W);
        pb.init_w(tid, p.Nw);
        if (!IN2BF) loadB(0);
    }
    if (RES_PF) {
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = min(y0 + wave * WROWS + px * PH + ty, p.H - 1);
            if (RESPRE) {  // residual = pre_conv(x0) at this pixel: one load, 16 FMAs (resunet.py:555,165)
                const float xv = p.res[(size_t)b * p.res_bs + (size_t)y * p.W + x];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    rtmp[px][r] = xv * p.pre_w[n] + p.pre_b[n];
                }
            } else {
                const float* src = p.res + (size_t)b * p.res_bs + (size_t)(n0 + 4 * khalf) * HW + (size_t)y * p.W + x;
#pragma unroll
                for (int r = 0; r < 16; ++r) rtmp[px][r] = src[(size_t)((r & 3) + 8 * (r >> 2)) * HW];
            }
        }
    }
    if (!INBF) PA::compute(lds4, wl_a, acc, lane, wave);  // last chunk of phase A (the INBF loop contracts all of them)
    if (HASB && IN2BF && !sc_folded) {
        // same schedule as the INBF main phase: image and weights by LDS-DMA into alternating buffers
        const long plane2 = (long)(p.Cin2 / 8) * HW * 16;
        const v4i32 r_rs = make_rsrc_words(reinterpret_cast<const char*>(p.in2_bf16) + (size_t)b * plane2, (unsigned)plane2);
        if constexpr (PB::WDMA) {
        const unsigned wbytes2 = (unsigned)(((long)(p.Cin2 / KB) * 2 * p.Nw - n0) * 16);
        const v4i32 wd_rs = make_rsrc_words(wb2, wbytes2);
        const v4i32 wdl_rs = SPLIT == 2 ? make_rsrc_words(wb2_lo, wbytes2) : wd_rs;
        const unsigned img0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
        const unsigned wl0 = img0 + (unsigned)(2 * PB::IN_U4 * 16);
        pb.init_dma(lane, wave, y0, x0, p.H, p.W);
        pb.init_wdma(lane, p.Nw);
        __syncthreads();  // phase A has finished with the LDS
#ifndef LASS_SC_TWO_BUFFERS
        {
            // A chunk of the 1x1 shortcut is ONE tap: NCO x NPX MFMAs (128 cycles) against a DMA round trip of a microsecond, and
            // Cin2 / 16 = 2 ... 48 of them follow each other - a two-buffer pipeline pays the whole latency per chunk.  The
            // chunks are small (image PB::IN_U4 + slab PB::W_U4 = 10 KB at 64 couts), so the region phase A leaves behind
            // holds a RING of RING_D slots: RING_D chunks are requested up front, chunk ch + RING_D - 1 as soon as every wave is
            // past chunk ch - 1 (one barrier per chunk), and a wave waits only until ITS pieces of chunk ch have landed
            // (vector-memory operations complete in order: all but the `later * ops` youngest).
            constexpr int SLOT = PB::IN_U4 + PB::W_U4;
            constexpr int RING_D = LDS_U4 / SLOT < 6 ? LDS_U4 / SLOT : 6;
            static_assert(RING_D >= 2, "the shortcut ring needs two slots");
            int ops = 0;  // DMA instructions this wave issues per chunk (wave-uniform)
#pragma unroll
            for (int i = 0; i < PB::NPC; ++i) ops += (wave + 4 * i < PB::NPIECE) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < PB::NWPC; ++i) ops += (wave + 4 * i < PB::NWPIECE) ? 1 : 0;
            auto request = [&](int ch) {
                const unsigned slot = img0 + (unsigned)((ch % RING_D) * SLOT * 16);
                pb.issue_dma(r_rs, r_rs, (unsigned)(ch * 2 * HW) * 16u, slot, wave);
                pb.issue_wdma(wd_rs, wdl_rs, (unsigned)(ch * 2 * p.Nw) * 16u, p.Nw, slot + (unsigned)(PB::IN_U4 * 16), wave);
            };
            for (int ch = 0; ch < RING_D && ch < nB; ++ch) request(ch);
            for (int ch = 0; ch < nB; ++ch) {
                const int issued = min(nB - 1, ch == 0 ? RING_D - 1 : ch + RING_D - 2);  // youngest chunk requested so far
                wait_vmcnt_dyn((issued - ch) * ops);
                lds_barrier();  // (no vmcnt drain) chunk ch is in LDS for every wave, and every wave is past chunk ch - 1
                if (ch >= 1 && ch + RING_D - 1 < nB) request(ch + RING_D - 1);  // into the slot of chunk ch - 1
                const uint4* slot = lds4 + (ch % RING_D) * SLOT;
                PB::compute(slot, slot + PB::IN_U4, acc, lane, wave);
            }
        }
#else   // the round-3 schedule (two buffers, one DMA round trip per chunk): -DLASS_SC_TWO_BUFFERS, for A/B
        pb.issue_dma(r_rs, r_rs, 0u, img0, wave);
        pb.issue_wdma(wd_rs, wdl_rs, 0u, p.Nw, wl0, wave);
        for (int ch = 0; ch < nB; ++ch) {
            const int cur = ch & 1;
            wait_vmcnt<0>();
            __syncthreads();
            if (ch + 1 < nB) {
                pb.issue_dma(r_rs, r_rs, (unsigned)((ch + 1) * 2 * HW) * 16u, img0 + (unsigned)((cur ^ 1) * PB::IN_U4 * 16), wave);
                pb.issue_wdma(wd_rs, wdl_rs, (unsigned)((ch + 1) * 2 * p.Nw) * 16u, p.Nw,
                              wl0 + (unsigned)((cur ^ 1) * PB::W_U4 * 16), wave);
            }
            PB::compute(lds4 + cur * PB::IN_U4, lds4 + 2 * PB::IN_U4 + cur * PB::W_U4, acc, lane, wave);
        }
#endif
        } else {
            uint4* wl_b = lds4 + 2 * PB::IN_U4;
            const unsigned img0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
            const unsigned img1 = img0 + (unsigned)(PB::IN_U4 * 16);
            pb.init_dma(lane, wave, y0, x0, p.H, p.W);
            __syncthreads();  // phase A has finished with the LDS
            pb.issue_dma(r_rs, r_rs, 0u, img0, wave);
            loadB(0);
            pb.store(lds4, wl_b, tid);
            wait_vmcnt<0>();
            __syncthreads();
            for (int ch = 0; ch < nB; ++ch) {
                const bool more = ch + 1 < nB;
                if (more) {
                    pb.issue_dma(r_rs, r_rs, (unsigned)((ch + 1) * 2 * HW) * 16u, (ch & 1) ? img0 : img1, wave);
                    loadB(ch + 1);
                }
                PB::compute(lds4 + ((ch & 1) ? PB::IN_U4 : 0), wl_b, acc, lane, wave);
                __syncthreads();
                if (more) {
                    pb.store(lds4, wl_b, tid);
                    wait_vmcnt<0>();
                }
                __syncthreads();
            }
        }
    } else if (HASB && !IN2BF) {
        uint4* wl_b = lds4 + PB::IN_U4;
        __syncthreads();
        pb.store(lds4, wl_b, tid);
        __syncthreads();
        for (int ch = 0; ch + 1 < nB; ++ch) {
            loadB(ch + 1);
            PB::compute(lds4, wl_b, acc, lane, wave);
            __syncthreads();
            pb.store(lds4, wl_b, tid);
            __syncthreads();
        }
        PB::compute(lds4, wl_b, acc, lane, wave);
    }
#ifdef LASS_CONV_DIAG
    dg_t3 = clock64();
#endif
    if (FLAGS & F_TCONV)
        tconv_store<NCO, NPX, PW>(p, acc, b, n0, y0, x0, lane, wave, lds_tact);
    else
        store_tile<NCO, NPX, PW, FLAGS, RES_PF>(p, acc, rtmp, lds_es, lds_eh, b, n0, y0, x0, lane, wave,
                                                 MASK ? lds_mw : nullptr, OUTBF ? lds_act : nullptr);
#ifdef LASS_CONV_DIAG
    if (p.dbg && tid == 0) {
        long long* d = p.dbg + 4 * (size_t)blockIdx.x;
        const long long te = clock64();
        d[0] = dg_t1 - dg_t0; d[1] = dg_t2 - dg_t1; d[2] = dg_t3 - dg_t2; d[3] = te - dg_t3;
    }
#endif
}

// dst[chunk][tap][octet][Cout][8] (bf16, RNE) = src[co][ci = chunk*16 + octet*8 + j][tap]   (taps = 9 or 1)
// lo != 0: dst = bf16(w - float(bf16(w))), the second term of the hi + lo split
__global__ __launch_bounds__(256) void weights_bf16_kernel(const float* __restrict__ src, int Cout, int Cin, int taps,
                                                           __bf16* __restrict__ dst, int lo, int transposed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)Cout * Cin * taps;
    if (i >= n) return;
    const int jj = (int)(i % 8);
    const int co = (int)((i / 8) % Cout);
    const int o = (int)((i / (8L * Cout)) % 2);
    const int tap = (int)((i / (16L * Cout)) % taps);
    const int chunk = (int)(i / (16L * Cout * taps));
    const int ci = chunk * 16 + o * 8 + jj;
    // transposed (taps == 1): src is [Cin][Cout] - the ConvTranspose2d weight (cin, cout, kh, kw) with n = (cout, kh, kw)
    const float w = transposed ? src[(size_t)ci * Cout + co] : src[((size_t)co * Cin + ci) * taps + tap];
    const __bf16 hi = (__bf16)w;
    dst[i] = lo ? (__bf16)(w - (float)hi) : hi;
}

template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
hipError_t launch_bf16_one(const ConvArgs& p0, hipStream_t stream) {
    ConvArgs p = p0;
#ifdef LASS_CONV_DIAG
    static const int exp_flags = [] { const char* e = getenv("LASS_EXP"); return e ? atoi(e) : 0; }();
    p.exp = exp_flags;
    static long long* dbuf = nullptr;
    static size_t dcap = 0;
    constexpr int PHTd = 4 * NPX * (32 / PW);
    const size_t nblk = (size_t)(p.W / PW) * ((p.H + PHTd - 1) / PHTd) * (p.N / (32 * NCO)) * p.B;
    if (nblk > dcap) {
        if (dbuf) (void)hipFree(dbuf);
        (void)hipMalloc((void**)&dbuf, nblk * 32);
        dcap = nblk;
    }
    p.dbg = dbuf;
    struct Report {
        const ConvArgs& p; size_t nblk; long long* dbuf;
        ~Report() {
            std::vector<long long> h(nblk * 4);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), dbuf, nblk * 32, hipMemcpyDeviceToHost);
            double s[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < nblk; ++i) for (int k = 0; k < 4; ++k) s[k] += (double)h[i * 4 + k];
            fprintf(stderr, "[bf16-diag] taps=%d NCO=%d NPX=%d flags=%d Cin=%d Cin2=%d N=%d %dx%d blocks=%zu | cycles per block: prologue %.0f  "
                    "main %.0f (%.0f per chunk)  shortcut %.0f  epilogue %.0f\n", TAPS, NCO, NPX, FLAGS, p.Cin, p.Cin2, p.N, p.H, p.W, nblk,
                    s[0] / nblk, s[1] / nblk, s[1] / nblk / (p.Cin / 16.0), s[2] / nblk, s[3] / nblk);
        }
    } report{p, nblk, dbuf};
#endif
    constexpr int PHT = 4 * NPX * (32 / PW);
    p.gx = (p.W / PW) * ((p.H + PHT - 1) / PHT);
    p.gy = p.N / (32 * NCO);
    static const int xcd = [] { const char* e = getenv("LASS_XCD_MAP"); return e ? atoi(e) : 2; }();  // 0 off, 1 tile by tile, 2 contiguous ranges
    p.xcd_map = (xcd && ((long)p.gx * p.B) % 8 == 0 && (p.gy > 1 || xcd == 2)) ? xcd : 0;
    dim3 grid((unsigned)((long)p.gx * p.gy * p.B));
    if constexpr ((FLAGS & F_NOSPLIT) != 0) {
        if (p.w_bf16_lo) return hipErrorInvalidValue;
        hipLaunchKernelGGL((conv_bf16_kernel<TAPS, NCO, NPX, PW, FLAGS, 1>), grid, dim3(NTHREADS), 0, stream, p);
    } else {
        if (p.w_bf16_lo)
            hipLaunchKernelGGL((conv_bf16_kernel<TAPS, NCO, NPX, PW, FLAGS, 2>), grid, dim3(NTHREADS), 0, stream, p);
        else
            hipLaunchKernelGGL((conv_bf16_kernel<TAPS, NCO, NPX, PW, FLAGS, 1>), grid, dim3(NTHREADS), 0, stream, p);
    }
    return hipGetLastError();
}

// Tile geometry as in conv.hip: 64-cout tiles at W >= 32, small tiles at the bottom of the U-Net.
template <int TAPS, int FLAGS>
hipError_t launch_bf16(const ConvArgs& p, hipStream_t stream) {
    const int pw = p.W >= 32 ? 32 : p.W;
    constexpr bool NARROW = (FLAGS & (F_MASK | F_RESPRE | F_PRECONV)) != 0;  // 32-cout-only kernels (host-checked N == 32)
    if constexpr (NARROW) {
        if (pw != 32 || p.N != 32) return hipErrorInvalidValue;
        if constexpr ((FLAGS & F_INBF16) != 0) {
            static const int force = [] { const char* e = getenv("LASS_BF16_NPX"); return e ? atoi(e) : 0; }();
            if (force == 4) return launch_bf16_one<TAPS, 1, 4, 32, FLAGS>(p, stream);
        }
        return launch_bf16_one<TAPS, 1, 2, 32, FLAGS>(p, stream);
    } else
    if (pw == 32) {
        if constexpr ((FLAGS & F_INBF16) != 0) {
            // DMA-fed kernels: 16-row tiles (each wave 4 px-tiles x NCO cout-tiles: 0.75 instead of 1 fragment read per
            // MFMA, the weight slab shared by twice the pixels) wherever that still leaves >= 2 workgroups per CU slot
            const long wgs16 = (long)(p.W / 32) * ((p.H + 15) / 16) * (p.N / (p.N % 64 == 0 ? 64 : 32)) * p.B;
            static const int force = [] { const char* e = getenv("LASS_BF16_NPX"); return e ? atoi(e) : 0; }();
            (void)wgs16;
            if (force == 4) {  // measured r2: 16-row tiles lose 15-20 % on every layer (occupancy beats LDS traffic)
                if (p.N % 64 == 0) return launch_bf16_one<TAPS, 2, 4, 32, FLAGS>(p, stream);
                return launch_bf16_one<TAPS, 1, 4, 32, FLAGS>(p, stream);
            }
        }
        if (p.N % 64 == 0) return launch_bf16_one<TAPS, 2, 2, 32, FLAGS>(p, stream);
        return launch_bf16_one<TAPS, 1, 2, 32, FLAGS>(p, stream);
    }
    if constexpr (!NARROW) {
        if (pw == 16) return launch_bf16_one<TAPS, 1, 1, 16, FLAGS>(p, stream);
        if (pw == 8) return launch_bf16_one<TAPS, 1, 2, 8, FLAGS>(p, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool lass_bf16_supported(const ConvArgs& p) {
    return (p.W == 8 || p.W == 16 || (p.W % 32) == 0) && p.Cin % 16 == 0 && p.N % 32 == 0;
}

hipError_t lass_launch_conv_bf16(ConvKind kind, const ConvArgs& p, hipStream_t stream) {
    if (!lass_bf16_supported(p) || !p.w_bf16 || (!p.in && !p.in_bf16) || (!p.out && !p.out_bf16 && !p.mask_re))
        return hipErrorInvalidValue;
    if ((p.in_bf16 || p.out_bf16) && (p.Cin % 8 != 0 || p.N % 8 != 0)) return hipErrorInvalidValue;
    if (p.w_bf16_lo && ((p.in_bf16 && !p.in_bf16_lo) || (p.out_bf16 && !p.out_bf16_lo))) return hipErrorInvalidValue;
    switch (kind) {
        case CONV1_ACT:
            if ((!p.in_bf16 && (!p.pro_scale || !p.pro_shift)) || !p.epi_scale || !p.epi_shift) return hipErrorInvalidValue;
            if (p.in_bf16) {  // decoder conv1 fed by the activated blocked copy of the concat: no prologue left to apply
                if (!p.out_bf16) return hipErrorInvalidValue;
                return launch_bf16<9, F_EPIACT | F_OUTBF16 | F_INBF16 | F_NOSPLIT>(p, stream);
            }
            if (p.out_bf16) return launch_bf16<9, F_PRO | F_EPIACT | F_OUTBF16>(p, stream);
            return launch_bf16<9, F_PRO | F_EPIACT>(p, stream);
        case CONV2_IDENT:
            if (!p.res) return hipErrorInvalidValue;
            if (p.in_bf16) return launch_bf16<9, F_RES | F_INBF16>(p, stream);
            return launch_bf16<9, F_RES>(p, stream);
        case CONV2_SHORTCUT:
            if ((!p.in2 && !p.in2_bf16) || !p.w2_bf16 || !p.bias || p.Cin2 % 16 != 0 || (p.w_bf16_lo && !p.w2_bf16_lo))
                return hipErrorInvalidValue;
            if (p.mask_re) {  // fused output head: decoder_block6 geometry only
                if (p.N != 32 || p.W + 1 != p.mask_nbins || !p.in_bf16 || !p.mask_w || !p.mask_b || !p.mask_mag || !p.mask_cos ||
                    !p.mask_sin || !p.mask_im || p.mask_T <= 0 || p.mask_T > p.H)
                    return hipErrorInvalidValue;
                if (p.in2_bf16) return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_IN2BF16 | F_MASK | F_NOSPLIT>(p, stream);
                return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_MASK>(p, stream);
            }
            if (p.in_bf16 && p.out_bf16 && !p.out_bf16_act) {
                // decoder output handed to the next transposed conv as ONE activated blocked bf16 tensor (its BN+FiLM+leaky
                // prologue applied here as the epilogue activation)
                if (!p.epi_scale || !p.epi_shift) return hipErrorInvalidValue;
                if (p.in2_bf16)
                    return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_IN2BF16 | F_EPIACT | F_OUTBF16 | F_NOSPLIT>(p, stream);
                return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_EPIACT | F_OUTBF16 | F_NOSPLIT>(p, stream);
            }
            if (p.in_bf16 && p.in2_bf16 && p.out_bf16)  // encoder block fed by, and feeding, blocked bf16 copies
                return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_IN2BF16 | F_OUTBF16 | F_NOSPLIT>(p, stream);
            if (p.in_bf16 && p.in2_bf16) return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_IN2BF16 | F_NOSPLIT>(p, stream);
            if (p.in_bf16 && p.out_bf16) return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16 | F_OUTBF16 | F_NOSPLIT>(p, stream);
            if (p.in_bf16) return launch_bf16<9, F_PHASEB | F_BIAS | F_INBF16>(p, stream);
            return launch_bf16<9, F_PHASEB | F_BIAS>(p, stream);
        case TCONV_ACT:
            if ((!p.in_bf16 && (!p.pro_scale || !p.pro_shift)) || (p.up_h != 1 && p.up_h != 2)) return hipErrorInvalidValue;
            if (p.out_bf16 && (p.up_h != 2 || !p.out_bf16_act || !p.act_scale || !p.act_shift || p.out_noct <= 0 || p.N % 32 != 0))
                return hipErrorInvalidValue;
            if (p.in_bf16) return launch_bf16<1, F_TCONV | F_INBF16 | F_NOSPLIT>(p, stream);  // input already activated bf16
            return launch_bf16<1, F_PRO | F_TCONV>(p, stream);
        case CONV1_ACT_PRE:  // encoder_block1 at full resolution: 32 -> 32 channels, W a multiple of 32
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift || !p.pre_w || !p.pre_b || p.N != 32 ||
                p.Cin != 32 || p.W % 32 != 0)
                return hipErrorInvalidValue;
            if (p.out_bf16) return launch_bf16_one<9, 1, 2, 32, F_PRO | F_EPIACT | F_PRECONV | F_OUTBF16>(p, stream);
            return launch_bf16_one<9, 1, 2, 32, F_PRO | F_EPIACT | F_PRECONV>(p, stream);
        case CONV2_IDENT_PRE:
            if (!p.res || !p.pre_w || !p.pre_b || p.N != 32 || p.W % 32 != 0) return hipErrorInvalidValue;
            if (p.in_bf16 && p.out_bf16) return launch_bf16<9, F_RES | F_RESPRE | F_INBF16 | F_OUTBF16 | F_NOSPLIT>(p, stream);
            if (p.in_bf16) return launch_bf16<9, F_RES | F_RESPRE | F_INBF16>(p, stream);
            return launch_bf16_one<9, 1, 2, 32, F_RES | F_RESPRE>(p, stream);
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t lass_launch_weights_bf16(const float* w, int Cout, int Cin, int taps, void* dst, int lo, int transposed,
                                    hipStream_t stream) {
    const long n = (long)Cout * Cin * taps;
    hipLaunchKernelGGL(weights_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin, taps,
                       (__bf16*)dst, lo, transposed);
    return hipGetLastError();
}
