// conv.hip - implicit-GEMM 3x3 / 1x1 / transposed convolutions of the ResUNet30 on gfx950 f32 MFMA.
//
// Replaces (reference: /root/reference/models/resunet.py):
//   ConvBlockRes.forward   :147-165   x1 = conv1(leaky(bn1(x)+b1)); x2 = conv2(leaky(bn2(x1)+b2)); out = sc(x) + x2
//   DecoderBlockRes1B.forward :254-255 x = conv1_T(leaky(bn1(x)+b1))   (kernel == stride, no overlap)
//
// GEMM view (per clip b):  D[n][p] = sum_k A[n][k] * B[k][p]
//   n = output channel (rows of D  -> the 16 accumulator registers of v_mfma_f32_32x32x2_f32)
//   p = output pixel   (cols of D  -> the lane, so NCHW stores are 128-B contiguous per half-wave)
//   k = (input channel, tap)
// A comes from the re-laid-out weights Wt[cin][tap][cout] (cout contiguous), B from a planar LDS halo tile
// [cin][row][col]; both fragments are single conflict-free ds_read_b32 per MFMA operand, register-prefetched one
// k-step ahead.  The f32 MFMA is a bit-exact f32 FMA chain, so results differ from the reference only by summation
// order.
//
// Fusions: BN(eval)+FiLM+leaky-ReLU as a prologue while staging the halo tile (zero padding is applied AFTER the
// activation, as conv padding does; the per-channel scale/shift are wave-uniform scalars because a staging pass
// covers exactly one channel pair), the block's second activation as conv1's epilogue (tables in LDS), the identity
// residual / shortcut bias as the INITIAL accumulator, the 1x1 shortcut conv as a second K-phase into the same
// accumulators, transposed-conv scatter, channel-slice ("virtual concat") output via batch strides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "conv_common.h"
#include "wino_common.h"  // block_coords: the XCD-aware workgroup order

namespace {

// One K-phase: per chunk, [KC] channels x (rows+halo) x (cols+halo) of input and [KC][TAPS][NT] of weights are staged
// in LDS; the registers of chunk c+1 are loaded from global memory while chunk c is contracted.
// Input staging walks the chunk in groups of G channels (G*CH_ELEMS elements, NPASS passes of 256 threads), so the
// channel of an element is (group, u >= CH_ELEMS): wave-uniform up to one select.
template <int TAPS, int KC, int NCO, int NPX, int PW, bool PRO, bool PRE = false>
struct Phase {
    static constexpr int PH = 32 / PW;
    static constexpr int WROWS = NPX * PH;
    static constexpr int PHT = 4 * WROWS;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int IR = PHT + 2 * HALO;
    static constexpr int IP = PW + 2 * HALO;
    static constexpr int NT = 32 * NCO;
    static constexpr int CH_ELEMS = IR * IP;
    static constexpr int G = (TAPS == 9) ? 2 : 1;
    static constexpr int NGRP = KC / G;
    static constexpr int GRP_ELEMS = G * CH_ELEMS;
    static constexpr int NPASS = (GRP_ELEMS + NTHREADS - 1) / NTHREADS;
    static constexpr int IN_ELEMS = KC * CH_ELEMS;
    static constexpr int W_V4 = KC * TAPS * NT / 4;
    static constexpr int NWLD = (W_V4 + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_FLOATS = IN_ELEMS + KC * TAPS * NT;
    static_assert((IN_ELEMS % 4) == 0, "weight region must stay 16-B aligned");
    static_assert(KC % G == 0 && G <= 2, "channel grouping");

    // Every global load below is UNCONDITIONAL (addresses clamped into the image / the weight slab, the padding zero
    // applied by a select when the element is written to LDS): a conditional load makes hipcc branch around it and
    // drain vmcnt at the join, which serialises the prefetch behind a full memory round trip per chunk.
    unsigned goff[NPASS];   // clamped BYTE offset of this thread's element in pass k of group 0 (lane part of a buffer address)
    unsigned woff[NWLD];    // BYTE offset of this thread's weight float4s inside a chunk's slab
    unsigned okbits;        // bit k: the element of pass k lies inside the image (else conv zero padding)
    float v[NGRP][NPASS];   // prefetched input elements
    float4 wv[NWLD];        // prefetched weights
    float psc[KC], psh[KC]; // wave-uniform prologue scale / shift of the prefetched chunk (SGPRs)
    float pcw[KC], pcb[KC]; // PRE: pre_conv weight / bias of the prefetched chunk's channels (resunet.py:555)
    float x0v[NPASS];       // PRE: the single-channel input at this thread's positions (same for every channel)

    __device__ __forceinline__ static int upos(int tid, int k) {  // element index within a channel group (clamped:
        const int u = tid + k * NTHREADS;                        // surplus threads of the last pass duplicate the
        return u < GRP_ELEMS ? u : GRP_ELEMS - 1;                 // group's last element)
    }

    __device__ __forceinline__ void init(int tid, int y0, int x0, int H, int W) {
        okbits = 0;
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int u = upos(tid, k);
            const int cl = (G == 2 && u >= CH_ELEMS) ? 1 : 0;
            const int w = u - cl * CH_ELEMS;
            const int r = w / IP, x = w % IP;
            const int gy = y0 + r - HALO, gx = x0 + x - HALO;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1);
            goff[k] = 4u * (unsigned)(cl * H * W + gyc * W + gxc);
            okbits |= (ok ? 1u : 0u) << k;
        }
    }

    // in_c0: channel c0 of this clip; w_c0: Wt[c0][0][n0]; sc/sh: prologue tables at channel c0 (PRO only)
    // PRE: fetch x0 once per tile (goff's channel-local part must be dropped: there is one plane only)
    __device__ __forceinline__ void load_x0(const float* __restrict__ x0_b, int HW) {
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const unsigned e = goff[k] / 4u;
            x0v[k] = x0_b[e >= (unsigned)HW ? e - HW : e];
        }
    }
    __device__ __forceinline__ void load_pre(const float* __restrict__ pw, const float* __restrict__ pb) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            pcw[c] = pw[c];
            pcb[c] = pb[c];
        }
    }

    __device__ __forceinline__ void init_w(int tid, int Nw) {
#pragma unroll
        for (int i = 0; i < NWLD; ++i) {
            const int e0 = tid + i * NTHREADS;  // float4 index into [KC*TAPS][NT/4]
            const int e = e0 < W_V4 ? e0 : W_V4 - 1;
            const int row = e / (NT / 4), col = e % (NT / 4);
            woff[i] = 4u * (unsigned)(row * Nw + col * 4);
        }
    }
    // Buffer-addressed (descriptor + scalar byte offset + constant lane offset: no VALU address arithmetic, which the f32
    // MFMA would have to share the SIMD's VALU issue with).  in_rs: this clip's planes, c0b: byte offset of channel c0;
    // w_rs: Wt from column n0 on, wb: byte offset of row c0*TAPS.
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t in_rs, unsigned c0b, int HW, __amdgpu_buffer_rsrc_t w_rs,
                                         unsigned wb, const float* __restrict__ sc, const float* __restrict__ sh) {
        if (!PRE) {
#pragma unroll
            for (int q = 0; q < NGRP; ++q)
#pragma unroll
                for (int k = 0; k < NPASS; ++k)
                    v[q][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                            in_rs, (int)goff[k], (int)(c0b + (unsigned)(q * G * HW) * 4u), 0));
        }
#pragma unroll
        for (int i = 0; i < NWLD; ++i)
            wv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)woff[i], (int)wb, 0));
        if (PRO) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                psc[c] = sc[c];
                psh[c] = sh[c];
            }
        }
    }

    __device__ __forceinline__ void store(float* lds, int tid) {
#pragma unroll
        for (int q = 0; q < NGRP; ++q)
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int u = upos(tid, k);
                float t = PRE ? 0.f : v[q][k];
                if (PRE) {
                    const bool hi0 = (G == 2) && (u >= CH_ELEMS);
                    t = x0v[k] * (hi0 ? pcw[q * G + G - 1] : pcw[q * G]) + (hi0 ? pcb[q * G + G - 1] : pcb[q * G]);
                }
                if (PRO) {
                    const bool hi = (G == 2) && (u >= CH_ELEMS);
                    const float s = hi ? psc[q * G + G - 1] : psc[q * G];
                    const float h = hi ? psh[q * G + G - 1] : psh[q * G];
                    t = leaky(t * s + h);
                }
                t = ((okbits >> k) & 1u) ? t : 0.f;  // conv zero padding comes after the activation
                lds[q * GRP_ELEMS + u] = t;
            }
        float4* lw = reinterpret_cast<float4*>(lds + IN_ELEMS);
#pragma unroll
        for (int i = 0; i < NWLD; ++i) {
            const int e0 = tid + i * NTHREADS;
            lw[e0 < W_V4 ? e0 : W_V4 - 1] = wv[i];
        }
    }

    __device__ __forceinline__ static void compute(const float* lds, f32x16 (&acc)[NCO][NPX], int lane, int wave) {
        const int khalf = lane >> 5, j = lane & 31;
        const int ty = j / PW, tx = j % PW;
        const float* bbase = lds + khalf * CH_ELEMS + (wave * WROWS + ty) * IP + tx;
        const float* abase = lds + IN_ELEMS + khalf * (TAPS * NT) + j;
        constexpr int S = (KC / 2) * TAPS;  // k-steps: (channel pair, tap)
        // register double-buffered fragments: the reads of step s+1 are issued before the MFMAs of step s
        float a[2][NCO], b[2][NPX];
        auto rd = [&](int s, float (&aa)[NCO], float (&bb)[NPX]) {
            const int kk = s / TAPS, tap = s % TAPS;
#pragma unroll
            for (int co = 0; co < NCO; ++co) aa[co] = abase[(kk * 2 * TAPS + tap) * NT + co * 32];
#pragma unroll
            for (int px = 0; px < NPX; ++px)
                bb[px] = bbase[kk * 2 * CH_ELEMS + (px * PH + (TAPS == 9 ? tap / 3 : 0)) * IP +
                               (TAPS == 9 ? tap % 3 : 0)];
        };
        rd(0, a[0], b[0]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) rd(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this step's MFMAs (hipcc sinks it otherwise)
#pragma unroll
            for (int co = 0; co < NCO; ++co)
#pragma unroll
                for (int px = 0; px < NPX; ++px)
                    acc[co][px] =
                        __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][co], b[s & 1][px], acc[co][px], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Double-buffered variant: chunk c+1 is written into the other LDS buffer at the top of iteration c (its registers
    // were loaded one iteration earlier), chunk c+2's global loads are issued, then chunk c is contracted: ONE barrier
    // per chunk.  The chunk loop is unrolled by two so both buffer addresses are compile-time constants.
    __device__ __forceinline__ void run_db(float* lds, int buf_stride, const float* __restrict__ in_b, int Cin, int HW,
                                           const float* __restrict__ Wt, int Nw, int n0,
                                           const float* __restrict__ sc, const float* __restrict__ sh,
                                           f32x16 (&acc)[NCO][NPX], int tid, int y0, int x0, int H, int W) {
        const int lane = tid & 63, wave = tid >> 6;
        float* buf0 = lds;
        float* buf1 = lds + buf_stride;
        init(tid, y0, x0, H, W);
        init_w(tid, Nw);
        const int nchunks = Cin / KC;  // even (host-checked)
        const __amdgpu_buffer_rsrc_t in_rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, Cin * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t w_rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt + n0), 0, (Cin * TAPS * Nw - n0) * 4, 0x00020000);
        auto ld = [&](int c) {
            load(in_rs, (unsigned)(c * KC * HW) * 4u, HW, w_rs, (unsigned)(c * KC * TAPS * Nw) * 4u, sc + c * KC, sh + c * KC);
        };
        ld(0);
        __syncthreads();  // previous phase's LDS reads complete; epilogue tables visible
        store(buf0, tid);
        ld(1);
        __syncthreads();
        for (int ch = 0; ch < nchunks; ch += 2) {
            store(buf1, tid);  // chunk ch+1 (always exists)
            if (ch + 2 < nchunks) ld(ch + 2);
            compute(buf0, acc, lane, wave);
            __syncthreads();
            if (ch + 2 < nchunks) {
                store(buf0, tid);  // chunk ch+2
                if (ch + 3 < nchunks) ld(ch + 3);
            }
            compute(buf1, acc, lane, wave);
            __syncthreads();
        }
    }
};

template <int A, int B>
struct MaxI {
    static constexpr int v = A > B ? A : B;
};


// ---- single-tile kernel, double-buffered LDS (one barrier per chunk) --------------------------------------------------
template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
__global__ __launch_bounds__(NTHREADS) void conv_kernel_db(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr int KCA = (TAPS == 9) ? 8 : 16;
    using PA = Phase<TAPS, KCA, NCO, NPX, PW, PRO>;
    using PB = Phase<1, 16, NCO, NPX, PW, false>;
    constexpr int LDS_ONE = (FLAGS & F_PHASEB) ? MaxI<PA::LDS_FLOATS, PB::LDS_FLOATS>::v : PA::LDS_FLOATS;
    constexpr int LDS_MAIN = 2 * LDS_ONE;
    constexpr int PH = PA::PH, WROWS = PA::WROWS, PHT = PA::PHT, NT = PA::NT;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;

    static_assert(LDS_MAIN % 4 == 0, "epilogue tables are read as float4");
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN + (EPI ? 2 * NT : 0)];
    float* lds_es = lds + LDS_MAIN;  // epilogue scale / shift for this block's NT output channels
    float* lds_eh = lds_es + NT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int bx_, by_, b;
    block_coords(p, bx_, by_, b);  // 1-D grid: the cout blocks of one input tile on one XCD (wino_common.h)
    const int n0 = by_ * NT;
    const int tiles_x = p.W / PW;
    const int y0 = (bx_ / tiles_x) * PHT;
    const int x0 = (bx_ % tiles_x) * PW;
    const int HW = p.H * p.W;
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;

    if (EPI) {
        if (tid < NT) {
            lds_es[tid] = p.epi_scale[n0 + tid];
            lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + n0 + tid];
        }
    }

    // ---- accumulator initialisation: 0, the shortcut bias, or the identity residual ---------------------------
    f32x16 acc[NCO][NPX];
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = y0 + wave * WROWS + px * PH + ty;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                float v = 0.f;
                if (FLAGS & F_BIAS) v = p.bias[n];
                if (FLAGS & F_RES)  // unconditional (row clamped; rows >= H are never stored)
                    v = p.res[(size_t)b * p.res_bs + (size_t)n * HW + (size_t)min(y, p.H - 1) * p.W + x];
                acc[co][px][r] = v;
            }
        }

    const float* sc = PRO ? p.pro_scale : nullptr;
    const float* sh = PRO ? p.pro_shift + (size_t)b * p.pro_shift_bs : nullptr;
    {
        PA ph;
        ph.run_db(lds, LDS_ONE, p.in + (size_t)b * p.in_bs, p.Cin, HW, p.w, p.Nw, n0, sc, sh, acc, tid, y0, x0, p.H,
                  p.W);
    }
    if (FLAGS & F_PHASEB) {
        PB ph;
        ph.run_db(lds, LDS_ONE, p.in2 + (size_t)b * p.in2_bs, p.Cin2, HW, p.w2, p.Nw, n0, nullptr, nullptr, acc, tid,
                  y0, x0, p.H, p.W);
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    if (FLAGS & F_TCONV) {
        tconv_store<NCO, NPX, PW>(p, acc, b, n0, y0, x0, lane, wave);
    } else {
        store_tile<NCO, NPX, PW, (FLAGS & ~F_RES), false>(p, acc, nullptr, lds_es, lds_eh, b, n0, y0, x0, lane, wave);
    }
}

// ---- single-tile kernel, single-buffered LDS (two barriers per chunk, highest occupancy) ------------------------------
// Used for the 32-wide output tiles, the few-chunk layers and the 1-tap kernels.  Everything that has to come from HBM
// before the next contraction can start is requested one contraction earlier: chunk c+1 during chunk c, the shortcut
// phase's first chunk and the identity residual during the main phase's last chunk.
// (A multi-tile variant with cross-tile prefetch was measured and brought nothing: on these layers the matrix pipe is
// already ~90 % busy and the chip holds only ~1.9 GHz under their HBM + LDS load - see DESIGN.md.)
template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
__global__ __launch_bounds__(NTHREADS) void conv_kernel_sb(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr bool HASB = (FLAGS & F_PHASEB) != 0;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;
    constexpr bool BIAS = (FLAGS & F_BIAS) != 0;
    constexpr bool RES = (FLAGS & F_RES) != 0;
    constexpr bool RES_PF = RES && NCO == 1;  // residual prefetched into registers during the last chunk
    constexpr bool PRE = (FLAGS & F_PRECONV) != 0;
    constexpr bool RESPRE = (FLAGS & F_RESPRE) != 0;
    static_assert(!RESPRE || RES_PF, "x0-derived residual needs the register-prefetch path");
    constexpr int KCA = (TAPS == 9) ? 8 : 16;
    using PA = Phase<TAPS, KCA, NCO, NPX, PW, PRO, PRE>;
    using PB = Phase<1, 16, NCO, NPX, PW, false>;
    constexpr int LDS_ONE = HASB ? MaxI<PA::LDS_FLOATS, PB::LDS_FLOATS>::v : PA::LDS_FLOATS;
    constexpr int PH = PA::PH, WROWS = PA::WROWS, PHT = PA::PHT, NT = PA::NT;
    constexpr bool MASK = (FLAGS & F_MASK) != 0;
    constexpr int NTAB = (EPI ? 2 * NT : 0) + (BIAS ? NT : 0) + (RESPRE ? 2 * NT : 0) + (MASK ? 100 : 0);

    static_assert(LDS_ONE % 4 == 0, "epilogue tables are read as float4");
    __shared__ __attribute__((aligned(16))) float lds[LDS_ONE + NTAB];
    float* lds_mw = lds + LDS_ONE + NTAB - 100;  // MASK: after_conv weight [3][32] + bias [3]
    float* lds_es = lds + LDS_ONE;  // epilogue scale / shift for this block's NT output channels
    float* lds_eh = lds_es + NT;
    float* lds_bias = lds + LDS_ONE + (EPI ? 2 * NT : 0);
    float* lds_rw = lds + LDS_ONE + (EPI ? 2 * NT : 0) + (BIAS ? NT : 0);  // RESPRE: pre_conv weight / bias of the
    float* lds_rb = lds_rw + NT;                                          // block's output channels

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int bx_, by_, b;
    block_coords(p, bx_, by_, b);  // 1-D grid: the cout blocks of one input tile on one XCD (wino_common.h)
    const int n0 = by_ * NT;
    const int tiles_x = p.W / PW;
    const int y0 = (bx_ / tiles_x) * PHT, x0 = (bx_ % tiles_x) * PW;
    const int HW = p.H * p.W;
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;
    const int nA = p.Cin / KCA;
    const int nB = HASB ? p.Cin2 / 16 : 0;
    const float* in_b = p.in + (size_t)b * p.in_bs;
    const float* in2_b = HASB ? p.in2 + (size_t)b * p.in2_bs : nullptr;
    const float* sc = PRO ? p.pro_scale : nullptr;
    const float* sh = PRO ? p.pro_shift + (size_t)b * p.pro_shift_bs : nullptr;
#ifdef LASS_CONV_DIAG
    const long long k_c0 = clock64(), k_r0 = wall_clock64();
    long long dsum[4] = {0, 0, 0, 0};
#endif

    if (EPI && tid < NT) {
        lds_es[tid] = p.epi_scale[n0 + tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + n0 + tid];
    }
    if (BIAS && tid < NT) lds_bias[tid] = p.bias[n0 + tid];
    if (RESPRE && tid < NT) {
        lds_rw[tid] = p.pre_w[n0 + tid];
        lds_rb[tid] = p.pre_b[n0 + tid];
    }
    if (MASK && tid < 99) lds_mw[tid] = tid < 96 ? p.mask_w[tid] : p.mask_b[tid - 96];

    PA pa;
    PB pb;
    const auto rs = [](const float* ptr, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t in_rs = rs(in_b, (long)(PRE ? 1 : p.Cin) * HW * 4);
    const __amdgpu_buffer_rsrc_t wa_rs = rs(p.w + n0, ((long)p.Cin * TAPS * p.Nw - n0) * 4);
    const __amdgpu_buffer_rsrc_t in2_rs = HASB ? rs(in2_b, (long)p.Cin2 * HW * 4) : in_rs;
    const __amdgpu_buffer_rsrc_t wb_rs = HASB ? rs(p.w2 + n0, ((long)p.Cin2 * p.Nw - n0) * 4) : wa_rs;
    auto loadA = [&](int c) {
        pa.load(in_rs, (unsigned)(c * KCA * HW) * 4u, HW, wa_rs, (unsigned)(c * KCA * TAPS * p.Nw) * 4u, sc + c * KCA,
                sh + c * KCA);
        if (PRE) pa.load_pre(p.pre_w + c * KCA, p.pre_b + c * KCA);
    };
    auto loadB = [&](int c) {
        pb.load(in2_rs, (unsigned)(c * 16 * HW) * 4u, HW, wb_rs, (unsigned)(c * 16 * p.Nw) * 4u, nullptr, nullptr);
    };

    pa.init(tid, y0, x0, p.H, p.W);
    pa.init_w(tid, p.Nw);
    if (PRE) pa.load_x0(in_b, HW);
    loadA(0);
    __syncthreads();  // tables visible
    pa.store(lds, tid);
    __syncthreads();
#ifdef LASS_CONV_DIAG
    const long long k_c1 = clock64();  // end of the block prologue
#endif

    f32x16 acc[NCO][NPX];
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int px = 0; px < NPX; ++px)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[co][px][r] = BIAS ? lds_bias[co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf] : 0.f;

    // ---- main phase, all chunks but the last
    for (int ch = 0; ch + 1 < nA; ++ch) {
#ifdef LASS_CONV_DIAG
        const long long t0 = clock64();
#endif
        loadA(ch + 1);
        PA::compute(lds, acc, lane, wave);
#ifdef LASS_CONV_DIAG
        const long long t1 = clock64();
#endif
        __syncthreads();
#ifdef LASS_CONV_DIAG
        const long long t2 = clock64();
#endif
        pa.store(lds, tid);
#ifdef LASS_CONV_DIAG
        const long long t3 = clock64();
#endif
        __syncthreads();
#ifdef LASS_CONV_DIAG
        dsum[0] += t1 - t0; dsum[1] += t2 - t1; dsum[2] += t3 - t2; dsum[3] += clock64() - t3;
#endif
    }
    // ---- last chunk of the main phase: prefetch what the shortcut phase / the epilogue need
    float rtmp[RES_PF ? NPX : 1][16];
    if (HASB) {
        pb.init(tid, y0, x0, p.H, p.W);
        pb.init_w(tid, p.Nw);
        loadB(0);
    }
    if (RES_PF) {
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = min(y0 + wave * WROWS + px * PH + ty, p.H - 1);
            if (RESPRE) {  // residual = pre_conv(x0) at this pixel: one load, 16 FMAs (resunet.py:555,165)
                const float xv = p.res[(size_t)b * p.res_bs + (size_t)y * p.W + x];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = 4 * khalf + (r & 3) + 8 * (r >> 2);
                    rtmp[px][r] = xv * lds_rw[nl] + lds_rb[nl];
                }
                continue;
            }
            const float* src = p.res + (size_t)b * p.res_bs + (size_t)(n0 + 4 * khalf) * HW + (size_t)y * p.W + x;
#pragma unroll
            for (int r = 0; r < 16; ++r) rtmp[px][r] = src[(size_t)((r & 3) + 8 * (r >> 2)) * HW];
        }
    }
    PA::compute(lds, acc, lane, wave);
    // ---- shortcut phase (1x1 over the raw block input)
    if (HASB) {
        __syncthreads();
        pb.store(lds, tid);
        __syncthreads();
        for (int ch = 0; ch + 1 < nB; ++ch) {
            loadB(ch + 1);
            PB::compute(lds, acc, lane, wave);
            __syncthreads();
            pb.store(lds, tid);
            __syncthreads();
        }
        PB::compute(lds, acc, lane, wave);
    }
#ifdef LASS_CONV_DIAG
    const long long k_c2 = clock64();
#endif
    // ---- epilogue
    if (FLAGS & F_TCONV)
        tconv_store<NCO, NPX, PW>(p, acc, b, n0, y0, x0, lane, wave);
    else
        store_tile<NCO, NPX, PW, FLAGS, RES_PF>(p, acc, rtmp, lds_es, lds_eh, b, n0, y0, x0, lane, wave,
                                                 MASK ? lds_mw : nullptr);
#ifdef LASS_CONV_DIAG
    if (p.dbg && tid == 0) {
        const long long k_c3 = clock64(), k_r3 = wall_clock64();
        long long* d = p.dbg + 8 * (size_t)blockIdx.x;
        d[0] = dsum[0]; d[1] = dsum[1]; d[2] = dsum[2]; d[3] = dsum[3];
        d[4] = k_c1 - k_c0;   // prologue
        d[5] = k_c3 - k_c0;   // whole block, shader cycles
        d[6] = k_r3 - k_r0;   // whole block, 100 MHz ticks
        d[7] = k_r0;
        (void)k_c2;
    }
#endif
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// -1 = automatic (double-buffered LDS for the 64-wide 3x3 tiles, where it measured 1-9 % faster; single-buffered for
// the 32-wide tiles and the 1-tap kernels, which prefer the higher occupancy); 0 / 1 force a variant (A/B runs).
int conv_variant() {
    static int v = [] {
        const char* e = getenv("LASS_CONV_VARIANT");
        return e ? atoi(e) : -1;
    }();
    return v;
}

#ifdef LASS_CONV_DIAG
void diag_report(long long* dbuf, size_t nblk, const ConvArgs& p, int taps) {
    std::vector<long long> h(nblk * 8);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dbuf, nblk * 64, hipMemcpyDeviceToHost);
    double s[7] = {0};
    long long rmin = h[7], rmax = 0;
    for (size_t i = 0; i < nblk; ++i) {
        for (int k = 0; k < 7; ++k) s[k] += (double)h[i * 8 + k];
        if (h[i * 8 + 7] < rmin) rmin = h[i * 8 + 7];
        if (h[i * 8 + 7] + h[i * 8 + 6] > rmax) rmax = h[i * 8 + 7] + h[i * 8 + 6];
    }
    for (double& v : s) v /= (double)nblk;
    const double clk_ghz = s[5] / s[6] * 0.1;
    fprintf(stderr,
            "[diag] taps=%d Cin=%d N=%d %dx%d blocks=%zu | per-block cycles: compute %.0f  bar1 %.0f  store %.0f  bar2 "
            "%.0f | Kphase %.0f total %.0f | clock %.3f GHz | kernel span %.3f ms\n",
            taps, p.Cin, p.N, p.H, p.W, nblk, s[0], s[1], s[2], s[3], s[4], s[5], clk_ghz, (rmax - rmin) * 1e-5);
}
#endif

template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
hipError_t launch_one(const ConvArgs& p0, hipStream_t stream) {
    constexpr int PHT = 4 * NPX * (32 / PW);
    ConvArgs p = p0;
    // 1-D grid decoded by block_coords(): the gy cout blocks of a (tile, clip) pair run on ONE XCD - a transposed conv is a
    // 1-tap conv with 4 x Cout output channels, i.e. 2-12 cout blocks that all read the same input tile (in the natural
    // 3-D order they ran a whole grid row apart and the tile came from HBM once per block: 2x the input at decoder_block6)
    static const int xcd = [] { const char* e = getenv("LASS_XCD_MAP"); return e ? atoi(e) : 2; }();
    p.gx = (p.W / PW) * ((p.H + PHT - 1) / PHT);
    p.gy = p.N / (32 * NCO);
    p.xcd_map = (xcd && ((long)p.gx * p.B) % 8 == 0 && (p.gy > 1 || xcd == 2)) ? xcd : 0;
    dim3 grid((unsigned)((long)p.gx * p.gy * p.B));
#ifdef LASS_CONV_DIAG
    static long long* dbuf = nullptr;
    static size_t dcap = 0;
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    if (nblk > dcap) {
        if (dbuf) (void)hipFree(dbuf);
        (void)hipMalloc((void**)&dbuf, nblk * 64);
        dcap = nblk;
    }
    p.dbg = dbuf;
    struct Rep {
        long long* d;
        size_t n;
        const ConvArgs& p;
        ~Rep() { diag_report(d, n, p, TAPS); }
    } rep{dbuf, nblk, p};
#endif
    const int var = conv_variant();
    const int kc = TAPS == 9 ? 8 : 16;
    const int nchunks = p.Cin / kc;
    // double-buffered kernel (one barrier per chunk) for the 64-wide tiles with many chunks, where it measured 1-9 %
    // faster; the single-buffered kernel (higher occupancy) everywhere else
    const bool db = var < 0 ? (TAPS == 9 && NCO == 2 && PW == 32 && nchunks >= 16 && !(FLAGS & F_RES))
                            : ((var & 1) != 0 && !(FLAGS & F_RES));
    constexpr bool HAS_DB = TAPS == 9 && NCO == 2 && PW == 32 && !(FLAGS & F_RES);  // geometries the db kernel exists for
    if constexpr (HAS_DB) {
        if (db) {
            hipLaunchKernelGGL((conv_kernel_db<TAPS, NCO, NPX, PW, FLAGS>), grid, dim3(NTHREADS), 0, stream, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((conv_kernel_sb<TAPS, NCO, NPX, PW, FLAGS>), grid, dim3(NTHREADS), 0, stream, p);
    return hipGetLastError();
}

// Tile geometry per layer shape.  Wave tile = NCO x NPX MFMA tiles of 32 couts x 32 pixels; block = 4 waves stacked
// over rows.  The bottom of the U-Net (W <= 16, 384 channels, a few thousand pixels per batch) needs small tiles to
// produce enough workgroups for 256 CUs.
template <int TAPS, int FLAGS>
hipError_t launch_geom(const ConvArgs& p, hipStream_t stream) {
    static const int small_env = env_int("LASS_SMALL", -1);
    // measured (B=16): 3x3 at W=16 -> 32-cout x 32-px wave tiles in 8-row blocks; 3x3 at W=8 -> 32-cout tiles;
    // the 1-tap transposed convs keep the 64-cout tiles
    const int small = small_env >= 0 ? small_env : (TAPS == 9 ? (p.W == 16 ? 3 : 2) : 0);
    const int pw = p.W >= 32 ? 32 : p.W;
    if (pw == 32) {
        if (p.N % 64 == 0) return launch_one<TAPS, 2, 2, 32, FLAGS>(p, stream);
        return launch_one<TAPS, 1, 2, 32, FLAGS>(p, stream);
    }
    if (p.N % 64 != 0) return hipErrorInvalidValue;
    if (pw == 16) {
        if (small == 3) return launch_one<TAPS, 1, 1, 16, FLAGS>(p, stream);
        return launch_one<TAPS, 2, 2, 16, FLAGS>(p, stream);
    }
    if (pw == 8) {
        if (small == 2) return launch_one<TAPS, 1, 2, 8, FLAGS>(p, stream);
        return launch_one<TAPS, 2, 2, 8, FLAGS>(p, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// Host-side shape validation shared by all conv launches: every assumption the kernel and its grid make.
static bool conv_args_ok(const ConvArgs& p, int taps, bool phaseb) {
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return false;
    if (p.W != 8 && p.W != 16 && (p.W % 32) != 0) return false;
    if (p.N % 32 != 0 || p.Nw < p.N || (p.Nw % 4) != 0) return false;
    const int kc = taps == 9 ? 8 : 16;
    if (p.Cin <= 0 || p.Cin % (2 * kc) != 0) return false;  // even chunk count (run_db)
    if (phaseb && (p.Cin2 <= 0 || p.Cin2 % 32 != 0)) return false;
    if (!p.in || !p.w || (!p.out && !p.mask_re)) return false;
    return true;
}

hipError_t lass_launch_conv(ConvKind kind, const ConvArgs& p, hipStream_t stream) {
    switch (kind) {
        case CONV1_ACT:  // 3x3, prologue act on x, epilogue act (conv2's BN+FiLM+leaky)
            if (!conv_args_ok(p, 9, false) || !p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift)
                return hipErrorInvalidValue;
            return launch_geom<9, F_PRO | F_EPIACT>(p, stream);
        case CONV2_IDENT:  // 3x3 over pre-activated input, + residual
            if (!conv_args_ok(p, 9, false) || !p.res) return hipErrorInvalidValue;
            return launch_geom<9, F_RES>(p, stream);
        case CONV2_SHORTCUT:  // 3x3 over pre-activated input, + 1x1(in2) + bias
            if (!conv_args_ok(p, 9, true) || !p.in2 || !p.w2 || !p.bias) return hipErrorInvalidValue;
            if (p.mask_re) {  // fused output head: decoder_block6 geometry only
                if (p.N != 32 || p.W + 1 != p.mask_nbins || !p.mask_w || !p.mask_b || !p.mask_mag || !p.mask_cos || !p.mask_sin ||
                    !p.mask_im || p.mask_T <= 0 || p.mask_T > p.H)
                    return hipErrorInvalidValue;
                return launch_one<9, 1, 2, 32, F_PHASEB | F_BIAS | F_MASK>(p, stream);
            }
            return launch_geom<9, F_PHASEB | F_BIAS>(p, stream);
        case CONV1_ACT_PRE:  // encoder_block1.conv1 reading x0 directly (pre_conv fused into the staging)
            if (!conv_args_ok(p, 9, false) || !p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift ||
                !p.pre_w || !p.pre_b || p.N != 32 || p.W < 32)
                return hipErrorInvalidValue;
            return launch_one<9, 1, 2, 32, F_PRO | F_EPIACT | F_PRECONV>(p, stream);
        case CONV2_IDENT_PRE:  // encoder_block1.conv2 with the residual recomputed from x0
            if (!conv_args_ok(p, 9, false) || !p.res || !p.pre_w || !p.pre_b || p.N != 32 || p.W < 32)
                return hipErrorInvalidValue;
            return launch_one<9, 1, 2, 32, F_RES | F_RESPRE>(p, stream);
        case TCONV_ACT:  // kernel==stride transposed conv with prologue act
            if (!conv_args_ok(p, 1, false) || !p.pro_scale || !p.pro_shift || (p.up_h != 1 && p.up_h != 2))
                return hipErrorInvalidValue;
            return launch_geom<1, F_PRO | F_TCONV>(p, stream);
    }
    return hipErrorInvalidValue;
}
