// conv_common.h - pieces shared by the f32 (conv.hip) and bf16 (conv_bf16.hip) implicit-GEMM kernels: kernel flags and the
// epilogue stores for the 32x32 MFMA accumulator layout (identical for every 32x32 MFMA dtype on gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "pixel_ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int F_PRO = 1;     // prologue: x*scale[c] + shift[b][c], leaky 0.01
constexpr int F_PHASEB = 2;  // second K-phase: 1x1 over raw in2 (shortcut conv)
constexpr int F_BIAS = 4;    // + bias[n]            (as initial accumulator)
constexpr int F_RES = 8;     // + res[b][n][y][x]    (as initial accumulator)
constexpr int F_EPIACT = 16; // epilogue: leaky(v*scale[n] + shift[b][n])
constexpr int F_TCONV = 32;  // n = (co, a, bb); scatter to (y*uh+a, x*uw+bb)
constexpr int F_PRECONV = 64;   // input is the 1-channel x0; channel c = pre_w[c]*x0 + pre_b[c] is formed while staging
constexpr int F_RESPRE = 128;   // with F_RES: the residual is pre_w[n]*x0 + pre_b[n] (never materialised)
constexpr int F_OUTBF16 = 256;  // bf16 kernels: the output is the blocked bf16 intermediate [C/8][H][W][8] (+ lo plane)
constexpr int F_MASK = 1024;    // epilogue = after_conv + complex ratio mask (see ConvArgs::mask_*); no tensor output
constexpr int F_IN2BF16 = 4096;  // bf16 kernels: phase B (shortcut) reads the blocked bf16 raw copy by LDS-DMA
constexpr int F_NOSPLIT = 8192;  // bf16 mode only (no split-operand instantiation)
constexpr int F_INBF16 = 512;   // bf16 kernels: phase A reads that intermediate by LDS-DMA

constexpr int NTHREADS = 256;

// The khalf pair of a pixel (lanes j, j + 32) holds the two 8-byte halves of every 16-byte unit of the blocked bf16 layout.
// For two units A, B of that pixel (adjacent octets) one v_permlane32_swap per dword (lanes 32-63 of its first operand swap
// with lanes 0-31 of the second) leaves lanes 0-31 with the whole unit A and lanes 32-63 with the whole unit B: ONE 16-byte
// store per lane instead of two 8-byte ones (an epilogue of 8-byte stores is store-issue-bound, not byte-bound).
__device__ __forceinline__ uint4 pair_units(uint2 a, uint2 b) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    const unsigned a0 = s0[0], b0 = s0[1], a1 = s1[0], b1 = s1[1];  // (named: hipcc's bit_cast / use of a vector ELEMENT is fragile)
    return uint4{a0, a1, b0, b1};
}

// Transposed-conv scatter: n = co_real*(uh*2) + a*2 + bb; registers (r, r+1), r even, are bb = 0/1 of one (co_real, a),
// so each lane writes 8 contiguous bytes and a half-wave a contiguous 256-B run of the up-sampled row.
template <int NCO, int NPX, int PW>
__device__ __forceinline__ void tconv_store(const ConvArgs& p, f32x16 (&acc)[NCO][NPX], int b, int n0, int y0, int x0,
                                            int lane, int wave, const float* lds_tact = nullptr) {  // [2][8*NCO]
    constexpr int PH = 32 / PW, WROWS = NPX * PH;
    const int HW = p.H * p.W;
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;
    const int uhw = p.up_h * 2;
    const size_t oHW = (size_t)HW * uhw;
    const int oW = p.W * 2;
    if (p.out_bf16) {
        // Blocked bf16 outputs (up_h == 2, host-checked): a 32-row co-tile is 8 channels x 4 sub-pixels = one octet; the
        // khalf pair holds all of it (channel 2g + khalf, sub-pixel i in register 4g + i).  The pair swaps halves so that
        // lane khalf owns output row a = khalf with all 8 channels, then stores two adjacent 16-B units (bb = 0, 1).
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        const size_t clip = (size_t)b * p.out_noct * oHW;
#pragma unroll
        for (int co = 0; co < NCO; ++co)
#pragma unroll
            for (int px = 0; px < NPX; ++px) {
                const int y = y0 + wave * WROWS + px * PH + ty;
                float ch[2][8];  // [bb][channel in octet]
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        // one v_permlane32_swap: lanes 32-63 of the a = 0 register (odd channel) swap with lanes 0-31 of the
                        // a = 1 register (even channel): first result = even channel, second = odd channel, row a = khalf
                        const float lo_a = acc[co][px][4 * g + bb], hi_a = acc[co][px][4 * g + 2 + bb];  // a = 0 / 1
                        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo_a), __builtin_bit_cast(unsigned, hi_a), false, false);
                        const unsigned r0 = sw[0], r1 = sw[1];
                        ch[bb][2 * g] = __builtin_bit_cast(float, r0);
                        ch[bb][2 * g + 1] = __builtin_bit_cast(float, r1);
                    }
                if (y >= p.H) continue;
                const int oct = (n0 + co * 32) / 32;  // octet of this co-tile among the launch's output channels
                const size_t unit = clip + (size_t)(p.out_oct0 + oct) * oHW + (size_t)(y * 2 + khalf) * oW + x * 2;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    bf16x8 raw, act;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        raw[k] = (__bf16)ch[bb][k];
                        act[k] = (__bf16)leaky(ch[bb][k] * lds_tact[co * 8 + k] + lds_tact[8 * NCO + co * 8 + k]);
                    }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(p.out_bf16) + (unit + bb) * 16) = raw;
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(p.out_bf16_act) + (unit + bb) * 16) = act;
                }
            }
        return;
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = y0 + wave * WROWS + px * PH + ty;
            if (y >= p.H) continue;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int n = n0 + co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int co_real = n / uhw, a = (n % uhw) >> 1;
                float2 o = make_float2(acc[co][px][r], acc[co][px][r + 1]);
                float* dst = p.out + (size_t)b * p.out_bs + co_real * oHW + (size_t)(y * p.up_h + a) * oW + x * 2;
                *reinterpret_cast<float2*>(dst) = o;
            }
        }
}

// Final store of one wave's accumulators (non-transposed kernels): optional prefetched residual, optional epilogue
// activation, and optionally the block's F.avg_pool2d (resunet.py:197) fused in: the 2x2 (or 1x2) window of a pooled
// pixel is (px-tile 0, px-tile 1) x (lane, lane^1), summed in the reference's row-major order.
template <int NCO, int NPX, int PW, int FLAGS, bool HAVE_RTMP>
__device__ __forceinline__ void store_tile(const ConvArgs& p, f32x16 (&acc)[NCO][NPX], const float (*rtmp)[16],
                                           const float* lds_es, const float* lds_eh, int b, int n0, int y0, int x0,
                                           int lane, int wave, const float* lds_mw = nullptr,
                                           const float* lds_act = nullptr) {  // [4][32*NCO]: skip act scale/shift, pool act scale/shift
    static_assert((FLAGS & F_MASK) == 0 || NCO == 1, "the fused output head needs all 32 channels in one wave");
    constexpr int PH = 32 / PW, WROWS = NPX * PH;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;
    constexpr bool RES = (FLAGS & F_RES) != 0;
    const int HW = p.H * p.W;
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
        float val[NPX][16];
        // This lane's 16 channels of the tile are 4 runs of 4 consecutive channels (run g at co*32 + 8g + 4*khalf): their
        // table entries are fetched as float4 per run, once per cout tile - 2-6 x 4 ds_read_b128 instead of up to 6 x 64
        // dependent ds_read_b32 inside the pixel loops (measured: 6 500-17 000 cycles of epilogue per workgroup before).
        float es4[4][4], eh4[4][4], as4[4][4], ah4[4][4], ps4[4][4], ph4[4][4];
        auto ld4 = [&](const float* tab, float (&dst)[4][4]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(tab + co * 32 + 8 * g + 4 * khalf);
                dst[g][0] = t.x; dst[g][1] = t.y; dst[g][2] = t.z; dst[g][3] = t.w;
            }
        };
        if (EPI) { ld4(lds_es, es4); ld4(lds_eh, eh4); }
        if ((FLAGS & F_OUTBF16) != 0 && p.out_bf16_act) { ld4(lds_act, as4); ld4(lds_act + 32 * NCO, ah4); }
        if ((FLAGS & F_OUTBF16) != 0 && p.pool_bf16) { ld4(lds_act + 2 * 32 * NCO, ps4); ld4(lds_act + 3 * 32 * NCO, ph4); }
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = y0 + wave * WROWS + px * PH + ty;
            const size_t pix = (size_t)(n0 + co * 32 + 4 * khalf) * HW + (size_t)min(y, p.H - 1) * p.W + x;
            float radd[16];
            if (RES && !HAVE_RTMP) {  // batch of 16 unconditional loads, then one wait
                const float* src = p.res + (size_t)b * p.res_bs + pix;
#pragma unroll
                for (int r = 0; r < 16; ++r) radd[r] = src[(size_t)((r & 3) + 8 * (r >> 2)) * HW];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2);  // + co*32 + 4*khalf
                float v = acc[co][px][r];
                if (RES && HAVE_RTMP) v += rtmp[px][r];
                if (RES && !HAVE_RTMP) v += radd[r];
                if (EPI) v = leaky(v * es4[r >> 2][r & 3] + eh4[r >> 2][r & 3]);
                val[px][r] = v;
            }
            if ((FLAGS & F_MASK) != 0) {
                // NCO == 1, N == 32: this lane and its khalf partner hold all 32 channels of the pixel
                float l[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += lds_mw[q * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf] * val[px][r];
                    s += __shfl_xor(s, 32, 64);
                    l[q] = s + lds_mw[96 + q];
                }
                if (khalf == 0 && y < p.mask_T) mask_pixel(p, b, y, x, l[0], l[1], l[2]);
            } else if ((FLAGS & F_OUTBF16) != 0) {
                // blocked bf16 layout: unit (octet, y, x) = 16 B = 8 channels; this lane holds channels 4*khalf..+3 of the
                // four octets g of its 32-cout tile (8 bytes each; the khalf partner holds the other 8): per octet PAIR the
                // halves are exchanged (pair_units) and lane khalf stores the whole unit of octet g + khalf
                {
                    const int noct = p.out_noct ? p.out_noct : p.N / 8;
                    const size_t clip = (size_t)b * noct * HW;
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        uint2 hi2[2], lo2[2], ac2[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            bf16x4 hi, lo, ac;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = val[px][4 * (g + h) + i];
                                hi[i] = (__bf16)v;
                                lo[i] = (__bf16)(v - (float)hi[i]);
                                ac[i] = p.out_bf16_act ? (__bf16)leaky(v * as4[g + h][i] + ah4[g + h][i]) : (__bf16)0.f;
                            }
                            hi2[h] = __builtin_bit_cast(uint2, hi);
                            lo2[h] = __builtin_bit_cast(uint2, lo);
                            ac2[h] = __builtin_bit_cast(uint2, ac);
                        }
                        const size_t unit = clip + (size_t)(p.out_oct0 + (n0 + co * 32) / 8 + g + khalf) * HW + (size_t)min(y, p.H - 1) * p.W + x;
                        const uint4 uh = pair_units(hi2[0], hi2[1]);
                        if (y < p.H) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out_bf16) + unit * 16) = uh;
                        if (p.out_bf16_lo) {
                            const uint4 ul = pair_units(lo2[0], lo2[1]);
                            if (y < p.H) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out_bf16_lo) + unit * 16) = ul;
                        }
                        if (p.out_bf16_act) {
                            const uint4 ua = pair_units(ac2[0], ac2[1]);
                            if (y < p.H) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out_bf16_act) + unit * 16) = ua;
                        }
                    }
                }
            } else if (y < p.H) {
                float* dst = p.out + (size_t)b * p.out_bs + pix;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * HW] = val[px][r];
            }
        }
        if (p.pool_out || p.pool_bf16) {  // wave-uniform
            const int Wo = p.W / 2;
            if (p.pool_h == 2) {
                if (PW == 32 && NPX % 2 == 0)
#pragma unroll
                for (int pp = 0; pp < NPX / 2; ++pp) {  // row pairs (px-tiles 2pp, 2pp+1) of this wave
                    const int y = y0 + wave * WROWS + 2 * pp;  // even row of the pair
                    const int Ho = p.H / 2;
                    float* dst = p.pool_out + (size_t)b * (p.pool_bs ? (size_t)p.pool_bs : (size_t)p.N * Ho * Wo) + (size_t)(n0 + co * 32 + 4 * khalf) * Ho * Wo +
                                 (size_t)(y >> 1) * Wo + (x >> 1);
                    float pooled[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float o0 = lane_xor1(val[2 * pp][r]), o1 = lane_xor1(val[2 * pp + 1][r]);
                        float sum = val[2 * pp][r] + o0;
                        sum += val[2 * pp + 1][r];
                        sum += o1;
                        pooled[r] = sum * 0.25f;
                    }
                    if (p.pool_bf16) {  // blocked bf16 copies for the next encoder block (raw + its conv1 prologue applied)
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        const size_t clipo = (size_t)b * (p.pool_noct ? p.pool_noct : p.N / 8) * Ho * Wo;
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {  // octet pairs: whole 16-byte units per lane (pair_units)
                            uint2 raw2[2], act2[2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                bf16x4 raw, act;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    raw[i] = (__bf16)pooled[4 * (g + h) + i];
                                    act[i] = (__bf16)leaky(pooled[4 * (g + h) + i] * ps4[g + h][i] + ph4[g + h][i]);
                                }
                                raw2[h] = __builtin_bit_cast(uint2, raw);
                                act2[h] = __builtin_bit_cast(uint2, act);
                            }
                            const size_t unit = clipo + (size_t)(p.pool_oct0 + (n0 + co * 32) / 8 + g + khalf) * Ho * Wo + (size_t)(y >> 1) * Wo + (x >> 1);
                            const uint4 ur = pair_units(raw2[0], raw2[1]), ua = pair_units(act2[0], act2[1]);
                            if (!(lane & 1) && y + 1 < p.H) {
                                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.pool_bf16) + unit * 16) = ur;
                                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.pool_bf16_act) + unit * 16) = ua;
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (!(lane & 1) && y + 1 < p.H) dst[(size_t)((r & 3) + 8 * (r >> 2)) * Ho * Wo] = pooled[r];
                    }
                }
            } else {
#pragma unroll
                for (int px = 0; px < NPX; ++px) {
                    const int y = y0 + wave * WROWS + px * PH + ty;
                    float* dst = p.pool_out + (size_t)b * (p.pool_bs ? (size_t)p.pool_bs : (size_t)p.N * p.H * Wo) + (size_t)(n0 + co * 32 + 4 * khalf) * p.H * Wo +
                                 (size_t)min(y, p.H - 1) * Wo + (x >> 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sum = val[px][r] + lane_xor1(val[px][r]);
                        if (!(lane & 1) && y < p.H) dst[(size_t)((r & 3) + 8 * (r >> 2)) * p.H * Wo] = sum * 0.5f;
                    }
                }
            }
        }
    }
}


}  // namespace
