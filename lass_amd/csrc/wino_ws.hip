// wino_ws.hip - persistent, wave-specialised Winograd F(2x2,3x3) kernel for the 3x3 convolutions at W >= 32.
//
// Same arithmetic and fusions as wino.hip (reference: models/resunet.py:147-165), different schedule.  wino.hip runs the
// stage / transform / MFMA phases of a chunk one after the other inside a 4-wave workgroup and relies on a second
// workgroup per CU to fill the matrix pipe meanwhile; phase stamps showed the pipe ~65 % busy because the non-MFMA part of
// a chunk is latency-bound (global -> LDS -> transform -> LDS, three barriers) and longer than the MFMA part.  Here a
// workgroup is 8 waves = one CONSUMER and one PRODUCER wave per SIMD (roles are picked from HW_ID so that this holds
// whatever the wave placement):
//   * producers (256 threads) load 4x4 input patches straight from global (two 8-byte loads per patch row, one chunk
//     ahead, unconditional with clamped addresses), apply the BN+FiLM+leaky prologue and the zero padding, form
//     V = B^T d B in registers and write it to LDS; they also move the U slab global -> registers (one chunk ahead) ->
//     LDS (they have the registers to spare; measured: LDS-DMA moves only ~8-11 B/clk per CU, which made both this
//     kernel and wino.hip DMA-throughput-bound);
//   * consumers do nothing but fragment reads + 64 MFMAs per chunk, and the output transform / epilogue per tile.
// One raw s_barrier per chunk is the only synchronisation (chunk j+1 is produced while chunk j is consumed; V is
// double-buffered).  The workgroup is persistent (grid = #CUs, each walks a contiguous, XCD-local slice of the tile
// list) so a producer runs ahead into the next tile while the consumers are in the epilogue of the current one.
// The 1x1 shortcut phase uses chunks of 32 channels x 4 xi (same 128 LDS rows, same 64 MFMAs per chunk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "kernels.h"
#include "wino_common.h"

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load at 4-byte alignment (dword-aligned)

namespace {

constexpr int F_PRO = 1, F_PHASEB = 2, F_BIAS = 4, F_RES = 8, F_EPIACT = 16, F_PRECONV = 64, F_RESPRE = 128;
constexpr int WS_THREADS = 512;
constexpr int ROWS = 128;          // LDS rows per chunk: 16 xi x 8 channels (3x3) or 4 xi x 32 channels (shortcut)
constexpr int KCA = 8, KCB = 32;
constexpr int PWT = 16;            // Winograd tiles per row pair (32 output columns)
#ifndef PRODUCER_PRIO
#define PRODUCER_PRIO 3
#endif

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

template <int WCO, int WWT, int FLAGS>
__global__ __launch_bounds__(WS_THREADS, 1) void wino_ws_kernel(ConvArgs p, int ntiles) {
    static_assert(WCO * WWT == 4, "4 consumer waves");
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr bool HASB = (FLAGS & F_PHASEB) != 0;
    constexpr bool EPI = (FLAGS & F_EPIACT) != 0;
    constexpr bool BIAS = (FLAGS & F_BIAS) != 0;
    constexpr bool RES = (FLAGS & F_RES) != 0;
    constexpr bool PRE = (FLAGS & F_PRECONV) != 0;
    constexpr bool RESPRE = (FLAGS & F_RESPRE) != 0;
    constexpr int NT = 32 * WCO;            // couts per tile
    constexpr int NWT = 16 * WWT;           // Winograd tiles per tile
    constexpr int OR_ = 2 * WWT, OC = 2 * PWT;
    constexpr int VP = NWT + 16;            // V row pitch (floats)
    constexpr int V_F = ROWS * VP, U_F = ROWS * NT;
    constexpr int CSTEP = 256 / NWT;        // a producer thread owns ONE tile position and channels cbase + i*CSTEP
    constexpr int NIT = KCA / CSTEP;        // 3x3 patches per producer thread and chunk
    constexpr int NITB = KCB / CSTEP;       // shortcut items per producer thread and chunk
    constexpr int NF2 = 8 * NIT;            // 8-byte registers per prefetch set (NITB * 2 == NF2 as well)
    static_assert(NITB * 2 == NF2, "both chunk kinds fill the same prefetch set");
    constexpr int RPI = 256 / NT;           // LDS rows per 1-KiB DMA piece
    constexpr int NINSTR = ROWS / RPI / 4;  // 1-KiB slab pieces per producer wave and chunk
    static_assert(KCA % RPI == 0, "a slab piece stays inside one xi slot");

    __shared__ __attribute__((aligned(16))) float lds[2 * V_F + 2 * U_F];
    __shared__ int role_cnt[4];
    float* lv = lds;
    float* lu = lds + 2 * V_F;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HW = p.H * p.W;
    const int tiles_x = p.W / OC, tiles_y = (p.H + OR_ - 1) / OR_, NB = p.N / NT;
    const int nchA = p.Cin / KCA;
    const int nchB = HASB ? p.Cin2 / KCB : 0;

    // ---- roles: one consumer + one producer per SIMD ----------------------------------------------------------------
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    if (threadIdx.x < 4) role_cnt[threadIdx.x] = 0;
    __syncthreads();
    int rank = 0;
    if (lane == 0) rank = atomicAdd(&role_cnt[simd], 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
    const bool balanced = role_cnt[0] == 2 && role_cnt[1] == 2 && role_cnt[2] == 2 && role_cnt[3] == 2;
    const bool consumer = balanced ? rank == 0 : wave < 4;
    const int rid = balanced ? simd : (wave & 3);  // index within the role
    __syncthreads();

    // ---- tile list slice of this workgroup (XCD x = blockIdx & 7 owns a contiguous eighth) --------------------------
    int t_begin, t_end, t_step;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntiles + 7) / 8, x = blockIdx.x & 7;
        t_begin = x * per + (int)(blockIdx.x >> 3);
        t_end = min(ntiles, (x + 1) * per);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t_begin = blockIdx.x; t_end = ntiles; t_step = gridDim.x;
    }
    auto decode = [&](int t, int& b, int& n0, int& y0, int& x0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        n0 = (t % NB) * NT; b = t / NB;
        y0 = ty * OR_; x0 = tx * OC;
    };
#ifdef LASS_CONV_DIAG
    const long long k_c0 = clock64(), k_r0 = wall_clock64();
    long long dg[3] = {0, 0, 0};
#endif

    if (consumer) {
        // =============================================== CONSUMER ===================================================
        const int wco = rid / WWT, wwt = rid % WWT;
        const int kq = lane >> 4, l15 = lane & 15;
        const int sw = (kq & 1) * 16;  // undo the LDS-DMA source swizzle: odd rows hold their 16-float halves swapped
        const float* bfrag = lv + kq * VP + wwt * 16 + l15;
        const float* afrag0 = lu + kq * NT + wco * 32 + sw + l15;
        const float* afrag1 = lu + kq * NT + wco * 32 + (16 - sw) + l15;
        int vpar = 0;
        for (int t = t_begin; t < t_end; t += t_step) {
            int b, n0, y0, x0;
            decode(t, b, n0, y0, x0);
            f32x4 acc[16][2];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[xi][q][r] = 0.f;
            for (int ch = 0; ch < nchA; ++ch) {
#ifdef LASS_CONV_DIAG
                const long long t0 = clock64();
#endif
                lds_barrier();  // chunk produced
#ifdef LASS_CONV_DIAG
                const long long t1 = clock64();
#endif
#ifdef LASS_WS_EXP
                if (!(p.dbg_mode & 1))
#endif
                gemm_steps<32, NT, VP>(afrag0 + vpar * U_F, afrag1 + vpar * U_F, bfrag + vpar * V_F, acc,
                                       [](int s) { return s / 2; });
#ifdef LASS_CONV_DIAG
                dg[0] += t1 - t0; dg[1] += clock64() - t1;
#endif
                vpar ^= 1;
            }
            if (HASB) {
                for (int ch = 0; ch < nchB; ++ch) {
                    lds_barrier();
                    gemm_steps<32, NT, VP>(afrag0 + vpar * U_F, afrag1 + vpar * U_F, bfrag + vpar * V_F, acc, [](int s) {
                        const int q = s / 8;
                        return (q >> 1) * 4 + (q & 1) + 5;  // 5, 6, 9, 10
                    });
                    vpar ^= 1;
                }
            }
            // ---- output transform Y = A^T M A and epilogue ----------------------------------------------------------
#ifdef LASS_CONV_DIAG
            const long long te = clock64();
#endif
            const int oy = y0 + 2 * wwt, ox = x0 + 2 * l15;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wco * 32 + q * 16 + kq * 4 + r;
                    float s[2][4];
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        s[0][jx] = acc[0 + jx][q][r] + acc[4 + jx][q][r] + acc[8 + jx][q][r];
                        s[1][jx] = acc[4 + jx][q][r] - acc[8 + jx][q][r] - acc[12 + jx][q][r];
                    }
                    float y[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        y[i][0] = s[i][0] + s[i][1] + s[i][2];
                        y[i][1] = s[i][1] - s[i][2] - s[i][3];
                    }
                    const size_t pix = (size_t)n * HW + (size_t)oy * p.W + ox;
                    if (BIAS) {
                        const float bb = p.bias[n];
                        y[0][0] += bb; y[0][1] += bb; y[1][0] += bb; y[1][1] += bb;
                    }
                    if (RES) {
                        const float* rp = p.res + (size_t)b * p.res_bs + (RESPRE ? 0 : (size_t)n * HW) +
                                          (size_t)min(oy, p.H - 1) * p.W + ox;
                        float2 r0 = *reinterpret_cast<const float2*>(rp);
                        float2 r1 = *reinterpret_cast<const float2*>(rp + (oy + 1 < p.H ? p.W : 0));
                        if (RESPRE) {  // residual = pre_conv(x0): resunet.py:555,165
                            const float pw = p.pre_w[n], pb = p.pre_b[n];
                            r0.x = r0.x * pw + pb; r0.y = r0.y * pw + pb; r1.x = r1.x * pw + pb; r1.y = r1.y * pw + pb;
                        }
                        y[0][0] += r0.x; y[0][1] += r0.y; y[1][0] += r1.x; y[1][1] += r1.y;
                    }
                    if (EPI) {
                        const float es = p.epi_scale[n], eh = p.epi_shift[(size_t)b * p.epi_shift_bs + n];
                        y[0][0] = leaky(y[0][0] * es + eh); y[0][1] = leaky(y[0][1] * es + eh);
                        y[1][0] = leaky(y[1][0] * es + eh); y[1][1] = leaky(y[1][1] * es + eh);
                    }
                    float* dst = p.out + (size_t)b * p.out_bs + pix;
                    if (oy < p.H) *reinterpret_cast<float2*>(dst) = make_float2(y[0][0], y[0][1]);
                    if (oy + 1 < p.H) *reinterpret_cast<float2*>(dst + p.W) = make_float2(y[1][0], y[1][1]);
                    if (p.pool_out) {
                        const int Wo = p.W / 2;
                        if (p.pool_h == 2) {
                            float sum = y[0][0] + y[0][1];  // reference summation order (row-major)
                            sum += y[1][0];
                            sum += y[1][1];
                            if (oy + 1 < p.H)
                                p.pool_out[((size_t)b * p.N + n) * (p.H / 2) * Wo + (size_t)(oy >> 1) * Wo + (ox >> 1)] =
                                    sum * 0.25f;
                        } else {
                            float* pd = p.pool_out + ((size_t)b * p.N + n) * p.H * Wo + (size_t)oy * Wo + (ox >> 1);
                            if (oy < p.H) pd[0] = (y[0][0] + y[0][1]) * 0.5f;
                            if (oy + 1 < p.H) pd[Wo] = (y[1][0] + y[1][1]) * 0.5f;
                        }
                    }
                }
            }
#ifdef LASS_CONV_DIAG
            dg[2] += clock64() - te;
#endif
        }
#ifdef LASS_CONV_DIAG
        if (p.dbg && rid == 0 && lane == 0) {
            long long* d = p.dbg + 8 * (size_t)blockIdx.x;
            d[0] = dg[0]; d[1] = dg[1]; d[2] = dg[2];
            d[6] = clock64() - k_c0;
            d[7] = wall_clock64() - k_r0;
        }
#endif
    } else {
        // =============================================== PRODUCER ===================================================
        // Two waves share a SIMD's VALU issue, arbitrated by priority then age: without this the producer only gets the
        // slots the consumer's MFMA stream leaves over (measured: +1850 cycles per chunk)
        __builtin_amdgcn_s_setprio(PRODUCER_PRIO);
        const int ptid = rid * 64 + lane;       // 0..255
        const int wt = ptid % NWT, cbase = ptid / NWT;
        const int wty = wt / PWT, wtx = wt % PWT;
        // per-lane part of a slab piece's source address: row-in-piece * Nw + swizzled column (see wino.hip UDma)
        const int rl = lane / (NT / 4);
        const int ucol = ((lane % (NT / 4)) * 4) ^ ((rl & 1) << 4);
        const float* sc = PRO ? p.pro_scale : nullptr;

        f2u rs[2][NF2];
        f32x4 ur[2][NINSTR];
        float rsc[2][NIT], rsh[2][NIT], rpw[2][NIT], rpb[2][NIT];

        // load cursor (runs one chunk ahead of the chunk being processed)
        int tL = t_begin, chL = 0, phL = 0;
        bool validL = tL < t_end;
        int bL = 0, n0L = 0;
        int roff[4], colA = 0, colB = 0, offB = 0;
        unsigned flagsL = 0, flagsP = 0;
        auto geometry = [&]() {  // of tile tL
            int y0, x0;
            decode(tL, bL, n0L, y0, x0);
            const int gy0 = y0 + 2 * wty - 1, gx0 = x0 + 2 * wtx - 1;
            flagsL = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gy = gy0 + i;
                roff[i] = min(max(gy, 0), p.H - 1) * p.W;
                flagsL |= (gy >= 0 && gy < p.H ? 1u : 0u) << i;
            }
            const bool left = gx0 < 0, right = gx0 + 3 >= p.W;
            colA = left ? 0 : gx0;
            colB = right ? p.W - 2 : gx0 + 2;
            flagsL |= (left ? 16u : 0u) | (right ? 32u : 0u);
            const bool okB = gy0 + 2 < p.H;
            offB = min(gy0 + 1, p.H - 2) * p.W + gx0 + 1;
            flagsL |= okB ? 64u : 0u;
        };
        auto advance = [&]() {  // past the end the cursor stays on the last chunk (its loads are re-issued, unused)
            int ch = chL + 1, ph = phL, t = tL;
            if (ph == 0 && ch == nchA) {
                ch = 0;
                if (HASB) ph = 1; else t += t_step;
            } else if (HASB && ph == 1 && ch == nchB) {
                ch = 0; ph = 0; t += t_step;
            }
            validL = t < t_end;
            if (validL) { chL = ch; phL = ph; tL = t; }
        };
        auto loadA = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
            const float* in_b = p.in + (size_t)bL * p.in_bs;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = chL * KCA + cbase + it * CSTEP;
                const float* pc = PRE ? in_b : in_b + (size_t)c * HW;  // PRE: every channel reads the one x0 plane
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rs[S][it * 8 + 2 * i] = *reinterpret_cast<const f2u*>(pc + roff[i] + colA);
                    rs[S][it * 8 + 2 * i + 1] = *reinterpret_cast<const f2u*>(pc + roff[i] + colB);
                }
                if (PRO) {
                    rsc[S][it] = sc[c];
                    rsh[S][it] = p.pro_shift[(size_t)bL * p.pro_shift_bs + c];
                }
                if (PRE) {
                    rpw[S][it] = p.pre_w[c];
                    rpb[S][it] = p.pre_b[c];
                }
            }
        };
        // shortcut-capable kernels (never PRO / PRE): one branch-free load sequence serves both chunk kinds, so that no
        // vector-memory load sits in divergent control flow (hipcc answers that with vmcnt(0) in front of the next use)
        auto loadAB = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
            const bool isA = phL == 0;
            const float* base = isA ? p.in + (size_t)bL * p.in_bs : p.in2 + (size_t)bL * p.in2_bs;
            const int c0 = (isA ? chL * KCA : chL * KCB) + cbase;
#pragma unroll
            for (int k = 0; k < NF2; ++k) {
                const int chan = c0 + (isA ? (k / 8) * CSTEP : (k / 2) * CSTEP);
                const int sp = isA ? roff[(k % 8) / 2] + ((k & 1) ? colB : colA) : offB + ((k & 1) ? p.W : 0);
                rs[S][k] = *reinterpret_cast<const f2u*>(base + (size_t)chan * HW + sp);
            }
        };
        // U slab of the chunk at the load cursor -> registers; written to LDS (lane-linear 1-KiB pieces, same layout as
        // the LDS-DMA of wino.hip) when that chunk is processed
        auto loadU = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
            const bool isA = !HASB || phL == 0;
            const float* lp = (isA ? p.w_wino : p.w2_wino) + (size_t)rl * p.Nw + n0L + ucol;
            const int cin_ = isA ? p.Cin : p.Cin2;
            const int c0 = isA ? chL * KCA : chL * KCB;
#pragma unroll
            for (int i = 0; i < NINSTR; ++i) {
                const int r0 = (rid * NINSTR + i) * RPI;
                const int xi = isA ? r0 / KCA : r0 / KCB, c = isA ? r0 % KCA : r0 % KCB;
                ur[S][i] = *reinterpret_cast<const f32x4*>(lp + ((size_t)xi * cin_ + c0 + c) * p.Nw);
            }
        };
        auto storeU = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int i = 0; i < NINSTR; ++i) {
                const int r0 = (rid * NINSTR + i) * RPI;
                *reinterpret_cast<f32x4*>(lu + S * U_F + r0 * NT + lane * 4) = ur[S][i];
            }
        };
        auto processA = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
            const bool left = (flagsP & 16u) != 0, right = (flagsP & 32u) != 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float d[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f2u a = rs[S][it * 8 + 2 * i], bq = rs[S][it * 8 + 2 * i + 1];
                    float v0 = a.x, v1 = left ? a.x : a.y, v2 = right ? bq.y : bq.x, v3 = bq.y;
                    if (PRE) {
                        const float pw = rpw[S][it], pb = rpb[S][it];
                        v0 = v0 * pw + pb; v1 = v1 * pw + pb; v2 = v2 * pw + pb; v3 = v3 * pw + pb;
                    }
                    if (PRO) {
                        const float s1 = rsc[S][it], s2 = rsh[S][it];
                        v0 = leaky(v0 * s1 + s2); v1 = leaky(v1 * s1 + s2);
                        v2 = leaky(v2 * s1 + s2); v3 = leaky(v3 * s1 + s2);
                    }
                    const bool rok = ((flagsP >> i) & 1u) != 0;  // zero padding comes AFTER the activation
                    d[i][0] = (rok && !left) ? v0 : 0.f;
                    d[i][1] = rok ? v1 : 0.f;
                    d[i][2] = rok ? v2 : 0.f;
                    d[i][3] = (rok && !right) ? v3 : 0.f;
                }
                float tt[4][4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    tt[0][jx] = d[0][jx] - d[2][jx];
                    tt[1][jx] = d[1][jx] + d[2][jx];
                    tt[2][jx] = d[2][jx] - d[1][jx];
                    tt[3][jx] = d[1][jx] - d[3][jx];
                }
                float* dst = lv + S * V_F + (cbase + it * CSTEP) * VP + wt;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dst[(4 * i + 0) * (KCA * VP)] = tt[i][0] - tt[i][2];
                    dst[(4 * i + 1) * (KCA * VP)] = tt[i][1] + tt[i][2];
                    dst[(4 * i + 2) * (KCA * VP)] = tt[i][2] - tt[i][1];
                    dst[(4 * i + 3) * (KCA * VP)] = tt[i][1] - tt[i][3];
                }
            }
        };
        auto processB = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
            const bool ok = (flagsP & 64u) != 0;
#pragma unroll
            for (int it = 0; it < NITB; ++it) {
                f2u r1 = rs[S][2 * it], r2 = rs[S][2 * it + 1];  // patch rows 1,2 x cols 1,2
                if (!ok) { r1.x = r1.y = r2.x = r2.y = 0.f; }
                const float t1a = r1.x + r2.x, t1b = r1.y + r2.y;
                const float t2a = r2.x - r1.x, t2b = r2.y - r1.y;
                float* dst = lv + S * V_F + (cbase + it * CSTEP) * VP + wt;
                dst[0 * (KCB * VP)] = t1a + t1b;
                dst[1 * (KCB * VP)] = t1b - t1a;
                dst[2 * (KCB * VP)] = t2a + t2b;
                dst[3 * (KCB * VP)] = t2b - t2a;
            }
        };

        int phP = 0;
        if (validL) {
            geometry();
            loadA(std::integral_constant<int, 0>{});
            loadU(std::integral_constant<int, 0>{});
            flagsP = flagsL;
        }
        bool validP = validL;
        auto iter = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
#ifdef LASS_CONV_DIAG
            const long long t0 = clock64();
#endif
            const int tprev = tL;
            advance();
            if (tL != tprev) geometry();
#ifdef LASS_WS_EXP
            if (!(p.dbg_mode & 2)) {
#endif
            if (HASB) loadAB(std::integral_constant<int, 1 - S>{});
            else loadA(std::integral_constant<int, 1 - S>{});
#ifdef LASS_WS_EXP
            }
            if (!(p.dbg_mode & 4))
#endif
            loadU(std::integral_constant<int, 1 - S>{});
            __builtin_amdgcn_sched_barrier(0);
#ifdef LASS_CONV_DIAG
            const long long t1 = clock64();
#endif
#ifdef LASS_WS_EXP
            if (!(p.dbg_mode & 8))
#endif
            if (!HASB || phP == 0) processA(setc); else processB(setc);
            storeU(setc);
            __builtin_amdgcn_sched_barrier(0);
#ifdef LASS_CONV_DIAG
            const long long t2 = clock64();
#endif
            lds_barrier();  // V and U of this chunk are in LDS (the next chunk's loads stay in flight)
#ifdef LASS_CONV_DIAG
            dg[0] += t1 - t0; dg[1] += t2 - t1; dg[2] += clock64() - t2;
#endif
            flagsP = flagsL; phP = phL; validP = validL;
        };
        while (validP) {
            iter(std::integral_constant<int, 0>{});
            if (!validP) break;
            iter(std::integral_constant<int, 1>{});
        }
#ifdef LASS_CONV_DIAG
        if (p.dbg && rid == 0 && lane == 0) {
            long long* d = p.dbg + 8 * (size_t)blockIdx.x;
            d[3] = dg[0]; d[4] = dg[1]; d[5] = dg[2];
        }
#endif
    }
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

template <int FLAGS>
hipError_t launch_ws(const ConvArgs& p0, hipStream_t stream) {
    ConvArgs p = p0;
    const bool wide = p.N % 64 == 0;
    const int NT = wide ? 64 : 32, OR_ = wide ? 4 : 8;
    const int ntiles = (p.W / 32) * ((p.H + OR_ - 1) / OR_) * (p.N / NT) * p.B;
    const int grid = ntiles < num_cus() ? ntiles : num_cus();
#ifdef LASS_CONV_DIAG
    static long long* dbuf = nullptr;
    static size_t dcap = 0;
    const size_t nblk = (size_t)grid;
    if (nblk > dcap) {
        if (dbuf) (void)hipFree(dbuf);
        (void)hipMalloc((void**)&dbuf, nblk * 64);
        dcap = nblk;
    }
    (void)hipMemsetAsync(dbuf, 0, nblk * 64, stream);
    p.dbg = dbuf;
#endif
#ifdef LASS_WS_EXP
    if (const char* e = getenv("LASS_WS_DBG")) p.dbg_mode = atoi(e);
#endif
    if (wide)
        hipLaunchKernelGGL((wino_ws_kernel<2, 2, FLAGS>), dim3(grid), dim3(WS_THREADS), 0, stream, p, ntiles);
    else
        hipLaunchKernelGGL((wino_ws_kernel<1, 4, FLAGS>), dim3(grid), dim3(WS_THREADS), 0, stream, p, ntiles);
#ifdef LASS_CONV_DIAG
    {
        std::vector<long long> h(nblk * 8);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), dbuf, nblk * 64, hipMemcpyDeviceToHost);
        double s[8] = {0};
        for (size_t i = 0; i < nblk; ++i)
            for (int k = 0; k < 8; ++k) s[k] += (double)h[i * 8 + k];
        const double nchunks = ((double)p.Cin / 8.0 + ((FLAGS & F_PHASEB) ? p.Cin2 / 32.0 : 0.0)) * ntiles / (double)nblk;
        for (double& v : s) v /= (double)nblk;
        fprintf(stderr,
                "[wino-ws-diag] Cin=%d Cin2=%d N=%d %dx%d tiles=%d wgs=%zu | consumer per chunk: barrier %.0f  mfma %.0f  "
                "epilogue %.0f | producer per chunk: issue %.0f  transform %.0f  wait+barrier %.0f | total %.0f cycles, "
                "clock %.3f GHz\n",
                p.Cin, (FLAGS & F_PHASEB) ? p.Cin2 : 0, p.N, p.H, p.W, ntiles, nblk, s[0] / nchunks, s[1] / nchunks,
                s[2] / nchunks, s[3] / nchunks, s[4] / nchunks, s[5] / nchunks, s[6], s[6] / s[7] * 0.1);
    }
#endif
    return hipGetLastError();
}

}  // namespace

bool lass_wino_ws_supported(const ConvArgs& p) {
    return p.W >= 32 && (p.W % 32) == 0 && (p.H % 2) == 0 && p.H >= 2 && p.Cin % KCA == 0 && p.N % 32 == 0 &&
           (p.Nw % 4) == 0;
}

hipError_t lass_launch_wino_ws(ConvKind kind, const ConvArgs& p, hipStream_t stream) {
    if (!lass_wino_ws_supported(p) || !p.w_wino || !p.in || !p.out) return hipErrorInvalidValue;
    switch (kind) {
        case CONV1_ACT:
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift) return hipErrorInvalidValue;
            return launch_ws<F_PRO | F_EPIACT>(p, stream);
        case CONV2_IDENT:
            if (!p.res) return hipErrorInvalidValue;
            return launch_ws<F_RES>(p, stream);
        case CONV2_SHORTCUT:
            if (!p.in2 || !p.w2_wino || !p.bias || p.Cin2 % KCB != 0) return hipErrorInvalidValue;
            return launch_ws<F_PHASEB | F_BIAS>(p, stream);
        case CONV1_ACT_PRE:
            if (!p.pro_scale || !p.pro_shift || !p.epi_scale || !p.epi_shift || !p.pre_w || !p.pre_b || p.N != 32 ||
                p.Cin != 32)
                return hipErrorInvalidValue;
            return launch_ws<F_PRO | F_EPIACT | F_PRECONV>(p, stream);
        case CONV2_IDENT_PRE:
            if (!p.res || !p.pre_w || !p.pre_b || p.N != 32) return hipErrorInvalidValue;
            return launch_ws<F_RES | F_RESPRE>(p, stream);
        default:
            return hipErrorInvalidValue;
    }
}
