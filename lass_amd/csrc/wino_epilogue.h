// wino_epilogue.h - output transform Y = A^T M A and the fused epilogue of the Winograd kernels (wino.hip, wino32.hip):
// bias / residual / epilogue activation / 2x2 (or 1x2) avg-pool / output head (after_conv + complex ratio mask).
// A wave holds 32 couts x 16 tiles with all 16 xi accumulators: acc[xi][t][r] = M_xi[cout = t*16 + kq*4 + r][tile = l15]
// (kq = lane >> 4, l15 = lane & 15), so everything below is register-local.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "pixel_ops.h"
#include "wino_common.h"

namespace {

constexpr int WF_BIAS = 4, WF_RES = 8, WF_EPIACT = 16, WF_RESPRE = 128, WF_MASK = 1024;  // = the F_* flags of wino.hip

// (oy, ox): top-left output pixel of this lane's 2x2 tile; n0: first output channel of the workgroup's cout block, nl0:
// this wave's first channel inside that block (index base of the LDS tables).
template <int FLAGS>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& p, f32x4 (&acc)[16][2], int b, int n0, int nl0, int oy, int ox,
                                              int lane, const float* lds_bias, const float* lds_es, const float* lds_eh,
                                              const float* lds_pw, const float* lds_pb, const float* lds_mw) {
    constexpr bool EPI = (FLAGS & WF_EPIACT) != 0;
    constexpr bool BIAS = (FLAGS & WF_BIAS) != 0;
    constexpr bool RES = (FLAGS & WF_RES) != 0;
    constexpr bool RESPRE = (FLAGS & WF_RESPRE) != 0;
    constexpr bool MASK = (FLAGS & WF_MASK) != 0;
    const int kq = lane >> 4;
    const int HW = p.H * p.W;
    float ml[3][2][2] = {};  // MASK: this lane's partial after_conv logits of its 2x2 pixels (8 of the 32 channels)
    // Residual values of all 8 channels first, in one batch of loads: inside the store loop below the compiler cannot
    // move a load of p.res above the preceding store to p.out (they may alias for all it knows), which serialised eight
    // global-load round trips per workgroup (16 000 cycles of a 44 000-cycle encoder_block1.conv2 block).
    float2 rres[(RES && !RESPRE) ? 2 : 1][(RES && !RESPRE) ? 4 : 1][2];
    if (RES) {
        const size_t rpix = (size_t)min(oy, p.H - 1) * p.W + ox;
        const int rrow = oy + 1 < p.H ? p.W : 0;
        if (RESPRE) {  // one x0 patch serves every channel
            const float* rp = p.res + (size_t)b * p.res_bs + rpix;
            rres[0][0][0] = *reinterpret_cast<const float2*>(rp);
            rres[0][0][1] = *reinterpret_cast<const float2*>(rp + rrow);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* rp = p.res + (size_t)b * p.res_bs + (size_t)(n0 + nl0 + t * 16 + kq * 4 + r) * HW + rpix;
                    rres[(RES && !RESPRE) ? t : 0][(RES && !RESPRE) ? r : 0][0] = *reinterpret_cast<const float2*>(rp);
                    rres[(RES && !RESPRE) ? t : 0][(RES && !RESPRE) ? r : 0][1] = *reinterpret_cast<const float2*>(rp + rrow);
                }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = nl0 + t * 16 + kq * 4 + r;  // channel within the block
            const int n = n0 + nl;
            float s[2][4];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                s[0][jx] = acc[0 + jx][t][r] + acc[4 + jx][t][r] + acc[8 + jx][t][r];
                s[1][jx] = acc[4 + jx][t][r] - acc[8 + jx][t][r] - acc[12 + jx][t][r];
            }
            float y[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[i][0] = s[i][0] + s[i][1] + s[i][2];
                y[i][1] = s[i][1] - s[i][2] - s[i][3];
            }
            const size_t pix = (size_t)n * HW + (size_t)oy * p.W + ox;
            if (BIAS) {
                const float bb = lds_bias[nl];
                y[0][0] += bb; y[0][1] += bb; y[1][0] += bb; y[1][1] += bb;
            }
            if (RES) {
                float2 r0 = rres[RESPRE ? 0 : t][RESPRE ? 0 : r][0], r1 = rres[RESPRE ? 0 : t][RESPRE ? 0 : r][1];
                if (RESPRE) {  // residual = pre_conv(x0): resunet.py:555,165
                    const float pw = lds_pw[nl], pb = lds_pb[nl];
                    r0.x = r0.x * pw + pb; r0.y = r0.y * pw + pb; r1.x = r1.x * pw + pb; r1.y = r1.y * pw + pb;
                }
                y[0][0] += r0.x; y[0][1] += r0.y; y[1][0] += r1.x; y[1][1] += r1.y;
            }
            if (EPI) {
                const float es = lds_es[nl], eh = lds_eh[nl];
                y[0][0] = leaky(y[0][0] * es + eh); y[0][1] = leaky(y[0][1] * es + eh);
                y[1][0] = leaky(y[1][0] * es + eh); y[1][1] = leaky(y[1][1] * es + eh);
            }
            if (MASK) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float wk = lds_mw[k * 32 + nl];
                    ml[k][0][0] += wk * y[0][0]; ml[k][0][1] += wk * y[0][1];
                    ml[k][1][0] += wk * y[1][0]; ml[k][1][1] += wk * y[1][1];
                }
                continue;
            }
            // 16-byte stores: the lane pair (l, l^1) holds two horizontally adjacent 2x2 tiles; the even lane takes the upper
            // row of both (4 consecutive floats), the odd lane the lower row - 8 dwordx4 instead of 16 dwordx2 stores per
            // lane (the store tail of a workgroup is issue-bound).  H is even, so both rows of a tile are in range together.
            {
                const bool odd = (lane & 1) != 0;
                const float sx = odd ? y[0][0] : y[1][0], sy = odd ? y[0][1] : y[1][1];  // the row the partner stores
                const float rx = lane_xor1(sx), ry = lane_xor1(sy);
                float* dst = p.out + (size_t)b * p.out_bs + pix;
                const float4 v = odd ? make_float4(rx, ry, y[1][0], y[1][1]) : make_float4(y[0][0], y[0][1], rx, ry);
                if (oy < p.H) *reinterpret_cast<float4*>(odd ? dst + p.W - 2 : dst) = v;
            }
            if (p.pool_out) {
                const int Wo = p.W / 2;
                const size_t pool_bs = p.pool_bs ? (size_t)p.pool_bs : (size_t)p.N * (p.H / p.pool_h) * Wo;
                if (p.pool_h == 2) {
                    float sum = y[0][0] + y[0][1];  // reference summation order (row-major)
                    sum += y[1][0];
                    sum += y[1][1];
                    if (oy + 1 < p.H)
                        p.pool_out[(size_t)b * pool_bs + (size_t)n * (p.H / 2) * Wo + (size_t)(oy >> 1) * Wo + (ox >> 1)] =
                            sum * 0.25f;
                } else {
                    float* pd = p.pool_out + (size_t)b * pool_bs + (size_t)n * p.H * Wo + (size_t)oy * Wo + (ox >> 1);
                    if (oy < p.H) pd[0] = (y[0][0] + y[0][1]) * 0.5f;
                    if (oy + 1 < p.H) pd[Wo] = (y[1][0] + y[1][1]) * 0.5f;
                }
            }
        }
    }
    if (MASK) {
        // the four lanes l15 + 16*kq hold the tile's 32 channels between them: butterfly over kq, then lane kq finishes
        // pixel (kq >> 1, kq & 1) of the 2x2 tile
        float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = ml[k][i][j];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (kq == i * 2 + j) l[k] = v + lds_mw[96 + k];
                }
        const int t = oy + (kq >> 1), f = ox + (kq & 1);
        if (t < p.mask_T) mask_pixel(p, b, t, f, l[0], l[1], l[2]);
    }
}

}  // namespace
