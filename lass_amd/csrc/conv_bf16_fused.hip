// conv_bf16_fused.hip - bf16 mode: encoder_block1's ConvBlockRes (models/resunet.py:147-165 at the shape of :315-323) as ONE
// kernel.  The block is byte-bound in bf16 (32 channels at 1024 x 512: MFMA busy 0.10-0.13 in the two-launch form), and half
// of its bytes were the 32-channel intermediate going out to HBM from conv1's epilogue and coming back into conv2.  Here it
// never leaves the CU:
//   x0 tile (f32, 12 x 36)  --pre_conv + BN/FiLM/leaky, bf16-->  LDS image [octet][12][36]
//   conv1 (3x3, 32 -> 32) over the 10 x 34 pixels conv2 needs (a 1-pixel recomputed halo: ten 32-pixel row tiles plus one
//   tile holding the two halo columns), + bn2/FiLM/leaky, zero outside the image  -->  LDS image [octet][10][34]
//   conv2 (3x3, 32 -> 32) over it  + pre_conv(x0) residual  -->  the skip as blocked bf16 copies and the 2x2 avg-pool as
//   blocked bf16 copies (the epilogue of conv_bf16.hip's kernels, store_tile).
// Same MFMA (v_mfma_f32_32x32x16_bf16), LDS layouts, fragment reads and epilogue as conv_bf16.hip; conv1 runs 11 pixel tiles
// where the two-launch form ran 8 (x 1.19 MFMAs for the block: free at this MFMA load).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "conv_common.h"
#include "kernels.h"
#include "wino_common.h"  // make_rsrc_words, lds_dma_16B, wait_vmcnt

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FPW = 32, FPHT = 8;                           // output tile: 8 rows x 32 columns, 4 waves x 2 rows
constexpr int IRI = FPHT + 4, IPI = FPW + 4, NPI = IRI * IPI;  // input image (halo 2): 12 x 36 pixels
constexpr int IRM = FPHT + 2, IPM = FPW + 2, NPM = IRM * IPM;  // conv1 output = conv2 input (halo 1): 10 x 34 pixels
constexpr int NPPI = (NPI + NTHREADS - 1) / NTHREADS;          // input pixels per thread
constexpr int W_U4 = 9 * 2 * 32;                               // 16-byte units of one 16-channel weight chunk (32 couts)
constexpr int IMG1_U4 = 2 * NPI, MID_U4 = 4 * NPM + 64;        // (+ slack: lanes of a discarded tile read past the image)

// p: conv1 of the block (CONV1_ACT_PRE arguments: x0, pre_w / pre_b, prologue and epilogue tables, w_bf16);
// q: conv2 (CONV2_IDENT_PRE arguments with blocked bf16 outputs: res = x0, out_bf16 (+ _act), pool_bf16 (+ _act), w_bf16).
__global__ __launch_bounds__(NTHREADS) void enc1_fused_bf16_kernel(ConvArgs p, ConvArgs q) {
    // conv2's weights (2 chunks) take over the input image + conv1 weight regions once conv1 is done: 47 KB per workgroup,
    // three workgroups per CU
    static_assert(2 * W_U4 <= IMG1_U4 + W_U4, "conv2's weights must fit the regions conv1 leaves behind");
    __shared__ uint4 lds4[IMG1_U4 + W_U4 + MID_U4 + (64 + 128) / 4];
    uint4* img1 = lds4;
    uint4* w1 = img1 + IMG1_U4;
    uint4* mid = w1 + W_U4;
    uint4* w2 = lds4;
    float* tabs = reinterpret_cast<float*>(mid + MID_U4);
    float* lds_es = tabs;        // conv1 epilogue (bn2 + FiLM) scale / shift
    float* lds_eh = tabs + 32;
    float* lds_act = tabs + 64;  // store_tile: skip activation scale / shift, pooled activation scale / shift

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, j = lane & 31;
    const int tiles_x = p.W / FPW;
    const int bz = blockIdx.x / ((unsigned)tiles_x * ((p.H + FPHT - 1) / FPHT));
    const int bxy = blockIdx.x - bz * tiles_x * ((p.H + FPHT - 1) / FPHT);
    const int y0 = (bxy / tiles_x) * FPHT, x0 = (bxy % tiles_x) * FPW;
    const int b = bz;
    const int HW = p.H * p.W;

    if (tid < 32) {
        lds_es[tid] = p.epi_scale[tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + tid];
        if (q.out_bf16_act) {
            lds_act[tid] = q.act_scale[tid];
            lds_act[32 + tid] = q.act_shift[(size_t)b * q.act_shift_bs + tid];
        }
        if (q.pool_bf16) {
            lds_act[64 + tid] = q.pool_act_scale[tid];
            lds_act[96 + tid] = q.pool_act_shift[(size_t)b * q.act_shift_bs + tid];
        }
    }

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
    const v4i32 w1_rs = make_rsrc_words(p.w_bf16, 2u * W_U4 * 16u);
    const v4i32 w2_rs = make_rsrc_words(q.w_bf16, 2u * W_U4 * 16u);

    // ---- x0 at this thread's pixels of the 12 x 36 input tile (loaded once; every channel is an affine function of it) -----
    const float* x0_b = p.in + (size_t)b * p.in_bs;
    float x0v[NPPI];
    unsigned okbits = 0;
#pragma unroll
    for (int k = 0; k < NPPI; ++k) {
        const int u = min(tid + k * NTHREADS, NPI - 1);
        const int gy = y0 + u / IPI - 2, gx = x0 + u % IPI - 2;
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        x0v[k] = x0_b[min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1)];
        okbits |= (ok ? 1u : 0u) << k;
    }

    // ---- conv1 over the 10 x 34 intermediate pixels: three pixel tiles per wave ---------------------------------------------
    // slot s of wave w: rows w, w + 4, w + 8 (< 10) at columns 1..32; the third slot of wave 2 holds the halo columns 0 and 33
    // (lane j: row j / 2, column 33 * (j & 1)); surplus lanes / slots compute on a valid address and are not written.
    int mir[3], mic[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int ir = wave + 4 * s, ic = 1 + j;
        if (s == 2 && wave >= 2) {
            ir = wave == 2 ? (j >> 1) : IRM;  // wave 3: nothing
            ic = (j & 1) ? IPM - 1 : 0;
        }
        mir[s] = ir;
        mic[s] = ic;
    }
    f32x16 acc1[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[s][r] = 0.f;

    for (int c = 0; c < 2; ++c) {
        if (c) __syncthreads();  // chunk 0 has been contracted: image and weight regions are free
        for (int piece = wave; piece < W_U4 / 64; piece += 4)
            lds_dma_16B(w1_rs, (unsigned)lane * 16u, (unsigned)((c * W_U4 + piece * 64) * 16), lds0 + (unsigned)((IMG1_U4 + piece * 64) * 16));
        float pcw[16], pcb[16], psc[16], psh[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            pcw[i] = p.pre_w[c * 16 + i];
            pcb[i] = p.pre_b[c * 16 + i];
            psc[i] = p.pro_scale[c * 16 + i];
            psh[i] = p.pro_shift[(size_t)b * p.pro_shift_bs + c * 16 + i];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NPPI; ++k) {
                const int u = tid + k * NTHREADS;
                const bool ok = (okbits >> k) & 1u;
                bf16x8 pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float t = x0v[k] * pcw[o * 8 + i] + pcb[o * 8 + i];       // pre_conv (resunet.py:555)
                    t = leaky(t * psc[o * 8 + i] + psh[o * 8 + i]);           // bn1 + FiLM + leaky (:150)
                    pk[i] = (__bf16)(ok ? t : 0.f);                           // conv zero padding comes after the activation
                }
                if (u < NPI) *reinterpret_cast<bf16x8*>(img1 + o * NPI + u) = pk;
            }
        wait_vmcnt<0>();
        __syncthreads();
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(w1) + khalf * 32 + j;
        const bf16x8* ibase = reinterpret_cast<const bf16x8*>(img1) + khalf * NPI;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16x8 a = abase[tap * 64];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ir = min(mir[s], IRM - 1);
                const bf16x8 bb = ibase[(ir + tap / 3) * IPI + mic[s] + tap % 3];
                acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc1[s], 0, 0, 0);
            }
        }
    }

    __syncthreads();  // every wave has finished with the input image and conv1's weights: conv2's weights move in (LDS-DMA)
    for (int piece = wave; piece < 2 * W_U4 / 64; piece += 4)
        lds_dma_16B(w2_rs, (unsigned)lane * 16u, (unsigned)piece * 1024u, lds0 + (unsigned)(piece * 1024));
    // ---- conv1 epilogue: bn2 + FiLM + leaky (resunet.py:151), zero outside the image, bf16 -> the intermediate image ------
    {
        float es4[4][4], eh4[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(lds_es + 8 * g + 4 * khalf);
            const float4 c = *reinterpret_cast<const float4*>(lds_eh + 8 * g + 4 * khalf);
            es4[g][0] = a.x; es4[g][1] = a.y; es4[g][2] = a.z; es4[g][3] = a.w;
            eh4[g][0] = c.x; eh4[g][1] = c.y; eh4[g][2] = c.z; eh4[g][3] = c.w;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int ir = mir[s], ic = mic[s];
            const int gy = y0 - 1 + ir, gx = x0 - 1 + ic;
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            if (ir < IRM) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = leaky(acc1[s][4 * g + i] * es4[g][i] + eh4[g][i]);
                        v[i] = (__bf16)(inside ? t : 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(mid + g * NPM + ir * IPM + ic) + khalf * 8) = v;
                }
            }
        }
    }
    // residual = pre_conv(x0) at this lane's output pixels (resunet.py:555,165), fetched while the intermediate settles
    const int x = x0 + j;
    float rtmp[2][16];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        const int y = min(y0 + wave * 2 + px, q.H - 1);
        const float xv = q.res[(size_t)b * q.res_bs + (size_t)y * q.W + x];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * khalf;
            rtmp[px][r] = xv * q.pre_w[n] + q.pre_b[n];
        }
    }
    wait_vmcnt<0>();  // conv2's weights
    __syncthreads();

    // ---- conv2 over the intermediate image: the fragment scheme of conv_bf16.hip (image [octet][10][34], 2 rows per wave) ---
    f32x16 acc2[1][2];
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][px][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bf16x8* bbase = reinterpret_cast<const bf16x8*>(mid) + (2 * c + khalf) * NPM + (wave * 2) * IPM + j;
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(w2 + c * W_U4) + khalf * 32 + j;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16x8 a = abase[tap * 64];
#pragma unroll
            for (int px = 0; px < 2; ++px)
                acc2[0][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bbase[(px + tap / 3) * IPM + tap % 3], acc2[0][px], 0, 0, 0);
        }
    }
    store_tile<1, 2, FPW, F_RES | F_RESPRE | F_OUTBF16, true>(q, acc2, rtmp, nullptr, nullptr, b, 0, y0, x0, lane, wave, nullptr,
                                                                 lds_act);
}

}  // namespace

bool lass_enc1_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q) {
    return p.Cin == 32 && p.N == 32 && p.Nw == 32 && q.Cin == 32 && q.N == 32 && q.Nw == 32 && p.W % FPW == 0 && p.W >= FPW &&
           p.w_bf16 && q.w_bf16 && !p.w_bf16_lo && !q.w_bf16_lo && p.pre_w && p.pre_b && p.pro_scale && p.pro_shift && p.epi_scale &&
           p.epi_shift && q.res && q.pre_w && q.pre_b && q.out_bf16 && !q.out_bf16_lo && q.out_noct > 0 && (!q.pool_out) &&
           (!q.pool_bf16 || (q.pool_h == 2 && q.pool_bf16_act && q.H % 2 == 0)) && q.H == p.H && q.W == p.W && q.B == p.B;
}

hipError_t lass_launch_enc1_fused_bf16(const ConvArgs& p, const ConvArgs& q, hipStream_t stream) {
    if (!lass_enc1_fused_bf16_supported(p, q)) return hipErrorInvalidValue;
    const long nblk = (long)(p.W / FPW) * ((p.H + FPHT - 1) / FPHT) * p.B;
    hipLaunchKernelGGL(enc1_fused_bf16_kernel, dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, p, q);
    return hipGetLastError();
}
