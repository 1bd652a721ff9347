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

// Tile index of this workgroup: workgroup g runs on XCD g % 8 (its own L2); every XCD walks ONE contiguous range of tiles,
// so the tiles that share halo rows / columns run side by side under the same L2 (as block_coords, wino_common.h).
__device__ __forceinline__ unsigned fused_tile_index() {
    const unsigned lin = blockIdx.x;
    return (gridDim.x & 7u) == 0 ? (lin & 7u) * (gridDim.x >> 3) + (lin >> 3) : lin;
}

constexpr int FPW = 32, FPHT = 8;                           // output tile: 8 rows x 32 columns, 4 waves x 2 rows
constexpr int IRI = FPHT + 4, IPI = FPW + 4, NPI = IRI * IPI;  // input image (halo 2): 12 x 36 pixels
constexpr int IRM = FPHT + 2, IPM = FPW + 2, NPM = IRM * IPM;  // conv1 output = conv2 input (halo 1): 10 x 34 pixels
constexpr int NPPI = (NPI + NTHREADS - 1) / NTHREADS;          // input pixels per thread
constexpr int W_U4 = 9 * 2 * 32;                               // 16-byte units of one 16-channel weight chunk (32 couts)
constexpr int IMG1_U4 = 2 * NPI, MID_U4 = 4 * NPM + 64;        // (+ slack: lanes of a discarded tile read past the image)

// p: conv1 of the block (CONV1_ACT_PRE arguments: x0, pre_w / pre_b, prologue and epilogue tables, w_bf16);
// q: conv2 (CONV2_IDENT_PRE arguments with blocked bf16 outputs: res = x0, out_bf16 (+ _act), pool_bf16 (+ _act), w_bf16).
__global__ __launch_bounds__(NTHREADS, 3) void enc1_fused_bf16_kernel(ConvArgs p, ConvArgs q) {
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
    const unsigned tile = fused_tile_index();
    const int bz = tile / ((unsigned)tiles_x * ((p.H + FPHT - 1) / FPHT));
    const int bxy = tile - bz * tiles_x * ((p.H + FPHT - 1) / FPHT);
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


// ---- decoder_block6's ConvBlockRes (resunet.py:147-165 at the shape of :408-418) with the output head, ONE kernel ----------
//   cat (activated blocked-bf16 copy, 64 channels, 12 x 36 tile)  --global -> registers -> LDS, 4 chunks of 16 channels-->
//   conv1 (3x3, 64 -> 32) over the 10 x 34 pixels conv2 needs, + bn2/FiLM/leaky, zero outside the image --> LDS image
//   conv2 (3x3, 32 -> 32) over it  +  the 1x1 shortcut over the RAW copy of the cat (operands straight from global memory
//   into the MFMA: a lane's B fragment is one 16-byte unit of the blocked layout, its A fragment one unit of the weights)
//   + bias  -->  after_conv + complex ratio mask (store_tile's F_MASK epilogue): out_real / out_imag.
// The 32-channel intermediate (0.54 GB written and read back with its halo per step) never leaves the CU.
// p: conv1 (CONV1_ACT arguments in the blocked pipeline: in_bf16 = activated cat copy, w_bf16, epilogue tables);
// q: conv2 (CONV2_SHORTCUT arguments: w_bf16, in2_bf16 = raw cat copy, w2_bf16, bias, mask head).
__global__ __launch_bounds__(NTHREADS, 3) void dec6_fused_bf16_kernel(ConvArgs p, ConvArgs q) {
    // conv1's image and weight chunk are double-buffered (one barrier per chunk, the next chunk's units and weights arrive
    // behind this chunk's MFMAs).  Once conv1 is done conv2's weights (2 chunks) take over buffer 0 and the intermediate
    // image buffer 1: 47 KB per workgroup, three workgroups per CU (launch bounds: 168 registers)
    constexpr int BUF_U4 = IMG1_U4 + W_U4;  // one (image, weights) buffer
    static_assert(2 * W_U4 <= BUF_U4 && MID_U4 <= BUF_U4, "conv2's weights / the intermediate must fit one buffer each");
    __shared__ uint4 lds4[2 * BUF_U4 + (64 + 100 + 28) / 4];
    uint4* mid = lds4 + BUF_U4;
    uint4* w2 = lds4;
    float* tabs = reinterpret_cast<float*>(lds4 + 2 * BUF_U4);
    float* lds_es = tabs;        // conv1 epilogue (bn2 + FiLM) scale / shift
    float* lds_eh = tabs + 32;
    float* lds_mw = tabs + 64;   // after_conv weight [3][32] + bias [3]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, j = lane & 31;
    const int tiles_x = p.W / FPW, tiles = tiles_x * ((p.H + FPHT - 1) / FPHT);
    const unsigned tile = fused_tile_index();
    const int b = tile / (unsigned)tiles;
    const int bxy = tile - b * tiles;
    const int y0 = (bxy / tiles_x) * FPHT, x0 = (bxy % tiles_x) * FPW;
    const int HW = p.H * p.W;

    if (tid < 32) {
        lds_es[tid] = p.epi_scale[tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + tid];
    }
    if (tid < 99) lds_mw[tid] = tid < 96 ? q.mask_w[tid] : q.mask_b[tid - 96];

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
    const v4i32 w1_rs = make_rsrc_words(p.w_bf16, 4u * W_U4 * 16u);
    const v4i32 w2_rs = make_rsrc_words(q.w_bf16, 2u * W_U4 * 16u);

    // ---- this thread's units of the 12 x 36 input tile: one 16-byte unit = 8 channels of a pixel; a chunk = 2 octets --------
    const uint4* act = reinterpret_cast<const uint4*>(p.in_bf16) + (size_t)b * (p.Cin / 8) * HW;
    unsigned uoff[NPPI], okbits = 0;
#pragma unroll
    for (int k = 0; k < NPPI; ++k) {
        const int u = min(tid + k * NTHREADS, NPI - 1);
        const int gy = y0 + u / IPI - 2, gx = x0 + u % IPI - 2;
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        uoff[k] = (unsigned)(min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1));
        okbits |= (ok ? 1u : 0u) << k;
    }
    uint4 stage[2][2][NPPI];  // two chunks in flight: chunk c in set c & 1
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NPPI; ++k) stage[c & 1][o][k] = act[(size_t)(2 * c + o) * HW + uoff[k]];
    };
    load_chunk(0);
    load_chunk(1);

    int mir[3], mic[3];  // conv1's pixel tiles: as in enc1_fused_bf16_kernel
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

    constexpr int NCH = 4;  // 64 input channels
    auto dma_w1 = [&](int c) {  // weight chunk c -> buffer c & 1
        for (int piece = wave; piece < W_U4 / 64; piece += 4)
            lds_dma_16B(w1_rs, (unsigned)lane * 16u, (unsigned)((c * W_U4 + piece * 64) * 16),
                        lds0 + (unsigned)(((c & 1) * BUF_U4 + IMG1_U4 + piece * 64) * 16));
    };
    auto put_chunk = [&](int c) {  // the staged units of chunk c -> image buffer c & 1 (conv zero padding applied here)
        uint4* img = lds4 + (c & 1) * BUF_U4;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NPPI; ++k) {
                const int u = tid + k * NTHREADS;
                const bool ok = (okbits >> k) & 1u;
                if (u < NPI) img[o * NPI + u] = ok ? stage[c & 1][o][k] : uint4{0u, 0u, 0u, 0u};
            }
    };
    dma_w1(0);
    put_chunk(0);  // (waits for the units of chunk 0)
    load_chunk(2);
    wait_vmcnt<2 * NPPI>();  // weights of chunk 0: only the units of chunk 2 are younger (those of chunk 1 were requested first)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) dma_w1(c + 1);  // its buffer was last read in chunk c-1: every wave is past that (barrier below)
        if (c == NCH - 1)                // buffer 0 is free for good: conv2's weights move in behind the last chunk's MFMAs
            for (int piece = wave; piece < 2 * W_U4 / 64; piece += 4)
                lds_dma_16B(w2_rs, (unsigned)lane * 16u, (unsigned)piece * 1024u, lds0 + (unsigned)(piece * 1024));
        const uint4* img = lds4 + (c & 1) * BUF_U4;
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(img + IMG1_U4) + khalf * 32 + j;
        const bf16x8* ibase = reinterpret_cast<const bf16x8*>(img) + khalf * NPI;
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
        if (c + 1 < NCH) {
            put_chunk(c + 1);  // image buffer (c+1) & 1 was last read in chunk c-1; the units were requested two chunks ago
            if (c + 3 < NCH) load_chunk(c + 3);
            // this wave's weight pieces of chunk c+1 must have landed before the barrier; younger than them are only the
            // units requested just now (in-order return): with nothing requested, everything is waited for
            if (c + 3 < NCH) wait_vmcnt<2 * NPPI>(); else wait_vmcnt<0>();
            __syncthreads();
        }
    }

    __syncthreads();  // every wave is done with image buffer 1 (chunk 3): the intermediate image goes there
    // ---- shortcut operands (1x1 over the raw cat copy, resunet.py:163): requested here (they arrive behind the epilogue below), contracted behind conv2 -------------
    const uint4* raw = reinterpret_cast<const uint4*>(q.in2_bf16) + (size_t)b * (q.Cin2 / 8) * HW;
    const uint4* wsc = reinterpret_cast<const uint4*>(q.w2_bf16);  // [chunk][octet][cout]
    const int x = x0 + j;
    unsigned poff[2];
#pragma unroll
    for (int px = 0; px < 2; ++px) poff[px] = (unsigned)(min(y0 + wave * 2 + px, q.H - 1) * q.W + x);
    uint4 sa[NCH], sb[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        sa[c] = wsc[c * 64 + khalf * 32 + j];
#pragma unroll
        for (int px = 0; px < 2; ++px) sb[c][px] = raw[(size_t)(2 * c + khalf) * HW + poff[px]];
    }
    // accumulators start at the shortcut's bias (resunet.py:163: conv bias of the 1x1)
    f32x16 acc2[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bb = q.bias[(r & 3) + 8 * (r >> 2) + 4 * khalf];
        acc2[0][0][r] = bb;
        acc2[0][1][r] = bb;
    }
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
    wait_vmcnt<0>();  // conv2's weights, the shortcut operands
    __syncthreads();

    // ---- conv2 over the intermediate image ---------------------------------------------------------------------------------
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
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int px = 0; px < 2; ++px)
            acc2[0][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sa[c]), __builtin_bit_cast(bf16x8, sb[c][px]),
                                                                   acc2[0][px], 0, 0, 0);
    store_tile<1, 2, FPW, F_MASK, false>(q, acc2, nullptr, nullptr, nullptr, b, 0, y0, x0, lane, wave, lds_mw, nullptr);
}


// ---- decoder_block6 with its transposed conv INSIDE (resunet.py:240-264 at the shape of :408-418) --------------------------
// The up-sampled half of the concat (32 channels at the full resolution: 1.1 GB written as two copies and read back with its
// halo per step) never exists in HBM.  A workgroup forms it for its own 12 x 36 input tile from the 6 x 18 low-resolution
// pixels underneath: kernel = stride, so ConvTranspose2d is a pointwise GEMM  up[(co, a, bb)][pixel] = sum_ci Wt[ci][(co, a, bb)]
// x[ci][pixel]  (x = decoder_block5's output with this conv's BN+FiLM+leaky already applied, blocked bf16) - 16 MFMAs per wave
// with BOTH operands straight from global memory (a lane's fragment is one 16-byte unit of either).  An accumulator tile is one
// channel octet x 4 sub-pixels; the khalf pair swaps halves (as tconv_store), applies conv_block2.bn1 + FiLM + leaky and writes
// 16-byte units into the image buffers of conv1's chunks 0 and 1; chunks 2 and 3 (the encoder skip) come from the activated cat
// copy as before.
// The 1x1 shortcut over the up-sampled half is linear in x as well:  Wsc[:, :32] up = W' x  with the composed weights
// W'[(a, bb, n)][ci] = sum_co Wsc[n][co] Wt[ci][(co, a, bb)]  (lass_finalize, f32 products rounded to bf16 once).  Its
// accumulator tile for sub-pixel (a, bb) is [32 couts][low-res pixels], so conv2 runs in THAT layout: wave w owns sub-pixel
// class (a, bb) = (w >> 1, w & 1) of the 8 x 32 output tile, column tile t, lane j = low-res pixel (2t + (j >> 4), j & 15); the
// intermediate image is kept parity-split ([row][column parity][20]) so that a wave's B fragments are contiguous 16-byte units.
// p, q: as dec6_fused_bf16_kernel (p.pro_scale / pro_shift = conv_block2.bn1 of the concat channels);  u: the transposed conv
// (in_bf16 = x, w_bf16 = Wt [4][2][128] units, w2_bf16 = W' [4][2][128] units, H x W = the low resolution).
constexpr int LRI = IRI / 2, LPI = IPI / 2, NLP = LRI * LPI;  // low-resolution halo tile: 6 x 18 = 108 pixels
constexpr int MPH = 20, MPITCH = 2 * MPH;                      // parity-split intermediate row (17 of 20 units used per parity;
                                                               // two rows = 80 units = 0 mod 16: lanes 16-31 fall on the banks behind lanes 0-15)
constexpr int MROWS_U4 = IRM * MPITCH, MIDP_U4 = 4 * MROWS_U4;

__global__ __launch_bounds__(NTHREADS, 3) void dec6u_fused_bf16_kernel(ConvArgs p, ConvArgs q, ConvArgs u) {
    constexpr int BUF_U4 = IMG1_U4 + W_U4;  // one (image, weights) buffer
    static_assert(2 * W_U4 + MIDP_U4 <= 2 * BUF_U4 && 2 * W_U4 <= BUF_U4, "conv2's weights fit buffer 0; the intermediate starts behind them");
    __shared__ uint4 lds4[2 * BUF_U4 + 57];
    uint4* mid = lds4 + 2 * W_U4;  // behind conv2's weights; written only after conv1's last chunk
    uint4* w2 = lds4;
    float* tabs = reinterpret_cast<float*>(lds4 + 2 * BUF_U4);
    float* lds_es = tabs;        // conv1 epilogue (bn2 + FiLM) scale / shift
    float* lds_eh = tabs + 32;
    float* lds_mw = tabs + 64;   // after_conv weight [3][32] + bias [3]
    float* lds_us = tabs + 164;  // conv_block2.bn1 + FiLM of the up-sampled channels (concat channels 0..31)
    float* lds_uh = tabs + 196;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, j = lane & 31;
    const int tiles_x = p.W / FPW, tiles = tiles_x * ((p.H + FPHT - 1) / FPHT);
    const unsigned tile = fused_tile_index();
    const int b = tile / (unsigned)tiles;
    const int bxy = tile - b * tiles;
    const int y0 = (bxy / tiles_x) * FPHT, x0 = (bxy % tiles_x) * FPW;
    const int HW = p.H * p.W;
    const int lh = u.H, lw = u.W, lhw = lh * lw;

    if (tid < 32) {
        lds_es[tid] = p.epi_scale[tid];
        lds_eh[tid] = p.epi_shift[(size_t)b * p.epi_shift_bs + tid];
        lds_us[tid] = p.pro_scale[tid];
        lds_uh[tid] = p.pro_shift[(size_t)b * p.pro_shift_bs + tid];
    }
    if (tid < 99) lds_mw[tid] = tid < 96 ? q.mask_w[tid] : q.mask_b[tid - 96];

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds4;
    const v4i32 w1_rs = make_rsrc_words(p.w_bf16, 4u * W_U4 * 16u);
    const v4i32 w2_rs = make_rsrc_words(q.w_bf16, 2u * W_U4 * 16u);
    constexpr int NCH = 4;  // 64 concat channels
    auto dma_w1 = [&](int c) {  // conv1's weight chunk c -> buffer c & 1
        for (int piece = wave; piece < W_U4 / 64; piece += 4)
            lds_dma_16B(w1_rs, (unsigned)lane * 16u, (unsigned)((c * W_U4 + piece * 64) * 16),
                        lds0 + (unsigned)(((c & 1) * BUF_U4 + IMG1_U4 + piece * 64) * 16));
    };
    dma_w1(0);
    dma_w1(1);

    // ---- phase 0: the transposed conv of this tile's 6 x 18 low-resolution pixels -> conv1's image chunks 0 and 1 ------------
    const uint4* xa = reinterpret_cast<const uint4*>(u.in_bf16) + (size_t)b * (u.Cin / 8) * lhw;
    const uint4* wt = reinterpret_cast<const uint4*>(u.w_bf16);
    const int qpx = wave * 32 + j;  // this lane's low-resolution pixel (MFMA column) in the 6 x 18 tile
    const bool qvalid = qpx < NLP;
    const int qq = qvalid ? qpx : NLP - 1;
    const int ly = qq / LPI, lx = qq - ly * LPI;
    const int gly = (y0 >> 1) - 1 + ly, glx = (x0 >> 1) - 1 + lx;
    const bool qin = qvalid && gly >= 0 && gly < lh && glx >= 0 && glx < lw;  // outside the image: conv1's zero padding
    const unsigned xoff = (unsigned)(min(max(gly, 0), lh - 1) * lw + min(max(glx, 0), lw - 1));
    // every operand of the 16 MFMAs is requested up front (80 registers, free at this point of the kernel): one exposed
    // round trip instead of one per accumulator tile (hipcc sinks the loads to their uses otherwise)
    uint4 xb[4], wa[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[s] = xa[(size_t)(2 * s + khalf) * lhw + xoff];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) wa[m][s] = wt[(s * 2 + khalf) * 128 + 32 * m + j];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // tables visible
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // accumulator tile m = concat channels 8m .. 8m+7 x 4 sub-pixels
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[m][s]), __builtin_bit_cast(bf16x8, xb[s]), acc, 0, 0, 0);
        // register r = row (r & 3) + 8 (r >> 2) + 4 khalf = channel 2 (r >> 2) + khalf, sub-pixel r & 3 = (a, bb).  One
        // v_permlane32_swap per (channel pair, bb) hands lane khalf output row a = khalf with all 8 channels (two adjacent
        // units, bb = 0, 1): it swaps lanes 32-63 of its first operand (a = 0 values of the odd channel) with lanes 0-31 of
        // the second (a = 1 values of the even channel)
        float ch[2][8];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float a0 = acc[4 * g + bb], a1 = acc[4 * g + 2 + bb];  // (named floats: hipcc's bit_cast of a vector ELEMENT reads element 0)
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1), false, false);
                const unsigned r0 = sw[0], r1 = sw[1];
                ch[bb][2 * g] = __builtin_bit_cast(float, r0);
                ch[bb][2 * g + 1] = __builtin_bit_cast(float, r1);
            }
        float us[8], uh[8];
        {
            const float4 a0 = *reinterpret_cast<const float4*>(lds_us + 8 * m), a1 = *reinterpret_cast<const float4*>(lds_us + 8 * m + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(lds_uh + 8 * m), h1 = *reinterpret_cast<const float4*>(lds_uh + 8 * m + 4);
            us[0] = a0.x; us[1] = a0.y; us[2] = a0.z; us[3] = a0.w; us[4] = a1.x; us[5] = a1.y; us[6] = a1.z; us[7] = a1.w;
            uh[0] = h0.x; uh[1] = h0.y; uh[2] = h0.z; uh[3] = h0.w; uh[4] = h1.x; uh[5] = h1.y; uh[6] = h1.z; uh[7] = h1.w;
        }
        uint4* img = lds4 + (m >> 1) * BUF_U4 + (m & 1) * NPI + (2 * ly + khalf) * IPI + 2 * lx;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = leaky(ch[bb][k] * us[k] + uh[k]);  // conv_block2.bn1 + FiLM + leaky (resunet.py:150)
                v[k] = (__bf16)(qin ? t : 0.f);
            }
            if (qvalid) *reinterpret_cast<bf16x8*>(img + bb) = v;
        }
    }

    // ---- the encoder skip (concat channels 32..63 = chunks 2, 3): this thread's units of the 12 x 36 tile, staged in registers
    const uint4* act = reinterpret_cast<const uint4*>(p.in_bf16) + (size_t)b * (p.Cin / 8) * HW;
    unsigned uoff[NPPI], okbits = 0;
#pragma unroll
    for (int k = 0; k < NPPI; ++k) {
        const int un = min(tid + k * NTHREADS, NPI - 1);
        const int gy = y0 + un / IPI - 2, gx = x0 + un % IPI - 2;
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        uoff[k] = (unsigned)(min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1));
        okbits |= (ok ? 1u : 0u) << k;
    }
    uint4 stage[2][2][NPPI];  // chunk c in set c & 1
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NPPI; ++k) stage[c & 1][o][k] = act[(size_t)(2 * c + o) * HW + uoff[k]];
    };
    auto put_chunk = [&](int c) {  // the staged units of chunk c -> image buffer c & 1 (conv zero padding applied here)
        uint4* img = lds4 + (c & 1) * BUF_U4;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < NPPI; ++k) {
                const int un = tid + k * NTHREADS;
                const bool ok = (okbits >> k) & 1u;
                if (un < NPI) img[o * NPI + un] = ok ? stage[c & 1][o][k] : uint4{0u, 0u, 0u, 0u};
            }
    };
    load_chunk(2);
    load_chunk(3);

    int mir[3], mic[3];  // conv1's pixel tiles: as in enc1_fused_bf16_kernel
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

    // conv2 / shortcut pixel classes (see the header): sub-pixel (spa, spb) = wave, column tile t, lane j
    const int spa = wave >> 1, spb = wave & 1;
    int oy[2];
    const int ox = 2 * (j & 15) + spb;
#pragma unroll
    for (int t = 0; t < 2; ++t) oy[t] = 2 * (2 * t + (j >> 4)) + spa;
    uint4 xs[2][4], au[4];  // shortcut over the up-sampled half: W' (rows of this wave's sub-pixel) and x at the inner 4 x 16 pixels

    wait_vmcnt<2 * 2 * NPPI>();  // conv1's weight chunks 0 and 1 have landed (only the skip units are younger)
    __syncthreads();             // ... and everyone's image units of chunks 0 and 1 are written
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c >= 1 && c + 1 < NCH) dma_w1(c + 1);  // its buffer was last read in chunk c-1: every wave is past that (barrier below)
        if (c == NCH - 1) {              // buffer 0 is free for good: conv2's weights move in behind the last chunk's MFMAs
            for (int piece = wave; piece < 2 * W_U4 / 64; piece += 4)
                lds_dma_16B(w2_rs, (unsigned)lane * 16u, (unsigned)piece * 1024u, lds0 + (unsigned)(piece * 1024));
            const uint4* wsu = reinterpret_cast<const uint4*>(u.w2_bf16);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                au[s] = wsu[(s * 2 + khalf) * 128 + wave * 32 + j];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    xs[t][s] = xa[(size_t)(2 * s + khalf) * lhw + (size_t)min((y0 >> 1) + 2 * t + (j >> 4), lh - 1) * lw + (x0 >> 1) + (j & 15)];
            }
        }
        const uint4* img = lds4 + (c & 1) * BUF_U4;
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(img + IMG1_U4) + khalf * 32 + j;
        const bf16x8* ibase = reinterpret_cast<const bf16x8*>(img) + khalf * NPI;
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
        if (c + 1 < NCH) {
            if (c + 1 >= 2) put_chunk(c + 1);  // image buffer (c+1) & 1 was last read in chunk c-1
            wait_vmcnt<0>();                   // this wave's weight pieces of chunk c+1
            __syncthreads();
        }
    }
    // accumulators of conv2 start at the shortcut's bias (resunet.py:163) + the shortcut over the up-sampled half
    f32x16 acc2[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bb = q.bias[(r & 3) + 8 * (r >> 2) + 4 * khalf];
        acc2[0][r] = bb;
        acc2[1][r] = bb;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au[s]), __builtin_bit_cast(bf16x8, xs[t][s]), acc2[t], 0, 0, 0);

    __syncthreads();  // every wave is done with buffer 1 (chunk 3): the intermediate image goes there
    // ---- shortcut over the skip half (1x1 over the RAW cat copy, resunet.py:163): requested here, contracted behind conv2 ------
    const uint4* raw = reinterpret_cast<const uint4*>(q.in2_bf16) + (size_t)b * (q.Cin2 / 8) * HW;
    const uint4* wsc = reinterpret_cast<const uint4*>(q.w2_bf16);  // [chunk][octet][cout]
    uint4 sa[2], sb[2][2];
#pragma unroll
    for (int c = 2; c < NCH; ++c) {
        sa[c - 2] = wsc[c * 64 + khalf * 32 + j];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            sb[c - 2][t] = raw[(size_t)(2 * c + khalf) * HW + (size_t)min(y0 + oy[t], q.H - 1) * q.W + x0 + ox];
    }
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
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(mid + g * MROWS_U4 + ir * MPITCH + (ic & 1) * MPH + (ic >> 1)) + khalf * 8) = v;
                }
            }
        }
    }
    wait_vmcnt<0>();  // conv2's weights, the shortcut operands
    __syncthreads();

    // ---- conv2 over the intermediate image, sub-pixel class (spa, spb) ----------------------------------------------------------
    int lbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) lbase[t] = 2 * (2 * t + (j >> 4)) * MPITCH + (j & 15);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bf16x8* bbase = reinterpret_cast<const bf16x8*>(mid) + (2 * c + khalf) * MROWS_U4;
        const bf16x8* abase = reinterpret_cast<const bf16x8*>(w2 + c * W_U4) + khalf * 32 + j;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16x8 a = abase[tap * 64];
            const int col = spb + tap % 3;  // intermediate column = 2 lx + col, row = 2 ly + spa + tap / 3
            const int off = (spa + tap / 3) * MPITCH + (col & 1) * MPH + (col >> 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bbase[lbase[t] + off], acc2[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sa[c]), __builtin_bit_cast(bf16x8, sb[c][t]), acc2[t], 0, 0, 0);

    // ---- after_conv + complex ratio mask (resunet.py:570-574,436-519): the khalf pair holds all 32 channels of a pixel; lane
    // khalf finishes the pixel of column tile t = khalf
    float lg[3];
#pragma unroll
    for (int qd = 0; qd < 3; ++qd) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float wv = lds_mw[qd * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf];
            s0 += wv * acc2[0][r];
            s1 += wv * acc2[1][r];
        }
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        lg[qd] = (khalf ? s1 : s0) + lds_mw[96 + qd];
    }
    const int yy = y0 + (khalf ? oy[1] : oy[0]);
    if (yy < q.mask_T) mask_pixel(q, b, yy, x0 + ox, lg[0], lg[1], lg[2]);
}


// W'[(sp, n)][ci] = sum_co Wsc[n][co] * Wt[ci][co][sp]  (sp = a * 2 + bb): the 1x1 shortcut of decoder_block6's ConvBlockRes
// composed with the transposed conv in front of it (both linear; resunet.py:122-128,163 and :216-224,250), f32.
// wsc (Nsc, Ccat, 1, 1) - only its first Cup input channels (the up-sampled half of torch.cat((x, skip), 1)); wt (Cin, Cup, 2, 2).
__global__ __launch_bounds__(256) void compose_up_shortcut_kernel(const float* __restrict__ wsc, const float* __restrict__ wt,
                                                                  int Cin, int Cup, int Ccat, int Nsc, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // ((sp * Nsc) + n) * Cin + ci
    if (i >= 4 * Nsc * Cin) return;
    const int ci = i % Cin, n = (i / Cin) % Nsc, sp = i / (Cin * Nsc);
    double s = 0.0;
    for (int co = 0; co < Cup; ++co) s += (double)wsc[(size_t)n * Ccat + co] * (double)wt[((size_t)ci * Cup + co) * 4 + sp];
    out[i] = (float)s;
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

bool lass_dec6_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q) {
    return p.Cin == 64 && p.N == 32 && p.Nw == 32 && q.Cin == 32 && q.N == 32 && q.Nw == 32 && q.Cin2 == 64 && p.W % FPW == 0 &&
           p.W >= FPW && p.in_bf16 && !p.in_bf16_lo && p.w_bf16 && q.w_bf16 && !p.w_bf16_lo && !q.w_bf16_lo && p.epi_scale &&
           p.epi_shift && q.in2_bf16 && q.w2_bf16 && !q.w2_bf16_lo && q.bias && q.mask_re && q.mask_im && q.mask_w && q.mask_b &&
           q.mask_mag && q.mask_cos && q.mask_sin && q.W + 1 == q.mask_nbins && q.mask_T > 0 && q.mask_T <= q.H && !q.pool_out &&
           !q.pool_bf16 && q.H == p.H && q.W == p.W && q.B == p.B && (unsigned long long)p.H * p.W * 8ull < 0x10000000ull;
}

hipError_t lass_launch_dec6_fused_bf16(const ConvArgs& p, const ConvArgs& q, hipStream_t stream) {
    if (!lass_dec6_fused_bf16_supported(p, q)) return hipErrorInvalidValue;
    const long nblk = (long)(p.W / FPW) * ((p.H + FPHT - 1) / FPHT) * p.B;
    hipLaunchKernelGGL(dec6_fused_bf16_kernel, dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, p, q);
    return hipGetLastError();
}

bool lass_dec6u_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q, const ConvArgs& u) {
    return lass_dec6_fused_bf16_supported(p, q) && p.pro_scale && p.pro_shift && u.in_bf16 && u.w_bf16 && u.w2_bf16 && u.Cin == 64 &&
           u.H * 2 == p.H && u.W * 2 == p.W && u.B == p.B && p.H % 2 == 0 && (unsigned long long)u.H * u.W * 8ull < 0x10000000ull;
}

hipError_t lass_launch_dec6u_fused_bf16(const ConvArgs& p, const ConvArgs& q, const ConvArgs& u, hipStream_t stream) {
    if (!lass_dec6u_fused_bf16_supported(p, q, u)) return hipErrorInvalidValue;
    const long nblk = (long)(p.W / FPW) * ((p.H + FPHT - 1) / FPHT) * p.B;
    hipLaunchKernelGGL(dec6u_fused_bf16_kernel, dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, p, q, u);
    return hipGetLastError();
}

hipError_t lass_launch_compose_up_shortcut(const float* wsc, const float* wt, int Cin, int Cup, int Ccat, int Nsc, float* out,
                                           hipStream_t stream) {
    const int n = 4 * Nsc * Cin;
    hipLaunchKernelGGL(compose_up_shortcut_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, wsc, wt, Cin, Cup, Ccat, Nsc, out);
    return hipGetLastError();
}
