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
// [cin][row][col]; both fragments are single conflict-free ds_read_b32 per MFMA operand.  The f32 MFMA is a
// bit-exact f32 FMA chain, so results differ from the reference only by summation order.
//
// Fusions: BN(eval)+FiLM+leaky-ReLU as a prologue while staging the halo tile (zero padding is applied AFTER the
// activation, as conv padding does), the block's second activation as conv1's epilogue, residual add / 1x1 shortcut
// conv (+bias) inside conv2, transposed-conv scatter, channel-slice ("virtual concat") output via batch strides.
#include <hip/hip_runtime.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int F_PRO = 1;     // prologue: x*scale[c] + shift[b][c], leaky 0.01
constexpr int F_PHASEB = 2;  // second K-phase: 1x1 over raw in2 (shortcut conv)
constexpr int F_BIAS = 4;    // + bias[n]
constexpr int F_RES = 8;     // + res[b][n][y][x]
constexpr int F_EPIACT = 16; // epilogue: leaky(v*scale[n] + shift[b][n])
constexpr int F_TCONV = 32;  // n = (co, a, bb); scatter to (y*uh+a, x*uw+bb)

constexpr int MAX_CIN = 768;
constexpr int NTHREADS = 256;

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.01f * v; }

// One K-phase: stages [KC] channels x (rows+halo) x (cols+halo) of input and [KC][TAPS][NT] of weights per chunk,
// register-prefetching chunk c+1 while chunk c is contracted out of LDS.
template <int TAPS, int KC, int NCO, int NPX, int PW, bool PRO>
struct Phase {
    static constexpr int PH = 32 / PW;
    static constexpr int WROWS = NPX * PH;
    static constexpr int PHT = 4 * WROWS;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int IR = PHT + 2 * HALO;
    static constexpr int IP = PW + 2 * HALO;
    static constexpr int NT = 32 * NCO;
    static constexpr int IN_ELEMS = KC * IR * IP;
    static constexpr int NLD = (IN_ELEMS + NTHREADS - 1) / NTHREADS;
    static constexpr int W_V4 = KC * TAPS * NT / 4;
    static constexpr int NWLD = (W_V4 + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_FLOATS = IN_ELEMS + KC * TAPS * NT;
    static_assert((IN_ELEMS % 4) == 0, "weight region must stay 16-B aligned");

    int goff[NLD];
    float v[NLD];
    float4 wv[NWLD];

    __device__ __forceinline__ void init(int tid, int y0, int x0, int H, int W) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NTHREADS;
            const int c = e / (IR * IP);
            const int r = (e / IP) % IR;
            const int x = e % IP;
            const int gy = y0 + r - HALO, gx = x0 + x - HALO;
            const bool ok = (e < IN_ELEMS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            goff[i] = ok ? (c * H * W + gy * W + gx) : -1;
        }
    }

    // in_c0: pointer to channel c0 of this clip; w_c0: pointer to Wt[c0][0][n0]
    __device__ __forceinline__ void load(const float* __restrict__ in_c0, const float* __restrict__ w_c0, int Nw,
                                         int tid) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = goff[i] >= 0 ? in_c0[goff[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < NWLD; ++i) {
            const int e = tid + i * NTHREADS;  // float4 index into [KC*TAPS][NT/4]
            if (e < W_V4) {
                const int row = e / (NT / 4), col = e % (NT / 4);
                wv[i] = *reinterpret_cast<const float4*>(w_c0 + (size_t)row * Nw + col * 4);
            }
        }
    }

    __device__ __forceinline__ void store(float* lds, const float* lds_sc, const float* lds_sh, int c0, int tid) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NTHREADS;
            if (e < IN_ELEMS) {
                float t = v[i];
                if (PRO) {
                    const int c = c0 + e / (IR * IP);
                    t = goff[i] >= 0 ? leaky(t * lds_sc[c] + lds_sh[c]) : 0.f;
                }
                lds[e] = t;
            }
        }
        float4* lw = reinterpret_cast<float4*>(lds + IN_ELEMS);
#pragma unroll
        for (int i = 0; i < NWLD; ++i) {
            const int e = tid + i * NTHREADS;
            if (e < W_V4) lw[e] = wv[i];
        }
    }

    __device__ __forceinline__ static void compute(const float* lds, f32x16 (&acc)[NCO][NPX], int lane, int wave) {
        const int khalf = lane >> 5, j = lane & 31;
        const int ty = j / PW, tx = j % PW;
        const float* bbase = lds + khalf * (IR * IP) + (wave * WROWS + ty) * IP + tx;
        const float* abase = lds + IN_ELEMS + khalf * (TAPS * NT) + j;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                float a[NCO], b[NPX];
#pragma unroll
                for (int co = 0; co < NCO; ++co) a[co] = abase[(kk * 2 * TAPS + tap) * NT + co * 32];
#pragma unroll
                for (int px = 0; px < NPX; ++px)
                    b[px] = bbase[kk * 2 * (IR * IP) + (px * PH + (TAPS == 9 ? tap / 3 : 0)) * IP +
                                  (TAPS == 9 ? tap % 3 : 0)];
#pragma unroll
                for (int co = 0; co < NCO; ++co)
#pragma unroll
                    for (int px = 0; px < NPX; ++px)
                        acc[co][px] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[co], b[px], acc[co][px], 0, 0, 0);
            }
        }
    }

    // Run the whole phase.  `in_b`: clip base of the input tensor; Wt: [Cin][TAPS][Nw] (+n0 applied here).
    __device__ __forceinline__ void run(float* lds, const float* lds_sc, const float* lds_sh,
                                        const float* __restrict__ in_b, int Cin, int HW,
                                        const float* __restrict__ Wt, int Nw, int n0, f32x16 (&acc)[NCO][NPX],
                                        int tid, int y0, int x0, int H, int W) {
        const int lane = tid & 63, wave = tid >> 6;
        init(tid, y0, x0, H, W);
        const int nchunks = Cin / KC;
        load(in_b, Wt + n0, Nw, tid);
        __syncthreads();  // previous phase's LDS reads (and the prologue tables) are complete
        store(lds, lds_sc, lds_sh, 0, tid);
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            const bool more = ch + 1 < nchunks;
            if (more)
                load(in_b + (size_t)(ch + 1) * KC * HW, Wt + (size_t)(ch + 1) * KC * TAPS * Nw + n0, Nw, tid);
            compute(lds, acc, lane, wave);
            __syncthreads();
            if (more) store(lds, lds_sc, lds_sh, (ch + 1) * KC, tid);
            __syncthreads();
        }
    }
};

template <int A, int B>
struct MaxI {
    static constexpr int v = A > B ? A : B;
};

template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
__global__ __launch_bounds__(NTHREADS) void conv_kernel(ConvArgs p) {
    constexpr bool PRO = (FLAGS & F_PRO) != 0;
    constexpr int KCA = (TAPS == 9) ? 8 : 16;
    using PA = Phase<TAPS, KCA, NCO, NPX, PW, PRO>;
    using PB = Phase<1, 16, NCO, NPX, PW, false>;
    constexpr int LDS_MAIN = (FLAGS & F_PHASEB) ? MaxI<PA::LDS_FLOATS, PB::LDS_FLOATS>::v : PA::LDS_FLOATS;
    constexpr int PH = PA::PH, WROWS = PA::WROWS, PHT = PA::PHT, NT = PA::NT;

    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN + (PRO ? 2 * MAX_CIN : 0)];
    float* lds_sc = lds + LDS_MAIN;
    float* lds_sh = lds_sc + MAX_CIN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int n0 = blockIdx.y * NT;
    const int tiles_x = p.W / PW;
    const int y0 = (blockIdx.x / tiles_x) * PHT;
    const int x0 = (blockIdx.x % tiles_x) * PW;
    const int HW = p.H * p.W;

    f32x16 acc[NCO][NPX];
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int px = 0; px < NPX; ++px)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[co][px][r] = 0.f;

    if (PRO) {
        for (int c = tid; c < p.Cin; c += NTHREADS) {
            lds_sc[c] = p.pro_scale[c];
            lds_sh[c] = p.pro_shift[(size_t)b * p.pro_shift_bs + c];
        }
    }
    {
        PA ph;
        ph.run(lds, lds_sc, lds_sh, p.in + (size_t)b * p.in_bs, p.Cin, HW, p.w, p.Nw, n0, acc, tid, y0, x0, p.H, p.W);
    }
    if (FLAGS & F_PHASEB) {
        PB ph;
        ph.run(lds, nullptr, nullptr, p.in2 + (size_t)b * p.in2_bs, p.Cin2, HW, p.w2, p.Nw, n0, acc, tid, y0, x0,
               p.H, p.W);
    }

    // ---- epilogue: D row (register) = output channel, D col (lane&31) = pixel -----------------------------------
    const int khalf = lane >> 5, j = lane & 31;
    const int ty = j / PW, tx = j % PW;
    const int x = x0 + tx;
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const int y = y0 + wave * WROWS + px * PH + ty;
            if (y >= p.H) continue;
            if (FLAGS & F_TCONV) {
                // n = co_real*(uh*uw) + a*uw + bb, uw == 2: registers (r, r+1), r even, are bb = 0/1 of one (co_real, a)
                const int uhw = p.up_h * 2;
                const size_t oHW = (size_t)HW * uhw;
                const int oW = p.W * 2;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int n = n0 + co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int co_real = n / uhw, a = (n % uhw) >> 1;
                    float2 o = make_float2(acc[co][px][r], acc[co][px][r + 1]);
                    float* dst = p.out + (size_t)b * p.out_bs + co_real * oHW + (size_t)(y * p.up_h + a) * oW + x * 2;
                    *reinterpret_cast<float2*>(dst) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + co * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v = acc[co][px][r];
                    const size_t pix = (size_t)n * HW + (size_t)y * p.W + x;
                    if (FLAGS & F_BIAS) v += p.bias[n];
                    if (FLAGS & F_RES) v += p.res[(size_t)b * p.res_bs + pix];
                    if (FLAGS & F_EPIACT) v = leaky(v * p.epi_scale[n] + p.epi_shift[(size_t)b * p.epi_shift_bs + n]);
                    p.out[(size_t)b * p.out_bs + pix] = v;
                }
            }
        }
    }
}

template <int TAPS, int NCO, int NPX, int PW, int FLAGS>
hipError_t launch_one(const ConvArgs& p, hipStream_t stream) {
    constexpr int PHT = 4 * NPX * (32 / PW);
    dim3 grid((p.W / PW) * ((p.H + PHT - 1) / PHT), p.N / (32 * NCO), p.B);
    hipLaunchKernelGGL((conv_kernel<TAPS, NCO, NPX, PW, FLAGS>), grid, dim3(NTHREADS), 0, stream, p);
    return hipGetLastError();
}

template <int TAPS, int FLAGS>
hipError_t launch_geom(const ConvArgs& p, hipStream_t stream) {
    const int pw = p.W >= 32 ? 32 : p.W;
    if (pw == 32) {
        if (p.N % 64 == 0) return launch_one<TAPS, 2, 2, 32, FLAGS>(p, stream);
        return launch_one<TAPS, 1, 2, 32, FLAGS>(p, stream);
    }
    if (p.N % 64 != 0) return hipErrorInvalidValue;
    if (pw == 16) return launch_one<TAPS, 2, 2, 16, FLAGS>(p, stream);
    if (pw == 8) return launch_one<TAPS, 2, 2, 8, FLAGS>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

// Host-side shape validation shared by all conv launches: every assumption the kernel and its grid make.
static bool conv_args_ok(const ConvArgs& p, int taps, bool phaseb) {
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return false;
    if (p.W != 8 && p.W != 16 && (p.W % 32) != 0) return false;
    if (p.N % 32 != 0 || p.Nw < p.N || (p.Nw % 4) != 0) return false;
    const int kc = taps == 9 ? 8 : 16;
    if (p.Cin <= 0 || p.Cin % kc != 0 || p.Cin > MAX_CIN) return false;
    if (phaseb && (p.Cin2 <= 0 || p.Cin2 % 16 != 0)) return false;
    if (!p.in || !p.w || !p.out) return false;
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
            return launch_geom<9, F_PHASEB | F_BIAS>(p, stream);
        case TCONV_ACT:  // kernel==stride transposed conv with prologue act
            if (!conv_args_ok(p, 1, false) || !p.pro_scale || !p.pro_shift || (p.up_h != 1 && p.up_h != 2))
                return hipErrorInvalidValue;
            return launch_geom<1, F_PRO | F_TCONV>(p, stream);
    }
    return hipErrorInvalidValue;
}
