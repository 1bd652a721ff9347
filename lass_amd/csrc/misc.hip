// misc.hip - the small HBM-bound kernels around the convolutions.
//
// Replaces (reference: /root/reference):
//   FiLM.forward                 models/resunet.py:59-81   (38 nn.Linear -> ONE pass over a concatenated matrix)
//   pre_conv                     models/resunet.py:555
//   F.avg_pool2d                 models/resunet.py:197
//   after_conv + feature_maps_to_wav (mask part)  models/resunet.py:570-574, :469-495
//   calculate_sdr / calculate_sisdr reductions    utils.py:148-200
//   BatchNorm2d eval folding     models/resunet.py:98-99,159-160  (weight preparation, one-off)
#include <hip/hip_runtime.h>
#include <float.h>
#include "kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per output row j: the 512 weights of the row live in 8 registers per lane; every clip's condition is
// dotted against them with a 64-lane shuffle reduction.
__global__ __launch_bounds__(256) void film_kernel(const float* __restrict__ cond, int B, const float* __restrict__ Wf,
                                                   const float* __restrict__ bf, const float* __restrict__ base, int n,
                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    float w[LASS_COND / 64];
#pragma unroll
    for (int i = 0; i < LASS_COND / 64; ++i) w[i] = Wf[(size_t)j * LASS_COND + lane + 64 * i];
    const float add = bf[j] + (base ? base[j] : 0.f);
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LASS_COND / 64; ++i) s += w[i] * cond[(size_t)b * LASS_COND + lane + 64 * i];
        s = wave_sum(s);
        if (lane == 0) out[(size_t)b * n + j] = s + add;
    }
}

__global__ __launch_bounds__(256) void preconv_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int C, long HW,
                                                      float* __restrict__ out) {
    const int b = blockIdx.z, c = blockIdx.y;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= HW) return;
    const float4 v = *reinterpret_cast<const float4*>(x0 + (size_t)b * HW + i);
    const float wc = w[c], bc = bias[c];
    float4 o = make_float4(v.x * wc + bc, v.y * wc + bc, v.z * wc + bc, v.w * wc + bc);
    *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * HW + i) = o;
}

template <int PH>
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ in, long in_bs, int C, int H, int W,
                                                   float* __restrict__ out) {
    // one thread per output pixel pair-row; horizontal factor is always 2
    const int Ho = H / PH, Wo = W / 2;
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ho * Wo) return;
    const int yo = i / Wo, xo = i % Wo;
    const float* src = in + (size_t)b * in_bs + (size_t)c * H * W + (size_t)(yo * PH) * W + xo * 2;
    const float2 r0 = *reinterpret_cast<const float2*>(src);
    float s = r0.x + r0.y;  // sequential row-major accumulation, as ATen's NCHW avg_pool2d kernel does
    if (PH == 2) {
        const float2 r1 = *reinterpret_cast<const float2*>(src + W);
        s += r1.x;
        s += r1.y;
    }
    out[((size_t)b * C + c) * Ho * Wo + i] = s / (float)(PH * 2);
}

// after_conv (1x1, 32 -> 3, +bias) fused with the complex ratio mask.  One block per (frame, clip); thread f handles
// bins f and f+256, thread 0 also writes the Nyquist bin, whose logits are the zero padding of resunet.py:573 so the
// output is exactly 0 (cos = sin = 0 from magphase's clamp).
__global__ __launch_bounds__(256) void mask_kernel(const float* __restrict__ x12, const float* __restrict__ wa,
                                                   const float* __restrict__ ba, const float* __restrict__ mag,
                                                   const float* __restrict__ cosv, const float* __restrict__ sinv,
                                                   int T, int Tpad, int fcrop, float* __restrict__ out_real,
                                                   float* __restrict__ out_imag) {
    __shared__ float sw[3 * 32 + 3];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < 99) sw[tid] = tid < 96 ? wa[tid] : ba[tid - 96];
    __syncthreads();
    const size_t plane = (size_t)Tpad * fcrop;
    const float* xb = x12 + (size_t)b * 32 * plane + (size_t)t * fcrop;
    const size_t row = ((size_t)b * T + t) * (fcrop + 1);
    for (int f = tid; f < fcrop; f += 256) {
        float l0 = sw[96], l1 = sw[97], l2 = sw[98];
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float v = xb[(size_t)c * plane + f];
            l0 += sw[c] * v;
            l1 += sw[32 + c] * v;
            l2 += sw[64 + c] * v;
        }
        const float mask_mag = 1.f / (1.f + expf(-l0));
        const float mr = tanhf(l1), mi = tanhf(l2);
        const float mm = sqrtf(mr * mr + mi * mi);
        const float den = fmaxf(mm, 1e-10f);  // torchlibrosa magphase: clamp on |M|
        const float mc = mr / den, ms = mi / den;
        const float ci = cosv[row + f], si = sinv[row + f];
        const float oc = ci * mc - si * ms;
        const float os = si * mc + ci * ms;
        const float om = fmaxf(mag[row + f] * mask_mag, 0.f);
        out_real[row + f] = om * oc;
        out_imag[row + f] = om * os;
    }
    if (tid == 0) {
        out_real[row + fcrop] = 0.f;
        out_imag[row + fcrop] = 0.f;
    }
}

__global__ __launch_bounds__(256) void sdr_pass1(const float* __restrict__ ref, const float* __restrict__ est, int L,
                                                 double* __restrict__ stats) {
    __shared__ double red[4][4];
    const int b = blockIdx.y;
    const float* r = ref + (size_t)b * L;
    const float* e = est + (size_t)b * L;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const double rv = r[i], ev = e[i], d = ev - rv;
        s0 += rv * rv;
        s1 += ev * ev;
        s2 += rv * ev;
        s3 += d * d;
    }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); s3 = wave_sum_d(s3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(&stats[(size_t)b * 6 + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(256) void sdr_pass2(const float* __restrict__ ref, const float* __restrict__ est, int L,
                                                 double* __restrict__ stats) {
    __shared__ double red[4][2];
    const int b = blockIdx.y;
    const float* r = ref + (size_t)b * L;
    const float* e = est + (size_t)b * L;
    const double eps = (double)FLT_EPSILON;  // np.finfo(float32).eps, utils.py:180
    const double a = (eps + stats[(size_t)b * 6 + 2]) / (stats[(size_t)b * 6 + 0] + eps);
    double s0 = 0, s1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const double tr = a * (double)r[i], d = (double)e[i] - tr;
        s0 += tr * tr;
        s1 += d * d;
    }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(&stats[(size_t)b * 6 + 4 + threadIdx.x], v);
    }
}

// ---- mixture creation at a given SNR + declipping (dcase_evaluator.py:77-89), device-resident ----------------------
// pass 1: ws[b] = {sum source^2, sum noise^2, (max |mixture| as float bits), unused}
__global__ __launch_bounds__(256) void mix_power_kernel(const float* __restrict__ src, const float* __restrict__ noise,
                                                        int L, double* __restrict__ ws) {
    __shared__ double red[4][2];
    const int b = blockIdx.y;
    const float* s = src + (size_t)b * L;
    const float* n = noise + (size_t)b * L;
    double s0 = 0, s1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const double sv = s[i], nv = n[i];
        s0 += sv * sv;
        s1 += nv * nv;
    }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(&ws[(size_t)b * 4 + threadIdx.x],
                  red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// noise scaling factor sqrt((P_source / 10^(snr/10)) / P_noise) (the 1/L of the two means cancels)
__device__ __forceinline__ float mix_scale(const double* ws, int b, float snr_db) {
    return (float)sqrt(ws[(size_t)b * 4] / pow(10.0, (double)snr_db / 10.0) / ws[(size_t)b * 4 + 1]);
}

// pass 2: mixture = source + noise * scale; per-clip max |mixture| (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void mix_apply_kernel(const float* __restrict__ src, const float* __restrict__ noise,
                                                        const float* __restrict__ snr_db, int L,
                                                        double* __restrict__ ws, float* __restrict__ mix) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float sc = mix_scale(ws, b, snr_db[b]);
    float mx = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const float nv = noise[(size_t)b * L + i] * sc;  // numpy: noise = noise * scaling_factor (f32), then the add
        const float m = src[(size_t)b * L + i] + nv;
        mix[(size_t)b * L + i] = m;
        mx = fmaxf(mx, fabsf(m));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(reinterpret_cast<unsigned int*>(&ws[(size_t)b * 4 + 2]), __float_as_uint(mx));
    }
}

// pass 3: declipping - if max |mixture| > 1, source and mixture are both scaled by 0.9 / max (dcase_evaluator.py:86-89)
__global__ __launch_bounds__(256) void mix_declip_kernel(float* __restrict__ src, float* __restrict__ mix, int L,
                                                         const double* __restrict__ ws) {
    const int b = blockIdx.y;
    const float mx = __uint_as_float(*reinterpret_cast<const unsigned int*>(&ws[(size_t)b * 4 + 2]));
    if (!(mx > 1.f)) return;
    const float g = 0.9f / mx;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        src[(size_t)b * L + i] *= g;
        mix[(size_t)b * L + i] *= g;
    }
}

// ---- SegmentMixer on the device (data/waveform_mixers.py:19-92; SURVEY section 8 row f4, mixer half) --------------------
// Clip n of a batch is mixed with its mix_num[n] - 1 successors (indices wrap): each is brought to the energy of clip n
// (ratio = clamp(sqrt(E_next / max(E_n, 1e-10)), 0.02, 50), next / ratio) and given an integer-dB gain; the summed noise
// gets the same treatment once more (`dynamic_loudnorm`, :85-92), mixture = segment + noise, and a mixture above 1 is
// brought to a 0.9 peak together with its segment (:49-53).  The random integers (mix_num, the dB draws) are INPUTS, so a
// restatement with the same draws is an oracle.  ws[n] = {sum x_n^2, sum noise_n^2, max |mixture_n| (float bits), -}.
__device__ __forceinline__ float seg_ratio(double e_audio, double e_ref, int L) {
    const float ea = (float)(e_audio / L);                      // get_energy = mean(x^2)   (:71-72)
    const float er = fmaxf((float)(e_ref / L), 1e-10f);         // max(get_energy(segment2), 1e-10)   (:78)
    return fminf(fmaxf(sqrtf(ea / er), 0.02f), 50.f);           // clamp(ratio, 0.02, 50)   (:79-80)
}
__device__ __forceinline__ float db_gain(float db) { return (float)pow(10.0, (double)db / 20.0); }  // np.power(10.0, d / 20.0)

__global__ __launch_bounds__(256) void seg_energy_kernel(const float* __restrict__ x, int L, double* __restrict__ ws) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    double s0 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const double v = x[(size_t)b * L + i];
        s0 += v * v;
    }
    s0 = wave_sum_d(s0);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&ws[(size_t)b * 4], red[0] + red[1] + red[2] + red[3]);
}

constexpr int SEG_MAXC = 7;  // components beside the primary segment (max_mix_num <= 8)

// STAGE 0: ws[n][1] = sum noise_n^2.  STAGE 1: mixture = segment + loudnorm(noise), segment copy, ws[n][2] = max |mixture|.
template <int STAGE>
__global__ __launch_bounds__(256) void seg_mix_kernel(const float* __restrict__ x, int B, int L, const int* __restrict__ mix_num,
                                                      const float* __restrict__ comp_db, int max_comp,
                                                      const float* __restrict__ noise_db, double* __restrict__ ws,
                                                      float* __restrict__ mixture, float* __restrict__ segment) {
    __shared__ double red[4];
    __shared__ float redf[4];
    const int n = blockIdx.y;
    int nc = mix_num[n] - 1;
    nc = nc < 0 ? 0 : (nc > max_comp ? max_comp : nc);
    const float* src[SEG_MAXC];
    float ratio[SEG_MAXC], gain[SEG_MAXC];
#pragma unroll
    for (int i = 0; i < SEG_MAXC; ++i) {
        const int o = (n + i + 1) % B;
        src[i] = x + (size_t)o * L;
        ratio[i] = i < nc ? seg_ratio(ws[(size_t)o * 4], ws[(size_t)n * 4], L) : 1.f;
        gain[i] = i < nc ? db_gain(comp_db[(size_t)n * max_comp + i]) : 0.f;
    }
    float r2 = 1.f, g2 = 0.f;
    if (STAGE == 1) {
        r2 = seg_ratio(ws[(size_t)n * 4 + 1], ws[(size_t)n * 4], L);
        g2 = db_gain(noise_db[n]);
    }
    double s0 = 0;
    float mx = 0.f;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < L; s += gridDim.x * 256) {
        float noise = 0.f;  // noise = zeros; noise += gain * (next / ratio), in component order   (:31-41)
#pragma unroll
        for (int i = 0; i < SEG_MAXC; ++i)
            if (i < nc) noise += gain[i] * (src[i][s] / ratio[i]);
        if (STAGE == 0) {
            s0 += (double)noise * (double)noise;
        } else {
            const float seg = x[(size_t)n * L + s];
            const float m = seg + g2 * (noise / r2);
            mixture[(size_t)n * L + s] = m;
            segment[(size_t)n * L + s] = seg;
            mx = fmaxf(mx, fabsf(m));
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (STAGE == 0) {
        s0 = wave_sum_d(s0);
        if (lane == 0) red[wv] = s0;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&ws[(size_t)n * 4 + 1], red[0] + red[1] + red[2] + red[3]);
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) redf[wv] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned int*>(&ws[(size_t)n * 4 + 2]),
                      __float_as_uint(fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]))));
    }
}

__global__ __launch_bounds__(256) void relayout_conv_kernel(const float* __restrict__ src, int Cout, int Cin, int taps,
                                                            float* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // index into dst [ci][tap][co]
    const long n = (long)Cout * Cin * taps;
    if (i >= n) return;
    const int co = (int)(i % Cout);
    const int tap = (int)((i / Cout) % taps);
    const int ci = (int)(i / ((long)Cout * taps));
    dst[i] = src[((size_t)co * Cin + ci) * taps + tap];
}

__global__ __launch_bounds__(256) void bnfold_kernel(const float* __restrict__ g, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     int C, float eps, float* __restrict__ scale,
                                                     float* __restrict__ base) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(var[c] + eps);
    const float s = g[c] * invstd;  // same factoring as ATen's eval-mode batch_norm: alpha = invstd*weight
    scale[c] = s;
    base[c] = beta[c] - mean[c] * s;
}

}  // namespace

hipError_t lass_launch_film(const float* cond, int B, const float* Wf, const float* bf, const float* base, int n,
                            float* out, hipStream_t stream) {
    if (B <= 0 || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(film_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, cond, B, Wf, bf, base, n, out);
    return hipGetLastError();
}

hipError_t lass_launch_preconv(const float* x0, const float* w, const float* bias, int B, int C, long HW, float* out,
                               hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || (HW % 4) != 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((HW / 4 + 255) / 256), C, B);
    hipLaunchKernelGGL(preconv_kernel, grid, dim3(256), 0, stream, x0, w, bias, C, HW, out);
    return hipGetLastError();
}

hipError_t lass_launch_pool(const float* in, long in_bs, int B, int C, int H, int W, int ph, int pw, float* out,
                            hipStream_t stream) {
    if (B <= 0 || C <= 0 || pw != 2 || (ph != 1 && ph != 2) || H < ph || (W % 2) != 0)  // odd H: floor, as F.avg_pool2d
        return hipErrorInvalidValue;
    const int npix = (H / ph) * (W / 2);
    dim3 grid((npix + 255) / 256, C, B);
    if (ph == 2)
        hipLaunchKernelGGL(pool_kernel<2>, grid, dim3(256), 0, stream, in, in_bs, C, H, W, out);
    else
        hipLaunchKernelGGL(pool_kernel<1>, grid, dim3(256), 0, stream, in, in_bs, C, H, W, out);
    return hipGetLastError();
}

// x0[b][t][f] = mag[b][t][f] * s0[f] + h0[f] for t < T, f < fcrop; 0 for T <= t < Tpad  (resunet.py:537-552 on a
// precomputed magnitude (B, T, fcrop+1))
__global__ __launch_bounds__(256) void x0_from_mag_kernel(const float* __restrict__ mag, int T, int Tpad, int fcrop,
                                                          const float* __restrict__ s0, const float* __restrict__ h0,
                                                          float* __restrict__ x0) {
    const int t = blockIdx.x, b = blockIdx.y;
    float* dst = x0 + ((size_t)b * Tpad + t) * fcrop;
    const float* src = mag + ((size_t)b * T + t) * (fcrop + 1);
    for (int f = threadIdx.x; f < fcrop; f += 256) dst[f] = t < T ? src[f] * s0[f] + h0[f] : 0.f;
}

hipError_t lass_launch_x0_from_mag(const float* mag, int B, int T, int Tpad, int fcrop, const float* s0,
                                   const float* h0, float* x0, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tpad < T || fcrop <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(x0_from_mag_kernel, dim3(Tpad, B), dim3(256), 0, stream, mag, T, Tpad, fcrop, s0, h0, x0);
    return hipGetLastError();
}

hipError_t lass_launch_mask(const float* x12, const float* wa, const float* ba, const float* mag, const float* cosv,
                            const float* sinv, int B, int T, int Tpad, int fcrop, float* out_real, float* out_imag,
                            hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tpad < T || fcrop <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mask_kernel, dim3(T, B), dim3(256), 0, stream, x12, wa, ba, mag, cosv, sinv, T, Tpad, fcrop,
                       out_real, out_imag);
    return hipGetLastError();
}

hipError_t lass_launch_sdr(const float* ref, const float* est, int B, int L, double* stats, hipStream_t stream) {
    if (B <= 0 || L <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 6 * (size_t)B, stream);
    if (e != hipSuccess) return e;
    int nb = (L + 256 * 16 - 1) / (256 * 16);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(sdr_pass1, dim3(nb, B), dim3(256), 0, stream, ref, est, L, stats);
    hipLaunchKernelGGL(sdr_pass2, dim3(nb, B), dim3(256), 0, stream, ref, est, L, stats);
    return hipGetLastError();
}

hipError_t lass_launch_relayout_conv(const float* src, int Cout, int Cin, int taps, float* dst, hipStream_t stream) {
    const long n = (long)Cout * Cin * taps;
    hipLaunchKernelGGL(relayout_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, Cout, Cin,
                       taps, dst);
    return hipGetLastError();
}

hipError_t lass_launch_bnfold(const float* g, const float* beta, const float* mean, const float* var, int C, float eps,
                              float* scale, float* base, hipStream_t stream) {
    hipLaunchKernelGGL(bnfold_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, g, beta, mean, var, C, eps, scale,
                       base);
    return hipGetLastError();
}

hipError_t lass_launch_mix_at_snr(float* source, const float* noise, const float* snr_db, float* mixture, int B, int L,
                                  double* ws, hipStream_t stream) {
    if (B <= 0 || L <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * 4 * B, stream);
    if (e != hipSuccess) return e;
    int nb = (L + 256 * 16 - 1) / (256 * 16);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(mix_power_kernel, dim3(nb, B), dim3(256), 0, stream, source, noise, L, ws);
    hipLaunchKernelGGL(mix_apply_kernel, dim3(nb, B), dim3(256), 0, stream, source, noise, snr_db, L, ws, mixture);
    hipLaunchKernelGGL(mix_declip_kernel, dim3(nb, B), dim3(256), 0, stream, source, mixture, L, ws);
    return hipGetLastError();
}

hipError_t lass_launch_segment_mix(const float* waveforms, int B, int L, const int* mix_num, const float* comp_db, int max_comp,
                                   const float* noise_db, float* mixture, float* segment, double* ws, hipStream_t stream) {
    if (B <= 0 || L <= 0 || max_comp < 1 || max_comp > SEG_MAXC) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * 4 * B, stream);
    if (e != hipSuccess) return e;
    int nb = (L + 256 * 16 - 1) / (256 * 16);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(seg_energy_kernel, dim3(nb, B), dim3(256), 0, stream, waveforms, L, ws);
    hipLaunchKernelGGL(seg_mix_kernel<0>, dim3(nb, B), dim3(256), 0, stream, waveforms, B, L, mix_num, comp_db, max_comp, noise_db,
                       ws, mixture, segment);
    hipLaunchKernelGGL(seg_mix_kernel<1>, dim3(nb, B), dim3(256), 0, stream, waveforms, B, L, mix_num, comp_db, max_comp, noise_db,
                       ws, mixture, segment);
    hipLaunchKernelGGL(mix_declip_kernel, dim3(nb, B), dim3(256), 0, stream, segment, mixture, L, ws);
    return hipGetLastError();
}
