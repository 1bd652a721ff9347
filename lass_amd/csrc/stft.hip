// stft.hip - STFT front end (fused magnitude/phase + bn0) and iSTFT back end for gfx950.
//
// Replaces (reference: /root/reference):
//   torchlibrosa STFT.forward  (call: models/base.py:84; cfg: models/resunet.py:284-292) - there a Conv1d against a
//       513x1024 windowed DFT matrix (1.05 GMAC/clip); here a 1024-point radix-4 Stockham FFT in LDS per frame.
//   Base.spectrogram_phase     (models/base.py:83-88, eps :91)
//   bn0 / T-pad / F-crop       (models/resunet.py:537-552)
//   torchlibrosa ISTFT.forward (call: models/resunet.py:510) - Hermitian extension, inverse DFT x Hann, overlap-add,
//       division by the window-sum-square envelope, trim.
// All of it is HBM-bound streaming: frames are read as contiguous 4-KB runs of the waveform, spectra are written as
// contiguous 513-float rows.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// u * exp(-i*theta) (forward) or u * exp(+i*theta) (inverse); w = (cos theta, sin theta)
template <bool INV>
__device__ __forceinline__ float2 ctw(float2 u, float2 w) {
    if (INV) return make_float2(u.x * w.x - u.y * w.y, u.y * w.x + u.x * w.y);
    return make_float2(u.x * w.x + u.y * w.y, u.y * w.x - u.x * w.y);
}

// 1024-point complex FFT, radix-4 Stockham autosort, 256 threads x 5 passes.  Input in `a`, output (natural order)
// in `b`.  tw[k] = (cos, sin)(2*pi*k/1024) in LDS.  Ends with a barrier.
template <bool INV>
__device__ __forceinline__ void fft1024(float2* a, float2* b, const float2* tw, int tid) {
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
        const int p = 1 << (2 * pass);
        const int k = tid & (p - 1);
        const int j = ((tid - k) << 2) + k;
        const int ts = 256 >> (2 * pass);
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (pass > 0) {
            u1 = ctw<INV>(u1, tw[k * ts]);
            u2 = ctw<INV>(u2, tw[2 * k * ts]);
            u3 = ctw<INV>(u3, tw[3 * k * ts]);
        }
        const float2 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3);
        float2 a3 = csub(u1, u3);
        a3 = INV ? make_float2(-a3.y, a3.x) : make_float2(a3.y, -a3.x);  // * (+i) inverse, * (-i) forward
        dst[j] = cadd(a0, a2);
        dst[j + p] = cadd(a1, a3);
        dst[j + 2 * p] = csub(a0, a2);
        dst[j + 3 * p] = csub(a1, a3);
        __syncthreads();
        float2* t = src;
        src = dst;
        dst = t;
    }
}

__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wav, int L, int T, int Tpad,
                                                   const float2* __restrict__ tw, const float* __restrict__ win,
                                                   float* __restrict__ mag, float* __restrict__ cosv,
                                                   float* __restrict__ sinv, float* __restrict__ real,
                                                   float* __restrict__ imag, float* __restrict__ x0,
                                                   const float* __restrict__ s0, const float* __restrict__ h0) {
    __shared__ float2 A[1024], Bf[1024], TW[1024];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (t >= T) {  // rows T..Tpad-1 of the network input are zeros AFTER bn0 (resunet.py:537-548)
        if (x0)
            for (int f = tid; f < LASS_FCROP; f += 256) x0[((size_t)b * Tpad + t) * LASS_FCROP + f] = 0.f;
        return;
    }
    const float* w = wav + (size_t)b * L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        TW[idx] = tw[idx];
        int n = t * LASS_HOP + idx - LASS_NFFT / 2;  // centre=True, reflect padding of n_fft/2
        if (n < 0) n = -n;
        if (n >= L) n = 2 * (L - 1) - n;
        A[idx] = make_float2(w[n] * win[idx], 0.f);
    }
    __syncthreads();
    fft1024<false>(A, Bf, TW, tid);
    const size_t row = ((size_t)b * T + t) * LASS_NBINS;
    for (int f = tid; f < LASS_NBINS; f += 256) {
        const float re = Bf[f].x, im = Bf[f].y;
        const float m = sqrtf(fmaxf(re * re + im * im, 1e-10f));  // clamp on |X|^2 (base.py:85)
        if (real) real[row + f] = re;
        if (imag) imag[row + f] = im;
        if (mag) mag[row + f] = m;
        if (cosv) cosv[row + f] = re / m;
        if (sinv) sinv[row + f] = im / m;
        if (x0 && f < LASS_FCROP) x0[((size_t)b * Tpad + t) * LASS_FCROP + f] = m * s0[f] + h0[f];
    }
}

// N-point complex forward FFT for N in {256, 512, 1024, 2048}: radix-4 Stockham passes plus one radix-2 pass when log2 N
// is odd; 256 threads, each taking (N/4)/256 butterflies per pass (or idling).  tw2k[m] = (cos, sin)(2*pi*m/2048).
template <int N>
__device__ __forceinline__ float2* fft_fwd(float2* a, float2* b, const float2* tw2k, int tid) {
    constexpr int LOG2 = N == 256 ? 8 : N == 512 ? 9 : N == 1024 ? 10 : 11;
    constexpr int NP4 = LOG2 / 2;
    constexpr int TWS = 2048 / N;  // stride into the 2048-entry table
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int pass = 0; pass < NP4; ++pass) {
        const int p = 1 << (2 * pass);
        for (int i = tid; i < N / 4; i += 256) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const int ts = (N / 4 / p) * TWS;
            float2 u0 = src[i], u1 = src[i + N / 4], u2 = src[i + N / 2], u3 = src[i + 3 * N / 4];
            if (pass > 0) {
                u1 = ctw<false>(u1, tw2k[k * ts]);
                u2 = ctw<false>(u2, tw2k[2 * k * ts]);
                u3 = ctw<false>(u3, tw2k[3 * k * ts]);
            }
            const float2 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3);
            float2 a3 = csub(u1, u3);
            a3 = make_float2(a3.y, -a3.x);  // * (-i)
            dst[j] = cadd(a0, a2);
            dst[j + p] = cadd(a1, a3);
            dst[j + 2 * p] = csub(a0, a2);
            dst[j + 3 * p] = csub(a1, a3);
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    if (LOG2 & 1) {  // final radix-2 pass, p = N/2
        for (int i = tid; i < N / 2; i += 256) {
            const float2 u0 = src[i];
            const float2 u1 = ctw<false>(src[i + N / 2], tw2k[i * TWS]);
            dst[i] = cadd(u0, u1);
            dst[i + N / 2] = csub(u0, u1);
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;  // natural-order spectrum
}

struct MultiStftArgs {
    int nwin;
    int n_fft[LASS_MAX_STFT_WINDOWS];
    float* mag[LASS_MAX_STFT_WINDOWS];
    float* cosv[LASS_MAX_STFT_WINDOWS];
    float* sinv[LASS_MAX_STFT_WINDOWS];
};

template <int N>
__device__ __forceinline__ void stft_frame(const float* __restrict__ w, int L, int hop, int t, size_t row,
                                           const float2* TW, float2* A, float2* Bf, float* __restrict__ mag,
                                           float* __restrict__ cosv, float* __restrict__ sinv, int tid) {
    constexpr int TWS = 2048 / N;
    for (int idx = tid; idx < N; idx += 256) {
        int n = t * hop + idx - N / 2;  // centre=True, reflect padding of n_fft/2
        if (n < 0) n = -n;
        if (n >= L) n = 2 * (L - 1) - n;
        const float wn = 0.5f - 0.5f * TW[idx * TWS].x;  // periodic Hann
        A[idx] = make_float2(w[n] * wn, 0.f);
    }
    __syncthreads();
    const float2* X = fft_fwd<N>(A, Bf, TW, tid);
    for (int f = tid; f <= N / 2; f += 256) {
        const float re = X[f].x, im = X[f].y;
        const float m = sqrtf(re * re + im * im);
        const float den = fmaxf(m, 1e-10f);  // torchlibrosa magphase: the clamp is on |X| (precompute_stfts.py:51)
        mag[row + f] = m;
        cosv[row + f] = re / den;
        sinv[row + f] = im / den;
    }
}

// One launch for all analysis windows (they share the hop, hence T): blockIdx.z selects the window.
__global__ __launch_bounds__(256) void multi_stft_kernel(const float* __restrict__ wav, int L, int T, int hop,
                                                         const float2* __restrict__ tw2k, MultiStftArgs a) {
    __shared__ float2 A[2048], Bf[2048], TW[2048];
    const int t = blockIdx.x, b = blockIdx.y, z = blockIdx.z, tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) TW[i] = tw2k[i];
    __syncthreads();
    const int N = a.n_fft[z];
    const float* w = wav + (size_t)b * L;
    const size_t row = ((size_t)b * T + t) * (N / 2 + 1);
    switch (N) {
        case 256: stft_frame<256>(w, L, hop, t, row, TW, A, Bf, a.mag[z], a.cosv[z], a.sinv[z], tid); break;
        case 512: stft_frame<512>(w, L, hop, t, row, TW, A, Bf, a.mag[z], a.cosv[z], a.sinv[z], tid); break;
        case 1024: stft_frame<1024>(w, L, hop, t, row, TW, A, Bf, a.mag[z], a.cosv[z], a.sinv[z], tid); break;
        default: stft_frame<2048>(w, L, hop, t, row, TW, A, Bf, a.mag[z], a.cosv[z], a.sinv[z], tid); break;
    }
}

__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ real,
                                                           const float* __restrict__ imag, int T,
                                                           const float2* __restrict__ tw,
                                                           const float* __restrict__ win,
                                                           float* __restrict__ frames) {
    __shared__ float2 A[1024], Bf[1024], TW[1024];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = ((size_t)b * T + t) * LASS_NBINS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        TW[k] = tw[k];
        // Hermitian extension 513 -> 1024 bins: Z[1024-k] = conj(Z[k])
        A[k] = (k <= 512) ? make_float2(real[row + k], imag[row + k])
                          : make_float2(real[row + 1024 - k], -imag[row + 1024 - k]);
    }
    __syncthreads();
    fft1024<true>(A, Bf, TW, tid);
    float* fr = frames + ((size_t)b * T + t) * LASS_NFFT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = tid + 256 * i;
        fr[n] = Bf[n].x * (win[n] * (1.0f / LASS_NFFT));
    }
}

// Gather-form overlap-add: each output sample sums the <= 7 frames that cover it and divides by the window
// sum-square envelope of the same frames (clamped at 1e-11), trimmed to [n_fft/2, n_fft/2 + L).
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int T, int L,
                                                        const float* __restrict__ win, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const int m = n + LASS_NFFT / 2;
    int t_hi = m / LASS_HOP;
    if (t_hi > T - 1) t_hi = T - 1;
    int t_lo = (m - (LASS_NFFT - 1) + LASS_HOP - 1) / LASS_HOP;
    if (m - (LASS_NFFT - 1) <= 0) t_lo = 0;
    float acc = 0.f, env = 0.f;
    const float* fb = frames + (size_t)b * T * LASS_NFFT;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int off = m - t * LASS_HOP;
        acc += fb[(size_t)t * LASS_NFFT + off];
        const float wv = win[off];
        env += wv * wv;
    }
    wav[(size_t)b * L + n] = acc / fmaxf(env, 1e-11f);
}

}  // namespace

hipError_t lass_launch_stft(const float* wav, int B, int L, int T, int Tpad, const float2* tw, const float* win,
                            float* mag, float* cosv, float* sinv, float* real, float* imag, float* x0,
                            const float* s0, const float* h0, hipStream_t stream) {
    if (B <= 0 || L <= LASS_NFFT / 2 || T != 1 + L / LASS_HOP || Tpad < T || (x0 && (!s0 || !h0)))
        return hipErrorInvalidValue;
    dim3 grid(x0 ? Tpad : T, B);
    hipLaunchKernelGGL(stft_kernel, grid, dim3(256), 0, stream, wav, L, T, Tpad, tw, win, mag, cosv, sinv, real, imag,
                       x0, s0, h0);
    return hipGetLastError();
}

hipError_t lass_launch_istft_frames(const float* real, const float* imag, int B, int T, const float2* tw,
                                    const float* win, float* frames, hipStream_t stream) {
    if (B <= 0 || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(istft_frames_kernel, dim3(T, B), dim3(256), 0, stream, real, imag, T, tw, win, frames);
    return hipGetLastError();
}

hipError_t lass_launch_istft_ola(const float* frames, int B, int T, int L, const float* win, float* wav,
                                 hipStream_t stream) {
    if (B <= 0 || T <= 0 || L <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((L + 255) / 256, B), dim3(256), 0, stream, frames, T, L, win, wav);
    return hipGetLastError();
}

hipError_t lass_launch_multi_stft(const float* wav, int B, int L, int hop, int nwin, const int* n_fft,
                                  const float2* tw2k, float* const* mag, float* const* cosv, float* const* sinv,
                                  hipStream_t stream) {
    if (B <= 0 || hop <= 0 || nwin <= 0 || nwin > LASS_MAX_STFT_WINDOWS) return hipErrorInvalidValue;
    MultiStftArgs a;
    a.nwin = nwin;
    for (int i = 0; i < nwin; ++i) {
        const int N = n_fft[i];
        if ((N != 256 && N != 512 && N != 1024 && N != 2048) || L <= N / 2 || !mag[i] || !cosv[i] || !sinv[i])
            return hipErrorInvalidValue;
        a.n_fft[i] = N; a.mag[i] = mag[i]; a.cosv[i] = cosv[i]; a.sinv[i] = sinv[i];
    }
    const int T = 1 + L / hop;
    hipLaunchKernelGGL(multi_stft_kernel, dim3(T, B, nwin), dim3(256), 0, stream, wav, L, T, hop, tw2k, a);
    return hipGetLastError();
}
