// stft.hip - STFT front end (fused magnitude/phase + bn0) and iSTFT back end for gfx950.
//
// Replaces (reference: /root/reference):
//   torchlibrosa STFT.forward  (call: models/base.py:84; cfg: models/resunet.py:284-292) - there a Conv1d against a
//       513x1024 windowed DFT matrix (1.05 GMAC/clip); here a radix-4 Stockham FFT in LDS, two real frames per complex
//       transform (n_fft 1024, or 2048 with zero-padded windows for the multi-STFT model).
//   Base.spectrogram_phase     (models/base.py:83-88, eps :91)
//   bn0 / T-pad / F-crop       (models/resunet.py:537-552)
//   torchlibrosa ISTFT.forward (call: models/resunet.py:510) - Hermitian extension, inverse DFT x Hann, overlap-add,
//       division by the window-sum-square envelope, trim: one fused kernel, overlap-add in LDS, no frame scratch.
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

#ifdef LASS_STFT_DBG
// Diagnostic build only (tools/stft_hazard_probe.py): every radix-4 pass of the forward transform of
// stft2_kernel records, per workgroup, [pass][kind][1024] float2 with kind 0 = the four points as READ from LDS, kind 1 = the
// three twiddles as READ from the LDS table (slot 0 unused), kind 2 = the four results as COMPUTED, in front of their LDS
// stores.  A wrong launch then says whether its first wrong value was read (LDS path) or computed (vector ALU).
__device__ float2* g_stft_dbg = nullptr;
#define STFT_DBG_REC(kind, slot, val) \
    if (dbg) dbg[(pass * 3 + (kind)) * N + i + (slot) * (N / 4)] = (val)
#else
#define STFT_DBG_REC(kind, slot, val)
#endif

// ---- generic pair-packed transforms (n_fft in {1024, 2048}, window length <= n_fft) ---------------------------------
// A real frame needs only half a complex transform: two frames ride in one complex FFT (frame a in the real part, frame b
// in the imaginary part) and are separated by the Hermitian symmetry of their spectra:
//     Z = FFT(xa + i xb):   Xa[k] = (Z[k] + conj(Z[N-k])) / 2,   Xb[k] = (Z[k] - conj(Z[N-k])) / (2i)
// and the other way round for the inverse.  Twiddles and the periodic Hann window come from the 2048-entry table.
template <int N, bool INV>
__device__ __forceinline__ float2* fft_c(float2* a, float2* b, const float2* tw2k, int tid, float2* dbg = nullptr) {
    constexpr int LOG2 = N == 1024 ? 10 : 11;
    constexpr int NP4 = LOG2 / 2;
    constexpr int TWS = 2048 / N;
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int pass = 0; pass < NP4; ++pass) {
        const int p = 1 << (2 * pass);
#pragma unroll
        for (int i = tid; i < N / 4; i += 256) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const int ts = (N / 4 / p) * TWS;
            float2 u0 = src[i], u1 = src[i + N / 4], u2 = src[i + N / 2], u3 = src[i + 3 * N / 4];
            STFT_DBG_REC(0, 0, u0); STFT_DBG_REC(0, 1, u1); STFT_DBG_REC(0, 2, u2); STFT_DBG_REC(0, 3, u3);
            if (pass > 0) {
                const float2 w1 = tw2k[k * ts], w2 = tw2k[2 * k * ts], w3 = tw2k[3 * k * ts];
                STFT_DBG_REC(1, 1, w1); STFT_DBG_REC(1, 2, w2); STFT_DBG_REC(1, 3, w3);
                u1 = ctw<INV>(u1, w1);
                u2 = ctw<INV>(u2, w2);
                u3 = ctw<INV>(u3, w3);
            }
            const float2 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3);
            float2 a3 = csub(u1, u3);
            a3 = INV ? make_float2(-a3.y, a3.x) : make_float2(a3.y, -a3.x);  // * (+i) inverse, * (-i) forward
            const float2 r0 = cadd(a0, a2), r1 = cadd(a1, a3), r2 = csub(a0, a2), r3 = csub(a1, a3);
            STFT_DBG_REC(2, 0, r0); STFT_DBG_REC(2, 1, r1); STFT_DBG_REC(2, 2, r2); STFT_DBG_REC(2, 3, r3);
            dst[j] = r0;
            dst[j + p] = r1;
            dst[j + 2 * p] = r2;
            dst[j + 3 * p] = r3;
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    if (LOG2 & 1) {  // final radix-2 pass, p = N/2
#pragma unroll
        for (int i = tid; i < N / 2; i += 256) {
            const float2 u0 = src[i];
            const float2 u1 = ctw<INV>(src[i + N / 2], tw2k[i * TWS]);
            dst[i] = cadd(u0, u1);
            dst[i + N / 2] = csub(u0, u1);
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;  // natural order
}

struct Stft2Args {
    int nbr;
    int wlen[LASS_MAX_STFT_WINDOWS];
    float* mag[LASS_MAX_STFT_WINDOWS];
    float* cosv[LASS_MAX_STFT_WINDOWS];
    float* sinv[LASS_MAX_STFT_WINDOWS];
    float* real[LASS_MAX_STFT_WINDOWS];
    float* imag[LASS_MAX_STFT_WINDOWS];
    float* x0[LASS_MAX_STFT_WINDOWS];
};

// Centred, reflect-padded STFT of frames (2*blockIdx.x, 2*blockIdx.x + 1) of clip blockIdx.y for branch blockIdx.z
// (periodic Hann of wlen samples zero-padded to N, centred), fused with magnitude / phase and the network-input prologue
// x0 = bn0(mag) with T zero-padded to Tpad and the Nyquist bin dropped (resunet.py:533-552).
// MAGPHASE: torchlibrosa magphase (clamp on |X|, precompute_stfts.py:51) instead of base.py:83-88 (clamp on |X|^2).
template <int N, bool MAGPHASE>
__global__ __launch_bounds__(256) void stft2_kernel(const float* __restrict__ wav, int L, int hop, int T, int Tpad,
                                                    const float2* __restrict__ tw2k, Stft2Args a,
                                                    const float* __restrict__ s0, const float* __restrict__ h0) {
    __shared__ float2 A[N], Bf[N], TW[2048];
    constexpr int NB = N / 2 + 1, FC = N / 2;
    const int ta = 2 * blockIdx.x, b = blockIdx.y, z = blockIdx.z, tid = threadIdx.x;
    float* x0 = a.x0[z];
    if (ta >= T) {  // rows T..Tpad-1 of the network input are zeros AFTER bn0
        if (x0)
            for (int r = 0; r < 2; ++r)
                if (ta + r < Tpad)
                    for (int f = tid; f < FC; f += 256) x0[((size_t)b * Tpad + ta + r) * FC + f] = 0.f;
        return;
    }
    for (int i = tid; i < 2048; i += 256) TW[i] = tw2k[i];
    __syncthreads();
    const int wlen = a.wlen[z], woff = (N - wlen) / 2, wstride = 2048 / wlen;
    const float* w = wav + (size_t)b * L;
    const bool have_b = ta + 1 < T;
    for (int idx = tid; idx < N; idx += 256) {
        float2 v = make_float2(0.f, 0.f);
        const int j = idx - woff;
        if (j >= 0 && j < wlen) {
            const float wn = 0.5f - 0.5f * TW[j * wstride].x;  // periodic Hann of wlen
            int n = ta * hop + idx - N / 2;                     // centre=True, reflect padding of n_fft/2
            int n2 = n + hop;
            if (n < 0) n = -n;
            if (n >= L) n = 2 * (L - 1) - n;
            if (n2 < 0) n2 = -n2;
            if (n2 >= L) n2 = 2 * (L - 1) - n2;
            v = make_float2(w[n] * wn, have_b ? w[n2] * wn : 0.f);
        }
        A[idx] = v;
    }
    __syncthreads();
#ifdef LASS_STFT_DBG
    float2* dbg = g_stft_dbg ? g_stft_dbg + (size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (N == 1024 ? 5 : 5) * 3 * N : nullptr;
    const float2* Z = fft_c<N, false>(A, Bf, TW, tid, dbg);
#else
    const float2* Z = fft_c<N, false>(A, Bf, TW, tid);
#endif
    float *mag = a.mag[z], *cosv = a.cosv[z], *sinv = a.sinv[z], *real = a.real[z], *imag = a.imag[z];
    for (int f = tid; f < NB; f += 256) {
        const float2 z1 = Z[f], z2 = Z[(N - f) & (N - 1)];
        const float re[2] = {0.5f * (z1.x + z2.x), 0.5f * (z1.y + z2.y)};
        const float im[2] = {0.5f * (z1.y - z2.y), -0.5f * (z1.x - z2.x)};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !have_b) break;
            const size_t row = ((size_t)b * T + ta + r) * NB + f;
            float m, c, s;
            if (MAGPHASE) {
                m = sqrtf(re[r] * re[r] + im[r] * im[r]);
                const float den = fmaxf(m, 1e-10f);
                c = re[r] / den; s = im[r] / den;
            } else {
                m = sqrtf(fmaxf(re[r] * re[r] + im[r] * im[r], 1e-10f));
                c = re[r] / m; s = im[r] / m;
            }
            if (real) real[row] = re[r];
            if (imag) imag[row] = im[r];
            if (mag) mag[row] = m;
            if (cosv) cosv[row] = c;
            if (sinv) sinv[row] = s;
            if (x0 && f < FC) x0[((size_t)b * Tpad + ta + r) * FC + f] = m * s0[f] + h0[f];
        }
    }
    if (x0 && !have_b && ta + 1 < Tpad)
        for (int f = tid; f < FC; f += 256) x0[((size_t)b * Tpad + ta + 1) * FC + f] = 0.f;
}

// Inverse STFT, fused: a workgroup owns SPAN consecutive samples of the (n_fft/2-padded) output, runs the inverse
// transforms of every frame whose window reaches into that span (two frames per complex transform) and overlap-adds
// them in LDS in ascending frame order, then divides by the window-sum-square envelope of the same frames (clamped at
// 1e-11) and writes the trimmed samples [n_fft/2, n_fft/2 + L).  No frame scratch in HBM.
constexpr int ISTFT_SPAN = 16 * LASS_HOP;

template <int N>
__global__ __launch_bounds__(256) void istft2_kernel(const float* __restrict__ real, const float* __restrict__ imag,
                                                     int T, int L, int hop, int wlen,
                                                     const float2* __restrict__ tw2k, float* __restrict__ wav) {
    __shared__ float2 A[N], Bf[N], TW[2048];
    __shared__ float acc[ISTFT_SPAN], env[ISTFT_SPAN];
    constexpr int NB = N / 2 + 1;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int m0 = blockIdx.x * ISTFT_SPAN;  // first padded position of this span
    const int woff = (N - wlen) / 2, wstride = 2048 / wlen;
    for (int i = tid; i < 2048; i += 256) TW[i] = tw2k[i];
    for (int i = tid; i < ISTFT_SPAN; i += 256) { acc[i] = 0.f; env[i] = 0.f; }
    // frames whose window support [t*hop + woff, t*hop + woff + wlen) meets [m0, m0 + SPAN)
    int t_lo = m0 - woff - wlen + 1;
    t_lo = t_lo <= 0 ? 0 : (t_lo + hop - 1) / hop;
    int t_hi = (m0 + ISTFT_SPAN - 1 - woff) / hop;
    if (m0 + ISTFT_SPAN - 1 - woff < 0) t_hi = -1;
    if (t_hi > T - 1) t_hi = T - 1;
    __syncthreads();
    for (int ta = t_lo; ta <= t_hi; ta += 2) {
        const bool have_b = ta + 1 <= t_hi;
        const size_t rowa = ((size_t)b * T + ta) * NB, rowb = rowa + NB;
        // Z = Xa + i Xb with both spectra Hermitian-extended: bin k > N/2 is the conjugate of bin N-k
        for (int k = tid; k < N; k += 256) {
            const int kk = k <= N / 2 ? k : N - k;
            // bins 0 and N/2 of a real frame's spectrum are real: their imaginary parts are ignored (sin(0) = sin(pi n)
            // = 0 in the reference's inverse-DFT matrix) - and must not leak into the partner frame of the packed pair
            const float sg = (kk == 0 || kk == N / 2) ? 0.f : (k <= N / 2 ? 1.f : -1.f);
            const float ar = real[rowa + kk], ai = sg * imag[rowa + kk];
            const float br = have_b ? real[rowb + kk] : 0.f, bi = have_b ? sg * imag[rowb + kk] : 0.f;
            A[k] = make_float2(ar - bi, ai + br);
        }
        __syncthreads();
        const float2* X = fft_c<N, true>(A, Bf, TW, tid);  // (xa, xb) * N
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !have_b) break;
            const int base = (ta + r) * hop + woff - m0;  // span position of window sample 0
            for (int j = tid; j < wlen; j += 256) {
                const int pos = base + j;
                if (pos >= 0 && pos < ISTFT_SPAN) {
                    const float wn = 0.5f - 0.5f * TW[j * wstride].x;
                    const float2 x = X[woff + j];
                    acc[pos] += (r == 0 ? x.x : x.y) * (wn * (1.0f / N));
                    env[pos] += wn * wn;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < ISTFT_SPAN; i += 256) {
        const int n = m0 + i - N / 2;
        if (n >= 0 && n < L) wav[(size_t)b * L + n] = acc[i] / fmaxf(env[i], 1e-11f);
    }
}

}  // namespace

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

hipError_t lass_launch_stft2(const float* wav, int B, int L, int n_fft, int hop, int T, int Tpad, int nbr,
                             const StftBranch* br, int magphase_sem, const float* s0, const float* h0,
                             const float2* tw2k, hipStream_t stream) {
    if (B <= 0 || hop <= 0 || nbr <= 0 || nbr > LASS_MAX_STFT_WINDOWS || (n_fft != 1024 && n_fft != 2048) ||
        L <= n_fft / 2 || T != 1 + L / hop || Tpad < T)
        return hipErrorInvalidValue;
    Stft2Args a;
    a.nbr = nbr;
    bool any_x0 = false;
    for (int i = 0; i < nbr; ++i) {
        const int wl = br[i].wlen;
        if (wl <= 0 || wl > n_fft || (2048 % wl) != 0 || ((n_fft - wl) & 1)) return hipErrorInvalidValue;
        if (br[i].x0 && (!s0 || !h0)) return hipErrorInvalidValue;
        any_x0 |= br[i].x0 != nullptr;
        a.wlen[i] = wl; a.mag[i] = br[i].mag; a.cosv[i] = br[i].cosv; a.sinv[i] = br[i].sinv;
        a.real[i] = br[i].real; a.imag[i] = br[i].imag; a.x0[i] = br[i].x0;
    }
    dim3 grid(((any_x0 ? Tpad : T) + 1) / 2, B, nbr);
#define LASS_STFT2(NN, MP) hipLaunchKernelGGL((stft2_kernel<NN, MP>), grid, dim3(256), 0, stream, wav, L, hop, T, Tpad, tw2k, a, s0, h0)
    if (n_fft == 1024) { if (magphase_sem) LASS_STFT2(1024, true); else LASS_STFT2(1024, false); }
    else               { if (magphase_sem) LASS_STFT2(2048, true); else LASS_STFT2(2048, false); }
#undef LASS_STFT2
    return hipGetLastError();
}

hipError_t lass_launch_istft2(const float* real, const float* imag, int B, int T, int L, int n_fft, int wlen, int hop,
                              const float2* tw2k, float* wav, hipStream_t stream) {
    if (B <= 0 || T <= 0 || L <= 0 || hop != LASS_HOP || (n_fft != 1024 && n_fft != 2048) || wlen <= 0 || wlen > n_fft ||
        (2048 % wlen) != 0 || ((n_fft - wlen) & 1))
        return hipErrorInvalidValue;
    dim3 grid((n_fft / 2 + L + ISTFT_SPAN - 1) / ISTFT_SPAN, B);
    if (n_fft == 1024)
        hipLaunchKernelGGL(istft2_kernel<1024>, grid, dim3(256), 0, stream, real, imag, T, L, hop, wlen, tw2k, wav);
    else
        hipLaunchKernelGGL(istft2_kernel<2048>, grid, dim3(256), 0, stream, real, imag, T, L, hop, wlen, tw2k, wav);
    return hipGetLastError();
}

#ifdef LASS_STFT_DBG
// diagnostic build: the record buffer of the NEXT stft2_kernel launches on `stream` (nullptr = off)
extern "C" int lass_dbg_stft_buffer(void* buf, hipStream_t stream) {
    float2* pbuf = (float2*)buf;
    return (int)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stft_dbg), &pbuf, sizeof(pbuf), 0, hipMemcpyHostToDevice, stream);
}
#endif
