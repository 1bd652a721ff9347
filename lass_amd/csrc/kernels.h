// kernels.h - internal launch interfaces between the C-ABI layer (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

constexpr int LASS_NFFT = 1024;
constexpr int LASS_HOP = 160;
constexpr int LASS_NBINS = 513;
constexpr int LASS_FCROP = 512;
constexpr int LASS_COND = 512;
constexpr int LASS_MAX_STFT_WINDOWS = 4;

// ---- conv.hip -----------------------------------------------------------------------------------------------------
struct ConvArgs {
    // phase A: K = Cin x TAPS
    const float* in = nullptr;  // (B, Cin, H, W); batch stride in_bs, channel stride H*W
    long in_bs = 0;
    int Cin = 0;
    const float* w = nullptr;  // Wt[Cin][TAPS][Nw]
    int Nw = 0;                // row pitch of the weight matrix (total output channels it holds)
    const float* pro_scale = nullptr;  // [Cin]
    const float* pro_shift = nullptr;  // [B][pro_shift_bs], column = channel
    int pro_shift_bs = 0;
    // phase B: 1x1 shortcut over the raw block input
    const float* in2 = nullptr;
    long in2_bs = 0;
    int Cin2 = 0;
    const float* w2 = nullptr;    // [Cin2][Nw]
    const float* bias = nullptr;  // [Nw]
    const float* res = nullptr;   // (B, N, H, W) residual
    long res_bs = 0;
    const float* epi_scale = nullptr;  // [N]
    const float* epi_shift = nullptr;  // [B][epi_shift_bs]
    int epi_shift_bs = 0;
    float* out = nullptr;
    long out_bs = 0;
    int B = 0, H = 0, W = 0;
    int N = 0;  // output channels computed by this launch (GEMM rows)
    int up_h = 1;  // TCONV: vertical stride (1 or 2); horizontal stride is always 2
    const float* w_wino = nullptr;   // Winograd-domain weights U[16][Cin][Nw] (wino.hip)
    const float* w2_wino = nullptr;  // Winograd-domain shortcut weights [4][Cin2][Nw]
    const float* w_wino4 = nullptr;    // Winograd F(4x4,3x3)-domain weights: LDS images of wino4.hip (36 * Cin * Nw floats)
    const float* w_wino32 = nullptr;   // 32-cout layers: the resident LDS image of wino32.hip (Cin * 512 floats)
    const float* w2_wino32 = nullptr;  // ... of the shortcut weights (Cin2 * 128 floats)
    const void* w_bf16 = nullptr;    // bf16 weights [Cin/16][tap][2][Nw][8] (conv_bf16.hip)
    const void* w2_bf16 = nullptr;   // bf16 shortcut weights [Cin2/16][1][2][Nw][8]
    const void* w_bf16_lo = nullptr;   // split mode: bf16(w - float(bf16(w))), same layouts; non-null selects the 3-MFMA kernels
    const void* w2_bf16_lo = nullptr;
    // bf16 modes: the block's intermediate (conv1 -> conv2) in the blocked bf16 layout [B][C/8][H][W][8], hi (+ lo) planes
    void* out_bf16 = nullptr;
    void* out_bf16_lo = nullptr;
    const void* in_bf16 = nullptr;
    const void* in_bf16_lo = nullptr;
    // bf16 mode, decoder inputs: the concat (transposed-conv output | encoder skip) is kept as TWO blocked bf16 copies
    // instead of f32 - `raw` for the 1x1 shortcut and `act` = leaky(v*act_scale[c] + act_shift[b][c]) (the consumer's
    // BN+FiLM prologue) for conv1.  Producers write octets [out_oct0, out_oct0 + N/8) of out_noct octets per clip.
    void* out_bf16_act = nullptr;
    const float* act_scale = nullptr;   // indexed by this launch's output channel
    const float* act_shift = nullptr;   // [B][act_shift_bs]
    int act_shift_bs = 0;
    int out_oct0 = 0;
    int out_noct = 0;                   // 0: N / 8
    const void* in2_bf16 = nullptr;     // phase B (1x1 shortcut) input as the blocked bf16 raw copy
    // bf16 mode, encoder chain: the fused avg-pool output as two blocked bf16 copies (raw + activated with the NEXT block's
    // conv1 prologue) instead of f32; replaces pool_out when set
    void* pool_bf16 = nullptr;
    void* pool_bf16_act = nullptr;
    int pool_oct0 = 0;                      // the launch writes octets [pool_oct0, pool_oct0 + N/8) of pool_noct per clip
    int pool_noct = 0;                      // (0: N / 8) - a channel slice of a wider pooled tensor (multi-STFT branches)
    const float* pool_act_scale = nullptr;  // indexed by this launch's output channel
    const float* pool_act_shift = nullptr;  // [B][act_shift_bs]
    const float* pre_w = nullptr;  // pre_conv (1x1, 1 -> 32) weight / bias for the *_PRE kinds
    const float* pre_b = nullptr;
    // fused output head (decoder_block6.conv2 only, N == 32, W == 512): after_conv (1x1, 32 -> 3, + bias) and the complex
    // ratio mask (resunet.py:570-574,436-519) in the epilogue; the block's own output is then not written at all
    const float* mask_w = nullptr;    // after_conv.weight [3][32]
    const float* mask_b = nullptr;    // after_conv.bias [3]
    const float* mask_mag = nullptr;  // (B, mask_T, 513) mixture magnitude / cos / sin
    const float* mask_cos = nullptr;
    const float* mask_sin = nullptr;
    float* mask_re = nullptr;         // (B, mask_T, 513) separated spectrum
    float* mask_im = nullptr;
    int mask_T = 0;                   // frames (rows y >= mask_T are the T padding: dropped)
    int mask_nbins = LASS_NBINS;      // bins per spectrum row = W + 1 (513; 1025 for the multi-STFT model)
    float* pool_out = nullptr;  // fused avg-pool of the output: (B, N, H/pool_h, W/2), batch stride pool_bs
    long pool_bs = 0;            // elements between clips of pool_out; 0 = dense (N * H/pool_h * W/2).  A launch that
                                 // produces a channel slice of a wider pooled tensor passes the wide tensor's stride.
    int pool_h = 2;              // vertical pool factor (1 or 2); horizontal is 2
    // XCD-aware block order (wino.hip, conv_bf16.hip): launched as a 1-D grid of gx * gy * B workgroups; set by the launcher
    int gx = 0, gy = 0;        // spatial tiles per clip, cout blocks
    int xcd_map = 0;           // 1, 2: the gy cout blocks of one (tile, clip) run on ONE XCD (they re-read the same input tile);
                               // 2: and every XCD walks one contiguous range of tiles (block_coords, wino_common.h)
    long long* dbg = nullptr;  // diagnostic builds (-DLASS_CONV_DIAG) only: 8 int64 per block
    int exp = 0;  // diagnostic builds only: timing-experiment switches (env LASS_EXP, see wino.hip)
};

enum ConvKind { CONV1_ACT = 0, CONV2_IDENT = 1, CONV2_SHORTCUT = 2, TCONV_ACT = 3, CONV1_ACT_PRE = 4, CONV2_IDENT_PRE = 5 };

hipError_t lass_launch_conv(ConvKind kind, const ConvArgs& p, hipStream_t stream);

// ---- wino.hip (Winograd F(2x2,3x3) variant of the 3x3 kinds; W must be a multiple of 32, H even) -------------------
bool lass_wino_supported(const ConvArgs& p);
hipError_t lass_launch_wino(ConvKind kind, const ConvArgs& p, hipStream_t stream);
hipError_t lass_launch_wino_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream);
hipError_t lass_launch_wino_shortcut_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream);

// ---- wino32.hip (the 32-cout full-resolution layers: weights resident in LDS, persistent, waves decoupled) ------------
bool lass_wino32_supported(ConvKind kind, const ConvArgs& p);
hipError_t lass_launch_wino32(ConvKind kind, const ConvArgs& p, hipStream_t stream);
hipError_t lass_launch_wino32_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream);            // w (Cout, Cin, 3, 3), Cout % 32 == 0: one image per 32-cout slice
hipError_t lass_launch_wino32_shortcut_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream);   // w (Cout, Cin, 1, 1)

// ---- wino4.hip (Winograd F(4x4,3x3): 36 instead of 64 MFMA multiplies per 16 outputs; the conv1 kind, W % 32 == 0) ----------
bool lass_wino4_supported(ConvKind kind, const ConvArgs& p);
hipError_t lass_launch_wino4(ConvKind kind, const ConvArgs& p, hipStream_t stream);
hipError_t lass_launch_wino4_weights(const float* w, int Cout, int Cin, float* U, hipStream_t stream);  // w (Cout, Cin, 3, 3)

// ---- conv_bf16.hip (bf16-MFMA variant of the 3x3 kinds; W multiple of 32, Cin multiple of 16) ----------------------
bool lass_bf16_supported(const ConvArgs& p);
hipError_t lass_launch_conv_bf16(ConvKind kind, const ConvArgs& p, hipStream_t stream);
hipError_t lass_launch_weights_bf16(const float* w, int Cout, int Cin, int taps, void* dst, int lo, int transposed,
                                    hipStream_t stream);

// ---- conv_bf16_fused.hip (bf16 mode: encoder_block1's ConvBlockRes as one kernel, intermediate in LDS) -------------------
// p = the block's CONV1_ACT_PRE arguments, q = its CONV2_IDENT_PRE arguments with blocked bf16 outputs
bool lass_enc1_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q);
hipError_t lass_launch_enc1_fused_bf16(const ConvArgs& p, const ConvArgs& q, hipStream_t stream);
// ... and decoder_block6's (conv1 64 -> 32 from the activated cat copy, conv2 + 1x1 shortcut over the raw copy + output head)
bool lass_dec6_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q);
hipError_t lass_launch_dec6_fused_bf16(const ConvArgs& p, const ConvArgs& q, hipStream_t stream);

// ... and the same block with decoder_block6's transposed conv computed inside (the up-sampled half of the concat never goes
// to HBM).  u: in_bf16 = the previous decoder's activated blocked-bf16 output (B, 64/8, H/2, W/2, 8), w_bf16 = the transposed
// conv's bf16 weights, w2_bf16 = the composed shortcut weights (lass_launch_compose_up_shortcut -> lass_launch_weights_bf16),
// H, W = the low resolution, Cin = 64
bool lass_dec6u_fused_bf16_supported(const ConvArgs& p, const ConvArgs& q, const ConvArgs& u);
hipError_t lass_launch_dec6u_fused_bf16(const ConvArgs& p, const ConvArgs& q, const ConvArgs& u, hipStream_t stream);
// out[((a*2+bb) * Nsc + n) * Cin + ci] = sum_co wsc[n][co] * wt[ci][co][a][bb]   (f32; wsc (Nsc, Ccat, 1, 1), wt (Cin, Cup, 2, 2))
hipError_t lass_launch_compose_up_shortcut(const float* wsc, const float* wt, int Cin, int Cup, int Ccat, int Nsc, float* out,
                                           hipStream_t stream);

// ---- stft.hip -----------------------------------------------------------------------------------------------------
// Multi-resolution analysis (scripts/precompute_stfts.py:19-58,573-590): nwin centred STFTs (n_fft = win in {256, 512,
// 1024, 2048}, periodic Hann, reflect pad, common hop) of the same waveforms in one launch, torchlibrosa-magphase
// semantics (clamp on |X| at 1e-10).  Outputs (B, T, n_fft/2+1) each, T = 1 + L/hop.  tw2k: 2048 (cos, sin)(2*pi*k/2048).
hipError_t lass_launch_multi_stft(const float* wav, int B, int L, int hop, int nwin, const int* n_fft,
                                  const float2* tw2k, float* const* mag, float* const* cosv, float* const* sinv,
                                  hipStream_t stream);
// Generic pair-packed transforms (stft.hip): n_fft in {1024, 2048}; each branch is a periodic Hann window of `wlen`
// samples (a divisor of 2048, <= n_fft) zero-padded to n_fft and centred.  Outputs per branch where non-null:
// mag/cos/sin/real/imag (B, T, n_fft/2+1) and x0 (B, Tpad, n_fft/2) = bn0(mag), zero rows T..Tpad-1, Nyquist dropped.
// magphase_sem: 0 = base.py:83-88 (clamp on |X|^2), 1 = torchlibrosa magphase (clamp on |X|).
struct StftBranch {
    int wlen = 0;
    float *mag = nullptr, *cosv = nullptr, *sinv = nullptr, *real = nullptr, *imag = nullptr, *x0 = nullptr;
};
hipError_t lass_launch_stft2(const float* wav, int B, int L, int n_fft, int hop, int T, int Tpad, int nbr,
                             const StftBranch* br, int magphase_sem, const float* s0, const float* h0,
                             const float2* tw2k, hipStream_t stream);
// Fused inverse STFT (inverse transforms + overlap-add + envelope + trim in one kernel, no frame scratch).
hipError_t lass_launch_istft2(const float* real, const float* imag, int B, int T, int L, int n_fft, int wlen, int hop,
                              const float2* tw2k, float* wav, hipStream_t stream);

// ---- misc.hip -----------------------------------------------------------------------------------------------------
// film[b][j] = dot(cond[b], Wf[j]) + bf[j] (+ base[j] if base)   for j < n
hipError_t lass_launch_film(const float* cond, int B, const float* Wf, const float* bf, const float* base, int n,
                            float* out, hipStream_t stream);
// out[b][c][t][f] = w[c]*x0[b][t][f] + bias[c]
hipError_t lass_launch_preconv(const float* x0, const float* w, const float* bias, int B, int C, long HW, float* out,
                               hipStream_t stream);
// average pool (ph x pw) of (B,C,H,W) with batch stride in_bs -> (B,C,H/ph,W/pw) dense
hipError_t lass_launch_pool(const float* in, long in_bs, int B, int C, int H, int W, int ph, int pw, float* out,
                            hipStream_t stream);
hipError_t lass_launch_mask(const float* x12, const float* wa, const float* ba, const float* mag, const float* cosv,
                            const float* sinv, int B, int T, int Tpad, int fcrop, float* out_real, float* out_imag,
                            hipStream_t stream);
// x0 (B,Tpad,fcrop) = bn0 affine of a precomputed magnitude (B,T,fcrop+1), zero rows T..Tpad-1
hipError_t lass_launch_x0_from_mag(const float* mag, int B, int T, int Tpad, int fcrop, const float* s0,
                                   const float* h0, float* x0, hipStream_t stream);
hipError_t lass_launch_sdr(const float* ref, const float* est, int B, int L, double* stats, hipStream_t stream);
// mixture = source + noise * sqrt(P_source / 10^(snr/10) / P_noise); if max|mixture| > 1 both source and mixture are
// scaled by 0.9/max (dcase_evaluator.py:77-89).  source is updated in place; ws: 4*B doubles of scratch.
hipError_t lass_launch_mix_at_snr(float* source, const float* noise, const float* snr_db, float* mixture, int B, int L,
                                  double* ws, hipStream_t stream);
// SegmentMixer.__call__ (data/waveform_mixers.py:19-62) with the random draws passed in: clip n + its mix_num[n] - 1 successors
// (energy-matched, comp_db[n][i] dB each), the noise sum energy-matched again (+ noise_db[n] dB), declipped to a 0.9 peak.
// ws: 4*B doubles of scratch; max_comp = columns of comp_db (1 ... 7).
hipError_t lass_launch_segment_mix(const float* waveforms, int B, int L, const int* mix_num, const float* comp_db, int max_comp,
                                   const float* noise_db, float* mixture, float* segment, double* ws, hipStream_t stream);
// dst[ci][tap][co] = src[co][ci][tap]
hipError_t lass_launch_relayout_conv(const float* src, int Cout, int Cin, int taps, float* dst, hipStream_t stream);
// scale[c] = g/sqrt(var+eps); base[c] = beta - mean*scale
hipError_t lass_launch_bnfold(const float* g, const float* beta, const float* mean, const float* var, int C, float eps,
                              float* scale, float* base, hipStream_t stream);
