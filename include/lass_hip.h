/*
 * lass_hip.h - C-ABI of liblass_hip.so: the MI355X (gfx950) text-conditioned separation hot path.
 *
 *     mixture (B,L) f32 + condition (B,512) f32  --STFT--> |X|,cos,sin --FiLM ResUNet30--> mask --iSTFT--> (B,L) f32
 *
 * The reference (reedrosenbluth/LASS) is pure Python and has no FFI of its own; this boundary replaces what its
 * evaluator calls two levels down:  `pl_model.ss_model(input_dict)["waveform"]`  (dcase_evaluator.py:99-107), i.e.
 * `ResUNet30.forward` (models/resunet.py:640-653) = `FiLM.forward` (:59-81) + `ResUNet30_Base.forward` (:522-595),
 * plus the numpy metrics `calculate_sdr` / `calculate_sisdr` (utils.py:148-200).  Each entry point below cites the
 * reference code it stands in for.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes; no torch / C++ types.  All tensor pointers are DEVICE pointers unless said otherwise.
 *   - every function returns 0 on success, <0 on error; `lass_last_error(ctx)` has the message.
 *   - kernels are enqueued on the caller's `stream` (a hipStream_t passed as void*) and never synchronise.
 *   - the caller owns inputs, outputs and the workspace; the context owns re-laid-out weights and constant tables.
 *     `lass_separate` allocates nothing.
 *   - one context per (process, device); not re-entrant per context.  Every entry point makes the context's device
 *     current (hipSetDevice) before it launches or allocates, so the caller's current device does not matter.
 *   - there is NO CPU fallback: without a gfx950 device `lass_create` fails.
 */
#ifndef LASS_HIP_H
#define LASS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lass_ctx lass_ctx;

/* error codes */
#define LASS_OK 0
#define LASS_ERR_ARG (-1)      /* bad argument / shape */
#define LASS_ERR_HIP (-2)      /* a HIP runtime call failed */
#define LASS_ERR_STATE (-3)    /* call order (e.g. separate before finalize, missing parameter) */
#define LASS_ERR_WORKSPACE (-4)/* workspace too small */

/* dtypes for lass_set_param */
#define LASS_F32 0
#define LASS_I64 1

/* compute modes for lass_finalize */
#define LASS_COMPUTE_F32 0     /* f32 storage, f32 MFMA contractions (plain f32 FMAs): Winograd F(4x4,3x3) for the 3x3 convs with
                                  >= 64 input channels at >= 32-bin resolutions (LASS_WINO4=0: none), F(2x2,3x3) for the rest */
#define LASS_COMPUTE_BF16 1    /* convolutions contracted on the bf16 MFMA (operands rounded to bf16, f32 accumulate):
                                  BASELINE configs[2].  Inside the workspace the tensors handed from one conv launch to
                                  the next (block intermediates, decoder concats, pooled encoder outputs, transposed-conv
                                  inputs) are stored as blocked bf16 [C/8][H][W][8]; spectra, masks and waveforms stay
                                  f32.  Reduced precision, looser parity. */
#define LASS_COMPUTE_BF16X3 2  /* as BF16 but every operand is split hi+lo (two bf16) and each product formed as
                                  hi*hi + hi*lo + lo*hi on the bf16 MFMA: ~16 mantissa bits per operand.  Block
                                  intermediates are two blocked bf16 planes (hi, lo); everything else stays f32. */

/* Library / ABI version (major*10000 + minor*100 + patch). */
int lass_version(void);

/* Create a context on HIP device `device_id`.  Builds the FFT twiddle / Hann tables.
 * Replaces: ResUNet30.__init__ (resunet.py:621-637) + torchlibrosa STFT/ISTFT construction (:284-302).
 * LIMIT: the context implements ResUNet30(input_channels=1, output_channels=1, condition_size=512) with
 * target_sources_num = 1 - the only configuration the reference ships (config/audiosep_base.yaml:23-30); resunet.py:425-470,
 * 621-637 is written for target_sources_num x output_channels x 3 mask channels, and the Python mirror raises
 * NotImplementedError for anything else (lass_amd/resunet.py). */
int lass_create(lass_ctx** out, int device_id);
/* Context for the multi-resolution-STFT separator (BASELINE configs[4]; reference intent:
 * models/resunet_with_multistft.py:40-118,137-216 - not runnable as shipped, see DESIGN.md section 9 for the authored
 * spec): `n_windows` periodic-Hann analysis windows (`win_lengths`, e.g. {256, 512, 2048}) at a common n_fft (2048),
 * hop 160; one pre_conv + encoder_block1 per window, pools and skips concatenated on channels, shared trunk; the mask is
 * applied to the `mask_window` branch and the waveform comes from an iSTFT with that window.  Parameter names are the
 * reference module tree's (`base.pre_convs.<w>.*`, `base.encoder_block1s.<w>.conv_block1.*`,
 * `film.encoder_block1s-><w>->conv_block1->beta1.*`, ...).  f32 compute only.  All entry points below work on either
 * kind of context. */
int lass_create_multistft(lass_ctx** out, int device_id, int n_fft, int n_windows, const int* win_lengths,
                          int mask_window);
int lass_destroy(lass_ctx* ctx);

/* Last error message of this context (or of the failed lass_create when ctx == NULL). Never NULL. */
const char* lass_last_error(const lass_ctx* ctx);

/* Upload one tensor of the reference's state_dict (key relative to ss_model, e.g.
 * "base.encoder_block1.conv_block1.conv1.weight", "film.decoder_block3->conv_block2->beta1.bias").
 * `data` may be a host or a device pointer (copied with hipMemcpyDefault); shape/dtype are checked against the
 * architecture.  Unknown keys that a reference checkpoint legitimately carries (base.stft.*, base.istft.*,
 * *.num_batches_tracked, decoder_block*.bn2.*, film.decoder_block*->beta2.*) are accepted and ignored (returns 1).
 * Replaces: nn.Module.load_state_dict as used by utils.load_ss_model (utils.py:356-400). */
int lass_set_param(lass_ctx* ctx, const char* name, const void* data, const int64_t* shape, int ndim, int dtype);

/* Fold BatchNorm (eval mode, eps 1e-5) into per-channel scale/shift tables, concatenate the 32 live FiLM linears into
 * one matrix, re-lay-out conv weights to [cin][tap][cout].  Fails if a required parameter is missing.
 * Must be called again after any lass_set_param.
 * LASS_ERR_STATE when this process already holds a live context of the other kind of the pair {bf16 / bf16x3 compute mode,
 * f32 context routed to wino32.hip by LASS_WINO4 != default}: the two may not run side by side (see lass_separate). */
int lass_finalize(lass_ctx* ctx, int compute_mode);

/* Bytes of workspace `lass_separate` needs for B clips of L samples.
 * Limits: B >= 1, L > n_fft/2, and the largest per-clip tensor - decoder_block6's concat, (32 + 32*n_windows) channels x
 * padded frames x n_fft/2 bins, f32 - below 4 GiB (2 GiB with the direct f32 kernels, LASS_WINO=0): the kernels address
 * one clip's tensors with unsigned 32-bit byte offsets.  ResUNet30: 327 s at 16 kHz; multi-STFT model: 40.9 s at 32 kHz.  Longer clips are an LASS_ERR_ARG here and in lass_separate; the reference's own long-form route,
 * chunk_inference (resunet.py:655-714), stays available. */
int lass_workspace_bytes(const lass_ctx* ctx, int B, int L, size_t* bytes);

/* The hot path.  mixture (B,L) f32, condition (B,512) f32 -> out (B,L) f32.
 * Replaces: ResUNet30.forward(input_dict)["waveform"] (resunet.py:640-653) with mixture/out squeezed of their
 * singleton channel axis.
 * Scheduling: the hipGraph that replays a recurring call runs an even batch of >= 8 clips as two independent
 * half-batches on two branches, so that one half's small launches and launch tails overlap the other half's
 * full-size ones (LASS_SPLIT=0: never; =2: eager launches too, the second half on an internal stream forked from /
 * joined to `stream` by events).  Clips are independent: the results are bit-identical to the unsplit run, and `stream`
 * still orders the whole call.  lass_workspace_bytes already accounts for it; lass_workspace_tensor reports
 * LASS_ERR_STATE while the last call of that shape ran split (its workspace holds the part-batch layouts).
 * CAUTION for callers with kernels of their own (gfx950, measured: DESIGN.md section 5b).  While a bf16-mode lass_separate
 * (LASS_COMPUTE_BF16 / _BF16X3: workgroups feeding v_mfma_f32_32x32x16_bf16 from LDS) runs on one stream, a kernel on ANOTHER
 * stream that executes packed-f32 vector instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 - hipcc's SLP vectoriser
 * forms them from float2-shaped code such as complex butterflies) can return wrong values when it shares a compute unit with
 * them: the x half of a packed result in lanes 48-63 came out wrong from right operands, about one launch in two of an FFT
 * kernel, never NaN.  Every kernel of this library is built without packed f32 (-fno-slp-vectorize, audited in the ISA by the
 * test suite) and lass_finalize refuses the one configuration that would bring its only packed-f32 kernel (wino32.hip, f32
 * contexts behind LASS_WINO4) beside a bf16 context.  Kernels the CALLER launches concurrently are the caller's to build the
 * same way (or to keep on the same stream as the bf16 lass_separate); f32-mode contexts are not affected. */
int lass_separate(lass_ctx* ctx, const float* mixture, const float* condition, float* out, int B, int L,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Graph replay.  The third consecutive lass_separate call with the same pointers and shape is captured into a hipGraph
 * and replayed on the caller's stream from then on (one graph per context; a different combination is captured anew
 * after it, too, has been seen three times in a row).  Nothing else about the call changes; profiled
 * contexts and LASS_GRAPH=0 stay eager.  Returns 1 if replay is enabled, 0 if not; counters are optional outputs. */
int lass_graph_stats(const lass_ctx* ctx, long* captures, long* replays);
/* Turns graph replay off (0: every call launches its kernels eagerly) or back on (cached graphs are replayed again).
 * A measurement switch - bench.py times both forms of the same step - with no effect on results. */
int lass_set_graph_replay(lass_ctx* ctx, int enabled);

/* The same path from a PRECOMPUTED analysis of the mixtures, as the reference's multi-STFT wrapper takes it
 * (resunet_with_multistft.py:233-241: input_dict["stft_mixture_mag" / "_cos" / "_sin"][win]; producer:
 * scripts/precompute_stfts.py:19-58 = lass_stft_components): mag[k] (B,T,n_fft/2+1) for each analysis window k in the
 * context's order, cos / sin of the mask window only.  L = target waveform length, T = 1 + L/160. */
int lass_separate_components(lass_ctx* ctx, const float* const* mag, const float* cos_mask, const float* sin_mask,
                             const float* condition, float* out, int B, int L, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ---- stage entry points (used by the parity tests; same kernels lass_separate launches) --------------------- */

/* Centred STFT (n_fft=win=1024, hop 160, reflect pad, periodic Hann) fused with magnitude/phase.
 * wav (B,L) -> mag, cos, sin (B,T,513), T = 1 + L/160.  Any of mag/cos/sin/real/imag may be NULL.
 * mag = sqrt(max(re^2+im^2, 1e-10)), cos = re/mag, sin = im/mag.
 * Replaces: Base.wav_to_spectrogram_phase (base.py:83-113) + torchlibrosa STFT.forward. */
int lass_stft_magphase(lass_ctx* ctx, const float* wav, int B, int L, float* mag, float* cos_out, float* sin_out,
                       float* real_out, float* imag_out, void* stream);

/* Multi-resolution analysis front end ("next" row f3): n_windows centred STFTs of the same waveforms in one launch
 * (n_fft = win_length in {256,512,1024,2048}, periodic Hann, reflect pad, common hop), each as magnitude / cos / sin
 * with torchlibrosa-magphase semantics (mag = sqrt(re^2+im^2); cos = re/max(mag,1e-10); sin likewise).
 * wav (B,L) -> mag[i], cos_out[i], sin_out[i] (B,T,win_lengths[i]/2+1), T = 1 + L/hop; at most 4 windows.
 * Replaces: calculate_stft_components (scripts/precompute_stfts.py:19-58) looped over stft_win_lengths (:573-590;
 * config stft_win_lengths [256,512,2048], hop 160). */
int lass_multi_stft(lass_ctx* ctx, const float* wav, int B, int L, int hop, int n_windows, const int* win_lengths,
                    float* const* mag, float* const* cos_out, float* const* sin_out, void* stream);

/* `calculate_stft_components(waveform, n_fft, hop, win_length, "hann", True, "reflect")` (scripts/precompute_stfts.py:
 * 19-58) for several win_length at one COMMON n_fft (1024 or 2048; each periodic Hann window zero-padded, centred) in one
 * launch, torchlibrosa-magphase semantics: wav (B,L) -> mag[i], cos_out[i], sin_out[i] (B,T,n_fft/2+1), T = 1 + L/hop.
 * This is the producer of the multi-STFT model's wire format. */
int lass_stft_components(lass_ctx* ctx, const float* wav, int B, int L, int n_fft, int hop, int n_windows,
                         const int* win_lengths, float* const* mag, float* const* cos_out, float* const* sin_out,
                         void* stream);

/* Inverse STFT: real, imag (B,T,513) -> wav (B,L): inverse transforms, overlap-add, division by the window-sum-square
 * envelope and trim fused in one kernel.  frames_ws is unused since 1.1 (no frame scratch exists any more); may be NULL.
 * Replaces: torchlibrosa ISTFT.forward as called at resunet.py:510. */
int lass_istft(lass_ctx* ctx, const float* real, const float* imag, int B, int T, int L, float* wav,
               float* frames_ws, void* stream);
/* The same for n_fft in {1024, 2048} and a synthesis window of win_length <= n_fft (zero-padded, centred), hop 160:
 * real, imag (B,T,n_fft/2+1) -> wav (B,L).  Replaces: ISTFT(n_fft, hop, win_length) of resunet_with_multistft.py:36-44. */
int lass_istft_nfft(lass_ctx* ctx, const float* real, const float* imag, int B, int T, int L, int n_fft, int win_length,
                    float* wav, void* stream);

/* FiLM + BN folding for a batch of conditions: shift (B, n_shift) where n_shift = lass_film_width(ctx); column
 * layout is given by lass_film_offset().  shift[b, off+c] = bn_beta[c] - mean[c]*s[c] + (W_site cond_b + b_site)[c].
 * Replaces: FiLM.forward (resunet.py:59-81) for the 32 sites whose output is read. */
int lass_film(lass_ctx* ctx, const float* condition, int B, float* shift, void* stream);
int lass_film_width(const lass_ctx* ctx);
/* Column offset of a site ("encoder_block1->conv_block1->beta1", "decoder_block2->beta1", ...) or -1. */
int lass_film_offset(const lass_ctx* ctx, const char* site);
/* Raw FiLM vectors (no BN folding): film (B, n_shift) = W cond + b.  For parity tests against the golden film vectors. */
int lass_film_raw(lass_ctx* ctx, const float* condition, int B, float* film, void* stream);

/* One residual block (resunet.py:147-165) by reference module prefix, e.g. "base.encoder_block2.conv_block1" or
 * "base.decoder_block3.conv_block2".  x (B,Cin,H,W) NCHW f32 -> y (B,Cout,H,W).  shift is the lass_film() output for
 * the same batch.  scratch: B*Cout*H*W floats. */
int lass_convblock(lass_ctx* ctx, const char* prefix, const float* x, int B, int H, int W, const float* shift,
                   float* y, float* scratch, void* stream);

/* One encoder block (resunet.py:186-198) by module name, e.g. "base.encoder_block3": the ConvBlockRes above plus
 * F.avg_pool2d with the block's downsample ((2,2); (1,2) for encoder_block6) - fused into conv2's epilogue exactly as
 * lass_separate does it (its own kernel when H is not divisible).  x (B,Cin,H,W) -> y (B,Cout,H,W) and
 * pool (B,Cout,H/dh,W/2).  "base.conv_block7a" (downsample (1,1) = identity) takes pool == NULL. */
int lass_encoder_block(lass_ctx* ctx, const char* name, const float* x, int B, int H, int W, const float* shift, float* y,
                       float* pool, float* scratch, void* stream);

/* STFT + magnitude/phase + the network-input prologue of ResUNet30_Base.forward (resunet.py:533-552): bn0 over
 * frequency, zero-padding of T to a multiple of 32 AFTER bn0, last bin dropped.  wav (B,L) -> x0 (B,Tpad,512) and,
 * where non-NULL, mag/cos/sin (B,T,513).  pre_conv (:555) is applied inside encoder_block1's staging.
 * Multi-STFT context: x0 is (n_windows,B,Tpad,n_fft/2), one slab per analysis window; mag/cos/sin (B,T,n_fft/2+1) are
 * the mask window's. */
int lass_front_end(lass_ctx* ctx, const float* wav, int B, int L, float* mag, float* cos_out, float* sin_out, float* x0,
                   void* stream);

/* Where lass_separate(B, L) leaves a named intermediate inside the caller's workspace (f32 compute mode; in the bf16
 * modes several of these hold blocked bf16 data instead): byte offset, shape (B,C,H,W) and element strides.  Names:
 * "mag" "cos" "sin" "x0" "out_real" "out_imag", "encoder_blockN" (the skip, stored in place inside the decoder's concat
 * buffer), "encoder_blockN.pool", "conv_block7a", "decoder_blockN.up" (transposed-conv half of the concat),
 * "decoder_blockN" (N = 6 is consumed by the fused output head and never written).  For parity tests of the fused
 * stages against the reference's own taps.  Returns LASS_ERR_ARG for an unknown name, and LASS_ERR_STATE when the LAST
 * lass_separate of this (B, L) ran as part-batches (the replayed graph of an even batch >= 8, or LASS_SPLIT=2): the workspace
 * then holds their layouts.  After an eager, unsplit call (the first calls of a key, LASS_GRAPH=0, changing pointers) the taps
 * of any batch size are readable. */
int lass_workspace_tensor(const lass_ctx* ctx, int B, int L, const char* name, size_t* offset, int64_t shape[4],
                          int64_t strides[4]);

/* Transposed conv of a decoder block (resunet.py:254-255): x (B,Cin,h,w) -> y (B,Cout,h*sh,w*sw), with the
 * bn1 + FiLM + leaky-ReLU prologue.  name: "base.decoder_blockN". */
int lass_upconv(lass_ctx* ctx, const char* name, const float* x, int B, int h, int w, const float* shift, float* y,
                void* stream);

/* after_conv (1x1, 32->3, +bias) + complex-mask application (resunet.py:570-574, :469-495):
 * x12 (B,32,Tpad,512), mag/cos/sin (B,T,513) -> out_real, out_imag (B,T,513); bin 512 is exactly 0. */
int lass_mask_apply(lass_ctx* ctx, const float* x12, const float* mag, const float* cos_in, const float* sin_in,
                    int B, int T, int Tpad, float* out_real, float* out_imag, void* stream);

/* Per-clip statistics for SDR / SI-SDR (utils.py:148-200): stats (B,6) f64 =
 * [sum ref^2, sum est^2, sum ref*est, sum (est-ref)^2, sum (a*ref)^2, sum (est-a*ref)^2],
 * a = (eps32 + <ref,est>)/(<ref,ref> + eps32), eps32 = FLT_EPSILON.  dB math is the host's. */
int lass_sdr_stats(lass_ctx* ctx, const float* ref, const float* est, int B, int L, double* stats, void* stream);

/* Evaluator data path ("next" row f1): build B mixtures at given SNRs on the device.
 * source (B,L) in/out, noise (B,L), snr_db (B) -> mixture (B,L):
 *   mixture = source + noise * sqrt(mean(source^2) / 10^(snr/10) / mean(noise^2));
 *   if max|mixture| > 1: source *= 0.9/max and mixture *= 0.9/max   (declipping; source is modified in place).
 * scratch: 4*B doubles.  Replaces: the numpy mixing block of DCASEEvaluator.__call__ (dcase_evaluator.py:77-89). */
int lass_mix_at_snr(lass_ctx* ctx, float* source, const float* noise, const float* snr_db, float* mixture, int B, int L,
                    double* scratch, void* stream);

/* Training-side data path ("next" row f4, mixer half; also the producer stage of the STFT pre-compute,
 * scripts/precompute_stfts.py:352-681): SegmentMixer.__call__ + dynamic_loudnorm (data/waveform_mixers.py:19-92) on the
 * device.  waveforms (B,L) -> mixture (B,L), segment (B,L).  For clip n:
 *   noise = sum_{i=1}^{mix_num[n]-1} 10^(comp_db[n][i-1]/20) * (x_{(n+i)%B} / clamp(sqrt(E_{(n+i)%B} / max(E_n,1e-10)), 0.02, 50));
 *   noise = 10^(noise_db[n]/20) * (noise / clamp(sqrt(E_noise / max(E_n,1e-10)), 0.02, 50));   E = mean(x^2)
 *   mixture = x_n + noise;  if max|mixture| > 1: segment = x_n * 0.9/max, mixture *= 0.9/max, else segment = x_n.
 * The reference draws mix_num = randint(2, max_mix_num) and the dB values = randint(lower_db, higher_db) with Python's
 * `random`; here they are INPUTS (device arrays: mix_num int32 (B), comp_db f32 (B, max_comp), noise_db f32 (B)), so the
 * host decides the draws (lass_amd.waveform_mixers.SegmentMixer draws them in the reference's order).
 * max_comp = max_mix_num - 1 (1 ... 7); scratch: 4*B doubles; the three waveform buffers must be distinct. */
int lass_segment_mix(lass_ctx* ctx, const float* waveforms, int B, int L, const int* mix_num, const float* comp_db, int max_comp,
                     const float* noise_db, float* mixture, float* segment, double* scratch, void* stream);

/* ---- instrumentation ---------------------------------------------------------------------------------------- */

/* When enabled, lass_separate brackets each kernel class with HIP events on `stream` (costs a few us per launch).
 * The event pool (512 pairs, ~12 separations) is created HERE, never inside lass_separate; scopes beyond the pool are
 * not timed, so read (lass_profile_get) or reset at least every few separations while profiling. */
int lass_set_profiling(lass_ctx* ctx, int enabled);
/* After a profiled lass_separate has completed (caller synchronised the stream): number of kernel classes, and for
 * class i its name, accumulated milliseconds and launch count since the last lass_profile_reset. */
int lass_profile_count(const lass_ctx* ctx);
int lass_profile_get(lass_ctx* ctx, int i, const char** name, double* ms, int* launches);
int lass_profile_reset(lass_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LASS_HIP_H */
