"""Thin torch-tensor front end over the C-ABI: owns one `lass_ctx`, hands device pointers and the current HIP stream
to liblass_hip.  torch is plumbing here (device memory, streams); every number is produced by the HIP kernels."""
from __future__ import annotations

from ctypes import byref, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, arch


def _ptr(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """One context per (process, device)."""

    def __init__(self, device="cuda:0", multistft=None):
        """multistft: None = ResUNet30 (models/resunet.py); (n_fft, win_lengths, mask_window) = the multi-resolution-STFT
        separator (lass_create_multistft)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LassError("lass_amd computes on an MI355X only (device must be a cuda/HIP device)")
        if not torch.cuda.is_available():
            raise _lib.LassError("no HIP device visible to torch; lass_amd has no CPU fallback")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        h = c_void_p()
        self.multistft = None
        if multistft is None:
            rc = self.lib.lass_create(byref(h), idx)
        else:
            n_fft, wins, mask_window = int(multistft[0]), [int(w) for w in multistft[1]], int(multistft[2])
            rc = self.lib.lass_create_multistft(byref(h), idx, n_fft, len(wins), (c_int * len(wins))(*wins), mask_window)
            self.multistft = (n_fft, tuple(wins), mask_window)
        if rc < 0:
            raise _lib.LassError(f"lass_create failed ({rc}): {self.lib.lass_last_error(None).decode()}")
        self.ctx = h
        self.n_fft = 1024 if multistft is None else self.multistft[0]
        self.n_branches = 1 if multistft is None else len(self.multistft[1])
        self._ws_buf: Optional[torch.Tensor] = None  # ONE workspace, as large as the largest (B, L) seen so far
        self._ws_last = None                         # (B, L) of the last call that used it (its intermediates live there)
        self.ws_allocations = 0
        self.finalized = False

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.lass_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------------------------
    def set_param(self, name: str, t) -> int:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        t = t.detach()
        if not t.is_floating_point():
            return 1  # num_batches_tracked & co: not used by inference
        t = t.to(torch.float32).contiguous()
        shape = (c_int64 * max(1, t.dim()))(*t.shape)
        if t.is_cuda:  # lass_set_param copies synchronously on the NULL stream: the source must be complete first
            torch.cuda.current_stream(t.device).synchronize()
        rc = self.lib.lass_set_param(self.ctx, name.encode(), _ptr(t), shape, t.dim(), 0)
        self.finalized = False
        return _lib.check(self.ctx, rc, f"lass_set_param({name})")

    COMPUTE_MODES = {"f32": 0, "bf16": 1, "bf16x3": 2}  # include/lass_hip.h: LASS_COMPUTE_*

    def load_state_dict(self, sd: Dict[str, object], compute_dtype: str = "f32"):
        for k, v in sd.items():
            self.set_param(k, v)
        self.finalize(compute_dtype)

    def finalize(self, compute_dtype: str = "f32"):
        """compute_dtype 'f32' (default; exact-f32 MFMA arithmetic) or 'bf16' (BASELINE configs[2]: 3x3 convs on the
        bf16 MFMA, f32 accumulate - reduced precision)."""
        mode = self.COMPUTE_MODES[compute_dtype]
        _lib.check(self.ctx, self.lib.lass_finalize(self.ctx, mode), "lass_finalize")
        self.finalized = True
        self.compute_dtype = compute_dtype

    # ---- hot path --------------------------------------------------------------------------------------------
    def workspace_bytes(self, B: int, L: int) -> int:
        n = c_size_t()
        _lib.check(self.ctx, self.lib.lass_workspace_bytes(self.ctx, B, L, byref(n)), "lass_workspace_bytes")
        return n.value

    def _workspace(self, B: int, L: int) -> torch.Tensor:
        """One buffer keyed on capacity: any (B, L) whose lass_workspace_bytes fits reuses it (lass_separate only needs
        workspace_bytes >= its plan), so the evaluator's ragged last batch neither frees and re-allocates gigabytes nor
        changes the workspace pointer the captured graph of the common batch is keyed on.  It grows when a larger shape
        arrives (the old buffer is released first: workspaces are GBs)."""
        need = self.workspace_bytes(B, L)
        if self._ws_buf is None or self._ws_buf.numel() < need:
            self._ws_buf = None
            self._ws_buf = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.ws_allocations += 1
        self._ws_last = (B, L)
        return self._ws_buf

    def separate(self, mixture: torch.Tensor, condition: torch.Tensor, out: Optional[torch.Tensor] = None):
        """mixture (B,L) f32, condition (B,512) f32 on this device -> (B,L) f32."""
        assert mixture.dim() == 2 and condition.dim() == 2 and condition.shape == (mixture.shape[0], arch_cond())
        mixture = self._dev(mixture)
        condition = self._dev(condition)
        B, L = mixture.shape
        if out is None:
            out = torch.empty_like(mixture)
        elif (out.shape != mixture.shape or out.dtype != torch.float32 or out.device != self.device
              or not out.is_contiguous()):
            raise _lib.LassError(f"out must be a contiguous float32 {tuple(mixture.shape)} tensor on {self.device}")
        ws = self._workspace(B, L)
        rc = self.lib.lass_separate(self.ctx, _ptr(mixture), _ptr(condition), _ptr(out), B, L, _ptr(ws), ws.numel(),
                                    _stream(self.device))
        _lib.check(self.ctx, rc, "lass_separate")
        return out

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self.device:
            raise _lib.LassError(f"tensor on {t.device}, engine on {self.device}")
        if t.dtype != torch.float32:
            raise _lib.LassError("float32 tensors only")
        return t.contiguous()

    # ---- stage entry points (parity tests) -------------------------------------------------------------------
    def stft_magphase(self, wav: torch.Tensor, want_complex: bool = False):
        wav = self._dev(wav)
        B, L = wav.shape
        T = arch.frames_for(L)
        mk = lambda: torch.empty(B, T, arch.N_BINS, dtype=torch.float32, device=self.device)  # noqa: E731
        mag, cos, sin = mk(), mk(), mk()
        re, im = (mk(), mk()) if want_complex else (None, None)
        rc = self.lib.lass_stft_magphase(self.ctx, _ptr(wav), B, L, _ptr(mag), _ptr(cos), _ptr(sin), _ptr(re),
                                         _ptr(im), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_stft_magphase")
        return (mag, cos, sin, re, im) if want_complex else (mag, cos, sin)

    def multi_stft(self, wav: torch.Tensor, win_lengths, hop: int = arch.HOP):
        """(B,L) -> {win: (mag, cos, sin)} each (B,1,T,win//2+1): every analysis window in ONE launch."""
        wav = self._dev(wav)
        B, L = wav.shape
        T = 1 + L // hop
        wins = [int(w) for w in win_lengths]
        n = len(wins)
        outs = {w: tuple(torch.empty(B, 1, T, w // 2 + 1, dtype=torch.float32, device=self.device) for _ in range(3))
                for w in wins}
        arr = lambda i: (c_void_p * n)(*[outs[w][i].data_ptr() for w in wins])  # noqa: E731
        rc = self.lib.lass_multi_stft(self.ctx, _ptr(wav), B, L, hop, n, (c_int * n)(*wins), arr(0), arr(1), arr(2),
                                      _stream(self.device))
        _lib.check(self.ctx, rc, "lass_multi_stft")
        return outs

    def istft(self, real: torch.Tensor, imag: torch.Tensor, length: int):
        real, imag = self._dev(real), self._dev(imag)
        B, T, F = real.shape
        assert F == arch.N_BINS and imag.shape == real.shape
        wav = torch.empty(B, length, dtype=torch.float32, device=self.device)
        rc = self.lib.lass_istft(self.ctx, _ptr(real), _ptr(imag), B, T, length, _ptr(wav), None, _stream(self.device))
        _lib.check(self.ctx, rc, "lass_istft")
        return wav

    def film_width(self) -> int:
        return self.lib.lass_film_width(self.ctx)

    def film_offset(self, site: str) -> int:
        return self.lib.lass_film_offset(self.ctx, site.encode())

    def film(self, cond: torch.Tensor, raw: bool = False) -> torch.Tensor:
        cond = self._dev(cond)
        B = cond.shape[0]
        out = torch.empty(B, self.film_width(), dtype=torch.float32, device=self.device)
        fn = self.lib.lass_film_raw if raw else self.lib.lass_film
        _lib.check(self.ctx, fn(self.ctx, _ptr(cond), B, _ptr(out), _stream(self.device)), "lass_film")
        return out

    def convblock(self, prefix: str, x: torch.Tensor, shift: torch.Tensor, cout: int) -> torch.Tensor:
        x = self._dev(x)
        B, _cin, H, W = x.shape
        y = torch.empty(B, cout, H, W, dtype=torch.float32, device=self.device)
        scratch = torch.empty_like(y)
        rc = self.lib.lass_convblock(self.ctx, prefix.encode(), _ptr(x), B, H, W, _ptr(shift), _ptr(y), _ptr(scratch),
                                     _stream(self.device))
        _lib.check(self.ctx, rc, "lass_convblock")
        return y

    def encoder_block(self, name: str, x: torch.Tensor, shift: torch.Tensor, cout: int, down):
        """One encoder block incl. its (fused) avg-pool: x (B,Cin,H,W) -> (y (B,Cout,H,W), pool or None)."""
        x = self._dev(x)
        B, _cin, H, W = x.shape
        y = torch.empty(B, cout, H, W, dtype=torch.float32, device=self.device)
        pool = (torch.empty(B, cout, H // down[0], W // down[1], dtype=torch.float32, device=self.device)
                if tuple(down) != (1, 1) else None)
        scratch = torch.empty_like(y)
        rc = self.lib.lass_encoder_block(self.ctx, name.encode(), _ptr(x), B, H, W, _ptr(shift), _ptr(y), _ptr(pool),
                                         _ptr(scratch), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_encoder_block")
        return y, pool

    def front_end(self, wav: torch.Tensor, out=None):
        """(B,L) -> (mag, cos, sin (B,T,n_fft/2+1), x0): STFT + bn0 + T-pad + F-crop (resunet.py:533-552).
        x0 is (B,Tpad,512) for ResUNet30 and (n_windows,B,Tpad,1024) for the multi-STFT model (mag/cos/sin are then the
        mask window's).  `out` = a (mag, cos, sin, x0) tuple of an earlier call to write into (no allocation in this call)."""
        wav = self._dev(wav)
        B, L = wav.shape
        T = arch.frames_for(L)
        nb = self.n_fft // 2 + 1
        shape = (B, arch.padded_frames(T), nb - 1)
        x0_shape = (self.n_branches,) + shape if self.multistft else shape
        if out is not None:
            mag, cos, sin, x0 = (self._dev(t) for t in out)
            if not (mag.shape == cos.shape == sin.shape == (B, T, nb) and tuple(x0.shape) == tuple(x0_shape)
                    and all(t.dtype == torch.float32 and t.is_contiguous() for t in (mag, cos, sin, x0))):
                raise ValueError("front_end: `out` does not match this input's shapes")
        else:
            mk = lambda: torch.empty(B, T, nb, dtype=torch.float32, device=self.device)  # noqa: E731
            mag, cos, sin = mk(), mk(), mk()
            x0 = torch.empty(x0_shape, dtype=torch.float32, device=self.device)
        rc = self.lib.lass_front_end(self.ctx, _ptr(wav), B, L, _ptr(mag), _ptr(cos), _ptr(sin), _ptr(x0),
                                     _stream(self.device))
        _lib.check(self.ctx, rc, "lass_front_end")
        return mag, cos, sin, x0

    def stft_components(self, wav: torch.Tensor, n_fft: int, win_lengths, hop: int = arch.HOP):
        """(B,L) -> {win: (mag, cos, sin)} each (B,1,T,n_fft//2+1): `calculate_stft_components` at a COMMON n_fft for
        every win_length (zero-padded windows), one launch."""
        wav = self._dev(wav)
        B, L = wav.shape
        T = 1 + L // hop
        wins = [int(w) for w in win_lengths]
        n = len(wins)
        outs = {w: tuple(torch.empty(B, 1, T, n_fft // 2 + 1, dtype=torch.float32, device=self.device) for _ in range(3))
                for w in wins}
        arr = lambda i: (c_void_p * n)(*[outs[w][i].data_ptr() for w in wins])  # noqa: E731
        rc = self.lib.lass_stft_components(self.ctx, _ptr(wav), B, L, n_fft, hop, n, (c_int * n)(*wins), arr(0), arr(1),
                                           arr(2), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_stft_components")
        return outs

    def istft_nfft(self, real: torch.Tensor, imag: torch.Tensor, length: int, n_fft: int, win_length: int):
        real, imag = self._dev(real), self._dev(imag)
        B, T, F = real.shape
        assert F == n_fft // 2 + 1 and imag.shape == real.shape
        wav = torch.empty(B, length, dtype=torch.float32, device=self.device)
        rc = self.lib.lass_istft_nfft(self.ctx, _ptr(real), _ptr(imag), B, T, length, n_fft, win_length, _ptr(wav),
                                      _stream(self.device))
        _lib.check(self.ctx, rc, "lass_istft_nfft")
        return wav

    def separate_components(self, mags, cos_mask: torch.Tensor, sin_mask: torch.Tensor, condition: torch.Tensor,
                            length: int) -> torch.Tensor:
        """Precomputed analysis in (the multi-STFT wrapper's input form): mags = [ (B,T,F) per analysis window in the
        context's order ], cos / sin of the mask window -> waveform (B, length)."""
        mags = [self._dev(m) for m in mags]
        cos_mask, sin_mask, condition = self._dev(cos_mask), self._dev(sin_mask), self._dev(condition)
        B, T, F = mags[0].shape
        if len(mags) != self.n_branches or any(m.shape != (B, T, F) for m in mags) or F != self.n_fft // 2 + 1 \
                or cos_mask.shape != (B, T, F) or sin_mask.shape != (B, T, F) or T != arch.frames_for(length) \
                or condition.shape != (B, arch_cond()):
            raise _lib.LassError("separate_components: inconsistent shapes")
        out = torch.empty(B, length, dtype=torch.float32, device=self.device)
        ws = self._workspace(B, length)
        arr = (c_void_p * len(mags))(*[m.data_ptr() for m in mags])
        rc = self.lib.lass_separate_components(self.ctx, arr, _ptr(cos_mask), _ptr(sin_mask), _ptr(condition), _ptr(out),
                                               B, length, _ptr(ws), ws.numel(), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_separate_components")
        return out

    def workspace_tensor(self, name: str, B: int, L: int) -> torch.Tensor:
        """View of a named intermediate that the last `separate` of shape (B, L) left in the workspace (f32 mode)."""
        off = c_size_t()
        shape, strides = (c_int64 * 4)(), (c_int64 * 4)()
        rc = self.lib.lass_workspace_tensor(self.ctx, B, L, name.encode(), byref(off), shape, strides)
        if rc == -3:  # LASS_ERR_STATE
            raise _lib.LassError(f"lass_workspace_tensor: the last separation of {B} clips ran as two half-batches (the workspace "
                                 "holds two half-batch layouts): tap after an eager call (set_graph_replay(False)), with "
                                 "LASS_SPLIT=0, or with a batch below 8")
        if rc < 0:
            raise _lib.LassError(f"lass_workspace_tensor: unknown tensor '{name}' or bad shape")
        ws = self._ws_buf if self._ws_last == (B, L) else None
        if ws is None:
            raise _lib.LassError("the workspace does not hold a run of that shape: call separate(B, L) first")
        assert off.value % 4 == 0
        flat = ws.view(torch.float32)
        return flat.as_strided(tuple(shape), tuple(strides), off.value // 4)

    def upconv(self, name: str, x: torch.Tensor, shift: torch.Tensor, cout: int, up) -> torch.Tensor:
        x = self._dev(x)
        B, _cin, h, w = x.shape
        y = torch.empty(B, cout, h * up[0], w * up[1], dtype=torch.float32, device=self.device)
        rc = self.lib.lass_upconv(self.ctx, name.encode(), _ptr(x), B, h, w, _ptr(shift), _ptr(y),
                                  _stream(self.device))
        _lib.check(self.ctx, rc, "lass_upconv")
        return y

    def mask_apply(self, x12, mag, cos, sin):
        x12, mag, cos, sin = map(self._dev, (x12, mag, cos, sin))
        B, C, Tp, F = x12.shape
        T = mag.shape[1]
        assert C == 32 and F == arch.F_CROP
        o_re = torch.empty_like(mag)
        o_im = torch.empty_like(mag)
        rc = self.lib.lass_mask_apply(self.ctx, _ptr(x12), _ptr(mag), _ptr(cos), _ptr(sin), B, T, Tp, _ptr(o_re),
                                      _ptr(o_im), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_mask_apply")
        return o_re, o_im

    def sdr_stats(self, ref: torch.Tensor, est: torch.Tensor) -> torch.Tensor:
        """(B,L),(B,L) f32 -> (B,6) f64 sums (see include/lass_hip.h)."""
        ref, est = self._dev(ref), self._dev(est)
        B, L = ref.shape
        stats = torch.empty(B, 6, dtype=torch.float64, device=self.device)
        rc = self.lib.lass_sdr_stats(self.ctx, _ptr(ref), _ptr(est), B, L, _ptr(stats), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_sdr_stats")
        return stats

    def mix_at_snr(self, source: torch.Tensor, noise: torch.Tensor, snr_db: torch.Tensor, out: Optional[torch.Tensor] = None,
                   scratch: Optional[torch.Tensor] = None):
        """dcase_evaluator.py:77-89 on the device.  source (B,L) is scaled IN PLACE when the mixture needs declipping;
        returns the mixture (B,L) (`out` / `scratch` (B,4) f64: caller-kept buffers, nothing allocated here)."""
        source, noise, snr_db = self._dev(source), self._dev(noise), self._dev(snr_db)
        B, L = source.shape
        assert noise.shape == source.shape and snr_db.shape == (B,)
        mixture = torch.empty_like(source) if out is None else self._dev(out)
        assert mixture.shape == source.shape and mixture.is_contiguous() and mixture.dtype == torch.float32
        if scratch is None:
            scratch = torch.empty(B, 4, dtype=torch.float64, device=self.device)
        assert scratch.dtype == torch.float64 and scratch.numel() >= 4 * B and scratch.is_contiguous()
        rc = self.lib.lass_mix_at_snr(self.ctx, _ptr(source), _ptr(noise), _ptr(snr_db), _ptr(mixture), B, L,
                                      _ptr(scratch), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_mix_at_snr")
        return mixture

    def segment_mix(self, waveforms: torch.Tensor, mix_num: torch.Tensor, comp_db: torch.Tensor, noise_db: torch.Tensor):
        """data/waveform_mixers.py:19-62 on the device with the random draws given: waveforms (B,L), mix_num (B) int32,
        comp_db (B, max_mix_num - 1) f32, noise_db (B) f32 -> (mixture (B,L), segment (B,L))."""
        waveforms = self._dev(waveforms)
        B, L = waveforms.shape
        mix_num = mix_num.to(device=self.device, dtype=torch.int32).contiguous()
        comp_db = comp_db.to(device=self.device, dtype=torch.float32).contiguous()
        noise_db = noise_db.to(device=self.device, dtype=torch.float32).contiguous()
        if mix_num.shape != (B,) or noise_db.shape != (B,) or comp_db.dim() != 2 or comp_db.shape[0] != B:
            raise ValueError("segment_mix: mix_num (B), comp_db (B, max_mix_num - 1), noise_db (B) expected")
        mixture, segment = torch.empty_like(waveforms), torch.empty_like(waveforms)
        scratch = torch.empty(B, 4, dtype=torch.float64, device=self.device)
        rc = self.lib.lass_segment_mix(self.ctx, _ptr(waveforms), B, L, _ptr(mix_num), _ptr(comp_db), int(comp_db.shape[1]),
                                       _ptr(noise_db), _ptr(mixture), _ptr(segment), _ptr(scratch), _stream(self.device))
        _lib.check(self.ctx, rc, "lass_segment_mix")
        return mixture, segment

    def graph_stats(self):
        """(enabled, captures, replays) of the hipGraph replay of lass_separate."""
        from ctypes import c_long
        cap, rep = c_long(), c_long()
        on = self.lib.lass_graph_stats(self.ctx, byref(cap), byref(rep))
        return bool(on), cap.value, rep.value

    def set_graph_replay(self, on: bool):
        """hipGraph replay of lass_separate on / off (off: every call launches eagerly; a measurement switch)."""
        _lib.check(self.ctx, self.lib.lass_set_graph_replay(self.ctx, 1 if on else 0), "lass_set_graph_replay")

    # ---- instrumentation -------------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self.lib.lass_set_profiling(self.ctx, 1 if on else 0)

    def profile(self, reset: bool = True) -> Dict[str, tuple]:
        """{kernel class: (ms, launches)} accumulated since the last reset.  Caller must have synchronised."""
        out = {}
        for i in range(self.lib.lass_profile_count(self.ctx)):
            name, ms, n = c_char_p(), c_double(), c_int()
            self.lib.lass_profile_get(self.ctx, i, byref(name), byref(ms), byref(n))
            out[name.value.decode()] = (ms.value, n.value)
        if reset:
            self.lib.lass_profile_reset(self.ctx)
        return out


def arch_cond() -> int:
    return 512


_ENGINES: Dict[int, Engine] = {}


def get_engine(device) -> Engine:
    """Process-wide engine for a device (weights are per ResUNet30 instance -> each model owns its own Engine;
    this shared one serves weight-free ops: STFT, iSTFT, SDR statistics)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ENGINES:
        _ENGINES[idx] = Engine(torch.device("cuda", idx))
    return _ENGINES[idx]
