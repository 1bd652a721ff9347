"""Oracle STFT / iSTFT / magphase (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates torchlibrosa==0.1.0 (`environment.yml:306`; call sites `models/resunet.py:284-302,473,510`,
`models/base.py:83-88`).  The library is absent from /root/reference and from this image, so this restatement is
"parity unpinned" against the library; it is pinned to torch.stft / torch.istft (tests/test_oracle_stft.py).

Two formulations are provided:
  *_dft : the library's own formulation - multiply by the windowed DFT / inverse-DFT matrix (float64 tables,
          computed in the working dtype), overlap-add, divide by the window-sum-square envelope.
  *_fft : torch.fft based, same conventions; used where speed matters (10 s clips, CPU baseline).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

N_FFT = 1024
HOP = 160


def hann_periodic(n: int = N_FFT, dtype=torch.float64) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * np.pi * k / n)).to(dtype)


def frame(x: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP) -> torch.Tensor:
    """(B, L) -> (B, T, n_fft) centred frames with reflect padding (STFT center=True, pad_mode='reflect')."""
    xp = F.pad(x[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")[:, 0, :]
    return xp.unfold(-1, n_fft, hop)


def stft_dft(x: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP):
    """(B, L) -> real, imag (B, 1, T, n_fft//2+1); X[k] = sum_n w[n] x[n] exp(-2*pi*i*n*k/N)."""
    fr = frame(x, n_fft, hop)
    n = torch.arange(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
    ang = 2.0 * np.pi * torch.outer(n, k) / n_fft
    w = hann_periodic(n_fft)
    wr = (torch.cos(ang) * w[:, None]).to(x.dtype)
    wi = (-torch.sin(ang) * w[:, None]).to(x.dtype)
    return (fr @ wr)[:, None], (fr @ wi)[:, None]


def stft_fft(x: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP):
    fr = frame(x, n_fft, hop) * hann_periodic(n_fft, x.dtype)
    s = torch.fft.rfft(fr, dim=-1)
    return s.real[:, None].contiguous(), s.imag[:, None].contiguous()


def spectrogram_phase(real: torch.Tensor, imag: torch.Tensor, eps: float = 1e-10):
    """models/base.py:83-88 with eps from :91 - the clamp is on |X|^2 (NOT on |X|)."""
    mag = torch.clamp(real ** 2 + imag ** 2, eps, np.inf) ** 0.5
    return mag, real / mag, imag / mag


def magphase(real: torch.Tensor, imag: torch.Tensor):
    """torchlibrosa.stft.magphase (used on the mask, resunet.py:473) - the clamp is on |M| at 1e-10."""
    mag = (real ** 2 + imag ** 2) ** 0.5
    den = torch.clamp(mag, 1e-10, np.inf)
    return mag, real / den, imag / den


def stft_components(x: torch.Tensor, n_fft: int, hop: int = HOP):
    """scripts/precompute_stfts.py:19-58 (`calculate_stft_components`): torchlibrosa STFT(n_fft = win_length, periodic
    Hann, centre, reflect) followed by torchlibrosa `magphase` -> (mag, cos, sin), each (B, 1, T, n_fft//2+1)."""
    return magphase(*stft_fft(x, n_fft, hop))


def ola_envelope(frames: int, n_fft: int = N_FFT, hop: int = HOP, dtype=torch.float64) -> torch.Tensor:
    """sum_t w^2[n - hop*t], clamped at 1e-11 (librosa window_sumsquare as used by torchlibrosa ISTFT)."""
    n = n_fft + hop * (frames - 1)
    wsq = hann_periodic(n_fft) ** 2
    env = torch.zeros(n, dtype=torch.float64)
    idx = (torch.arange(frames)[:, None] * hop + torch.arange(n_fft)[None, :]).reshape(-1)
    env.index_add_(0, idx, wsq.repeat(frames))
    return torch.clamp(env, min=1e-11).to(dtype)


def _overlap_add(fr: torch.Tensor, hop: int) -> torch.Tensor:
    b, t, n_fft = fr.shape
    n = n_fft + hop * (t - 1)
    y = F.fold(fr.transpose(1, 2), output_size=(1, n), kernel_size=(1, n_fft), stride=(1, hop))
    return y[:, 0, 0, :]


def _finish(fr: torch.Tensor, length: int, n_fft: int, hop: int) -> torch.Tensor:
    y = _overlap_add(fr, hop)
    y = y / ola_envelope(fr.shape[1], n_fft, hop, fr.dtype)[None, :]
    y = y[:, n_fft // 2: n_fft // 2 + length]
    if y.shape[-1] < length:
        y = F.pad(y, (0, length - y.shape[-1]))
    return y


def istft_dft(real: torch.Tensor, imag: torch.Tensor, length: int, n_fft: int = N_FFT, hop: int = HOP):
    """(B,1,T,F)x2 -> (B, length): Hermitian extension, windowed inverse DFT matrix, OLA, envelope, trim."""
    re, im = real[:, 0], imag[:, 0]                                   # (B,T,F)
    full_re = torch.cat((re, torch.flip(re[..., 1:-1], dims=[-1])), dim=-1)
    full_im = torch.cat((im, -torch.flip(im[..., 1:-1], dims=[-1])), dim=-1)
    n = torch.arange(n_fft, dtype=torch.float64)
    ang = 2.0 * np.pi * torch.outer(n, n) / n_fft                     # [k, n]
    w = hann_periodic(n_fft)
    vr = (torch.cos(ang) / n_fft * w[None, :]).to(re.dtype)
    vi = (torch.sin(ang) / n_fft * w[None, :]).to(re.dtype)
    fr = full_re @ vr - full_im @ vi                                   # (B,T,n_fft)
    return _finish(fr, length, n_fft, hop)


def istft_fft(real: torch.Tensor, imag: torch.Tensor, length: int, n_fft: int = N_FFT, hop: int = HOP):
    spec = torch.complex(real[:, 0], imag[:, 0])
    fr = torch.fft.irfft(spec, n=n_fft, dim=-1) * hann_periodic(n_fft, real.dtype)
    return _finish(fr, length, n_fft, hop)
