"""Builder-written stand-in for `torchlibrosa.stft` (torchlibrosa==0.1.0, environment.yml:306 of the reference).

torchlibrosa is a third-party dependency of the reference that is NOT vendored under /root/reference and is NOT
installed in this image (no network).  This file exists ONLY so that tools/gen_golden.py can import the reference's
own `models/resunet.py` unmodified in the build container; it is never imported by the product, by bench.py or by
any test that runs on the GPU box.

It restates the library's published semantics in the library's own formulation (conv1d with a windowed DFT matrix,
conv1d with a windowed inverse-DFT matrix + `fold` overlap-add), so golden vectors carry the same kind of float32
rounding the real library produces.  Because the real library could not be run here, results at THIS boundary are
"parity unpinned" against torchlibrosa itself; they are cross-checked against torch.stft / torch.istft instead
(tests/test_oracle_stft.py).

Semantics restated:
  STFT(n_fft, hop, win, 'hann', center=True, pad_mode='reflect'):
      x -> reflect-pad n_fft//2 both sides -> real = conv1d(x, Re(W*w)), imag = conv1d(x, Im(W*w)), stride=hop,
      W[n,k] = exp(-2*pi*i*n*k/N), w = periodic Hann (scipy get_window('hann', N, fftbins=True));
      outputs (B, 1, T, n_fft//2+1).
  magphase(real, imag): mag = sqrt(re^2+im^2); cos = re/clamp(mag,1e-10); sin = im/clamp(mag,1e-10).
  ISTFT: Hermitian-extend to n_fft bins; s = conv1x1(Re(V*w)) @ re - conv1x1(Im(V*w)) @ im with V = exp(+2*pi*i*n*k/N)/N;
      fold overlap-add (hop); divide by clamp(sum_t w^2[n - hop*t], 1e-11); trim [n_fft//2 : n_fft//2 + length].
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def magphase(real, imag):
    mag = (real ** 2 + imag ** 2) ** 0.5
    cos = real / torch.clamp(mag, 1e-10, np.inf)
    sin = imag / torch.clamp(mag, 1e-10, np.inf)
    return mag, cos, sin


class STFT(nn.Module):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                 pad_mode="reflect", freeze_parameters=True):
        super().__init__()
        assert window == "hann" and pad_mode in ("reflect", "constant")
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 4
        self.center = center
        self.pad_mode = pad_mode
        w = _hann_periodic(self.win_length)
        lpad = (n_fft - self.win_length) // 2
        w = np.pad(w, (lpad, n_fft - self.win_length - lpad))
        n = np.arange(n_fft)
        W = np.exp(-2j * np.pi * np.outer(n, n) / n_fft)
        out_channels = n_fft // 2 + 1
        self.conv_real = nn.Conv1d(1, out_channels, n_fft, stride=self.hop_length, padding=0, bias=False)
        self.conv_imag = nn.Conv1d(1, out_channels, n_fft, stride=self.hop_length, padding=0, bias=False)
        self.conv_real.weight.data = torch.Tensor(np.real(W[:, 0:out_channels] * w[:, None]).T)[:, None, :]
        self.conv_imag.weight.data = torch.Tensor(np.imag(W[:, 0:out_channels] * w[:, None]).T)[:, None, :]
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        x = input[:, None, :]
        if self.center:
            x = F.pad(x, pad=(self.n_fft // 2, self.n_fft // 2), mode=self.pad_mode)
        real = self.conv_real(x)
        imag = self.conv_imag(x)
        real = real[:, None, :, :].transpose(2, 3)
        imag = imag[:, None, :, :].transpose(2, 3)
        return real, imag


class ISTFT(nn.Module):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                 pad_mode="reflect", freeze_parameters=True, onnx=False, frames_num=None, device=None):
        super().__init__()
        assert window == "hann"
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 4
        self.center = center
        w = _hann_periodic(self.win_length)
        lpad = (n_fft - self.win_length) // 2
        w = np.pad(w, (lpad, n_fft - self.win_length - lpad))
        self._w = w
        n = np.arange(n_fft)
        V = np.exp(2j * np.pi * np.outer(n, n) / n_fft) / n_fft
        self.conv_real = nn.Conv1d(n_fft, n_fft, kernel_size=1, bias=False)
        self.conv_imag = nn.Conv1d(n_fft, n_fft, kernel_size=1, bias=False)
        self.conv_real.weight.data = torch.Tensor(np.real(V * w[None, :]).T)[:, :, None]
        self.conv_imag.weight.data = torch.Tensor(np.imag(V * w[None, :]).T)[:, :, None]
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def _window_sumsquare(self, frames_num, device):
        n = self.n_fft + self.hop_length * (frames_num - 1)
        x = np.zeros(n, dtype=np.float32)
        wsq = (self._w ** 2).astype(np.float32)
        for i in range(frames_num):
            s = i * self.hop_length
            x[s:min(n, s + self.n_fft)] += wsq[:max(0, min(self.n_fft, n - s))]
        return torch.from_numpy(np.clip(x, 1e-11, np.inf)).to(device)

    def forward(self, real_stft, imag_stft, length):
        assert real_stft.ndimension() == 4 and imag_stft.ndimension() == 4
        real_stft = real_stft[:, 0, :, :].transpose(1, 2)
        imag_stft = imag_stft[:, 0, :, :].transpose(1, 2)
        full_real = torch.cat((real_stft, torch.flip(real_stft[:, 1:-1, :], dims=[1])), dim=1)
        full_imag = torch.cat((imag_stft, -torch.flip(imag_stft[:, 1:-1, :], dims=[1])), dim=1)
        s_real = self.conv_real(full_real) - self.conv_imag(full_imag)
        frames_num = s_real.shape[-1]
        output_samples = (frames_num - 1) * self.hop_length + self.win_length
        y = F.fold(input=s_real, output_size=(1, output_samples), kernel_size=(1, self.win_length),
                   stride=(1, self.hop_length))
        y = y[:, 0, 0, :]
        y = y / self._window_sumsquare(frames_num, y.device)[None, 0:y.shape[1]]
        if length is None:
            if self.center:
                y = y[:, self.n_fft // 2: -self.n_fft // 2]
        else:
            start = self.n_fft // 2 if self.center else 0
            y = y[:, start:start + length]
            if y.shape[-1] < length:
                y = F.pad(y, (0, length - y.shape[-1]))
        return y
