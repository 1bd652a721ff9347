"""Pin oracle/stft.py: the STFT / iSTFT conventions restated from torchlibrosa 0.1.0 ("parity unpinned" against the
absent library) must equal torch.stft / torch.istft exactly, the DFT-matrix and FFT formulations must agree, and
STFT -> iSTFT must be the identity.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import metrics as om
from oracle import stft as ost


@pytest.mark.parametrize("L", [16000, 8077, 160000])
def test_stft_equals_torch_stft(L):
    x = torch.randn(2, L, dtype=torch.float64, generator=torch.Generator().manual_seed(L))
    re, im = ost.stft_fft(x)
    ts = torch.stft(x, 1024, 160, 1024, torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True,
                    pad_mode="reflect", normalized=False, onesided=True, return_complex=True).transpose(1, 2)
    assert re.shape == (2, 1, 1 + L // 160, 513)
    assert float((ts.real - re[:, 0]).abs().max()) < 1e-10 and float((ts.imag - im[:, 0]).abs().max()) < 1e-10
    if L <= 16000:
        r2, i2 = ost.stft_dft(x)
        assert float((r2 - re).abs().max()) < 1e-9 and float((i2 - im).abs().max()) < 1e-9


@pytest.mark.parametrize("L", [16000, 8077])
def test_istft_equals_torch_istft_and_roundtrip(L):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, L, dtype=torch.float64, generator=g)
    re, im = ost.stft_fft(x)
    assert float((ost.istft_fft(re, im, L) - x).abs().max()) < 1e-12
    assert float((ost.istft_dft(re, im, L) - x).abs().max()) < 1e-10
    T = 1 + L // 160
    sr, si = torch.randn(2, 1, T, 513, dtype=torch.float64, generator=g), torch.randn(2, 1, T, 513, dtype=torch.float64, generator=g)
    # torch.istft requires real DC/Nyquist; torchlibrosa's conv form ignores their imaginary parts - same thing here
    si0 = si.clone()
    si0[..., 0] = 0
    si0[..., 512] = 0
    y = ost.istft_fft(sr, si, L)
    yt = torch.istft(torch.complex(sr[:, 0], si0[:, 0]).transpose(1, 2), 1024, 160, 1024,
                     torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True, length=L)
    assert float((y - yt).abs().max()) < 1e-10
    assert float((ost.istft_dft(sr, si, L) - y).abs().max()) < 1e-10


def test_magphase_clamps():
    z = torch.zeros(1, 1, 2, 3)
    mag, c, s = ost.spectrogram_phase(z, z, 1e-10)           # clamp on |X|^2  (base.py:85)
    assert torch.allclose(mag, torch.full_like(mag, 1e-5)) and not c.any() and not s.any()
    mag, c, s = ost.magphase(z, z)                            # clamp on |M|    (torchlibrosa magphase)
    assert not mag.any() and not c.any() and not s.any()
    mag, c, s = ost.magphase(torch.tensor([3.0]), torch.tensor([4.0]))
    assert torch.allclose(mag, torch.tensor([5.0])) and torch.allclose(c, torch.tensor([0.6])) and torch.allclose(s, torch.tensor([0.8]))


def test_sdr_closed_forms():
    rng = np.random.default_rng(0)
    ref = rng.standard_normal(16000).astype(np.float32)
    assert abs(om.calculate_sdr(ref, 0.5 * ref) - 20 * np.log10(2)) < 1e-4
    assert om.calculate_sdr(ref, ref) == pytest.approx(10 * np.log10(np.mean(ref ** 2) / 1e-10), abs=1e-3)
    noise = rng.standard_normal(16000).astype(np.float32)
    noise -= (noise @ ref) / (ref @ ref) * ref                     # orthogonal to ref
    noise *= np.sqrt((ref @ ref) / (noise @ noise)) * 10 ** (-10 / 20)   # -10 dB relative power
    assert abs(om.calculate_sisdr(ref, 0.3 * ref + 0.3 * noise) - 10.0) < 1e-2     # scale-invariant
    assert abs(om.calculate_sdr(ref, ref + noise) - 10.0) < 1e-2
