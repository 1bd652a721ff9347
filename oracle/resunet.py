"""Oracle ResUNet30 forward (TEST INFRASTRUCTURE - see oracle/__init__.py).

Functional torch-CPU restatement of /root/reference/models/resunet.py + models/base.py over an explicit
{state_dict key: tensor} weight dict.  Same op order as the reference (BN, +beta, leaky-ReLU, conv, residual ...),
no fusion, no folding - so it is an independent check of the fused HIP kernels.

Pinned against the reference's own code by tests/golden/*.npz (tools/gen_golden.py), see tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import stft as ostft

# (name, cin, cout, downsample) - resunet.py:315-370
_ENC = (
    ("encoder_block1", 32, 32, (2, 2)),
    ("encoder_block2", 32, 64, (2, 2)),
    ("encoder_block3", 64, 128, (2, 2)),
    ("encoder_block4", 128, 256, (2, 2)),
    ("encoder_block5", 256, 384, (2, 2)),
    ("encoder_block6", 384, 384, (1, 2)),
    ("conv_block7a", 384, 384, (1, 1)),
)
# (name, cin, cout, upsample) - resunet.py:371-418
_DEC = (
    ("decoder_block1", 384, 384, (1, 2)),
    ("decoder_block2", 384, 384, (2, 2)),
    ("decoder_block3", 384, 256, (2, 2)),
    ("decoder_block4", 256, 128, (2, 2)),
    ("decoder_block5", 128, 64, (2, 2)),
    ("decoder_block6", 64, 32, (2, 2)),
)


def to_torch(sd: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def film(sd, cond: torch.Tensor, site: str) -> torch.Tensor:
    """resunet.py:68-81: nn.Linear(cond)[:, :, None, None] for the FiLM module named `site` ('a->b->beta1')."""
    return F.linear(cond, sd[f"film.{site}.weight"], sd[f"film.{site}.bias"])[:, :, None, None]


def _bn(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Eval-mode BatchNorm2d (running statistics), eps 1e-5."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=1e-5)


def conv_block_res(sd, prefix: str, x: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """resunet.py:147-165."""
    h = F.conv2d(F.leaky_relu(_bn(sd, prefix + ".bn1", x) + b1, 0.01), sd[prefix + ".conv1.weight"], padding=1)
    h = F.conv2d(F.leaky_relu(_bn(sd, prefix + ".bn2", h) + b2, 0.01), sd[prefix + ".conv2.weight"], padding=1)
    if (prefix + ".shortcut.weight") in sd:
        return F.conv2d(x, sd[prefix + ".shortcut.weight"], sd[prefix + ".shortcut.bias"]) + h
    return x + h


def base_forward(sd, mixtures: torch.Tensor, cond: torch.Tensor, taps: Optional[dict] = None,
                 stft_form: str = "fft") -> torch.Tensor:
    """ResUNet30_Base.forward (resunet.py:522-595) with FiLM applied from `cond` (resunet.py:640-653).

    mixtures (B,1,L), cond (B,512) -> waveform (B,1,L).  `taps`, if given, receives intermediate tensors.
    """
    assert mixtures.dim() == 3 and mixtures.shape[1] == 1
    x_wav = mixtures[:, 0, :]
    length = x_wav.shape[-1]
    real, imag = (ostft.stft_fft if stft_form == "fft" else ostft.stft_dft)(x_wav)
    mag, cos_in, sin_in = ostft.spectrogram_phase(real, imag, 1e-10)               # base.py:91-113
    x = _bn(sd, "base.bn0", mag.transpose(1, 3)).transpose(1, 3)                   # resunet.py:537-539
    t0 = x.shape[2]
    pad_len = int(np.ceil(t0 / 32)) * 32 - t0
    x = F.pad(x, (0, 0, 0, pad_len))                                               # :548
    x = x[..., 0:x.shape[-1] - 1]                                                  # :552
    if taps is not None:
        taps.update(mag=mag, cos=cos_in, sin=sin_in, x0=x)
    x = F.conv2d(x, sd["base.pre_conv.weight"], sd["base.pre_conv.bias"])          # :555
    if taps is not None:
        taps["pre"] = x
    skips = []
    for name, _cin, _cout, down in _ENC:                                           # :556-562
        p = f"base.{name}.conv_block1"
        enc = conv_block_res(sd, p, x, film(sd, cond, f"{name}->conv_block1->beta1"),
                             film(sd, cond, f"{name}->conv_block1->beta2"))
        x = F.avg_pool2d(enc, kernel_size=down)
        skips.append(enc)
        if taps is not None:
            taps[name] = enc
            taps[name + ".pool"] = x
    skips.pop()                                                                    # conv_block7a's skip is unused
    for name, _cin, _cout, up in _DEC:                                             # :563-568, :240-264
        h = F.leaky_relu(_bn(sd, f"base.{name}.bn1", x) + film(sd, cond, f"{name}->beta1"), 0.01)
        h = F.conv_transpose2d(h, sd[f"base.{name}.conv1.weight"], stride=up)
        if taps is not None:
            taps[name + ".up"] = h
        h = torch.cat((h, skips.pop()), dim=1)
        x = conv_block_res(sd, f"base.{name}.conv_block2", h, film(sd, cond, f"{name}->conv_block2->beta1"),
                           film(sd, cond, f"{name}->conv_block2->beta2"))
        if taps is not None:
            taps[name] = x
    x = F.conv2d(x, sd["base.after_conv.weight"], sd["base.after_conv.bias"])      # :570
    x = F.pad(x, (0, 1))[:, :, 0:t0, :]                                            # :573-574
    if taps is not None:
        taps["logits"] = x
    # feature_maps_to_wav, resunet.py:436-519 (target_sources_num = output_channels = 1, K = 3)
    mask_mag = torch.sigmoid(x[:, 0:1])
    mask_real = torch.tanh(x[:, 1:2])
    mask_imag = torch.tanh(x[:, 2:3])
    _, mask_cos, mask_sin = ostft.magphase(mask_real, mask_imag)
    out_cos = cos_in * mask_cos - sin_in * mask_sin
    out_sin = sin_in * mask_cos + cos_in * mask_sin
    out_mag = F.relu(mag * mask_mag)
    out_real = out_mag * out_cos
    out_imag = out_mag * out_sin
    if taps is not None:
        taps.update(out_real=out_real, out_imag=out_imag)
    wav = (ostft.istft_fft if stft_form == "fft" else ostft.istft_dft)(out_real, out_imag, length)
    return wav[:, None, :]


def forward(sd, input_dict: dict, taps: Optional[dict] = None, stft_form: str = "fft") -> dict:
    """ResUNet30.forward (resunet.py:640-653)."""
    with torch.no_grad():
        return {"waveform": base_forward(sd, input_dict["mixture"], input_dict["condition"], taps, stft_form)}


def film_all(sd, cond: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Every FiLM vector (B,C) keyed by site name, including the 6 unused decoder beta2 (resunet.py:59-81)."""
    out = {}
    for k in sd:
        if k.startswith("film.") and k.endswith(".weight"):
            site = k[len("film."):-len(".weight")]
            out[site] = film(sd, cond, site)[:, :, 0, 0]
    return out


def chunk_inference(sd, input_dict: dict, stft_form: str = "fft") -> np.ndarray:
    """ResUNet30.chunk_inference (resunet.py:655-714): overlap-discard stitching, RATE hard-coded 32000,
    batch 1, returns float64 ndarray (1, L); zeros when L <= WINDOW."""
    mixtures, cond = input_dict["mixture"], input_dict["condition"]
    rate = 32000
    nl, nc, nr = int(1.0 * rate), int(3.0 * rate), int(1.0 * rate)
    length = mixtures.shape[2]
    out_np = np.zeros([1, length])
    window = nl + nc + nr
    idx = 0
    with torch.no_grad():
        while idx + window < length:
            chunk = base_forward(sd, mixtures[:, :, idx:idx + window], cond, None, stft_form)
            c = chunk.squeeze(0).numpy()
            if idx == 0:
                out_np[:, idx:idx + window - nr] = c[:, :-nr]
            else:
                out_np[:, idx + nl:idx + window - nr] = c[:, nl:-nr]
            idx += nc
            if idx < length:
                chunk = base_forward(sd, mixtures[:, :, idx:idx + window], cond, None, stft_form)
                c = chunk.squeeze(0).numpy()
                seg_len = c.shape[1]
                out_np[:, idx + nl:idx + seg_len] = c[:, nl:]
    return out_np
