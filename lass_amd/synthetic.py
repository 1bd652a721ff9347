"""Seeded synthetic weights, conditions and mixtures.

There is no checkpoint and no audio in the build/test environment, so every parity test, the smoke test and
bench.py regenerate the same tensors from a seed on both sides (HIP path and oracle).  Nothing here is committed
as data.

Weights: conv/linear ~ U(+-xavier bound) (reference init: models/base.py:9-15), but BatchNorm statistics are
randomised (gamma~U(.5,1.5), beta,mean~U(-.1,.1), var~U(.5,1.5)) and biases are non-zero so that BN-folding,
FiLM and bias bugs are visible - the reference's own init (gamma=1, beta=0, mean=0, var=1, bias=0) would hide them.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from . import arch

SEED = 1234  # config/audiosep_base.yaml:48 (random_seed)


def _xavier_bound(shape: Tuple[int, ...], kind: str) -> float:
    if kind == "linear_w":
        fan_out, fan_in = shape
    else:  # conv_w (O,I,kh,kw) / tconv_w (I,O,kh,kw): torch's fan computation treats dim0 as "out", dim1 as "in"
        rf = int(np.prod(shape[2:]))
        fan_out, fan_in = shape[0] * rf, shape[1] * rf
    return math.sqrt(6.0 / (fan_in + fan_out))


def make_state_dict(seed: int = SEED, input_channels: int = 1, output_channels: int = 1,
                    condition_size: int = 512, specs=None) -> Dict[str, np.ndarray]:
    """Seeded numpy state_dict with the reference's keys (float32; num_batches_tracked int64).  `specs` defaults to
    arch.param_specs(...) (ResUNet30); pass arch.ms_param_specs(...) for the multi-STFT model."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    if specs is None:
        specs = arch.param_specs(input_channels, output_channels, condition_size)
    for name, shape, kind in specs:
        if kind in ("conv_w", "tconv_w", "linear_w"):
            b = _xavier_bound(shape, kind)
            v = rng.uniform(-b, b, size=shape)
        elif kind in ("bias", "linear_b", "bn_bias", "bn_mean"):
            v = rng.uniform(-0.1, 0.1, size=shape)
        elif kind in ("bn_weight", "bn_var"):
            v = rng.uniform(0.5, 1.5, size=shape)
        elif kind == "bn_nbt":
            sd[name] = np.asarray(0, dtype=np.int64)
            continue
        else:
            raise ValueError(kind)
        sd[name] = v.astype(np.float32)
    # Centre the mask logits so the synthetic net emits a usable mask (|M|~0.5, phase rotation mostly < 45 deg)
    # instead of a random-phase one whose iSTFT cancels to ~0: keeps SDR / SI-SDR parity checks well conditioned.
    sd["base.after_conv.bias"] = (sd["base.after_conv.bias"]
                                  + np.asarray([1.5, 2.0, -0.2] * output_channels, dtype=np.float32))
    return sd


def make_state_dict_ms(seed: int = SEED + 4, win_lengths=arch.MS_WIN_LENGTHS) -> Dict[str, np.ndarray]:
    """Seeded weights of the multi-STFT separator (arch.ms_param_specs)."""
    return make_state_dict(seed, specs=arch.ms_param_specs(win_lengths=win_lengths))


def make_condition(batch: int, seed: int = SEED, condition_size: int = 512, distinct: bool = True) -> np.ndarray:
    """Unit-norm query embeddings (CLAP text embeddings are L2-normalised: CLAP/open_clip/model.py:750)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    n = batch if distinct else 1
    g = rng.standard_normal((n, condition_size))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    if not distinct:
        g = np.repeat(g, batch, axis=0)
    return g.astype(np.float32)


_SNRS = (-15, -10, -5, 0, 5, 10, 15)


def make_clip(index: int, length: int = 160000, sr: int = 16000, seed: int = SEED):
    """(source, noise, snr_db) for synthetic validation clip `index` (float32, mono, already at `sr`)."""
    rng = np.random.Generator(np.random.PCG64([seed, 7919, index]))
    t = np.arange(length) / sr
    src = np.zeros(length)
    for _ in range(3):
        f = rng.uniform(100.0, 4000.0)
        a = rng.uniform(0.05, 0.3)
        ph = rng.uniform(0, 2 * np.pi)
        src += a * np.sin(2 * np.pi * f * t + ph)
    am = 0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(0.2, 2.0) * t + rng.uniform(0, 2 * np.pi))
    src = src * am + 0.01 * rng.standard_normal(length)
    noise = np.cumsum(rng.standard_normal(length))
    noise -= np.linspace(noise[0], noise[-1], length)          # remove the random-walk drift
    k = 64
    noise = noise - np.convolve(noise, np.ones(k) / k, mode="same")  # pink-ish: kill the sub-audio part
    noise *= 0.1 / max(np.sqrt(np.mean(noise ** 2)), 1e-12)
    return src.astype(np.float32), noise.astype(np.float32), _SNRS[index % len(_SNRS)]


def mix_at_snr(source: np.ndarray, noise: np.ndarray, snr: int):
    """The evaluator's deterministic mixer (dcase_evaluator.py:76-89). Returns (source', mixture)."""
    source = source.copy()
    source_power = np.mean(source ** 2)
    noise_power = np.mean(noise ** 2)
    desired_noise_power = source_power / (10 ** (snr / 10))
    scaling_factor = np.sqrt(desired_noise_power / noise_power)
    noise = noise * scaling_factor
    mixture = source + noise
    max_value = np.max(np.abs(mixture))
    if max_value > 1:
        source *= 0.9 / max_value
        mixture *= 0.9 / max_value
    return source, mixture


def make_mixtures(batch: int, length: int = 160000, first: int = 0, seed: int = SEED):
    """(sources (B,L), mixtures (B,L)) float32."""
    srcs, mixes = [], []
    for i in range(first, first + batch):
        s, n, snr = make_clip(i, length, seed=seed)
        s2, m = mix_at_snr(s, n, snr)
        srcs.append(s2)
        mixes.append(m.astype(np.float32))
    return np.stack(srcs).astype(np.float32), np.stack(mixes).astype(np.float32)


def write_validation_set(root: str, n_clips: int = 8, length: int = 160000, sr: int = 16000, seed: int = SEED) -> str:
    """Write `<root>/lass_validation/*.wav` + `<root>/lass_synthetic_validation.csv` in the DCASE layout
    (header `source,noise,snr,caption`; dcase_evaluator.py:42-47,67-71).  Returns the csv path."""
    import os
    from .wavio import write_wav_f32

    adir = os.path.join(root, "lass_validation")
    os.makedirs(adir, exist_ok=True)
    rows = ["source,noise,snr,caption"]
    for i in range(n_clips):
        s, n, snr = make_clip(i, length, sr, seed)
        write_wav_f32(os.path.join(adir, f"src_{i:04d}.wav"), s, sr)
        write_wav_f32(os.path.join(adir, f"noise_{i:04d}.wav"), n, sr)
        rows.append(f"src_{i:04d},noise_{i:04d},{snr},synthetic tone cluster {i % 4}")
    csv_path = os.path.join(root, "lass_synthetic_validation.csv")
    with open(csv_path, "w") as f:
        f.write("\n".join(rows) + "\n")
    return csv_path
