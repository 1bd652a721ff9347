"""Oracle DCASE evaluator loop (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates /root/reference/dcase_evaluator.py:49-122 over in-memory clips (the reference's `librosa.load` reduces to a
PCM read for audio that is already 16 kHz mono - librosa is absent here, "parity unpinned" at that boundary).
"""
import numpy as np
import torch

from . import metrics, resunet


def mix(source: np.ndarray, noise: np.ndarray, snr: int):
    """dcase_evaluator.py:76-89.  Returns (source, mixture); both rescaled when the mixture clips."""
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


def evaluate(sd, clips, conditions, forward=None):
    """clips: iterable of (source, noise, snr); conditions: (N,512).  Returns
    (mean_sisdr, mean_sdri, mean_sdr), per-clip array (N,3) of [sdr, sdri, sisdr]  (dcase_evaluator.py:91-122).
    `forward(sd, input_dict) -> {'waveform': (1,1,L)}` defaults to the ResUNet30 oracle (pl_model.ss_model of :104)."""
    forward = forward or resunet.forward
    rows = []
    for i, (source, noise, snr) in enumerate(clips):
        source, mixture = mix(source, noise, int(snr))
        sdr_no_sep = metrics.calculate_sdr(ref=source, est=mixture)
        inp = {"mixture": torch.Tensor(mixture)[None, None, :],
               "condition": torch.as_tensor(conditions[i:i + 1])}
        sep = forward(sd, inp)["waveform"].squeeze(0).squeeze(0).numpy()
        sdr = metrics.calculate_sdr(ref=source, est=sep)
        rows.append([sdr, sdr - sdr_no_sep, metrics.calculate_sisdr(ref=source, est=sep)])
    rows = np.asarray(rows, dtype=np.float64)
    return (float(np.mean(rows[:, 2])), float(np.mean(rows[:, 1])), float(np.mean(rows[:, 0]))), rows
