"""Oracle SegmentMixer (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates /root/reference/data/waveform_mixers.py:19-92 (`SegmentMixer.__call__`, `dynamic_loudnorm`, `rescale_to_match_energy`,
`get_energy_ratio`, `get_energy`) in torch-CPU float32, same operations in the same order, with the integers the reference draws
from Python's `random` (:34, :88) passed in as arrays so that both sides mix with the same draws.

PARITY UNPINNED: the reference module imports `pyloudnorm` (absent here, no network) and the reference ships no vectors for it,
so this restatement is checked against the source text only.  `draws_like_reference` replays the reference's draw ORDER.
"""
import random

import numpy as np
import torch


def get_energy(x):                                   # waveform_mixers.py:71-72
    return torch.mean(x ** 2)


def get_energy_ratio(segment1, segment2):            # :75-81
    energy1 = get_energy(segment1)
    energy2 = max(get_energy(segment2), 1e-10)
    ratio = (energy1 / energy2) ** 0.5
    return torch.clamp(ratio, 0.02, 50)


def dynamic_loudnorm(audio, reference, delta_loudness):   # :84-92, the randint(lower_db, higher_db) of :88 given
    rescaled_audio = audio / get_energy_ratio(audio, reference)   # rescale_to_match_energy, :64-68
    gain = np.power(10.0, delta_loudness / 20.0)
    return gain * rescaled_audio


def segment_mix(waveforms: torch.Tensor, mix_num, comp_db, noise_db):
    """waveforms (B, ...) float32 -> (mixture, segment) (:19-62)."""
    batch_size = waveforms.shape[0]
    segs, mixes = [], []
    for n in range(batch_size):
        segment = waveforms[n].clone()
        noise = torch.zeros_like(segment)
        assert int(mix_num[n]) >= 2
        for i in range(1, int(mix_num[n])):
            next_segment = waveforms[(n + i) % batch_size]
            noise += dynamic_loudnorm(next_segment, segment, int(comp_db[n][i - 1]))
        noise = dynamic_loudnorm(noise, segment, int(noise_db[n]))
        mixture = segment + noise
        max_value = torch.max(torch.abs(mixture))
        if max_value > 1:
            segment *= 0.9 / max_value
            mixture *= 0.9 / max_value
        segs.append(segment)
        mixes.append(mixture)
    return torch.stack(mixes, dim=0), torch.stack(segs, dim=0)


def draws_like_reference(batch_size, max_mix_num, lower_db, higher_db):
    """The calls the reference's loop makes on `random` for one batch, in its order (:34, then :88 once per component at :39,
    then :88 for the noise sum at :43)."""
    mix_num = np.zeros(batch_size, dtype=np.int32)
    comp_db = np.zeros((batch_size, max_mix_num - 1), dtype=np.float32)
    noise_db = np.zeros(batch_size, dtype=np.float32)
    for n in range(batch_size):
        mix_num[n] = random.randint(2, max_mix_num)
        for i in range(1, mix_num[n]):
            comp_db[n, i - 1] = random.randint(lower_db, higher_db)
        noise_db[n] = random.randint(lower_db, higher_db)
    return mix_num, comp_db, noise_db
