"""`SegmentMixer` with the reference's interface (data/waveform_mixers.py:9-62; SURVEY §8 row f4, mixer half) on the device.

Same constructor, same `__call__(waveforms) -> (mixture, segment)`; the arithmetic (energy matching with the ratio clamped to
[0.02, 50], integer-dB gains, the second `dynamic_loudnorm` over the summed noise, declipping to a 0.9 peak) runs in
`lass_segment_mix` (lass_amd/csrc/misc.hip).  What stays on the host is the reference's use of Python's `random`: the draws are
made here IN THE REFERENCE'S ORDER - per clip `randint(2, max_mix_num)`, then one `randint(lower_db, higher_db)` per mixed-in
clip, then one for the noise sum (waveform_mixers.py:34-44,88) - so a run under `random.seed(s)` consumes the generator exactly
as the reference's loop does, and are handed to the kernel as arrays (`draw()` / `mix_with_draws()` expose the two halves; the
oracle, oracle/waveform_mixers.py, takes the same arrays).  The reference module cannot be imported here (it imports
`pyloudnorm`, absent) and ships no vectors: parity of this path is against the restatement only ("parity unpinned").
"""
from __future__ import annotations

import random
from typing import Tuple

import numpy as np
import torch

from ._lib import LassError
from .engine import get_engine


class SegmentMixer:
    def __init__(self, max_mix_num, lower_db, higher_db):
        if not 2 <= int(max_mix_num) <= 8:
            raise NotImplementedError("max_mix_num must be 2 ... 8 (config/audiosep_base.yaml: 2)")
        self.max_mix_num = int(max_mix_num)
        self.loudness_param = {"lower_db": lower_db, "higher_db": higher_db}

    def draw(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """The random integers of one `__call__`, drawn from Python's global `random` in the reference's order.
        -> mix_num (B) int32, comp_db (B, max_mix_num - 1) float32 (unused entries 0), noise_db (B) float32."""
        lo, hi = self.loudness_param["lower_db"], self.loudness_param["higher_db"]
        mix_num = np.zeros(batch_size, dtype=np.int32)
        comp_db = np.zeros((batch_size, self.max_mix_num - 1), dtype=np.float32)
        noise_db = np.zeros(batch_size, dtype=np.float32)
        for n in range(batch_size):
            mix_num[n] = random.randint(2, self.max_mix_num)
            for i in range(1, mix_num[n]):
                comp_db[n, i - 1] = random.randint(lo, hi)
            noise_db[n] = random.randint(lo, hi)
        return mix_num, comp_db, noise_db

    def mix_with_draws(self, waveforms: torch.Tensor, mix_num, comp_db, noise_db):
        """waveforms (B, L) or (B, C=1, L) on the device -> (mixture, segment) of the same shape."""
        if waveforms.device.type != "cuda":
            raise LassError("lass_amd computes on an MI355X only: move the waveforms to 'cuda' (no CPU fallback)")
        shape = waveforms.shape
        if waveforms.dim() == 3:
            if shape[1] != 1:
                raise NotImplementedError("mono segments only (the reference's datasets yield (B, 1, L))")
            x = waveforms[:, 0, :]
        elif waveforms.dim() == 2:
            x = waveforms
        else:
            raise ValueError("waveforms must be (batch, time) or (batch, 1, time)")
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))  # noqa: E731
        mixture, segment = get_engine(x.device).segment_mix(x.float().contiguous(), as_t(mix_num), as_t(comp_db), as_t(noise_db))
        return mixture.view(shape), segment.view(shape)

    def __call__(self, waveforms: torch.Tensor):
        return self.mix_with_draws(waveforms, *self.draw(waveforms.shape[0]))
