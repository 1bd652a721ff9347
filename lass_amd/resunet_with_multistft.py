"""Host-side mirror of the multi-resolution-STFT separator (BASELINE configs[4]).

Reference intent: /root/reference/models/resunet_with_multistft.py - `ResUNet30(input_channels, output_channels,
condition_size, win_lengths=(256, 512, 2048))` whose `forward(input_dict, target_waveform)` reads PRECOMPUTED spectra
`input_dict["stft_mixture_mag" | "stft_mixture_cos" | "stft_mixture_sin"][win]` (written by scripts/precompute_stfts.py)
plus `input_dict["condition"]`.  That file does not run as shipped; lass_amd/arch.py ("multi-resolution-STFT
separator") and DESIGN.md §9 state the authored, coherent spec this class and liblass_hip implement: every window analysed
at a common n_fft = 2048, one `pre_convs[w]` + `encoder_block1s[w]` per window, channel-concatenated pools / skips, the
ResUNet30 trunk, mask + iSTFT on the 512 window.  Parity is "unpinned" against the reference (nothing to run); the HIP
path is held to oracle/resunet_multistft.py.

Two input forms:
  * the reference wrapper's: precomputed dicts {win: (B,1,T,1025)} + `target_waveform` (only its length is read);
  * `input_dict["mixture"]` (B,1,L): the analysis runs on the device inside the same call (one launch for all windows).
The module holds parameters only; all arithmetic happens in liblass_hip.  Inference only, f32.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import arch
from ._lib import LassError
from .engine import Engine
from .resunet import ResUNet30 as _ResUNet30


class ResUNet30(_ResUNet30):
    def __init__(self, input_channels: int = 1, output_channels: int = 1, condition_size: int = 512,
                 win_lengths: Sequence[int] = arch.MS_WIN_LENGTHS):
        self.win_lengths = tuple(int(w) for w in win_lengths)
        if arch.MS_MASK_WINDOW not in self.win_lengths:
            raise NotImplementedError("the mask / re-synthesis window (512, resunet_with_multistft.py:36-44,185-188) "
                                      "must be one of win_lengths")
        super().__init__(input_channels, output_channels, condition_size)

    def _param_specs(self):
        return arch.ms_param_specs(self.input_channels, self.output_channels, self.condition_size, self.win_lengths)

    def _film_sites(self):
        return arch.ms_film_sites(self.win_lengths)

    def _make_engine(self, dev) -> Engine:
        return Engine(dev, multistft=(arch.MS_N_FFT, self.win_lengths, arch.MS_MASK_WINDOW))

    @torch.no_grad()
    def forward(self, input_dict: Dict, target_waveform: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """resunet_with_multistft.py:233-241.  Returns {'waveform': (B, L)} for the precomputed form (what the
        reference's base returns, :213-214) and {'waveform': (B, 1, L)} for the 'mixture' form (ResUNet30's shape, so the
        DCASE evaluator can drive either model)."""
        if self.training:
            raise LassError("lass_amd separators are inference-only (call .eval()); training is out of scope")
        if "mixture" in input_dict:
            return {"waveform": self._separate(input_dict["mixture"], input_dict["condition"])}
        if target_waveform is None:
            raise ValueError("the precomputed form needs target_waveform (its last dimension is the output length)")
        eng = self._ensure_engine()
        dev = eng.device

        def plane(d, w):  # (B,1,T,F) [or (B,1,1,T,F) from collation, :155-156] -> (B,T,F) on the device
            t = d[w]
            if t.dim() == 5 and t.shape[1] == 1:
                t = t.squeeze(1)
            return t.to(device=dev, dtype=torch.float32)[:, 0].contiguous()

        nb = arch.MS_N_BINS
        for w in self.win_lengths:  # files written with n_fft = win_length (the reference's own writer) have w//2+1 bins
            got = input_dict["stft_mixture_mag"][w].shape[-1]
            if got != nb:
                raise ValueError(f"window {w}: {got} bins; this model reads spectra at the common n_fft = {arch.MS_N_FFT} "
                                 f"({nb} bins per window) - write them with make_precomputed_items(..., n_fft={arch.MS_N_FFT}) "
                                 "(stft_common_params['n_fft'] records it)")
        mags = [plane(input_dict["stft_mixture_mag"], w) for w in self.win_lengths]
        cos = plane(input_dict["stft_mixture_cos"], arch.MS_MASK_WINDOW)
        sin = plane(input_dict["stft_mixture_sin"], arch.MS_MASK_WINDOW)
        cond = input_dict["condition"].to(device=dev, dtype=torch.float32).contiguous()
        return {"waveform": eng.separate_components(mags, cos, sin, cond, int(target_waveform.shape[-1]))}

    def chunk_inference(self, input_dict, max_batch: int = 4):
        """Overlap-discard stitching exactly as ResUNet30.chunk_inference (resunet.py:655-714) on the 'mixture' form."""
        return super().chunk_inference(input_dict, max_batch=max_batch)
