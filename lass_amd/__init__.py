"""lass_amd - MI355X-native (gfx950) text-conditioned separation hot path: STFT -> FiLM ResUNet30 -> mask -> iSTFT.

Host side mirrors the reference's operator interface (`ResUNet30(input_dict)['waveform']`, `chunk_inference`,
`DCASEEvaluator`, `calculate_sdr/sisdr`); all arithmetic runs in hand-written HIP kernels behind the C-ABI declared
in include/lass_hip.h.  There is no CPU fallback: importing works anywhere, computing requires the built extension
and a GPU.
"""
from . import arch  # noqa: F401

__all__ = ["arch"]
