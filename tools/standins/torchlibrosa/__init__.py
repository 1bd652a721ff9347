"""Stand-in package for the absent third-party `torchlibrosa` (see stft.py). Golden-vector tooling only."""
from . import stft  # noqa: F401
