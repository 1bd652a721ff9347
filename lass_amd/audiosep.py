"""`AudioSep`-shaped holder (reference: models/audiosep.py:14-50, :148-154) without Lightning.

The evaluator only touches `.ss_model`, `.query_encoder.get_query_embed(...)`, `.eval()` and `.device`
(dcase_evaluator.py:57-58,93,104).  The reference's `forward` is a stub (`pass`, audiosep.py:49-50); north_star asks
for a `separate()` entry, provided here as a thin alias of `ss_model(input_dict)['waveform']`."""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .resunet import ResUNet30


class PrecomputedQueryEncoder(nn.Module):
    """Query encoder serving precomputed / deterministic 512-d unit-norm embeddings.

    The frozen CLAP text tower stays on PyTorch-ROCm and is out of this package's scope (needs the LAION checkpoint
    and HF weights, both absent offline).  Anything with the same `get_query_embed(modality, text, device)` signature
    (models/clap_encoder.py:93-106) can be passed to AudioSep instead; this class is the stand-in used with synthetic
    data: a caption maps to a fixed unit vector (a table entry if given, else seeded from the caption's SHA-256)."""

    encoder_type = "precomputed"

    def __init__(self, table: Optional[Dict[str, np.ndarray]] = None, dim: int = 512):
        super().__init__()
        self.table = dict(table or {})
        self.dim = dim

    def _embed_one(self, caption: str) -> np.ndarray:
        v = self.table.get(caption)
        if v is None:
            seed = int.from_bytes(hashlib.sha256(caption.encode()).digest()[:8], "little")
            g = np.random.Generator(np.random.PCG64(seed)).standard_normal(self.dim)
            v = (g / np.linalg.norm(g)).astype(np.float32)
            self.table[caption] = v
        return v

    def get_query_embed(self, modality="text", text: Optional[List[str]] = None, audio=None, use_text_ratio=1.0,
                        device=None) -> torch.Tensor:
        if modality != "text" or text is None:
            raise NotImplementedError("only modality='text' is served by PrecomputedQueryEncoder")
        e = torch.from_numpy(np.stack([self._embed_one(t) for t in text]))
        return e.to(device) if device is not None else e


class AudioSep(nn.Module):
    def __init__(self, ss_model: nn.Module = None, waveform_mixer=None, query_encoder: nn.Module = None,
                 loss_function=None, optimizer_type: str = None, learning_rate: float = None, lr_lambda_func=None,
                 use_text_ratio: float = 1.0):
        super().__init__()
        self.ss_model = ss_model
        self.waveform_mixer = waveform_mixer
        self.query_encoder = query_encoder if query_encoder is not None else PrecomputedQueryEncoder()
        self.query_encoder_type = getattr(self.query_encoder, "encoder_type", "unknown")
        self.use_text_ratio = use_text_ratio
        self.loss_function = loss_function
        self.optimizer_type = optimizer_type
        self.learning_rate = learning_rate
        self.lr_lambda_func = lr_lambda_func

    @property
    def device(self) -> torch.device:
        return next(self.ss_model.parameters()).device

    @torch.no_grad()
    def separate(self, mixture: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        """mixture (B,1,L), condition (B,512) -> waveform (B,1,L)."""
        return self.ss_model({"mixture": mixture, "condition": condition})["waveform"]

    def forward(self, mixture: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        return self.separate(mixture, condition)


def get_model_class(model_type: str):
    """models/audiosep.py:148-154."""
    if model_type == "ResUNet30":
        return ResUNet30
    raise NotImplementedError
