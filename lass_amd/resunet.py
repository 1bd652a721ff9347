"""Host-side mirror of the reference separator interface (drop-in for /root/reference/models/resunet.py:621-714).

`ResUNet30(input_channels, output_channels, condition_size)` is an nn.Module whose state_dict keys, shapes and init
are the reference's, so reference checkpoints load unchanged; `forward(input_dict) -> {'waveform': (B,1,L)}` and
`chunk_inference(input_dict) -> ndarray (1,L) float64` keep the reference signatures.  The module holds parameters
only - all arithmetic happens in liblass_hip (HIP kernels, include/lass_hip.h).  Inference (eval mode) only:
BatchNorm uses running statistics (dcase_evaluator.py:57).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import arch
from ._lib import LassError
from .engine import Engine

_IGNORED_PREFIXES = ("base.stft.", "base.istft.")  # torchlibrosa's frozen DFT buffers in reference checkpoints


class _Holder(nn.Module):
    """Parameter container; the tree of these reproduces the reference's module names."""


def _init_tensor(kind: str, shape) -> torch.Tensor:
    """Reference initialisation: models/base.py:9-21 (xavier_uniform weights, zero bias; BN gamma=1, beta=0)."""
    if kind in ("conv_w", "tconv_w", "linear_w"):
        t = torch.empty(*shape)
        nn.init.xavier_uniform_(t)
        return t
    if kind in ("bias", "linear_b", "bn_bias", "bn_mean"):
        return torch.zeros(*shape)
    if kind in ("bn_weight", "bn_var"):
        return torch.ones(*shape)
    if kind == "bn_nbt":
        return torch.tensor(0, dtype=torch.long)
    raise ValueError(kind)


class ResUNet30(nn.Module):
    def __init__(self, input_channels: int = 1, output_channels: int = 1, condition_size: int = 512):
        super().__init__()
        if (input_channels, output_channels, condition_size) != (1, 1, 512):
            # config/audiosep_base.yaml:23-30 is the only configuration the reference ships or evaluates
            raise NotImplementedError("lass_amd.ResUNet30 supports input_channels=1, output_channels=1, "
                                      "condition_size=512 (config/audiosep_base.yaml)")
        self.input_channels, self.output_channels, self.condition_size = input_channels, output_channels, condition_size
        self.film_meta = self._film_meta()
        for name, shape, kind in self._param_specs():
            parts = name.split(".")
            mod: nn.Module = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            t = _init_tensor(kind, shape)
            if kind in ("bn_mean", "bn_var", "bn_nbt"):
                mod.register_buffer(parts[-1], t)
            else:
                mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
        self._engine: Optional[Engine] = None
        self._uploaded_sig = None
        self.compute_dtype = "f32"  # "bf16" = BASELINE configs[2] (bf16-MFMA convolutions, reduced precision)
        self.eval()

    # ---- what a variant of the model overrides (lass_amd/resunet_with_multistft.py) ---------------------------------
    def _param_specs(self):
        return arch.param_specs(self.input_channels, self.output_channels, self.condition_size)

    def _film_sites(self):
        return arch.film_sites()

    def _make_engine(self, dev) -> Engine:
        return Engine(dev)

    def _film_meta(self) -> Dict:
        """Nested {module: {'beta1': C, 'beta2': C}} as get_film_meta returns (resunet.py:598-618)."""
        meta: Dict = {}
        for site, c, _used in self._film_sites():
            d = meta
            parts = site.split("->")
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = c
        return meta

    # ---- state handling ------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.startswith(_IGNORED_PREFIXES)}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._uploaded_sig = None
        return r

    def set_compute_dtype(self, compute_dtype: str) -> "ResUNet30":
        if compute_dtype not in Engine.COMPUTE_MODES:
            raise ValueError(f"compute_dtype must be one of {sorted(Engine.COMPUTE_MODES)}")
        self.compute_dtype = compute_dtype
        self._uploaded_sig = None
        return self

    def _signature(self):
        return (self.compute_dtype,) + tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())

    def _device(self) -> torch.device:
        return self.base.after_conv.weight.device

    def _ensure_engine(self) -> Engine:
        dev = self._device()
        if dev.type != "cuda":
            raise LassError("lass_amd.ResUNet30 computes on an MI355X only: move the module with .to('cuda'). "
                            "There is no CPU fallback.")
        if self._engine is None or self._engine.device != dev:
            self._engine = self._make_engine(dev)
            self._uploaded_sig = None
        sig = self._signature()
        if sig != self._uploaded_sig:
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()}, self.compute_dtype)
            self._uploaded_sig = sig
        return self._engine

    @property
    def engine(self) -> Engine:
        return self._ensure_engine()

    # ---- reference interface -------------------------------------------------------------------------------------
    def _separate(self, mixtures: torch.Tensor, conditions: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise LassError("lass_amd.ResUNet30 is inference-only (call .eval()); training is out of scope")
        if mixtures.dim() != 3 or mixtures.shape[1] != 1:
            raise ValueError("mixture must be (batch_size, 1, segment_samples)")
        eng = self._ensure_engine()
        dev = eng.device
        mix = mixtures.to(device=dev, dtype=torch.float32)[:, 0, :].contiguous()
        cond = conditions.to(device=dev, dtype=torch.float32).contiguous()
        return eng.separate(mix, cond)[:, None, :]

    @torch.no_grad()
    def separate_into(self, mixture: torch.Tensor, condition: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """The same call with caller-kept buffers: mixture (B,L), condition (B,512) -> `out` (B,L), all float32 on the model's
        device.  A caller that presents the SAME buffers again (the evaluator's resident batches) reaches lass_separate's
        replayed hipGraph and its two overlapping half-batches; `forward` allocates its output and therefore launches eagerly."""
        if self.training:
            raise LassError("lass_amd.ResUNet30 is inference-only (call .eval()); training is out of scope")
        return self._ensure_engine().separate(mixture, condition, out)

    @torch.no_grad()
    def forward(self, input_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """resunet.py:640-653."""
        return {"waveform": self._separate(input_dict["mixture"], input_dict["condition"])}

    @torch.no_grad()
    def chunk_inference(self, input_dict: Dict[str, torch.Tensor], max_batch: int = 16) -> np.ndarray:
        """resunet.py:655-714, including its quirks: RATE hard-coded to 32000, batch 1, float64 result, zeros when
        the input is not longer than one window.  The reference runs a second, overlapping forward inside each loop
        iteration whose write is overwritten by the next iteration except at the tail; the same writes are made here in
        the same order, so outputs are identical sample for sample.  Execution differs: the windows are independent
        (eval-mode BatchNorm), so every DISTINCT window is separated once, `max_batch` windows per launch, instead of
        two batch-1 forwards per iteration (the second forward of iteration i is the first of iteration i+1)."""
        mixtures, conditions = input_dict["mixture"], input_dict["condition"]
        rate = 32000
        nl, nc, nr = int(1.0 * rate), int(3.0 * rate), int(1.0 * rate)
        length = mixtures.shape[2]
        out_np = np.zeros([1, length])
        window = nl + nc + nr
        # pass 1: the reference's control flow, recording (segment, destination slice, source slice) per write
        writes = []
        idx = 0
        while idx + window < length:
            seg = (idx, idx + window)
            if idx == 0:
                writes.append((seg, slice(idx, idx + window - nr), slice(None, -nr)))
            else:
                writes.append((seg, slice(idx + nl, idx + window - nr), slice(nl, -nr)))
            idx += nc
            if idx < length:
                seg = (idx, min(idx + window, length))
                writes.append((seg, slice(idx + nl, seg[1]), slice(nl, None)))
        # pass 2: separate every distinct segment once, batching segments of equal length
        results: Dict[tuple, np.ndarray] = {}
        by_len: Dict[int, list] = {}
        for seg, _, _ in writes:
            if seg not in results:
                results[seg] = None
                by_len.setdefault(seg[1] - seg[0], []).append(seg)
        for segs in by_len.values():
            for i in range(0, len(segs), max_batch):
                group = segs[i:i + max_batch]
                batch = torch.cat([mixtures[:1, :, a:b] for a, b in group], dim=0)
                sep = self._separate(batch, conditions[:1].expand(len(group), -1)).squeeze(1).cpu().numpy()
                for seg, row in zip(group, sep):
                    results[seg] = row[None, :]
        # pass 3: replay the writes in the reference's order
        for seg, dst, src in writes:
            out_np[:, dst] = results[seg][:, src]
        return out_np
