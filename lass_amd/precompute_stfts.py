"""Multi-resolution STFT front end and the precomputed-STFT wire format.

Host-side mirror of the reference's pre-compute path:
  * `calculate_stft_components`      scripts/precompute_stfts.py:19-58   (one window; same signature and outputs)
  * `multi_resolution_stfts`         the loop at scripts/precompute_stfts.py:573-590, as ONE kernel launch
  * `make_precomputed_items`         the per-item dict at scripts/precompute_stfts.py:596-622
  * `save_batch_precomputed_data`    scripts/precompute_stfts.py:60-122  (`batch_%06d.pt`: a list of dicts)
  * `PrecomputedSTFTDataset`         data/precomputed_stft_dataset.py:7-134 (interface: numeric shard order, global index)

All spectra are computed by liblass_hip (`lass_multi_stft`, lass_amd/csrc/stft.hip); there is no CPU fallback.
The consumer of these files in the reference (`models/resunet_with_multistft.py`) is not runnable as shipped
(SURVEY §2a): with n_fft = win_length its three branches have 129 / 257 / 1025 bins and cannot be concatenated.  The
consumer built here (lass_amd/resunet_with_multistft.py, authored spec in DESIGN.md §9) reads the same wire format
written with `n_fft=2048` for every window, which `calculate_stft_components(..., n_fft=2048, win_length=w, ...)` -
a legal call of the reference's own function - produces.
"""
from __future__ import annotations

import bisect
import pathlib
from typing import Any, Dict, List, Optional, Sequence

import torch

from ._lib import LassError
from .engine import get_engine

SUPPORTED_WINDOWS = (256, 512, 1024, 2048)


COMMON_N_FFT = (1024, 2048)  # n_fft > win_length: every window zero-padded, centred, to one transform size


def _check_cfg(n_fft, win_length, window, center, pad_mode):
    if n_fft != win_length and not (n_fft in COMMON_N_FFT and win_length < n_fft):
        raise NotImplementedError("n_fft must equal win_length (scripts/precompute_stfts.py:577 sets n_fft = win_length) "
                                  f"or be a common transform size {COMMON_N_FFT} larger than win_length")
    if win_length not in SUPPORTED_WINDOWS:
        raise NotImplementedError(f"win_length must be one of {SUPPORTED_WINDOWS}")
    if window != "hann" or not center or pad_mode != "reflect":
        raise NotImplementedError("only window='hann', center=True, pad_mode='reflect' (config/*.yaml stft_* keys)")


def _as_2d(waveform: torch.Tensor) -> torch.Tensor:
    if waveform.dim() == 3:
        waveform = waveform.squeeze(1)
    if waveform.dim() != 2:
        raise ValueError("waveform must be (batch, time) or (batch, 1, time)")
    if waveform.device.type != "cuda":
        raise LassError("lass_amd computes on an MI355X only: move the waveform to 'cuda' (no CPU fallback)")
    return waveform.float().contiguous()


def multi_resolution_stfts(waveform: torch.Tensor, win_lengths: Sequence[int], hop_length: int = 160,
                           window: str = "hann", center: bool = True, pad_mode: str = "reflect",
                           n_fft: Optional[int] = None):
    """{win_length: (magnitude, cos, sin)}, T = 1 + L // hop_length.  n_fft=None: each window at n_fft = win_length
    (scripts/precompute_stfts.py:577), shapes (B, 1, T, win_length//2+1).  n_fft=2048 (or 1024): every window at that
    common transform size - the input format of lass_amd.resunet_with_multistft.ResUNet30 - shapes (B, 1, T, n_fft//2+1)."""
    for w in win_lengths:
        _check_cfg(w if n_fft is None else n_fft, w, window, center, pad_mode)
    x = _as_2d(waveform)
    if n_fft is None:
        return get_engine(x.device).multi_stft(x, list(win_lengths), hop_length)
    return get_engine(x.device).stft_components(x, n_fft, list(win_lengths), hop_length)


def calculate_stft_components(waveform, n_fft, hop_length, win_length, window, center, pad_mode):
    """scripts/precompute_stfts.py:19-58: (magnitude, cos_phase, sin_phase), each (B, 1, T, n_fft//2+1), contiguous."""
    _check_cfg(n_fft, win_length, window, center, pad_mode)
    return multi_resolution_stfts(waveform, [win_length], hop_length, window, center, pad_mode,
                                  None if n_fft == win_length else n_fft)[win_length]


def make_precomputed_items(mixtures: torch.Tensor, segments: torch.Tensor, texts: Sequence[str],
                           mixture_component_texts: Sequence[Sequence[str]], win_lengths: Sequence[int],
                           hop_length: int = 160, window: str = "hann", center: bool = True,
                           pad_mode: str = "reflect", n_fft: Optional[int] = None) -> List[Dict[str, Any]]:
    """Per-item dicts exactly as scripts/precompute_stfts.py:596-622 builds them (tensors stay on the device; slices
    `t[k:k+1]` keep the leading batch axis of size 1)."""
    if mixtures.shape != segments.shape:
        raise ValueError("mixtures and segments must have the same shape")
    mix = multi_resolution_stfts(mixtures, win_lengths, hop_length, window, center, pad_mode, n_fft)
    seg = multi_resolution_stfts(segments, win_lengths, hop_length, window, center, pad_mode, n_fft)
    # "n_fft": None = every window at n_fft = win_length (what the reference writes); an int = the common transform size
    # of the multi-STFT consumer.  The 2048 window has 1025 bins either way, so the shapes alone cannot tell the two apart.
    common = {"hop_length": hop_length, "window": window, "center": center, "pad_mode": pad_mode, "n_fft": n_fft}
    items = []
    for k in range(mixtures.shape[0]):
        items.append({
            "stfts": {
                "mixture": {w: tuple(t[k:k + 1] for t in mix[w]) for w in win_lengths},
                "segment": {w: tuple(t[k:k + 1] for t in seg[w]) for w in win_lengths},
            },
            "target_waveform": segments[k],
            "text": texts[k],
            "mixture_component_texts": list(mixture_component_texts[k]),
            "stft_common_params": dict(common),
            "stft_win_lengths": list(win_lengths),
        })
    return items


def mix_and_make_precomputed_items(waveforms: torch.Tensor, texts: Sequence[str], mixer, win_lengths: Sequence[int],
                                   hop_length: int = 160, window: str = "hann", center: bool = True,
                                   pad_mode: str = "reflect", n_fft: Optional[int] = None) -> List[Dict[str, Any]]:
    """The producer stage in front of `make_precomputed_items` (scripts/precompute_stfts.py:352-571 mixes a batch of loaded
    segments with the SegmentMixer recipe before the STFTs): `mixer` = a lass_amd.waveform_mixers.SegmentMixer; clip n is mixed
    with its mix_num[n] - 1 successors on the device (`lass_segment_mix`), `mixture_component_texts[n]` lists the captions of the
    primary segment and of the clips mixed in (the recipe's component texts, :160-166), and the spectra of mixtures and declipped
    segments follow in the same call chain - nothing leaves the device."""
    if len(texts) != waveforms.shape[0]:
        raise ValueError("one text per waveform")
    B = waveforms.shape[0]
    mix_num, comp_db, noise_db = mixer.draw(B)
    mixtures, segments = mixer.mix_with_draws(waveforms, mix_num, comp_db, noise_db)
    comp_texts = [[texts[n]] + [texts[(n + i) % B] for i in range(1, int(mix_num[n]))] for n in range(B)]
    return make_precomputed_items(mixtures, segments, texts, comp_texts, win_lengths, hop_length, window, center, pad_mode, n_fft)


def _to_cpu(value):
    if isinstance(value, torch.Tensor):
        return value.detach().cpu()
    if isinstance(value, tuple):
        return tuple(_to_cpu(v) for v in value)
    if isinstance(value, dict):
        return {k: _to_cpu(v) for k, v in value.items()}
    return value


def save_batch_precomputed_data(output_dir, batch_index: int, batch_data_list: List[Dict[str, Any]]) -> int:
    """scripts/precompute_stfts.py:60-122: one `batch_%06d.pt` per batch holding a list of CPU dicts; returns the number
    of items written (0 for an empty list: no file is created)."""
    if not batch_data_list:
        print(f"save_batch_precomputed_data: batch {batch_index} is empty, no file written")
        return 0
    output_dir = pathlib.Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    torch.save([_to_cpu(d) for d in batch_data_list], output_dir / f"batch_{batch_index:06d}.pt")
    return len(batch_data_list)


class _Shard:
    """One `batch_*.pt` file of the wire format: where it is and which global indices it serves."""
    __slots__ = ("path", "first", "count")

    def __init__(self, path: pathlib.Path, first: int, count: int):
        self.path, self.first, self.count = path, first, count


def _batch_number(path: pathlib.Path) -> int:
    return int(path.stem.rsplit("_", 1)[1])


def _read_shard(path: pathlib.Path) -> List[Dict[str, Any]]:
    # weights_only=True: a shard holds tensors, strings, numbers, lists, tuples and dicts - nothing is executed
    items = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(items, list) or not items:
        raise ValueError("not a non-empty list of items")
    return items


class PrecomputedSTFTDataset(torch.utils.data.Dataset):
    """Map-style dataset over a directory of `batch_*.pt` shards (interface of data/precomputed_stft_dataset.py:7-134):
    `len()` is the number of items of all readable shards, `ds[i]` the i-th item counting through the shards in NUMERIC
    batch order.  Unreadable or empty shards are reported on stdout and left out.  Only the shard touched last stays in
    memory, so a sequential sweep reads every file once."""

    def __init__(self, data_dir: str, expected_num_items: Optional[int] = None):
        root = pathlib.Path(data_dir)
        if not root.is_dir():
            raise FileNotFoundError(f"no such directory of precomputed STFT shards: {root}")
        self.data_dir = root
        self._shards: List[_Shard] = []
        n = 0
        for path in sorted(root.glob("batch_*.pt"), key=_batch_number):
            try:
                count = len(_read_shard(path))
            except Exception as err:
                print(f"PrecomputedSTFTDataset: leaving out {path.name} ({err})")
                continue
            self._shards.append(_Shard(path, n, count))
            n += count
        self._starts = [sh.first for sh in self._shards]
        self._size = n
        self._open: Optional[tuple] = None  # (shard position, its items)
        if expected_num_items is not None and expected_num_items != n:
            print(f"PrecomputedSTFTDataset: {n} items under {root}, {expected_num_items} were expected")

    @property
    def shard_sizes(self) -> List[int]:
        return [sh.count for sh in self._shards]

    # read-only mirrors of the reference class's public attributes (data/precomputed_stft_dataset.py:27-29,62)
    @property
    def file_paths(self) -> List[pathlib.Path]:
        return [sh.path for sh in self._shards]

    @property
    def item_counts(self) -> List[int]:
        return [sh.count for sh in self._shards]

    @property
    def cumulative_counts(self) -> List[int]:
        """cumulative_counts[i] = items before shard i; one trailing entry = the total (as the reference keeps it)."""
        return [sh.first for sh in self._shards] + [self._size]

    @property
    def total_items(self) -> int:
        return self._size

    def __len__(self) -> int:
        return self._size

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        if idx < 0 or idx >= self._size:
            raise IndexError(f"item {idx} requested from a PrecomputedSTFTDataset of {self._size} items")
        pos = bisect.bisect_right(self._starts, idx) - 1
        if self._open is None or self._open[0] != pos:
            self._open = None
            try:
                self._open = (pos, _read_shard(self._shards[pos].path))
            except Exception as err:
                raise RuntimeError(f"shard {self._shards[pos].path} became unreadable: {err}") from err
        return self._open[1][idx - self._shards[pos].first]
