"""SDR / SI-SDR with the reference's signatures (utils.py:148-200), reductions on the MI355X.

The six sums per clip come from the `sdr_stats` HIP kernel (wavefront-shuffle + f64 accumulation); only the
log10 arithmetic is done on the host.  `stats_to_db` is the batched form the evaluator uses on device-resident
waveforms (no D2H of audio)."""
from __future__ import annotations

import numpy as np
import torch

from .engine import get_engine


def stats_to_db(stats: np.ndarray, length: int, eps: float = 1e-10):
    """stats (N,6) f64 from Engine.sdr_stats -> (sdr (N,), sisdr (N,)) in dB.

    SDR   = 10 log10( clip(mean ref^2, eps) / clip(mean (est-ref)^2, eps) )            utils.py:148-169
    SISDR = 10 log10( (eps32 + sum (a ref)^2) / (eps32 + sum (est - a ref)^2) )         utils.py:172-200
    """
    stats = np.asarray(stats, dtype=np.float64).reshape(-1, 6)
    eps32 = float(np.finfo(np.float32).eps)
    num = np.clip(stats[:, 0] / length, eps, None)
    den = np.clip(stats[:, 3] / length, eps, None)
    sdr = 10.0 * np.log10(num / den)
    sisdr = 10.0 * np.log10((eps32 + stats[:, 4]) / (eps32 + stats[:, 5]))
    return sdr, sisdr


def _stats(ref, est, device=None) -> tuple[np.ndarray, int]:
    if not torch.is_tensor(ref):
        ref = torch.from_numpy(np.ascontiguousarray(ref, dtype=np.float32))
    if not torch.is_tensor(est):
        est = torch.from_numpy(np.ascontiguousarray(est, dtype=np.float32))
    if device is None:
        device = ref.device if ref.is_cuda else (est.device if est.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    eng = get_engine(device)
    r = ref.reshape(1, -1).to(eng.device, torch.float32)
    e = est.reshape(1, -1).to(eng.device, torch.float32)
    return eng.sdr_stats(r, e).cpu().numpy(), r.shape[1]


def calculate_sdr(ref, est, eps=1e-10) -> float:
    st, n = _stats(ref, est)
    return float(stats_to_db(st, n, eps)[0][0])


def calculate_sisdr(ref, est) -> float:
    st, n = _stats(ref, est)
    return float(stats_to_db(st, n)[1][0])
