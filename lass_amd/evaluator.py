"""DCASE-2024 T9 evaluator with the reference's interface (dcase_evaluator.py:27-145).

Same constructor, same `__call__(pl_model) -> (mean_sisdr, mean_sdri, mean_sdr)`, same mixing / declipping / metric
definitions and the same printed line.  Differences are in execution only:
  * clips are separated in batches (`batch_size`, default 16) instead of one by one - eval-mode BatchNorm has no
    cross-clip coupling, so results are unchanged;
  * SDR / SI-SDR reductions run on the device over the resident waveforms (no D2H of audio);
  * under torch.distributed the clip list is block-sharded over ranks and the per-clip metric rows are all-gathered
    once at the end (RCCL when the backend is "nccl").
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List

import numpy as np
import torch

from . import dist as ldist
from .engine import get_engine
from .metrics import stats_to_db
from .utils import load_ss_model, parse_yaml
from .wavio import read_wav


class DCASEEvaluator:
    def __init__(self, sampling_rate=16000, eval_indexes="lass_synthetic_validation.csv", audio_dir="lass_validation",
                 batch_size: int = 16) -> None:
        r"""DCASE T9 LASS evaluator (dcase_evaluator.py:28-47)."""
        self.sampling_rate = sampling_rate
        with open(eval_indexes) as csv_file:
            csv_reader = csv.reader(csv_file, delimiter=",")
            eval_list = [row for row in csv_reader][1:]
        self.eval_list = eval_list
        self.audio_dir = audio_dir
        self.batch_size = batch_size
        self.last_rows = None  # (N,3) per-clip [sdr, sdri, sisdr] of the last call (all ranks)

    def _load_clip(self, eval_data):
        """dcase_evaluator.py:67-89 for one csv row -> (source, mixture, caption), float32."""
        source, noise, snr, caption = eval_data
        snr = int(snr)
        source, _ = read_wav(os.path.join(self.audio_dir, f"{source}.wav"), self.sampling_rate)
        noise, _ = read_wav(os.path.join(self.audio_dir, f"{noise}.wav"), self.sampling_rate)
        source = source.copy()
        # create audio mixture with a specific SNR level
        source_power = np.mean(source ** 2)
        noise_power = np.mean(noise ** 2)
        desired_noise_power = source_power / (10 ** (snr / 10))
        scaling_factor = np.sqrt(desired_noise_power / noise_power)
        noise = noise * scaling_factor
        mixture = source + noise
        # declipping if need be
        max_value = np.max(np.abs(mixture))
        if max_value > 1:
            source *= 0.9 / max_value
            mixture *= 0.9 / max_value
        return source.astype(np.float32), mixture.astype(np.float32), caption

    def __call__(self, pl_model) -> tuple:
        r"""Evaluate (dcase_evaluator.py:49-122)."""
        rank, ws = ldist.world()
        if rank == 0:
            print("Evaluation on DCASE T9 synthetic validation set.")
        pl_model.eval()
        device = pl_model.device
        eng = get_engine(device)
        n_total = len(self.eval_list)
        lo, hi = ldist.shard_range(n_total, rank, ws)
        rows: List[np.ndarray] = []
        with torch.no_grad():
            i = lo
            while i < hi:
                group = [self._load_clip(self.eval_list[i])]
                i += 1
                # batch consecutive clips of identical length (DCASE clips are all 10 s)
                while i < hi and len(group) < self.batch_size:
                    nxt = self._load_clip(self.eval_list[i])
                    if nxt[0].shape != group[0][0].shape:
                        break
                    group.append(nxt)
                    i += 1
                src = torch.from_numpy(np.stack([g[0] for g in group])).to(device)
                mix = torch.from_numpy(np.stack([g[1] for g in group])).to(device)
                conditions = pl_model.query_encoder.get_query_embed(modality="text", text=[g[2] for g in group],
                                                                    device=device)
                input_dict = {"mixture": mix[:, None, :], "condition": conditions}
                sep = pl_model.ss_model(input_dict)["waveform"][:, 0, :]
                length = src.shape[1]
                st_sep = eng.sdr_stats(src, sep.contiguous()).cpu().numpy()
                st_mix = eng.sdr_stats(src, mix).cpu().numpy()
                sdr, sisdr = stats_to_db(st_sep, length)
                sdr_no_sep, _ = stats_to_db(st_mix, length)
                rows.append(np.stack([sdr, sdr - sdr_no_sep, sisdr], axis=1))
        local = np.concatenate(rows, axis=0) if rows else np.zeros((0, 3))
        allrows = ldist.gather_rows(local, n_total, device)
        self.last_rows = allrows
        mean_sdr, mean_sdri, mean_sisdr = (float(np.mean(allrows[:, k])) for k in range(3))
        return mean_sisdr, mean_sdri, mean_sdr


def eval(evaluator, checkpoint_path, config_yaml="config/audiosep_base.yaml", device="cuda", query_encoder=None):
    """dcase_evaluator.py:126-145.  `query_encoder` defaults to the precomputed-embedding stand-in (CLAP itself stays
    on PyTorch-ROCm and needs its own checkpoint; pass a CLAP_Encoder-compatible object to use it)."""
    from .audiosep import PrecomputedQueryEncoder

    configs = parse_yaml(config_yaml)
    if query_encoder is None:
        query_encoder = PrecomputedQueryEncoder()
    pl_model = load_ss_model(configs=configs, checkpoint_path=checkpoint_path, query_encoder=query_encoder).to(device)
    print("-------  Start Evaluation  -------")
    SISDR, SDRi, SDR = evaluator(pl_model)
    msg_clotho = "SDR: {:.3f}, SDRi: {:.3f}, SISDR: {:.3f}".format(SDR, SDRi, SISDR)
    print(msg_clotho)
    print("-------------------------  Done  ---------------------------")
    return SDR, SDRi, SISDR
