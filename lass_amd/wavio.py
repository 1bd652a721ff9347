"""Minimal RIFF/WAVE reader-writer (PCM16, PCM32, float32), mono down-mix.

Stands in for `librosa.load(path, sr=16000, mono=True)` (dcase_evaluator.py:73-74) for files that are ALREADY at the
target rate: the load then reduces to PCM decode + mono down-mix + int->float scaling (x/32768 for int16, as
soundfile/librosa do).  Whether the Zenodo validation audio is at 16 kHz is NOT established from the reference (its
scripts/process_audio.sh converts TRAINING data with `sox -r 16000 -c 1`; the validation set's rate is not stated):
`librosa.load` would resample such files (librosa 0.10: `res_type="soxr_hq"`).  librosa / soxr are absent here, so a file at
another rate is resampled by `scipy.signal.resample_poly` (polyphase FIR, Kaiser window) with a one-time warning: the
result is band-limited correctly but is NOT bit-comparable with librosa's - **parity unpinned** at this boundary (SURVEY
8c; the reference holds no fixture for it).  `read_wav(..., strict_rate=True)` (or LASS_WAV_STRICT_RATE=1) raises instead,
for callers that would rather resample offline (`sox -r 16000 -c 1`, as scripts/process_audio.sh does for training data).
"""
from __future__ import annotations

import struct

import numpy as np


_warned_resample = False


def _resample(x: np.ndarray, rate: int, sr: int) -> np.ndarray:
    """Polyphase resampling rate -> sr (parity unpinned against librosa.load's soxr_hq, see the module docstring)."""
    from math import gcd

    from scipy.signal import resample_poly
    g = gcd(int(rate), int(sr))
    return resample_poly(x.astype(np.float64), sr // g, rate // g).astype(np.float32)


def read_wav(path: str, sr: int | None = None, strict_rate: bool | None = None) -> tuple[np.ndarray, int]:
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, rate, _, _, bits = fmt
    if tag == 0xFFFE and len(data) > 0:  # WAVE_FORMAT_EXTENSIBLE: sub-format in the extension; infer from bits
        tag = 3 if bits == 32 and b"\x03\x00\x00\x00\x00\x00\x10\x00" in data[:128] else 1
    if tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV encoding tag={tag} bits={bits}")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1).astype(np.float32)  # librosa mono=True
    if sr is not None and rate != sr:
        import os
        if strict_rate is None:
            strict_rate = os.environ.get("LASS_WAV_STRICT_RATE", "0") not in ("", "0")
        if strict_rate:
            raise ValueError(f"{path}: sample rate {rate} != {sr}; resample offline (e.g. sox -r {sr} -c 1)")
        global _warned_resample
        if not _warned_resample:
            _warned_resample = True
            import warnings
            warnings.warn(f"{path}: {rate} Hz resampled to {sr} Hz with scipy.signal.resample_poly - librosa.load would use "
                          "soxr_hq; parity with the reference is unpinned for resampled files (further files: no message)")
        x, rate = _resample(x, rate, sr), sr
    return np.ascontiguousarray(x), rate


def write_wav_f32(path: str, x: np.ndarray, sr: int) -> None:
    x = np.ascontiguousarray(x, dtype="<f4")
    body = x.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 3, 1, sr, sr * 4, 4, 32) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def write_wav_pcm16(path: str, x: np.ndarray, sr: int) -> None:
    q = np.clip(np.round(np.asarray(x, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    body = q.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)
