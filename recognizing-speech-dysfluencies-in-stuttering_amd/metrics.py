"""Per-file audio metrics of the reference (pipeline1.py:151-186: ``snr_db``, ``spectral_flatness_mean``,
``high_freq_energy_ratio`` -- the columns of output_results/per_file_analysis.csv) on the GPU, behind the C ABI
(``dysf_metrics_batch_dev``).  Same names and argument meaning as the reference; the batched form is what the
preprocessing loop p1:371-417 would call once per batch instead of three times per file.  No CPU implementation here."""
import ctypes
from typing import Sequence

import numpy as np

from . import _lib
from .features import get_extractor
from .params import REFERENCE_149, TARGET_SR

import dataclasses


def audio_metrics_batch(clips: Sequence[np.ndarray], sr: int = TARGET_SR, device: int = 0) -> np.ndarray:
    """[n, 3] float32: snr_db, spectral_flatness_mean, high_freq_energy_ratio of every clip (int16 PCM or float32)."""
    import torch
    ex = get_extractor(REFERENCE_149 if sr == TARGET_SR else dataclasses.replace(REFERENCE_149, sr=int(sr)), device)
    pcm, offsets, kind = ex._pack(clips)
    n = len(clips)
    out = torch.zeros((n, 3), dtype=torch.float32, device=f"cuda:{device}")
    if n == 0:
        return out.cpu().numpy()
    dpcm = torch.from_numpy(pcm).to(out.device)
    doff = torch.from_numpy(offsets).to(out.device)
    _lib.check(_lib.lib().dysf_metrics_batch_dev(
        ex._h, ctypes.c_void_p(dpcm.data_ptr()), kind, ctypes.c_void_p(doff.data_ptr()), n, int(np.max(np.diff(offsets))),
        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
    return out.cpu().numpy()


def _one(y, sr, col):
    if y is None:
        return 0.0
    try:
        return float(audio_metrics_batch([np.asarray(y)], sr)[0, col])
    except _lib.DysfError:
        raise
    except Exception:       # the reference's metric functions swallow their own exceptions and return 0.0
        return 0.0


def snr_db(y: np.ndarray, sr: int = TARGET_SR) -> float:                      # pipeline1.py:151-165
    return _one(y, sr, 0)


def spectral_flatness_mean(y: np.ndarray, sr: int = TARGET_SR) -> float:      # pipeline1.py:168-174
    return _one(y, sr, 1)


def high_freq_energy_ratio(y: np.ndarray, sr: int) -> float:                  # pipeline1.py:177-186
    return _one(y, sr, 2)
