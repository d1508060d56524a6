"""Host side of the feature path: the reference's Python signatures on top of the C-ABI HIP library.

Mirrors (names, argument meaning, error behaviour) pipeline1.py:
  load_audio                p1:100-106     PCM_16 WAV at its native rate only (== int16 / 32768)
  save_cache / load_cache   p1:113-121     NPY v1.0 '<f4' (149,), byte-compatible with cache_features/
  extract_audio_features    p1:206-239     -> float32 (144,), zeros on any failure
  extract_text_features     p1:242-254
  extract_features          p1:257-265     -> float32 (149,)
  cached_extract_features   p1:429-440
All numerics run in libdysf_hip.so on the GPU; there is no CPU implementation in this package.
"""
import ctypes
import dataclasses
import logging
import os
import re
import struct
from collections import Counter
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .params import REFERENCE_149, TARGET_SR, TEXT_FEATURE_LEN, FeatureParams

log = logging.getLogger("rsd_amd")
CACHE_DIR = "cache_features"  # pipeline1.py:31


def _vp(arr: Optional[np.ndarray]):
    return None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)


class FeatureExtractor:
    """One DysfHandle: device tables + workspace for one parameter set on one GPU.  Not thread-safe."""

    def __init__(self, params: FeatureParams = REFERENCE_149, device: int = 0):
        self.params = params
        self.device = device
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        c = params.to_c()
        _lib.check(L.dysf_create(ctypes.byref(c), device, ctypes.byref(self._h)))
        self.pooled_dim = L.dysf_pooled_dim(self._h)
        self.frame_rows = L.dysf_frame_rows(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().dysf_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- host buffers
    @staticmethod
    def _pack(clips: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, int]:
        if len(clips) == 0:
            return np.zeros(0, np.int16), np.zeros(1, np.int64), _lib.PCM_I16
        kinds = {np.asarray(c).dtype == np.int16 for c in clips}
        if kinds == {True}:
            dt, kind = np.int16, _lib.PCM_I16
        else:
            dt, kind = np.float32, _lib.PCM_F32
        flat = [np.ascontiguousarray(np.asarray(c).reshape(-1), dtype=dt) for c in clips]
        offsets = np.zeros(len(flat) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(f) for f in flat])
        pcm = np.concatenate(flat) if flat else np.zeros(0, dt)
        return pcm, offsets, kind

    def extract_batch(self, clips: Sequence[np.ndarray], frames: bool = False, tuning: bool = False) -> Dict[str, np.ndarray]:
        """Batched extraction from host arrays (int16 PCM or float32 in [-1, 1)).

        Returns ``pooled`` [n, pooled_dim], ``status`` [n] (+ ``frames`` [n, 3*n_mfcc, Tmax], ``tuning`` [n])."""
        pcm, offsets, kind = self._pack(clips)
        n = len(clips)
        out = {"pooled": np.zeros((n, self.pooled_dim), np.float32), "status": np.zeros(n, np.int32)}
        t_max = self.params.n_frames(int(np.max(np.diff(offsets)))) if n else 0
        if frames:
            out["frames"] = np.zeros((n, self.frame_rows, t_max), np.float32)
        if tuning:
            out["tuning"] = np.zeros(n, np.int32)
        if n == 0:
            return out
        _lib.check(_lib.lib().dysf_extract_batch(
            self._h, _vp(pcm), kind, _vp(offsets), n, _vp(out["pooled"]), _vp(out.get("frames")), t_max,
            _vp(out["status"]), _vp(out.get("tuning"))))
        return out

    # ---------------------------------------------------------------- device buffers (torch is plumbing only)
    def extract_batch_device(self, pcm, offsets, max_samples: int, pooled: bool = True, frames: bool = False,
                             frame_stride: Optional[int] = None, tuning: bool = False, out: Optional[dict] = None):
        """Hot path: ``pcm`` (int16 or float32) and ``offsets`` (int64, [n+1]) are CUDA/HIP torch tensors that
        stay resident; outputs are torch tensors on the same device.  Only enqueues work on the current stream.
        ``out``: optional dict of preallocated, contiguous output tensors (``status`` and whichever of ``pooled`` /
        ``frames`` / ``tuning`` are wanted) to write into instead of allocating -- e.g. row slices of larger tensors."""
        import torch
        assert pcm.is_cuda and offsets.is_cuda and offsets.dtype == torch.int64 and pcm.is_contiguous()
        assert offsets.is_contiguous()
        kind = {torch.int16: _lib.PCM_I16, torch.float32: _lib.PCM_F32}[pcm.dtype]
        n = offsets.numel() - 1
        dev = pcm.device
        t_max = self.params.n_frames(max_samples)
        fs = int(frame_stride or t_max)
        if out is not None:
            res = out
            for k, shape, dt in (("status", (n,), torch.int32), ("pooled", (n, self.pooled_dim), torch.float32),
                                 ("frames", (n, self.frame_rows, fs), torch.float32), ("tuning", (n,), torch.int32)):
                if k in res:
                    assert res[k].is_contiguous() and tuple(res[k].shape) == shape and res[k].dtype == dt, k
        else:
            res = {}
            res["status"] = torch.empty(n, dtype=torch.int32, device=dev)
            if pooled:
                res["pooled"] = torch.empty((n, self.pooled_dim), dtype=torch.float32, device=dev)
            if frames:
                res["frames"] = torch.empty((n, self.frame_rows, fs), dtype=torch.float32, device=dev)
            if tuning:
                res["tuning"] = torch.empty(n, dtype=torch.int32, device=dev)
        ptr = lambda k: ctypes.c_void_p(res[k].data_ptr()) if k in res else None  # noqa: E731
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().dysf_extract_batch_dev(
            self._h, ctypes.c_void_p(pcm.data_ptr()), kind, ctypes.c_void_p(offsets.data_ptr()), n, int(max_samples),
            ptr("pooled"), ptr("frames"), fs, ptr("status"), ptr("tuning"), stream))
        return res

    def kernel_timing(self, enable: bool = True):
        """Measurement hook: returns {'stft_mel', 'mfcc', 'chroma'} durations in ms of the most recent device-resident
        launch (HIP events recorded on the launch stream), then keeps event recording on/off as requested."""
        ms = (ctypes.c_float * 3)()
        _lib.check(_lib.lib().dysf_debug_timing(self._h, 1 if enable else 0, ms))
        return {"stft_mel": ms[0], "mfcc": ms[1], "chroma": ms[2]}

    # ---------------------------------------------------------------- kernel-level test hooks
    def _debug_dev(self, fn, y: np.ndarray, cols: int) -> np.ndarray:
        import torch
        y = np.asarray(y)
        kind = _lib.PCM_I16 if y.dtype == np.int16 else _lib.PCM_F32
        yt = torch.from_numpy(np.ascontiguousarray(y, dtype=np.int16 if kind == _lib.PCM_I16 else np.float32)).to(f"cuda:{self.device}")
        T = self.params.n_frames(len(y))
        out = torch.zeros((T, cols), dtype=torch.float32, device=yt.device)
        _lib.check(fn(self._h, ctypes.c_void_p(yt.data_ptr()), kind, len(y), ctypes.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(yt.device)
        return out.cpu().numpy()

    def stft_power(self, y: np.ndarray) -> np.ndarray:
        """[T, n_bins] power spectrogram of one clip (device kernel, debug hook)."""
        return self._debug_dev(_lib.lib().dysf_debug_stft_power_dev, y, self.params.n_bins)

    def logmel(self, y: np.ndarray) -> np.ndarray:
        """[T, n_mels] un-clamped log-mel spectrogram of one clip (device kernel, debug hook)."""
        return self._debug_dev(_lib.lib().dysf_debug_logmel_dev, y, self.params.n_mels)

    def peaks(self, y: np.ndarray):
        """(mag, bin) arrays of the piptrack peaks the tuning estimate saw for one clip (chroma presets, debug hook)."""
        self.extract_batch([y])
        cap = 1 << 16
        mag, b = np.zeros(cap, np.float32), np.zeros(cap, np.int32)
        n = _lib.check(_lib.lib().dysf_debug_peaks(self._h, _vp(mag), _vp(b), cap))
        return mag[:n], b[:n]

    def table(self, which: int, arg: int = 0) -> np.ndarray:
        p = self.params
        size = {0: p.n_fft, 1: p.n_mels * p.n_bins, 2: p.n_mfcc * p.n_mels, 3: 12 * p.n_bins}[which]
        buf = np.zeros(size, np.float32)
        _lib.check(_lib.lib().dysf_debug_table(self._h, which, arg, _vp(buf), size))
        return buf


class HostTables:
    """Tables only (dysf_create with device = -1): usable without a GPU, for table tests."""

    def __init__(self, params: FeatureParams):
        self.params = params
        self._h = ctypes.c_void_p()
        c = params.to_c()
        _lib.check(_lib.lib().dysf_create(ctypes.byref(c), -1, ctypes.byref(self._h)))

    table = FeatureExtractor.table
    close = FeatureExtractor.close
    __del__ = FeatureExtractor.__del__


# -------------------------------------------------------------------------------- reference-shaped functions
_extractors: Dict[Tuple[FeatureParams, int], FeatureExtractor] = {}


def get_extractor(params: FeatureParams = REFERENCE_149, device: int = 0) -> FeatureExtractor:
    key = (params, device)
    if key not in _extractors:
        _extractors[key] = FeatureExtractor(params, device)
    return _extractors[key]


_RESAMPLERS = {}


def _resampler(sr_in: int, sr_out: int):
    from .resample import Resampler
    key = (sr_in, sr_out)
    if key not in _RESAMPLERS:
        _RESAMPLERS[key] = Resampler(sr_in, sr_out, 0)
    return _RESAMPLERS[key]


def load_audio(path: str, sr: int = TARGET_SR):
    """pipeline1.py:100-106 for mono PCM_16 WAV files.  A file already at ``sr`` (the case parity is pinned on) is
    returned as ``int16 / 32768``; a file at another rate is converted on the GPU by ``rsd_amd.resample.Resampler``
    (polyphase, scipy.signal.resample_poly's filter -- librosa would use soxr here, so these samples are not
    bit-comparable with the reference's).  Returns (float32 y, sr) or (None, None) on failure, like the
    reference.  Decoding other containers is upstream of this path."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
        if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
            raise ValueError("only RIFF/WAVE PCM_16 is supported on this path")
        pos, fmt, data = 12, None, None
        while pos + 8 <= len(blob):
            cid, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
            if cid == b"fmt ":
                fmt = struct.unpack_from("<HHIIHH", blob, pos + 8)
            elif cid == b"data":
                data = blob[pos + 8:pos + 8 + size]
            pos += 8 + size + (size & 1)
        if fmt is None or data is None:
            raise ValueError("missing fmt/data chunk")
        tag, channels, file_sr, _, _, bits = fmt
        if tag != 1 or channels != 1 or bits != 16:
            raise ValueError("only mono PCM_16 is supported on this path")
        pcm = np.frombuffer(data[: len(data) // 2 * 2], dtype="<i2")
        if file_sr != sr:
            return _resampler(int(file_sr), int(sr)).resample(pcm), int(sr)
        return (pcm.astype(np.float32) / np.float32(32768.0)), int(sr)
    except Exception as e:  # same convention as the reference
        logging.error(f"load_audio fail {path}: {e}")
        return None, None


def cached_path(path: str):
    return os.path.normpath(path)


def save_cache(arr: np.ndarray, cache_path: str):
    np.save(cache_path, arr)


def load_cache(cache_path: str):
    try:
        return np.load(cache_path, allow_pickle=False)
    except Exception:
        return None


def extract_audio_features(y, sr: int, params: FeatureParams = REFERENCE_149, device: int = 0, strict: bool = False) -> np.ndarray:
    """pipeline1.py:206-239: float32 [audio_feature_len]; zeros when y is None or the clip is rejected."""
    n_out = params.audio_feature_len
    if y is None:
        return np.zeros(n_out, dtype=np.float32)
    try:
        if sr != params.sr:
            params = dataclasses.replace(params, sr=int(sr))
        y = np.asarray(y)
        if y.ndim != 1:
            raise ValueError("mono audio expected")
        res = get_extractor(params, device).extract_batch([y])
        st = int(res["status"][0])
        if st != _lib.DYSF_OK:
            raise ValueError(f"clip rejected: {_lib.STATUS_NAMES.get(st, st)}")
        return res["pooled"][0].astype(np.float32)
    except _lib.DysfError:
        raise  # a missing / broken HIP extension is never silently turned into zeros
    except Exception as e:
        if strict:
            raise
        logging.error(f"extract_audio_features error: {e}")
        return np.zeros(n_out, dtype=np.float32)


def repetition_stats_from_text(text: str):
    """pipeline1.py:191-201."""
    words = re.findall(r"\b\w+\b", text.lower()) if text else []
    if not words:
        return {"repetition_count": 0, "repetition_ratio": 0.0, "unique_ratio": 0.0}
    repeats = sum(c - 1 for c in Counter(words).values() if c > 1)
    return {"repetition_count": float(repeats), "repetition_ratio": repeats / len(words),
            "unique_ratio": len(set(words)) / len(words)}


def extract_text_features(text: str) -> np.ndarray:
    """pipeline1.py:242-254 (the transcript is always "" in the reference, giving five zeros)."""
    if not text:
        return np.zeros(TEXT_FEATURE_LEN, dtype=np.float32)
    rep = repetition_stats_from_text(text)
    words = re.findall(r"\b\w+\b", text.lower())
    return np.array([float(len(text)), float(len(words)), rep["repetition_count"], rep["repetition_ratio"],
                     rep["unique_ratio"]], dtype=np.float32)


def extract_features(y, sr: int, transcript: str = "", params: FeatureParams = REFERENCE_149, device: int = 0) -> np.ndarray:
    """pipeline1.py:257-265: float32 [149] = audio features (144) + text features (5)."""
    feats = np.hstack([extract_audio_features(y, sr, params, device), extract_text_features(transcript)]).astype(np.float32)
    total = params.total_feature_len
    if feats.size != total:
        out = np.zeros(total, dtype=np.float32)
        out[: min(feats.size, total)] = feats[:total]
        return out
    return feats


def cached_extract_features(path: str, transcript: str, suffix: str, cache_dir: str = CACHE_DIR,
                            params: FeatureParams = REFERENCE_149, device: int = 0) -> np.ndarray:
    """pipeline1.py:429-440: memoise per basename + suffix as <cache_dir>/<stem>_<suffix>_feats.npy."""
    base = os.path.basename(path).rsplit(".", 1)[0]
    cache_file = cached_path(os.path.join(cache_dir, f"{base}_{suffix}_feats.npy"))
    cached = load_cache(cache_file)
    if cached is not None:
        return np.array(cached)
    y, sr = load_audio(path, sr=params.sr)
    feats = extract_features(y, sr, transcript, params, device)
    save_cache(feats, cache_file)
    return feats
