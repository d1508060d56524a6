"""The reference's audio cleaning stage (pipeline1.py:126-146 ``clean_audio_and_cache``; main.py:649-663 and
main1.py:598-611 are the same with ``prop_decrease=0.8``) on the GPU, behind the C ABI (``dysf_clean_batch_dev``):

    y_clean = nr.reduce_noise(y=y, sr=sr)         ->  reduce_noise(y, sr)            (noisereduce's default gate)
    y_clean = librosa.util.normalize(y_clean)     ->  normalised float32 / the int16 samples sf.write stores
    sf.write(out_path, y_clean, sr)               ->  write_wav_pcm16

Same names and argument meaning as the reference and as ``noisereduce.reduce_noise`` (the non-stationary defaults; a
stationary gate or a noise clip is not part of the reference's calls and raises).  No CPU implementation here: without
the HIP library every call fails."""
import ctypes
import logging
import os
import struct
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from .features import FeatureExtractor, cached_path, load_audio
from .params import TARGET_SR

CLEAR_DIR = "clear_audio"      # pipeline1.py:32

_CLEANERS: Dict[tuple, "AudioCleaner"] = {}


class AudioCleaner:
    """One configured spectral gate on one device (``nr.reduce_noise``'s keyword values)."""

    def __init__(self, sr: int = TARGET_SR, device: int = 0, prop_decrease: float = 1.0, time_constant_s: float = 2.0,
                 freq_mask_smooth_hz: float = 500, time_mask_smooth_ms: float = 50, thresh_n_mult_nonstationary: float = 2,
                 sigmoid_slope_nonstationary: float = 10, padding: int = 30000):
        self.sr, self.device = int(sr), int(device)
        h = ctypes.c_void_p()
        rc = _lib.lib().dysf_clean_create(self.device, self.sr, float(prop_decrease), float(time_constant_s),
                                          float(freq_mask_smooth_hz), float(time_mask_smooth_ms),
                                          float(thresh_n_mult_nonstationary), float(sigmoid_slope_nonstationary),
                                          int(padding), ctypes.byref(h))
        if rc < 0:
            raise _lib.DysfError(f"dysf_clean_create: {_lib.lib().dysf_clean_last_error().decode(errors='replace')}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dysf_clean_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clean_batch_device(self, pcm, offsets, max_samples: int, denoised: bool = False, normalized: bool = False,
                           pcm16: bool = True):
        """pcm: int16 or float32 CUDA tensor of all clips back to back; offsets: int64 CUDA tensor [n + 1].
        Returns a dict with the requested outputs (CUDA tensors indexed like ``pcm``) and ``status`` [n] int32."""
        import torch
        n = int(offsets.numel()) - 1
        total = int(pcm.numel())
        dev = pcm.device
        out = {"status": torch.zeros((max(n, 0),), dtype=torch.int32, device=dev)}
        if denoised:
            out["denoised"] = torch.zeros((total,), dtype=torch.float32, device=dev)
        if normalized:
            out["normalized"] = torch.zeros((total,), dtype=torch.float32, device=dev)
        if pcm16:
            out["pcm16"] = torch.zeros((total,), dtype=torch.int16, device=dev)
        if n <= 0:
            return out
        kind = {torch.int16: _lib.PCM_I16, torch.float32: _lib.PCM_F32}[pcm.dtype]
        ptr = lambda k: ctypes.c_void_p(out[k].data_ptr()) if k in out else None
        rc = _lib.lib().dysf_clean_batch_dev(
            self._h, ctypes.c_void_p(pcm.data_ptr()), kind, ctypes.c_void_p(offsets.data_ptr()), n, int(max_samples), total,
            ptr("denoised"), ptr("normalized"), ptr("pcm16"), ctypes.c_void_p(out["status"].data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise _lib.DysfError(f"dysf_clean_batch_dev: {_lib.lib().dysf_clean_last_error().decode(errors='replace')}")
        return out

    def clean_batch(self, clips: Sequence[np.ndarray], **want) -> Dict[str, list]:
        """Host convenience: a list of 1-D int16 / float32 clips in, per-clip NumPy arrays out (plus ``status``)."""
        import torch
        pcm, offsets, _ = FeatureExtractor._pack(clips)
        dev = torch.device(f"cuda:{self.device}")
        res = self.clean_batch_device(torch.from_numpy(pcm).to(dev), torch.from_numpy(offsets).to(dev),
                                      int(np.max(np.diff(offsets))) if len(clips) else 0, **want)
        out = {"status": res["status"].cpu().numpy()}
        for k in ("denoised", "normalized", "pcm16"):
            if k in res:
                flat = res[k].cpu().numpy()
                out[k] = [flat[offsets[i]:offsets[i + 1]] for i in range(len(clips))]
        return out


def get_cleaner(sr: int = TARGET_SR, device: int = 0, prop_decrease: float = 1.0) -> AudioCleaner:
    key = (int(sr), int(device), float(prop_decrease))
    if key not in _CLEANERS:
        _CLEANERS[key] = AudioCleaner(sr=sr, device=device, prop_decrease=prop_decrease)
    return _CLEANERS[key]


def reduce_noise(y: np.ndarray, sr: int, stationary: bool = False, y_noise=None, prop_decrease: float = 1.0,
                 device: int = 0, **kw) -> np.ndarray:
    """``noisereduce.reduce_noise(y=y, sr=sr[, prop_decrease=...])`` as the reference calls it (p1:139, m:655, m1:604):
    mono float32 in, float32 of the same length out."""
    if stationary or y_noise is not None:
        raise NotImplementedError("the reference only uses noisereduce's non-stationary gate without a noise clip")
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float32))
    if y.ndim != 1:
        raise ValueError("mono clips only (librosa.load(..., mono=True), pipeline1.py:102)")
    cleaner = AudioCleaner(sr=sr, device=device, prop_decrease=prop_decrease, **kw) if kw else get_cleaner(sr, device, prop_decrease)
    return cleaner.clean_batch([y], denoised=True, pcm16=False)["denoised"][0]


def write_wav_pcm16(path: str, pcm16: np.ndarray, sr: int):
    """The file ``sf.write(path, float_data, sr)`` produces for mono data: a 44-byte RIFF/WAVE header (fmt chunk of 16
    bytes, PCM tag 1, 16 bits) followed by the little-endian samples -- the layout of every reference clear_audio/*.wav."""
    data = np.ascontiguousarray(pcm16, dtype="<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(data)) + data)


def clean_audio_and_cache(in_path: str, clear_dir: str = CLEAR_DIR, prop_decrease: float = 1.0, device: int = 0) -> Optional[str]:
    """pipeline1.py:126-146: writes the cleaned WAV to ``clear_dir/<basename>.wav`` and returns its path; returns the
    existing path if it is already there, None on failure (including a non-finite result, where librosa's normalize raises)."""
    base = os.path.basename(in_path).rsplit(".", 1)[0]
    out_path = cached_path(os.path.join(clear_dir, f"{base}.wav"))
    if os.path.exists(out_path):
        return out_path
    y, sr = load_audio(in_path, sr=TARGET_SR)
    if y is None:
        return None
    try:
        res = get_cleaner(sr, device, prop_decrease).clean_batch([y], pcm16=True)
        if int(res["status"][0]) != 0:
            raise ValueError("Input must be finite")       # librosa.util.normalize's ParameterError
        os.makedirs(clear_dir, exist_ok=True)
        write_wav_pcm16(out_path, res["pcm16"][0], sr)
        return out_path
    except Exception as e:
        logging.error(f"clean_audio fail {in_path}: {e}")
        return None
