"""Rational sample-rate conversion on the GPU (second half): lets raw-rate audio (the reference's MP3s
are 22.05 kHz) enter the 16 kHz path the way ``librosa.load(path, sr=16000)`` does at pipeline1.py:100-106.

librosa resamples with soxr, an un-vendored dependency that is absent here, so the reference's own resampled samples
cannot be reproduced ("parity unpinned").  The conversion implemented here is the published polyphase method of
``scipy.signal.resample_poly(x, up, down)`` with its default Kaiser(5.0) design -- that call is the oracle
(``oracle.dysf_oracle.resample_poly_ref``) and the GPU kernel matches it to float32 rounding.

The filter is designed on the host in float64 with NumPy only (windowed sinc, 2*10*max(up,down)+1 taps, unity DC gain,
times ``up``), padded in front as resample_poly does so that output sample n is y_upfirdn[n + n_pre_remove]."""
from __future__ import annotations

import ctypes
import math
from typing import Tuple

import numpy as np

from . import _lib


def design_taps(up: int, down: int) -> Tuple[np.ndarray, int]:
    """(front-padded float64 taps, n_pre_remove) of scipy.signal.resample_poly's default filter for the reduced
    ratio up/down."""
    max_rate = max(up, down)
    f_c = 1.0 / max_rate
    half_len = 10 * max_rate
    numtaps = 2 * half_len + 1
    alpha = 0.5 * (numtaps - 1)
    m = np.arange(numtaps) - alpha
    h = f_c * np.sinc(f_c * m)                                           # firwin, low-pass, cutoff relative to Nyquist
    n = np.arange(numtaps)
    h *= np.i0(5.0 * np.sqrt(1.0 - ((n - alpha) / alpha) ** 2)) / np.i0(5.0)   # symmetric Kaiser(beta = 5)
    h /= h.sum()                                                         # unity gain at DC
    h *= up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return np.concatenate((np.zeros(n_pre_pad), h)), int(n_pre_remove)


def output_length(n_in: int, up: int, down: int) -> int:
    n_out = n_in * up
    return n_out // down + (1 if n_out % down else 0)


class Resampler:
    """``sr_in -> sr_out`` for batches of clips resident on the GPU (int16 or float32 in, float32 out)."""

    def __init__(self, sr_in: int, sr_out: int, device: int = 0):
        g = math.gcd(int(sr_in), int(sr_out))
        self.up, self.down = int(sr_out) // g, int(sr_in) // g
        self.sr_in, self.sr_out, self.device = int(sr_in), int(sr_out), device
        self._h = ctypes.c_void_p()
        if self.up == self.down:
            self.taps, self.n_pre_remove = np.ones(1), 0
        else:
            self.taps, self.n_pre_remove = design_taps(self.up, self.down)
        taps = np.ascontiguousarray(self.taps, np.float64)
        L = _lib.lib()
        rc = L.dysf_resampler_create(device, self.up, self.down, taps.ctypes.data_as(ctypes.c_void_p), len(taps),
                                     self.n_pre_remove, ctypes.byref(self._h))
        if rc != 0:
            raise _lib.DysfError(f"dysf_resampler_create failed ({rc}): {L.dysf_resampler_last_error().decode()}")

    def out_len(self, n_in: int) -> int:
        return output_length(int(n_in), self.up, self.down)

    def resample_batch_device(self, pcm, offsets, max_samples=None):
        """pcm: flat int16 / float32 CUDA tensor of all clips, offsets: int64 [n+1] (CUDA).  Returns (out float32 flat,
        out_offsets int64 [n+1]) on the same device; only enqueues work on the current stream (the output offsets are
        computed with torch ops, no host round trip).  ``max_samples``: an upper bound on a clip's input length (sizes
        the launch; defaults to the whole buffer)."""
        import torch
        assert pcm.is_cuda and offsets.is_cuda and offsets.dtype == torch.int64 and pcm.is_contiguous()
        kind = {torch.int16: _lib.PCM_I16, torch.float32: _lib.PCM_F32}[pcm.dtype]
        n = offsets.numel() - 1
        lens = offsets[1:] - offsets[:-1]
        olens = (lens * self.up + self.down - 1) // self.down
        out_off = torch.zeros(n + 1, dtype=torch.int64, device=pcm.device)
        torch.cumsum(olens, 0, out=out_off[1:])
        total = self.out_len(int(pcm.numel())) + n   # upper bound without reading the device: sum ceil <= ceil(sum) + n
        out = torch.empty(total, dtype=torch.float32, device=pcm.device)
        L = _lib.lib()
        rc = L.dysf_resample_batch_dev(self._h, ctypes.c_void_p(pcm.data_ptr()), kind, ctypes.c_void_p(offsets.data_ptr()),
                                       n, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out_off.data_ptr()),
                                       self.out_len(int(max_samples if max_samples is not None else pcm.numel())),
                                       ctypes.c_void_p(torch.cuda.current_stream(pcm.device).cuda_stream))
        if rc != 0:
            raise _lib.DysfError(f"dysf_resample_batch_dev failed ({rc}): {L.dysf_resampler_last_error().decode()}")
        return out, out_off

    def resample(self, y: np.ndarray) -> np.ndarray:
        """One host clip (int16 or float32) -> float32 at sr_out."""
        import torch
        y = np.array(y, dtype=np.int16 if np.asarray(y).dtype == np.int16 else np.float32, copy=True)  # writable, contiguous
        dev = f"cuda:{self.device}"
        pcm = torch.from_numpy(y).to(dev)
        off = torch.tensor([0, len(y)], dtype=torch.int64, device=dev)
        out, out_off = self.resample_batch_device(pcm, off)
        return out[: self.out_len(len(y))].cpu().numpy()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().dysf_resampler_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
