"""CPU oracle for the audio cleaning stage  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/`` (and the smoke / cpu_baseline legs named in DESIGN.md) may import this module; the product path
(``rsd_amd.cleaning``) never does and has no CPU fallback.

What it restates
----------------
``/root/reference/pipeline1.py:126-146`` ``clean_audio_and_cache``:

    y_clean = nr.reduce_noise(y=y, sr=sr)          # p1:139   (main.py:655 / main1.py:604 pass prop_decrease=0.8)
    y_clean = librosa.util.normalize(y_clean)      # p1:140
    sf.write(out_path, y_clean, sr)                # p1:141   -> PCM_16 WAV

PARITY UNPINNED for ``reduce_noise``: ``noisereduce`` is an un-vendored, unversioned dependency
(``requirements.txt:7``) that is not installed in this image, and the reference's raw inputs
(``segrigated_samples/*.mp3``) cannot be decoded here, so no input/output pair of this stage exists.  The gate below
restates the published algorithm of noisereduce 2.x / 3.x with its defaults (``stationary=False``:
``SpectralGateNonStationary``), which is thin glue over SciPy calls that ARE importable here and are used verbatim:
``scipy.signal.stft / istft / filtfilt / fftconvolve``.

Pinned by reference data: the last two steps.  ``librosa.util.normalize`` (peak -> 1.0) followed by python-soundfile's
PCM_16 write (libsndfile ``f2s_clip_array``: python-soundfile switches SFC_SET_CLIPPING on, so the scale is 32768 with
clipping at +32767) must leave every ``clear_audio/*.wav`` with a peak of exactly +32767 or -32768 --
``tests/test_oracle_golden.py::test_reference_clean_wavs_pin_normalise_and_pcm16`` checks all 892 files.
"""
from __future__ import annotations

import numpy as np
import scipy.signal


# ---------------------------------------------------------------------------------------------
# noisereduce.spectralgate.base.SpectralGate (defaults of nr.reduce_noise)
N_FFT = 1024
PADDING = 30000
CHUNK_SIZE = 600000
TIME_CONSTANT_S = 2.0
FREQ_MASK_SMOOTH_HZ = 500
TIME_MASK_SMOOTH_MS = 50
THRESH_N_MULT_NONSTATIONARY = 2
SIGMOID_SLOPE_NONSTATIONARY = 10


def smoothing_filter(n_grad_freq: int, n_grad_time: int) -> np.ndarray:
    """noisereduce.spectralgate.base._smoothing_filter: outer product of two triangles, normalised to sum 1."""
    def tri(n):
        return np.concatenate([np.linspace(0, 1, n + 1, endpoint=False), np.linspace(1, 0, n + 2)])[1:-1]
    f = np.outer(tri(n_grad_freq), tri(n_grad_time))
    return f / np.sum(f)


def mask_smoothing_sizes(sr: int, n_fft: int, hop: int, freq_mask_smooth_hz=FREQ_MASK_SMOOTH_HZ,
                         time_mask_smooth_ms=TIME_MASK_SMOOTH_MS):
    """SpectralGate._generate_mask_smoothing_filter: tap half-widths in bins / frames."""
    n_grad_freq = int(freq_mask_smooth_hz / (sr / (n_fft / 2)))
    n_grad_time = int(time_mask_smooth_ms / ((hop / sr) * 1000))
    if n_grad_freq < 1 or n_grad_time < 1:
        raise ValueError("mask smoothing narrower than one bin / frame")
    return n_grad_freq, n_grad_time


def time_smoothing_coefficient(sr: int, hop: int, time_constant_s: float = TIME_CONSTANT_S) -> float:
    """noisereduce.spectralgate.nonstationary.get_time_smoothed_representation: the one-pole coefficient b."""
    t_frames = time_constant_s * sr / float(hop)
    return (np.sqrt(1 + 4 * t_frames ** 2) - 1) / (2 * t_frames ** 2)


def spectral_gate_nonstationary(chunk: np.ndarray, sr: int, prop_decrease: float = 1.0, n_fft: int = N_FFT) -> np.ndarray:
    """SpectralGateNonStationary.spectral_gating_nonstationary on one (already padded) float64 channel."""
    win_length, hop = n_fft, n_fft // 4
    _, _, sig_stft = scipy.signal.stft(chunk, nfft=n_fft, noverlap=win_length - hop, nperseg=win_length, padded=False)
    abs_sig_stft = np.abs(sig_stft)
    b = time_smoothing_coefficient(sr, hop)
    sig_stft_smooth = scipy.signal.filtfilt([b], [1, b - 1], abs_sig_stft, axis=-1, padtype=None)
    with np.errstate(all="ignore"):
        sig_mult_above_thresh = (abs_sig_stft - sig_stft_smooth) / sig_stft_smooth
        sig_mask = 1 / (1 + np.exp(-(sig_mult_above_thresh + -THRESH_N_MULT_NONSTATIONARY) * SIGMOID_SLOPE_NONSTATIONARY))
    ngf, ngt = mask_smoothing_sizes(sr, n_fft, hop)
    if not (ngf == 1 and ngt == 1):
        sig_mask = scipy.signal.fftconvolve(sig_mask, smoothing_filter(ngf, ngt), mode="same")
    sig_mask = sig_mask * prop_decrease + np.ones(np.shape(sig_mask)) * (1.0 - prop_decrease)
    sig_stft_denoised = sig_stft * sig_mask
    _, denoised = scipy.signal.istft(sig_stft_denoised, nfft=n_fft, noverlap=win_length - hop, nperseg=win_length)
    out = np.zeros(chunk.shape, chunk.dtype)
    out[: len(denoised)] = denoised[: len(out)]
    return out


def reduce_noise(y: np.ndarray, sr: int, prop_decrease: float = 1.0) -> np.ndarray:
    """nr.reduce_noise(y=y, sr=sr[, prop_decrease=...]) for a mono clip, as SpectralGate.get_traces runs it with the
    defaults chunk_size=600000, padding=30000:
      * n <= chunk_size: filter_chunk(0, n) -- the clip between 30 000 zeros each side in a float64 chunk (``_read_chunk``
        builds it with np.zeros), gate, crop;
      * longer clips (the reference's sidebar uploads of 126.5 s went this way, main1.py:598-611): chunk k is the FULL span
        [k * chunk_size, (k + 1) * chunk_size) -- also the last one, read past the end as zeros -- padded with the 30 000
        REAL neighbouring samples either side (zeros outside the clip), filtered on its own, cropped to the span, and
        the spans are concatenated up to n.
    The result is cast back to the input dtype."""
    y = np.asarray(y)
    n = y.shape[-1]

    def filter_chunk(start, end):   # SpectralGate.filter_chunk / _read_chunk
        i1, i2 = start - PADDING, end + PADDING
        chunk = np.zeros(i2 - i1)
        i1b, i2b = max(i1, 0), min(i2, n)
        chunk[i1b - i1:i2b - i1] = y[i1b:i2b]
        return spectral_gate_nonstationary(chunk, sr, prop_decrease)[start - i1:end - i1]

    if n <= CHUNK_SIZE:
        return filter_chunk(0, n).astype(y.dtype)
    out = np.zeros(n, dtype=y.dtype)
    for k in range((n - 1) // CHUNK_SIZE + 1):
        start0, end0 = k * CHUNK_SIZE, min((k + 1) * CHUNK_SIZE, n)
        out[start0:end0] = filter_chunk(k * CHUNK_SIZE, (k + 1) * CHUNK_SIZE)[:end0 - start0]
    return out


# ---------------------------------------------------------------------------------------------
def normalize(y: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(S) with its defaults (norm=inf, axis=0, threshold=tiny, fill=None): float64 peak,
    float64 quotient stored into an array of S's dtype; non-finite input raises (librosa's ParameterError)."""
    y = np.asarray(y)
    if not np.all(np.isfinite(y)):
        raise ValueError("Input must be finite")
    mag = np.abs(y).astype(float)
    length = np.max(mag, axis=0, keepdims=True) if y.size else np.ones(1)
    length[length < np.finfo(y.dtype).tiny] = 1.0
    out = np.empty_like(y)
    out[:] = y / length
    return out


def float_to_pcm16(y: np.ndarray) -> np.ndarray:
    """python-soundfile's sf.write of float32 data as PCM_16 (libsndfile pcm.c f2s_clip_array, clipping on):
    float32 product with 32768, >= 32767 -> 32767, <= -32768 -> -32768, else lrintf (round half to even)."""
    s = np.asarray(y, dtype=np.float32) * np.float32(32768.0)
    out = np.rint(s)
    out = np.where(s >= np.float32(32767.0), 32767, np.where(s <= np.float32(-32768.0), -32768, out))
    return out.astype(np.int16)


def clean_audio(y: np.ndarray, sr: int, prop_decrease: float = 1.0) -> np.ndarray:
    """pipeline1.py:139-141 end to end: int16 samples of the WAV the reference writes."""
    return float_to_pcm16(normalize(reduce_noise(np.asarray(y, dtype=np.float32), sr, prop_decrease)))
