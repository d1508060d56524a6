"""Seed-stable synthetic clip bank of SURVEY.md section 8d (BASELINE.json configs[1..3]).

Pure NumPy input generator shared by bench.py and the tests; it produces INPUTS only (int16 PCM) and has no
numerics of the feature path in it.  Clip ``i`` depends only on ``i``: white noise + three harmonic stacks under
random on/off envelopes, peak-normalised to 0.9; every 64th clip is an edge case (all-zero, pure 1 kHz tone,
full-scale white noise, 0.2 s of content followed by silence)."""
import numpy as np


def synth_clip(i: int, n: int = 48000, sr: int = 16000) -> np.ndarray:
    """Seed-stable speech-like int16 clip ``i`` of the benchmark bank."""
    rng = np.random.Generator(np.random.PCG64(1234 + i))
    if i % 64 == 63:  # every 64th clip is an edge case
        kind = (i // 64) % 4
        if kind == 0:
            return np.zeros(n, dtype=np.int16)
        if kind == 1:
            t = np.arange(n) / sr
            return np.round(0.9 * 32767 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.int16)
        if kind == 2:
            return rng.integers(-32768, 32768, size=n, dtype=np.int64).astype(np.int16)
        y = np.zeros(n, dtype=np.int16)  # 0.2 s of real content, rest silence
        m = int(0.2 * sr)
        y[:m] = np.round(0.5 * 32767 * rng.standard_normal(m).clip(-1, 1)).astype(np.int16)
        return y
    t = np.arange(n) / sr
    y = 0.05 * rng.standard_normal(n)
    edges = np.sort(rng.integers(0, n, size=3))
    bounds = np.concatenate(([0], edges, [n]))
    for s in range(3):
        f0 = rng.uniform(80.0, 400.0)
        stack = np.zeros(n)
        for k in range(1, 9):
            stack += np.sin(2 * np.pi * f0 * k * t + rng.uniform(0, 2 * np.pi)) / k
        env = np.zeros(n)
        for seg in range(4):
            if rng.random() < 0.6:
                env[bounds[seg]:bounds[seg + 1]] = 1.0
        y += 0.3 * stack * env
    peak = np.max(np.abs(y))
    if peak > 0:
        y = 0.9 * y / peak
    return np.round(y * 32767).astype(np.int16)


def synth_bank(n_clips: int, n: int = 48000, sr: int = 16000, workers: int = 1) -> np.ndarray:
    """[n_clips, n] int16.  ``workers`` > 1 spreads the generation over a process pool."""
    if workers <= 1 or n_clips < 2 * workers:
        return np.stack([synth_clip(i, n, sr) for i in range(n_clips)])
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        rows = pool.starmap(synth_clip, [(i, n, sr) for i in range(n_clips)], chunksize=8)
    return np.stack(rows)
