"""GPU diagnostic: PCIe-inclusive rate of the streaming ingest against the chunk size (clips per H2D copy / launch), three
repeats each, pinned source, default preset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.ingest import StreamingExtractor, pinned_empty
from rsd_amd.synth import synth_bank
bank = synth_bank(256, workers=8)
n = 32768
flat = pinned_empty(n * 48000)
flat.reshape(n // 256, -1)[:] = bank.reshape(1, -1)
offsets = np.arange(n + 1, dtype=np.int64) * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for rep in range(3):
    for chunk in (1024, 2048, 4096, 8192):
        se = StreamingExtractor(ex, chunk_clips=chunk)
        if rep == 0: se.run(flat[: 2 * chunk * 48000], offsets[: 2 * chunk + 1])
        t0 = time.perf_counter(); out = se.run(flat, offsets); dt = time.perf_counter() - t0
        print(f"rep {rep} chunk {chunk:5d}: {n/dt:9.0f} clips/s  ({n*96000/dt/1e9:5.1f} GB/s of PCM; copy stream {se.h2d_gbs:.1f} GB/s)", flush=True)
