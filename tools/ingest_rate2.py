"""GPU diagnostic: PCIe-inclusive rate of the streaming ingest with kernel A2 / kernel A (DYSF_STFT=1), pinned source."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.ingest import StreamingExtractor, pinned_empty
from rsd_amd.synth import synth_bank
bank = synth_bank(256, workers=8)
n = 16384
flat = pinned_empty(n * 48000)
flat.reshape(n // 256, -1)[:] = bank.reshape(1, -1)
offsets = np.arange(n + 1, dtype=np.int64) * 48000
for name, env, preset in (("A2", None, rsd_amd.BENCH_512), ("A", "1", rsd_amd.BENCH_512), ("ref149", None, rsd_amd.REFERENCE_149), ("A2", None, rsd_amd.BENCH_512)):
    if env: os.environ["DYSF_STFT"] = env
    ex = rsd_amd.FeatureExtractor(preset, 0)
    os.environ.pop("DYSF_STFT", None)
    for chunk in (2048, 4096):
        se = StreamingExtractor(ex, chunk_clips=chunk)
        se.run(flat[: 4096 * 48000], offsets[:4097])
        t0 = time.perf_counter(); out = se.run(flat, offsets); dt = time.perf_counter() - t0
        print(f"{name:7s} chunk {chunk}: {n/dt:9.0f} clips/s  ({n*96000/dt/1e9:5.1f} GB/s of PCM; copy stream {se.h2d_gbs:.1f} GB/s)")
