"""GPU diagnostic: PCIe-inclusive clips/s of the pooled REFERENCE-free bench preset via the streaming ingest
(host int16 -> pinned double-buffered H2D -> kernels -> pooled rows on the host)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.ingest import StreamingExtractor
from rsd_amd.synth import synth_bank
bank = synth_bank(256, workers=8)
n = 16384
flat = np.tile(bank.reshape(-1), n // 256)
offsets = np.arange(n + 1, dtype=np.int64) * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for threads in (1, 4, 8, 16):
    se = StreamingExtractor(ex, chunk_clips=2048, copy_threads=threads)
    se.run(flat[: 4096 * 48000], offsets[:4097])
    t0 = time.perf_counter(); out = se.run(flat, offsets); dt = time.perf_counter() - t0
    assert int(np.abs(out["status"]).sum()) == 0
    print(f"streaming ingest, {threads:2d} copy threads: {n} clips in {dt*1e3:.1f} ms -> {n/dt:.0f} clips/s ({n*96000/dt/1e9:.1f} GB/s of int16 PCM over PCIe)")
