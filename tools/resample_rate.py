"""GPU diagnostic: throughput of the 22.05 kHz -> 16 kHz resampler on the bench batch shape (1024 clips x 3 s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.resample import Resampler
n, L = 1024, 66150
rng = np.random.default_rng(0)
pcm = torch.from_numpy(rng.integers(-20000, 20000, n * L, dtype=np.int16)).cuda()
off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
rs = Resampler(22050, 16000, 0)
for _ in range(3): rs.resample_batch_device(pcm, off, max_samples=L)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out, oo = rs.resample_batch_device(pcm, off, max_samples=L)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
byts = n * L * 2 + n * 48000 * 4
print("resample 1024 x 66150 int16 -> 48000 f32: %.1f us per batch, %.2f M clips/s, %.0f GB/s of algorithmic traffic" % (us, n / us, byts / us / 1e3))
