"""GPU diagnostic: throughput of the cleaning stage (dysf_clean_batch_dev) on 3 s clips resident in HBM, and its
agreement with oracle/dysf_clean_oracle.py on a few of them (test infrastructure use of the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
from oracle import dysf_clean_oracle as CO

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bank = synth_bank(64)
rng = np.random.default_rng(0)
noisy = (bank.astype(np.float32) / 32768 * 0.5 + 0.01 * rng.standard_normal(bank.shape)).astype(np.float32)
pcm = torch.from_numpy(noisy[np.arange(n) % 64].reshape(-1)).cuda()
off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * 48000
cl = rsd_amd.AudioCleaner(16000, 0)
for _ in range(2):
    res = cl.clean_batch_device(pcm, off, 48000, denoised=True, pcm16=True)
torch.cuda.synchronize()
t = time.perf_counter()
reps = 5
for _ in range(reps):
    res = cl.clean_batch_device(pcm, off, 48000, denoised=True, pcm16=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"cleaning stage: {n} clips of 3 s in {dt * 1e3:.2f} ms = {n / dt:.0f} clips/s ({n * 3 / dt:.0f}x real time)")
q = res["pcm16"].cpu().numpy().reshape(n, -1)
den = res["denoised"].cpu().numpy().reshape(n, -1)
t = time.perf_counter()
mism = tot = 0
worst = 0.0
for i in range(8):
    ref = CO.reduce_noise(noisy[i], 16000)
    worst = max(worst, float(np.abs(den[i].astype(np.float64) - ref).max() / np.abs(ref).max()))
    rq = CO.float_to_pcm16(CO.normalize(ref))
    mism += int((rq != q[i]).sum()); tot += rq.size
cpu = (time.perf_counter() - t) / 8
print(f"vs oracle on 8 clips: {mism} of {tot} int16 samples differ, max relative float32 difference {worst:.2e}; "
      f"oracle (SciPy, 1 core) {cpu * 1e3:.1f} ms per clip = {1 / cpu:.1f} clips/s")
