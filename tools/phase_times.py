"""GPU diagnostic: per-phase cycle shares of the clip kernel (needs `make -C .../csrc stamp`).
usage: DYSF_THREADS=256|512 python tools/phase_times.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DYSF_LIB"] = os.path.join(ROOT, "recognizing-speech-dysfluencies-in-stuttering_amd", "libdysf_hip_stamp.so")
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
L = rsd_amd._lib.lib()
L.dysf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
bank = synth_bank(64)
idx = np.arange(1024) % 64
pcm = torch.from_numpy(bank[idx].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
buf = (ctypes.c_uint64 * 32)()
L.dysf_debug_stamps(ex._h, buf, 1)
for _ in range(3):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 1)
n = 10
for _ in range(n):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 0)
names = ["setup+stage0", "fft", "bar1", "commit+mel", "bar2", "phaseB setup", "barB1", "dct", "barB2", "delta/stats/store"]
v = np.array(list(buf), dtype=np.float64) / n
for w, base in (("wave0", 0), ("waveLast", 16)):
    tot = v[base:base + 10].sum()
    print(w, "total cycles/launch summed over WGs: %.3e" % tot)
    for i, nm in enumerate(names):
        print("   %-20s %5.1f%%" % (nm, 100 * v[base + i] / tot))
