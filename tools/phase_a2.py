"""GPU diagnostic: per-phase cycle shares of kernel A2 (needs `make -C .../csrc stamp`).  Shares only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DYSF_LIB"] = os.path.join(ROOT, "recognizing-speech-dysfluencies-in-stuttering_amd", "libdysf_hip_stamp.so")
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
L = rsd_amd._lib.lib()
L.dysf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
bank = synth_bank(32)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
buf = (ctypes.c_uint64 * 16)()
L.dysf_debug_stamps(ex._h, buf, 1)
for _ in range(20): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 1)
n = 10
for _ in range(n): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 0)
v = np.array(list(buf), dtype=np.float64) / n
names = ["setup", "cvt+win+stage0", "exchange", "tw+stage1", "unpack+rows", "mel mfma", "db+store"]
tot = v[:7].sum()
print(f"kernel A2: thread-0 cycles per launch summed over workgroups {tot:.3e}  (per workgroup {tot / 1024:.0f})")
for i, nm in enumerate(names):
    print(f"   {nm:18s} {100 * v[i] / tot:5.1f}%   per quad {v[i] / 1024 / 19:8.0f} cycles")
