"""GPU diagnostic: per-phase cycle shares of kernel A-exact (thread 0 of every workgroup; needs `make -C .../csrc stamp`)."""
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
ex = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0)
buf = (ctypes.c_uint64 * 16)()
L.dysf_debug_stamps(ex._h, buf, 1)
for _ in range(10): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 1)
n = 10
for _ in range(n): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
L.dysf_debug_stamps(ex._h, buf, 0)
v = np.array(list(buf), dtype=np.float64) / n
names = {0: "setup (tables, first loads)", 1: "convert + window + stage 0", 2: "two exchanges", 3: "twiddles + stage 1", 4: "twiddles + stage 2",
         5: "un-pack, |X|^2, rows", 6: "local maxima", 7: "issue next loads", 13: "barriers", 14: "mel product", 15: "finisher"}
tot = sum(v[i] for i in names)
frames = 1024 * 94 / 8
print(f"kernel A-exact: thread-0 counts per launch {tot:.3e}; per frame of wave 0: {tot / frames:.0f}")
for i, nm in names.items():
    print(f"   {nm:30s} {100 * v[i] / tot:5.1f}%   per frame {v[i] / frames:8.0f}")
