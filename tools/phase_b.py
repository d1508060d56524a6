"""GPU diagnostic: per-phase cycle shares of the MFCC kernel in the parity presets (thread 0 of every workgroup; needs
`make -C .../csrc stamp`).  Run with DYSF_EXACT=1 and without chroma so that no other kernel writes the same stamp slots."""
import ctypes, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DYSF_LIB"] = os.path.join(ROOT, "recognizing-speech-dysfluencies-in-stuttering_amd", "libdysf_hip_stamp.so")
os.environ["DYSF_EXACT"] = "1"
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
L = rsd_amd._lib.lib()
L.dysf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
bank = synth_bank(32)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(dataclasses.replace(rsd_amd.REFERENCE_149, chroma=False), 0)
buf = (ctypes.c_uint64 * 16)()
L.dysf_debug_stamps(ex._h, buf, 1)
for _ in range(10): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=False)
L.dysf_debug_stamps(ex._h, buf, 1)
n = 10
for _ in range(n): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=False)
L.dysf_debug_stamps(ex._h, buf, 0)
v = np.array(list(buf), dtype=np.float64) / n / 1024
names = {8: "setup (tables, first fetch)", 9: "barrier (+ deltas of earlier chunks)", 10: "DCT", 11: "barrier", 12: "deltas, frame rows", 13: "statistics"}
tot = sum(v[i] for i in names)
print(f"MFCC kernel, parity preset: thread-0 counts per workgroup {tot:.0f}")
for i, nm in names.items():
    print(f"   {nm:38s} {v[i]:9.0f}  {100 * v[i] / tot:5.1f}%")
