"""GPU diagnostic: per-phase cycle shares of the STFT+mel and MFCC kernels (needs `make -C .../csrc stamp`).
Shares only: the stamps themselves perturb the timing."""
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
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
preset = {"bench": rsd_amd.BENCH_512, "ref": rsd_amd.REFERENCE_149}[sys.argv[1] if len(sys.argv) > 1 else "bench"]
want_frames = preset is rsd_amd.BENCH_512
ex = rsd_amd.FeatureExtractor(preset, 0)
buf = (ctypes.c_uint64 * 16)()
L.dysf_debug_stamps(ex._h, buf, 1)
for _ in range(3):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=want_frames)
L.dysf_debug_stamps(ex._h, buf, 1)
n = 10
for _ in range(n):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=want_frames)
L.dysf_debug_stamps(ex._h, buf, 0)
v = np.array(list(buf), dtype=np.float64) / n
v[6] += 0.0
for title, names, base in (("stft+mel kernel", ["setup", "prefetch+fft", "barrier", "commit", "mel", "barrier", "finisher", "piptrack"], 0),
                           ("stft+mel kernel, spectrogram store", ["S store"], 14),
                           ("mfcc kernel", ["setup", "barrier(+deltas)", "dct", "barrier", "last deltas", "final reduce"], 8)):
    tot = v[base:base + len(names)].sum()
    print(f"{title}: thread-0 cycles per launch summed over workgroups {tot:.3e}")
    for i, nm in enumerate(names):
        print(f"   {nm:18s} {100 * v[base + i] / tot:5.1f}%   {v[base + i]:.3e}")
