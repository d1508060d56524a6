"""GPU diagnostic: feature launches of several presets on 3 lanes while the float16 CNN runs on a side stream over fixed
frames.  Which feature kernels change their output when that CNN kernel shares the GPU with them?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import CNNHead, FusedCNNHead
from rsd_amd.pipeline import LaneExecutor
from rsd_amd.synth import synth_bank
dev = torch.device("cuda", 0)
bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(512) % 64].reshape(-1)).to(dev)
off = (torch.arange(0, 513, dtype=torch.int64) * 48000).to(dev)
torch.manual_seed(0)
cnn = FusedCNNHead(CNNHead().to(dev))
fixed = torch.randn(128, 60, 301, device=dev) * 20 - 30
import ctypes
L = rsd_amd._lib.lib()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(dev)
presets = {"BENCH_512 (A2 + MFCC f32-MFMA)": (rsd_amd.BENCH_512, dict(pooled=True, frames=True)),
           "BENCH_512 kernel A (DYSF_STFT=1)": (rsd_amd.BENCH_512, dict(pooled=True, frames=True)),
           "REFERENCE_149 (f64 FFT, f64-MFMA DCT, chroma)": (rsd_amd.REFERENCE_149, dict(pooled=True, tuning=True)),
           "22.05 kHz 2048/512/128 non-parity": (rsd_amd.FeatureParams(sr=22050, chroma=False, parity=False), dict(pooled=True, frames=True))}
for name, (p, kw) in presets.items():
    if "DYSF_STFT=1" in name: os.environ["DYSF_STFT"] = "1"
    lanes = LaneExecutor(p, 0, 3)
    os.environ.pop("DYSF_STFT", None)
    def run(with_cnn, poison=False, stress=None):
        outs = []
        lanes.fork()
        for i in range(9):
            k = i % 3
            with torch.cuda.stream(lanes.streams[k]):
                outs.append(lanes.extractors[k].extract_batch_device(pcm, off, 48000, **kw))
            if with_cnn:
                with torch.cuda.stream(side):
                    cnn(fixed)
            if stress is not None:
                for _ in range(2):
                    L.dysf_debug_stress(ctypes.c_void_p(side.cuda_stream), stress, 1024, 3000, 40 * 1024, ctypes.c_void_p(sink.data_ptr()))
            if poison:
                for _ in range(6):
                    L.dysf_debug_poison_vgpr(ctypes.c_void_p(side.cuda_stream), 2048, ctypes.c_void_p(sink.data_ptr()))
        lanes.join(); torch.cuda.synchronize()
        return outs
    ref = run(False)
    if hasattr(L, "dysf_debug_poison_vgpr"):
        got = run(False, poison=True)
        print(name, "| VGPR poison on the side stream | launches whose frames / pooled differ:",
              [i for i in range(9) if not torch.equal(ref[0]["pooled"], got[i]["pooled"])], " non-finite:", [i for i in range(9) if not bool(torch.isfinite(got[i]["pooled"]).all())])
    if hasattr(L, "dysf_debug_stress"):
        for kind, what in enumerate(("barriers + LDS", "float16 MFMA", "float32 MFMA", "ds_read_b128")):
            got = run(False, stress=kind)
            print(name, "| stress:", what, "| launches whose pooled differ:", [i for i in range(9) if not torch.equal(ref[0]["pooled"], got[i]["pooled"])])
    for rep in range(2):
        got = run(True)
        bad = {}
        for key in ("pooled", "frames", "tuning"):
            if key in ref[0]:
                bad[key] = [i for i in range(9) if not torch.equal(ref[0][key], got[i][key])]
        print(name, "| pass", rep, "| launches (of 9) whose output differs from the CNN-free run:", bad)
