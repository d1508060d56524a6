"""GPU diagnostic: throughput of the other BASELINE.json configurations on one MI355X (they are parity-test cases, not the
bench line): configs[2] = config 2 + the CNN head on the frame-level rows (torch / MIOpen, fp32 and fp16), the
REFERENCE_149 preset on 3 s clips, configs[4]-shaped variable-length 22.05 kHz clips, and the MAIN_286 rich preset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import CNNHead
from rsd_amd.synth import synth_bank

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
with torch.no_grad():
    t = timeit(lambda: ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True), 50)
    print("config 2  (BENCH_512, frames + pooled)            : %7.3f ms / 1024 clips  %9.0f clips/s" % (t * 1e3, 1024 / t))
    for dt, name in ((torch.float32, "fp32"), (torch.float16, "fp16")):
        cnn = CNNHead().cuda().to(dt)
        def f():
            r = ex.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)
            return cnn(r["frames"].to(dt))
        t = timeit(f, 10)
        print("config 3  (+ CNN head, torch/MIOpen %s)          : %7.3f ms / 1024 clips  %9.0f clips/s" % (name, t * 1e3, 1024 / t))
    ex149 = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0)
    t = timeit(lambda: ex149.extract_batch_device(pcm, off, 48000, pooled=True), 20)
    print("REFERENCE_149 (2048/512/128 mel/20 mfcc + chroma)   : %7.3f ms / 1024 clips  %9.0f clips/s" % (t * 1e3, 1024 / t))
    exm = rsd_amd.FeatureExtractor(rsd_amd.MAIN_286, 0)
    t = timeit(lambda: exm.extract_batch_device(pcm, off, 48000, pooled=True), 20)
    print("MAIN_286  (512/256/128 mel/40 mfcc + chroma + rich) : %7.3f ms / 1024 clips  %9.0f clips/s" % (t * 1e3, 1024 / t))
    # configs[4] shape: 1-10 s at 22.05 kHz, ragged, seed 4321
    rng = np.random.default_rng(4321)
    lens = (rng.uniform(1.0, 10.0, size=1024) * 22050).astype(np.int64)
    tot = int(lens.sum())
    big = torch.from_numpy(np.resize(bank.reshape(-1), tot)).cuda()
    o5 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).cuda()
    ex5 = rsd_amd.FeatureExtractor(rsd_amd.FeatureParams(sr=22050, chroma=True), 0)
    t = timeit(lambda: ex5.extract_batch_device(big, o5, int(lens.max()), pooled=True), 10)
    print("config 5 shape (1-10 s @ 22.05 kHz, 128 mel, ragged) : %7.3f ms / 1024 clips  %9.0f clips/s  (%.1f h of audio per s)" % (t * 1e3, 1024 / t, tot / 22050 / 3600 / t))
    from rsd_amd.classifier import FusedCNNHead
    cnn32 = CNNHead().cuda()
    fused = FusedCNNHead(cnn32)
    def g():
        r = ex.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)
        return fused(r["frames"])
    t = timeit(g, 20)
    print("config 3  (+ CNN head, HIP kernels, f16 MFMA conv2)   : %7.3f ms / 1024 clips  %9.0f clips/s" % (t * 1e3, 1024 / t))
    fr = ex.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)["frames"]
    t = timeit(lambda: fused(fr), 20)
    print("          CNN head alone (HIP)                        : %7.3f ms / 1024 clips" % (t * 1e3))
