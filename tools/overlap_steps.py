"""GPU diagnostic: alternate whole steps between two streams (two extractor handles, so two log-mel workspaces): does
the MFCC kernel of one batch overlap usefully with the STFT kernel of the next?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
from rsd_amd.synth import synth_bank
bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
exs = [rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0) for _ in range(2)]
rng = np.random.default_rng(0)
dims = [exs[0].pooled_dim, 128, 64, 3]
mk = lambda: FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)], [np.zeros(dims[i + 1]) for i in range(3)])
heads = [mk(), mk()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, two):
    outs = []
    for i in range(n):
        k = i & 1 if two else 0
        with torch.cuda.stream(streams[k]):
            res = exs[k].extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
            lg = heads[k](res["pooled"])
        outs = [lg, res]
    return outs
for name, two in (("one stream", False), ("two streams", True), ("one stream", False), ("two streams", True)):
    run(10, two); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(400, two); torch.cuda.synchronize()
    print("%-12s %.1f us/step" % (name, (time.perf_counter() - t0) / 400 * 1e6))
