"""GPU diagnostic: is the bucketed configs[4] pass (ragged feature launch + float16 CNN per bucket, lanes) deterministic?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import CNNHead, FusedCNNHead
from rsd_amd.pipeline import LaneExecutor
from rsd_amd.synth import synth_bank
dev = torch.device("cuda", 0)
bank = synth_bank(256, workers=8)
src = torch.from_numpy(bank.reshape(-1)).to(dev)
sr = 22050
rng = np.random.default_rng(4321)
n = 1024
lens = (rng.uniform(1.0, 10.0, size=n) * sr).astype(np.int64)
starts = rng.integers(0, src.numel() - int(lens.max()), size=n)
params = rsd_amd.FeatureParams(sr=22050, chroma=False, parity=False)
sec = np.minimum(10, np.ceil(lens / sr).astype(np.int64))
buckets = []
for b in range(1, 11):
    idx = np.nonzero(sec == b)[0]
    if len(idx) == 0: continue
    bl = lens[idx]
    boff = np.concatenate([[0], np.cumsum(bl)]).astype(np.int64)
    bpcm = torch.cat([src[int(s0):int(s0) + int(m)] for s0, m in zip(starts[idx], bl)])
    buckets.append({"idx": torch.from_numpy(idx).to(dev), "pcm": bpcm, "off": torch.from_numpy(boff).to(dev), "max": int(bl.max())})
torch.manual_seed(0)
cnn = FusedCNNHead(CNNHead().to(dev))
torch.cuda.synchronize()

USE_CNN = True
def run(n_lanes, lanes):
    logits = torch.empty((n, 3), dtype=torch.float32, device=dev)
    feats = []
    for i, b in enumerate(buckets):
        k = i % n_lanes
        with torch.cuda.stream(lanes.streams[k]):
            r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
            cols = ((b["off"][1:] - b["off"][:-1]) // params.hop + 1).to(torch.int32)
            if USE_CNN: logits.index_copy_(0, b["idx"], cnn(r["frames"], cols=cols))
            feats.append(r["frames"])
    lanes.join(); torch.cuda.synchronize()
    return logits, feats

for nl in (1, 3):
    lanes = LaneExecutor(params, 0, nl)
    lanes.fork()
    a, fa = run(nl, lanes)
    lanes.fork()
    b, fb = run(nl, lanes)
    print("lanes", nl, "logits max |d| between two passes:", float((a - b).abs().max()), " |logits| max", float(a.abs().max()),
          " frames max |d|:", max(float((x - y).abs().max()) for x, y in zip(fa, fb)))
    if nl == 1: ref = a; ref_f = fa
    else: print("3 lanes vs 1 lane: logits max |d|", float((a - ref).abs().max()))
# the CNN alone, same frames twice
fr = fa[3]; b3 = buckets[3]
cols = ((b3["off"][1:] - b3["off"][:-1]) // params.hop + 1).to(torch.int32)
o1 = cnn(fr, cols=cols).clone(); o2 = cnn(fr, cols=cols).clone(); torch.cuda.synchronize()
print("CNN twice on the same frames:", float((o1 - o2).abs().max()))

# which buckets differ, and does the feature path alone (no CNN on the lanes) stay deterministic?
lanes = LaneExecutor(params, 0, 3)
lanes.fork(); a, fa = run(3, lanes)
for rep in range(3):
    lanes.fork(); b, fb = run(3, lanes)
    bad = [(i, int(((x - y).abs() > 0).sum()), tuple(x.shape)) for i, (x, y) in enumerate(zip(fa, fb)) if not torch.equal(x, y)]
    print("3 lanes with CNN, pass", rep, "buckets whose frames differ from the first pass:", [(i, c) for i, c, _ in bad])
    if bad and rep == 0:
        i = bad[0][0]
        x, y = fa[i], fb[i]
        dm = (x - y).abs() > 0
        clips_bad = torch.nonzero(dm.any(dim=2).any(dim=1)).flatten().tolist()
        print("  bucket", i, "clips affected:", len(clips_bad), "of", x.shape[0], clips_bad[:12])
        c = clips_bad[0]
        rows_bad = torch.nonzero(dm[c].any(dim=1)).flatten().tolist()
        cols_bad = torch.nonzero(dm[c].any(dim=0)).flatten().tolist()
        print("  clip", c, "rows", rows_bad[:8], "..", len(rows_bad), "cols", cols_bad[:6], "..", cols_bad[-3:], len(cols_bad))
        print("  first pass values", x[c, rows_bad[0], cols_bad[0]:cols_bad[0] + 4].tolist(), "second", y[c, rows_bad[0], cols_bad[0]:cols_bad[0] + 4].tolist())
        print("  which equals the 1-lane reference? first:", bool(torch.equal(x, ref_f[i])), "second:", bool(torch.equal(y, ref_f[i])))
USE_CNN = False
lanes.fork(); a, fa = run(3, lanes)
for rep in range(3):
    lanes.fork(); b, fb = run(3, lanes)
    bad = [(i, int(((x - y).abs() > 0).sum())) for i, (x, y) in enumerate(zip(fa, fb)) if not torch.equal(x, y)]
    print("3 lanes, features only, pass", rep, "differing buckets:", bad)

# E1: the lanes run the feature launches + an unrelated torch GEMM instead of the CNN
USE_CNN = False
A_ = torch.randn(2048, 2048, device=dev); 
def run_gemm(n_lanes, lanes):
    feats = []
    for i, b in enumerate(buckets):
        k = i % n_lanes
        with torch.cuda.stream(lanes.streams[k]):
            r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
            for _ in range(3): (A_ @ A_).sum()
            feats.append(r["frames"])
    lanes.join(); torch.cuda.synchronize()
    return feats
lanes.fork(); fa = run_gemm(3, lanes)
for rep in range(3):
    lanes.fork(); fb = run_gemm(3, lanes)
    print("E1 3 lanes, features + torch GEMMs, pass", rep, "differing buckets:", [i for i, (x, y) in enumerate(zip(fa, fb)) if not torch.equal(x, y)],
          "vs 1-lane reference:", [i for i, (x, y) in enumerate(zip(ref_f, fb)) if not torch.equal(x, y)])
# E2: features on 3 lanes, the CNN on its own stream over FIXED frames (bucket 5 of the reference pass), concurrently
side = torch.cuda.Stream(dev)
cols_of = [((b["off"][1:] - b["off"][:-1]) // params.hop + 1).cpu().numpy() for b in buckets]
fr5 = ref_f[5].clone(); b5 = buckets[5]
cols5 = ((b5["off"][1:] - b5["off"][:-1]) // params.hop + 1).to(torch.int32)
torch.cuda.synchronize()
def run_side(n_lanes, lanes):
    feats = []
    for i, b in enumerate(buckets):
        k = i % n_lanes
        with torch.cuda.stream(lanes.streams[k]):
            r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
            feats.append(r["frames"])
        with torch.cuda.stream(side):
            cnn(fr5, cols=cols5)
    lanes.join(); torch.cuda.synchronize()
    return feats
for rep in range(3):
    lanes.fork(); fb = run_side(3, lanes)
    if rep == 0:
        for i in (1, 4):
            dm = (ref_f[i] - fb[i]).abs() > 0
            for c in torch.nonzero(dm.any(dim=2).any(dim=1)).flatten().tolist()[:6]:
                cols_bad = torch.nonzero(dm[c].any(dim=0)).flatten().tolist()
                nrows = int(dm[c].any(dim=1).sum())
                Tc = int(cols_of[i][c])
                print("   E2 bucket", i, "clip", c, "T", Tc, "rows", nrows, "cols", cols_bad[0], "..", cols_bad[-1], "n", len(cols_bad),
                      "max |d|", float((ref_f[i][c] - fb[i][c]).abs().max()))
    print("E2 3 lanes features, CNN on a side stream over fixed frames, pass", rep, "buckets differing from the 1-lane reference:",
          [i for i, (x, y) in enumerate(zip(ref_f, fb)) if not torch.equal(x, y)], " fixed frames intact:", bool(torch.equal(fr5, ref_f[5])))

# E3 (stamp library only): features on 3 lanes while a side stream runs a kernel that fills LDS with NaN patterns
import ctypes
L = rsd_amd._lib.lib()
if hasattr(L, "dysf_debug_poison_lds"):
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    def run_poison(n_lanes, lanes, nbytes):
        feats = []
        for i, b in enumerate(buckets):
            k = i % n_lanes
            with torch.cuda.stream(lanes.streams[k]):
                r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
                feats.append(r["frames"])
            for _ in range(4):
                L.dysf_debug_poison_lds(ctypes.c_void_p(side.cuda_stream), 4096, nbytes, ctypes.c_void_p(sink.data_ptr()))
        lanes.join(); torch.cuda.synchronize()
        return feats
    for nbytes in (160 * 1024, 64 * 1024):
        for rep in range(2):
            lanes.fork(); fb = run_poison(3, lanes, nbytes)
            print("E3 LDS poison", nbytes, "pass", rep, "buckets differing from the 1-lane reference:",
                  [i for i, (x, y) in enumerate(zip(ref_f, fb)) if not torch.equal(x, y)], "non-finite:", [i for i, y in enumerate(fb) if not bool(torch.isfinite(y).all())])

# E4: the same with the float32 CNN build, and with the float16 CNN on the narrowest bucket's frames
torch.manual_seed(0)
cnn32 = FusedCNNHead(CNNHead().to(dev), precision="f32")
fr0 = ref_f[0].clone(); b0 = buckets[0]
cols0 = ((b0["off"][1:] - b0["off"][:-1]) // params.hop + 1).to(torch.int32)
torch.cuda.synchronize()
def run_side2(n_lanes, lanes, fn):
    feats = []
    for i, b in enumerate(buckets):
        k = i % n_lanes
        with torch.cuda.stream(lanes.streams[k]):
            r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
            feats.append(r["frames"])
        with torch.cuda.stream(side):
            fn()
    lanes.join(); torch.cuda.synchronize()
    return feats
for name, fn in (("f32 CNN, bucket 5 frames", lambda: cnn32(fr5, cols=cols5)), ("f16 CNN, bucket 0 frames (narrow)", lambda: cnn(fr0, cols=cols0)),
                 ("f16 CNN, bucket 5 frames, no per-clip widths", lambda: cnn(fr5))):
    for rep in range(2):
        lanes.fork(); fb = run_side2(3, lanes, fn)
        print("E4", name, "pass", rep, "buckets differing from the 1-lane reference:", [i for i, (x, y) in enumerate(zip(ref_f, fb)) if not torch.equal(x, y)])
