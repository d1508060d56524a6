"""GPU diagnostic: kernel A2 (wave-private STFT + mel) against kernel A on the same inputs, then same-process timing.

  python tools/a2_check.py [--no-time]

Compares the frame-level rows and pooled vectors of the two STFT kernels (DYSF_STFT=1 forces kernel A) on the bench
batch, on a ragged batch whose clips start at odd samples of the flat buffer, and on float32 PCM; then times both
(per-kernel HIP events inside the library, interleaved rounds in one process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank


def make(variant):
    os.environ["DYSF_STFT"] = str(variant)
    ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
    os.environ.pop("DYSF_STFT")
    return ex


def cmp(name, a, b):
    ok = True
    for k in ("pooled", "frames", "status"):
        if k not in a: continue
        x, y = a[k], b[k]
        if x.dtype.kind == "f":
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            tol = 2e-4 + 2e-6 * np.abs(x)
            bad = int((d > tol).sum())
            print(f"  {name:18s} {k:7s} max |d| {d.max():.3e}  over tol: {bad} of {d.size}  finite: {bool(np.isfinite(y).all())}")
            ok &= bad == 0
        else:
            eq = bool((x == y).all())
            print(f"  {name:18s} {k:7s} equal: {eq}")
            ok &= eq
    return ok


def main():
    ex1, ex2 = make(1), make(0)
    bank = synth_bank(64, workers=8)
    ok = True
    # 1. bench-shaped batch through the host entry point
    clips = [bank[i % 64] for i in range(256)]
    ok &= cmp("bench batch", ex1.extract_batch(clips, frames=True), ex2.extract_batch(clips, frames=True))
    # 2. ragged lengths, odd starts, short and empty clips
    rng = np.random.default_rng(7)
    rag = []
    for i in range(200):
        n = int(rng.integers(1281, 48001))
        rag.append(bank[i % 64][:n].copy())
    rag += [bank[0][:1279], bank[1][:1280], bank[2][:1281], bank[3][:0], bank[4][:1], bank[5][:47999], bank[6][:161 * 7 + 3]]
    ok &= cmp("ragged batch", ex1.extract_batch(rag, frames=True), ex2.extract_batch(rag, frames=True))
    # 3. float32 PCM, one clip with a NaN
    ragf = [(c.astype(np.float32) / 32768.0) for c in rag[:40]]
    ragf[3] = ragf[3].copy(); ragf[3][1000] = np.nan
    ragf[5] = ragf[5].copy(); ragf[5][5] = np.inf
    r1, r2 = ex1.extract_batch(ragf, frames=True), ex2.extract_batch(ragf, frames=True)
    ok &= cmp("float32 PCM", r1, r2)
    print("  statuses:", r2["status"][:8])
    print("A2 == A within tolerance:", ok)
    if "--no-time" in sys.argv:
        return 0 if ok else 1
    pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
    off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
    exs = {"A": ex1, "A2": ex2}
    for ex in exs.values():
        for _ in range(60): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
        ex.kernel_timing(True)
    ts = {k: [] for k in exs}
    for rnd in range(6):
        for k, ex in exs.items():
            for _ in range(20):
                ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
                ts[k].append(ex.kernel_timing(True))
    for k in exs:
        st = 1e3 * np.array([t["stft_mel"] for t in ts[k]]); mf = 1e3 * np.array([t["mfcc"] for t in ts[k]])
        print(f"{k:3s} stft+mel median {np.median(st):.1f} us  min {st.min():.1f}   mfcc median {np.median(mf):.1f} us")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
