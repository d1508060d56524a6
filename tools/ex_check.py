"""GPU diagnostic: kernel A-exact (parity presets at n_fft 2048) against kernel A's float64 path on the same inputs.

  python tools/ex_check.py [--no-time]

DYSF_EXACT=1 keeps kernel A.  The two run the same arithmetic, so pooled vectors, frame rows, tuning indices and statuses
must be bit-equal: on the bench batch, on ragged clips that start at odd samples, on float32 PCM with non-finite samples,
and on silent clips (the general |X|^2 sequence).  Then per-kernel HIP-event timings of both, interleaved in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank


def make(variant):
    os.environ["DYSF_EXACT"] = str(variant)
    ex = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0)
    os.environ.pop("DYSF_EXACT")
    return ex


def cmp(name, a, b):
    ok = True
    for k in ("pooled", "frames", "status", "tuning"):
        if k not in a or a[k] is None: continue
        x, y = np.asarray(a[k]), np.asarray(b[k])
        eq = bool(np.array_equal(x, y, equal_nan=True))
        extra = ""
        if not eq and x.dtype.kind == "f":
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            extra = f"  max |d| {np.nanmax(d):.3e}, {int((d > 0).sum())} of {d.size} differ"
        print(f"  {name:18s} {k:7s} bit-equal: {eq}{extra}")
        ok &= eq
    return ok


def main():
    ex1, ex2 = make(1), make(0)
    sr = rsd_amd.REFERENCE_149.sr
    bank = synth_bank(64, n=3 * sr, sr=sr, workers=8)
    ok = True
    if "--time-only" in sys.argv:
        return timing(ex1, ex2, bank, sr)
    clips = [bank[i % 64] for i in range(128)]
    ok &= cmp("bench batch", ex1.extract_batch(clips, frames=True, tuning=True), ex2.extract_batch(clips, frames=True, tuning=True))
    rng = np.random.default_rng(7)
    rag = []
    for i in range(100):
        n = int(rng.integers(4097, len(bank[0]) + 1))
        rag.append(bank[i % 64][:n].copy())
    rag += [bank[0][:4095], bank[1][:4096], bank[2][:4097], bank[3][:0], bank[4][:1], bank[5][:len(bank[5]) - 1], bank[6][:512 * 9 + 3],
            np.zeros(20000, np.int16), np.concatenate([bank[7][:9000], np.zeros(9000, np.int16), bank[7][:9000]])]
    ok &= cmp("ragged batch", ex1.extract_batch(rag, frames=True, tuning=True), ex2.extract_batch(rag, frames=True, tuning=True))
    ragf = [(c.astype(np.float32) / 32768.0) for c in rag[:40]]
    ragf[3] = ragf[3].copy(); ragf[3][1000] = np.nan
    ragf[5] = ragf[5].copy(); ragf[5][5] = np.inf
    r1, r2 = ex1.extract_batch(ragf, frames=True, tuning=True), ex2.extract_batch(ragf, frames=True, tuning=True)
    ok &= cmp("float32 PCM", r1, r2)
    print("  statuses:", r2["status"][:8])
    print("A-exact == A bit for bit:", ok)
    if "--no-time" in sys.argv:
        return 0 if ok else 1
    timing(ex1, ex2, bank, sr)
    return 0 if ok else 1


def timing(ex1, ex2, bank, sr):
    n3 = 3 * sr
    pcm = torch.from_numpy(np.stack([bank[i % 64][:n3] for i in range(1024)]).reshape(-1)).cuda()
    off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * n3
    exs = {"A": ex1, "A-exact": ex2}
    for ex in exs.values():
        for _ in range(20): ex.extract_batch_device(pcm, off, n3, pooled=True, frames=True)
        ex.kernel_timing(True)
    ts = {k: [] for k in exs}
    for rnd in range(4):
        for k, ex in exs.items():
            for _ in range(10):
                ex.extract_batch_device(pcm, off, n3, pooled=True, frames=True)
                ts[k].append(ex.kernel_timing(True))
    for k in exs:
        st = 1e3 * np.array([t["stft_mel"] for t in ts[k]]); mf = 1e3 * np.array([t["mfcc"] for t in ts[k]]); ch = 1e3 * np.array([t["chroma"] for t in ts[k]])
        print(f"{k:8s} stft+mel median {np.median(st):.1f} us  min {st.min():.1f}   mfcc median {np.median(mf):.1f} us   chroma etc. median {np.median(ch):.1f} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
