"""Diagnostic: parity preset on a few fixture clips -- frame rows vs oracle, pooled vs NumPy statistics of the GPU's own rows."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rsd_amd
from oracle import dysf_oracle as O
d = os.path.join(ROOT, "tests", "golden", "ref_pairs")
ex = rsd_amd.get_extractor(rsd_amd.REFERENCE_149, 0)
for cid in ("00", "12", "13", "16"):
    pcm, sr = O.read_wav_pcm16(os.path.join(d, cid + ".wav")); gold = np.load(os.path.join(d, cid + ".npy"))
    y = O.pcm16_to_float(pcm)
    a = ex.extract_batch([pcm], frames=True, tuning=True)
    b = ex.extract_batch([pcm], tuning=True)
    fr = a["frames"][0]
    ref = O.frame_level_stack(y, O.REFERENCE_149)
    want = np.hstack([O.stat_pair(fr[:20]), O.stat_pair(fr[20:40]), O.stat_pair(fr[40:])])
    S = ex.stft_power(pcm); Sr = O.stft_power(y, O.REFERENCE_149).T
    print(cid, "T", fr.shape[1], "frames max|d|", float(np.abs(fr - ref).max()),
          "pooled(frames) vs np.stat of rows", float(np.abs(a["pooled"][0][:120] - want).max()),
          "pooled(stash) vs same", float(np.abs(b["pooled"][0][:120] - want).max()),
          "vs gold", float(np.abs(a["pooled"][0] - gold[:144]).max()), float(np.abs(b["pooled"][0] - gold[:144]).max()),
          "S mismatches", int((S != Sr).sum()), "of", S.size, "max rel", float((np.abs(S - Sr) / (Sr.max(axis=1, keepdims=True) + 1e-30)).max()),
          "tuning", int(a["tuning"][0]))
pcm, sr = O.read_wav_pcm16(os.path.join(d, "00.wav"))
np.save(os.path.join(ROOT, "gpurun_out", "S00.npy"), ex.stft_power(pcm))
a = ex.extract_batch([pcm], frames=True)
np.save(os.path.join(ROOT, "gpurun_out", "F00.npy"), a["frames"][0])
np.save(os.path.join(ROOT, "gpurun_out", "L00.npy"), ex.logmel(pcm))
