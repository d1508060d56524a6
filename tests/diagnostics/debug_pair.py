"""GPU-side diagnostic: where does the pooled-vector error of one reference pair come from?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsd_amd
from oracle import dysf_oracle as O

ids = sys.argv[1:] or ["02"]
d = os.path.join(ROOT, "tests", "golden", "ref_pairs")
ex = rsd_amd.get_extractor(rsd_amd.REFERENCE_149, 0)
p = O.REFERENCE_149
for cid in ids:
    pcm, sr = O.read_wav_pcm16(os.path.join(d, cid + ".wav"))
    gold = np.load(os.path.join(d, cid + ".npy"))
    y = O.pcm16_to_float(pcm)
    gpu = rsd_amd.extract_features(y, sr)
    orc = O.extract_features(y, sr)
    print(f"== pair {cid}  n={len(pcm)} T={p.n_frames(len(pcm))}")
    print(" |gpu-gold| max %.3e at %d ; |orc-gold| max %.3e ; |gpu-orc| max %.3e" % (
        np.abs(gpu - gold).max(), np.abs(gpu - gold).argmax(), np.abs(orc - gold).max(), np.abs(gpu - orc).max()))
    print(" gold[0], gold[20]:", gold[0], gold[20])
    S_g = ex.stft_power(pcm).T          # [K, T]
    S_o = O.stft_power(y, p)
    print(" S rel-to-max err: %.3e" % (np.abs(S_g - S_o).max() / S_o.max()))
    L_g = ex.logmel(pcm).T
    mel_o = O.mel_filterbank(p) @ S_o
    L_o = np.float32(10) * np.log10(np.maximum(np.float32(1e-10), mel_o))
    live = L_o > L_o.max() - 80
    print(" logmel err live max %.3e  mean %.3e ; frac live %.2f" % (np.abs(L_g - L_o)[live].max(), np.abs(L_g - L_o)[live].mean(), live.mean()))
    # oracle back-end on GPU front-ends
    def backend(L):
        Lc = np.maximum(L, L.max() - np.float32(80))
        m = O.mfcc_from_logmel(Lc.astype(np.float32), p)
        return np.hstack([O.stat_pair(m), O.stat_pair(O.delta(m, 1)), O.stat_pair(O.delta(m, 2))])
    f_LG = backend(L_g)
    f_SG = backend(np.float32(10) * np.log10(np.maximum(np.float32(1e-10), O.mel_filterbank(p) @ S_g)))
    print(" oracle backend on GPU logmel: |.-gold| %.3e ; on GPU S: %.3e ; on oracle S: %.3e" % (
        np.abs(f_LG - gold[:120]).max(), np.abs(f_SG - gold[:120]).max(), np.abs(backend(L_o) - gold[:120]).max()))
    # float64 exact back-end on GPU logmel
    def backend64(L):
        L = L.astype(np.float64)
        Lc = np.maximum(L, L.max() - 80.0)
        m = O.dct_matrix(p.n_mels, p.n_mfcc) @ Lc
        def dl(x, o):
            return O.delta(x, o)
        return np.hstack([m.mean(1), m.std(1), dl(m, 1).mean(1), dl(m, 1).std(1), dl(m, 2).mean(1), dl(m, 2).std(1)])
    print(" f64 backend on GPU logmel: |.-gold| %.3e ; |gpu - f64backend(gpu logmel)| %.3e" % (
        np.abs(backend64(L_g) - gold[:120]).max(), np.abs(gpu[:120] - backend64(L_g)).max()))
    print(" f64 backend on oracle logmel: |.-gold| %.3e" % np.abs(backend64(L_o) - gold[:120]).max())
