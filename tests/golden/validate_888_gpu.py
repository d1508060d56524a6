"""One-off GPU validation against ALL 888 reference pairs (clear_audio/<stem>.wav -> cache_features/<stem>_clean_feats.npy).
The pairs are copied into tests/golden/_all888/ (git-ignored, never committed) by the build container; this script runs on
the GPU box and writes the summary JSON that IS committed (tests/golden/validate_888_gpu.json)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rsd_amd
from rsd_amd.ingest import read_wav_pcm16
from oracle import dysf_oracle as O

d = os.path.join(ROOT, "tests", "golden", "_all888")
idx = json.load(open(os.path.join(d, "index.json")))
ex = rsd_amd.get_extractor(rsd_amd.REFERENCE_149, 0)
clips, golds = [], []
for e in idx:
    pcm, sr = read_wav_pcm16(os.path.join(d, e["id"] + ".wav"))
    assert sr == 16000
    clips.append(np.array(pcm)); golds.append(np.load(os.path.join(d, e["id"] + ".npy")))
res = ex.extract_batch(clips, tuning=True)
G = np.stack(golds)
F = np.hstack([res["pooled"], np.zeros((len(clips), 5), np.float32)])
D = np.abs(F.astype(np.float64) - G.astype(np.float64))
blocks = {"mfcc_stats": (0, 40), "delta_stats": (40, 80), "delta2_stats": (80, 120), "chroma_stats": (120, 144), "text": (144, 149)}
per_clip_audio = D[:, :120].max(axis=1)
per_clip_chroma = D[:, 120:144].max(axis=1)
# tuning index vs the oracle on the clips whose chroma block disagrees
tun_mis = []
# tuning index of EVERY clip against the oracle's (the histogram arg-max is a tie or near-tie in most clips)
tun_all_equal = 0
for i in range(len(clips)):
    _, dbg = O.estimate_tuning(O.stft_power(O.pcm16_to_float(clips[i]), O.REFERENCE_149), O.REFERENCE_149)
    tun_all_equal += int(res["tuning"][i] == dbg["index"])
for i in np.nonzero(per_clip_chroma > 1e-4)[0]:
    _, dbg = O.estimate_tuning(O.stft_power(O.pcm16_to_float(clips[i]), O.REFERENCE_149), O.REFERENCE_149)
    tun_mis.append({"stem": idx[i]["stem"], "gpu_index": int(res["tuning"][i]), "oracle_index": int(dbg["index"]),
                    "chroma_max_abs_diff": float(per_clip_chroma[i])})
# per-file metrics (pipeline1.py:151-186) of all 888 clips against the reference's per_file_analysis.csv values
met = None
have = [i for i, e in enumerate(idx) if e.get("metrics_after")]
if have:
    gm = rsd_amd.audio_metrics_batch([clips[i] for i in have], 16000)
    wm = np.array([idx[i]["metrics_after"] for i in have])
    met = {"n_clips": len(have), "snr_db_max_abs_diff": float(np.abs(gm[:, 0] - wm[:, 0]).max()),
           "flatness_max_rel_diff": float((np.abs(gm[:, 1] - wm[:, 1]) / np.maximum(np.abs(wm[:, 1]), 1e-12)).max()),
           "hf_ratio_max_rel_diff": float((np.abs(gm[:, 2] - wm[:, 2]) / np.maximum(np.abs(wm[:, 2]), 1e-12)).max())}
rec = {
    "what": "GPU path (rsd_amd REFERENCE_149, one batched launch of 888 clips) vs reference cache_features/*_clean_feats.npy",
    "n_pairs": len(clips), "all_status_ok": bool((res["status"] == 0).all()),
    "max_abs_diff": float(D.max()), "block_max_abs_diff": {k: float(D[:, a:b].max()) for k, (a, b) in blocks.items()},
    "mfcc_delta_blocks": {"max": float(per_clip_audio.max()), "p99": float(np.percentile(per_clip_audio, 99)),
                          "median": float(np.median(per_clip_audio)), "n_over_1e-4": int((per_clip_audio > 1e-4).sum()),
                          "n_over_6.2e-5": int((per_clip_audio > 6.2e-5).sum()),
                          "clips_over_6.2e-5": [{"stem": idx[i]["stem"], "max_abs_diff": float(per_clip_audio[i]),
                                                 "index": int(D[i, :120].argmax())} for i in np.nonzero(per_clip_audio > 6.2e-5)[0]]},
    "chroma_block": {"n_over_1e-4": int((per_clip_chroma > 1e-4).sum()), "median": float(np.median(per_clip_chroma)),
                     "clips_over_1e-4": tun_mis},
    "tuning_index_equal_to_oracle": tun_all_equal,
    "metrics_vs_per_file_analysis_csv": met,
    "n_clips_all_149_within_1e-4": int((D.max(axis=1) <= 1e-4).sum()),
}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "validate_888_gpu.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
