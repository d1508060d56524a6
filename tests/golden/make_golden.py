#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own data files.

Run ONCE in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

It never imports reference code (the three scripts hard-import streamlit/librosa, which are absent);
it only READS the reference's data artefacts, which are the de-facto known-answer tests of the path
(SURVEY.md section 4):

  clear_audio/<stem>.wav  ->  cache_features/<stem>_clean_feats.npy     (888 pairs, pins extract_features)
  cache_features/*_{raw,clean}_feats.npy + segrigated_samples/<class>/   (pins the classify step)
  output_results/metrics_summary.csv, confusion_after_MLP.csv           (evidence for the re-fit)

Outputs (all under tests/golden/):
  validate_888.json        oracle vs all 888 pairs (max |diff| per block, per-clip worst cases)
  ref_pairs/NN.wav, NN.npy the shortest N real pairs; travels to the GPU box
  classifier_after.npz     StandardScaler + MLP(128,64) re-fit exactly as pipeline1.py:462-531 does,
                           with the 905x149 feature matrix and expected predict / predict_proba
  forest_after.npz         the RandomForest(200) of pipeline1.py:496 as flat node arrays + expected outputs
  clean_pin.json           peak / header statistics of all clear_audio/*.wav: what the last two steps of
                           clean_audio_and_cache (librosa.util.normalize + sf.write PCM_16, p1:140-141) leave behind
  tuning_888.json          (--tuning-888) the oracle's estimate_tuning index of every one of the 888 reference clips
  deployed_models.npz      (--deployed) the models main1.py:859-868 defines -- RandomForest(600), MLP(256,128,64),
                           SVC(C=10), soft VotingClassifier -- validated by re-running its 5-fold CV (m1:874-893) against
                           output_results/FINAL_PERFORMANCE_TABLE.csv, then fitted on all 905 scaled rows as m1:896 does
                           for the forest it serves (m1:984-989); scikit-learn's predict / predict_proba on the 905 rows
"""
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dysf_oracle as O  # noqa: E402

REF = os.environ.get("DYSF_REFERENCE", "/root/reference")
N_PAIRS = 12
CONFIG1_CLIP = "F_0050_10y9m_1 (mp3cut.net)10.wav"
# Clips on which a GPU build has missed the reference before (tests/golden/validate_888_gpu.json of round 1): the two whose
# tuning histogram arg-max flipped under float32 FFT noise, and two whose mfcc_mean_0 sits one float32 ulp from the 1e-4
# limit (the clip maximum of the log-mel decides the top_db floor of every clamped cell).
EDGE_CLIPS = ("M_0091_11y4m_1 (mp3cut.net) (2).wav", "M_1107_38y0m_1 (mp3cut.net)2.wav",
              "M_0098_07y8m_1 (mp3cut.net) (4).wav", "M_1105_21y0m_1  (mp3cut.net)35.wav")
AUDIO_EXTS = (".wav", ".mp3", ".flac", ".m4a", ".ogg")  # pipeline1.py:77


def reference_metrics():
    """stem -> (snr_after, flat_after, hf_after) from the reference's output_results/per_file_analysis.csv: the metrics of
    pipeline1.py:151-186 evaluated on the cleaned clip, i.e. on clear_audio/<stem>.wav."""
    out = {}
    with open(os.path.join(REF, "output_results", "per_file_analysis.csv")) as f:
        for r in csv.DictReader(f):
            out[r["file"].rsplit(".", 1)[0]] = (float(r["snr_after"]), float(r["flat_after"]), float(r["hf_after"]))
    return out


def validate_all():
    wavs = sorted(glob.glob(os.path.join(REF, "clear_audio", "*.wav")))
    ref_m = reference_metrics()
    m_err = np.zeros(3)
    m_n = 0
    blocks = {"mfcc_stats": (0, 40), "delta_stats": (40, 80), "delta2_stats": (80, 120),
              "chroma_stats": (120, 144), "text": (144, 149)}
    bmax = {k: 0.0 for k in blocks}
    rows, lengths = [], {}
    for w in wavs:
        stem = os.path.basename(w)[:-4]
        g = os.path.join(REF, "cache_features", stem + "_clean_feats.npy")
        if not os.path.exists(g):
            continue
        pcm, sr = O.read_wav_pcm16(w)
        feats = O.extract_features(O.pcm16_to_float(pcm), sr)
        gold = np.load(g)
        assert gold.dtype == np.float32 and gold.shape == (149,)
        d = np.abs(feats.astype(np.float64) - gold.astype(np.float64))
        for k, (a, b) in blocks.items():
            bmax[k] = max(bmax[k], float(d[a:b].max()))
        rows.append((stem, float(d.max()), int(d.argmax()), len(pcm)))
        lengths[w] = len(pcm)
        if stem in ref_m:
            got = O.audio_metrics(O.pcm16_to_float(pcm), sr)
            want = np.array(ref_m[stem])
            m_err = np.maximum(m_err, [abs(got[0] - want[0]), abs(got[1] - want[1]) / max(abs(want[1]), 1e-12),
                                       abs(got[2] - want[2]) / max(abs(want[2]), 1e-12)])
            m_n += 1
    diffs = np.array([r[1] for r in rows])
    rec = {
        "what": "oracle/dysf_oracle.py extract_features vs reference cache_features/*_clean_feats.npy",
        "n_pairs": len(rows),
        "max_abs_diff": float(diffs.max()),
        "p99_abs_diff": float(np.percentile(diffs, 99)),
        "median_abs_diff": float(np.median(diffs)),
        "n_over_1e-4": int((diffs > 1e-4).sum()),
        "block_max_abs_diff": bmax,
        "worst_clips": [dict(stem=r[0], max_abs_diff=r[1], index=r[2], n_samples=r[3])
                        for r in sorted(rows, key=lambda r: -r[1])[:5]],
        "metrics": {"what": "oracle audio_metrics vs per_file_analysis.csv (snr_after, flat_after, hf_after)", "n_clips": m_n,
                    "snr_db_max_abs_diff": float(m_err[0]), "flatness_max_rel_diff": float(m_err[1]),
                    "hf_ratio_max_rel_diff": float(m_err[2])},
        "numpy": np.__version__,
    }
    import scipy
    rec["scipy"] = scipy.__version__
    with open(os.path.join(HERE, "validate_888.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))
    return lengths


def copy_pairs(lengths):
    out = os.path.join(HERE, "ref_pairs")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    chosen = sorted(lengths, key=lambda w: (lengths[w], w))[:N_PAIRS]
    # plus the 3 s-class clip SURVEY.md section 8d names for BASELINE configs[0]
    chosen += [w for w in lengths if os.path.basename(w) == CONFIG1_CLIP]
    for name in EDGE_CLIPS:
        chosen += [w for w in lengths if os.path.basename(w) == name]
    index = []
    ref_m = reference_metrics()
    for i, w in enumerate(chosen):
        stem = os.path.basename(w)[:-4]
        shutil.copyfile(w, os.path.join(out, f"{i:02d}.wav"))
        shutil.copyfile(os.path.join(REF, "cache_features", stem + "_clean_feats.npy"),
                        os.path.join(out, f"{i:02d}.npy"))
        index.append({"id": f"{i:02d}", "reference_stem": stem, "n_samples": lengths[w],
                      # per_file_analysis.csv: snr_after, flat_after, hf_after of this clip (pipeline1.py:394-396)
                      "metrics_after": list(ref_m[stem]) if stem in ref_m else None})
    with open(os.path.join(out, "index.json"), "w") as f:
        json.dump(index, f, indent=1)


def export_all(lengths):
    """All 888 pairs into tests/golden/_all888/ (git-ignored, never committed): input of tests/golden/validate_888_gpu.py, which
    runs on the GPU box where /root/reference does not exist."""
    out = os.path.join(HERE, "_all888")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    index = []
    ref_m = reference_metrics()
    for i, w in enumerate(sorted(lengths)):
        stem = os.path.basename(w)[:-4]
        shutil.copyfile(w, os.path.join(out, f"{i:03d}.wav"))
        shutil.copyfile(os.path.join(REF, "cache_features", stem + "_clean_feats.npy"), os.path.join(out, f"{i:03d}.npy"))
        index.append({"id": f"{i:03d}", "stem": stem, "n_samples": lengths[w], "metrics_after": list(ref_m[stem]) if stem in ref_m else None})
    with open(os.path.join(out, "index.json"), "w") as f:
        json.dump(index, f)


def list_audio_files(root):  # same walk order as pipeline1.py:91-97
    files = []
    for r, _, fs in os.walk(root):
        for f in fs:
            if f.lower().endswith(AUDIO_EXTS):
                files.append(os.path.join(r, f))
    return sorted(files)


def classifier_fixture():
    from sklearn.metrics import accuracy_score, confusion_matrix, log_loss
    from sklearn.model_selection import train_test_split
    from sklearn.neural_network import MLPClassifier
    from sklearn.preprocessing import LabelEncoder, StandardScaler

    files = list_audio_files(os.path.join(REF, "segrigated_samples"))
    labels = [os.path.basename(os.path.dirname(f)) for f in files]  # pipeline1.py:358-362
    X_after = np.vstack([np.load(os.path.join(REF, "cache_features",
                                              os.path.basename(f).rsplit(".", 1)[0] + "_clean_feats.npy"))
                         for f in files])
    le = LabelEncoder()
    y = le.fit_transform(labels)
    scaler = StandardScaler().fit(X_after)
    Xs = scaler.transform(X_after)
    Xtr, Xte, ytr, yte = train_test_split(Xs, y, test_size=0.2, stratify=y, random_state=42)
    idx_tr, idx_te = train_test_split(np.arange(len(y)), test_size=0.2, stratify=y, random_state=42)
    mlp = MLPClassifier(hidden_layer_sizes=(128, 64), max_iter=400, random_state=42)
    mlp.fit(Xtr, ytr)
    pred_te = mlp.predict(Xte)
    acc = float(accuracy_score(yte, pred_te) * 100.0)
    loss = float(log_loss(yte, mlp.predict_proba(Xte)))
    cm = confusion_matrix(yte, pred_te)
    with open(os.path.join(REF, "output_results", "confusion_after_MLP.csv")) as f:
        ref_cm = np.array([[int(v) for v in row[1:]] for row in list(csv.reader(f))[1:]])
    evidence = {"accuracy": acc, "log_loss": loss, "confusion": cm.tolist(),
                "reference_confusion_after_MLP": ref_cm.tolist(),
                "confusion_matches_reference": bool((cm == ref_cm).all()),
                "reference_metrics_summary_after_MLP": [69.06077348066299, 1.8513111489493788]}
    print(json.dumps(evidence, indent=1))
    np.savez_compressed(
        os.path.join(HERE, "classifier_after.npz"),
        X=X_after.astype(np.float32), y=y.astype(np.int64), classes=np.array(le.classes_),
        mean=scaler.mean_, scale=scaler.scale_,
        W1=mlp.coefs_[0], b1=mlp.intercepts_[0], W2=mlp.coefs_[1], b2=mlp.intercepts_[1],
        W3=mlp.coefs_[2], b3=mlp.intercepts_[2],
        test_index=idx_te.astype(np.int64),
        expected_pred=mlp.predict(Xs).astype(np.int64), expected_proba=mlp.predict_proba(Xs),
        evidence=json.dumps(evidence))


def forest_fixture():
    """RandomForestClassifier(n_estimators=200, random_state=42) re-fit exactly as pipeline1.py:496, 512-514 does on the
    cleaned ("after") features; exported as flat node arrays for the GPU traversal kernel (the
    deployed predictor of main1.py:984-989 is the forest)."""
    from sklearn.ensemble import RandomForestClassifier
    from sklearn.metrics import accuracy_score, confusion_matrix, log_loss
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import LabelEncoder, StandardScaler

    files = list_audio_files(os.path.join(REF, "segrigated_samples"))
    labels = [os.path.basename(os.path.dirname(f)) for f in files]
    X_after = np.vstack([np.load(os.path.join(REF, "cache_features",
                                              os.path.basename(f).rsplit(".", 1)[0] + "_clean_feats.npy"))
                         for f in files])
    y = LabelEncoder().fit_transform(labels)
    scaler = StandardScaler().fit(X_after)
    Xs = scaler.transform(X_after)
    Xtr, Xte, ytr, yte = train_test_split(Xs, y, test_size=0.2, stratify=y, random_state=42)
    rf = RandomForestClassifier(n_estimators=200, random_state=42).fit(Xtr, ytr)
    pred_te, proba_te = rf.predict(Xte), rf.predict_proba(Xte)
    acc, loss = float(accuracy_score(yte, pred_te) * 100.0), float(log_loss(yte, proba_te))
    cm = confusion_matrix(yte, pred_te)
    with open(os.path.join(REF, "output_results", "confusion_after_RandomForest.csv")) as f:
        ref_cm = np.array([[int(v) for v in row[1:]] for row in list(csv.reader(f))[1:]])
    evidence = {"accuracy": acc, "log_loss": loss, "confusion_matches_reference": bool((cm == ref_cm).all()),
                "reference_metrics_summary_after_RandomForest": [65.74585635359117, 0.8522499830310881]}
    print(json.dumps(evidence, indent=1))
    feat, thr, left, right, value, root = [], [], [], [], [], []
    for est in rf.estimators_:
        t = est.tree_
        root.append(len(feat))
        base = len(feat)
        feat += t.feature.tolist()
        thr += t.threshold.tolist()
        left += [c + base if c >= 0 else -1 for c in t.children_left]
        right += [c + base if c >= 0 else -1 for c in t.children_right]
        value += t.value[:, 0, :].tolist()  # class fractions as stored (sklearn >= 1.4): what tree.predict_proba returns
    np.savez_compressed(
        os.path.join(HERE, "forest_after.npz"),
        feature=np.array(feat, np.int32), threshold=np.array(thr, np.float64), left=np.array(left, np.int32),
        right=np.array(right, np.int32), value=np.array(value, np.float64), root=np.array(root, np.int32),
        mean=scaler.mean_, scale=scaler.scale_,
        expected_pred=rf.predict(Xs).astype(np.int64), expected_proba=rf.predict_proba(Xs),
        evidence=json.dumps(evidence))


def svc_fixture():
    """SVC(probability=True, kernel='rbf', C=1.0, random_state=42) re-fit exactly as pipeline1.py:498, 512-514 does on the
    cleaned ("after") features; exported as the libsvm model arrays (support vectors, dual coefficients, intercepts,
    Platt A/B) for the GPU kernel, with sklearn's own decision values / predictions / probabilities as expected outputs."""
    from sklearn.metrics import accuracy_score, confusion_matrix, log_loss
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import LabelEncoder, StandardScaler
    from sklearn.svm import SVC

    files = list_audio_files(os.path.join(REF, "segrigated_samples"))
    labels = [os.path.basename(os.path.dirname(f)) for f in files]
    X_after = np.vstack([np.load(os.path.join(REF, "cache_features",
                                              os.path.basename(f).rsplit(".", 1)[0] + "_clean_feats.npy"))
                         for f in files])
    y = LabelEncoder().fit_transform(labels)
    scaler = StandardScaler().fit(X_after)
    Xs = scaler.transform(X_after)
    Xtr, Xte, ytr, yte = train_test_split(Xs, y, test_size=0.2, stratify=y, random_state=42)
    svc = SVC(probability=True, kernel="rbf", C=1.0, random_state=42).fit(Xtr, ytr)
    pred_te, proba_te = svc.predict(Xte), svc.predict_proba(Xte)
    acc, loss = float(accuracy_score(yte, pred_te) * 100.0), float(log_loss(yte, proba_te))
    cm = confusion_matrix(yte, pred_te)
    with open(os.path.join(REF, "output_results", "confusion_after_SVM.csv")) as f:
        ref_cm = np.array([[int(v) for v in row[1:]] for row in list(csv.reader(f))[1:]])
    ref_row = None
    with open(os.path.join(REF, "output_results", "metrics_summary.csv")) as f:
        for row in csv.DictReader(f):
            if row["dataset"] == "after" and row["model"] == "SVM":
                ref_row = [float(row["accuracy"]), float(row["test_loss"])]
    evidence = {"accuracy": acc, "log_loss": loss, "confusion_matches_reference": bool((cm == ref_cm).all()),
                "reference_metrics_summary_after_SVM": ref_row}
    print(json.dumps(evidence, indent=1))
    svc.decision_function_shape = "ovo"
    np.savez_compressed(
        os.path.join(HERE, "svc_after.npz"),
        support_vectors=svc.support_vectors_.astype(np.float64), dual_coef=svc.dual_coef_.astype(np.float64),
        intercept=svc.intercept_.astype(np.float64), prob_a=svc.probA_.astype(np.float64),
        prob_b=svc.probB_.astype(np.float64), n_support=svc.n_support_.astype(np.int32),
        gamma=np.float64(svc._gamma), mean=scaler.mean_, scale=scaler.scale_,
        expected_pred=svc.predict(Xs).astype(np.int64), expected_proba=svc.predict_proba(Xs),
        expected_decision=svc.decision_function(Xs), evidence=json.dumps(evidence))


def clean_pin():
    """Every WAV in clear_audio/ went through librosa.util.normalize (peak -> 1.0) and python-soundfile's PCM_16 write:
    record the peaks and the header layout that pin oracle/dysf_clean_oracle.py's normalize + float_to_pcm16."""
    rec = {"files": 0, "peak_pos_32767": 0, "peak_neg_32768": 0, "other_peaks": [], "header_44_bytes_pcm16_mono_16k": 0}
    for f in sorted(glob.glob(os.path.join(REF, "clear_audio", "*.wav"))):
        b = open(f, "rb").read()
        d = np.frombuffer(b[44:], dtype="<i2")
        rec["files"] += 1
        hdr_ok = (b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[16:36] == np.array([16], "<u4").tobytes() +
                  np.array([1, 1], "<u2").tobytes() + np.array([16000, 32000], "<u4").tobytes() + np.array([2, 16], "<u2").tobytes()
                  and b[36:40] == b"data" and int(np.frombuffer(b[40:44], "<u4")[0]) == len(b) - 44
                  and int(np.frombuffer(b[4:8], "<u4")[0]) == len(b) - 8)
        rec["header_44_bytes_pcm16_mono_16k"] += int(hdr_ok)
        mx, mn = int(d.max()), int(d.min())
        if mx == 32767 and mn > -32768:
            rec["peak_pos_32767"] += 1
        elif mn == -32768 and mx < 32767:
            rec["peak_neg_32768"] += 1
        else:
            rec["other_peaks"].append([os.path.basename(f), mx, mn])
    with open(os.path.join(HERE, "clean_pin.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print("clean_pin:", {k: (v if not isinstance(v, list) else len(v)) for k, v in rec.items()})


def tuning_888():
    """The oracle's tuning index (librosa.estimate_tuning inside chroma_stft, pipeline1.py:227) of all 888 reference
    clips, by stem: the histogram arg-max is a tie or near-tie in most clips, so the GPU test compares indices exactly."""
    from multiprocessing import Pool
    stems = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(REF, "clear_audio", "*.wav"))
                   if os.path.exists(os.path.join(REF, "cache_features", os.path.basename(f)[:-4] + "_clean_feats.npy")))
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        idx = pool.map(_tuning_of, stems, chunksize=8)
    with open(os.path.join(HERE, "tuning_888.json"), "w") as fh:
        json.dump({"what": "oracle/dysf_oracle.py estimate_tuning index (0..99) of the reference clips clear_audio/<stem>.wav, "
                           "by stem; tests/golden/make_golden.py --tuning-888", "tuning_index": dict(zip(stems, idx))}, fh, indent=0)
    print("tuning_888:", len(stems), "clips")


def _tuning_of(stem):
    pcm, sr = O.read_wav_pcm16(os.path.join(REF, "clear_audio", stem + ".wav"))
    return int(O.estimate_tuning(O.stft_power(O.pcm16_to_float(pcm), O.REFERENCE_149), O.REFERENCE_149)[1]["index"])


def _flat_forest(rf):
    feat, thr, left, right, value, root = [], [], [], [], [], []
    for est in rf.estimators_:
        t = est.tree_
        root.append(len(feat))
        base = len(feat)
        feat += t.feature.tolist()
        thr += t.threshold.tolist()
        left += [c + base if c >= 0 else -1 for c in t.children_left]
        right += [c + base if c >= 0 else -1 for c in t.children_right]
        value += t.value[:, 0, :].tolist()
    return {"feature": np.array(feat, np.int32), "threshold": np.array(thr, np.float64), "left": np.array(left, np.int32),
            "right": np.array(right, np.int32), "value": np.array(value, np.float64), "root": np.array(root, np.int32)}


def deployed_fixture():
    """The classifiers main1.py defines (m1:859-868), cross-validated exactly as m1:856, 874-893 does, compared with the
    reference's own FINAL_PERFORMANCE_TABLE.csv, then fitted on ALL 905 scaled rows -- what m1:896-897 does for the forest
    the sidebar serves (m1:984-989; the stored model_rf.pkl is a missing blob).  The soft-voting ensemble of m1:867 clones
    and re-fits the three base models, so its probabilities are the mean of theirs (checked below)."""
    from sklearn.ensemble import RandomForestClassifier, VotingClassifier
    from sklearn.metrics import accuracy_score, precision_recall_fscore_support
    from sklearn.model_selection import StratifiedKFold
    from sklearn.neural_network import MLPClassifier
    from sklearn.preprocessing import LabelEncoder, StandardScaler
    from sklearn.svm import SVC

    files = list_audio_files(os.path.join(REF, "segrigated_samples"))
    labels = [os.path.basename(os.path.dirname(f)) for f in files]
    X_after = np.vstack([np.load(os.path.join(REF, "cache_features",
                                              os.path.basename(f).rsplit(".", 1)[0] + "_clean_feats.npy")) for f in files])
    le = LabelEncoder()
    y = le.fit_transform(labels)
    scaler = StandardScaler()
    Xs = scaler.fit_transform(X_after)
    skf = StratifiedKFold(n_splits=5, shuffle=True, random_state=42)

    def make():
        base = {
            "RandomForest": RandomForestClassifier(n_estimators=600, max_depth=None, min_samples_split=2, min_samples_leaf=1,
                                                   random_state=42, n_jobs=-1),
            "MLP": MLPClassifier(hidden_layer_sizes=(256, 128, 64), max_iter=1200, alpha=1e-4, learning_rate="adaptive",
                                 random_state=42),
            "SVM": SVC(probability=True, C=10, gamma="scale", random_state=42),
        }
        ens = VotingClassifier(estimators=[(k, v) for k, v in base.items()], voting="soft")
        return {**base, "Ensemble": ens}

    ref_table = {}
    with open(os.path.join(REF, "output_results", "FINAL_PERFORMANCE_TABLE.csv")) as f:
        for r in csv.DictReader(f):
            ref_table[r["Model"]] = [float(r["Accuracy (%)"]), float(r["Precision (%)"]), float(r["Recall (%)"]), float(r["F1-Score (%)"])]
    models = make()
    cv = {}
    for name, model in models.items():
        accs, precs, recs, f1s = [], [], [], []
        for tr, te in skf.split(Xs, y):
            model.fit(Xs[tr], y[tr])
            pred = model.predict(Xs[te])
            pr, rc, f1, _ = precision_recall_fscore_support(y[te], pred, average="macro", zero_division=0)
            accs.append(accuracy_score(y[te], pred)); precs.append(pr); recs.append(rc); f1s.append(f1)
        cv[name] = [float(np.mean(v) * 100) for v in (accs, precs, recs, f1s)]
        print(name, "cv", cv[name], "reference", ref_table.get(name))
    evidence = {"cv_5fold": cv, "reference_FINAL_PERFORMANCE_TABLE": ref_table,
                "max_abs_diff_vs_reference": {k: float(np.max(np.abs(np.array(cv[k]) - np.array(ref_table[k])))) for k in cv},
                "sklearn": __import__("sklearn").__version__}
    # the deployed fits: all 905 rows (m1:896)
    rf, mlp, svc, ens = models["RandomForest"], models["MLP"], models["SVM"], models["Ensemble"]
    rf.fit(Xs, y); mlp.fit(Xs, y); svc.fit(Xs, y); ens.fit(Xs, y)
    mean_proba = (rf.predict_proba(Xs) + mlp.predict_proba(Xs) + svc.predict_proba(Xs)) / 3.0
    evidence["ensemble_is_mean_of_the_three"] = float(np.abs(ens.predict_proba(Xs) - mean_proba).max())
    print(json.dumps(evidence, indent=1))
    svc.decision_function_shape = "ovo"
    out = {"X": X_after.astype(np.float32), "y": y.astype(np.int64), "classes": np.array(le.classes_),
           "mean": scaler.mean_, "scale": scaler.scale_, "evidence": json.dumps(evidence)}
    out.update({"rf_" + k: v for k, v in _flat_forest(rf).items()})
    out.update({"rf_pred": rf.predict(Xs).astype(np.int64), "rf_proba": rf.predict_proba(Xs)})
    for i, (W, b) in enumerate(zip(mlp.coefs_, mlp.intercepts_)):
        out["mlp_W%d" % (i + 1)] = W; out["mlp_b%d" % (i + 1)] = b
    out.update({"mlp_pred": mlp.predict(Xs).astype(np.int64), "mlp_proba": mlp.predict_proba(Xs)})
    out.update({"svc_support_vectors": svc.support_vectors_.astype(np.float64), "svc_dual_coef": svc.dual_coef_.astype(np.float64),
                "svc_intercept": svc.intercept_.astype(np.float64), "svc_prob_a": svc.probA_.astype(np.float64),
                "svc_prob_b": svc.probB_.astype(np.float64), "svc_n_support": svc.n_support_.astype(np.int32),
                "svc_gamma": np.float64(svc._gamma), "svc_pred": svc.predict(Xs).astype(np.int64),
                "svc_proba": svc.predict_proba(Xs), "svc_decision": svc.decision_function(Xs)})
    out.update({"ens_pred": ens.predict(Xs).astype(np.int64), "ens_proba": ens.predict_proba(Xs)})
    np.savez_compressed(os.path.join(HERE, "deployed_models.npz"), **out)


if __name__ == "__main__":
    if "--tuning-888" in sys.argv:
        tuning_888()
        sys.exit(0)
    if "--deployed" in sys.argv:
        deployed_fixture()
        sys.exit(0)
    if "--clean-pin-only" in sys.argv:
        clean_pin()
        sys.exit(0)
    if "--forest-only" in sys.argv:
        forest_fixture()
        sys.exit(0)
    if "--svc-only" in sys.argv:
        svc_fixture()
        sys.exit(0)
    lengths = validate_all()
    copy_pairs(lengths)
    if "--all888" in sys.argv:
        export_all(lengths)
    if "--pairs-only" in sys.argv:
        sys.exit(0)
    classifier_fixture()
    forest_fixture()
    svc_fixture()
    clean_pin()
