"""The reference's two per-file loops as batched GPU passes (pipeline1.py:371-417 "Preprocess each file" and
pipeline1.py:447-456 "Feature extraction"), with the reference's memoisation contracts kept:

    for f in files:                                   # p1:371-417
        y_raw = load_audio(f)                         #   -> one decode per file on the host
        snr / flatness / hf ratio of y_raw            #   -> one dysf_metrics_batch_dev per batch
        cleaned = clean_audio_and_cache(f)            #   -> one dysf_clean_batch_dev per batch; clear_dir/<stem>.wav written,
                                                      #      existing files reused (p1:134-135), failures fall back to f (p1:386-388)
        y_clean = load_audio(cleaned)
        snr / flatness / hf ratio of y_clean          #   -> one dysf_metrics_batch_dev per batch
    for raw_p, clean_p in ...:                        # p1:447-456
        cached_extract_features(raw_p, "", "raw")     #   -> one extract_batch per batch, cache_dir/<stem>_{raw,clean}_feats.npy
        cached_extract_features(clean_p, "", "clean") #      written / reused exactly as p1:429-440 does

Inputs must be decodable by ``load_audio`` (mono PCM_16 WAV; MP3 decode is upstream of this path).  Row keys are those of
the dicts pipeline1.py:401-412 appends.  No CPU implementation of any numeric step here."""
import csv
import logging
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .cleaning import CLEAR_DIR, get_cleaner, write_wav_pcm16
from .features import CACHE_DIR, cached_path, get_extractor, load_audio, load_cache, save_cache
from .metrics import audio_metrics_batch
from .params import REFERENCE_149, TARGET_SR

AUDIO_EXTS = (".wav", ".mp3", ".flac", ".m4a", ".ogg")     # pipeline1.py:77
PER_FILE_COLUMNS = ("file", "label", "duration_sec", "snr_before_db", "snr_after_db", "spectral_flatness_before",
                    "spectral_flatness_after", "hf_energy_ratio_before", "hf_energy_ratio_after", "transcript")


def list_audio_files(root: str) -> List[str]:
    """pipeline1.py:91-97: every audio file under ``root``, sorted."""
    files = []
    for r, _, fs in os.walk(root):
        for f in fs:
            if f.lower().endswith(AUDIO_EXTS):
                files.append(os.path.join(r, f))
    return sorted(files)


def _stem(path: str) -> str:
    return os.path.basename(path).rsplit(".", 1)[0]


def preprocess_files(files: Sequence[str], clear_dir: str = CLEAR_DIR, batch: int = 256, prop_decrease: float = 1.0,
                     device: int = 0) -> Dict[str, object]:
    """pipeline1.py:371-417.  Returns {"per_file": [row dicts], "raw_paths", "clean_paths", "labels", "transcripts", "skipped"}."""
    out = {"per_file": [], "raw_paths": [], "clean_paths": [], "labels": [], "transcripts": [], "skipped": 0}
    for b0 in range(0, len(files), max(1, batch)):
        group = []
        for f in files[b0:b0 + max(1, batch)]:
            y_raw, sr_raw = load_audio(f, sr=TARGET_SR)
            if y_raw is None:
                out["skipped"] += 1
                continue
            group.append((f, y_raw, sr_raw))
        if not group:
            continue
        before = audio_metrics_batch([y for _, y, _ in group], TARGET_SR, device)
        # cleaning, memoised per basename: only the files without a clear_dir/<stem>.wav go to the GPU
        cleaned_paths = [cached_path(os.path.join(clear_dir, f"{_stem(f)}.wav")) for f, _, _ in group]
        todo = [i for i, p in enumerate(cleaned_paths) if not os.path.exists(p)]
        # two inputs with the same basename share one output file (the reference's second call finds the first one's file)
        first = {}
        todo = [i for i in todo if first.setdefault(cleaned_paths[i], i) == i]
        if todo:
            os.makedirs(clear_dir, exist_ok=True)
            cleaner = get_cleaner(TARGET_SR, device, prop_decrease)
            try:   # one launch for the whole batch; clips beyond noisereduce's chunk_size are gated chunk by chunk inside
                res = cleaner.clean_batch([group[i][1] for i in todo], pcm16=True)
                done = [(i, int(res["status"][k]), res["pcm16"][k]) for k, i in enumerate(todo)]
            except Exception as e:   # e.g. the batch's workspace does not fit: file by file, as the reference works (p1:384)
                logging.error(f"batched cleaning failed, cleaning file by file: {e}")
                done = []
                for i in todo:
                    try:
                        r = cleaner.clean_batch([group[i][1]], pcm16=True)
                        done.append((i, int(r["status"][0]), r["pcm16"][0]))
                    except Exception as e1:   # p1:144-146: "Cleaning failed" -> the original path is used (p1:386-388)
                        logging.error(f"Cleaning failed for {group[i][0]}: {e1}")
            for i, status, pcm in done:
                if status == 0:
                    write_wav_pcm16(cleaned_paths[i], pcm, TARGET_SR)
        clean_audio = []
        for i, (f, y_raw, _) in enumerate(group):
            path = cleaned_paths[i] if os.path.exists(cleaned_paths[i]) else f        # p1:386-388: fall back to the original
            y_clean, sr_clean = load_audio(path, sr=TARGET_SR)
            clean_audio.append((path, y_clean, sr_clean))
        ok = [i for i, (_, y, _) in enumerate(clean_audio) if y is not None]
        after = audio_metrics_batch([clean_audio[i][1] for i in ok], TARGET_SR, device) if ok else np.zeros((0, 3), np.float32)
        after_of = {i: after[k] for k, i in enumerate(ok)}
        for i, (f, y_raw, sr_raw) in enumerate(group):
            if i not in after_of:
                out["skipped"] += 1
                continue
            label = os.path.basename(os.path.dirname(f)) or "unknown"
            out["per_file"].append({
                "file": os.path.basename(f), "label": label,
                "duration_sec": float(len(y_raw) / sr_raw) if sr_raw else 0.0,
                "snr_before_db": float(before[i][0]), "snr_after_db": float(after_of[i][0]),
                "spectral_flatness_before": float(before[i][1]), "spectral_flatness_after": float(after_of[i][1]),
                "hf_energy_ratio_before": float(before[i][2]), "hf_energy_ratio_after": float(after_of[i][2]),
                "transcript": "",
            })
            out["raw_paths"].append(f)
            out["clean_paths"].append(clean_audio[i][0])
            out["labels"].append(label)
            out["transcripts"].append("")
    return out


def write_per_file_analysis(per_file: Sequence[dict], csv_path: str):
    """pipeline1.py:415-417 (``analysis_df.to_csv(analysis_csv, index=False)``)."""
    with open(csv_path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(PER_FILE_COLUMNS))
        w.writeheader()
        for row in per_file:
            w.writerow(row)


def cached_extract_features_batch(paths: Sequence[str], suffix: str, cache_dir: str = CACHE_DIR, batch: int = 1024,
                                  params=REFERENCE_149, device: int = 0) -> np.ndarray:
    """pipeline1.py:429-456 for a list of files: rows come from ``<cache_dir>/<stem>_<suffix>_feats.npy`` when that file
    exists, the others are extracted in batches of ``batch`` clips and cached.  [n, 149] float32, the order of ``paths``."""
    rows: List[Optional[np.ndarray]] = [None] * len(paths)
    cache_files = [cached_path(os.path.join(cache_dir, f"{_stem(p)}_{suffix}_feats.npy")) for p in paths]
    missing = []
    for i, cf in enumerate(cache_files):
        cached = load_cache(cf)
        if cached is not None:
            rows[i] = np.array(cached)
        else:
            missing.append(i)
    ex = get_extractor(params, device)
    n_text = 5                                                      # pipeline1.py:243-244: five zeros for the empty transcript
    width = ex.pooled_dim + n_text
    for b0 in range(0, len(missing), max(1, batch)):
        idx = missing[b0:b0 + max(1, batch)]
        audio = [load_audio(paths[i], sr=params.sr)[0] for i in idx]
        good = [k for k, y in enumerate(audio) if y is not None and len(y) > 0]
        pooled = ex.extract_batch([audio[k] for k in good])["pooled"] if good else np.zeros((0, ex.pooled_dim), np.float32)
        row_of = {k: r for r, k in enumerate(good)}
        for k, i in enumerate(idx):
            feats = np.zeros(width, np.float32)                    # p1:213-214: a clip that failed to load gives zeros
            if k in row_of:
                feats[:ex.pooled_dim] = pooled[row_of[k]]
            os.makedirs(cache_dir, exist_ok=True)
            save_cache(feats, cache_files[i])
            rows[i] = feats
    return np.vstack(rows).astype(np.float32) if rows else np.empty((0, width), np.float32)
