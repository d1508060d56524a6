"""CPU tier: host logic that mirrors the reference's Python (cache format, text features, audio loader, sharding,
logits all-gather over gloo with world_size 2, MLP head arg-max against the scikit-learn fixture)."""
import os
import socket

import numpy as np
import pytest
import torch

import rsd_amd
from oracle import dysf_oracle as O
from rsd_amd.classifier import CNNHead, MLPHead
from rsd_amd.pipeline import gather_rows, shard_by_samples, shard_range
from rsd_amd.synth import synth_bank, synth_clip


def test_cache_file_is_byte_compatible_with_reference(tmp_path, golden_dir):
    gold_path = os.path.join(golden_dir, "ref_pairs", "00.npy")
    gold = np.load(gold_path)
    p = tmp_path / "x_clean_feats.npy"
    rsd_amd.save_cache(gold.copy(), str(p))
    assert p.read_bytes() == open(gold_path, "rb").read()  # NPY v1.0, '<f4', (149,), 724 bytes
    assert len(p.read_bytes()) == 724
    assert np.array_equal(rsd_amd.load_cache(str(p)), gold)
    assert rsd_amd.load_cache(str(tmp_path / "missing.npy")) is None


def test_cached_extract_features_hit_short_circuits(tmp_path, golden_dir):
    gold = np.load(os.path.join(golden_dir, "ref_pairs", "03.npy"))
    np.save(tmp_path / "some clip (1)_clean_feats.npy", gold)
    got = rsd_amd.cached_extract_features("/data/whatever/some clip (1).wav", "", "clean", cache_dir=str(tmp_path))
    assert np.array_equal(got, gold)  # keyed by basename only, like pipeline1.py:431-432


def test_load_audio_matches_soundfile_scaling_and_failure_convention(golden_dir, tmp_path):
    wav = os.path.join(golden_dir, "ref_pairs", "01.wav")
    y, sr = rsd_amd.load_audio(wav)
    pcm, osr = O.read_wav_pcm16(wav)
    assert sr == osr == 16000 and y.dtype == np.float32 and np.array_equal(y, O.pcm16_to_float(pcm))
    assert rsd_amd.load_audio(str(tmp_path / "nope.wav")) == (None, None)
    # a file at another rate goes through the GPU resampler (test_resampler_matches_scipy_resample_poly, -m gpu);
    # without a GPU that raises inside load_audio and becomes the reference's (None, None)
    import torch
    if not torch.cuda.is_available():
        assert rsd_amd.load_audio(wav, sr=22050) == (None, None)
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    assert rsd_amd.load_audio(str(tmp_path / "junk.wav")) == (None, None)


def test_none_input_gives_zero_vector_without_touching_the_gpu():
    f = rsd_amd.extract_features(None, None)
    assert f.shape == (149,) and f.dtype == np.float32 and (f == 0).all()
    assert rsd_amd.extract_audio_features(None, 16000).shape == (144,)


def test_text_features_follow_reference_rules():
    assert (rsd_amd.extract_text_features("") == 0).all()
    t = "I I want want want to go"
    f = rsd_amd.extract_text_features(t)
    assert f.dtype == np.float32 and f[0] == len(t) and f[1] == 7 and f[2] == 3
    assert abs(f[3] - 3 / 7) < 1e-6 and abs(f[4] - 4 / 7) < 1e-6
    assert rsd_amd.repetition_stats_from_text("...")["repetition_count"] == 0


def test_synthetic_bank_is_seed_stable():
    a, b = synth_clip(5), synth_clip(5)
    assert a.dtype == np.int16 and a.shape == (48000,) and np.array_equal(a, b)
    assert np.abs(synth_clip(63)).max() == 0 and np.abs(a).max() == round(0.9 * 32767)
    bank = synth_bank(6, n=4000, workers=1)
    assert bank.shape == (6, 4000) and np.array_equal(bank[3], synth_clip(3, n=4000))


def test_shard_ranges_cover_the_index_space():
    for n, w in ((100000, 8), (1024, 3), (5, 8), (0, 4)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_range(100000, 3, 8) == (37500, 50000)
    lens = np.array([10] * 8 + [40] * 2)
    b = shard_by_samples(lens, 2)
    assert b.tolist() == [0, 8, 10] and lens[:8].sum() == lens[8:].sum()
    rng = np.random.default_rng(0)
    lens = rng.integers(16000, 160000, size=1000)
    b = shard_by_samples(lens, 8)
    per = np.array([lens[b[i]:b[i + 1]].sum() for i in range(8)])
    assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) > 0).all() and np.abs(per - lens.sum() / 8).max() <= lens.max()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gather_worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 10.0, 100.0])
    out = gather_rows(local, n_total, rank, world)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [11, 8])
def test_logits_all_gather_world_size_2_gloo(n_total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    want = np.arange(n_total, dtype=np.float32)[:, None] * np.array([1.0, 10.0, 100.0], np.float32)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
    one = gather_rows(torch.ones(3, 2), 3, 0, 1)  # world size 1 degenerates to a copy
    assert one.shape == (3, 2)


def _gather_bounds_worker(rank, world, port, lens, q):
    import torch.distributed as dist
    from rsd_amd.pipeline import gather_rows_by_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = shard_by_samples(np.array(lens), world)
    local = torch.arange(int(b[rank]), int(b[rank + 1]), dtype=torch.float32)[:, None] * torch.tensor([1.0, 10.0, 100.0])
    out = gather_rows_by_bounds(local, b, rank, world)
    q.put((rank, out.numpy(), b.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_unequal_shards_all_gather_world_size_2_gloo():
    """configs[4] at N > 1: shards balanced by cumulative sample count have different clip counts; one all-gather of the
    padded shards puts every clip's logits back at its index of the global list on every rank."""
    import torch.multiprocessing as mp
    from rsd_amd.pipeline import gather_rows_by_bounds
    lens = [9, 1, 1, 1, 1, 1, 1, 1, 8, 8, 2]   # 34 samples: the first shard holds fewer clips than the second
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_gather_bounds_worker, args=(r, 2, port, lens, q)) for r in range(2)]
    [p.start() for p in procs]
    got = {r: (o, b) for r, o, b in (q.get(timeout=120) for _ in range(2))}
    [p.join(timeout=60) for p in procs]
    want = np.arange(len(lens), dtype=np.float32)[:, None] * np.array([1.0, 10.0, 100.0], np.float32)
    b = got[0][1]
    assert b[0] == 0 and b[-1] == len(lens) and b[1] - b[0] != b[2] - b[1]
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)
    assert gather_rows_by_bounds(torch.ones(3, 2), [0, 3], 0, 1).shape == (3, 2)   # world size 1: a copy


def test_mlp_head_reproduces_sklearn_fixture_on_cpu(golden_dir):
    path = os.path.join(golden_dir, "classifier_after.npz")
    z = np.load(path)
    head = MLPHead.from_npz(path)
    x = torch.from_numpy(z["X"])
    assert np.array_equal(head.predict(x).numpy(), z["expected_pred"])
    assert np.abs(head.predict_proba(x).numpy() - z["expected_proba"]).max() <= 1e-4
    te = z["test_index"]
    acc = (z["expected_pred"][te] == z["y"][te]).mean() * 100
    assert abs(acc - 69.06077348066299) < 1e-9  # output_results/metrics_summary.csv:6 (after, MLP)
    assert head.inverse_transform([0, 2]) == ["Prolongatio sample", "word repetition"]
    with pytest.raises(ValueError):
        head(torch.zeros(1, 148))
    # the decision margins are wide enough for float32: no row sits within 1e-3 of a tie
    top2 = torch.topk(head(x), 2, dim=1).values
    assert float((top2[:, 0] - top2[:, 1]).min()) > 1e-3


def test_cnn_head_shape_and_seed():
    a, b = CNNHead(), CNNHead()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    assert a(torch.zeros(2, 120, 301)).shape == (2, 3)


def test_forest_fixture_reproduces_reference_metrics(golden_dir):
    import json
    z = np.load(os.path.join(golden_dir, "forest_after.npz"))
    ev = json.loads(str(z["evidence"]))
    assert ev["confusion_matches_reference"] is True
    assert abs(ev["accuracy"] - 65.74585635359117) < 1e-9 and abs(ev["log_loss"] - 0.8522499830310881) < 1e-9
    assert len(z["root"]) == 200 and z["value"].shape[1] == 3 and np.allclose(z["value"].sum(axis=1), 1.0)
    leaf = z["feature"] < 0
    assert (z["left"][~leaf] > np.nonzero(~leaf)[0]).all()  # children follow parents: every walk terminates
    # a pure-NumPy walk of the flat arrays gives the stored expectations (pins the export format on CPU)
    X = np.load(os.path.join(golden_dir, "classifier_after.npz"))["X"]
    xs = (((X.astype(np.float64) - z["mean"]).astype(np.float32)).astype(np.float64) / z["scale"]).astype(np.float32)
    rows = [17, 904]
    for r in rows:
        acc = np.zeros(3)
        for root in z["root"]:
            node = int(root)
            while z["feature"][node] >= 0:
                node = int(z["left"][node] if xs[r, z["feature"][node]] <= z["threshold"][node] else z["right"][node])
            acc += z["value"][node]
        assert np.abs(acc / 200 - z["expected_proba"][r]).max() < 1e-12


def test_resampler_filter_design_equals_scipy():
    """rsd_amd.resample.design_taps (NumPy only) against the filter scipy.signal.resample_poly builds for 22.05 -> 16 kHz
    and 44.1 -> 16 kHz (firwin + Kaiser(5.0), times up, front padding, n_pre_remove)."""
    from scipy.signal import firwin
    import rsd_amd.resample as R
    for up, down in ((320, 441), (160, 441), (2, 1), (1, 3)):
        taps, n_pre_remove = R.design_taps(up, down)
        half_len = 10 * max(up, down)
        h = firwin(2 * half_len + 1, 1.0 / max(up, down), window=("kaiser", 5.0)) * up
        n_pre_pad = down - half_len % down
        assert len(taps) == n_pre_pad + len(h) and not taps[:n_pre_pad].any()
        assert np.abs(taps[n_pre_pad:] - h).max() <= 1e-13 * up
        assert n_pre_remove == (half_len + n_pre_pad) // down
    assert R.output_length(66150, 320, 441) == 48000 and R.output_length(1, 320, 441) == 1 and R.output_length(0, 320, 441) == 0


def test_sharding_properties_hypothesis():
    """shard_range / shard_by_samples partition the clip index space: contiguous, disjoint, complete, balanced."""
    from hypothesis import given, settings, strategies as st
    from rsd_amd.pipeline import shard_by_samples, shard_range

    @settings(max_examples=200, deadline=None)
    @given(n=st.integers(0, 5000), world=st.integers(1, 16))
    def check_range(n, world):
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert all(s >= 0 for s in sizes) and max(sizes) <= -(-n // world)

    @settings(max_examples=100, deadline=None)
    @given(lens=st.lists(st.integers(1, 100000), min_size=1, max_size=200), world=st.integers(1, 8))
    def check_samples(lens, world):
        b = shard_by_samples(np.array(lens), world)
        assert b[0] == 0 and b[-1] == len(lens) and (np.diff(b) >= 0).all()
        per = [int(np.sum(lens[b[i]:b[i + 1]])) for i in range(world)]
        assert sum(per) == sum(lens) and max(per) <= sum(lens) / world + max(lens)

    check_range()
    check_samples()


def test_bench_self_launch_propagates_a_failed_rank(tmp_path):
    """`python bench.py --gpus 2` starts its own rank processes before anything touches the GPU and returns their exit
    code: on this GPU-less box both ranks die in torch.cuda.set_device and the parent must report it (no hang)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--bank-workers", "1"], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" not in out.stderr
