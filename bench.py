#!/usr/bin/env python3
"""bench.py -- clips/s of the feature hot path, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]                 (N > 1: starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                          (the driver's form: ranks come from the env)

Workloads (BASELINE.json `configs`):
  default            configs[1]: per GPU 1024 synthetic 3 s @ 16 kHz int16 clips resident in HBM -> 512-pt STFT (hop 160)
                     -> 64 Slaney mels -> dB (top_db 80) -> 40 MFCC + delta + delta-delta, written as frame-level rows
                     [1024, 120, 301] fp32 and pooled mean/std rows [1024, 240]; pooled rows -> MLP head -> logits and,
                     for N > 1, the single collective of the path: an RCCL all-gather of the [1024, 3] logits.  Weak
                     scaling (every rank owns its own 1024-clip shard).
  --head cnn         configs[2]: the same launch with the 2-layer CNN head (HIP kernels, MFMA) on the frame-level rows.
  --config 4         configs[3]: 100 000 clips (clip j = bank clip j mod 1024) sharded contiguously over the ranks
                     (12 500 per GPU at N = 8), resident; pooled rows -> MLP head -> one all-gather of [n_local, 3]; the
                     gathered arg-max is checked against the single-GPU result of the bank.  Strong scaling.
  --config 5         configs[4]: ONE list of 1024 x N clips of 1-10 s at 22.05 kHz (lengths ~ U[1, 10] s, seed 4321), cut into
                     contiguous shards balanced by cumulative sample count (rsd_amd.pipeline.shard_by_samples);
                     2048-pt STFT hop 512, 128 mels, 20 MFCC + delta + delta2; every rank groups ITS clips by ceil(seconds)
                     into 10 buckets, each bucket is one ragged launch padded to its own longest clip (frame rows
                     [n_b, 60, T_b]) followed by the CNN head in float16 MFMA with the clips' true widths; logits go back
                     into clip order, one all-gather of the (unequal) shards, and the gathered arg-max is checked against
                     a single-GPU pass over the whole list.
  --preset reference_149   the reference's own parameterisation (2048 / 512 / 128 mels / 20 MFCC + chroma, parity mode)
                     on the same 1024 clips per GPU: pooled 144-d rows + tuning index -> MLP(149 -> 128 -> 64 -> 3).

A step = one pass of the hot path over the rank's resident batch.  Steps are independent batches, so they alternate
between `--lanes` HIP streams (rsd_amd.pipeline.LaneExecutor, default 3, one extractor handle each); all lanes are joined
before the closing fence.  Order of a run: CPU baseline (host only), then -- at N = 1 with the default workload -- the
PCIe-inclusive streaming measurement and the device copy-bandwidth probe (both reported, `config.before_warmup`; they also
bring a cold GPU to its working clocks), then exactly W untimed warm-up steps of the timed workload, then K timed steps
between barrier + synchronize fences; the slowest rank's wall time counts.  After the timed region: the same loop once more on
the by then warm GPU (`config.steady_state`: both feature kernels run ~14 % faster after the first ~100 launches of a
process -- reported next to `value`, never instead of it), the one-stream figure and the per-kernel timing launches.

Rank 0 prints ONE JSON line: the metric, a `roofline` object for the dominant kernel (durations from HIP events on the launch
stream, bytes = SURVEY 8d's algorithmic figure, no intermediates) and, at N = 1, a `cpu_baseline` object (the NumPy oracle
timed on the host cores on a bounded sample of the same clips) and the PCIe-inclusive rate from host memory.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 1024           # clips per GPU per step (BASELINE.json configs[1])
N_SAMPLES = 48000      # 3 s @ 16 kHz
TOTAL_CONFIG4 = 100000  # BASELINE.json configs[3]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TF = 2500.0   # dense fp16 / bf16 MFMA
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA (= the fp32 vector rate)
CSRC = os.path.join(ROOT, "recognizing-speech-dysfluencies-in-stuttering_amd", "csrc")


def kernel_source_id():
    """Short hash of the kernel sources: ties counter files under profiles/ to the build they were measured on."""
    h = hashlib.sha256()
    for name in ("dysf_kernels.hip", "dysf_fft.hpp", "dysf_kernels.h", "dysf_cnn.hip", "dysf_mlp.hip"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def _cpu_worker(job):
    from oracle import dysf_oracle as O
    clip, preset = job
    y = O.pcm16_to_float(clip)
    if preset == "reference_149":
        return O.extract_audio_features(y, 16000, O.REFERENCE_149).shape[0]
    return O.frame_level_stack(y, O.BENCH_512).shape[1]


def cpu_baseline(bank, preset, budget_s=15.0):
    """Reported baseline only: the CPU oracle (a NumPy/SciPy port of the reference's librosa calls) on a bounded
    sample of the same clips, one clip per task over the host cores available to this process."""
    import multiprocessing as mp

    from threadpoolctl import threadpool_limits

    cores = min(16, len(os.sched_getaffinity(0)))  # a one-GPU box's CPU share is 16 cores
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        n1 = 0
        while n1 < 8 or time.perf_counter() - t0 < 2.0:
            _cpu_worker((bank[n1 % len(bank)], preset))
            n1 += 1
        one = n1 / (time.perf_counter() - t0)
        sample = int(min(len(bank), max(cores * 4, one * cores * budget_s * 0.5)))
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_cpu_worker, [(bank[i], preset) for i in range(min(cores, len(bank)))])  # warm the workers
            t0 = time.perf_counter()
            pool.map(_cpu_worker, [(bank[i % len(bank)], preset) for i in range(sample)], chunksize=4)
            dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    what = ("extract_audio_features (REFERENCE_149: 2048-pt STFT, 128 mels, 20 MFCC + deltas + chroma incl. the tuning estimate)"
            if preset == "reference_149" else "frame_level_stack (BENCH_512)")
    return {"value": round(sample / dt, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{sample} clips of the same synthetic bank through oracle/dysf_oracle.py {what}, one clip per task, "
                      f"{cores} processes x 1 BLAS thread",
            "one_thread_clips_per_s": round(one, 2), "cpu_model": model}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 20 with --config 4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 100, 10 with --config 4: the core clock needs ~100 launches of this workload to settle, tools/ramp.py)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state re-measurement after the timed region")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="2 = BASELINE configs[1] (1024 clips per GPU, weak scaling); 4 = configs[3] (100k clips, strong scaling); "
                         "5 = configs[4] (1-10 s clips at 22.05 kHz in 10 length buckets, 128 mels, float16-MFMA CNN head)")
    ap.add_argument("--preset", default="bench_512", choices=("bench_512", "reference_149"))
    ap.add_argument("--head", default="mlp", choices=("mlp", "cnn"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the PCIe-inclusive measurement (N = 1 only)")
    ap.add_argument("--no-reference-leg", action="store_true",
                    help="skip the short REFERENCE_149 leg (config.reference_149) the default run adds after its timed region")
    ap.add_argument("--eager-head", action="store_true",
                    help="run the MLP head as the eager torch module instead of the one-kernel HIP head (dysf_mlp_*)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid: all ranks run on cuda:0 and the collective goes over gloo (never a measurement)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="HIP streams (and extractor handles) that consecutive batches alternate between; 1 = serial")
    ap.add_argument("--bank-workers", type=int, default=0,
                    help="processes generating the synthetic bank (0 = auto; use 1 under rocprofv3 --pmc, which "
                         "initialises the GPU before main() and must not be followed by a fork)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell: start the N rank processes (fresh interpreters, nothing has touched
    the GPU in this one), wait for them and pass on the first non-zero exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:  # a failed rank leaves the others in a collective: do not wait for their timeout
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    args = parse_args()
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if args.steps is None:
        args.steps = 20 if args.config == 4 else 200
    if args.warmup is None:
        args.warmup = 10 if args.config == 4 else 100

    import numpy as np
    import torch
    import torch.distributed as dist

    import rsd_amd
    from rsd_amd.classifier import FusedMLPHead, MLPHead
    from rsd_amd.pipeline import LaneExecutor, gather_rows, gather_rows_by_bounds, shard_by_samples, shard_range
    from rsd_amd.synth import synth_bank

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.share_gpu:   # rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0, gloo carries the gather
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.head == "cnn" and (args.preset != "bench_512" or args.config != 2):
        raise SystemExit("--head cnn runs on the frame-level rows of the default workload (configs[2])")
    # host-only work first (process pools fork; keep that ahead of any GPU initialisation)
    workers = args.bank_workers or max(1, min(8, len(os.sched_getaffinity(0)) // max(1, world)))
    bank = synth_bank(BATCH, N_SAMPLES, 16000, workers=workers)
    cpu = cpu149 = None
    if world == 1 and not args.no_cpu_baseline and args.config != 5:
        cpu = cpu_baseline(bank[:256], args.preset)
        if args.config == 2 and args.preset == "bench_512" and args.head == "mlp" and not args.no_reference_leg:
            cpu149 = cpu_baseline(bank[:256], "reference_149", budget_s=8.0)   # for config.reference_149

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    bank_dev = torch.from_numpy(bank).to(dev)                      # [1024, 48000] int16
    if args.config == 4:   # this rank's contiguous shard of the 100 000-clip index space, clip j = bank clip j mod 1024
        total = TOTAL_CONFIG4
        lo, hi = shard_range(total, rank, world)
        n_local = hi - lo
        pcm = bank_dev[(torch.arange(lo, hi, device=dev) % BATCH)].reshape(-1).contiguous()
    else:
        total = BATCH * world
        lo, n_local = rank * BATCH, BATCH
        pcm = bank_dev.reshape(-1)
    offsets = (torch.arange(0, n_local + 1, dtype=torch.int64) * N_SAMPLES).to(dev)

    buckets = None
    if args.config == 5:
        if args.preset != "bench_512" or args.head != "mlp":
            raise SystemExit("--config 5 fixes its own parameters (2048 / 512 / 128 mels at 22.05 kHz) and head (CNN, float16 MFMA)")
        sr5 = 22050
        # one global clip list (seed 4321): lengths, and where in the 49 M samples of synthetic audio each clip is cut from
        total = BATCH * world
        rng5 = np.random.default_rng(4321)
        lens_all = (rng5.uniform(1.0, 10.0, size=total) * sr5).astype(np.int64)
        src5 = torch.from_numpy(bank.reshape(-1)).to(dev)
        starts_all = rng5.integers(0, src5.numel() - int(lens_all.max()), size=total)
        bounds5 = shard_by_samples(lens_all, world)          # contiguous shards, balanced by cumulative sample count
        lo, n_local = int(bounds5[rank]), int(bounds5[rank + 1] - bounds5[rank])

        def make_buckets(first, count):
            """Clips [first, first + count) of the list grouped by ceil(seconds): resident PCM + offsets per bucket; `idx` is
            the clip's position inside the range."""
            ln, st = lens_all[first:first + count], starts_all[first:first + count]
            sec = np.minimum(10, np.ceil(ln / sr5).astype(np.int64))
            out = []
            for b in range(1, 11):
                idx = np.nonzero(sec == b)[0]
                if len(idx) == 0:
                    continue
                bl = ln[idx]
                boff = np.concatenate([[0], np.cumsum(bl)]).astype(np.int64)
                bpcm = torch.cat([src5[int(s0):int(s0) + int(n)] for s0, n in zip(st[idx], bl)])
                out.append({"idx": torch.from_numpy(idx).to(dev), "pcm": bpcm, "off": torch.from_numpy(boff).to(dev),
                            "max": int(bl.max()), "n": len(idx), "samples": int(bl.sum())})
            return out
        buckets = make_buckets(lo, n_local)
        lens = lens_all[lo:lo + n_local]
        total_samples5 = int(lens_all.sum())
    params = rsd_amd.REFERENCE_149 if args.preset == "reference_149" else rsd_amd.BENCH_512
    if args.config == 5:
        params = rsd_amd.FeatureParams(sr=22050, chroma=False, parity=False)   # 2048 / 512 / 128 mels / 20 MFCC, librosa defaults
    want_frames = args.preset == "bench_512" and args.config == 2
    ex = rsd_amd.FeatureExtractor(params, local_rank)
    T = params.n_frames(N_SAMPLES)
    rng = np.random.default_rng(0)
    n_in = params.total_feature_len if args.preset == "reference_149" else ex.pooled_dim
    dims = [n_in, 128, 64, 3]
    head_w = [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)]
    head_b = [np.zeros(dims[i + 1]) for i in range(3)]
    cnn = None
    if args.head == "cnn" or args.config == 5:
        from rsd_amd.classifier import CNNHead, FusedCNNHead
        torch.manual_seed(0)
        cnn = FusedCNNHead(CNNHead().to(dev))
        head = None
    elif args.eager_head:   # the torch module (seven small launches); the default is the one-kernel HIP head
        head = MLPHead(np.zeros(dims[0]), np.ones(dims[0]), head_w, head_b).to(dev)
    else:
        head = FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), head_w, head_b, device=local_rank)

    def run_head(res):
        if cnn is not None:
            return cnn(res["frames"])
        x = res["pooled"]
        if x.shape[1] < n_in:   # the five always-zero text features (pipeline1.py:243-244)
            x = torch.cat([x, x.new_zeros((x.shape[0], n_in - x.shape[1]))], dim=1)
        return head(x)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    n_lanes = max(1, args.lanes)
    lanes = LaneExecutor(params, local_rank, n_lanes, head=None, first_extractor=ex)
    kw = dict(pooled=True, frames=want_frames, tuning=(args.preset == "reference_149"))
    frame_pitch = T
    if want_frames and cnn is None and not params.parity and os.environ.get("BENCH_FRAME_PITCH", "") != "T":
        # the frame-level rows [3 * n_mfcc][T] are written on a 64-byte row pitch (T rounded up to 16 floats, padding columns zero):
        # aligned 256-byte stores that the MFCC kernel streams past L2; the bytes counted are T per row
        frame_pitch = (T + 15) // 16 * 16
        kw["frame_stride"] = frame_pitch

    def after(res):
        res["logits"] = run_head(res)
        res["all_logits"] = gather_rows(res["logits"], total, rank, world)

    def run_buckets(bks, n_rows):
        """One ragged launch + CNN per length bucket (lanes alternate), logits scattered back into clip order."""
        logits = torch.empty((n_rows, 3), dtype=torch.float32, device=dev)
        status = torch.empty(n_rows, dtype=torch.int32, device=dev)
        for i, b in enumerate(bks):
            k = i % n_lanes
            with torch.cuda.stream(lanes.streams[k]):
                r = lanes.extractors[k].extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)
                cols = ((b["off"][1:] - b["off"][:-1]) // params.hop + 1).to(torch.int32)
                logits.index_copy_(0, b["idx"], cnn(r["frames"], cols=cols))
                status.index_copy_(0, b["idx"], r["status"])
                for t in (logits, status):
                    t.record_stream(lanes.streams[k])
        lanes.join()
        lanes.fork()
        return logits, status

    def step5():
        """configs[4]: this rank's shard of the list, bucket by bucket, then the one all-gather of the (unequal) shards."""
        logits, status = run_buckets(buckets, n_local)
        return {"logits": logits, "status": status, "all_logits": gather_rows_by_bounds(logits, bounds5, rank, world)}

    def step():
        if args.config == 5:
            return step5()
        if args.config == 4:   # one big resident shard: 2048-clip chunks alternate between the lanes, then head + gather
            res = lanes.extract_large(pcm, offsets, N_SAMPLES, chunk_clips=2048, pooled=True, frames=False)
            after(res)
            return res
        return lanes.submit(pcm, offsets, N_SAMPLES, after=after, **kw)

    # The two measurements that do not belong to the timed workload run FIRST (reported in the JSON, listed in
    # config.before_warmup): the PCIe-inclusive streaming rate and the device copy-bandwidth probe.  The timed region then
    # starts on a GPU that is already at its working clocks -- with --warmup 5 alone the first ~50 steps of a cold process run
    # 15-20 % slower (W 5 / K 20: 4.15 M clips/s, W 200 / K 20: 4.89 M on the same box), which says nothing about the kernels.
    # PCIe-inclusive rate (N = 1): host int16 -> H2D overlapped with the kernels -> pooled rows back on the host
    host_rate = None
    if world == 1 and not args.no_host_rate and args.config == 2 and cnn is None:
        from rsd_amd.ingest import StreamingExtractor, pinned_empty
        n_host = 32768   # 3.1 GB of int16 PCM: the pipeline's fill (first chunk's copy, not overlapped) is 1/16 of the run
        hoff = np.arange(n_host + 1, dtype=np.int64) * N_SAMPLES
        host_rate = {"clips": n_host}
        # (i) the producer decodes into page-locked memory (pinned_empty): H2D starts straight from it;
        # (ii) an ordinary pageable array, page-locked for the call by hipHostRegister (its cost is inside the time)
        for name, flat in (("pinned_source", pinned_empty(n_host * N_SAMPLES)), ("registered_pageable_source", np.empty(n_host * N_SAMPLES, np.int16))):
            flat.reshape(n_host // BATCH, -1)[:] = bank.reshape(1, -1)
            se = StreamingExtractor(ex, chunk_clips=2048)
            se.run(flat[: 4096 * N_SAMPLES], hoff[:4097])
            # three consecutive passes, the last one reported: the first ones still run on a GPU whose clocks are coming up
            # from idle (tools/ingest_chunks.py: 0.28-0.40 M clips/s, then 0.50, then 0.57 M on every later pass)
            passes = []
            for _ in range(3):
                th = time.perf_counter()
                o2 = se.run(flat, hoff)
                passes.append(time.perf_counter() - th)
            th = passes[-1]
            assert int(np.abs(o2["status"]).sum()) == 0
            host_rate[name] = {"clips_per_s": round(n_host / th, 1), "first_pass_clips_per_s": round(n_host / passes[0], 1),
                               "pcm_gb_per_s": round(n_host * N_SAMPLES * 2 / th / 1e9, 2),
                               "h2d_gb_per_s_on_the_copy_stream": round(se.h2d_gbs, 2) if se.h2d_gbs else None, "path": se.describe()}
            del flat
        host_rate["clips_per_s"] = host_rate["pinned_source"]["clips_per_s"]
    # measured device-to-device copy rate (report the fraction of the measured copy bandwidth too)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(5):
        dst.copy_(src)
    b1.record()
    torch.cuda.synchronize(dev)
    copy_gbs = 5 * 2 * src.numel() * 4 / (b0.elapsed_time(b1) * 1e-3) / 1e9
    del src, dst
    per_kernel = []
    with torch.no_grad():
        lanes.fork()
        for _ in range(args.warmup):
            step()
        lanes.join()
        fence()
        # one event pair around the whole timed loop: hundreds of live timing events (one pair per step) exhaust the
        # runtime's signal pool and serialise the launches
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        lanes.fork()
        for _ in range(args.steps):
            res = step()
        lanes.join()
        ev1.record()
        fence()
        dt = time.perf_counter() - t0
        all_logits = res["all_logits"]
        # for transparency: the same loop again, now that the GPU has been running this workload for a while (both feature
        # kernels get ~14 % faster over the first ~100 launches of a process, tools/ramp.py).  Reported as
        # config.steady_state, never as `value`.
        steady = None
        if args.config == 2 and not args.no_steady:
            n_steady = max(200, args.steps)
            lanes.fork()
            for _ in range(n_steady):   # (its own warm-up)
                step()
            lanes.join()
            fence()
            ts0 = time.perf_counter()
            lanes.fork()
            for _ in range(n_steady):
                step()
            lanes.join()
            fence()
            ts = torch.tensor([time.perf_counter() - ts0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            steady = {"steps": n_steady, "ms_per_step": round(float(ts.item()) / n_steady * 1e3, 4),
                      "clips_per_s": round(total * n_steady / float(ts.item()), 1)}
        # for transparency: the same step strictly serially on one stream (outside the timed region, not `value`)
        serial_ms = None
        if n_lanes > 1 and args.config == 2:
            sv0, sv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_serial = min(50, args.steps)
            sv0.record()
            for _ in range(n_serial):
                r1 = ex.extract_batch_device(pcm, offsets, N_SAMPLES, **kw)
                gather_rows(run_head(r1), total, rank, world)
            sv1.record()
            torch.cuda.synchronize(dev)
            serial_ms = sv0.elapsed_time(sv1) / n_serial
        # per-kernel durations: a few extra launches outside the timed region, HIP events inside the library (on the
        # launch stream) around each kernel
        k_pcm, k_off, k_n = (pcm, offsets, n_local) if args.config == 2 else (bank_dev.reshape(-1), offsets[:BATCH + 1], BATCH)
        cnn_ms = None
        feat5_ms = cnn5_ms = None
        if args.config == 5:   # feature launches and CNN launches of all buckets, each timed back to back on one stream
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            frs = []
            e[0].record()
            for b in buckets:
                frs.append(ex.extract_batch_device(b["pcm"], b["off"], b["max"], pooled=False, frames=True)["frames"])
            e[1].record()
            for b, fr in zip(buckets, frs):
                cnn(fr, cols=((b["off"][1:] - b["off"][:-1]) // params.hop + 1).to(torch.int32))
            e[2].record()
            torch.cuda.synchronize(dev)
            feat5_ms, cnn5_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
            del frs
        ex.kernel_timing(True)
        for _ in range(10):
            # three launches back to back, the events of the last one read: like the launches of the timed region, it starts
            # on a GPU that is already busy (a launch that follows a host synchronisation measures 2-4 % longer)
            for _ in range(3):
                ex.extract_batch_device(k_pcm, k_off, N_SAMPLES, **kw)
            per_kernel.append(ex.kernel_timing(True))
        ex.kernel_timing(False)
        if cnn is not None and args.config != 5:
            fr = ex.extract_batch_device(k_pcm, k_off, N_SAMPLES, **kw)["frames"]
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnn(fr)
            c0.record()
            for _ in range(10):
                cnn(fr)
            c1.record()
            torch.cuda.synchronize(dev)
            cnn_ms = c0.elapsed_time(c1) / 10
        # configs[3]: the gathered arg-max must equal the single-GPU result of the same clips (clip j = bank clip j mod 1024)
        argmax_ok = None
        if args.config == 4:
            ref = run_head(ex.extract_batch_device(bank_dev.reshape(-1), offsets[:BATCH + 1], N_SAMPLES, pooled=True)).argmax(dim=1)
            got = all_logits.argmax(dim=1)
            argmax_ok = bool(torch.equal(got, ref[torch.arange(total, device=dev) % BATCH]))
            assert argmax_ok, "gathered arg-max differs from the single-GPU result"
        logits_close = None
        if args.config == 5:
            # configs[4]: a single-GPU pass over the WHOLE list (its buckets hold other clips and are padded to other widths
            # than the shards' buckets) must give the gathered result.  A clip's features and CNN output do not depend on
            # its batch neighbours; the float16 head's partial sums may be ordered differently, hence the arg-max is
            # compared where the two leading logits are more than 1e-3 apart and the logits within 1e-3.
            all_buckets = make_buckets(0, total)   # (built on the main stream:
            lanes.fork()                            #  the lanes must not start before it is there)
            ref_logits, _ = run_buckets(all_buckets, total)
            lanes.join()
            torch.cuda.synchronize(dev)
            top2 = torch.topk(ref_logits, 2, dim=1).values
            clear = (top2[:, 0] - top2[:, 1]) > 1e-3
            logits_close = float((all_logits - ref_logits).abs().max())
            argmax_ok = bool(torch.equal(all_logits.argmax(dim=1)[clear], ref_logits.argmax(dim=1)[clear])) and logits_close <= 1e-3
            assert argmax_ok, f"gathered logits differ from the single-GPU pass over the whole list (max |d| {logits_close})"
    assert int(res["status"].abs().sum()) == 0 and all_logits.shape == (total, 3), (int(res["status"].abs().sum()), all_logits.shape)
    assert bool(torch.isfinite(all_logits).all())
    # The reference's own parameterisation (pipeline1.py:216-228) as a short extra leg of the default run, outside the timed
    # region: the same 1024 clips through REFERENCE_149 (parity mode) + MLP(149), W 20 / K 50 on the lanes.
    ref149 = None
    if world == 1 and args.config == 2 and args.preset == "bench_512" and cnn is None and not args.no_reference_leg:
        p149 = rsd_amd.REFERENCE_149
        ex149 = rsd_amd.FeatureExtractor(p149, local_rank)
        lanes149 = LaneExecutor(p149, local_rank, n_lanes, head=None, first_extractor=ex149)
        d149 = [p149.total_feature_len, 128, 64, 3]
        head149 = FusedMLPHead(np.zeros(d149[0]), np.ones(d149[0]), [rng.standard_normal((d149[i], d149[i + 1])) / np.sqrt(d149[i]) for i in range(3)],
                               [np.zeros(d149[i + 1]) for i in range(3)], device=local_rank)

        def after149(r):
            x = r["pooled"]
            r["logits"] = head149(torch.cat([x, x.new_zeros((x.shape[0], d149[0] - x.shape[1]))], dim=1))

        with torch.no_grad():
            lanes149.fork()
            for _ in range(20):
                lanes149.submit(pcm, offsets, N_SAMPLES, after=after149, pooled=True, tuning=True)
            lanes149.join()
            torch.cuda.synchronize(dev)
            t149 = time.perf_counter()
            lanes149.fork()
            for _ in range(50):
                r149 = lanes149.submit(pcm, offsets, N_SAMPLES, after=after149, pooled=True, tuning=True)
            lanes149.join()
            torch.cuda.synchronize(dev)
            t149 = time.perf_counter() - t149
            assert int(r149["status"].abs().sum()) == 0
            ex149.kernel_timing(True)
            pk = []
            for _ in range(6):
                for _ in range(2):
                    ex149.extract_batch_device(pcm, offsets, N_SAMPLES, pooled=True, tuning=True)
                pk.append(ex149.kernel_timing(True))
            ex149.kernel_timing(False)
        km = {k: float(np.median([q[k] for q in pk])) for k in ("stft_mel", "mfcc", "chroma")}
        b149 = 2 * N_SAMPLES + 4 * ex149.pooled_dim
        kp = km["stft_mel"] + km["mfcc"] + km["chroma"]
        ref149 = {"what": "pipeline1.py:216-228 parameterisation (2048 / 512 / 128 mels / 20 MFCC + deltas + chroma_stft with its tuning estimate, "
                          "parity mode) on the same 1024 resident clips + MLP(149 -> 128 -> 64 -> 3); W 20 / K 50 after the timed region",
                  "clips_per_s": round(BATCH * 50 / t149, 1), "ms_per_step": round(t149 / 50 * 1e3, 4),
                  "kernel_ms": {"dysf_stft_mel_exact2048_kernel": round(km["stft_mel"], 4), "dysf_mfcc_kernel": round(km["mfcc"], 4),
                                "dysf_chroma_kernel": round(km["chroma"], 4)},
                  "bytes_per_clip": b149, "achieved_gbs": round(BATCH * b149 / (kp * 1e-3) / 1e9, 1),
                  "frac": round(BATCH * b149 / (kp * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        del lanes149, ex149
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kms = ev0.elapsed_time(ev1) / max(1, args.steps)  # GPU time per step on the launch stream

    if rank == 0:
        clips_per_s = total * args.steps / dt if args.config == 4 else BATCH * world * args.steps / dt
        out_elems = ex.pooled_dim + (ex.frame_rows * T if want_frames else 0)
        bytes_per_clip = 2 * N_SAMPLES + 4 * out_elems        # PCM read once + outputs written once
        algo_bytes = BATCH * bytes_per_clip                   # per 1024-clip launch (the unit the kernels are timed on)
        med = lambda k: float(np.median([p[k] for p in per_kernel]))  # noqa: E731
        k_stft, k_mfcc, k_chroma = med("stft_mel"), med("mfcc"), med("chroma")
        k_path = k_stft + k_mfcc + (k_chroma if params.chroma else 0.0)
        achieved = algo_bytes / (k_path * 1e-3) / 1e9
        # useful arithmetic of the path: FFT 2.5 n log2 n per frame, the non-zero mel taps, the DCT
        import math
        n_bins = params.n_fft // 2 + 1
        mel_nnz = int(np.count_nonzero(ex.table(1))) if hasattr(ex, "table") else params.n_mels * n_bins
        useful_flops = BATCH * T * (2.5 * params.n_fft * math.log2(params.n_fft) + 2 * mel_nnz + 2 * params.n_mfcc * params.n_mels)
        a2 = params.n_fft == 512 and not params.parity and not params.chroma and os.environ.get("DYSF_STFT") != "1"
        ex2048 = params.n_fft == 2048 and params.parity and os.environ.get("DYSF_EXACT") != "1"
        stft_name = "dysf_stft_mel_wave_kernel" if a2 else ("dysf_stft_mel_exact2048_kernel" if ex2048 else "dysf_stft_mel_kernel")
        # hardware counters are collected in separate rocprofv3 --pmc passes (tools/pmc.sh) and committed with the id
        # of the kernel sources they were measured on; a stale file is not quoted
        counters = None
        cpath = os.path.join(ROOT, "profiles", "r03_counters.json")
        if os.path.exists(cpath):
            try:
                c = json.load(open(cpath)).get(args.preset if cnn is None else "cnn")
                if c and c.get("kernel_source_id") == kernel_source_id():
                    counters = c
            except Exception:
                counters = None
        workload = {
            (2, "bench_512", "mlp"): "BASELINE.json configs[1]: 1024 clips x 48000 int16 samples per GPU resident in HBM, 512-pt STFT hop "
                                     "160, 64 Slaney mels, top_db 80, 40 MFCC + delta + delta2; outputs frame-level [1024,120,301] fp32 + "
                                     "pooled [1024,240] fp32, then MLP head -> logits [1024,3] and (N>1) RCCL all-gather of the logits",
            (2, "bench_512", "cnn"): "BASELINE.json configs[2]: configs[1] + the 2-layer CNN head (conv 1->32, pool, conv 32->64, global "
                                     "average, linear 64->3; random weights) on the frame-level rows [1024,120,301], HIP kernels",
            (4, "bench_512", "mlp"): "BASELINE.json configs[3]: 100000 clips (clip j = bank clip j mod 1024) x 48000 int16 samples, contiguous "
                                     "shards resident in HBM, BENCH_512 pooled rows [n_local,240] -> MLP head -> one all-gather of [n_local,3] "
                                     "logits; gathered arg-max checked against the single-GPU result",
            (2, "reference_149", "mlp"): "the reference's own parameterisation (pipeline1.py:216-228): 1024 clips x 48000 int16 samples per "
                                         "GPU resident in HBM, 2048-pt STFT hop 512, 128 Slaney mels, 20 MFCC + delta + delta2 + chroma_stft "
                                         "(tuning estimate), parity mode (float64 FFT), pooled [1024,144] fp32 -> MLP(149->128->64->3)",
        }.get((args.config, args.preset, args.head), "")
        out = {
            "metric": "clips/sec (3 s@16 kHz, 40-MFCC) end-to-end", "value": round(clips_per_s, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.config == 4 else "weak", "vs_baseline": None,
            "dtype": "f64" if params.parity else "f32", "data": "synthetic",
            "config": {"workload": workload, "input": "PCM resident in HBM before the timed region (PCIe-inclusive rate: config.from_host)",
                       "clips_per_gpu": n_local, "total_clips": total, "samples_per_clip": N_SAMPLES, "n_fft": params.n_fft,
                       "hop": params.hop, "n_mels": params.n_mels, "n_mfcc": params.n_mfcc, "frames_per_clip": T, "frame_row_pitch": frame_pitch,
                       "head": args.head, "gathered_argmax_equals_single_gpu": argmax_ok,
                       "one_stream_ms_per_step": None if serial_ms is None else round(serial_ms, 4),
                       "step_gpu_ms": round(kms, 4), "parallelism": f"clip-sharded x{world}",
                       "launch": ("eager, batches alternate between %d streams" % n_lanes) if n_lanes > 1 else "eager, one stream",
                       "from_host": host_rate,
                       "steady_state": steady,
                       "before_warmup": ["from_host streaming measurement (2 sources x 3 passes x 32768 clips)" if host_rate else None, "device copy-bandwidth probe (6 x 1 GiB)"],
                       "pcie_ceiling_clips_per_s": round(63e9 / (2 * N_SAMPLES), 1)},
            # HBM is the roofline SURVEY 8d prices this path against: bytes = PCM read once + outputs written once (no
            # intermediates), time = the feature launch's kernels back to back on one stream.  The kernels are NOT bandwidth
            # bound: `limited_by` says what the counters show.
            "roofline": {"bound": "hbm", "kernel": "feature launch: " + stft_name + " + dysf_mfcc_kernel" + (" + dysf_chroma_kernel" if params.chroma else ""),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (counters or {}).get("hbm_bytes_per_launch"),
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_clip": bytes_per_clip, "clips_per_launch": BATCH,
                         "kernel_ms": round(k_path, 4), "measured_copy_gbs": round(copy_gbs, 1),
                         "frac_of_measured_copy_bw": round(achieved / copy_gbs, 4),
                         "limited_by": (counters or {}).get("limited_by", "the vector pipe and dependent latencies, not HBM (see DESIGN.md section 5)"),
                         # the other roof: useful float32 arithmetic (FFT butterflies + non-zero mel taps + DCT) against the
                         # vector / matrix peak -- the path sits at the float32 ridge, so both are reported
                         "valu": {"useful_flops_per_launch": int(useful_flops), "achieved_tflops": round(useful_flops / (k_path * 1e-3) / 1e12, 2),
                                  "peak_tflops": MFMA_F32_PEAK_TF, "frac": round(useful_flops / (k_path * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)},
                         "valu_busy": (counters or {}).get("valu_busy"), "mfma_busy": (counters or {}).get("mfma_busy"),
                         "counters_from": (counters or {}).get("source"),
                         "kernels": {stft_name: {"ms": round(k_stft, 4), "algorithmic_bytes": BATCH * 2 * N_SAMPLES,
                                                              "achieved_gbs": round(BATCH * 2 * N_SAMPLES / (k_stft * 1e-3) / 1e9, 1)},
                                     "dysf_mfcc_kernel": {"ms": round(k_mfcc, 4), "algorithmic_bytes": BATCH * 4 * out_elems,
                                                          "achieved_gbs": round(BATCH * 4 * out_elems / (k_mfcc * 1e-3) / 1e9, 1)}},
                         "clips_per_s_kernels_only": round(BATCH / (k_path * 1e-3), 1)},
        }
        if params.chroma:
            out["roofline"]["kernels"]["dysf_chroma_kernel"] = {"ms": round(k_chroma, 4)}
        if cnn is not None and args.config != 5:   # configs[2]: the dominant kernel is the CNN's conv2 (implicit GEMM on MFMA)
            flops = BATCH * (2 * 32 * 9 * 120 * 301 + 2 * 64 * 32 * 9 * 60 * 150)
            tf = flops / (cnn_ms * 1e-3) / 1e12
            prec = getattr(cnn, "precision", "f16")
            peak = MFMA_F16_PEAK_TF if prec == "f16" else MFMA_F32_PEAK_TF
            out["roofline"] = {"bound": "mfma", "kernel": "dysf_cnn_* (conv1 + pool + conv2 + average + linear)", "achieved": round(tf, 2),
                               "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                               "algorithmic_flops_per_launch": flops, "kernel_ms": round(cnn_ms, 4), "conv2_precision": prec,
                               "feature_kernels_ms": round(k_path, 4)}
        if args.config == 5:
            T5 = lens // params.hop + 1
            flops5 = int((2 * 32 * 9 * 60 * T5 + 2 * 64 * 32 * 9 * 30 * (T5 // 2)).sum())
            tf5 = flops5 / (cnn5_ms * 1e-3) / 1e12
            out["config"].update({
                "workload": "BASELINE.json configs[4]: per GPU 1024 clips with lengths ~ U[1, 10] s at 22 050 Hz (seed 4321 + rank; no resampling, the "
                            "rate is a parameter), 2048-pt STFT hop 512, 128 Slaney mels, 20 MFCC + delta + delta2; clips grouped by ceil(seconds) into "
                            "10 buckets, one ragged launch per bucket padded to the bucket's longest clip, CNN head in float16 MFMA with the clips' "
                            "true widths, logits back in clip order, (N>1) RCCL all-gather",
                "sr": sr5, "seed": 4321, "clips_per_bucket": {str(int(np.ceil(b["max"] / sr5))): b["n"] for b in buckets},
                "shards": "contiguous, balanced by cumulative sample count (rsd_amd.pipeline.shard_by_samples); clips per rank: "
                          + str([int(bounds5[r + 1] - bounds5[r]) for r in range(world)]),
                "gathered_logits_max_abs_diff_vs_single_gpu_pass": logits_close,
                "samples_per_gpu": int(lens.sum()), "samples_total": total_samples5, "samples_per_clip": None, "frames_per_clip": None,
                "hours_of_audio_per_s": round(total_samples5 / sr5 / 3600 / (dt / args.steps), 1),
                "feature_launches_ms": round(feat5_ms, 4), "cnn_launches_ms": round(cnn5_ms, 4), "head": "cnn (float16 MFMA conv2)"})
            out["dtype"] = "f16"
            out["roofline"] = {"bound": "mfma", "kernel": "dysf_cnn_* over the ten buckets (conv1 + pool + conv2 + average + linear, per-clip widths)",
                               "achieved": round(tf5, 2), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf5 / MFMA_F16_PEAK_TF, 4),
                               "traffic": None, "algorithmic_flops_per_launch": flops5, "kernel_ms": round(cnn5_ms, 4),
                               "feature_kernels_ms": round(feat5_ms, 4),
                               "feature_algorithmic_bytes": int(2 * int(lens.sum()) + 4 * 60 * int(T5.sum())),
                               "feature_achieved_gbs": round((2 * int(lens.sum()) + 4 * 60 * int(T5.sum())) / (feat5_ms * 1e-3) / 1e9, 1)}
        if ref149 is not None:
            if cpu149 is not None:
                ref149["cpu_baseline"] = cpu149
            out["config"]["reference_149"] = ref149
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
