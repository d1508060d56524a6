#!/usr/bin/env python3
"""bench.py -- clips/s of the feature hot path on BASELINE.json configs[1], one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one resident batch: 1024 synthetic 3 s @ 16 kHz int16 clips ->
512-pt STFT (hop 160) -> 64 Slaney mels -> dB (top_db 80) -> 40 MFCC + delta + delta-delta, written both as
frame-level rows [1024, 120, 301] fp32 (the CNN input of configs[2]) and as pooled mean/std rows [1024, 240];
then the pooled rows go through a small MLP head to per-clip logits and, for N > 1, the single collective of the
path: an RCCL all-gather of the [1024, 3] logits.  Every rank owns its own 1024-clip shard (weak scaling); the
PCM is already in HBM when the timed region starts.  Steps are independent batches, so they are submitted round-robin to
`--lanes` HIP streams (rsd_amd.pipeline.LaneExecutor, default 3, each lane with its own extractor handle): the STFT kernel
of one step overlaps the MFCC kernel / head / gather of the previous one.  All lanes are joined before the closing fence;
`--lanes 1` is the strictly serial loop.  Ahead of the W warm-up steps the script issues 60 more untimed launches
(first-use work of every lane, GPU clock ramp; `--no-preheat` skips them); the timed region is exactly K steps.

Rank 0 prints ONE JSON line with the metric, a `roofline` object for the dominant kernel (HIP-event timed on the
launch stream) and, at N = 1, a `cpu_baseline` object (the NumPy oracle timed on the host cores on a bounded
sample of the same clips).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 1024          # clips per GPU per step (BASELINE.json configs[1])
N_SAMPLES = 48000     # 3 s @ 16 kHz
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def _cpu_worker(clip):
    from oracle import dysf_oracle as O
    return O.frame_level_stack(O.pcm16_to_float(clip), O.BENCH_512).shape[1]


def cpu_baseline(bank, budget_s=15.0):
    """Reported baseline only: the CPU oracle (a NumPy/SciPy port of the reference's librosa calls) on a bounded
    sample of the same clips, one clip per task over all host cores available to this process."""
    import multiprocessing as mp

    from threadpoolctl import threadpool_limits

    cores = min(16, len(os.sched_getaffinity(0)))  # a one-GPU box's CPU share is 16 cores
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        n1 = 0
        while n1 < 8 or time.perf_counter() - t0 < 2.0:
            _cpu_worker(bank[n1 % len(bank)])
            n1 += 1
        one = n1 / (time.perf_counter() - t0)
        sample = int(min(len(bank), max(cores * 4, one * cores * budget_s * 0.5)))
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_cpu_worker, [bank[i] for i in range(min(cores, len(bank)))])  # warm the workers
            t0 = time.perf_counter()
            pool.map(_cpu_worker, [bank[i % len(bank)] for i in range(sample)], chunksize=4)
            dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(sample / dt, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{sample} clips of the same synthetic bank through oracle/dysf_oracle.py frame_level_stack "
                      f"(BENCH_512), one clip per task, {cores} processes x 1 BLAS thread",
            "one_thread_clips_per_s": round(one, 2), "cpu_model": model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-head", action="store_true",
                    help="run the MLP head as the eager torch module instead of the one-kernel HIP head (dysf_mlp_*)")
    ap.add_argument("--no-preheat", action="store_true", help="skip the 60 extra untimed launches ahead of the warm-up steps")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid: all ranks run on cuda:0 and the collective goes over gloo (never a measurement)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="HIP streams (and extractor handles) that consecutive batches alternate between; 1 = serial")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step (feature launch + head) into a hipGraph and replay it; measured equal to eager "
                         "launches on MI355X (the head's small GEMMs are GPU time, not launch overhead), so off by default")
    ap.add_argument("--bank-workers", type=int, default=0,
                    help="processes generating the synthetic bank (0 = auto; use 1 under rocprofv3 --pmc, which "
                         "initialises the GPU before main() and must not be followed by a fork)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import rsd_amd
    from rsd_amd.classifier import FusedMLPHead, MLPHead
    from rsd_amd.pipeline import LaneExecutor
    from rsd_amd.pipeline import gather_rows
    from rsd_amd.synth import synth_bank

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.share_gpu:   # rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0, gloo carries the gather
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # host-only work first (process pools fork; keep that ahead of any GPU initialisation):
    # every rank owns an identical-size shard of the clip index space
    workers = args.bank_workers or max(1, min(8, len(os.sched_getaffinity(0)) // max(1, world)))
    bank = synth_bank(BATCH, N_SAMPLES, 16000, workers=workers)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(bank[:256])

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pcm = torch.from_numpy(bank.reshape(-1)).to(dev)
    offsets = (torch.arange(0, BATCH + 1, dtype=torch.int64) * N_SAMPLES).to(dev)

    params = rsd_amd.BENCH_512
    ex = rsd_amd.FeatureExtractor(params, local_rank)
    T = params.n_frames(N_SAMPLES)
    rng = np.random.default_rng(0)
    dims = [ex.pooled_dim, 128, 64, 3]
    head_w = [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)]
    head_b = [np.zeros(dims[i + 1]) for i in range(3)]
    if args.eager_head:   # the torch module (seven small launches); the default is the one-kernel HIP head
        head = MLPHead(np.zeros(dims[0]), np.ones(dims[0]), head_w, head_b).to(dev)
    else:
        head = FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), head_w, head_b, device=local_rank)

    per_kernel = []  # HIP events the library records around each of its kernels (dysf_debug_timing)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Optional: the step (feature launch + head) captured once into a hipGraph and replayed; the all-gather of the
    # logits (N > 1) stays an eager RCCL call after the replay.
    graph = None
    if args.graph:
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(3):
                    g_res = ex.extract_batch_device(pcm, offsets, N_SAMPLES, pooled=True, frames=True)
                    g_logits = head(g_res["pooled"])
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                g_res = ex.extract_batch_device(pcm, offsets, N_SAMPLES, pooled=True, frames=True)
                g_logits = head(g_res["pooled"])
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize(dev)

    # Consecutive batches are independent, so they alternate between `--lanes` HIP streams (one extractor handle, i.e.
    # one log-mel workspace, per lane): the STFT kernel of one batch overlaps the MFCC kernel, the head and (N > 1) the
    # logits gather of the previous one.  Both kernels are latency bound with idle pipes, so the
    # overlap is real throughput: measured 257 -> 225 us per step with two lanes.  Everything is joined back into the
    # main stream before the closing fence, so the timed region still contains all of its work.
    n_lanes = 1 if graph is not None else max(1, args.lanes)
    lanes = LaneExecutor(params, local_rank, n_lanes, head=head, first_extractor=ex)

    def gather(res):
        res["all_logits"] = gather_rows(res["logits"], BATCH * world, rank, world)

    def step():
        if graph is not None:
            graph.replay()
            return gather_rows(g_logits, BATCH * world, rank, world), g_res
        res = lanes.submit(pcm, offsets, N_SAMPLES, after=gather, pooled=True, frames=True)
        return res["all_logits"], res

    with torch.no_grad():
        lanes.fork()
        # untimed: first-use work of every lane (workspace allocation, kernel attributes) and the GPU's clock ramp are
        # kept out of short timed regions by a fixed number of extra launches ahead of the W requested warm-up steps
        for _ in range(args.warmup + (0 if args.no_preheat else 60)):
            step()
        lanes.join()
        fence()
        # one event pair around the whole timed loop: hundreds of live timing events (one pair per step) exhaust the
        # runtime's signal pool and serialise the launches (measured: 0.29 ms/step at 50 steps, 0.43-0.52 at 200)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        lanes.fork()
        for _ in range(args.steps):
            all_logits, res = step()
        lanes.join()
        ev1.record()
        fence()
        dt = time.perf_counter() - t0
        # for transparency: the same step strictly serially on one stream (outside the timed region, not `value`)
        serial_ms = None
        if graph is None and n_lanes > 1:
            sv0, sv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_serial = min(50, args.steps)
            sv0.record()
            for _ in range(n_serial):
                r1 = ex.extract_batch_device(pcm, offsets, N_SAMPLES, pooled=True, frames=True)
                gather_rows(head(r1["pooled"]), BATCH * world, rank, world)
            sv1.record()
            torch.cuda.synchronize(dev)
            serial_ms = sv0.elapsed_time(sv1) / n_serial
        # per-kernel durations: a few extra launches outside the timed region, events inside the library
        ex.kernel_timing(True)
        for _ in range(min(10, args.steps)):
            ex.extract_batch_device(pcm, offsets, N_SAMPLES, pooled=True, frames=True)
            per_kernel.append(ex.kernel_timing(True))
        ex.kernel_timing(False)
    assert int(res["status"].abs().sum()) == 0 and all_logits.shape == (BATCH * world, 3)
    assert bool(torch.isfinite(res["frames"]).all())
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kms = ev0.elapsed_time(ev1) / max(1, args.steps)  # GPU time per step on the launch stream

    if rank == 0:
        clips_per_s = BATCH * world * args.steps / dt
        algo_bytes = BATCH * (2 * N_SAMPLES + 4 * (ex.frame_rows * T + ex.pooled_dim))  # PCM read once + outputs written once
        k_stft = float(np.median([p["stft_mel"] for p in per_kernel]))
        k_mfcc = float(np.median([p["mfcc"] for p in per_kernel]))
        k_path = k_stft + k_mfcc  # the feature launch: both kernels back to back on one stream
        achieved = algo_bytes / (k_path * 1e-3) / 1e9
        logmel_bytes = BATCH * T * ((params.n_mels + 15) // 16 * 16) * 4  # the one HBM intermediate (written by one kernel, read by the other)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "clips/sec (3 s@16 kHz, 40-MFCC) end-to-end",
            "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 1024 clips x 48000 int16 samples per GPU resident in HBM, "
                                   "512-pt STFT hop 160, 64 Slaney mels, top_db 80, 40 MFCC + delta + delta2; outputs "
                                   "frame-level [1024,120,301] fp32 + pooled [1024,240] fp32, then MLP head -> logits "
                                   "[1024,3] and (N>1) RCCL all-gather of the logits",
                       "clips_per_gpu": BATCH, "samples_per_clip": N_SAMPLES, "n_fft": 512, "hop": 160,
                       "n_mels": 64, "n_mfcc": 40, "frames_per_clip": T, "untimed_preheat_launches": 0 if args.no_preheat else 60, "one_stream_ms_per_step": None if serial_ms is None else round(serial_ms, 4), "parallelism": f"clip-sharded x{world}", "launch": "hipGraph replay" if graph is not None else ("eager, batches alternate between %d streams" % n_lanes if n_lanes > 1 else "eager, one stream")},
            # dominant kernel = the STFT+mel kernel; its own algorithmic bytes are the PCM it reads plus the log-mel it
            # writes.  `path` prices the whole feature launch (both kernels) with the SURVEY 8d bytes (no intermediates).
            "roofline": {"bound": "hbm", "kernel": "dysf_stft_mel_kernel<256, i16, 256>",
                         "achieved": round((BATCH * 2 * N_SAMPLES + logmel_bytes) / (k_stft * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((BATCH * 2 * N_SAMPLES + logmel_bytes) / (k_stft * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel_ms": round(k_stft, 4),
                         "algorithmic_bytes_per_launch": BATCH * 2 * N_SAMPLES + logmel_bytes,
                         "second_kernel": {"kernel": "dysf_mfcc_kernel<512>", "kernel_ms": round(k_mfcc, 4),
                                           "algorithmic_bytes_per_launch": logmel_bytes + algo_bytes - BATCH * 2 * N_SAMPLES,
                                           "achieved": round((logmel_bytes + algo_bytes - BATCH * 2 * N_SAMPLES) / (k_mfcc * 1e-3) / 1e9, 1)},
                         "path": {"kernels": "stft_mel + mfcc", "ms": round(k_path, 4), "step_gpu_ms": round(kms, 4),
                                  "algorithmic_bytes_per_launch": algo_bytes, "achieved": round(achieved, 1),
                                  "frac": round(achieved / HBM_PEAK_GBS, 4),
                                  "clips_per_s": round(BATCH / (k_path * 1e-3), 1)}},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
