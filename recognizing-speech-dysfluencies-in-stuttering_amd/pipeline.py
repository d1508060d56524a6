"""Batch driver: the reference's sequential per-file loop (pipeline1.py:447-456) as one sharded, GPU-resident
pass -- PCM shard -> HIP feature kernel -> classifier head -> logits -- with one process per GPU.

Clips are independent (the clip-global max and the tuning estimate are intra-clip), so ranks take contiguous
blocks of the clip index space and the only collective is one all-gather of the per-clip logits
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests)."""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r owns clips [r*ceil(n/W), min(n, (r+1)*ceil(n/W)))  (SURVEY.md section 8e)."""
    per = -(-n // world) if world > 0 else n
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_by_samples(lengths: np.ndarray, world: int) -> np.ndarray:
    """Variable-length clips: contiguous shards balanced by cumulative sample count.  Returns [world+1] bounds."""
    lengths = np.asarray(lengths, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(lengths)])
    targets = csum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(csum, targets, side="left")
    return np.concatenate([[0], cuts, [len(lengths)]]).astype(np.int64)


def gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather [n_local, d] row blocks of a shard_range partition into [n_total, d] on every rank."""
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return local
    per = -(-n_total // world)
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_total]


def gather_rows_by_bounds(local: torch.Tensor, bounds, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather the row blocks of a partition with unequal shards (``shard_by_samples`` bounds [world + 1]) into
    [bounds[-1], d] on every rank: shards are padded to the largest one for the single collective."""
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return local
    sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
    assert local.shape[0] == sizes[rank]
    per = max(sizes)
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * per: r * per + sizes[r]] for r in range(world)], dim=0)


class DysfluencyPipeline:
    """extract -> (pooled | frame-level) features -> head -> logits, everything resident on one GPU."""

    def __init__(self, extractor, head: Optional[torch.nn.Module] = None, head_input: str = "pooled"):
        assert head_input in ("pooled", "frames")
        self.extractor = extractor
        self.head = head
        self.head_input = head_input

    @torch.no_grad()
    def run_resident(self, pcm: torch.Tensor, offsets: torch.Tensor, max_samples: int):
        """pcm/offsets already in HBM.  Returns dict with the features and, if a head is set, ``logits``."""
        want_frames = self.head_input == "frames"
        res = self.extractor.extract_batch_device(pcm, offsets, max_samples, pooled=not want_frames, frames=want_frames)
        if self.head is not None:
            if want_frames:
                res["logits"] = self.head(res["frames"])
            else:
                x = res["pooled"]
                pad = self.head.n_features_in_ - x.shape[1] if hasattr(self.head, "n_features_in_") else 0
                if pad > 0:  # the five always-zero text features (pipeline1.py:243-244)
                    x = torch.cat([x, x.new_zeros((x.shape[0], pad))], dim=1)
                res["logits"] = self.head(x)
        return res

    @torch.no_grad()
    def run_sharded(self, pcm: torch.Tensor, offsets: torch.Tensor, max_samples: int, n_total: int, rank: int, world: int):
        """This rank's shard (already resident) -> logits, then the single all-gather of [n_local, n_classes]."""
        res = self.run_resident(pcm, offsets, max_samples)
        res["all_logits"] = gather_rows(res["logits"], n_total, rank, world)
        return res


class LaneExecutor:
    """Throughput driver for a stream of independent batches: batch i runs on HIP stream ``i % n_lanes`` with that
    lane's own ``FeatureExtractor`` handle (a handle owns one log-mel workspace, so lanes never share one).  The STFT
    kernel of one batch then overlaps the MFCC kernel, the head and the logits gather of the previous batch -- both
    feature kernels are latency bound with idle pipes, so this is real throughput (MI355X, 1024-clip batches:
    257 -> 225 us per batch with two lanes).  Tensors returned by ``submit`` belong to their lane's stream: call
    ``join()`` (or synchronise) before using them on the current stream."""

    def __init__(self, params, device: int = 0, n_lanes: int = 3, head=None, first_extractor=None):
        from .features import FeatureExtractor
        dev = torch.device("cuda", device)
        self.device, self.head = dev, head
        self.main = torch.cuda.current_stream(dev)
        n_lanes = max(1, int(n_lanes))
        self.extractors = [first_extractor if (i == 0 and first_extractor is not None) else FeatureExtractor(params, device)
                           for i in range(n_lanes)]
        self.streams = [self.main] if n_lanes == 1 else [torch.cuda.Stream(dev) for _ in range(n_lanes)]
        self._i = 0

    @property
    def n_lanes(self) -> int:
        return len(self.streams)

    def fork(self):
        """Lanes start after everything queued so far on the main stream."""
        for st in self.streams:
            if st is not self.main:
                st.wait_stream(self.main)

    def join(self):
        """The main stream waits for every lane."""
        for st in self.streams:
            if st is not self.main:
                self.main.wait_stream(st)

    @torch.no_grad()
    def submit(self, pcm: torch.Tensor, offsets: torch.Tensor, max_samples: int, after=None, **kw):
        """Queue one batch on the next lane.  ``after(res)`` (optional) runs inside the lane's stream context, e.g. to
        queue a collective on the logits.  Returns the result dict (+ ``logits`` when a head is set)."""
        k = self._i % len(self.streams)
        self._i += 1
        with torch.cuda.stream(self.streams[k]):
            res = self.extractors[k].extract_batch_device(pcm, offsets, max_samples, **kw)
            if self.head is not None:
                res["logits"] = self.head(res["pooled"])
            if after is not None:
                after(res)
        return res

    @torch.no_grad()
    def extract_large(self, pcm: torch.Tensor, offsets: torch.Tensor, max_samples: int, chunk_clips: int = 2048,
                      pooled: bool = True, frames: bool = False, frame_stride=None, tuning: bool = False):
        """One big resident batch (BASELINE configs[3]: tens of thousands of clips per GPU) cut into chunks of
        ``chunk_clips`` that alternate between the lanes and write into row slices of ONE set of output tensors, so that
        chunk i+1's STFT kernel overlaps chunk i's MFCC kernel.  Same numbers as a single call; forks from and joins
        into the current stream, so the result is ready for it on return (asynchronously)."""
        n = offsets.numel() - 1
        ex0 = self.extractors[0]
        dev = pcm.device
        fs = int(frame_stride or ex0.params.n_frames(max_samples))
        out = {"status": torch.empty(n, dtype=torch.int32, device=dev)}
        if pooled:
            out["pooled"] = torch.empty((n, ex0.pooled_dim), dtype=torch.float32, device=dev)
        if frames:
            out["frames"] = torch.empty((n, ex0.frame_rows, fs), dtype=torch.float32, device=dev)
        if tuning:
            out["tuning"] = torch.empty(n, dtype=torch.int32, device=dev)
        self.fork()
        for a in range(0, n, chunk_clips):
            b = min(n, a + chunk_clips)
            k = self._i % len(self.streams)
            self._i += 1
            with torch.cuda.stream(self.streams[k]):
                for t in out.values():
                    t.record_stream(self.streams[k])
                self.extractors[k].extract_batch_device(pcm, offsets[a:b + 1], max_samples, frame_stride=fs,
                                                        out={key: t[a:b] for key, t in out.items()})
        self.join()
        return out
