"""Multi-GPU: the batch dimension shards trivially (no environment reads another's state — the reference
vmaps over axis 0 everywhere, core_env.py:566,612), one process per GPU, zero communication while stepping.
The only collective is an all-gather of observations (RCCL over xGMI when the backend is "nccl") for a
consumer that needs the whole batch on every rank; it runs on its own stream so it overlaps the next chunk.
The reference has no counterpart (single device)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [start, stop) of the global batch owned by `rank` (remainder spread over the first ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(global_batch: int, world_size: int):
    return [shard_range(global_batch, world_size, r)[1] - shard_range(global_batch, world_size, r)[0]
            for r in range(world_size)]


def _slice_leaf(v, global_batch, lo, hi):
    import numpy as np

    if isinstance(v, (torch.Tensor, np.ndarray)) and v.ndim >= 1 and v.shape[0] == global_batch:
        return v[lo:hi]
    return v


def make_sharded_env(env_type, global_batch_size: int, group: Optional[dist.ProcessGroup] = None, rank: Optional[int] = None,
                     world_size: Optional[int] = None, **env_kwargs):
    """Build THIS rank's environment over its contiguous slice of a global batch: `batch_size` becomes the shard size
    and every property leaf given as a [global_batch_size] array (static_params values, MinMaxNormalization.min / .max) is
    sliced to the shard. Returns (env, (start, stop)). `env_type` is an EnvironmentRegistry member."""
    from .utils import MinMaxNormalization

    if world_size is None:
        world_size = dist.get_world_size(group)
    if rank is None:
        rank = dist.get_rank(group)
    lo, hi = shard_range(global_batch_size, world_size, rank)
    kw = dict(env_kwargs)
    if kw.get("static_params"):
        kw["static_params"] = {k: _slice_leaf(v, global_batch_size, lo, hi) for k, v in kw["static_params"].items()}
    for name in ("physical_normalizations", "action_normalizations"):
        if kw.get(name):
            kw[name] = {k: MinMaxNormalization(min=_slice_leaf(n.min, global_batch_size, lo, hi),
                                               max=_slice_leaf(n.max, global_batch_size, lo, hi)) for k, n in kw[name].items()}
    kw["batch_size"] = hi - lo
    return env_type.make(**kw), (lo, hi)


class ObservationGatherer:
    """All-gather of per-rank observation shards [B_r, ...] into the global [B, ...] array on every rank.

    ``start`` enqueues the collective on a side stream (after the producer stream's work) and returns at once;
    ``wait`` makes the current stream of the same device wait for it. The collective is chosen once, up front:
    even shards use ``all_gather_into_tensor`` (one ncclAllGather with the "nccl" backend = RCCL; gloo implements it too),
    ragged shards use the list form on shards padded to the largest one. Errors of the collective propagate — there
    is no fallback from one form to the other.
    """

    def __init__(self, global_batch: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.global_batch = global_batch
        self.sizes = shard_sizes(global_batch, self.world)
        self.even = len(set(self.sizes)) == 1
        self.collective = "all_gather_into_tensor" if self.even else "all_gather(list, padded)"
        self._stream = None
        self._event = None
        self._device = None
        self._held = None  # the shard being gathered: kept referenced until wait()

    def _side_stream(self, device):
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _gather(self, local: torch.Tensor, out: torch.Tensor):
        if self.even:
            dist.all_gather_into_tensor(out, local, group=self.group)
            return
        pad = max(self.sizes)
        buf = local
        if local.shape[0] != pad:
            buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
            buf[: local.shape[0]] = local
        parts = [local.new_empty((pad,) + tuple(local.shape[1:])) for _ in range(self.world)]
        dist.all_gather(parts, buf.contiguous(), group=self.group)
        off = 0
        for r, n in enumerate(self.sizes):
            out[off:off + n] = parts[r][:n]
            off += n

    def start(self, local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert local.shape[0] == self.sizes[self.rank], "local shard has the wrong batch size"
        local = local.contiguous()
        if out is None:
            out = local.new_empty((self.global_batch,) + tuple(local.shape[1:]))
        if local.is_cuda:
            self._device = local.device
            side = self._side_stream(local.device)
            side.wait_stream(torch.cuda.current_stream(local.device))
            with torch.cuda.stream(side):
                self._gather(local, out)
                local.record_stream(side)
                out.record_stream(side)
            self._event = side.record_event()
            # record_stream only tells TORCH's allocator; the environments' own output pools (vmap_step slots, trajectory sets)
            # recycle a buffer when nothing refers to it and the stream is the same — so the shard stays referenced here until
            # wait() has ordered the consumer's stream behind the collective
            self._held = local
        else:
            self._gather(local, out)
        return out

    def wait(self):
        if self._event is not None:
            torch.cuda.current_stream(self._device).wait_event(self._event)
            self._event = None
        self._held = None
