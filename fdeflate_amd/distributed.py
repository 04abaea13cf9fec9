"""Multi-GPU layer: one process per GPU, streams sharded by contiguous index range, no
data-path collective; the only exchange is the final gather of per-stream metadata (and,
optionally, payload) over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo"
in the CPU tests).  SURVEY.md 8(e)."""
import torch
import torch.distributed as dist


def shard_range(n_streams, rank, world_size):
    """Contiguous [lo, hi) of the streams owned by `rank` (ceil split, last ranks may be short)."""
    per = (n_streams + world_size - 1) // world_size
    lo = min(rank * per, n_streams)
    hi = min(lo + per, n_streams)
    return lo, hi


def gather_metadata(status, out_len, adler, group=None):
    """All-gathers the fixed-size per-stream results {status, out_len, adler} (12 B per stream).
    Every rank must pass the same shard size (pad the last shard).  Returns [world, 3, n_local]."""
    meta = torch.stack([status.to(torch.int32), out_len.to(torch.int32), adler.to(torch.int32)])
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(meta.shape), dtype=meta.dtype, device=meta.device)
    if dist.get_backend(group) == "nccl":  # RCCL: one flat all-gather over xGMI
        dist.all_gather_into_tensor(out, meta.contiguous(), group=group)
    else:  # gloo (CPU tests)
        dist.all_gather(list(out.unbind(0)), meta.contiguous(), group=group)
    return out


def gather_payload(out, group=None, into=None):
    """Optional payload gather (link-bound over xGMI: timed separately by the bench).  `into`
    reuses a [world, ...] buffer from an earlier call."""
    world = dist.get_world_size(group)
    full = into if into is not None else torch.empty((world,) + tuple(out.shape), dtype=out.dtype, device=out.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(full, out.contiguous(), group=group)
    else:
        dist.all_gather(list(full.unbind(0)), out.contiguous(), group=group)
    return full
