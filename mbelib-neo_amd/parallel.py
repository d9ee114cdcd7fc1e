"""Multi-GPU plumbing.  The path shards by stream: streams are independent, frames inside a
stream are sequential and stay on one wavefront, so N GPUs run N disjoint stream shards with
NO data-path collective.  The one collective is the start-up broadcast of the constant-table
blob from rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests), after which every rank
uploads the blob itself and the checksums are compared."""
import numpy as np


def shard_range(total_streams, world_size, rank):
    """Contiguous block partition of `total_streams`: returns (first_stream, count) for `rank`.
    Ranks [0, total % world) get one extra stream."""
    base, extra = divmod(int(total_streams), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def broadcast_tables(blob, device=None, force=False):
    """Rank 0 passes the blob bytes, the others None; returns the bytes on every rank.  Uses the
    default process group if one is initialised, otherwise returns the input unchanged (a group of one
    rank runs the collectives only when `force` is set: bench.py --force-dist on a 1-GPU box)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        if blob is None:
            raise ValueError("rank 0 must provide the table blob")
        return bytes(blob)
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([len(blob) if blob is not None else 0], dtype=torch.int64, device=dev)
    dist.broadcast(size, src=0)
    n = int(size.item())
    if blob is not None:
        buf = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    else:
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=0)
    return buf.cpu().numpy().tobytes()


def blob_checksum(blob):
    """FNV-1a-32 over the blob body, the same value mbx_init() verifies and mbx_table_checksum() returns."""
    h = 2166136261
    for b in blob[16:]:
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h
