"""Splitting one stream's frames over several GPUs (one process per GPU).

FLAC frames are independent of each other once the frame number is known (the reference encodes
them one after another in process_frame_, stream_encoder.c:2950, carrying nothing across frames but
the frame counter and, for the loose mid/side presets, the decision made at the start of each
period, :3074-3124).  So rank r encodes a contiguous range of frames, numbered from the range's
first frame, and the only thing ranks exchange is how many bytes each produced: an all-gather of
one integer per rank, from which every rank knows where its bytes go in the stream.
"""
from typing import List, Tuple


def loose_mid_side_period(sample_rate: int, blocksize: int) -> int:
    """Frames per loose mid/side decision (stream_encoder.c:877-879): round(0.4 s / frame), >= 1."""
    n = int(sample_rate * 0.4 / blocksize + 0.5)
    return n if n > 0 else 1


def frame_range(total_frames: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Frames [first, first+count) owned by `rank`.

    Ranges are contiguous, cover 0..total_frames, differ in size by at most `align`, and every
    range starts at a multiple of `align` (pass the loose mid/side period for presets that use it,
    so that each range starts where the reference would make a fresh decision).
    """
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank %d of %d" % (rank, world))
    if align < 1:
        raise ValueError("align must be positive")
    units = (total_frames + align - 1) // align
    base, extra = divmod(units, world)
    first_unit = rank * base + min(rank, extra)
    n_units = base + (1 if rank < extra else 0)
    first = min(first_unit * align, total_frames)
    last = min((first_unit + n_units) * align, total_frames)
    return first, last - first


def sample_range(total_samples: int, blocksize: int, world: int, rank: int, align: int = 1) -> Tuple[int, int, int]:
    """(first_frame, first_sample, sample_count) of `rank`'s share of a stream of total_samples."""
    total_frames = (total_samples + blocksize - 1) // blocksize
    first, count = frame_range(total_frames, world, rank, align)
    s0 = first * blocksize
    s1 = min((first + count) * blocksize, total_samples)
    return first, s0, max(0, s1 - s0)


def stream_offsets(local_bytes: int, group=None, device=None) -> Tuple[int, int, List[int]]:
    """All-gather every rank's byte total; returns (this rank's offset, stream total, all totals).

    Works on any initialised torch.distributed backend (RCCL on GPUs, gloo on CPUs); with no
    process group it is the single-rank identity.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 0, int(local_bytes), [int(local_bytes)]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(local_bytes)], dtype=torch.int64, device=device)
    totals = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(totals, mine, group=group)
    sizes = [int(x) for x in totals.tolist()]
    return sum(sizes[:rank]), sum(sizes), sizes


def gather_stream(local, group=None, dst: int = 0):
    """Put the ranks' frame bytes together into the one stream they are shards of.

    `local` is this rank's bytes: a 1-D uint8 tensor (on the GPU under RCCL, on the CPU under gloo),
    or a bytes object.  Every rank calls it; rank `dst` gets the whole stream (same kind as
    `local`), the others get None.  The sizes travel first (stream_offsets), then one all-gather of
    the shards padded to the largest one; rank dst cuts the padding out.  This is the output
    collection step of a sharded encode, the counterpart of the reference handing each frame to its
    write callback in order (stream_encoder.c:2379-2513): frame f of rank r lands after every frame
    of ranks 0..r-1.
    """
    import torch
    import torch.distributed as dist

    as_bytes = isinstance(local, (bytes, bytearray, memoryview))
    t = torch.frombuffer(bytearray(local), dtype=torch.uint8) if as_bytes else local
    if t.dtype != torch.uint8 or t.dim() != 1:
        raise ValueError("gather_stream wants a 1-D uint8 tensor or bytes")
    if not (dist.is_available() and dist.is_initialized()):
        return bytes(local) if as_bytes else t
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # a CPU-only backend (gloo) moves the bytes through host memory
    home = t.device
    via = home if dist.get_backend(group) == "nccl" else torch.device("cpu")
    _off, _total, sizes = stream_offsets(t.numel(), group=group, device=via)
    width = max(max(sizes), 1)
    padded = torch.zeros(width, dtype=torch.uint8, device=via)
    padded[:t.numel()] = t.to(via)
    everything = torch.empty(world * width, dtype=torch.uint8, device=via)
    dist.all_gather_into_tensor(everything, padded, group=group)
    if rank != dst:
        return None
    stream = torch.cat([everything[r * width:r * width + sizes[r]] for r in range(world)]).to(home)
    return stream.cpu().numpy().tobytes() if as_bytes else stream


def deal_streams(nstreams: int, world: int, rank: int) -> List[int]:
    """Batch decode is replica-style: stream i goes to rank i mod world, nothing is exchanged
    (each stream is a whole file with its own frame numbering, stream_decoder.c:1941-2113 keeps
    no state between streams).  Returns the stream indices `rank` decodes, in order."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank %d of %d" % (rank, world))
    return list(range(rank, nstreams, world))


def undeal(per_rank: List[list]) -> list:
    """Inverse of deal_streams for results: per_rank[r][k] is the result of stream r + k * world."""
    world = len(per_rank)
    total = sum(len(x) for x in per_rank)
    out = [None] * total
    for r, items in enumerate(per_rank):
        for k, item in enumerate(items):
            out[r + k * world] = item
    return out
