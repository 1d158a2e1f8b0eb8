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
