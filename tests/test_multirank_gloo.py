"""Two-process (gloo, CPU) test of the multi-GPU sharding scheme.

Each rank takes its contiguous range of frames (flac_amd.shard.frame_range), encodes it with the
CPU oracle numbered from the range's first frame, learns its byte offset from the all-gather of byte
totals (flac_amd.shard.stream_offsets) and writes its bytes there.  The assembled stream must equal
the golden bytes the reference wrote for the whole signal in one go: the N>1 path moves no sample
data between ranks and still produces the reference's stream.  On GPUs the same two functions run
over RCCL with the HIP encoder in place of the oracle.
"""
import os
import socket

import numpy as np
import pytest

import golden_util


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, case_name, out_path):
    import torch.distributed as dist

    import oracle_lib as O
    from flac_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = next(c for c in golden_util.load_index() if c["name"] == case_name)
        pcm, _meta, _frames = golden_util.load_case(case)
        p = O.preset(*golden_util.preset_args(case), **golden_util.field_overrides(case))
        align = shard.loose_mid_side_period(case["rate"], p.blocksize) if p.loose_mid_side_stereo else 1
        first, s0, n = shard.sample_range(pcm.shape[0], p.blocksize, world, rank, align)
        frames = O.encode_frames(p, pcm[s0:s0 + n], first_frame=first)[0] if n else []
        blob = b"".join(frames)
        offset, total, sizes = shard.stream_offsets(len(blob))
        assert len(sizes) == world and sizes[rank] == len(blob)
        if rank == 0:
            with open(out_path, "wb") as f:
                f.truncate(total)
        dist.barrier()
        with open(out_path, "r+b") as f:
            f.seek(offset)
            f.write(blob)
        dist.barrier()
        # the same through the product's own collection step: rank 0 receives the whole stream
        whole = shard.gather_stream(blob)
        if rank == 0:
            assert len(whole) == total
            with open(out_path + ".gathered", "wb") as f:
                f.write(whole)
        else:
            assert whole is None
        import torch
        as_tensor = shard.gather_stream(torch.frombuffer(bytearray(blob), dtype=torch.uint8) if blob else torch.zeros(0, dtype=torch.uint8), dst=1)
        if rank == 1:
            assert as_tensor.numpy().tobytes() == open(out_path + ".gathered", "rb").read()
        else:
            assert as_tensor is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name", ["s16_l8", "s16_l1_loose", "s16_l5_bs256"])
def test_two_ranks_assemble_the_reference_stream(tmp_path, case_name):
    import torch.multiprocessing as mp
    out = str(tmp_path / "stream.bin")
    mp.spawn(worker, args=(2, free_port(), case_name, out), nprocs=2, join=True)
    case = next(c for c in golden_util.load_index() if c["name"] == case_name)
    _pcm, _meta, frames = golden_util.load_case(case)
    assert open(out, "rb").read() == b"".join(frames)
    assert open(out + ".gathered", "rb").read() == b"".join(frames)


def deal_worker(rank, world, port, names, out_dir):
    """Batch decode, replica-style: each rank decodes the streams dealt to it (with the CPU oracle
    here, with the HIP decoder on GPUs), nothing is exchanged; the parent puts the results back in
    stream order with shard.undeal."""
    import torch.distributed as dist

    import oracle_lib as O
    from flac_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cases = {c["name"]: c for c in golden_util.load_index()}
        for k, i in enumerate(shard.deal_streams(len(names), world, rank)):
            case = cases[names[i]]
            pcm, _meta, frames = golden_util.load_case(case)
            n, got, _nf = O.decode_frames(b"".join(frames), case["channels"], case["bps"], case["rate"], pcm.shape[0])
            assert n == pcm.shape[0]
            np.save(os.path.join(out_dir, "r%d_%d.npy" % (rank, k)), got)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_decode_a_dealt_batch(tmp_path):
    import torch.multiprocessing as mp

    from flac_amd import shard
    names = ["s16_l8", "s16_l5_bs256", "s16_l1_loose", "s24_l8", "m16_l8"]
    have = {c["name"] for c in golden_util.load_index()}
    names = [n for n in names if n in have]
    assert len(names) >= 3
    mp.spawn(deal_worker, args=(2, free_port(), names, str(tmp_path)), nprocs=2, join=True)
    per_rank = [[np.load(str(tmp_path / ("r%d_%d.npy" % (r, k)))) for k in range(len(shard.deal_streams(len(names), 2, r)))] for r in range(2)]
    cases = {c["name"]: c for c in golden_util.load_index()}
    for name, got in zip(names, shard.undeal(per_rank)):
        pcm, _meta, _frames = golden_util.load_case(cases[name])
        assert np.array_equal(got, pcm), name


def test_frame_ranges_cover_and_align():
    from flac_amd import shard
    for total in (0, 1, 5, 6460, 6461, 51680):
        for world in (1, 2, 3, 8):
            for align in (1, 4, 15):
                got = [shard.frame_range(total, world, r, align) for r in range(world)]
                pos = 0
                for first, count in got:
                    if count:
                        assert first == pos and first % align == 0
                        pos = first + count
                assert pos == total
                counts = [c for _f, c in got]
                assert max(counts) - min(counts) < 2 * align
    assert shard.loose_mid_side_period(44100, 1152) == 15
    assert shard.loose_mid_side_period(44100, 4096) == 4
    assert shard.loose_mid_side_period(8000, 65535) == 1
    with pytest.raises(ValueError):
        shard.frame_range(10, 2, 2)


def test_stream_offsets_single_process():
    from flac_amd import shard
    assert shard.stream_offsets(123) == (0, 123, [123])
    assert shard.gather_stream(b"abc") == b"abc"


def test_deal_streams_round_robin():
    from flac_amd import shard
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 8):
            dealt = [shard.deal_streams(n, world, r) for r in range(world)]
            assert shard.undeal(dealt) == list(range(n))
            assert max(len(d) for d in dealt) - min(len(d) for d in dealt) <= 1
    with pytest.raises(ValueError):
        shard.deal_streams(4, 2, 2)


def plan_worker(rank, world, port, seconds, out_dir):
    """What `bench.py --config C3 --gpus 2` and `--config C5 --gpus 2` do per rank, with the CPU oracle in the
    HIP library's place: the plans come from flac_amd.workloads.plan, the same call bench.py makes."""
    import torch.distributed as dist

    import oracle_lib as O
    from flac_amd import shard
    from flac_amd import workloads as W

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # C3, strong scaling: ONE stream cut into contiguous frame ranges
        cfg = W.CONFIGS["C3"]
        p = O.preset(cfg["level"], cfg["channels"], cfg["bps"], cfg["rate"], **cfg["over"])
        pl = W.plan("C3", "strong", world, rank, p.blocksize, seconds=seconds)
        pcm = W.shard_pcm("C3", pl)
        assert pcm.shape[0] == pl["wide"]
        frames, _ = O.encode_frames(p, pcm, first_frame=pl["first_frame"])
        whole = shard.gather_stream(b"".join(frames))
        if rank == 0:
            with open(os.path.join(out_dir, "c3.bin"), "wb") as f:
                f.write(whole)
        # C3, weak scaling: the stream is world times as long, rank r's range is signal r
        plw = W.plan("C3", "weak", world, rank, p.blocksize, seconds=seconds)
        assert plw["first_frame"] == rank * -(-W.wide_samples("C3", seconds) // p.blocksize) and plw["signal"] == rank
        assert plw["is_last"] == (rank == world - 1)
        # C5: the batch dealt round-robin, nothing exchanged
        c5 = W.CONFIGS["C5"]
        p5 = O.preset(c5["level"], c5["channels"], c5["bps"], c5["rate"])
        pl5 = W.plan("C5", "strong", world, rank, p5.blocksize, seconds=0.25, streams=7)
        assert pl5["mine"] == shard.deal_streams(7, world, rank) and pl5["job_samples"] == 7 * pl5["stream_wide"] * 2
        for k, i in enumerate(pl5["mine"]):
            src = W.pcm("C5", stream_id=i % pl5["distinct"], wide=pl5["stream_wide"], stream_wide=pl5["stream_wide"])
            fr, _ = O.encode_frames(p5, src)
            n, got, _nf = O.decode_frames(b"".join(fr), 2, 16, c5["rate"], src.shape[0])
            assert n == src.shape[0] and np.array_equal(got, src)
            np.save(os.path.join(out_dir, "c5_r%d_%d.npy" % (rank, k)), got)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_run_the_bench_plans_for_c3_and_c5(tmp_path):
    import torch.multiprocessing as mp

    import oracle_lib as O
    from flac_amd import shard
    from flac_amd import workloads as W
    seconds = 0.2   # 19 200 wide samples of 24-bit / 96 kHz: five blocks, the last one short
    mp.spawn(plan_worker, args=(2, free_port(), seconds, str(tmp_path)), nprocs=2, join=True)
    cfg = W.CONFIGS["C3"]
    p = O.preset(cfg["level"], cfg["channels"], cfg["bps"], cfg["rate"], **cfg["over"])
    want, _ = O.encode_frames(p, W.pcm("C3", seconds=seconds, stream_wide=W.wide_samples("C3", seconds)))
    assert len(want) == 5 and open(str(tmp_path / "c3.bin"), "rb").read() == b"".join(want)
    per_rank = [[np.load(str(tmp_path / ("c5_r%d_%d.npy" % (r, k)))) for k in range(len(shard.deal_streams(7, 2, r)))] for r in range(2)]
    for i, got in enumerate(shard.undeal(per_rank)):
        assert np.array_equal(got, W.pcm("C5", stream_id=i, wide=got.shape[0], stream_wide=got.shape[0]))


def test_bench_plans_cover_the_job():
    from flac_amd import workloads as W
    for name, bs in (("C2", 4096), ("C3", 4096), ("C4", 4096)):
        for scaling in ("weak", "strong"):
            for world in (1, 2, 4, 8):
                plans = [W.plan(name, scaling, world, r, bs) for r in range(world)]
                pos = 0
                for pl in plans:
                    assert pl["first_sample"] == pos and pl["first_frame"] * bs == pos
                    pos += pl["wide"]
                assert pos == plans[0]["stream_wide"] and plans[-1]["is_last"] and not any(pl["is_last"] for pl in plans[:-1])
                assert plans[0]["job_samples"] == pos * W.CONFIGS[name]["channels"]
                one = W.wide_samples(name)
                assert pos == one if scaling == "strong" else pos >= world * (one - bs)
    for world in (1, 2, 8):
        mine = [W.plan("C5", "strong", world, r, 4096)["mine"] for r in range(world)]
        assert sorted(i for m in mine for i in m) == list(range(10000))
