#!/usr/bin/env python3
"""Benchmark of the FLAC hot path on MI355X: `flac -8` frame encode (and decode) of synthetic PCM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the whole encode pipeline over one batch: a 600 s 16-bit / 44.1 kHz stereo
stream (26.46 M wide samples, 6460 frames of 4096) that already lies in HBM as interleaved int16.
With N > 1 every rank owns the frames [rank*6460, (rank+1)*6460) of an N-times longer stream
(weak scaling); the only exchange is an all-gather of per-rank byte totals for the stream offsets.

Rank 0 prints one JSON line; see DESIGN.md for what each field is computed from.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SECONDS = 600.0
RATE = 44100
CHANNELS = 2
BPS = 16
LEVEL = 8
HBM_PEAK_GBS = 8000.0


def make_pcm(seed, seconds):
    from flac_amd.synth import synth_pcm
    return synth_pcm(seconds=seconds, sample_rate=RATE, channels=CHANNELS, bits_per_sample=BPS, seed=seed)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the real reference libFLAC (oracle/_ref, compiled from the reference's sources in
# the build container) when it is present, else this repo's CPU restatement; one process per core.
# ---------------------------------------------------------------------------------------------
def _cpu_worker(args):
    kind, seconds, seed = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    pcm = make_pcm(seed, seconds)
    t0 = time.perf_counter()
    if kind == "reference":
        import libflac_ctypes as L
        ref = L.load(os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so"))
        _meta, frames, ok, _ = L.encode(ref, pcm, BPS, RATE, level=LEVEL, chunk=1 << 20)
        assert ok
    else:
        import oracle_lib as O
        p = O.preset(LEVEL, CHANNELS, BPS, RATE)
        frames, _ = O.encode_frames(p, pcm)
    dt = time.perf_counter() - t0
    import hashlib
    return dt, sum(len(f) for f in frames), pcm.shape[0] * CHANNELS, hashlib.sha256(b"".join(frames)).hexdigest()


def cpu_baseline(seconds=SECONDS):
    import multiprocessing as mp
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")) else "port"
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(kind, seconds, 1234 + i) for i in range(cores)])
    wall_all = time.perf_counter() - t0
    slowest = max(r[0] for r in res)
    samples = sum(r[2] for r in res)
    return {
        "value": samples / slowest / 1e6,
        "unit": "MSamples/s",
        "cores": cores,
        "kind": kind,
        "sample": "%d processes x %.0f s of the same synthetic 16-bit/44.1kHz stereo signal, flac -8, encode only; "
                  "throughput = all samples / slowest process (%.2f s; pool wall %.2f s)" % (cores, seconds, slowest, wall_all),
        "per_core": samples / cores / slowest / 1e6,
    }


def launch_ranks(n):
    """Run this script as n ranks under torch.distributed.run (one per GPU, RCCL rendezvous on
    127.0.0.1) and return the launcher's exit code; rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=SECONDS, help="stream length per GPU (default: the 600 s workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-stream-api", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU) as children
        # of this process, which has not touched the GPU and never will
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    import flac_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (torch.cuda.is_available() is False) [rank %d of %d]" % (rank, world))
    # FLAC_AMD_BENCH_REHEARSE=1: every rank on GPU 0 and the collectives over gloo through host
    # memory - a rehearsal of the N-rank path on a one-GPU box, not a measurement
    rehearse = bool(os.environ.get("FLAC_AMD_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs GPU %d, this node shows %d" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    coll_dev = torch.device("cpu") if rehearse else dev

    # ---- workload: this rank's shard of the stream, resident in HBM as interleaved int16 ----
    # weak scaling: the job is ONE stream of world x 6460 frames cut into contiguous frame ranges;
    # every block is full except the stream's last one (the 600 s workload ends 160 samples short of
    # a block), so the ranks' bytes put end to end are a valid stream
    params = flac_amd.preset(LEVEL, CHANNELS, BPS, RATE)
    bs = params.blocksize
    from flac_amd import shard
    wide_one = int(round(args.seconds * RATE))
    frames_one = (wide_one + bs - 1) // bs
    stream_wide = (world - 1) * frames_one * bs + wide_one
    first_frame, first_sample, wide = shard.sample_range(stream_wide, bs, world, rank)
    nframes = (wide + bs - 1) // bs
    assert nframes == frames_one and first_frame == rank * frames_one
    pcm32 = make_pcm(1234 + 1000 * rank, (frames_one * bs + 1) / RATE)[:wide]
    assert pcm32.shape[0] == wide
    is_last = rank == world - 1
    pcm_dev = torch.from_numpy(pcm32.astype(np.int16)).to(dev)
    enc = flac_amd.Encoder(params, device=local_rank, max_frames=nframes)
    out_cap = enc.max_output_bytes(nframes)
    out_dev = torch.empty(out_cap, dtype=torch.uint8, device=dev)
    fbytes_dev = torch.zeros(nframes, dtype=torch.int32, device=dev)
    offs_dev = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
    # the exchange runs beside the next step's kernels (nothing on the device waits for the offsets: they are
    # needed when the shards are written out), so its buffers rotate: step k uses pair k & 1, whose previous
    # use, two steps back, is waited for first
    totals = [torch.zeros(world, dtype=torch.int64, device=coll_dev) for _ in range(2)]
    mine = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(2)]
    pending = []
    stream = torch.cuda.current_stream().cuda_stream

    def step(k):
        enc.encode_device(pcm_dev.data_ptr(), 2, wide, out_dev.data_ptr(), out_cap, fbytes_dev.data_ptr(),
                          offs_dev.data_ptr(), first_frame=first_frame, is_last=is_last, stream=stream)
        if world > 1:
            # stream offsets of the shards: all-gather of the per-rank byte totals over RCCL
            if len(pending) >= 2:
                pending.pop(0).wait()
            mine[k & 1].copy_(offs_dev[nframes:nframes + 1], non_blocking=True)
            pending.append(dist.all_gather_into_tensor(totals[k & 1], mine[k & 1], async_op=True))

    def drain():
        while pending:
            pending.pop(0).wait()

    for k in range(args.warmup):
        step(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    status_checked = not os.environ.get("FLAC_AMD_BENCH_NOCHECK")  # kernel experiments whose output is knowingly wrong
    if status_checked:
        enc.status()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    samples_per_step = stream_wide * CHANNELS
    value = samples_per_step * args.steps / elapsed / 1e6
    flac_bytes = int(offs_dev[nframes].item())

    # ---- per-kernel durations, HIP events on the launch stream, separate passes ----
    enc.set_profiling(True)
    acc = {}
    nprof = max(3, min(10, args.steps))
    for _ in range(nprof):
        enc.encode_device(pcm_dev.data_ptr(), 2, wide, out_dev.data_ptr(), out_cap, fbytes_dev.data_ptr(),
                          offs_dev.data_ptr(), first_frame=first_frame, is_last=is_last, stream=stream)
        torch.cuda.synchronize()
        for k, v in enc.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / nprof
    enc.set_profiling(False)
    dominant = max(acc, key=acc.get)
    alg_bytes = wide * CHANNELS * (BPS // 8) + flac_bytes  # PCM read once at container width + frames written once
    achieved = alg_bytes / (acc[dominant] * 1e-3) / 1e9
    pipeline_gbs = alg_bytes / (sum(acc.values()) * 1e-3) / 1e9

    result = {
        "metric": "encode/decode MSamples/s per node at -8 (1/2/4/8 GPU); bytes vs reference",
        "value": value,
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        **({"rehearsal": "all ranks on GPU 0, gloo collectives through host memory: not a measurement"} if rehearse else {}),
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "status_checked": bool(status_checked),
        "config": {
            "workload": "C2: flac -8 (max LPC order 12, tukey(0.5), Rice partition order 0-6), 16-bit/44.1kHz stereo, "
                        "blocksize 4096, %.0f s (%d frames) per GPU, PCM resident in HBM as int16" % (args.seconds, nframes),
            "what": "encode",
            "frames_per_gpu": nframes,
            "samples_per_step": samples_per_step,
            "flac_bytes_per_gpu": flac_bytes,
            "compression_ratio": flac_bytes / (wide * CHANNELS * 2),
            "sharding": "contiguous frame ranges per GPU; all-gather of byte totals" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dominant,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": acc,
            "all_kernels_achieved": pipeline_gbs,
            "all_kernels_frac": pipeline_gbs / HBM_PEAK_GBS,
        },
    }

    # ---- decode of the same stream (device resident), reported beside the encode number ----
    if not args.no_decode and hasattr(flac_amd.lib(), "flacamd_decode_streams"):
        try:
            result["decode"] = bench_decode(torch, flac_amd, dev, local_rank, world, dist, out_dev, flac_bytes, wide, bs,
                                            nframes, args, pcm_dev, stream_wide)
        except Exception as exc:  # the encode line must still be printed
            result["decode"] = {"error": str(exc)}

    # ---- N > 1: the shards put together into the one stream they belong to (output collection,
    # outside the timed region), which rank 0 then decodes whole: every frame of every rank must
    # be found, in order, with a good CRC-16 ----
    if world > 1:
        whole = shard.gather_stream(out_dev[:flac_bytes])
        if rank == 0:
            try:
                dec = flac_amd.Decoder(CHANNELS, BPS, bs, device=local_rank, max_input_bytes=whole.numel() + 64, max_frames=nframes * world + 16)
                back = torch.empty(stream_wide * CHANNELS, dtype=torch.int16, device=dev)
                dec.decode_device(whole.data_ptr(), whole.numel(), [(0, whole.numel(), 0, stream_wide, 0)], back.data_ptr(), 2, stream_wide)
                frames, got_wide, bad = dec.result()
                dec.close()
                result["assembled_stream"] = {"bytes": int(whole.numel()), "frames": frames, "wide_samples": got_wide, "bad_frames": bad,
                                              "rank0_shard_bit_exact": bool(torch.equal(back[:wide * CHANNELS].view(-1, CHANNELS), pcm_dev)),
                                              "ok": frames == nframes * world and bad == 0 and got_wide == stream_wide}
                del back
            except Exception as exc:
                result["assembled_stream"] = {"error": str(exc)}
        del whole

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    # the gfx950 note prescribes, + WRITE_SIZE); collected by separate runs of this same command
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as f:
            tr = json.load(f)
        if dominant in tr.get("kernels", {}):
            result["roofline"]["traffic"] = tr["kernels"][dominant]["bytes_per_launch"]
            result["roofline"]["traffic_source"] = tr.get("source")
    except (OSError, ValueError):
        pass

    if rank == 0 and world == 1 and not args.no_stream_api:
        try:
            bench_stream_api(flac_amd, pcm32[:1 << 20], do_md5=False)  # first use of the stream layer's buffers and threads
            result["end_to_end"] = bench_stream_api(flac_amd, pcm32)
            result["end_to_end"]["without_md5"] = bench_stream_api(flac_amd, pcm32, do_md5=False)
            result["end_to_end"]["without_md5_null_sink"] = bench_stream_api(flac_amd, pcm32, do_md5=False, sink="null")
            result["end_to_end"]["four_encoders_four_threads"] = bench_stream_api(flac_amd, pcm32, do_md5=True, threads=4)
            result["end_to_end"]["four_encoders_four_threads_without_md5"] = bench_stream_api(flac_amd, pcm32, do_md5=False, threads=4)
        except Exception as exc:
            result["end_to_end"] = {"error": str(exc)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline()
        # the same bounded sample through the GPU path: output bytes vs the CPU run
        sample = make_pcm(1234, 60.0)
        genc = flac_amd.Encoder(params, device=local_rank, max_frames=(sample.shape[0] + bs - 1) // bs)
        gframes, _ = genc.encode_host(sample)
        genc.close()
        kind, seconds, seed = cb["kind"], 60.0, 1234
        import hashlib
        _dt, cpu_bytes, _n, cpu_sha = _cpu_worker((kind, seconds, seed))
        gpu_bytes = sum(len(f) for f in gframes)
        result["cpu_baseline"] = cb
        result["bytes_vs_reference"] = {"gpu_bytes": gpu_bytes, "cpu_bytes": cpu_bytes,
                                        "delta_percent": 100.0 * (gpu_bytes - cpu_bytes) / cpu_bytes,
                                        "identical": hashlib.sha256(b"".join(gframes)).hexdigest() == cpu_sha,
                                        "what": "frame bytes of a 60 s sample of the workload, GPU path against the CPU baseline's "
                                                "encoder (sha256 of all frames); the whole 600 s stream is compared byte for byte in "
                                                "tests/test_gpu_reference_accepts.py"}
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def bench_stream_api(flac_amd, pcm32, do_md5=True, threads=1, sink="tmpfs"):
    """The same stream through the libFLAC-compatible API from host memory: int32 samples in through
    FLAC__stream_encoder_process_interleaved in 1 Mi-sample calls, MD5, PCIe both ways, the library's
    own FILE write / seek / tell callbacks (FLAC__stream_encoder_init_file, what the `flac` tool uses)
    onto a tmpfs file.  Never the headline value.  do_md5=False switches the digest off
    (FLAC__stream_encoder_set_do_md5, stream_encoder.c:1414), which shows what the copies and the GPU
    work take on their own.  threads > 1: that many encoder objects on as many host threads, each
    coding its own copy of the stream on the one GPU (aggregate rate)."""
    import ctypes as C
    import tempfile
    import threading
    lib = flac_amd.lib()
    wide, ch = pcm32.shape
    data = np.ascontiguousarray(pcm32, dtype=np.int32)
    lib.FLAC__stream_encoder_new.restype = C.c_void_p
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    results = [None] * threads
    encs = []
    for t in range(threads):
        enc = C.c_void_p(lib.FLAC__stream_encoder_new())
        for name, v in (("channels", ch), ("bits_per_sample", BPS), ("sample_rate", RATE), ("compression_level", LEVEL)):
            getattr(lib, "FLAC__stream_encoder_set_" + name)(enc, C.c_uint(v))
        lib.FLAC__stream_encoder_set_total_samples_estimate(enc, C.c_uint64(wide))
        if not do_md5:
            lib.FLAC__stream_encoder_set_do_md5(enc, C.c_int(0))
        encs.append(enc)

    def run(t):
        enc = encs[t]
        path = "/dev/null" if sink == "null" else os.path.join(tmpdir, "flac_amd_bench_%d_%d.flac" % (os.getpid(), t))
        t0 = time.perf_counter()
        st = lib.FLAC__stream_encoder_init_file(enc, path.encode(), None, None)
        if st != 0:
            results[t] = RuntimeError("init_file status %d" % st)
            return
        t1 = time.perf_counter()
        chunk = 1 << 20
        ok = True
        for pos in range(0, wide, chunk):
            n = min(chunk, wide - pos)
            if not lib.FLAC__stream_encoder_process_interleaved(enc, data[pos:pos + n].ctypes.data_as(C.c_void_p), C.c_uint(n)):
                ok = False
                break
        ok = bool(lib.FLAC__stream_encoder_finish(enc)) and ok
        t2 = time.perf_counter()
        size = os.path.getsize(path) if os.path.exists(path) else 0
        if sink != "null":
            try:
                os.remove(path)
            except OSError:
                pass
        results[t] = (t1 - t0, t2 - t1, size) if ok else RuntimeError("encode failed")

    t_start = time.perf_counter()
    if threads == 1:
        run(0)
    else:
        ths = [threading.Thread(target=run, args=(t,)) for t in range(threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    t_all = time.perf_counter() - t_start
    for enc in encs:
        lib.FLAC__stream_encoder_delete(enc)
    for r in results:
        if isinstance(r, Exception):
            raise r
    init_s = max(r[0] for r in results)
    proc_s = max(r[1] for r in results)
    out = {"value": threads * wide * ch / (t_all - init_s if threads > 1 else proc_s) / 1e6, "unit": "MSamples/s", "seconds": proc_s, "init_seconds": init_s,
           "file_bytes": results[0][2], "threads": threads,
           "what": "FLAC__stream_encoder_init_file + process_interleaved + finish of the same 600 s stream from host int32: %shost<->device "
                   "copies and the library's FILE callbacks onto %s included; %d host thread(s) calling, %d encoder object(s) on one GPU"
                   % ("MD5, " if do_md5 else "no MD5, ", "/dev/null" if sink == "null" else "a tmpfs file", threads, threads)}
    return out


def bench_decode(torch, flac_amd, dev, local_rank, world, dist, flac_dev, flac_bytes, wide, bs, nframes, args, pcm_dev, stream_wide):
    dec = flac_amd.Decoder(CHANNELS, BPS, bs, device=local_rank, max_input_bytes=flac_bytes + 64, max_frames=nframes + 16)
    out = torch.zeros(wide * CHANNELS, dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # the encoder numbered this rank's frames from rank*nframes; tell the decoder where sample 0 is
    first_sample = int(os.environ.get("RANK", "0")) * nframes * bs
    streams = [(0, flac_bytes, 0, wide, first_sample)]

    def step():
        dec.decode_device(flac_dev.data_ptr(), flac_bytes, streams, out.data_ptr(), 2, wide, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    frames, nwide, bad = dec.result()
    exact = bool(torch.equal(out.view(-1, CHANNELS), pcm_dev))
    dec.set_profiling(True)
    step()
    torch.cuda.synchronize()
    ms = dec.stage_ms()
    dec.close()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if os.environ.get("FLAC_AMD_BENCH_REHEARSE") else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    alg = flac_bytes + wide * CHANNELS * 2
    dom = max(ms, key=ms.get)
    return {
        "value": stream_wide * CHANNELS * args.steps / elapsed / 1e6,
        "unit": "MSamples/s",
        "ms_per_step": elapsed / args.steps * 1e3,
        "frames": frames, "bad_frames": bad, "bit_exact_vs_input": exact,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": alg / (ms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg / (ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms},
    }


if __name__ == "__main__":
    main()
