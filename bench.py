#!/usr/bin/env python3
"""Benchmark of the FLAC hot path on MI355X: frame encode (and decode) of synthetic PCM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default (the bench line): C2, `flac -8` on a 600 s 16-bit / 44.1 kHz stereo stream (26.46 M wide samples, 6460
frames of 4096) that already lies in HBM as interleaved int16; one step = one pass of the whole encode pipeline
over it.  With N > 1 every rank owns the frames [rank*6460, (rank+1)*6460) of an N-times longer stream (weak
scaling); the only exchange is an all-gather of per-rank byte totals for the stream offsets.

--config C1 | C3 | C4 run the other encode configurations the same way; their default is strong scaling: the one
stream of the configuration is cut into contiguous frame ranges over the ranks (--scaling weak makes the stream N
times longer instead).  --config C5 is the batch decode: 10 000 five-second `-8` streams (100 distinct signals x
100 copies, --streams to change the count) dealt round-robin over the ranks, nothing exchanged.

Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
METRIC = "encode/decode MSamples/s per node at -8 (1/2/4/8 GPU); bytes vs reference"
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "flac_ref")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")
HELPER = os.path.join(ROOT, "tools", "bin", "libbench_helper.so")


# ---------------------------------------------------------------------------------------------
# CPU baseline: the reference's own `flac` tool (oracle/_ref/flac_ref, compiled from the reference's sources
# in the build container, plain gcc -O2, C only) run as one process per CPU of this job's share, each pinned to
# its CPU, each on its own copy of the input in tmpfs: encode, `-d` and `-t`.  Without that build, this repo's
# scalar CPU restatement (encode only, kind "port").
# ---------------------------------------------------------------------------------------------
def host_cpus():
    """(CPUs this process may run on, the share of them the job is meant to use, the machine's count)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    share = len(allowed)
    # On a shared GPU host the job's CPU share is smaller than what the affinity mask shows; the pool says how
    # large it is through the worker-count variables it sets (16 per GPU).  FLAC_AMD_BENCH_CPU_PROCS overrides.
    for var in ("FLAC_AMD_BENCH_CPU_PROCS", "OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            share = min(share, int(v)) if var != "FLAC_AMD_BENCH_CPU_PROCS" else min(len(allowed), int(v))
            break
    return allowed, max(1, share), os.cpu_count() or len(allowed)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _pin(cpu):
    def fn():
        try:
            os.sched_setaffinity(0, {cpu})
        except (AttributeError, OSError):
            pass
    return fn


def _fan_out(cmds, cpus):
    """Start one command per CPU, each pinned, all at once; returns (wall of the slowest, per-process walls)."""
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, preexec_fn=_pin(cpu), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for cmd, cpu in zip(cmds, cpus)]
    walls = []
    for p in procs:
        _out, err = p.communicate()
        walls.append(time.perf_counter() - t0)
        if p.returncode != 0:
            raise RuntimeError("reference flac failed (%d): %s" % (p.returncode, err.decode(errors="replace")[-300:]))
    return max(walls), walls


def raw_bytes(pcm, bps):
    """Interleaved little-endian signed samples at the container width `flac --force-raw-format` reads."""
    if bps <= 8:
        return pcm.astype(np.int8).tobytes()
    if bps <= 16:
        return pcm.astype("<i2").tobytes()
    if bps <= 24:
        b = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        return np.ascontiguousarray(b).tobytes()
    return pcm.astype("<i4").tobytes()


def strip_metadata(flac_file_bytes):
    """The frames of a FLAC file: everything behind the metadata blocks (format.h: 'fLaC', then blocks of a 4-byte
    header whose first bit marks the last one)."""
    assert flac_file_bytes[:4] == b"fLaC"
    pos = 4
    while True:
        last = flac_file_bytes[pos] & 0x80
        length = int.from_bytes(flac_file_bytes[pos + 1:pos + 4], "big")
        pos += 4 + length
        if last:
            return flac_file_bytes[pos:]


def flac_options(cfg):
    opts = ["-%d" % cfg["level"]]
    if cfg["over"].get("do_exhaustive_model_search"):
        opts.append("-e")
    if cfg["over"].get("do_qlp_coeff_prec_search"):
        opts.append("-p")
    return opts


def cpu_baseline_config(name, seconds, nprocs, cpus, tmpdir, keep_first=False, exact_prefix=False):
    """Reference CLI on `nprocs` pinned processes: encode, decode to raw, test.  Returns the figures and, if asked,
    the first process's encoded file bytes."""
    from flac_amd import workloads as W
    cfg = W.CONFIGS[name]
    pcm = W.pcm(name, seconds=seconds) if exact_prefix else W.sample(name, seconds)
    raw = raw_bytes(pcm, cfg["bps"])
    samples = pcm.shape[0] * cfg["channels"]
    ins, outs, backs = [], [], []
    for i in range(nprocs):
        ins.append(os.path.join(tmpdir, "%s_in_%d.raw" % (name, i)))
        outs.append(os.path.join(tmpdir, "%s_out_%d.flac" % (name, i)))
        backs.append(os.path.join(tmpdir, "%s_back_%d.raw" % (name, i)))
        with open(ins[-1], "wb") as f:
            f.write(raw)
    fmt = ["--force-raw-format", "--endian=little", "--sign=signed", "--channels=%d" % cfg["channels"], "--bps=%d" % cfg["bps"],
           "--sample-rate=%d" % cfg["rate"]]
    enc_wall, _ = _fan_out([[REF_CLI, "--totally-silent", "-f"] + flac_options(cfg) + fmt + ["-o", outs[i], ins[i]] for i in range(nprocs)], cpus)
    dec_wall, _ = _fan_out([[REF_CLI, "--totally-silent", "-f", "-d", "--force-raw-format", "--endian=little", "--sign=signed", "-o", backs[i], outs[i]]
                            for i in range(nprocs)], cpus)
    test_wall, _ = _fan_out([[REF_CLI, "--totally-silent", "-t", outs[i]] for i in range(nprocs)], cpus)
    with open(backs[0], "rb") as f:
        lossless = f.read() == raw
    first = None
    if keep_first:
        with open(outs[0], "rb") as f:
            first = f.read()
    flac_size = os.path.getsize(outs[0])
    for p in ins + outs + backs:
        try:
            os.remove(p)
        except OSError:
            pass
    total = samples * nprocs
    return {
        "options": "flac " + " ".join(flac_options(cfg)),
        "sample_seconds": seconds,
        "encode": total / enc_wall / 1e6, "decode": total / dec_wall / 1e6, "test": total / test_wall / 1e6,
        "encode_per_core": samples / enc_wall / 1e6, "decode_per_core": samples / dec_wall / 1e6,
        "walls_s": {"encode": enc_wall, "decode": dec_wall, "test": test_wall},
        "flac_file_bytes": flac_size, "decoded_equals_input": bool(lossless),
    }, first


# bounded samples per configuration (seconds of signal per process): about 10 s of CPU work per process in all
CPU_SAMPLE_SECONDS = {"C1": 60.0, "C2": 120.0, "C3": 8.0, "C4": 4.0}


def cpu_baseline(workload):
    """Returns (cpu_baseline object for the JSON line, frame bytes the reference wrote for the workload's sample)."""
    allowed, nprocs, machine = host_cpus()
    cpus = allowed[:nprocs]
    info = {"cores": nprocs, "host_cpus_visible": len(allowed), "host_cpus_machine": machine, "cpu_model": cpu_model(),
            "unit": "MSamples/s"}
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    if os.path.exists(REF_CLI):
        tmpdir = tempfile.mkdtemp(prefix="flac_amd_cpu_", dir=tmp_root)
        try:
            per = {}
            ref_frames = None
            names = [workload] + [n for n in ("C2", "C1", "C3", "C4") if n != workload and workload != "C5"]
            if workload == "C5":
                names = ["C2"]
            t0 = time.perf_counter()
            for n in names:
                # (the workload's own sample is a true prefix of its stream, so that the bytes can be compared with the GPU's; the
                # other configurations get a short run of their generator)
                per[n], first = cpu_baseline_config(n, CPU_SAMPLE_SECONDS[n], nprocs, cpus, tmpdir, keep_first=(n == names[0]), exact_prefix=(n == names[0]))
                if first is not None:
                    ref_frames = strip_metadata(first)
            main = per[names[0]]
            info.update({
                "value": main["decode"] if workload == "C5" else main["encode"],
                "kind": "reference",
                "what": "decode (flac -d)" if workload == "C5" else "encode",
                "sample": "%d pinned processes of the reference's own flac tool (oracle/_ref/flac_ref: the reference's sources, gcc -O2, C only), "
                          "each on its own copy in tmpfs of the first %.0f s of the workload's stream, %s; throughput = all samples / wall of "
                          "the slowest process; then flac -d and flac -t of what they wrote.  The job's CPU share is %d of the %d CPUs this "
                          "host shows, so an all-core figure for this host is per_core x %d (not measured: the other CPUs are not this job's)"
                          % (nprocs, main["sample_seconds"], main["options"], nprocs, machine, machine),
                "per_core": main["decode_per_core"] if workload == "C5" else main["encode_per_core"],
                "decode": main["decode"], "test": main["test"],
                "configs": per, "wall_s": time.perf_counter() - t0,
            })
            return info, ref_frames, CPU_SAMPLE_SECONDS[names[0]]
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)
    # no reference build here: this repo's scalar restatement of the encoder, one process per CPU
    import multiprocessing as mp
    secs = 20.0
    with mp.get_context("spawn").Pool(nprocs) as pool:
        res = pool.map(_port_worker, [("C2" if workload == "C5" else workload, secs)] * nprocs)
    slowest = max(r[0] for r in res)
    info.update({"value": sum(r[1] for r in res) / slowest / 1e6, "kind": "port", "what": "encode",
                 "sample": "%d processes x %.0f s of the workload's signal through the scalar CPU restatement (oracle/), encode only" % (nprocs, secs),
                 "per_core": res[0][1] / slowest / 1e6})
    return info, res[0][2], secs


def _port_worker(args):
    name, seconds = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from flac_amd import workloads as W
    cfg = W.CONFIGS[name]
    pcm = W.pcm(name, seconds=seconds)
    p = O.preset(cfg["level"], cfg["channels"], cfg["bps"], cfg["rate"])
    for k, v in cfg["over"].items():
        setattr(p, k, v)
    t0 = time.perf_counter()
    frames, _ = O.encode_frames(p, pcm)
    return time.perf_counter() - t0, pcm.shape[0] * cfg["channels"], b"".join(frames)


def launch_ranks(n):
    """Run this script as n ranks under torch.distributed.run (one per GPU, RCCL rendezvous on
    127.0.0.1) and return the launcher's exit code; rank 0's JSON line passes through on stdout."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def roofline_block(alg_bytes, stage_ms):
    dom = max(stage_ms, key=stage_ms.get)
    ach = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    allk = alg_bytes / (sum(stage_ms.values()) * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": stage_ms, "all_kernels_achieved": allk, "all_kernels_frac": allk / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--seconds", type=float, default=None, help="stream length (default: the configuration's)")
    ap.add_argument("--streams", type=int, default=None, help="C5: streams in the batch (default 10 000)")
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
    from flac_amd import shard
    from flac_amd import workloads as W

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
    ctx = dict(torch=torch, dist=dist, flac_amd=flac_amd, shard=shard, W=W, rank=rank, local_rank=local_rank, world=world, dev=dev,
               coll_dev=coll_dev, rehearse=rehearse, args=args)
    name = args.config
    scaling = args.scaling or W.default_scaling(name)
    if name == "C5":
        result = bench_batch_decode(ctx, scaling)
    else:
        result = bench_encode(ctx, name, scaling)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def timed(ctx, step, drain=None):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; the maximum over ranks."""
    torch, dist, world, args = ctx["torch"], ctx["dist"], ctx["world"], ctx["args"]
    for k in range(args.warmup):
        step(k)
    if drain:
        drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    if drain:
        drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx["coll_dev"])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def bench_encode(ctx, name, scaling):
    torch, dist, flac_amd, shard, W = ctx["torch"], ctx["dist"], ctx["flac_amd"], ctx["shard"], ctx["W"]
    rank, local_rank, world, dev, coll_dev, args = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["dev"], ctx["coll_dev"], ctx["args"]
    cfg = W.CONFIGS[name]
    channels, bps = cfg["channels"], cfg["bps"]
    # ---- workload: this rank's shard of the stream, resident in HBM at the engine's container width ----
    # The job is ONE stream cut into contiguous frame ranges; every block is full except the stream's last one,
    # so the ranks' bytes put end to end are a valid stream
    params = W.params(name, flac_amd)
    bs = params.blocksize
    pl = W.plan(name, scaling, world, rank, bs, seconds=args.seconds)
    wide, first_frame, is_last, stream_wide = pl["wide"], pl["first_frame"], pl["is_last"], pl["stream_wide"]
    nframes = (wide + bs - 1) // bs
    if nframes == 0:
        raise SystemExit("bench.py: rank %d of %d has no frame of this stream to encode; use fewer GPUs or a longer stream" % (rank, world))
    pcm32 = W.shard_pcm(name, pl)
    assert pcm32.shape == (wide, channels)
    sample_bytes = 2 if bps <= 16 else 4
    pcm_dev = torch.from_numpy(pcm32.astype(np.int16) if sample_bytes == 2 else pcm32).to(dev)
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
        enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, out_dev.data_ptr(), out_cap, fbytes_dev.data_ptr(),
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

    elapsed = timed(ctx, step, drain)
    status_checked = not os.environ.get("FLAC_AMD_BENCH_NOCHECK")  # kernel experiments whose output is knowingly wrong
    if status_checked:
        enc.status()
    samples_per_step = pl["job_samples"]
    value = samples_per_step * args.steps / elapsed / 1e6
    flac_bytes = int(offs_dev[nframes].item())

    # ---- per-kernel durations, HIP events on the launch stream, separate passes ----
    enc.set_profiling(True)
    acc = {}
    nprof = max(3, min(10, args.steps))
    for _ in range(nprof):
        enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, out_dev.data_ptr(), out_cap, fbytes_dev.data_ptr(),
                          offs_dev.data_ptr(), first_frame=first_frame, is_last=is_last, stream=stream)
        torch.cuda.synchronize()
        for k, v in enc.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / nprof
    enc.set_profiling(False)
    cont = W.container_bytes(name)
    alg_bytes = wide * channels * cont + flac_bytes  # PCM read once at container width + frames written once

    result = {
        "metric": METRIC,
        "value": value,
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        **({"rehearsal": "all ranks on GPU 0, gloo collectives through host memory: not a measurement"} if ctx["rehearse"] else {}),
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "status_checked": bool(status_checked),
        "config": {
            "workload": "%s%s; %d frames on this GPU, PCM resident in HBM as int%d" % (
                cfg["title"], "" if args.seconds is None else " (cut to %.0f s)" % args.seconds, nframes, 8 * sample_bytes),
            "what": "encode",
            "frames_per_gpu": nframes,
            "samples_per_step": samples_per_step,
            "flac_bytes_per_gpu": flac_bytes,
            "compression_ratio": flac_bytes / (wide * channels * cont),
            "sharding": ("one stream of %d wide samples cut into contiguous frame ranges, one per GPU; all-gather of byte totals" % stream_wide)
                        if world > 1 else "single GPU",
        },
        "roofline": roofline_block(alg_bytes, acc),
    }

    # ---- decode of the same stream (device resident), reported beside the encode number ----
    if not args.no_decode:
        try:
            result["decode"] = bench_decode(ctx, cfg, out_dev, flac_bytes, wide, bs, nframes, pcm_dev, first_frame, pl, sample_bytes, cont)
        except Exception as exc:  # the encode line must still be printed
            result["decode"] = {"error": str(exc)}

    # ---- N > 1: the shards put together into the one stream they belong to (output collection,
    # outside the timed region), which rank 0 then decodes whole: every frame of every rank must
    # be found, in order, with a good CRC-16 ----
    if world > 1:
        whole = shard.gather_stream(out_dev[:flac_bytes])
        if rank == 0:
            try:
                total_frames = (stream_wide + bs - 1) // bs
                dec = flac_amd.Decoder(channels, bps, bs, device=local_rank, max_input_bytes=whole.numel() + 64, max_frames=total_frames + 16)
                back = torch.empty(stream_wide * channels, dtype=pcm_dev.dtype, device=dev)
                dec.decode_device(whole.data_ptr(), whole.numel(), [(0, whole.numel(), 0, stream_wide, 0)], back.data_ptr(), sample_bytes, stream_wide)
                frames, got_wide, bad = dec.result()
                dec.close()
                result["assembled_stream"] = {"bytes": int(whole.numel()), "frames": frames, "wide_samples": got_wide, "bad_frames": bad,
                                              "rank0_shard_bit_exact": bool(torch.equal(back[:wide * channels].view(-1, channels), pcm_dev)),
                                              "ok": frames == total_frames and bad == 0 and got_wide == stream_wide}
                del back
            except Exception as exc:
                result["assembled_stream"] = {"error": str(exc)}
        del whole

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    # the gfx950 note prescribes, + WRITE_SIZE); collected by separate runs of this same command
    if name == "C2" and args.seconds is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tr = json.load(f)
            dominant = result["roofline"]["kernel"]
            if dominant in tr.get("kernels", {}):
                result["roofline"]["traffic"] = tr["kernels"][dominant]["bytes_per_launch"]
                result["roofline"]["traffic_source"] = tr.get("source")
        except (OSError, ValueError):
            pass

    if rank == 0 and world == 1 and not args.no_stream_api:
        try:
            result["end_to_end"] = end_to_end(flac_amd, W, name, pcm32)
        except Exception as exc:
            result["end_to_end"] = {"error": str(exc)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cb, ref_frames, sample_seconds = cpu_baseline(name)
            result["cpu_baseline"] = cb
            # the same bounded sample through the GPU path: output bytes vs what the reference wrote
            sample = W.pcm(name, seconds=sample_seconds)
            genc = flac_amd.Encoder(params, device=local_rank, max_frames=(sample.shape[0] + bs - 1) // bs)
            gframes, _ = genc.encode_host(sample)
            genc.close()
            gpu = b"".join(gframes)
            result["bytes_vs_reference"] = {
                "gpu_bytes": len(gpu), "cpu_bytes": len(ref_frames), "delta_percent": 100.0 * (len(gpu) - len(ref_frames)) / len(ref_frames),
                "identical": gpu == ref_frames,
                "what": "frame bytes of the first %.0f s of the workload's signal, GPU path against what the CPU baseline's encoder wrote for the "
                        "same samples; the whole stream is compared byte for byte in tests/test_gpu_reference_accepts.py" % sample_seconds}
        except Exception as exc:
            result["cpu_baseline"] = {"error": str(exc)}
    elif rank == 0:
        result["cpu_baseline"] = None
    enc.close()
    return result


def bench_decode(ctx, cfg, flac_dev, flac_bytes, wide, bs, nframes, pcm_dev, first_frame, pl, sample_bytes, cont):
    torch, flac_amd, args, dev = ctx["torch"], ctx["flac_amd"], ctx["args"], ctx["dev"]
    channels, bps = cfg["channels"], cfg["bps"]
    dec = flac_amd.Decoder(channels, bps, bs, device=ctx["local_rank"], max_input_bytes=flac_bytes + 64, max_frames=nframes + 16)
    out = torch.zeros(wide * channels, dtype=pcm_dev.dtype, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # the encoder numbered this rank's frames from first_frame; tell the decoder where sample 0 is
    streams = [(0, flac_bytes, 0, wide, first_frame * bs)]

    def step(_k):
        dec.decode_device(flac_dev.data_ptr(), flac_bytes, streams, out.data_ptr(), sample_bytes, wide, stream=stream)

    elapsed = timed(ctx, step)
    frames, _nwide, bad = dec.result()
    exact = bool(torch.equal(out.view(-1, channels), pcm_dev))
    dec.set_profiling(True)
    step(0)
    torch.cuda.synchronize()
    ms = dec.stage_ms()
    dec.close()
    alg = flac_bytes + wide * channels * cont
    rf = roofline_block(alg, ms)
    del rf["traffic"]
    return {"value": pl["job_samples"] * args.steps / elapsed / 1e6, "unit": "MSamples/s", "ms_per_step": elapsed / args.steps * 1e3,
            "frames": frames, "bad_frames": bad, "bit_exact_vs_input": exact, "roofline": rf}


def bench_batch_decode(ctx, scaling):
    """C5: this rank's share of the batch of short streams, decoded in one call per step."""
    torch, flac_amd, W = ctx["torch"], ctx["flac_amd"], ctx["W"]
    rank, local_rank, world, dev, args = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["dev"], ctx["args"]
    name = "C5"
    cfg = W.CONFIGS[name]
    params = W.params(name, flac_amd)
    bs = params.blocksize
    pl = W.plan(name, scaling, world, rank, bs, seconds=args.seconds, streams=args.streams)
    wide = pl["stream_wide"]
    nfr1 = -(-wide // bs)
    distinct = pl["distinct"]
    mine = pl["mine"]
    # the distinct streams, made and encoded here (stream i of the batch is signal i mod distinct)
    enc = flac_amd.Encoder(params, device=local_rank, max_frames=nfr1)
    cap = enc.max_output_bytes(nfr1)
    needed = sorted({i % distinct for i in mine})
    sources, encoded = {}, {}
    for sid in needed:
        pcm = W.pcm(name, stream_id=sid, wide=wide)
        d = torch.from_numpy(pcm.astype(np.int16)).to(dev)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        offs = torch.zeros(nfr1 + 1, dtype=torch.int64, device=dev)
        sizes = torch.zeros(nfr1, dtype=torch.int32, device=dev)
        enc.encode_device(d.data_ptr(), 2, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        sources[sid] = d
        encoded[sid] = out[:int(offs[nfr1].item())].clone()
    enc.status()
    enc.close()
    # streams start on 4-byte boundaries, as separately stored files would after a copy
    pieces, descs, pos = [], [], 0
    zero_pad = torch.zeros(4, dtype=torch.uint8, device=dev)
    for k, i in enumerate(mine):
        e = encoded[i % distinct]
        n = int(e.numel())
        descs.append((pos, n, k * wide, wide, 0))
        pieces.append(e)
        pad = (-n) % 4
        if pad:
            pieces.append(zero_pad[:pad])
        pos += n + pad
    flac = torch.cat(pieces)
    del pieces
    nstreams = len(mine)
    flac_bytes = int(flac.numel())
    dec = flac_amd.Decoder(cfg["channels"], cfg["bps"], bs, device=local_rank, max_input_bytes=flac_bytes + 64, max_frames=nstreams * nfr1 + 64)
    out = torch.zeros((nstreams * wide, cfg["channels"]), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    desc_array = dec.stream_array(descs)

    def step(_k):
        dec.decode_device(flac.data_ptr(), flac_bytes, desc_array, out.data_ptr(), 2, nstreams * wide, stream=stream)

    elapsed = timed(ctx, step)
    frames, got_wide, bad = dec.result()
    which = np.array([i % distinct for i in mine])
    got = out.view(nstreams, wide, cfg["channels"])
    exact = all(bool((got[torch.from_numpy(np.nonzero(which == sid)[0]).to(dev)] == sources[sid].unsqueeze(0)).all()) for sid in needed)
    dec.set_profiling(True)
    step(0)
    torch.cuda.synchronize()
    ms = dec.stage_ms()
    dec.close()
    samples = nstreams * wide * cfg["channels"]
    alg = flac_bytes + samples * 2
    result = {
        "metric": METRIC,
        "value": pl["job_samples"] * args.steps / elapsed / 1e6,
        "unit": "MSamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%s; %d streams in all, %d (%d frames) on this GPU, dealt round-robin, nothing exchanged; FLAC bytes resident in HBM, "
                               "int16 PCM out" % (cfg["title"] if pl["total_streams"] == cfg["streams"] else cfg["title"].replace("10 000", str(pl["total_streams"])),
                                                  pl["total_streams"], nstreams, frames),
                   "what": "decode", "streams_per_gpu": nstreams, "frames_per_gpu": frames, "samples_per_step": pl["job_samples"],
                   "flac_bytes_per_gpu": flac_bytes},
        "decoded": {"frames": frames, "wide_samples": got_wide, "bad_frames": bad, "every_stream_bit_exact": bool(exact) and bad == 0 and got_wide == nstreams * wide},
        "roofline": roofline_block(alg, ms),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"], _frames, _secs = cpu_baseline("C5")
        except Exception as exc:
            result["cpu_baseline"] = {"error": str(exc)}
    elif rank == 0:
        result["cpu_baseline"] = None
    return result


# ---------------------------------------------------------------------------------------------
# The libFLAC-compatible stream API from host memory: never the headline value.
# ---------------------------------------------------------------------------------------------
def end_to_end(flac_amd, W, name, pcm32):
    cfg = W.CONFIGS[name]
    stream_api_encode(flac_amd, cfg, pcm32[:1 << 20], do_md5=False)  # first use of the stream layer's buffers and threads
    path = os.path.join(_tmp_root(), "flac_amd_bench_%d_keep.flac" % os.getpid())
    try:
        e2e = stream_api_encode(flac_amd, cfg, pcm32, keep=path)
        e2e["without_md5"] = stream_api_encode(flac_amd, cfg, pcm32, do_md5=False)
        e2e["without_md5_null_sink"] = stream_api_encode(flac_amd, cfg, pcm32, do_md5=False, sink="null")
        # K encoder objects on K host threads, one GPU: where the aggregate stops growing is the per-GPU ceiling of the
        # host side (staging copies, MD5 threads, PCIe)
        for k in (4, 16, 32):
            e2e["encoders_x%d" % k] = stream_api_encode(flac_amd, cfg, pcm32, do_md5=True, threads=k)
            e2e["encoders_x%d_without_md5" % k] = stream_api_encode(flac_amd, cfg, pcm32, do_md5=False, threads=k)
        if os.path.exists(HELPER):
            e2e["decode"] = stream_api_decode(flac_amd, cfg, path, pcm32, md5=True)
            e2e["decode_without_md5"] = stream_api_decode(flac_amd, cfg, path, pcm32, md5=False)
            e2e["decoders_x4"] = stream_api_decode(flac_amd, cfg, path, pcm32, md5=True, threads=4)
        else:
            e2e["decode"] = {"error": "tools/bin/libbench_helper.so has not been built"}
        if cfg["bps"] <= 16:
            # a stream wider than 16 bits crosses PCIe as int32 and its MD5 input is re-packed: 60 s of C3's signal at -8
            c3 = dict(W.CONFIGS["C3"], over={}, settings={})
            pcm24 = W.pcm("C3", seconds=60)
            e2e["encode_24bit_96k"] = stream_api_encode(flac_amd, c3, pcm24)
            e2e["encode_24bit_96k"]["without_md5"] = stream_api_encode(flac_amd, c3, pcm24, do_md5=False)
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    return e2e


def _tmp_root():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()


def stream_api_encode(flac_amd, cfg, pcm32, do_md5=True, threads=1, sink="tmpfs", keep=None):
    """The stream through the libFLAC-compatible API from host memory: int32 samples in through
    FLAC__stream_encoder_process_interleaved in 1 Mi-sample calls, MD5, PCIe both ways, the library's
    own FILE write / seek / tell callbacks (FLAC__stream_encoder_init_file, what the `flac` tool uses)
    onto a tmpfs file.  do_md5=False switches the digest off (FLAC__stream_encoder_set_do_md5,
    stream_encoder.c:1414), which shows what the copies and the GPU work take on their own.  threads > 1:
    that many encoder objects on as many host threads, each coding its own copy of the stream on the one
    GPU (aggregate rate)."""
    import threading
    lib = flac_amd.lib()
    wide, ch = pcm32.shape
    data = np.ascontiguousarray(pcm32, dtype=np.int32)
    lib.FLAC__stream_encoder_new.restype = C.c_void_p
    tmpdir = _tmp_root()
    results = [None] * threads
    encs = []
    for t in range(threads):
        enc = C.c_void_p(lib.FLAC__stream_encoder_new())
        for key, v in (("channels", ch), ("bits_per_sample", cfg["bps"]), ("sample_rate", cfg["rate"]), ("compression_level", cfg["level"])):
            getattr(lib, "FLAC__stream_encoder_set_" + key)(enc, C.c_uint(v))
        for key, v in cfg["settings"].items():
            getattr(lib, "FLAC__stream_encoder_" + key)(enc, C.c_uint(v))
        lib.FLAC__stream_encoder_set_total_samples_estimate(enc, C.c_uint64(wide))
        if not do_md5:
            lib.FLAC__stream_encoder_set_do_md5(enc, C.c_int(0))
        encs.append(enc)

    def run(t):
        enc = encs[t]
        path = "/dev/null" if sink == "null" else (keep if keep and t == 0 else os.path.join(tmpdir, "flac_amd_bench_%d_%d.flac" % (os.getpid(), t)))
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
        if sink != "null" and not (keep and t == 0):
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
    return {"value": threads * wide * ch / (t_all - init_s if threads > 1 else proc_s) / 1e6, "unit": "MSamples/s", "seconds": proc_s, "init_seconds": init_s,
            "file_bytes": results[0][2], "threads": threads,
            "what": "FLAC__stream_encoder_init_file + process_interleaved + finish of %.0f s of %d-bit/%d Hz x %d from host int32: %shost<->device "
                    "copies and the library's FILE callbacks onto %s included; %d host thread(s) calling, %d encoder object(s) on one GPU"
                    % (wide / cfg["rate"], cfg["bps"], cfg["rate"], ch, "MD5, " if do_md5 else "no MD5, ", "/dev/null" if sink == "null" else "a tmpfs file",
                       threads, threads)}


class _Sink(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("wide_samples", C.c_uint64), ("errors", C.c_uint64), ("sum", C.c_int64)]


def stream_api_decode(flac_amd, cfg, path, pcm32, md5=True, threads=1):
    """The file the encode line wrote, back through FLAC__stream_decoder_init_file + process_until_end_of_stream
    (what `flac -d` / `flac -t` call): the library's FILE read callbacks, PCIe both ways, the write callback once
    per frame with int32 planes in host memory (a C callback that sums every sample), MD5 checked against STREAMINFO."""
    import threading
    lib = flac_amd.lib()
    helper = C.CDLL(HELPER)
    wcb = C.cast(helper.fa_bench_write_cb, C.c_void_p)
    ecb = C.cast(helper.fa_bench_error_cb, C.c_void_p)
    lib.FLAC__stream_decoder_new.restype = C.c_void_p
    wide, ch = pcm32.shape
    want_sum = int(pcm32.sum(dtype=np.int64))
    sinks = [_Sink() for _ in range(threads)]
    results = [None] * threads

    def run(t):
        dec = C.c_void_p(lib.FLAC__stream_decoder_new())
        try:
            lib.FLAC__stream_decoder_set_md5_checking(dec, C.c_int(1 if md5 else 0))
            t0 = time.perf_counter()
            st = lib.FLAC__stream_decoder_init_file(dec, path.encode(), wcb, None, ecb, C.byref(sinks[t]))
            if st != 0:
                results[t] = RuntimeError("decoder init_file status %d" % st)
                return
            ok = bool(lib.FLAC__stream_decoder_process_until_end_of_stream(dec))
            fin = bool(lib.FLAC__stream_decoder_finish(dec))  # false on an MD5 mismatch
            results[t] = (time.perf_counter() - t0, ok, fin)
        finally:
            lib.FLAC__stream_decoder_delete(dec)

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
    for r in results:
        if isinstance(r, Exception):
            raise r
    good = all(r[1] and r[2] for r in results) and all(s.wide_samples == wide and s.errors == 0 and s.sum == want_sum for s in sinks)
    return {"value": threads * wide * ch / t_all / 1e6, "unit": "MSamples/s", "seconds": t_all, "threads": threads, "every_sample_back": bool(good),
            "what": "FLAC__stream_decoder_init_file + process_until_end_of_stream + finish of the file the encode line wrote: file reads, host<->device "
                    "copies, %sone write callback per frame (int32 planes in host memory, every sample summed by a C callback); %d host thread(s), "
                    "%d decoder object(s) on one GPU" % ("MD5 of the decoded samples checked against STREAMINFO, " if md5 else "no MD5, ", threads, threads)}


if __name__ == "__main__":
    main()
