#!/usr/bin/env python3
"""Encode / decode rates of the benchmark configurations that are not the bench line (device-resident data, one GPU):
C1  -5, 16-bit stereo, 600 s;  C3  -8 -e -p, 24-bit / 96 kHz stereo, 600 s;  C4  -8, eight channels 24-bit / 192 kHz, 60 s;
C5  one GPU's share (1250 of 10 000) of the batch decode of five-second -8 streams.
Prints one JSON line per configuration, each with the same roofline block as the bench line: algorithmic bytes (PCM at
container width + FLAC bytes) over the dominant kernel's HIP-event time, against 8 TB/s.
Every stream is the benchmark's own signal (flac_amd.workloads), and every encode line carries `bytes_vs_reference`: the leading frames
against what the reference encoder (oracle/_ref/libFLAC_ref.so, in worker processes) writes for the same samples.
usage: python tools/bench_configs.py [steps] [--no-reference]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import flac_amd
from flac_amd import workloads as W

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
with_reference = "--no-reference" not in sys.argv
HBM_PEAK_GBS = 8000.0
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")
# seconds of each stream that the reference encoder is given for the byte comparison (C4: the whole stream)
REF_SECONDS = {"C1": 60, "C3": 20, "C4": 60}


def roofline(alg_bytes, stage_ms):
    dom = max(stage_ms, key=stage_ms.get)
    ach = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    allk = alg_bytes / (sum(stage_ms.values()) * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes, "all_kernels_achieved": allk, "all_kernels_frac": allk / HBM_PEAK_GBS}


def reference_frames(args):
    """Worker process: what the reference encoder writes for the first `seconds` of a configuration's signal."""
    name, seconds = args
    import libflac_ctypes as L
    c = W.CONFIGS[name]
    pcm = W.pcm(name, seconds=seconds)
    ref = L.load(REF_LIB)
    _meta, frames, ok, _ = L.encode(ref, pcm, c["bps"], c["rate"], level=c["level"], settings=c["settings"], chunk=1 << 20)
    assert ok
    return [len(f) for f in frames], b"".join(frames)


def main():
    pool, jobs = None, {}
    if with_reference and os.path.exists(REF_LIB):
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(len(REF_SECONDS))
        jobs = {n: pool.apply_async(reference_frames, ((n, s),)) for n, s in REF_SECONDS.items()}
    dev = torch.device("cuda", 0)
    for name in ("C1", "C3", "C4"):
        c = W.CONFIGS[name]
        pcm = W.pcm(name)
        wide = pcm.shape[0]
        p = W.params(name, flac_amd)
        sample_bytes = 2 if c["bps"] <= 16 else 4
        host = pcm.astype(np.int16) if sample_bytes == 2 else pcm
        pcm_dev = torch.from_numpy(host).to(dev)
        nframes = -(-wide // p.blocksize)
        enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
        cap = enc.max_output_bytes(nframes)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
        offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
        enc.set_profiling(True)

        def encode():
            enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        encode()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            encode()
        torch.cuda.synchronize()
        enc_s = (time.perf_counter() - t0) / steps
        enc.status()
        enc_ms = enc.stage_ms()
        total = int(offs[nframes].item())
        dec = flac_amd.Decoder(c["channels"], c["bps"], p.blocksize, device=0, max_input_bytes=total + 64, max_frames=nframes + 8)
        back = torch.empty_like(pcm_dev)
        dec.set_profiling(True)

        def decode():
            dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide)], back.data_ptr(), sample_bytes, wide)
        decode()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            decode()
        torch.cuda.synchronize()
        dec_s = (time.perf_counter() - t0) / steps
        frames, got_wide, bad = dec.result()
        samples = wide * c["channels"]
        cont = W.container_bytes(name)
        alg = samples * cont + total
        dec_ms = dec.stage_ms()
        line = {"config": c["title"], "signal": "flac_amd.synth.synth_pcm, seed %d" % W.BASE_SEED, "frames": nframes, "samples": samples, "flac_bytes": total,
                "encode_roofline": roofline(alg, enc_ms), "decode_roofline": roofline(alg, dec_ms),
                "ratio": total / (samples * cont),
                "encode_MSamples_s": samples / enc_s / 1e6, "encode_ms": enc_s * 1e3, "encode_kernel_ms": {k: round(v, 3) for k, v in enc_ms.items()},
                "decode_MSamples_s": samples / dec_s / 1e6, "decode_ms": dec_s * 1e3, "decode_kernel_ms": {k: round(v, 3) for k, v in dec_ms.items()},
                "decoded": [frames, got_wide, bad], "round_trip_exact": bool(torch.equal(back, pcm_dev)) and bad == 0 and got_wide == wide}
        if name in jobs:
            # the reference's frames for the leading stretch against this encode's (the slice's own last block is short
            # where the long stream's is not, so a slice is compared up to its last full block)
            rsizes, rbytes = jobs[name].get(timeout=900)
            n = len(rsizes) if REF_SECONDS[name] * c["rate"] >= wide else (REF_SECONDS[name] * c["rate"]) // p.blocksize
            gsz = sizes[:n].cpu().numpy().astype(np.int64)
            gbytes = out[:int(gsz.sum())].cpu().numpy().tobytes()
            rb = rbytes[:int(sum(rsizes[:n]))]
            line["bytes_vs_reference"] = {"frames_compared": int(n), "gpu_bytes": len(gbytes), "reference_bytes": len(rb),
                                          "delta_percent": 100.0 * (len(gbytes) - len(rb)) / len(rb), "identical": gbytes == rb,
                                          "what": "the first %d frames of the stream against the reference encoder's (libFLAC_ref.so) for the same samples" % n}
        print(json.dumps(line), flush=True)
        enc.close()
        dec.close()
        del pcm_dev, out, back
    if pool:
        pool.terminate()
        pool.join()
    batch_decode(dev)


def batch_decode(dev):
    # ---- C5: batch decode, one GPU's share: 1250 five-second -8 streams (25 distinct, 50 copies each, shuffled) ----
    rate, seconds, distinct, copies = 44100, 5, 25, 50
    wide = rate * seconds
    p = W.params("C5", flac_amd)
    nfr1 = -(-wide // p.blocksize)
    enc = flac_amd.Encoder(p, device=0, max_frames=nfr1)
    cap = enc.max_output_bytes(nfr1)
    sources, encoded = [], []
    for i in range(distinct):
        pcm = W.pcm("C5", stream_id=i)
        d = torch.from_numpy(pcm.astype(np.int16)).to(dev)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        offs = torch.zeros(nfr1 + 1, dtype=torch.int64, device=dev)
        sizes = torch.zeros(nfr1, dtype=torch.int32, device=dev)
        enc.encode_device(d.data_ptr(), 2, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        sources.append(d)
        encoded.append(out[:int(offs[nfr1].item())].clone())
    enc.close()
    order = np.random.default_rng(1).permutation(distinct * copies) % distinct
    pieces, descs, pos = [], [], 0
    for k, which in enumerate(order):
        n = int(encoded[which].numel())
        descs.append((pos, n, k * wide, wide, 0))
        pieces.append(encoded[which])
        pad = (-n) % 4
        if pad:
            pieces.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
        pos += n + pad
    flac = torch.cat(pieces)
    nstreams = len(order)
    dec = flac_amd.Decoder(2, 16, p.blocksize, device=0, max_input_bytes=int(flac.numel()) + 64, max_frames=nstreams * nfr1 + 64)
    out = torch.zeros((nstreams * wide, 2), dtype=torch.int16, device=dev)
    dec.set_profiling(True)

    desc_array = dec.stream_array(descs)

    def decode5():
        dec.decode_device(flac.data_ptr(), int(flac.numel()), desc_array, out.data_ptr(), 2, nstreams * wide)

    decode5()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        decode5()
    torch.cuda.synchronize()
    dec_s = (time.perf_counter() - t0) / steps
    frames, got_wide, bad = dec.result()
    dec_ms = dec.stage_ms()
    samples = nstreams * wide * 2
    alg = samples * 2 + int(flac.numel())
    exact = all(bool((out.view(nstreams, wide, 2)[torch.from_numpy(np.nonzero(order == w)[0]).to(dev)] == sources[w].unsqueeze(0)).all()) for w in range(distinct))
    print(json.dumps({"config": "C5 batch decode, 1250 of 10 000 five-second -8 streams (one GPU's share)", "signal": "flac_amd.synth.synth_pcm, seeds 1234 + 1000 i",
                      "streams": nstreams, "frames": frames, "samples": samples,
                      "flac_bytes": int(flac.numel()), "decode_roofline": roofline(alg, dec_ms), "decode_MSamples_s": samples / dec_s / 1e6, "decode_ms": dec_s * 1e3,
                      "decode_kernel_ms": {k: round(v, 3) for k, v in dec_ms.items()}, "decoded": [frames, got_wide, bad], "round_trip_exact": exact and bad == 0}), flush=True)
    dec.close()


if __name__ == "__main__":
    main()
