#!/usr/bin/env python3
"""Encode / decode rates of the benchmark configurations that are not the bench line (device-resident data, one GPU):
C1  -5, 16-bit stereo, 600 s;  C3  -8 -e -p, 24-bit / 96 kHz stereo, 600 s;  C4  -8, eight channels 24-bit / 192 kHz, 60 s;
C5  one GPU's share (1250 of 10 000) of the batch decode of five-second -8 streams.
Prints one JSON line per configuration, each with the same roofline block as the bench line: algorithmic bytes (PCM at
container width + FLAC bytes) over the dominant kernel's HIP-event time, against 8 TB/s.
usage: python tools/bench_configs.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import flac_amd
from test_gpu_configs_full_size import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
HBM_PEAK_GBS = 8000.0


def roofline(alg_bytes, stage_ms):
    dom = max(stage_ms, key=stage_ms.get)
    ach = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    allk = alg_bytes / (sum(stage_ms.values()) * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes, "all_kernels_achieved": allk, "all_kernels_frac": allk / HBM_PEAK_GBS}


CONFIGS = [
    ("C1 -5 16-bit/44.1k stereo 600 s", dict(level=5, channels=2, bps=16, rate=44100, seconds=600), {}),
    ("C3 -8 -e -p 24-bit/96k stereo 600 s", dict(level=8, channels=2, bps=24, rate=96000, seconds=600), dict(do_qlp_coeff_prec_search=1)),
    ("C4 -8 8ch 24-bit/192k 60 s", dict(level=8, channels=8, bps=24, rate=192000, seconds=60), {}),
]
dev = torch.device("cuda", 0)
for name, c, over in CONFIGS:
    wide = c["seconds"] * c["rate"]
    pcm = synth(99, wide, c["channels"], c["bps"], c["rate"])
    p = flac_amd.preset(c["level"], c["channels"], c["bps"], c["rate"], **over)
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
    alg = samples * ((c["bps"] + 7) // 8) + total
    dec_ms = dec.stage_ms()
    print(json.dumps({"config": name, "frames": nframes, "samples": samples, "flac_bytes": total,
                      "encode_roofline": roofline(alg, enc_ms), "decode_roofline": roofline(alg, dec_ms),
                      "ratio": total / (samples * ((c["bps"] + 7) // 8)),
                      "encode_MSamples_s": samples / enc_s / 1e6, "encode_ms": enc_s * 1e3, "encode_kernel_ms": {k: round(v, 3) for k, v in enc_ms.items()},
                      "decode_MSamples_s": samples / dec_s / 1e6, "decode_ms": dec_s * 1e3, "decode_kernel_ms": {k: round(v, 3) for k, v in dec_ms.items()},
                      "decoded": [frames, got_wide, bad], "round_trip_exact": bool(torch.equal(back, pcm_dev)) and bad == 0 and got_wide == wide}), flush=True)
    enc.close()
    dec.close()
    del pcm_dev, out, back


# ---- C5: batch decode, one GPU's share: 1250 five-second -8 streams (25 distinct, 50 copies each, shuffled) ----
rate, seconds, distinct, copies = 44100, 5, 25, 50
wide = rate * seconds
p = flac_amd.preset(8, 2, 16, rate)
nfr1 = -(-wide // p.blocksize)
enc = flac_amd.Encoder(p, device=0, max_frames=nfr1)
cap = enc.max_output_bytes(nfr1)
sources, encoded = [], []
for i in range(distinct):
    pcm = synth(100 + i, wide, 2, 16, rate)
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


def decode5():
    dec.decode_device(flac.data_ptr(), int(flac.numel()), descs, out.data_ptr(), 2, nstreams * wide)


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
print(json.dumps({"config": "C5 batch decode, 1250 of 10 000 five-second -8 streams (one GPU's share)", "streams": nstreams, "frames": frames, "samples": samples,
                  "flac_bytes": int(flac.numel()), "decode_roofline": roofline(alg, dec_ms), "decode_MSamples_s": samples / dec_s / 1e6, "decode_ms": dec_s * 1e3,
                  "decode_kernel_ms": {k: round(v, 3) for k, v in dec_ms.items()}, "decoded": [frames, got_wide, bad], "round_trip_exact": exact and bad == 0}), flush=True)
dec.close()
