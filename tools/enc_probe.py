#!/usr/bin/env python3
"""Encode kernel times of one benchmark configuration under settings of the library's experiment switches.
usage: python tools/enc_probe.py C3 [seconds] -- VAR=value,VAR=value ...   (an empty variant '-' is the default build)
e.g.   python tools/enc_probe.py C3 120 -- - FLAC_AMD_SEARCH=generic"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import flac_amd
from flac_amd import workloads as W

name = sys.argv[1]
rest = sys.argv[2:]
seconds = None
if rest and rest[0] != "--":
    seconds = float(rest[0])
    rest = rest[1:]
variants = rest[1:] if rest and rest[0] == "--" else ["-"]
c = W.CONFIGS[name]
pcm = W.pcm(name, seconds=seconds)
wide = pcm.shape[0]
dev = torch.device("cuda", 0)
sample_bytes = 2 if c["bps"] <= 16 else 4
pcm_dev = torch.from_numpy(pcm.astype(np.int16) if sample_bytes == 2 else pcm).to(dev)
ref = None
for var in variants:
    sets = [kv.split("=", 1) for kv in var.split(",") if "=" in kv]
    for k, v in sets:
        os.environ[k] = v
    p = W.params(name, flac_amd)
    nframes = -(-wide // p.blocksize)
    enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
    cap = enc.max_output_bytes(nframes)
    out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
    offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
    enc.set_profiling(True)
    acc = {}
    t0 = None
    for it in range(6):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        if it:
            for k, v in enc.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 5
    ms = (time.perf_counter() - t0) / 5 * 1e3
    enc.status()
    total = int(offs[nframes].item())
    got = out[:total].clone()
    same = "first" if ref is None else ("same bytes" if ref.numel() == got.numel() and bool(torch.equal(ref, got)) else "BYTES DIFFER")
    if ref is None:
        ref = got
    print("%-40s %8.3f ms  %s  %d bytes (%s)" % (var, ms, " ".join("%s=%.3f" % (k, v) for k, v in acc.items() if v > 0.01), total, same), flush=True)
    enc.close()
    for k, _v in sets:
        os.environ.pop(k, None)
