#!/usr/bin/env python3
"""Phase timeline of k_search workgroups from in-kernel cycle stamps (FLAC_AMD_STAMPS=1).

Stamps: 0 start, 1 entry loads landed in LDS, 2 candidates gathered, 3 tables filled, 4 samples in registers,
5 residual loop done (thread 0), 6 all threads past the loop, 7 Rice estimates done, 8 winner kept, 9 plan written.
"""
import os
import sys

os.environ["FLAC_AMD_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np

import flac_amd
from flac_amd.synth import synth_pcm

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
pcm = synth_pcm(seconds=seconds, seed=1234)
p = flac_amd.preset(8, 2, 16, 44100)
nfr = (pcm.shape[0] + p.blocksize - 1) // p.blocksize
import torch
dev = torch.device("cuda", 0)
pcm_dev = torch.from_numpy(pcm.astype(np.int16)).to(dev)
enc = flac_amd.Encoder(p, 0, nfr)
out_cap = enc.max_output_bytes(nfr)
out_dev = torch.empty(out_cap, dtype=torch.uint8, device=dev)
fb = torch.zeros(nfr, dtype=torch.int32, device=dev)
offs = torch.zeros(nfr + 1, dtype=torch.int64, device=dev)
for _ in range(3):
    enc.encode_device(pcm_dev.data_ptr(), 2, pcm.shape[0], out_dev.data_ptr(), out_cap, fb.data_ptr(), offs.data_ptr(),
                      first_frame=0, is_last=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
ntasks = nfr * 4
raw = enc.debug_read(1, ntasks * 128).view(np.uint64).reshape(ntasks, 16)
t = raw[:, :10].astype(np.int64)
d = np.diff(t, axis=1)
names = ["entry loads", "gather", "tables", "chunk+zero", "residuals(t0)", "wait others", "rice est", "winner", "plan out"]
ok = (d >= 0).all(axis=1) & (d < 10_000_000).all(axis=1)
d = d[ok]
print("tasks %d (%d usable); cycles per phase: median / mean / p90" % (ntasks, len(d)))
for i, n in enumerate(names):
    print("  %-14s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
tot = t[ok][:, 9] - t[ok][:, 0]
print("  %-14s %8.0f %8.0f %8.0f" % ("total", np.median(tot), tot.mean(), np.percentile(tot, 90)))
span = t[ok][:, 9].max() - t[ok][:, 0].min()
print("first start to last end: %d cycles" % span)
enc.close()
