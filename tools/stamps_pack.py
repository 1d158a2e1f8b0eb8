#!/usr/bin/env python3
"""Phase timeline of k_pack workgroups from in-kernel cycle stamps (FLAC_AMD_STAMPS=pack).

Stamps: 0 start, 1 entry loads in LDS, 2 frame header out, then per written channel c (0/1): 3+4c subframe header
out, 4+4c residual and code lengths known, 5+4c block scan done, 6+4c codes in the window; 11 both channels done,
12 CRC-16 known, 13 frame flushed to its slot.
"""
import os
import sys

os.environ["FLAC_AMD_STAMPS"] = "pack"
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
raw = enc.debug_read(1, nfr * 128).view(np.uint64).reshape(nfr, 16)
t = raw[:-1, :14].astype(np.int64)   # the last frame is short
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]
names = ["entry loads", "frame header", "ch0 subframe hdr", "ch0 residual+lengths", "ch0 scan", "ch0 emit",
         "ch1 subframe hdr", "ch1 residual+lengths", "ch1 scan", "ch1 emit", "(loop exit)", "crc16", "flush"]
d = np.diff(t[:, order], axis=1)
ok = (d >= 0).all(axis=1) & (d < 10_000_000).all(axis=1)
d = d[ok]
print("frames %d (%d usable); cycles per phase: median / mean / p90" % (nfr, len(d)))
for i, n in enumerate(names):
    print("  %-22s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
tot = t[ok][:, 13] - t[ok][:, 0]
print("  %-22s %8.0f %8.0f %8.0f" % ("total", np.median(tot), tot.mean(), np.percentile(tot, 90)))
enc.close()
