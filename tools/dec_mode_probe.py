#!/usr/bin/env python3
"""Batch decode time of the two decoder forms (a lane per frame / a wave per frame) against the number of frames in the batch:
`flac -8` 16-bit stereo and eight-channel 24-bit.  usage: python tools/dec_mode_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import flac_amd
from test_gpu_configs_full_size import synth

dev = torch.device("cuda", 0)
for channels, bps, rate in ((2, 16, 44100), (8, 24, 192000)):
    p = flac_amd.preset(8, channels, bps, rate)
    bs = p.blocksize
    sample_bytes = 2 if bps <= 16 else 4
    for nframes in (16, 54, 256, 1024, 2048, 4096, 8192):
        if nframes * channels > 20000:
            continue
        wide = nframes * bs
        pcm = synth(5, wide, channels, bps, rate)
        host = pcm.astype(np.int16) if sample_bytes == 2 else pcm
        pcm_dev = torch.from_numpy(host).to(dev)
        enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
        cap = enc.max_output_bytes(nframes)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
        sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
        enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        enc.close()
        total = int(offs[nframes].item())
        res = {}
        for mode in ("lane", "coop"):
            os.environ["FLAC_AMD_DEC_MODE"] = mode
            dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=total + 64, max_frames=nframes + 8)
            back = torch.empty_like(pcm_dev)

            def run():
                dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide)], back.data_ptr(), sample_bytes, wide)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / 10 * 1e3
            assert torch.equal(back, pcm_dev) and dec.result() == (nframes, wide, 0)
            dec.close()
        print("%d ch %2d bit  frames %5d   lane %7.3f ms   coop %7.3f ms" % (channels, bps, nframes, res["lane"], res["coop"]), flush=True)
