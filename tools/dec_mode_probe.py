#!/usr/bin/env python3
"""Batch decode time of the decoder's forms against the shape of the batch.  A variant is a FLAC_AMD_DEC_MODE value
(lane / split), optionally followed by frames per wave and frames per wave of the split form's walk:
usage: python tools/dec_mode_probe.py lane split split,8 split,16,4 ..."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import flac_amd
from test_gpu_configs_full_size import synth

variants = sys.argv[1:] or ["lane", "split"]
dev = torch.device("cuda", 0)
for channels, bps, rate, counts in ((2, 16, 44100, (256, 1024, 4096, 6460, 16384)), (2, 24, 96000, (14063,)), (8, 24, 192000, (256, 2813))):
    p = flac_amd.preset(8, channels, bps, rate)
    bs = p.blocksize
    sample_bytes = 2 if bps <= 16 else 4
    for nframes in counts:
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
        line = []
        for var in variants:
            parts = var.split(",")
            os.environ["FLAC_AMD_DEC_MODE"] = parts[0]
            os.environ.pop("FLAC_AMD_DEC_LW", None)
            os.environ.pop("FLAC_AMD_DEC_SKIP_LW", None)
            if len(parts) > 1:
                os.environ["FLAC_AMD_DEC_LW"] = parts[1]
            if len(parts) > 2:
                os.environ["FLAC_AMD_DEC_SKIP_LW"] = parts[2]
            dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=total + 64, max_frames=nframes + 8)
            dec.set_profiling(True)
            back = torch.empty_like(pcm_dev)

            def run():
                dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide)], back.data_ptr(), sample_bytes, wide)
            run()
            torch.cuda.synchronize()
            fr = []
            t0 = time.perf_counter()
            for _ in range(6):
                run()
                torch.cuda.synchronize()
                fr.append(dec.stage_ms()["frames"])
            ms = (time.perf_counter() - t0) / 6 * 1e3
            assert os.environ.get("FLAC_AMD_PROBE_NOCHECK") or (torch.equal(back, pcm_dev) and dec.result() == (nframes, wide, 0))
            dec.close()
            line.append("%s %.3f (frames %.3f)" % (var, ms, sorted(fr)[len(fr) // 2]))
        print("%d ch %2d bit  frames %5d   %s" % (channels, bps, nframes, "   ".join(line)), flush=True)
