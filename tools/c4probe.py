import json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import flac_amd
from test_gpu_configs_full_size import synth
dev = torch.device("cuda", 0)
c = dict(level=8, channels=8, bps=24, rate=192000, seconds=60)
wide = c["seconds"] * c["rate"]
pcm = synth(99, wide, c["channels"], c["bps"], c["rate"])
p = flac_amd.preset(c["level"], c["channels"], c["bps"], c["rate"])
pcm_dev = torch.from_numpy(pcm).to(dev)
nframes = -(-wide // p.blocksize)
enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
cap = enc.max_output_bytes(nframes)
out = torch.empty(cap, dtype=torch.uint8, device=dev)
sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
enc.encode_device(pcm_dev.data_ptr(), 4, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
torch.cuda.synchronize()
total = int(offs[nframes].item())
dec = flac_amd.Decoder(c["channels"], c["bps"], p.blocksize, device=0, max_input_bytes=total + 64, max_frames=nframes + 8)
back = torch.empty_like(pcm_dev)
for _ in range(4):
    dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide)], back.data_ptr(), 4, wide)
torch.cuda.synchronize()
print(dec.result(), bool(torch.equal(back, pcm_dev)))
