import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import flac_amd, oracle_lib as O
from test_gpu_search_forms import signals
channels, bps, bs, level = 2, 24, 16384, 2
sigs = signals(bps, channels)
pcm = np.ascontiguousarray(sigs["square"][:bs * 3 + bs // 5 + 2])
p = flac_amd.preset(level, channels, bps, 48000, bs)
op = O.preset(level, channels, bps, 48000, bs)
want, _ = O.encode_frames(op, pcm)
print("oracle frame bytes", [len(w) for w in want], "params: ms", p.do_mid_side_stereo, "max_po", p.max_residual_partition_order, "min_po", p.min_residual_partition_order)
enc = flac_amd.Encoder(p, device=0, max_frames=8)
try:
    frames, _ = enc.encode_host(pcm)
    print("gpu ok", [len(f) for f in frames])
except Exception as ex:
    print("ERR", ex)
plans = enc.debug_read(4, 2 * 4 * 400)
st = enc.debug_read(0, 2 * 4 * 40).view(np.uint32).reshape(8, 10)
asg = enc.debug_read(5, 8).view(np.uint32)
fb = enc.debug_read(6, 8).view(np.uint32)
print("assign", asg, "frame_bytes", fb)
for i in range(8):
    pl = plans[i * 400:(i + 1) * 400]
    bits = pl[:4].view(np.uint32)[0]
    print("frame %d sig %d: bits %d (%d bytes) type %d order %d po %d rice2 %d wide %d | wasted %d sbps %d guess %d | params %s" % (i // 4, i % 4, bits, bits // 8, pl[4], pl[5], pl[8], pl[9], pl[10], st[i, 0], st[i, 1], st[i, 2], list(pl[16 + 128:16 + 128 + (1 << pl[8])])))
enc.close()
