import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import flac_amd, oracle_lib as O
from test_gpu_search_forms import signals
from test_gpu_pack_forms import encode
ch, bps, bs, level = 8, 24, 4096, 5
sigs = signals(bps, ch)
pcm = np.ascontiguousarray(sigs["music"][:bs * 3 + bs // 5])
p = flac_amd.preset(level, ch, bps, 48000, bs)
op = O.preset(level, ch, bps, 48000, bs)
want, _ = O.encode_frames(op, pcm)
a = encode(p, pcm, "generic"); b = encode(p, pcm, None)
for i in range(len(want)):
    w, x, y = bytes(want[i]), a[i], b[i]
    print(i, len(w), len(x), len(y), "generic ok" if w == x else "generic BAD", "sub ok" if w == y else "sub BAD")
    for nm, z in (("generic", x), ("sub", y)):
        if z != w and len(z) == len(w):
            d = [k for k in range(len(w)) if w[k] != z[k]]
            print("  ", nm, "differs at", len(d), "bytes, first", d[:12], "last", d[-4:])
