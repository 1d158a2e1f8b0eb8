import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import flac_amd, oracle_lib as O
from test_gpu_search_forms import signals
from test_gpu_analysis_forms import encode
channels, bps = 2, 24
sigs = signals(bps, channels)
for bs in (4096, 16384):
    for level in (8, 5, 2, 0, 1):
        p = flac_amd.preset(level, channels, bps, 48000, bs)
        op = O.preset(level, channels, bps, 48000, bs)
        for name, pcm in sigs.items():
            pcm = np.ascontiguousarray(pcm[:bs * 3 + bs // 5 + 2])
            want, _ = O.encode_frames(op, pcm)
            for form in ("split", None):
                try:
                    got = encode(p, pcm, form)
                    r = "ok" if got == [bytes(w) for w in want] else "DIFF"
                except Exception as ex:
                    r = "ERR"
                if r != "ok":
                    print(bs, level, name, form, r, [len(w) for w in want], "verbatim", bs * 6 + 20)
print("done")
