import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import flac_amd, oracle_lib as O
from test_gpu_pack_forms import encode
channels, bps, bs = 8, 16, 65535
rng = np.random.default_rng(bs + channels)
full = 1 << (bps - 1)
n = bs * 2 + 999
t = np.arange(n, dtype=np.float64)
music = np.stack([0.5 * np.sin(2 * np.pi * (0.003 + 0.0017 * c) * t + c) + 0.1 * np.sin(2 * np.pi * 0.041 * t) for c in range(channels)], axis=1)
sigs = {"music": (music * full * 0.9 + rng.normal(0, full / 300.0, size=(n, channels))), "loud": (music * full * 0.5 + rng.normal(0, full / 8.0, size=(n, channels))),
        "noise": rng.integers(-full, full, size=(n, channels)).astype(np.float64)}
for level in (8, 5):
    p = flac_amd.preset(level, channels, bps, 96000, bs)
    op = O.preset(level, channels, bps, 96000, bs)
    e = flac_amd.Encoder(p, device=0, max_frames=8)
    print("slot bytes per frame", e.max_output_bytes(1)); e.close()
    for name, x in sigs.items():
        pcm = np.ascontiguousarray(np.clip(x, -full, full - 1).astype(np.int32))
        want, _ = O.encode_frames(op, pcm)
        try:
            got = encode(p, pcm, None)
            print(level, name, [len(w) for w in want], "ok" if [bytes(w) for w in want] == got else "DIFF")
        except Exception as ex:
            print(level, name, [len(w) for w in want], "ERR", str(ex)[:80])
