import sys, time, hashlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import flac_amd, oracle_lib as O
from flac_amd.synth import synth_pcm
pcm = synth_pcm(seconds=30.0, seed=1234)
for level in (0,5,8):
    p = flac_amd.preset(level, 2, 16, 44100)
    nfr = (pcm.shape[0] + p.blocksize - 1)//p.blocksize
    enc = flac_amd.Encoder(p, 0, nfr)
    enc.set_profiling(True)
    t=time.time(); frames,_ = enc.encode_host(pcm); dt=time.time()-t
    ms = enc.stage_ms()
    op = O.preset(level, 2, 16, 44100)
    t=time.time(); of,_ = O.encode_frames(op, pcm); dto=time.time()-t
    print("level", level, "frames", len(frames), "bytes", sum(map(len,frames)), "match", frames==of, "host call %.3fs oracle %.3fs"%(dt,dto))
    print("  stages ms:", {k: round(v,3) for k,v in ms.items()}, "sum", round(sum(ms.values()),3))
    enc.close()
