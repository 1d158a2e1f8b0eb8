#!/usr/bin/env python3
"""Where the stream API's time goes: the 600 s stream through FLAC__stream_encoder_process_interleaved with the digest off,
for several numbers of staging helper threads and two sinks (a tmpfs file, /dev/null).  usage: python tools/e2e_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench
import flac_amd

pcm = bench.make_pcm(1234, 600.0)
lib = flac_amd.lib()
lib.FLAC__stream_encoder_new.restype = C.c_void_p
wide, ch = pcm.shape
data = np.ascontiguousarray(pcm, dtype=np.int32)


def once(path, md5, chunk):
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    for name, v in (("channels", ch), ("bits_per_sample", 16), ("sample_rate", 44100), ("compression_level", 8)):
        getattr(lib, "FLAC__stream_encoder_set_" + name)(enc, C.c_uint(v))
    lib.FLAC__stream_encoder_set_total_samples_estimate(enc, C.c_uint64(wide))
    lib.FLAC__stream_encoder_set_do_md5(enc, C.c_int(md5))
    t0 = time.perf_counter()
    assert lib.FLAC__stream_encoder_init_file(enc, path.encode(), None, None) == 0
    t1 = time.perf_counter()
    for pos in range(0, wide, chunk):
        n = min(chunk, wide - pos)
        assert lib.FLAC__stream_encoder_process_interleaved(enc, data[pos:pos + n].ctypes.data_as(C.c_void_p), C.c_uint(n))
    t2 = time.perf_counter()
    assert lib.FLAC__stream_encoder_finish(enc)
    t3 = time.perf_counter()
    lib.FLAC__stream_encoder_delete(enc)
    return t1 - t0, t2 - t1, t3 - t2


for threads, batch in ((0, 0), (3, 0), (3, 128), (3, 256), (3, 512), (7, 256)):
    os.environ["FLAC_AMD_COPY_THREADS"] = str(threads)
    if batch:
        os.environ["FLAC_AMD_BATCH_FRAMES"] = str(batch)
    else:
        os.environ.pop("FLAC_AMD_BATCH_FRAMES", None)
    for path in ("/dev/shm/e2e_probe.flac", "/dev/null"):
        for chunk in (1 << 20,):
            once(path, 0, chunk)
            best = min((once(path, 0, chunk) for _ in range(3)), key=lambda r: r[1] + r[2])
            print("batch %4d copy threads %d  sink %-24s chunk %7d  init %.1f ms  process %.1f ms  finish %.1f ms  -> %.0f MSamples/s" % (
                batch, threads, path, chunk, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, wide * ch / (best[1] + best[2]) / 1e6), flush=True)
