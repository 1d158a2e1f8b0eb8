"""The reference decoder accepts every full-size stream the GPU encoder writes, and the bytes are the reference's.

Each benchmark configuration is encoded at full size, on the signal the benchmark itself uses
(flac_amd.workloads: synth_pcm, seed 1234), through the libFLAC-compatible stream API of this library
(seekable, so that finish() writes the MD5 and the totals into STREAMINFO), and the whole stream is
then handed to the REFERENCE's stream decoder (oracle/_ref/libFLAC_ref.so, compiled from the
reference's sources in the build container) with MD5 checking on: no error callback, every sample
equal to the input, finish() true - which is what `flac -t` checks.

Bytes: all three streams - the 600 s `-8` stream (the bench line's own workload), the 600 s `-8 -e -p`
stream of 24 bits at 96 kHz and the one-minute eight-channel stream - are compared whole, prologue
included, with what the reference encoder writes for the same samples.  The reference encodes run in
worker processes while the GPU does its part (the `-8 -e -p` one is a good minute of one core).
"""
import multiprocessing as mp
import os

import numpy as np
import pytest

import libflac_ctypes as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")

# name, seconds of the stream the reference encoder is given (None: all of it)
CONFIGS = [("C2", None), ("C4", None), ("C3", None)]   # (the long reference encode is waited for last)


def _reference_encode(args):
    """Worker process: the reference encoder on the configuration's signal."""
    import sys
    sys.path.insert(0, ROOT)
    from flac_amd import workloads as W
    name, seconds = args
    cfg = W.CONFIGS[name]
    pcm = W.pcm(name, seconds=seconds)
    ref = L.load(REF_LIB)
    meta, frames, ok, _ = L.encode(ref, pcm, cfg["bps"], cfg["rate"], level=cfg["level"], settings=cfg["settings"], chunk=1 << 20,
                                   seekable=True, total_samples_estimate=pcm.shape[0])
    return meta, frames, ok


@pytest.fixture(scope="module")
def reference_streams():
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libFLAC_ref.so has not been built")
    pool = mp.get_context("spawn").Pool(len(CONFIGS))
    jobs = {name: pool.apply_async(_reference_encode, ((name, seconds),)) for name, seconds in CONFIGS}
    yield jobs
    pool.terminate()
    pool.join()


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libFLAC_ref.so has not been built")
    import flac_amd
    return L.load(flac_amd.LIB_PATH), L.load(REF_LIB)


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_reference_decoder_accepts_the_full_size_stream_and_the_bytes_are_its_own(libs, reference_streams, cfg):
    from flac_amd import workloads as W
    gpu, ref = libs
    name, ref_seconds = cfg
    c = W.CONFIGS[name]
    pcm = W.pcm(name)
    wide = pcm.shape[0]
    meta, frames, ok, _writes = L.encode(gpu, pcm, c["bps"], c["rate"], level=c["level"], settings=c["settings"], chunk=1 << 20, seekable=True,
                                         total_samples_estimate=wide)
    assert ok and len(frames) == -(-wide // 4096)
    assert meta[26:42] != bytes(16), "STREAMINFO carries no MD5"
    stream = meta + b"".join(frames)
    got, errors, fine = L.decode(ref, stream, md5_check=True)
    assert errors == []
    assert fine, "the reference decoder's finish() failed: MD5 mismatch"
    assert got.shape == pcm.shape and np.array_equal(got, pcm)
    del got
    rmeta, rframes, rok = reference_streams[name].get(timeout=900)
    assert rok
    if ref_seconds is None:
        assert len(rframes) == len(frames)
        gbytes, rbytes = sum(map(len, frames)), sum(map(len, rframes))
        assert gbytes == rbytes, "size differs from the reference's by %+.4f %%" % (100.0 * (gbytes - rbytes) / rbytes)
        assert b"".join(rframes) == b"".join(frames), "frame bytes differ from the reference encoder's"
        assert rmeta == meta, "stream prologue (STREAMINFO with MD5, sizes; vendor comment) differs"
    else:
        # the slice's last block is short, and coded differently from the full block the long stream has there
        n = (ref_seconds * c["rate"]) // 4096
        assert n > 1000 and len(rframes) >= n
        gbytes, rbytes = sum(map(len, frames[:n])), sum(map(len, rframes[:n]))
        assert gbytes == rbytes, "size of the first %d frames differs from the reference's by %+.4f %%" % (n, 100.0 * (gbytes - rbytes) / rbytes)
        assert b"".join(rframes[:n]) == b"".join(frames[:n]), "frame bytes differ from the reference encoder's"
