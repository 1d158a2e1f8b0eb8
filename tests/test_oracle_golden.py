"""CPU tests: the oracle (oracle/flac_oracle.c) against vectors written by the real reference.

These pin the checker itself.  Every case must match the reference byte for byte: stream
prologue, every frame, and the decode of the reference's stream.
"""
import hashlib

import numpy as np
import pytest

import golden_util
import oracle_lib as O

CASES = golden_util.load_index()


def params_for(case):
    return O.preset(*golden_util.preset_args(case), **golden_util.field_overrides(case))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_encode_bytes_match_reference(case):
    pcm, meta, frames = golden_util.load_case(case)
    p = params_for(case)
    got, _ = O.encode_frames(p, pcm)
    assert len(got) == len(frames)
    for i, (a, b) in enumerate(zip(got, frames)):
        assert a == b, "frame %d differs (%d vs %d bytes)" % (i, len(a), len(b))
    stream = O.encode_stream(p, pcm)
    assert stream[:len(meta)] == meta
    assert hashlib.sha256(stream).hexdigest() == case["stream_sha256"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_decode_reference_stream(case):
    pcm, meta, frames = golden_util.load_case(case)
    si = O.parse_header(meta + b"".join(frames))
    assert si.channels == case["channels"] and si.bits_per_sample == case["bps"]
    assert si.total_samples == pcm.shape[0]
    assert si.audio_offset == len(meta)
    assert bytes(si.md5) == O.md5_pcm(pcm, case["bps"])
    n, out, nf = O.decode_frames(b"".join(frames), case["channels"], case["bps"], case["rate"], pcm.shape[0])
    assert n == pcm.shape[0] and nf == len(frames)
    assert np.array_equal(out, pcm)


def test_decode_detects_corruption():
    case = next(c for c in CASES if c["name"] == "s16_l8")
    pcm, _meta, frames = golden_util.load_case(case)
    blob = bytearray(b"".join(frames))
    blob[len(frames[0]) // 2] ^= 0x10
    rc, _out, nf = O.decode_frames(bytes(blob), 2, 16, 44100, pcm.shape[0])
    assert rc < 0 and nf == 0
