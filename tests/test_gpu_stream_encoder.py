"""GPU parity tests of the libFLAC-compatible stream encoder API.

The same ctypes driver that produced tests/golden/ from the reference library (libflac_ctypes.py:
FLAC__stream_encoder_new / set_* / init_stream / process[_interleaved] / finish / delete, the call
sequence of src/test_libFLAC/encoders.c) is pointed at this repository's library.  The bytes that
reach the write callback (stream prologue after the finish()-time STREAMINFO rewrite, and every
frame) must equal what the reference wrote.
"""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util
import libflac_ctypes as L

pytestmark = pytest.mark.gpu

CASES = golden_util.load_index()
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flac_amd", "lib", "libflac_amd.so")


@pytest.fixture(scope="module")
def lib():
    return L.load(LIB)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_stream_matches_reference(lib, case):
    pcm, meta, frames = golden_util.load_case(case)
    got_meta, got_frames, ok, writes = L.encode(
        lib, pcm, case["bps"], case["rate"], level=case["level"], blocksize=case["blocksize"],
        settings=case["settings"], chunk=case["chunk"], planar=case["planar"], seekable=True)
    assert ok
    assert len(got_frames) == len(frames)
    for i, (a, b) in enumerate(zip(got_frames, frames)):
        assert a == b, "frame %d differs" % i
    assert got_meta == meta
    # the per-frame callback arguments: samples = blocksize of that frame, current_frame counts up
    fr = [(s, f) for _d, s, f in writes if s > 0]
    assert [f for _s, f in fr] == list(range(len(fr)))
    assert sum(s for s, _f in fr) == pcm.shape[0]


def test_small_batches_change_nothing(lib, monkeypatch):
    """Device batching is invisible: 3-frame batches give the same stream as one big batch."""
    monkeypatch.setenv("FLAC_AMD_BATCH_FRAMES", "3")
    case = next(c for c in CASES if c["name"] == "s16_l1_loose")
    pcm, meta, frames = golden_util.load_case(case)
    got_meta, got_frames, ok, _ = L.encode(lib, pcm, case["bps"], case["rate"], level=case["level"],
                                           chunk=1000, seekable=True)
    assert ok and got_frames == frames and got_meta == meta


def test_unseekable_prologue(lib):
    """Without seek/tell callbacks the STREAMINFO stays as first written (zero MD5, sizes, total)."""
    case = next(c for c in CASES if c["name"] == "s16_l5")
    pcm, meta, frames = golden_util.load_case(case)
    got_meta, got_frames, ok, _ = L.encode(lib, pcm, case["bps"], case["rate"], level=5, seekable=False)
    assert ok and got_frames == frames
    assert len(got_meta) == len(meta)
    assert got_meta[:4] == b"fLaC"
    # min/max blocksize, sample rate / channels / bps are written up front and agree
    assert got_meta[8:12] == meta[8:12]
    assert got_meta[18:21] == meta[18:21]
    # frame sizes, total samples and the MD5 stay zero
    assert got_meta[12:18] == b"\0" * 6
    assert got_meta[26:42] == b"\0" * 16
    # everything after STREAMINFO (the VORBIS_COMMENT block) is identical
    assert got_meta[42:] == meta[42:]


def test_verify_mode(lib):
    case = next(c for c in CASES if c["name"] == "s24_l8")
    pcm, meta, frames = golden_util.load_case(case)
    got_meta, got_frames, ok, _ = L.encode(lib, pcm, case["bps"], case["rate"], level=8,
                                           settings={"set_verify": 1}, seekable=True)
    assert ok and got_frames == frames and got_meta == meta


def test_caller_metadata_and_seektable_fill(lib):
    """set_metadata blocks: prologue order, vendor string, seek points filled in at finish()."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gen_golden
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metadata.json")) as f:
        want = {v["name"]: v for v in json.load(f)}
    by_name = {c["name"]: c for c in CASES}
    for name, cname, specs in gen_golden.METADATA_CASES:
        cs = by_name[cname]
        pcm, _meta, frames = golden_util.load_case(cs)
        meta, got_frames, ok, writes = L.encode(lib, pcm, cs["bps"], cs["rate"], level=cs["level"], blocksize=cs["blocksize"],
                                                settings=cs["settings"], seekable=True, metadata=specs)
        assert ok and got_frames == frames, name
        assert meta.hex() == want[name]["meta"], name
        assert [len(d) for d, smp, _f in writes if smp == 0] == want[name]["first_writes"], name


def test_getters_and_state(lib):
    u = C.c_uint
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    try:
        assert lib.FLAC__stream_encoder_get_state(enc) == 1  # UNINITIALIZED
        assert lib.FLAC__stream_encoder_set_channels(enc, u(2))
        assert lib.FLAC__stream_encoder_set_compression_level(enc, u(8))
        assert lib.FLAC__stream_encoder_get_max_lpc_order(enc) == 12
        assert lib.FLAC__stream_encoder_get_do_exhaustive_model_search(enc) == 1
        assert lib.FLAC__stream_encoder_get_max_residual_partition_order(enc) == 6
        sink = []
        wcb = L.WRITE_CB(lambda _e, buf, n, s, f, _c: sink.append(bytes(buf[:n])) or 0)
        st = lib.FLAC__stream_encoder_init_stream(enc, wcb, C.cast(None, L.SEEK_CB), C.cast(None, L.TELL_CB), None, None)
        assert st == 0
        assert lib.FLAC__stream_encoder_get_state(enc) == 0  # OK
        assert lib.FLAC__stream_encoder_get_blocksize(enc) == 4096
        # setters are refused once initialised; a second init reports ALREADY_INITIALIZED
        assert not lib.FLAC__stream_encoder_set_channels(enc, u(1))
        st2 = lib.FLAC__stream_encoder_init_stream(enc, wcb, C.cast(None, L.SEEK_CB), C.cast(None, L.TELL_CB), None, None)
        assert st2 == 13
        assert lib.FLAC__stream_encoder_finish(enc)
        assert lib.FLAC__stream_encoder_get_state(enc) == 1
        assert sink and sink[0] == b"fLaC"
    finally:
        lib.FLAC__stream_encoder_delete(enc)
