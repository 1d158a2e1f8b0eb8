"""GPU parity tests of the libFLAC-compatible stream decoder API.

The ctypes driver that decodes through the reference library (libflac_ctypes.py: the call sequence
of src/test_libFLAC/decoders.c) is pointed at this repository's library.  Clean streams must come
back bit-exact with the STREAMINFO MD5 accepted; for damaged and oddly framed streams the blocks
handed to the write callback, the error callbacks and the return values are compared with what the
reference reported for the same bytes (tests/golden/damaged.json, made by oracle/gen_golden.py).
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import golden_util
import libflac_ctypes as L

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = golden_util.load_index()
BY_NAME = {c["name"]: c for c in CASES}
LIB = os.path.join(ROOT, "flac_amd", "lib", "libflac_amd.so")
with open(os.path.join(ROOT, "tests", "golden", "damaged.json")) as _f:
    DAMAGED = json.load(_f)


@pytest.fixture(scope="module")
def lib():
    return L.load(LIB)


def stream_of(case):
    pcm, meta, frames = golden_util.load_case(case)
    return pcm, meta, frames, meta + b"".join(frames)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_clean_stream(lib, case):
    pcm, _meta, _frames, stream = stream_of(case)
    out, errors, ok = L.decode(lib, stream)
    assert errors == []
    assert ok, "process or finish (MD5 check) failed"
    assert out.shape == pcm.shape and np.array_equal(out, pcm)


def test_small_reads_and_small_batches(lib, monkeypatch):
    """Neither the size of what a read callback returns nor the device batch size shows."""
    monkeypatch.setenv("FLAC_AMD_DEC_BATCH_SAMPLES", "20000")
    pcm, _m, _f, stream = stream_of(BY_NAME["s16_l1_loose"])
    out, errors, ok = L.decode(lib, stream, max_read=777)
    assert errors == [] and ok and np.array_equal(out, pcm)


def test_write_callback_header_and_getters(lib):
    case = BY_NAME["s24_l5"]
    pcm, _m, frames, stream = stream_of(case)
    sd = L.StreamDecoder(lib, stream, respond_all=True)
    try:
        assert sd.init_status == 0
        assert sd.state() == 0  # SEARCH_FOR_METADATA
        assert sd.call("process_until_end_of_metadata")
        assert sd.state() == 2  # SEARCH_FOR_FRAME_SYNC
        assert sd.metadata == [(0, 0, 34), (4, 1, 40)]
        assert sd.call("get_total_samples") == pcm.shape[0]
        assert sd.call("process_single")
        assert len(sd.blocks) == 1
        bs, rate, ch, _assign, bps, number_type, number, crc = sd.last_header
        assert (bs, rate, ch, bps, number_type, number) == (4096, 96000, 2, 24, 1, 0)
        assert crc == frames[0][5]  # 4 fixed bytes + 1 number byte, then the CRC-8
        assert sd.call("get_channels") == 2 and sd.call("get_bits_per_sample") == 24
        assert sd.call("get_sample_rate") == 96000 and sd.call("get_blocksize") == 4096
        assert sd.call("process_until_end_of_stream")
        assert sd.state() == 4  # END_OF_STREAM
        assert np.array_equal(sd.pcm(), pcm)
        assert sd.call("finish")
        assert sd.state() == 9  # UNINITIALIZED
    finally:
        sd.close()


def test_metadata_filters(lib):
    _pcm, _m, _f, stream = stream_of(BY_NAME["m16_l2"])
    sd = L.StreamDecoder(lib, stream)  # default: STREAMINFO only
    try:
        assert sd.call("process_until_end_of_metadata")
        assert [t for t, _l, _n in sd.metadata] == [0]
    finally:
        sd.close()
    sd = L.StreamDecoder(lib, stream)
    try:
        # refused after init
        assert not sd.call("set_metadata_ignore_all")
    finally:
        sd.close()
    dec = C.c_void_p(lib.FLAC__stream_decoder_new())
    try:
        assert lib.FLAC__stream_decoder_set_metadata_ignore_all(dec)
        assert lib.FLAC__stream_decoder_set_metadata_respond(dec, C.c_int(4))
        assert not lib.FLAC__stream_decoder_set_metadata_respond(dec, C.c_int(127))
    finally:
        lib.FLAC__stream_decoder_delete(dec)


def test_skip_single_frame(lib):
    pcm, _m, _f, stream = stream_of(BY_NAME["s16_l2_bs1152x"])
    sd = L.StreamDecoder(lib, stream, md5_check=False)
    try:
        assert sd.call("process_until_end_of_metadata")
        assert sd.call("skip_single_frame")
        assert sd.blocks == []
        assert sd.call("process_single")
        assert sd.blocks[0][0] == 1152 and np.array_equal(sd.blocks[0][1], pcm[1152:2304])
    finally:
        sd.close()


def test_write_abort(lib):
    _pcm, _m, _f, stream = stream_of(BY_NAME["s16_l2_bs1152x"])
    sd = L.StreamDecoder(lib, stream, md5_check=False)
    try:
        sd.abort_writes = True
        assert not sd.call("process_until_end_of_stream")
        assert len(sd.blocks) == 1
    finally:
        sd.close()


@pytest.mark.parametrize("name", ["s16_l5", "s16_l1_loose", "s24_l8", "s16_l5_bs256", "c8_24_l0"])
def test_seek_absolute(lib, name):
    pcm, _m, _f, stream = stream_of(BY_NAME[name])
    n = pcm.shape[0]
    sd = L.StreamDecoder(lib, stream, seekable=True)
    try:
        assert sd.init_status == 0
        for target in (n // 2, 1, n - 1, 0, n // 3 + 7):
            sd.blocks.clear()
            assert sd.call("seek_absolute", C.c_uint64(target)), "seek to %d" % target
            first, blk = sd.blocks[0]
            assert first == target
            assert np.array_equal(blk, pcm[target:target + blk.shape[0]])
            # decoding carries on from there
            assert sd.call("process_single")
            if len(sd.blocks) > 1:
                nxt, blk2 = sd.blocks[1]
                assert nxt == target + blk.shape[0]
                assert np.array_equal(blk2, pcm[nxt:nxt + blk2.shape[0]])
        assert not sd.call("seek_absolute", C.c_uint64(n))  # past the end
        assert sd.errors == []
    finally:
        sd.close()


def test_flush_and_reset(lib):
    pcm, _m, _f, stream = stream_of(BY_NAME["s16_l2_bs1152x"])
    sd = L.StreamDecoder(lib, stream, seekable=True)
    try:
        assert sd.call("process_until_end_of_stream")
        assert np.array_equal(sd.pcm(), pcm)
        sd.blocks.clear()
        assert sd.call("reset")
        assert sd.state() == 0
        assert sd.call("process_until_end_of_stream")
        assert np.array_equal(sd.pcm(), pcm)
        assert sd.call("finish")  # MD5 checking is back on after reset, and matches
    finally:
        sd.close()


def test_decode_position(lib):
    _pcm, meta, frames, stream = stream_of(BY_NAME["s16_l2_bs1152x"])
    sd = L.StreamDecoder(lib, stream, seekable=True)
    try:
        pos = C.c_uint64(0)
        assert sd.call("process_until_end_of_metadata")
        assert sd.call("get_decode_position", C.byref(pos)) and pos.value == len(meta)
        assert sd.call("process_single")
        assert sd.call("get_decode_position", C.byref(pos)) and pos.value == len(meta) + len(frames[0])
    finally:
        sd.close()


@pytest.mark.parametrize("vec", DAMAGED, ids=[v["name"] for v in DAMAGED])
def test_damaged_stream_matches_reference(lib, vec):
    import gen_golden
    _pcm, meta, frames = golden_util.load_case(BY_NAME[vec["case"]])
    stream = gen_golden.apply_damage(meta, frames, tuple(vec["edit"]))
    assert hashlib.sha256(stream).hexdigest() == vec["stream_sha256"]
    sd = L.StreamDecoder(lib, stream, md5_check=True)
    try:
        assert sd.init_status == 0
        ok = bool(sd.call("process_until_end_of_stream"))
        state = sd.state()
        fin = bool(sd.call("finish"))
    finally:
        sd.close()
    got_blocks = [[n, int(b.shape[0]), hashlib.sha256(np.ascontiguousarray(b).tobytes()).hexdigest()] for n, b in sd.blocks]
    assert got_blocks == vec["blocks"]
    assert sd.errors == vec["errors"]
    assert (ok, state, fin) == (vec["process_ok"], vec["state"], vec["finish_ok"])


def test_file_api(lib, tmp_path):
    pcm, _m, _f, stream = stream_of(BY_NAME["s16_l5"])
    path = tmp_path / "a.flac"
    path.write_bytes(stream)
    blocks = []

    def on_write(_d, frame, buffers, _c):
        hdr = C.cast(frame, C.POINTER(L.FrameHeader)).contents
        blk = np.empty((hdr.blocksize, hdr.channels), dtype=np.int32)
        for c in range(hdr.channels):
            blk[:, c] = np.ctypeslib.as_array(buffers[c], shape=(hdr.blocksize,))
        blocks.append(blk)
        return 0

    wcb, ecb = L.DEC_WRITE_CB(on_write), L.DEC_ERROR_CB(lambda _d, s, _c: None)
    dec = C.c_void_p(lib.FLAC__stream_decoder_new())
    try:
        assert lib.FLAC__stream_decoder_set_md5_checking(dec, C.c_int(1))
        assert lib.FLAC__stream_decoder_init_file(dec, str(path).encode(), wcb, None, ecb, None) == 0
        assert lib.FLAC__stream_decoder_process_until_end_of_stream(dec)
        assert lib.FLAC__stream_decoder_finish(dec)
        assert np.array_equal(np.concatenate(blocks), pcm)
        assert lib.FLAC__stream_decoder_init_file(dec, b"/nonexistent/x.flac", wcb, None, ecb, None) == 4
        assert lib.FLAC__stream_decoder_init_ogg_file(dec, str(path).encode(), wcb, None, ecb, None) == 1
    finally:
        lib.FLAC__stream_decoder_delete(dec)


def _long_stream(lib, seconds=40, seektable=None):
    from flac_amd.synth import synth_pcm
    pcm = synth_pcm(seconds=seconds, sample_rate=44100, channels=2, bits_per_sample=16, seed=77)
    specs = [("seektable", seektable)] if seektable else None
    meta, frames, ok, _w = L.encode(lib, pcm, 16, 44100, level=5, chunk=1 << 18, seekable=True, total_samples_estimate=pcm.shape[0],
                                    metadata=specs)
    assert ok
    return pcm, meta, frames


@pytest.mark.parametrize("variant", ["plain", "seektable", "sparse_seektable", "unknown_total"])
def test_random_seeks(lib, variant):
    """What the reference's test_seeking does (src/test_seeking/main.c): many seeks to random samples, with and without a
    seek table and with a STREAMINFO that does not know the stream's length; every seek must land exactly and decoding
    must carry on from there."""
    rng = np.random.default_rng(9)
    n_est = 40 * 44100
    table = None
    if variant == "seektable":
        table = list(range(0, n_est, 44100 * 2))
    elif variant == "sparse_seektable":
        table = [0, n_est // 2 + 12345]
    pcm, meta, frames = _long_stream(lib, seektable=table)
    n = pcm.shape[0]
    if variant == "unknown_total":
        # STREAMINFO: the 36-bit total sample count (low 4 bits of byte 21 and bytes 22-25 of the block body) set to zero
        m = bytearray(meta)
        m[8 + 13] &= 0xf0
        m[8 + 14:8 + 18] = bytes(4)
        meta = bytes(m)
    stream = meta + b"".join(frames)
    sd = L.StreamDecoder(lib, stream, seekable=True)
    seeks = [0]
    plain_seek = sd._seek

    def counting_seek(d, off, c):
        seeks[0] += 1
        return plain_seek(d, off, c)
    try:
        assert sd.init_status == 0
        targets = [int(t) for t in rng.integers(0, n, size=60)] + [0, n - 1, n - 4096, 4095, 4096, 4097]
        for target in targets:
            sd.blocks.clear()
            assert sd.call("seek_absolute", C.c_uint64(target)), "seek to %d (%s)" % (target, variant)
            first, blk = sd.blocks[0]
            assert first == target
            assert np.array_equal(blk, pcm[target:target + blk.shape[0]])
            if target + blk.shape[0] < n:
                assert sd.call("process_single")
                nxt, blk2 = sd.blocks[1]
                assert nxt == target + blk.shape[0] and np.array_equal(blk2, pcm[nxt:nxt + blk2.shape[0]])
        if variant != "unknown_total":
            assert not sd.call("seek_absolute", C.c_uint64(n))
        else:
            # nobody knows where the stream ends: the hunt runs off the end and reports a seek error, which a flush clears
            assert not sd.call("seek_absolute", C.c_uint64(n + 5000))
            assert sd.state() == 6  # FLAC__STREAM_DECODER_SEEK_ERROR
            assert sd.call("flush")
            sd.blocks.clear()
            assert sd.call("seek_absolute", C.c_uint64(1000)) and sd.blocks[0][0] == 1000
        assert sd.errors == []
    finally:
        sd.close()


@pytest.mark.parametrize("order", ["silence_then_noise", "noise_then_silence"])
@pytest.mark.parametrize("known_total", [True, False])
def test_seeks_in_a_stream_of_very_uneven_density(lib, order, known_total):
    """Half of the stream is digital silence (a few bytes per frame), the other half full-scale noise (verbatim frames),
    no seek table: a place worked out from the average density is wrong by almost half the stream, so the hunt has to
    bracket the target as the reference's does (stream_decoder.c:2973ff) instead of aiming from one global figure."""
    rng = np.random.default_rng(5)
    half = 44100 * 12
    noise = rng.integers(-30000, 30000, size=(half, 2)).astype(np.int32)
    quiet = np.zeros((half, 2), dtype=np.int32)
    pcm = np.concatenate([quiet, noise] if order == "silence_then_noise" else [noise, quiet])
    n = pcm.shape[0]
    meta, frames, ok, _w = L.encode(lib, pcm, 16, 44100, level=5, chunk=1 << 18, seekable=True, total_samples_estimate=n)
    assert ok
    if not known_total:
        m = bytearray(meta)
        m[8 + 13] &= 0xf0
        m[8 + 14:8 + 18] = bytes(4)
        meta = bytes(m)
    stream = meta + b"".join(frames)
    sd = L.StreamDecoder(lib, stream, seekable=True)
    try:
        assert sd.init_status == 0
        targets = [int(t) for t in rng.integers(0, n, size=24)] + [half - 1, half, half + 1, 5, n - 5, half // 2, half + half // 2]
        for target in targets:
            sd.blocks.clear()
            assert sd.call("seek_absolute", C.c_uint64(target)), "seek to %d" % target
            first, blk = sd.blocks[0]
            assert first == target
            assert np.array_equal(blk, pcm[target:target + blk.shape[0]])
        assert sd.errors == []
    finally:
        sd.close()
