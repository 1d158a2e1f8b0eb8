"""CPU tests of the libFLAC-compatible stream layer: everything that happens before the first
frame needs the GPU.  Settings and their validation order (init_stream_internal_,
src/libFLAC/stream_encoder.c:660-759), the apodization parser (:1556-1660), format helpers
(format.c:213-314), the decoder's metadata phase on the reference's golden streams, and the loud
failure when a frame has to be coded and no device is there."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util
import libflac_ctypes as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("FLAC_TEST_LIB") or os.path.join(ROOT, "flac_amd", "lib", "libflac_amd.so")
u = C.c_uint


@pytest.fixture(scope="module")
def lib():
    return L.load(LIB)


def has_gpu():
    import torch
    return torch.cuda.is_available()


def new_encoder(lib, **settings):
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    for k, v in settings.items():
        assert getattr(lib, "FLAC__stream_encoder_set_" + k)(enc, u(v)), k
    return enc


def init(lib, enc):
    wcb = L.WRITE_CB(lambda *_a: 0)
    st = lib.FLAC__stream_encoder_init_stream(enc, wcb, C.cast(None, L.SEEK_CB), C.cast(None, L.TELL_CB), None, None)
    return st, wcb


def test_fresh_encoder_settings(lib):
    """What FLAC__stream_encoder_new() really leaves behind in this reference (preset 0, not 5)."""
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    try:
        g = lambda n: getattr(lib, "FLAC__stream_encoder_get_" + n)(enc)
        assert g("state") == 1
        assert (g("channels"), g("bits_per_sample"), g("sample_rate"), g("blocksize")) == (2, 16, 44100, 0)
        assert (g("verify"), g("streamable_subset"), g("do_mid_side_stereo"), g("loose_mid_side_stereo")) == (0, 1, 0, 0)
        assert (g("max_lpc_order"), g("qlp_coeff_precision"), g("do_exhaustive_model_search")) == (0, 0, 0)
        assert (g("min_residual_partition_order"), g("max_residual_partition_order")) == (0, 0)
        lib.FLAC__stream_encoder_get_total_samples_estimate.restype = C.c_uint64
        assert g("total_samples_estimate") == 0
    finally:
        lib.FLAC__stream_encoder_delete(enc)


@pytest.mark.parametrize("settings,status", [
    (dict(channels=0), 4), (dict(channels=9), 4),
    (dict(bits_per_sample=3), 5), (dict(bits_per_sample=25), 5),
    (dict(sample_rate=0), 6), (dict(sample_rate=655351), 6),
    (dict(blocksize=15), 7), (dict(blocksize=65536), 7),
    (dict(max_lpc_order=33), 8),
    (dict(compression_level=5, blocksize=16, max_lpc_order=32, streamable_subset=0), 10),
    (dict(compression_level=5, qlp_coeff_precision=4), 9),
    (dict(compression_level=5, qlp_coeff_precision=16), 9),
    (dict(compression_level=5, sample_rate=96000, blocksize=16385), 11),
    (dict(compression_level=5, bits_per_sample=17), 11),
    (dict(compression_level=5, blocksize=8192), 11),               # > 4608 at <= 48 kHz
    (dict(compression_level=5, max_lpc_order=13), 11),             # > 12 at <= 48 kHz
    (dict(compression_level=5, max_residual_partition_order=9), 11),
])
def test_init_refusals_in_reference_order(lib, settings, status):
    enc = new_encoder(lib, **settings)
    try:
        st, _cb = init(lib, enc)
        assert st == status
        assert lib.FLAC__stream_encoder_get_state(enc) == 1  # still uninitialised
    finally:
        lib.FLAC__stream_encoder_delete(enc)


def test_init_needs_a_write_callback(lib):
    enc = new_encoder(lib, compression_level=5)
    try:
        st = lib.FLAC__stream_encoder_init_stream(enc, C.cast(None, L.WRITE_CB), C.cast(None, L.SEEK_CB), C.cast(None, L.TELL_CB), None, None)
        assert st == 3
        # a seek callback without a tell callback is refused too
        wcb, scb = L.WRITE_CB(lambda *_a: 0), L.SEEK_CB(lambda *_a: 0)
        assert lib.FLAC__stream_encoder_init_stream(enc, wcb, scb, C.cast(None, L.TELL_CB), None, None) == 3
        assert lib.FLAC__stream_encoder_init_ogg_stream(enc, None, wcb, None, None, None, None) == 2
    finally:
        lib.FLAC__stream_encoder_delete(enc)


def test_apodization_parser(lib):
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    try:
        ok = lambda spec: lib.FLAC__stream_encoder_set_apodization(enc, C.c_char_p(spec))
        for spec in (b"hann", b"tukey(0.25);hamming;blackman", b"gauss(0.2)", b"rectangle;welch;triangle;bartlett;nuttall;flattop;connes",
                     b"blackman_harris_4term_92db;kaiser_bessel;bartlett_hann", b"bogus", b"tukey(1.5)", b"gauss(0.7)", b""):
            assert ok(spec)  # never refused: unknown names are skipped, an empty list falls back to tukey(0.5)
    finally:
        lib.FLAC__stream_encoder_delete(enc)


def test_no_gpu_is_a_loud_init_failure(lib, capfd):
    if has_gpu():
        pytest.skip("a GPU is present")
    enc = new_encoder(lib, compression_level=8)
    try:
        st, _cb = init(lib, enc)
        assert st == 1  # ENCODER_ERROR
        assert lib.FLAC__stream_encoder_get_state(enc) == 8  # MEMORY_ALLOCATION_ERROR: no device buffers
        assert not lib.FLAC__stream_encoder_process_interleaved(enc, (C.c_int32 * 8)(), u(4))
    finally:
        lib.FLAC__stream_encoder_delete(enc)


def test_format_helpers(lib):
    assert lib.FLAC__format_sample_rate_is_valid(u(44100)) and not lib.FLAC__format_sample_rate_is_valid(u(0))
    assert not lib.FLAC__format_sample_rate_is_valid(u(655351))
    assert lib.FLAC__format_sample_rate_is_subset(u(96000)) and lib.FLAC__format_sample_rate_is_subset(u(65537 * 10 - 7 * 10))
    assert not lib.FLAC__format_sample_rate_is_subset(u(65537))
    assert lib.FLAC__format_blocksize_is_subset(u(4608), u(44100)) and not lib.FLAC__format_blocksize_is_subset(u(4609), u(44100))
    assert lib.FLAC__format_blocksize_is_subset(u(16384), u(96000)) and not lib.FLAC__format_blocksize_is_subset(u(16385), u(96000))
    pts = (L.SeekPoint * 6)()
    for i, sn in enumerate((500, L.PLACEHOLDER, 100, 500, 0, L.PLACEHOLDER)):
        pts[i].sample_number, pts[i].stream_offset, pts[i].frame_samples = sn, i, 1
    st = L.SeekTable(6, C.cast(pts, C.POINTER(L.SeekPoint)))
    assert not lib.FLAC__format_seektable_is_legal(C.byref(st))
    assert lib.FLAC__format_seektable_sort(C.byref(st)) == 5  # one duplicate dropped, placeholders last
    assert [pts[i].sample_number for i in range(6)] == [0, 100, 500, L.PLACEHOLDER, L.PLACEHOLDER, L.PLACEHOLDER]
    assert lib.FLAC__format_seektable_is_legal(C.byref(st))
    assert C.c_char_p.in_dll(lib, "FLAC__VERSION_STRING").value == b"1.2.1"
    assert C.c_char_p.in_dll(lib, "FLAC__VENDOR_STRING").value == b"reference libFLAC 1.2.1 20070917"
    assert C.c_uint.in_dll(lib, "FLAC__STREAM_SYNC").value == 0x664C6143
    assert (C.c_char_p * 10).in_dll(lib, "FLAC__StreamDecoderStateString")[9] == b"FLAC__STREAM_DECODER_UNINITIALIZED"


def test_decoder_metadata_phase_needs_no_gpu(lib):
    case = next(c for c in golden_util.load_index() if c["name"] == "s24_l8")
    pcm, meta, frames = golden_util.load_case(case)
    sd = L.StreamDecoder(lib, meta + b"".join(frames), respond_all=True, seekable=True)
    try:
        assert sd.init_status == 0 and sd.state() == 0
        assert sd.call("process_single")      # finds fLaC, reads STREAMINFO
        assert sd.metadata == [(0, 0, 34)]
        assert sd.call("process_until_end_of_metadata")
        assert sd.metadata == [(0, 0, 34), (4, 1, 40)]
        assert sd.state() == 2
        assert sd.call("get_total_samples") == pcm.shape[0]
        pos = C.c_uint64()
        assert sd.call("get_decode_position", C.byref(pos)) and pos.value == len(meta)
        if not has_gpu():
            assert not sd.call("process_single")
            assert sd.state() == 8
    finally:
        sd.close()


def test_decoder_init_refusals(lib):
    dec = C.c_void_p(lib.FLAC__stream_decoder_new())
    try:
        assert lib.FLAC__stream_decoder_get_state(dec) == 9
        assert lib.FLAC__stream_decoder_init_stream(dec, None, None, None, None, None, None, None, None, None) == 2
        def empty(_d, _buf, pbytes, _c):
            pbytes[0] = 0
            return 1  # END_OF_STREAM
        rcb = L.DEC_READ_CB(empty)
        wcb = L.DEC_WRITE_CB(lambda *_a: 0)
        ecb = L.DEC_ERROR_CB(lambda *_a: None)
        scb = L.DEC_SEEK_CB(lambda *_a: 0)
        # a seek callback needs tell, length and eof callbacks
        assert lib.FLAC__stream_decoder_init_stream(dec, rcb, scb, None, None, None, wcb, None, ecb, None) == 2
        assert lib.FLAC__stream_decoder_init_ogg_stream(dec, rcb, None, None, None, None, wcb, None, ecb, None) == 1
        assert lib.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, None, ecb, None) == 0
        assert lib.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, None, ecb, None) == 5
        assert not lib.FLAC__stream_decoder_set_md5_checking(dec, C.c_int(1))
        # an empty input: end of stream while looking for metadata
        assert not lib.FLAC__stream_decoder_process_until_end_of_metadata(dec)
        assert lib.FLAC__stream_decoder_get_state(dec) == 4
        assert lib.FLAC__stream_decoder_finish(dec)
        assert lib.FLAC__stream_decoder_get_state(dec) == 9
    finally:
        lib.FLAC__stream_decoder_delete(dec)


def test_decoder_reads_every_metadata_type(lib):
    """Blocks serialised by hand: APPLICATION filters, SEEKTABLE, CUESHEET, PICTURE, unknown type."""
    def block(t, body, last=False):
        return bytes([(0x80 if last else 0) | t]) + len(body).to_bytes(3, "big") + body
    streaminfo = bytes.fromhex("10001000000000000000" "0ac442f0" "00000000" + "00" * 16)
    app1 = b"aaaa" + b"\x01\x02\x03"
    app2 = b"bbbb"
    seek = (0).to_bytes(8, "big") + (0).to_bytes(8, "big") + (4096).to_bytes(2, "big") + b"\xff" * 8 + bytes(10)
    vc = (3).to_bytes(4, "little") + b"abc" + (1).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"A=b"
    track = (0).to_bytes(8, "big") + bytes([1]) + b"ISRC12345678" + bytes([0x80]) + bytes(13) + bytes([1]) + (0).to_bytes(8, "big") + bytes([1]) + bytes(3)
    leadout = (44100).to_bytes(8, "big") + bytes([170]) + bytes(12) + bytes([0]) + bytes(13) + bytes([0])
    cue = bytes(128) + (88200).to_bytes(8, "big") + bytes([0x80]) + bytes(258) + bytes([2]) + track + leadout
    pic = (3).to_bytes(4, "big") + (9).to_bytes(4, "big") + b"image/png" + (2).to_bytes(4, "big") + b"hi" + \
        (1).to_bytes(4, "big") * 4 + (5).to_bytes(4, "big") + b"\x89PNG\r"
    stream = b"fLaC" + block(0, streaminfo) + block(2, app1) + block(2, app2) + block(3, seek) + block(4, vc) + block(5, cue) + \
        block(6, pic) + block(1, bytes(20)) + block(100, b"xyz", last=True)
    sd = L.StreamDecoder(lib, stream, respond_all=True, max_read=13)
    try:
        assert sd.call("process_until_end_of_metadata")
        assert sd.metadata == [(0, 0, 34), (2, 0, 7), (2, 0, 4), (3, 0, 36), (4, 0, len(vc)), (5, 0, len(cue)), (6, 0, len(pic)), (1, 0, 20), (100, 1, 3)]
        assert sd.state() == 2 and sd.errors == []
    finally:
        sd.close()
    # respond to one application id only
    d2 = C.c_void_p(lib.FLAC__stream_decoder_new())
    seen = []
    try:
        assert lib.FLAC__stream_decoder_set_metadata_ignore_all(d2)
        assert lib.FLAC__stream_decoder_set_metadata_respond_application(d2, (C.c_ubyte * 4)(*b"bbbb"))
        pos = [0]

        def rd(_d, buf, pbytes, _c):
            n = min(pbytes[0], len(stream) - pos[0])
            if n == 0:
                pbytes[0] = 0
                return 1
            C.memmove(buf, stream[pos[0]:pos[0] + n], n)
            pos[0] += n
            pbytes[0] = n
            return 0
        rcb, wcb, ecb = L.DEC_READ_CB(rd), L.DEC_WRITE_CB(lambda *_a: 0), L.DEC_ERROR_CB(lambda *_a: None)
        mcb = L.DEC_METADATA_CB(lambda _d, blk, _c: seen.append(tuple(C.cast(blk, C.POINTER(C.c_uint))[0:3]) +
                                                               (bytes(C.cast(blk, C.POINTER(L.StreamMetadata)).contents.data.application.id),)))
        assert lib.FLAC__stream_decoder_init_stream(d2, rcb, None, None, None, None, wcb, mcb, ecb, None) == 0
        assert lib.FLAC__stream_decoder_process_until_end_of_metadata(d2)
        assert seen == [(2, 0, 4, b"bbbb")]
    finally:
        lib.FLAC__stream_decoder_delete(d2)
