"""ctypes driver for the libFLAC 1.2.1 stream encoder / decoder C API.

Works against any shared object that exports that ABI: the reference library built into
oracle/_ref/ (to generate golden vectors, in the build container only) and this repository's own
drop-in library (to test it the way the reference's own API tests do).
"""
import ctypes as C
import os

import numpy as np

WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t, C.c_uint, C.c_uint, C.c_void_p)
SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)

DEC_READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.POINTER(C.c_size_t), C.c_void_p)
DEC_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
DEC_ERROR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


class FrameHeader(C.Structure):
    """Leading fields of FLAC__Frame (FLAC__FrameHeader, include/FLAC/format.h:446)."""
    _fields_ = [("blocksize", C.c_uint), ("sample_rate", C.c_uint), ("channels", C.c_uint),
                ("channel_assignment", C.c_int), ("bits_per_sample", C.c_uint),
                ("number_type", C.c_int), ("number", C.c_uint64), ("crc", C.c_uint8)]


def load(path):
    if os.path.basename(path).startswith("libflac_amd"):
        # PyTorch ships its own HIP runtime: in a process that uses both, torch's has to be loaded
        # first, or whichever initialises second finds no device (flac_amd.lib() does the same)
        import flac_amd
        flac_amd.lib()
    lib = C.CDLL(path)
    for kind in ("encoder",) + (("decoder",) if hasattr(lib, "FLAC__stream_decoder_new") else ()):
        getattr(lib, "FLAC__stream_%s_new" % kind).restype = C.c_void_p
        getattr(lib, "FLAC__stream_%s_delete" % kind).argtypes = [C.c_void_p]
        getattr(lib, "FLAC__stream_%s_delete" % kind).restype = None
    return lib


# ---- FLAC__StreamMetadata (include/FLAC/format.h:478-846), as far as the tests build blocks ----
class SeekPoint(C.Structure):
    _fields_ = [("sample_number", C.c_uint64), ("stream_offset", C.c_uint64), ("frame_samples", C.c_uint)]


class SeekTable(C.Structure):
    _fields_ = [("num_points", C.c_uint), ("points", C.POINTER(SeekPoint))]


class VCEntry(C.Structure):
    _fields_ = [("length", C.c_uint32), ("entry", C.POINTER(C.c_ubyte))]


class VorbisComment(C.Structure):
    _fields_ = [("vendor_string", VCEntry), ("num_comments", C.c_uint32), ("comments", C.POINTER(VCEntry))]


class Application(C.Structure):
    _fields_ = [("id", C.c_ubyte * 4), ("data", C.POINTER(C.c_ubyte))]


class StreamInfo(C.Structure):
    _fields_ = [("min_blocksize", C.c_uint), ("max_blocksize", C.c_uint), ("min_framesize", C.c_uint),
                ("max_framesize", C.c_uint), ("sample_rate", C.c_uint), ("channels", C.c_uint),
                ("bits_per_sample", C.c_uint), ("total_samples", C.c_uint64), ("md5sum", C.c_ubyte * 16)]


class MetadataBody(C.Union):
    _fields_ = [("stream_info", StreamInfo), ("application", Application), ("seek_table", SeekTable),
                ("vorbis_comment", VorbisComment), ("raw", C.c_ubyte * 192)]


class StreamMetadata(C.Structure):
    _fields_ = [("type", C.c_int), ("is_last", C.c_int), ("length", C.c_uint), ("data", MetadataBody)]


PLACEHOLDER = 0xFFFFFFFFFFFFFFFF


def _bytes_ptr(b, keep):
    buf = (C.c_ubyte * max(1, len(b)))(*b)
    keep.append(buf)
    return C.cast(buf, C.POINTER(C.c_ubyte))


def build_metadata(specs):
    """specs: list of ("padding", n) | ("seektable", [sample numbers]) | ("application", id4, data) |
    ("vorbis_comment", [b"K=V", ...]).  Returns (FLAC__StreamMetadata*[] array, count, keepalive)."""
    keep, blocks = [], []
    for spec in specs:
        m = StreamMetadata()
        kind = spec[0]
        if kind == "padding":
            m.type, m.length = 1, spec[1]
        elif kind == "seektable":
            pts = (SeekPoint * max(1, len(spec[1])))()
            for i, sn in enumerate(spec[1]):
                pts[i].sample_number, pts[i].stream_offset, pts[i].frame_samples = sn, 0, 0
            keep.append(pts)
            m.type, m.length = 3, 18 * len(spec[1])
            m.data.seek_table.num_points = len(spec[1])
            m.data.seek_table.points = C.cast(pts, C.POINTER(SeekPoint))
        elif kind == "application":
            m.type, m.length = 2, 4 + len(spec[2])
            for i in range(4):
                m.data.application.id[i] = spec[1][i]
            m.data.application.data = _bytes_ptr(spec[2], keep)
        elif kind == "vorbis_comment":
            ents = (VCEntry * max(1, len(spec[1])))()
            total = 4 + 4
            for i, kv in enumerate(spec[1]):
                ents[i].length = len(kv)
                ents[i].entry = _bytes_ptr(kv, keep)
                total += 4 + len(kv)
            keep.append(ents)
            m.type, m.length = 4, total
            m.data.vorbis_comment.vendor_string.length = 0
            m.data.vorbis_comment.vendor_string.entry = None
            m.data.vorbis_comment.num_comments = len(spec[1])
            m.data.vorbis_comment.comments = C.cast(ents, C.POINTER(VCEntry))
        else:
            raise ValueError(kind)
        blocks.append(m)
    arr = (C.POINTER(StreamMetadata) * max(1, len(blocks)))(*[C.pointer(b) for b in blocks])
    keep.extend(blocks)
    return arr, len(blocks), keep


def encode(lib, pcm, bits_per_sample, sample_rate, level=None, blocksize=0, settings=None,
           chunk=None, planar=False, seekable=False, total_samples_estimate=0, metadata=None):
    """Encode interleaved int32 pcm (n, channels) through the stream API.

    Returns (metadata_bytes, [frame_bytes, ...], final_state_ok, writes) where writes is the raw
    list of (bytes, samples, current_frame) callbacks.  With seekable=True seek/tell callbacks are
    supplied and the returned metadata reflects the finish()-time STREAMINFO rewrite.
    """
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, channels = pcm.shape
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    u = C.c_uint
    ok = True
    ok &= bool(lib.FLAC__stream_encoder_set_streamable_subset(enc, C.c_int(0)))
    ok &= bool(lib.FLAC__stream_encoder_set_channels(enc, u(channels)))
    ok &= bool(lib.FLAC__stream_encoder_set_bits_per_sample(enc, u(bits_per_sample)))
    ok &= bool(lib.FLAC__stream_encoder_set_sample_rate(enc, u(sample_rate)))
    if level is not None:
        ok &= bool(lib.FLAC__stream_encoder_set_compression_level(enc, u(level)))
    if blocksize:
        ok &= bool(lib.FLAC__stream_encoder_set_blocksize(enc, u(blocksize)))
    if total_samples_estimate:
        lib.FLAC__stream_encoder_set_total_samples_estimate(enc, C.c_uint64(total_samples_estimate))
    for key, val in (settings or {}).items():
        fn = getattr(lib, "FLAC__stream_encoder_" + key)
        if key == "set_apodization":
            ok &= bool(fn(enc, C.c_char_p(val.encode())))
        elif key.startswith("disable_"):
            ok &= bool(fn(enc, C.c_int(1 if val else 0)))
        else:
            ok &= bool(fn(enc, u(int(val))))
    keepalive = None
    if metadata:
        arr, count, keepalive = build_metadata(metadata)
        ok &= bool(lib.FLAC__stream_encoder_set_metadata(enc, arr, u(count)))
    assert ok, "a setter failed"

    stream = bytearray()
    state = {"pos": 0}
    writes = []

    def on_write(_e, buf, nbytes, samples, frame, _c):
        data = bytes(buf[:nbytes])
        writes.append((data, samples, frame))
        end = state["pos"] + nbytes
        if end > len(stream):
            stream.extend(b"\0" * (end - len(stream)))
        stream[state["pos"]:end] = data
        state["pos"] = end
        return 0

    def on_seek(_e, off, _c):
        state["pos"] = off
        return 0

    def on_tell(_e, poff, _c):
        poff[0] = state["pos"]
        return 0

    wcb = WRITE_CB(on_write)
    scb = SEEK_CB(on_seek) if seekable else C.cast(None, SEEK_CB)
    tcb = TELL_CB(on_tell) if seekable else C.cast(None, TELL_CB)
    st = lib.FLAC__stream_encoder_init_stream(enc, wcb, scb, tcb, None, None)
    if st != 0:
        lib.FLAC__stream_encoder_delete(enc)
        raise RuntimeError("init_stream failed with status %d" % st)
    if chunk is None:
        chunk = n if n else 1
    pos = 0
    good = True
    while pos < n and good:
        m = min(chunk, n - pos)
        if planar:
            cols = [np.ascontiguousarray(pcm[pos:pos + m, c]) for c in range(channels)]
            arr = (C.POINTER(C.c_int32) * channels)(*[c.ctypes.data_as(C.POINTER(C.c_int32)) for c in cols])
            good = bool(lib.FLAC__stream_encoder_process(enc, arr, u(m)))
        else:
            blk = pcm[pos:pos + m]
            good = bool(lib.FLAC__stream_encoder_process_interleaved(
                enc, blk.ctypes.data_as(C.POINTER(C.c_int32)), u(m)))
        pos += m
    good = bool(lib.FLAC__stream_encoder_finish(enc)) and good
    lib.FLAC__stream_encoder_delete(enc)
    frames = [d for d, s, _f in writes if s > 0]
    meta_len = 0
    for d, s, _f in writes:
        if s > 0:
            break
        meta_len += len(d)
    return bytes(stream[:meta_len]), frames, good, writes


DEC_SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
DEC_TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
DEC_LENGTH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
DEC_EOF_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
DEC_METADATA_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


class MetadataHead(C.Structure):
    """Leading fields of FLAC__StreamMetadata (include/FLAC/format.h:811)."""
    _fields_ = [("type", C.c_int), ("is_last", C.c_int), ("length", C.c_uint)]


class StreamDecoder:
    """A stream decoder over an in-memory FLAC stream, with every callback recorded.

    blocks: [(first_sample, int32 array (blocksize, channels))]; errors: [status];
    metadata: [(type, is_last, length)].  max_read caps what one read callback returns.
    """

    def __init__(self, lib, stream_bytes, md5_check=True, seekable=False, max_read=None, respond_all=False):
        self.lib = lib
        self.data = bytes(stream_bytes)
        self.pos = 0
        self.blocks, self.errors, self.metadata = [], [], []
        self.max_read = max_read
        self.abort_writes = False
        lib.FLAC__stream_decoder_get_total_samples.restype = C.c_uint64
        lib.FLAC__stream_decoder_get_resolved_state_string.restype = C.c_char_p
        self.dec = C.c_void_p(lib.FLAC__stream_decoder_new())
        if md5_check:
            assert lib.FLAC__stream_decoder_set_md5_checking(self.dec, C.c_int(1))
        if respond_all:
            assert lib.FLAC__stream_decoder_set_metadata_respond_all(self.dec)
        self._cbs = (DEC_READ_CB(self._read), DEC_SEEK_CB(self._seek), DEC_TELL_CB(self._tell), DEC_LENGTH_CB(self._length),
                     DEC_EOF_CB(self._eof), DEC_WRITE_CB(self._write), DEC_METADATA_CB(self._meta), DEC_ERROR_CB(self._error))
        r, s_, t, l_, e, w, m, err = self._cbs
        if seekable:
            self.init_status = lib.FLAC__stream_decoder_init_stream(self.dec, r, s_, t, l_, e, w, m, err, None)
        else:
            self.init_status = lib.FLAC__stream_decoder_init_stream(self.dec, r, None, None, None, None, w, m, err, None)

    def _read(self, _d, buf, pbytes, _c):
        want = pbytes[0]
        if self.max_read:
            want = min(want, self.max_read)
        take = min(want, len(self.data) - self.pos)
        if take == 0:
            pbytes[0] = 0
            return 1  # END_OF_STREAM
        C.memmove(buf, self.data[self.pos:self.pos + take], take)
        self.pos += take
        pbytes[0] = take
        return 0

    def _seek(self, _d, off, _c):
        if off > len(self.data):
            return 1
        self.pos = off
        return 0

    def _tell(self, _d, poff, _c):
        poff[0] = self.pos
        return 0

    def _length(self, _d, plen, _c):
        plen[0] = len(self.data)
        return 0

    def _eof(self, _d, _c):
        return 1 if self.pos >= len(self.data) else 0

    def _write(self, _d, frame, buffers, _c):
        hdr = C.cast(frame, C.POINTER(FrameHeader)).contents
        bs, ch = hdr.blocksize, hdr.channels
        blk = np.empty((bs, ch), dtype=np.int32)
        for c in range(ch):
            blk[:, c] = np.ctypeslib.as_array(buffers[c], shape=(bs,))
        self.blocks.append((int(hdr.number), blk))
        self.last_header = (hdr.blocksize, hdr.sample_rate, hdr.channels, hdr.channel_assignment, hdr.bits_per_sample,
                            hdr.number_type, int(hdr.number), hdr.crc)
        return 1 if self.abort_writes else 0

    def _meta(self, _d, block, _c):
        h = C.cast(block, C.POINTER(MetadataHead)).contents
        self.metadata.append((h.type, h.is_last, h.length))

    def _error(self, _d, status, _c):
        self.errors.append(status)

    def call(self, name, *args):
        return getattr(self.lib, "FLAC__stream_decoder_" + name)(self.dec, *args)

    def state(self):
        return self.call("get_state")

    def pcm(self):
        if not self.blocks:
            return np.zeros((0, 1), dtype=np.int32)
        return np.concatenate([b for _n, b in self.blocks], axis=0)

    def close(self):
        if self.dec:
            self.lib.FLAC__stream_decoder_delete(self.dec)
            self.dec = None


def decode(lib, stream_bytes, md5_check=True, max_read=None):
    """Decode a whole FLAC stream through the stream decoder API.

    Returns (pcm int32 (n, channels), list of error codes, finish_ok).
    """
    sd = StreamDecoder(lib, stream_bytes, md5_check=md5_check, max_read=max_read)
    if sd.init_status != 0:
        sd.close()
        raise RuntimeError("decoder init_stream failed with status %d" % sd.init_status)
    ok = bool(sd.call("process_until_end_of_stream"))
    fin = bool(sd.call("finish"))
    sd.close()
    return sd.pcm(), sd.errors, ok and fin
