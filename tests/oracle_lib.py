"""ctypes binding of oracle/liboracle.so (the CPU restatement used as the checker in tests)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")
MAX_APOD = 32
MAX_ORDER = 32
MAX_CH = 8


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side_stereo",
        "loose_mid_side_stereo", "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search",
        "do_exhaustive_model_search", "min_residual_partition_order", "max_residual_partition_order",
        "disable_constant_subframes", "disable_fixed_subframes", "disable_verbatim_subframes",
        "num_apodizations")] + [("apod_type", C.c_uint32 * MAX_APOD), ("apod_param", C.c_float * MAX_APOD)]


class SubframeInfo(C.Structure):
    _fields_ = [("type", C.c_uint32), ("wasted_bits", C.c_uint32), ("subframe_bps", C.c_uint32),
                ("order", C.c_uint32), ("estimate_bits", C.c_uint32), ("qlp_precision", C.c_uint32),
                ("qlp_shift", C.c_int32), ("qlp_coeff", C.c_int32 * MAX_ORDER),
                ("partition_order", C.c_uint32), ("is_rice2", C.c_uint32), ("rice_param0", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [("channel_assignment", C.c_uint32), ("frame_bytes", C.c_uint32),
                ("sub", SubframeInfo * MAX_CH)]


class StreamInfo(C.Structure):
    _fields_ = [("min_blocksize", C.c_uint32), ("max_blocksize", C.c_uint32),
                ("min_framesize", C.c_uint32), ("max_framesize", C.c_uint32),
                ("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("total_samples", C.c_uint64), ("md5", C.c_uint8 * 16), ("audio_offset", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "flac_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        _lib = C.CDLL(so)
        _lib.fo_encode_frames.restype = C.c_size_t
        _lib.fo_write_stream_header.restype = C.c_size_t
        _lib.fo_decode_frames.restype = C.c_int64
        _lib.fo_max_frame_bytes.restype = C.c_size_t
        _lib.fo_utf8_encode.restype = C.c_size_t
        _lib.fo_bitpack.restype = C.c_size_t
        _lib.fo_crc8.restype = C.c_uint8
        _lib.fo_crc16.restype = C.c_uint16
    return _lib


def preset(level, channels, bps, sample_rate, blocksize=0, **over):
    p = Params()
    rc = lib().fo_params_preset(C.byref(p), level, channels, bps, sample_rate, blocksize)
    if rc:
        raise ValueError("fo_params_preset rc=%d" % rc)
    if over:
        for k, v in over.items():
            if k == "apodizations":
                p.num_apodizations = len(v)
                for i, (t, prm) in enumerate(v):
                    p.apod_type[i] = t
                    p.apod_param[i] = prm
            else:
                setattr(p, k, v)
        rc = lib().fo_params_resolve(C.byref(p))
        if rc:
            raise ValueError("fo_params_resolve rc=%d" % rc)
    return p


def encode_frames(p, pcm, first_frame=0, want_info=False):
    """pcm: int32 (n, channels).  Returns (list of frame bytes, infos or None)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n = pcm.shape[0]
    nframes_max = (n + p.blocksize - 1) // p.blocksize + 1
    cap = lib().fo_max_frame_bytes(C.byref(p)) * nframes_max
    out = np.empty(cap, dtype=np.uint8)
    fbytes = np.zeros(nframes_max, dtype=np.uint32)
    nf = C.c_uint32(0)
    infos = (FrameInfo * nframes_max)() if want_info else None
    total = lib().fo_encode_frames(
        C.byref(p), pcm.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(first_frame),
        out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), fbytes.ctypes.data_as(C.c_void_p),
        C.c_uint32(nframes_max), C.byref(nf), infos)
    if total == C.c_size_t(-1).value:
        raise RuntimeError("oracle encode failed")
    frames = []
    pos = 0
    for i in range(nf.value):
        frames.append(out[pos:pos + fbytes[i]].tobytes())
        pos += int(fbytes[i])
    return frames, (list(infos)[:nf.value] if want_info else None)


def stream_header(p, total_samples, min_fs, max_fs, md5):
    buf = (C.c_uint8 * 256)()
    md = (C.c_uint8 * 16)(*md5)
    n = lib().fo_write_stream_header(C.byref(p), C.c_uint64(total_samples), min_fs, max_fs, md, buf, C.c_size_t(256))
    return bytes(buf[:n])


def md5_pcm(pcm, bps):
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    d = (C.c_uint8 * 16)()
    lib().fo_md5_pcm(pcm.ctypes.data_as(C.c_void_p), C.c_uint64(pcm.shape[0]), C.c_uint32(pcm.shape[1]), C.c_uint32(bps), d)
    return bytes(d)


def encode_stream(p, pcm):
    """Whole native FLAC stream exactly as the reference writes it through a seekable init_stream."""
    frames, _ = encode_frames(p, pcm)
    sizes = [len(f) for f in frames]
    hdr = stream_header(p, pcm.shape[0], min(sizes) if sizes else 0, max(sizes) if sizes else 0,
                        md5_pcm(pcm, p.bits_per_sample))
    return hdr + b"".join(frames)


def parse_header(data):
    si = StreamInfo()
    buf = np.frombuffer(data, dtype=np.uint8)
    rc = lib().fo_parse_stream_header(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(si))
    if rc:
        raise ValueError("not a FLAC stream")
    return si


def decode_frames(data, channels, bps, sample_rate, max_wide):
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty((max_wide, channels), dtype=np.int32)
    nf = C.c_uint32(0)
    rc = lib().fo_decode_frames(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), channels, bps, sample_rate,
                                out.ctypes.data_as(C.c_void_p), C.c_uint64(max_wide), C.byref(nf))
    if rc < 0:
        return rc, None, nf.value
    return rc, out[:rc], nf.value
