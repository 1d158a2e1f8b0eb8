"""Python binding (ctypes) of libflac_amd.so, the MI355X FLAC frame engine.

The library is HIP code for gfx950; there is no CPU implementation behind it.  Importing this
package without the built library, or using it on a machine without an AMD GPU, raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLAC_AMD_LIB: load another build of the library (kernel experiments)
LIB_PATH = os.environ.get("FLAC_AMD_LIB") or os.path.join(_HERE, "lib", "libflac_amd.so")

MAX_APOD = 32
ENC_STAGES = 8
DEC_STAGES = 4

APOD = {"bartlett": 0, "bartlett_hann": 1, "blackman": 2, "blackman_harris_4term_92db": 3, "connes": 4,
        "flattop": 5, "gauss": 6, "hamming": 7, "hann": 8, "kaiser_bessel": 9, "nuttall": 10,
        "rectangle": 11, "triangle": 12, "tukey": 13, "welch": 14}


class EncodeParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side_stereo",
        "loose_mid_side_stereo", "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search",
        "do_exhaustive_model_search", "min_residual_partition_order", "max_residual_partition_order",
        "disable_constant_subframes", "disable_fixed_subframes", "disable_verbatim_subframes",
        "num_apodizations")] + [("apod_type", C.c_uint32 * MAX_APOD), ("apod_param", C.c_float * MAX_APOD)]


class StreamDesc(C.Structure):
    _fields_ = [("byte_offset", C.c_uint64), ("byte_length", C.c_uint64),
                ("sample_offset", C.c_uint64), ("total_samples", C.c_uint64), ("first_sample", C.c_uint64)]


class FlacAmdError(RuntimeError):
    """A refused call; `code` is the library's negative FLACAMD_E_* value."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


_lib = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlacAmdError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or make -C flac_amd/csrc)" % LIB_PATH)
        # PyTorch ships its own HIP runtime; when both are to live in one process torch's has to be
        # the one that gets loaded, or whichever initialises second finds no device.
        if "torch" not in sys.modules and not os.environ.get("FLAC_AMD_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(LIB_PATH)
        _lib.flacamd_last_error.restype = C.c_char_p
        _lib.flacamd_version.restype = C.c_char_p
        _lib.flacamd_encoder_max_output_bytes.restype = C.c_size_t
        _lib.flacamd_encoder_stage_name.restype = C.c_char_p
        if hasattr(_lib, "flacamd_decoder_stage_name"):
            _lib.flacamd_decoder_stage_name.restype = C.c_char_p
        _lib.flacamd_debug_sizeof.restype = C.c_size_t
    return _lib


def _check(rc):
    if rc != 0:
        raise FlacAmdError("flac_amd error %d: %s" % (rc, lib().flacamd_last_error().decode()), rc)


def preset(level, channels, bits_per_sample, sample_rate, blocksize=0, **over):
    """Settings of `flac -<level>`, optionally overridden field by field, then resolved the way
    the reference's init does (stream_encoder.c:660-759)."""
    p = EncodeParams()
    _check(lib().flacamd_params_preset(C.byref(p), level, channels, bits_per_sample, sample_rate, blocksize))
    if over:
        for k, v in over.items():
            if k == "apodizations":
                p.num_apodizations = len(v)
                for i, (t, prm) in enumerate(v):
                    p.apod_type[i] = APOD[t] if isinstance(t, str) else t
                    p.apod_param[i] = prm
            else:
                setattr(p, k, v)
        _check(lib().flacamd_params_resolve(C.byref(p)))
    return p


def window(apod, param, length):
    w = np.zeros(length, dtype=np.float32)
    _check(lib().flacamd_window(APOD[apod] if isinstance(apod, str) else apod, C.c_float(param), length,
                                w.ctypes.data_as(C.c_void_p)))
    return w


class Encoder:
    """Batch frame encoder bound to one GPU."""

    def __init__(self, params, device=0, max_frames=1024):
        self.params = params
        self.max_frames = max_frames
        self._h = C.c_void_p()
        _check(lib().flacamd_encoder_create(C.byref(params), device, max_frames, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().flacamd_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_output_bytes(self, nframes):
        return lib().flacamd_encoder_max_output_bytes(self._h, nframes)

    def encode_host(self, pcm, first_frame=0, is_last=True, loose_state=0):
        """pcm: int32 (n, channels) on the host.  Returns (list of frame bytes, loose_state)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        n = pcm.shape[0]
        bs = self.params.blocksize
        frames = []
        pos = 0
        per_batch = self.max_frames * bs
        ls = C.c_uint32(loose_state)
        while pos < n:
            m = min(per_batch, n - pos)
            last = is_last and pos + m == n
            if not last:
                m -= m % bs
                if m == 0:
                    raise FlacAmdError("a non-final call needs whole blocks")
            nfr = (m + bs - 1) // bs
            cap = self.max_output_bytes(nfr)
            out = np.empty(cap, dtype=np.uint8)
            fbytes = np.zeros(nfr, dtype=np.uint32)
            nf = C.c_uint32(0)
            total = C.c_size_t(0)
            blk = pcm[pos:pos + m]
            _check(lib().flacamd_encode_frames_host(
                self._h, blk.ctypes.data_as(C.c_void_p), C.c_uint64(m), C.c_uint32(first_frame + len(frames)),
                C.c_int(1 if last else 0), C.byref(ls), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap),
                fbytes.ctypes.data_as(C.c_void_p), C.byref(nf), C.byref(total)))
            o = 0
            for i in range(nf.value):
                frames.append(out[o:o + int(fbytes[i])].tobytes())
                o += int(fbytes[i])
            pos += m
        return frames, ls.value

    def encode_device(self, pcm_ptr, sample_bytes, wide_samples, out_ptr, out_capacity, frame_bytes_ptr=None,
                      offsets_ptr=None, first_frame=0, is_last=True, stream=None):
        """Enqueue one batch whose PCM already lies in device memory; returns the frame count."""
        nf = C.c_uint32(0)
        _check(lib().flacamd_encode_frames(
            self._h, C.c_void_p(pcm_ptr), C.c_uint32(sample_bytes), C.c_uint64(wide_samples), C.c_uint32(first_frame),
            C.c_int(1 if is_last else 0), None, C.c_void_p(out_ptr), C.c_size_t(out_capacity),
            C.c_void_p(frame_bytes_ptr or 0), C.c_void_p(offsets_ptr or 0), C.byref(nf), C.c_void_p(stream or 0)))
        return nf.value

    def status(self):
        _check(lib().flacamd_encoder_status(self._h))

    def set_profiling(self, on):
        _check(lib().flacamd_encoder_set_profiling(self._h, 1 if on else 0))

    def stage_ms(self):
        ms = (C.c_float * ENC_STAGES)()
        _check(lib().flacamd_encoder_stage_ms(self._h, ms))
        return {lib().flacamd_encoder_stage_name(i).decode(): float(ms[i]) for i in range(ENC_STAGES)}

    def debug_read(self, what, nbytes):
        buf = np.zeros(nbytes, dtype=np.uint8)
        _check(lib().flacamd_debug_read(self._h, what, buf.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes)))
        return buf


class FrameInfo(C.Structure):
    """flacamd_frame_info"""
    _fields_ = [("pos", C.c_uint64), ("sample", C.c_uint64), ("end_byte", C.c_uint64), ("hdr_bytes", C.c_uint32), ("blocksize", C.c_uint32),
                ("assignment", C.c_uint32), ("stream", C.c_uint32), ("status", C.c_uint32), ("reserved", C.c_uint32)]


class SubframeInfo(C.Structure):
    """flacamd_subframe_info"""
    _fields_ = [("type", C.c_uint8), ("order", C.c_uint8), ("wasted_bits", C.c_uint8), ("precision", C.c_uint8), ("shift", C.c_int8),
                ("partition_order", C.c_uint8), ("rice2", C.c_uint8), ("pad", C.c_uint8), ("constant", C.c_int32), ("pad2", C.c_uint32 * 2),
                ("qlp_coeff", C.c_int32 * 32), ("warmup", C.c_int32 * 32), ("params", C.c_uint8 * 256)]


class Decoder:
    """Batch frame decoder bound to one GPU; every stream shares channels / bits / max blocksize."""

    def __init__(self, channels, bits_per_sample, max_blocksize, device=0, max_input_bytes=1 << 24, max_frames=4096):
        self.channels = channels
        self.bits_per_sample = bits_per_sample
        self._h = C.c_void_p()
        _check(lib().flacamd_decoder_create(channels, bits_per_sample, max_blocksize, device,
                                            C.c_uint64(max_input_bytes), max_frames, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().flacamd_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_planar_output(self, on):
        """flacamd_decoder_set_planar_output: a frame's stretch of the output holds its channels one after the other."""
        _check(lib().flacamd_decoder_set_planar_output(self._h, 1 if on else 0))

    def enable_subframe_info(self, on=True):
        _check(lib().flacamd_decoder_enable_subframe_info(self._h, 1 if on else 0))

    def subframe_info(self, first_frame, nframes):
        """flacamd_decoder_subframe_info: `channels` SubframeInfo records per frame of the table of frames()."""
        rows = (SubframeInfo * (nframes * self.channels))()
        _check(lib().flacamd_decoder_subframe_info(self._h, first_frame, nframes, rows))
        return [rows[i] for i in range(nframes * self.channels)]

    @staticmethod
    def stream_array(streams):
        """The descriptors of a batch as the C array flacamd_decode_streams takes: (byte_offset, byte_length, sample_offset,
        total_samples[, first_sample]) each.  Building it costs a few microseconds per stream; a caller that decodes the same
        batch layout again and again builds it once and passes it to decode_device in place of the list."""
        return (StreamDesc * len(streams))(*[StreamDesc(*(tuple(s) + (0,) * (5 - len(s)))) for s in streams])

    def decode_device(self, flac_ptr, flac_bytes, streams, out_ptr, out_sample_bytes, out_capacity_wide, stream=None):
        arr = streams if isinstance(streams, C.Array) else self.stream_array(streams)
        _check(lib().flacamd_decode_streams(self._h, C.c_void_p(flac_ptr), C.c_uint64(flac_bytes), arr, len(streams),
                                            C.c_void_p(out_ptr), out_sample_bytes, C.c_uint64(out_capacity_wide),
                                            C.c_void_p(stream or 0)))

    def decode_host(self, flac_bytes, streams, out_capacity_wide, out_sample_bytes=4):
        """flac_bytes: the frame data of one or more streams; streams: (byte_offset, byte_length,
        sample_offset, total_samples[, first_sample]).  Returns (pcm (n, channels), frames, wide, bad)."""
        buf = np.frombuffer(flac_bytes, dtype=np.uint8)
        arr = (StreamDesc * len(streams))(*[StreamDesc(*(tuple(s) + (0,) * (5 - len(s)))) for s in streams])
        out = np.zeros((out_capacity_wide, self.channels), dtype=np.int16 if out_sample_bytes == 2 else np.int32)
        frames, wide, bad = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
        _check(lib().flacamd_decode_streams_host(
            self._h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(len(flac_bytes)), arr, len(streams),
            out.ctypes.data_as(C.c_void_p), out_sample_bytes, C.c_uint64(out_capacity_wide),
            C.byref(frames), C.byref(wide), C.byref(bad)))
        return out, frames.value, wide.value, bad.value

    def result(self):
        frames, wide, bad = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
        _check(lib().flacamd_decoder_result(self._h, C.byref(frames), C.byref(wide), C.byref(bad)))
        return frames.value, wide.value, bad.value

    def frames(self, capacity=1 << 16):
        """flacamd_decoder_frames: one FrameInfo per frame header the last decode found, in byte order."""
        rows = (FrameInfo * capacity)()
        count, truncated = C.c_uint32(0), C.c_int(0)
        _check(lib().flacamd_decoder_frames(self._h, rows, capacity, C.byref(count), C.byref(truncated)))
        if truncated.value:
            raise FlacAmdError("the frame table holds more than the %d rows asked for (or more frames than the decoder's capacity)" % capacity)
        return [rows[i] for i in range(min(count.value, capacity))]

    def set_profiling(self, on):
        _check(lib().flacamd_decoder_set_profiling(self._h, 1 if on else 0))

    def stage_ms(self):
        ms = (C.c_float * DEC_STAGES)()
        _check(lib().flacamd_decoder_stage_ms(self._h, ms))
        return {lib().flacamd_decoder_stage_name(i).decode(): float(ms[i]) for i in range(DEC_STAGES)}


# ---------------------------------------------------------------------------------------------
# PCM containers (include/flac_amd_pcm.h)
# ---------------------------------------------------------------------------------------------
PCM_RAW, PCM_WAVE, PCM_RF64, PCM_WAVE64, PCM_AIFF, PCM_AIFF_C = range(6)


class PcmFormat(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "container", "channels", "sample_rate", "bits_per_sample", "shift", "is_big_endian", "is_unsigned",
        "channel_mask", "bytes_per_wide_sample", "reserved")] + [
        ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64), ("total_wide_samples", C.c_uint64)]


def pcm_parse_header(data, channel_map_none=False):
    """flacamd_pcm_parse_header on the first bytes of a WAVE / RF64 / Wave64 / AIFF / AIFF-C file."""
    fmt = PcmFormat()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    _check(lib().flacamd_pcm_parse_header(buf, C.c_size_t(len(data)), int(bool(channel_map_none)), C.byref(fmt)))
    return fmt


def pcm_write_header(container, channels, bits_per_sample, sample_rate, channel_mask, wide_samples):
    """flacamd_pcm_write_header: the header bytes of a file that holds wide_samples samples."""
    n = C.c_size_t(0)
    _check(lib().flacamd_pcm_write_header(container, channels, bits_per_sample, sample_rate, channel_mask,
                                          C.c_uint64(wide_samples), None, C.c_size_t(0), C.byref(n)))
    out = (C.c_uint8 * n.value)()
    _check(lib().flacamd_pcm_write_header(container, channels, bits_per_sample, sample_rate, channel_mask,
                                          C.c_uint64(wide_samples), out, C.c_size_t(n.value), C.byref(n)))
    return bytes(out)


def pcm_unpack(src_ptr, fmt, wide_samples, dst_ptr, dst_sample_bytes, flag_ptr, stream=0):
    """flacamd_pcm_unpack on device pointers (integers)."""
    _check(lib().flacamd_pcm_unpack(C.c_void_p(src_ptr), C.byref(fmt), C.c_uint64(wide_samples), C.c_void_p(dst_ptr),
                                    C.c_uint32(dst_sample_bytes), C.c_void_p(flag_ptr), C.c_void_p(stream)))


def pcm_pack(src_ptr, src_sample_bytes, channels, bits_per_sample, container, wide_samples, dst_ptr, stream=0):
    """flacamd_pcm_pack on device pointers (integers)."""
    _check(lib().flacamd_pcm_pack(C.c_void_p(src_ptr), C.c_uint32(src_sample_bytes), C.c_uint32(channels),
                                  C.c_uint32(bits_per_sample), C.c_uint32(container), C.c_uint64(wide_samples),
                                  C.c_void_p(dst_ptr), C.c_void_p(stream)))
