"""The reference decoder accepts every full-size stream the GPU encoder writes.

Each benchmark configuration is encoded at full size through the libFLAC-compatible stream API of
this library (seekable, so that finish() writes the MD5 and the totals into STREAMINFO), and the
whole stream is then handed to the REFERENCE's stream decoder (oracle/_ref/libFLAC_ref.so, compiled
from the reference's sources in the build container) with MD5 checking on: no error callback, every
sample equal to the input, finish() true - which is what `flac -t` checks.  The 16-bit `-8` stream
is in addition compared byte for byte with what the reference encoder writes for the same samples.
"""
import os

import numpy as np
import pytest

import libflac_ctypes as L
from test_gpu_configs_full_size import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")

CONFIGS = [
    # name, channels, bps, rate, seconds, extra settings, compare bytes with the reference encoder
    ("C2_l8_16bit_stereo_600s", 2, 16, 44100, 600, {}, True),
    ("C3_l8_e_p_24bit_96k_600s", 2, 24, 96000, 600, {"set_do_exhaustive_model_search": 1, "set_do_qlp_coeff_prec_search": 1}, False),
    ("C4_l8_8ch_24bit_192k_60s", 8, 24, 192000, 60, {}, False),
]


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libFLAC_ref.so has not been built")
    import flac_amd
    return L.load(flac_amd.LIB_PATH), L.load(REF_LIB)


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_reference_decoder_accepts_the_full_size_stream(libs, cfg):
    gpu, ref = libs
    name, channels, bps, rate, seconds, settings, compare_bytes = cfg
    wide = rate * seconds
    pcm = synth(77, wide, channels, bps, rate)
    meta, frames, ok, _writes = L.encode(gpu, pcm, bps, rate, level=8, settings=settings, chunk=1 << 20, seekable=True,
                                         total_samples_estimate=wide)
    assert ok and len(frames) == -(-wide // 4096)
    assert meta[26:42] != bytes(16), "STREAMINFO carries no MD5"
    stream = meta + b"".join(frames)
    got, errors, fine = L.decode(ref, stream, md5_check=True)
    assert errors == []
    assert fine, "the reference decoder's finish() failed: MD5 mismatch"
    assert got.shape == pcm.shape and np.array_equal(got, pcm)
    if compare_bytes:
        rmeta, rframes, rok, _ = L.encode(ref, pcm, bps, rate, level=8, settings=settings, chunk=1 << 20, seekable=True,
                                          total_samples_estimate=wide)
        assert rok and len(rframes) == len(frames)
        assert b"".join(rframes) == b"".join(frames), "frame bytes differ from the reference encoder's"
        assert rmeta == meta, "stream prologue (STREAMINFO with MD5, sizes; vendor comment) differs"
