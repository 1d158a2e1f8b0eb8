"""ReplayGain analysis against the reference's (oracle/_ref/libreplaygain_ref.so, compiled from the reference's sources by
oracle/Makefile): the same samples, fed in the same pieces, must give the same gains and peaks, bit for bit as floats.
Host code only.  When the reference library has not been built, the committed answers in tests/golden/replaygain.json
(written by this file run as a script where the reference is present) are used instead."""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "flac_amd", "lib", "libflac_amd.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libreplaygain_ref.so")
GOLD = os.path.join(ROOT, "tests", "golden", "replaygain.json")

RATES = [8000, 11025, 12000, 16000, 18900, 22050, 24000, 32000, 37800, 44100, 48000, 56000, 64000, 88200, 96000, 112000, 128000,
         144000, 176400, 192000]


def bind(path):
    lib = C.CDLL(path)
    lib.InitGainAnalysis.argtypes = [C.c_long]
    lib.ResetSampleFrequency.argtypes = [C.c_long]
    lib.AnalyzeSamples.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t, C.c_int]
    lib.GetTitleGain.restype = C.c_float
    lib.GetAlbumGain.restype = C.c_float
    lib.grabbag__replaygain_analyze.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.c_int, C.c_uint, C.c_uint]
    lib.grabbag__replaygain_get_album.argtypes = [C.POINTER(C.c_float)] * 2
    lib.grabbag__replaygain_get_title.argtypes = [C.POINTER(C.c_float)] * 2
    return lib


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def signal(rate, seconds, channels, bps, seed, level):
    rng = np.random.default_rng(seed)
    n = int(rate * seconds)
    t = np.arange(n) / rate
    full = 1 << (bps - 1)
    out = []
    for c in range(channels):
        x = level * (0.5 * np.sin(2 * np.pi * (300 + 170 * c) * t) + 0.3 * np.sin(2 * np.pi * 2500 * t + c) + 0.1 * rng.standard_normal(n))
        out.append(np.clip(np.round(x * full), -full, full - 1).astype(np.int32))
    return out


def script(lib):
    """Every answer of one fixed series of calls, as float bit patterns."""
    out = {}
    out["valid"] = [lib.grabbag__replaygain_is_valid_sample_frequency(r) for r in (0, 1, 8000, 44100, 44101, 48000, 96000, 192000, 384000)]
    out["bad_rate"] = [lib.InitGainAnalysis(44101), lib.grabbag__replaygain_init(7)]
    # core API, every sample rate, stereo and mono, odd piece sizes (shorter and longer than the filter order)
    core = []
    for k, rate in enumerate(RATES):
        assert lib.InitGainAnalysis(rate) == 1
        for track in range(2):
            chans = signal(rate, 0.9, 2 if (k + track) % 2 == 0 else 1, 16, 100 * k + track, 0.2 + 0.5 * track)
            fl = [np.ascontiguousarray(c, dtype=np.float32) for c in chans]
            pos, piece = 0, [3, 1, 9, 10, 11, 257, 4096, 2, 7777]
            i = 0
            while pos < len(fl[0]):
                n = min(piece[i % len(piece)], len(fl[0]) - pos)
                i += 1
                l = fl[0][pos:pos + n].ctypes.data_as(C.POINTER(C.c_float))
                r = fl[1][pos:pos + n].ctypes.data_as(C.POINTER(C.c_float)) if len(fl) > 1 else None
                assert lib.AnalyzeSamples(l, r, n, len(fl)) == 1
                pos += n
            core.append(bits(lib.GetTitleGain()))
        core.append(bits(lib.GetAlbumGain()))
    out["core"] = core
    # too little for one 50 ms window, and three channels
    assert lib.InitGainAnalysis(44100) == 1
    few = np.zeros(100, dtype=np.float32)
    p = few.ctypes.data_as(C.POINTER(C.c_float))
    out["short"] = [lib.AnalyzeSamples(p, p, 100, 2), bits(lib.GetTitleGain()), bits(lib.GetAlbumGain()), lib.AnalyzeSamples(p, p, 100, 3), lib.AnalyzeSamples(p, p, 0, 3)]
    # the wrappers: integer samples of several widths, gains and peaks per title and per album
    wrap = []
    for bps in (8, 12, 16, 20, 24):
        for stereo in (1, 0):
            assert lib.grabbag__replaygain_init(48000) == 1
            for track in range(2):
                chans = signal(48000, 1.3, 2 if stereo else 1, bps, 7 * bps + track, 0.1 + 0.8 * track)
                pos = 0
                while pos < len(chans[0]):
                    n = min(4608 if track else 1152, len(chans[0]) - pos)
                    arr = (C.POINTER(C.c_int32) * 2)()
                    keep = [np.ascontiguousarray(c[pos:pos + n]) for c in chans]
                    for c, a in enumerate(keep):
                        arr[c] = a.ctypes.data_as(C.POINTER(C.c_int32))
                    assert lib.grabbag__replaygain_analyze(arr, stereo, bps, n) == 1
                    pos += n
                g, pk = C.c_float(), C.c_float()
                lib.grabbag__replaygain_get_title(C.byref(g), C.byref(pk))
                wrap += [bits(g.value), bits(pk.value)]
            g, pk = C.c_float(), C.c_float()
            lib.grabbag__replaygain_get_album(C.byref(g), C.byref(pk))
            wrap += [bits(g.value), bits(pk.value)]
    out["wrap"] = wrap
    out["loudness"] = bits(C.c_float.in_dll(lib, "ReplayGainReferenceLoudness").value)
    return out


def expected():
    if os.path.exists(REF):
        return script(bind(REF)), "reference library"
    with open(GOLD) as f:
        return json.load(f), "committed answers"


def test_gains_and_peaks_match_the_reference():
    want, source = expected()
    got = script(bind(OURS))
    assert got == want, source
    # the answers are not trivial: a full histogram walk gives plausible gains
    gains = [struct.unpack("<f", struct.pack("<I", b))[0] for b in got["core"]]
    assert all(-30.0 < g < 40.0 for g in gains) and len(set(got["core"])) > 30
    assert struct.unpack("<f", struct.pack("<I", got["short"][1]))[0] == -24601.0


def test_committed_answers_are_the_reference_s():
    if not os.path.exists(REF):
        pytest.skip("the reference library has not been built (oracle/Makefile)")
    with open(GOLD) as f:
        assert json.load(f) == script(bind(REF))


if __name__ == "__main__":
    with open(GOLD, "w") as f:
        json.dump(script(bind(REF)), f)
    print("wrote", GOLD)
