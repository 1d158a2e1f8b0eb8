"""CPU tests of the C-ABI boundary: the library loads without a GPU, exports every function the
headers declare, resolves settings like the reference's init, builds the reference's window
tables, and refuses to run without a device (there is no CPU code path behind it)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import golden_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    # a name followed by "(" is a function, unless the parenthesis opens a function-pointer typedef
    names = set(re.findall(r"\b((?:flacamd|FLAC__)\w+)\s*\((?!\s*\*)", text))
    return sorted(n for n in names if not n.isupper())


def declared_data(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^extern\s+const\s[^;(]*?\b(FLAC__\w+)\s*(?:\[[^\]]*\])?\s*;", text, flags=re.M)))


def test_every_declared_symbol_is_exported():
    import flac_amd
    lib = flac_amd.lib()
    headers = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert headers
    missing = []
    for h in headers:
        for name in declared_functions(h):
            if not hasattr(lib, name):
                missing.append("%s:%s" % (h, name))
        for name in declared_data(h):
            try:
                C.c_int.in_dll(lib, name)
            except ValueError:
                missing.append("%s:%s (data)" % (h, name))
    assert not missing, missing
    assert len(declared_functions("flac_amd_stream.h")) > 90 and len(declared_data("flac_amd_stream.h")) > 60


def test_presets_resolve_like_the_reference():
    import flac_amd
    # (level) -> (mid/side, loose, max lpc, exhaustive, max partition order, blocksize, precision)
    want = {0: (0, 0, 0, 0, 3, 1152, 10), 1: (1, 1, 0, 0, 3, 1152, 10), 2: (1, 0, 0, 0, 3, 1152, 10),
            3: (0, 0, 6, 0, 4, 4096, 12), 4: (1, 1, 8, 0, 4, 4096, 12), 5: (1, 0, 8, 0, 5, 4096, 12),
            6: (1, 0, 8, 0, 6, 4096, 12), 7: (1, 0, 8, 1, 6, 4096, 12), 8: (1, 0, 12, 1, 6, 4096, 12)}
    for lv, w in want.items():
        p = flac_amd.preset(lv, 2, 16, 44100)
        got = (p.do_mid_side_stereo, p.loose_mid_side_stereo, p.max_lpc_order, p.do_exhaustive_model_search,
               p.max_residual_partition_order, p.blocksize, p.qlp_coeff_precision)
        assert got == w, lv
    p = flac_amd.preset(8, 8, 24, 192000)
    assert (p.do_mid_side_stereo, p.qlp_coeff_precision) == (0, 15)
    with pytest.raises(flac_amd.FlacAmdError):
        flac_amd.preset(5, 9, 16, 44100)
    with pytest.raises(flac_amd.FlacAmdError):
        flac_amd.preset(5, 2, 16, 44100, blocksize=8)
    with pytest.raises(flac_amd.FlacAmdError):
        flac_amd.preset(5, 2, 25, 44100)


def test_windows_match_reference_tables():
    import flac_amd
    k = golden_util.load_kats()["windows_sha256"]
    shapes = {"tukey_0.5_4096": ("tukey", 0.5, 4096), "tukey_0.25_1000": ("tukey", 0.25, 1000),
              "hann_576": ("hann", 0, 576), "hamming_100": ("hamming", 0, 100), "blackman_100": ("blackman", 0, 100),
              "welch_33": ("welch", 0, 33), "triangle_33": ("triangle", 0, 33), "bartlett_32": ("bartlett", 0, 32)}
    for name, (apod, prm, n) in shapes.items():
        w = flac_amd.window(apod, prm, n)
        assert hashlib.sha256(w.tobytes()).hexdigest() == k[name], name


def test_no_device_means_loud_failure():
    import torch

    import flac_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = flac_amd.preset(5, 2, 16, 44100)
    with pytest.raises(flac_amd.FlacAmdError, match="HIP|device|GPU"):
        flac_amd.Encoder(p, device=0, max_frames=4)
    with pytest.raises(flac_amd.FlacAmdError, match="HIP|device|GPU"):
        flac_amd.Decoder(2, 16, 4096, device=0)
