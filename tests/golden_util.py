"""Loading of tests/golden/ (vectors produced by the real reference libFLAC; see oracle/gen_golden.py)."""
import json
import os
import re

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

APOD_CODES = {"tukey": 0, "hann": 1, "rectangle": 2, "hamming": 3, "blackman": 4, "welch": 5,
              "triangle": 6, "bartlett": 7}

APOD_NAMES = {v: k for k, v in APOD_CODES.items()}

SETTER_TO_FIELD = {
    "set_do_mid_side_stereo": "do_mid_side_stereo",
    "set_loose_mid_side_stereo": "loose_mid_side_stereo",
    "set_max_lpc_order": "max_lpc_order",
    "set_qlp_coeff_precision": "qlp_coeff_precision",
    "set_do_qlp_coeff_prec_search": "do_qlp_coeff_prec_search",
    "set_do_exhaustive_model_search": "do_exhaustive_model_search",
    "set_min_residual_partition_order": "min_residual_partition_order",
    "set_max_residual_partition_order": "max_residual_partition_order",
    "disable_constant_subframes": "disable_constant_subframes",
    "disable_fixed_subframes": "disable_fixed_subframes",
    "disable_verbatim_subframes": "disable_verbatim_subframes",
}


def load_index():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return json.load(f)


def load_kats():
    with open(os.path.join(GOLD, "kats.json")) as f:
        return json.load(f)


def load_case(case):
    z = np.load(os.path.join(GOLD, case["name"] + ".npz"))
    pcm = z["pcm"].astype(np.int32)
    sizes = z["sizes"].astype(np.int64)
    blob = z["frames"].tobytes()
    frames, pos = [], 0
    for s in sizes:
        frames.append(blob[pos:pos + int(s)])
        pos += int(s)
    return pcm, z["meta"].tobytes(), frames


def parse_apodization(spec):
    """'tukey(0.25);hann' -> [(code, param), ...] the way stream_encoder.c:1526 reads it."""
    out = []
    for tok in spec.split(";"):
        m = re.match(r"^tukey\(([^)]*)\)$", tok)
        if m:
            out.append((APOD_CODES["tukey"], float(m.group(1))))
        elif tok in APOD_CODES:
            out.append((APOD_CODES[tok], 0.0))
    return out


def preset_args(case):
    """(level, channels, bps, rate, blocksize) for a golden case."""
    level = 0 if case["level"] is None else case["level"]
    return level, case["channels"], case["bps"], case["rate"], case["blocksize"]


def field_overrides(case):
    """Encoder-settings overrides of a golden case as {resolved-parameter field: value}."""
    over = {}
    if case["level"] is None:
        # A freshly created reference encoder is NOT at preset 5: FLAC__stream_encoder_new() calls
        # set_defaults_() while the calloc'ed state still reads OK (0), so the
        # set_compression_level(5) inside it is refused (stream_encoder.c:557,589,1464,2161).
        # What is left is preset 0 with a maximum Rice partition order of 0.
        over["max_residual_partition_order"] = 0
    for k, v in case["settings"].items():
        if k == "set_apodization":
            over["apodizations"] = parse_apodization(v)
        else:
            over[SETTER_TO_FIELD[k]] = int(v)
    return over
