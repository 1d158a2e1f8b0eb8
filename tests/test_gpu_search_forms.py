"""The model search has two forms on the device: a workgroup per subframe (any block size, any
sample width) and a wave per subframe (blocks of 4096 samples of streams up to 16 bits, the
benchmark's shape).  Both restate process_subframe_ (stream_encoder.c:3164) and must choose the same
model for every subframe, so the frames they lead to must be the same bytes.  The golden cases pin the
workgroup form (and, where their block size is 4096, the wave form) to the reference's bytes; this test
holds the two forms against each other on signals and settings the golden set does not have at that
block size: full-scale noise and square waves (coefficients at their limits, the exact 64-bit path),
wasted bits, constant and near-silent blocks, precision search, partition-order limits, mono, 8- and
12-bit samples, and a short last block."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BS = 4096


def signals(bps, channels):
    rng = np.random.default_rng(bps * 10 + channels)
    n = BS * 4 + 1500
    full = 1 << (bps - 1)
    t = np.arange(n, dtype=np.float64)
    out = {}
    music = np.stack([0.4 * np.sin(2 * np.pi * (0.005 + 0.002 * c) * t + c) + 0.2 * np.sin(2 * np.pi * 0.031 * t) for c in range(channels)], axis=1)
    out["music"] = (music * full * 0.9 + rng.normal(0, full / 2000.0, size=(n, channels))).astype(np.int64)
    out["noise"] = rng.integers(-full, full, size=(n, channels))
    sq = np.where((np.arange(n) // 37) % 2 == 0, full - 1, -full)
    out["square"] = np.stack([np.roll(sq, 11 * c) for c in range(channels)], axis=1)
    out["wasted"] = (out["music"] >> 3) << 3
    quiet = rng.integers(-1, 2, size=(n, channels))
    quiet[BS:2 * BS] = 5          # one constant block
    quiet[2 * BS:3 * BS] = 0      # one silent block
    out["quiet"] = quiet
    imp = np.zeros((n, channels), dtype=np.int64)
    imp[rng.integers(0, n, size=40), rng.integers(0, channels, size=40)] = full - 1
    out["impulses"] = imp
    # a resonance right at the stability limit: large coefficients, small shifts
    x = np.zeros(n)
    e = rng.normal(0, 1, size=n)
    for i in range(2, n):
        x[i] = 1.9995 * x[i - 1] - 0.9999 * x[i - 2] + e[i]
    x = x / np.max(np.abs(x)) * (full - 1)
    out["resonance"] = np.stack([np.roll(x, 5 * c) for c in range(channels)], axis=1).astype(np.int64)
    return {k: np.clip(v, -full, full - 1).astype(np.int32) for k, v in out.items()}


SETTINGS = [
    ("level5", 5, {}),
    ("level8", 8, {}),
    ("level8_p", 8, {"do_qlp_coeff_prec_search": 1}),
    ("level8_po66", 8, {"min_residual_partition_order": 6, "max_residual_partition_order": 6}),
    ("level8_po23", 8, {"min_residual_partition_order": 2, "max_residual_partition_order": 3}),
    ("level8_po0", 8, {"min_residual_partition_order": 0, "max_residual_partition_order": 0}),
    ("level8_prec15", 8, {"qlp_coeff_precision": 15}),
    ("level6_guess", 6, {"max_lpc_order": 12}),
    ("level2_fixed", 2, {}),
]


def encode(params, pcm, form):
    import flac_amd
    if form == "generic":
        os.environ["FLAC_AMD_SEARCH"] = "generic"
    else:
        os.environ.pop("FLAC_AMD_SEARCH", None)
    try:
        enc = flac_amd.Encoder(params, device=0, max_frames=8)
        try:
            frames, _ = enc.encode_host(pcm)
        finally:
            enc.close()
    finally:
        os.environ.pop("FLAC_AMD_SEARCH", None)
    return frames


@pytest.mark.parametrize("bps,channels", [(16, 2), (16, 1), (8, 2), (12, 2), (16, 3)])
def test_wave_form_chooses_what_the_workgroup_form_chooses(bps, channels):
    import flac_amd
    sigs = signals(bps, channels)
    for sname, level, over in SETTINGS:
        p = flac_amd.preset(level, channels, bps, 44100, BS, **over)
        assert p.blocksize == BS
        for name, pcm in sigs.items():
            a = encode(p, pcm, "generic")
            b = encode(p, pcm, "wave")
            assert len(a) == len(b) == 5
            for i, (fa, fb) in enumerate(zip(a, b)):
                assert fa == fb, "%s / %s / %d-bit x %d: frame %d differs between the two search forms" % (sname, name, bps, channels, i)


def test_wave_form_against_the_oracle():
    """The same signals through the CPU restatement of the reference at -8: the wave form's frames
    are the reference's bytes, not merely the other form's."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(16, 2)
    p = flac_amd.preset(8, 2, 16, 44100, BS)
    op = O.preset(8, 2, 16, 44100, BS)
    for name, pcm in sigs.items():
        got = encode(p, pcm, "wave")
        want, _ = O.encode_frames(op, pcm)
        assert got == list(want), name


@pytest.mark.parametrize("bps,channels", [(16, 1), (8, 2), (12, 2), (16, 3), (24, 2), (24, 3), (20, 1)])
def test_both_forms_against_the_oracle_on_other_shapes(bps, channels):
    """Mono, three channels (no mid/side), 8-, 12-, 20- and 24-bit samples (the wide residual arithmetic): whichever forms
    take the shape are held against the CPU restatement of the reference, not only against each other."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(bps, channels)
    for sname, level, over in (("level8", 8, {}), ("level8_p", 8, {"do_qlp_coeff_prec_search": 1}), ("level5", 5, {})):
        p = flac_amd.preset(level, channels, bps, 44100, BS, **over)
        op = O.preset(level, channels, bps, 44100, BS, **over)
        for name, pcm in sigs.items():
            want, _ = O.encode_frames(op, pcm)
            for form in ("generic", "wave"):
                assert encode(p, pcm, form) == list(want), (sname, name, bps, channels, form)


def test_fixed_predictors_on_a_side_signal_of_25_bits():
    """-l 0 on 24-bit stereo with mid/side, blocks beyond 4096 (the workgroup form of the search): a side signal that uses its 25th
    bit (full-scale channels in opposite phase) is not for 24-bit multiplies; against the CPU restatement of the reference."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(24, 2)
    for bs in (16384, 8192):
        for level in (2, 1):
            p = flac_amd.preset(level, 2, 24, 48000, bs)
            op = O.preset(level, 2, 24, 48000, bs)
            for name in ("square", "noise", "music"):
                pcm = np.ascontiguousarray(sigs[name])
                want, _ = O.encode_frames(op, pcm)
                enc = flac_amd.Encoder(p, device=0, max_frames=8)
                try:
                    got, _ = enc.encode_host(pcm)
                finally:
                    enc.close()
                assert got == [bytes(w) for w in want], (bs, level, name)
