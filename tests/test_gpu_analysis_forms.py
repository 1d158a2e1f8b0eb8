"""Streams of three or more channels are laid out again as planes before anything else reads them, and the pass that does it
(for one or two channels: the same pass without the planes, with mid and side formed on the way) also makes the statistics of get_wasted_bits_ (stream_encoder.c:4108), FLAC__fixed_compute_best_predictor[_wide] (fixed.c:224,282)
and the all-samples-equal test (stream_encoder.c:3210-3227): sums over the samples as they are, shifted by the wasted bits once
those are known.  FLAC_AMD_ANALYZE=split keeps the two passes apart (k_planes, then k_analyze over the planes with the wasted bits
taken out first).  Both must lead to the same frames: the statistics decide the wasted bits, the constant subframes, the fixed
predictor's order guess and its price."""
import os

import numpy as np
import pytest

from test_gpu_search_forms import signals

pytestmark = pytest.mark.gpu

SHAPES = [(3, 16), (4, 24), (5, 20), (6, 24), (8, 24), (8, 8), (7, 12), (3, 4), (2, 24), (2, 20), (2, 8), (1, 16), (1, 24), (2, 16)]


def encode(params, pcm, form):
    import flac_amd
    if form:
        os.environ["FLAC_AMD_ANALYZE"] = form
    else:
        os.environ.pop("FLAC_AMD_ANALYZE", None)
    try:
        enc = flac_amd.Encoder(params, device=0, max_frames=64)
        try:
            frames, _ = enc.encode_host(pcm)
        finally:
            enc.close()
    finally:
        os.environ.pop("FLAC_AMD_ANALYZE", None)
    return frames


@pytest.mark.parametrize("channels,bps", SHAPES)
def test_one_pass_statistics_lead_to_the_same_frames(channels, bps):
    import flac_amd
    sigs = signals(bps, channels)
    # more wasted bits, and a channel that is constant from the second block on
    deep = (sigs["music"] >> (bps // 2)) << (bps // 2)
    deep[4096:, 0] = deep[4096, 0]
    sigs["deep_wasted"] = deep
    for bs in (4096, 1152, 576, 192, 16) + ((16384,) if channels <= 2 else ()):
        for level, over in ((8, {}), (5, {}), (2, {}), (0, {}), (1, {}), (8, {"disable_constant_subframes": 1})):
            p = flac_amd.preset(level, channels, bps, 48000, bs, **over)
            for name, pcm in sigs.items():
                pcm = np.ascontiguousarray(pcm[:bs * 3 + bs // 5 + 2])
                a = encode(p, pcm, "split")
                b = encode(p, pcm, None)
                assert len(a) == len(b) == (len(pcm) + bs - 1) // bs
                for i, (fa, fb) in enumerate(zip(a, b)):
                    assert fa == fb, "level %d / %s / %d ch %d bits blocks of %d: frame %d differs" % (level, name, channels, bps, bs, i)


def test_one_pass_statistics_against_the_oracle():
    import flac_amd
    import oracle_lib as O
    for channels, bps in ((8, 24), (3, 16), (6, 20), (2, 24), (1, 24), (2, 12)):
        sigs = signals(bps, channels)
        p = flac_amd.preset(8, channels, bps, 96000, 4096)
        op = O.preset(8, channels, bps, 96000, 4096)
        for name in ("music", "quiet", "wasted", "noise"):
            pcm = np.ascontiguousarray(sigs[name][:4096 * 2 + 333])
            want, _ = O.encode_frames(op, pcm)
            assert encode(p, pcm, None) == list(want), (channels, bps, name)
