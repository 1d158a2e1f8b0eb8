"""The frame writer has three forms on the device: k_pack (any stream; a window of the frame's bits that
slides, one subframe after the other), k_pack_frame (full blocks of 16-bit stereo with mid/side,
the benchmark's shape; the whole frame in one window, both subframes in one pass) and k_pack_sub (a
workgroup per written subframe, the pieces meeting at any bit of the output; what every other stream
with blocks of up to 4096 samples takes).  All restate
process_frame_'s output (stream_encoder.c:2861: frame header, subframes, padding, CRC-16) for plans the
model search has made, so they must write the same bytes.  The golden cases pin whichever form their
shape selects to the reference's bytes; this test holds the two forms against each other on signals and
settings the golden set does not have at these block sizes: full-scale noise (verbatim subframes, the
longest frames there are), square waves (very long Rice codes next to very short ones), wasted bits,
constant and silent blocks, every channel assignment, Rice parameters above 14, partition-order
limits, fixed-only presets, block sizes below 4096, frame numbers whose UTF-8 form is one to five
bytes, and a short last block (which stays with k_pack)."""
import os

import numpy as np
import pytest

from test_gpu_search_forms import signals

pytestmark = pytest.mark.gpu


SETTINGS = [
    ("level1_loose", 1, {}),
    ("level2_fixed", 2, {}),
    ("level5", 5, {}),
    ("level8", 8, {}),
    ("level8_po66", 8, {"min_residual_partition_order": 6, "max_residual_partition_order": 6}),
    ("level8_po0", 8, {"min_residual_partition_order": 0, "max_residual_partition_order": 0}),
    ("level8_po8", 8, {"max_residual_partition_order": 8}),
    ("level8_prec12", 8, {"qlp_coeff_precision": 12}),
    ("level8_noverbatim", 8, {"disable_verbatim_subframes": 1}),
    ("level8_noconstant", 8, {"disable_constant_subframes": 1}),
    ("order32_prec10", 8, {"max_lpc_order": 32, "qlp_coeff_precision": 10}),   # the 32-sample history class of both forms
    ("order20_prec9_po4", 8, {"max_lpc_order": 20, "qlp_coeff_precision": 9, "max_residual_partition_order": 4}),
]


def encode(params, pcm, form, first_frame=0):
    import flac_amd
    if form in ("generic", "sub"):
        os.environ["FLAC_AMD_PACK"] = form
    else:
        os.environ.pop("FLAC_AMD_PACK", None)
    try:
        enc = flac_amd.Encoder(params, device=0, max_frames=64)
        try:
            frames, _ = enc.encode_host(pcm, first_frame=first_frame)
        finally:
            enc.close()
    finally:
        os.environ.pop("FLAC_AMD_PACK", None)
    return frames


@pytest.mark.parametrize("bs", [4096, 2048, 1024, 512, 256])
def test_frame_form_writes_what_the_window_form_writes(bs):
    import flac_amd
    sigs = signals(16, 2)
    for sname, level, over in SETTINGS:
        p = flac_amd.preset(level, 2, 16, 44100, bs, **over)
        assert p.blocksize == bs
        for name, pcm in sigs.items():
            pcm = pcm[:bs * 4 + bs // 3]
            a = encode(p, pcm, "generic")
            b = encode(p, pcm, "frame")
            assert len(a) == len(b) == 5
            for i, (fa, fb) in enumerate(zip(a, b)):
                assert fa == fb, "%s / %s / blocks of %d: frame %d differs between the two forms" % (sname, name, bs, i)


@pytest.mark.parametrize("first_frame", [0, 0x7f, 0x7fe, 0xfffe, 0x1ffffe, 0x3fffffe])
def test_frame_numbers_of_every_length(first_frame):
    import flac_amd
    pcm = signals(16, 2)["music"]
    p = flac_amd.preset(8, 2, 16, 44100, 4096)
    a = encode(p, pcm, "generic", first_frame)
    b = encode(p, pcm, "frame", first_frame)
    assert a == b


def test_odd_sample_rates():
    import flac_amd
    pcm = signals(16, 2)["music"]
    for rate in (44100, 96000, 11025, 12345, 65535, 655350, 64000):
        p = flac_amd.preset(8, 2, 16, rate, 4096)
        assert encode(p, pcm, "generic") == encode(p, pcm, "frame"), rate


def test_frame_form_against_the_oracle():
    """The same signals through the CPU restatement of the reference: the frame form's bytes are the
    reference's, not merely the other form's."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(16, 2)
    for level in (5, 8):
        p = flac_amd.preset(level, 2, 16, 44100, 4096)
        op = O.preset(level, 2, 16, 44100, 4096)
        for name, pcm in sigs.items():
            got = encode(p, pcm, "frame")
            want, _ = O.encode_frames(op, pcm)
            assert got == list(want), (level, name)


@pytest.mark.parametrize("bs", [1024, 256])
def test_both_forms_against_the_oracle_at_smaller_block_sizes(bs):
    """Blocks below 4096 samples: both writers against the CPU restatement of the reference, so that something the two
    forms share (the residual recomputation, the Rice writer's pieces) cannot be wrong in both unnoticed."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(16, 2)
    for sname, level, over in (("level8", 8, {}), ("level1_loose", 1, {}), ("level8_po8", 8, {"max_residual_partition_order": 8})):
        p = flac_amd.preset(level, 2, 16, 44100, bs, **over)
        op = O.preset(level, 2, 16, 44100, bs, **over)
        for name, pcm in sigs.items():
            pcm = pcm[:bs * 4 + bs // 3]
            want, _ = O.encode_frames(op, pcm)
            for form in ("generic", "frame"):
                assert encode(p, pcm, form) == list(want), (sname, name, bs, form)


def _encode_device(p, pcm16, nframes, shift, capacity=None, form=None):
    """Frames of one batch from device memory, the output starting `shift` bytes into its buffer."""
    import torch
    import flac_amd
    if form:
        os.environ["FLAC_AMD_PACK"] = form
    else:
        os.environ.pop("FLAC_AMD_PACK", None)
    try:
        dev = torch.device("cuda", 0)
        enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
        try:
            cap = enc.max_output_bytes(nframes) if capacity is None else capacity
            src = torch.from_numpy(pcm16).to(dev)
            out = torch.full((enc.max_output_bytes(nframes) + 64,), 0xA5, dtype=torch.uint8, device=dev)
            sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
            offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
            enc.encode_device(src.data_ptr(), 2, pcm16.shape[0], out.data_ptr() + shift, cap, sizes.data_ptr(), offs.data_ptr())
            torch.cuda.synchronize()
            enc.status()
            return out.cpu().numpy(), sizes.cpu().numpy(), offs.cpu().numpy()
        finally:
            enc.close()
    finally:
        os.environ.pop("FLAC_AMD_PACK", None)


@pytest.mark.parametrize("shift", [0, 1, 2, 3, 7, 13])
def test_frames_placed_directly_are_the_frames_copied_from_slots(shift):
    """k_pack_frame puts every frame straight where it belongs (sizes summed over the frames in front by a look-back
    over per-frame tickets); FLAC_AMD_PACK=slots keeps the slots, the scan and the copy.  Same bytes, sizes and places
    for 1 500 frames (more than the chip holds workgroups at once), wherever the output starts; not a byte outside."""
    import flac_amd
    from flac_amd import synth
    nframes = 1500
    p = flac_amd.preset(8, 2, 16, 44100, 4096)
    pcm = synth.synth_pcm(wide_samples=nframes * 4096 - 1000, sample_rate=44100, channels=2, bits_per_sample=16, seed=77).astype(np.int16)
    a, sa, oa = _encode_device(p, pcm, nframes, shift, form="slots")
    b, sb, ob = _encode_device(p, pcm, nframes, shift)
    assert np.array_equal(sa, sb) and np.array_equal(oa, ob)
    total = int(ob[nframes])
    assert total == int(sb.sum()) and np.array_equal(ob[:-1], np.concatenate(([0], np.cumsum(sb[:-1]))))
    assert np.array_equal(a[shift:shift + total], b[shift:shift + total])
    assert (b[:shift] == 0xA5).all() and (b[shift + total:] == 0xA5).all()


def test_direct_placement_reports_an_output_that_is_too_small():
    import flac_amd
    from flac_amd import synth
    p = flac_amd.preset(8, 2, 16, 44100, 4096)
    pcm = synth.synth_pcm(wide_samples=64 * 4096, sample_rate=44100, channels=2, bits_per_sample=16, seed=5).astype(np.int16)
    with pytest.raises(flac_amd.FlacAmdError):
        _encode_device(p, pcm, 64, 0, capacity=100000)


# ---- the subframe form ----

@pytest.mark.parametrize("bs", [4096, 1024, 256])
def test_subframe_form_writes_what_the_window_form_writes_16_bit_stereo(bs):
    """The settings and signals of the frame form's test, with every frame cut into its two subframes' workgroups."""
    import flac_amd
    sigs = signals(16, 2)
    for sname, level, over in SETTINGS:
        p = flac_amd.preset(level, 2, 16, 44100, bs, **over)
        for name, pcm in sigs.items():
            pcm = pcm[:bs * 4 + bs // 3]
            a = encode(p, pcm, "generic")
            b = encode(p, pcm, "sub")
            assert len(a) == len(b) == 5
            for i, (fa, fb) in enumerate(zip(a, b)):
                assert fa == fb, "%s / %s / blocks of %d: frame %d differs" % (sname, name, bs, i)


SHAPES = [(1, 16), (1, 24), (2, 24), (2, 8), (3, 16), (5, 20), (6, 24), (8, 24), (8, 12), (2, 17), (7, 4)]


@pytest.mark.parametrize("channels,bps", SHAPES)
def test_subframe_form_for_every_shape(channels, bps):
    """One to eight channels, 4 to 24 bits, stereo with and without mid/side: the default writer of these streams against the
    sliding window, frame by frame (pieces of 8 + bps bits for constant blocks up to whole verbatim subframes, so boundaries at
    every bit of a byte), and against the CPU restatement of the reference for a level each."""
    import flac_amd
    import oracle_lib as O
    sigs = signals(bps, channels)
    settings = [("level8", 8, {}), ("level5", 5, {}), ("level0", 0, {}), ("level3", 3, {}),
                ("level8_po8", 8, {"max_residual_partition_order": 8}), ("level8_p", 8, {"do_qlp_coeff_prec_search": 1})]
    for bs in (4096, 1152, 192):
        for sname, level, over in settings:
            p = flac_amd.preset(level, channels, bps, 48000, bs, **over)
            for name, pcm in sigs.items():
                pcm = np.ascontiguousarray(pcm[:bs * 3 + bs // 5])
                a = encode(p, pcm, "generic")
                b = encode(p, pcm, None)
                assert len(a) == len(b) == 4
                for i, (fa, fb) in enumerate(zip(a, b)):
                    assert fa == fb, "%s / %s / %d ch %d bits blocks of %d: frame %d differs" % (sname, name, channels, bps, bs, i)
    p = flac_amd.preset(8, channels, bps, 48000, 4096)
    op = O.preset(8, channels, bps, 48000, 4096)
    for name in ("music", "quiet", "wasted"):
        pcm = np.ascontiguousarray(sigs[name][:4096 * 2 + 77])
        want, _ = O.encode_frames(op, pcm)
        assert encode(p, pcm, None) == list(want), name


def _encode_device_wide(p, pcm32, nframes, shift, form=None, capacity=None):
    import torch
    import flac_amd
    if form:
        os.environ["FLAC_AMD_PACK"] = form
    else:
        os.environ.pop("FLAC_AMD_PACK", None)
    try:
        dev = torch.device("cuda", 0)
        enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
        try:
            cap = enc.max_output_bytes(nframes) if capacity is None else capacity
            src = torch.from_numpy(pcm32).to(dev)
            out = torch.full((enc.max_output_bytes(nframes) + 64,), 0xA5, dtype=torch.uint8, device=dev)
            sizes = torch.zeros(nframes, dtype=torch.int32, device=dev)
            offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
            enc.encode_device(src.data_ptr(), 4, pcm32.shape[0], out.data_ptr() + shift, cap, sizes.data_ptr(), offs.data_ptr())
            torch.cuda.synchronize()
            enc.status()
            return out.cpu().numpy(), sizes.cpu().numpy(), offs.cpu().numpy()
        finally:
            enc.close()
    finally:
        os.environ.pop("FLAC_AMD_PACK", None)


@pytest.mark.parametrize("channels,shift", [(8, 0), (8, 5), (3, 1), (2, 3), (1, 2)])
def test_subframe_form_places_a_long_batch(channels, shift):
    """More workgroups than the chip holds at once (the waits are for workgroups that started earlier): 700 frames of up to
    eight channels against the sliding window with its slots, scan and copy; sizes, places, bytes, and nothing outside."""
    import flac_amd
    from flac_amd import synth
    nframes = 700
    p = flac_amd.preset(8, channels, 24, 96000, 4096)
    pcm = synth.synth_pcm(wide_samples=nframes * 4096 - 1234, sample_rate=96000, channels=channels, bits_per_sample=24, seed=channels).astype(np.int32)
    a, sa, oa = _encode_device_wide(p, pcm, nframes, shift, form="generic")
    b, sb, ob = _encode_device_wide(p, pcm, nframes, shift)
    assert np.array_equal(sa, sb) and np.array_equal(oa, ob)
    total = int(ob[nframes])
    assert total == int(sb.sum())
    assert np.array_equal(a[shift:shift + total], b[shift:shift + total])
    assert (b[:shift] == 0xA5).all() and (b[shift + total:] == 0xA5).all()


def test_subframe_form_reports_an_output_that_is_too_small():
    import flac_amd
    from flac_amd import synth
    p = flac_amd.preset(8, 6, 24, 96000, 4096)
    pcm = synth.synth_pcm(wide_samples=64 * 4096, sample_rate=96000, channels=6, bits_per_sample=24, seed=5).astype(np.int32)
    with pytest.raises(flac_amd.FlacAmdError):
        _encode_device_wide(p, pcm, 64, 0, capacity=100000)


@pytest.mark.parametrize("channels,bps,bs", [(8, 24, 16384), (8, 24, 4608), (6, 24, 8192), (8, 16, 65535), (2, 24, 32768), (5, 20, 10000)])
def test_window_form_for_frames_of_many_windows(channels, bps, bs):
    """Blocks beyond 4096 samples stay with the sliding window, and with many channels a frame is many windows long (up to 1.5 MB):
    against the CPU restatement of the reference on signals whose subframes begin at every distance from a window's end (the fields
    in front of a subframe's codes are written once and need room in the window)."""
    import flac_amd
    import oracle_lib as O
    rng = np.random.default_rng(bs + channels)
    full = 1 << (bps - 1)
    n = bs * 2 + 999
    t = np.arange(n, dtype=np.float64)
    music = np.stack([0.5 * np.sin(2 * np.pi * (0.003 + 0.0017 * c) * t + c) + 0.1 * np.sin(2 * np.pi * 0.041 * t) for c in range(channels)], axis=1)
    sigs = {
        "music": (music * full * 0.9 + rng.normal(0, full / 300.0, size=(n, channels))),
        "loud": (music * full * 0.5 + rng.normal(0, full / 8.0, size=(n, channels))),
        "noise": rng.integers(-full, full, size=(n, channels)).astype(np.float64),
    }
    for level in (8, 5):
        p = flac_amd.preset(level, channels, bps, 96000, bs)
        op = O.preset(level, channels, bps, 96000, bs)
        for name, x in sigs.items():
            pcm = np.ascontiguousarray(np.clip(x, -full, full - 1).astype(np.int32))
            want, _ = O.encode_frames(op, pcm)
            # Full-scale noise in blocks of 65 535: the reference's 32-bit sums of residual magnitudes wrap (stream_encoder.c:3704), it
            # picks Rice parameters far too small and writes frames of 1.5 MB where verbatim would be 1.0 MB.  The device reserves what a
            # frame can take when its estimate holds, and says so when a frame does not fit - a limit, not a silent difference.
            verbatim = channels * ((bs * bps + 7) // 8 + 2) + 32
            if max(len(w) for w in want) > verbatim + bs * channels // 8:
                with pytest.raises(flac_amd.FlacAmdError):
                    encode(p, pcm, None)
                continue
            got = encode(p, pcm, None)
            assert len(got) == len(want) == 3
            for i in range(3):
                assert got[i] == bytes(want[i]), "%s level %d: frame %d differs" % (name, level, i)
