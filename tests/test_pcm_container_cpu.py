"""PCM container headers (host side of include/flac_amd_pcm.h) against files the reference `flac` read and wrote.

The fixtures (tests/golden/pcm_*.npz) hold, per case, an input file, the .flac the reference made of it and the
file the reference decoded that .flac to.  The parser must read from the input what the reference read (its
STREAMINFO says what that was), and the header writer must reproduce the decoded file's header byte for byte.
No device work here.
"""
import ctypes as C

import numpy as np
import pytest

import pcm_golden_util as G

CASES = G.load_index()
IDS = [c["name"] for c in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_parse_matches_what_the_reference_read(case):
    import flac_amd
    infile, flac, _decoded = G.load_case(case)
    fmt = flac_amd.pcm_parse_header(infile, case["channel_map_none"])
    _min_bs, _max_bs, rate, channels, bps, total, _audio = G.streaminfo(flac)
    assert (fmt.channels, fmt.sample_rate, fmt.bits_per_sample - fmt.shift, fmt.total_wide_samples) == (channels, rate, bps, total)
    assert fmt.bytes_per_wide_sample == channels * fmt.bits_per_sample // 8
    assert fmt.data_offset + fmt.data_bytes <= len(infile)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_sample_data_is_where_the_parser_says(case):
    """The reference's decoder wrote the same samples back: the data regions of the two files agree."""
    import flac_amd
    infile, _flac, decoded = G.load_case(case)
    a = flac_amd.pcm_parse_header(infile, case["channel_map_none"])
    b = flac_amd.pcm_parse_header(decoded, case["channel_map_none"])
    sa = G.samples_of(infile[a.data_offset:a.data_offset + a.data_bytes], a.channels, a.bits_per_sample, a.is_big_endian, a.is_unsigned, a.shift)
    sb = G.samples_of(decoded[b.data_offset:b.data_offset + b.data_bytes], b.channels, b.bits_per_sample, b.is_big_endian, b.is_unsigned, b.shift)
    assert sa.shape == sb.shape and np.array_equal(sa, sb)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_written_header_matches_the_reference_decoder(case):
    import flac_amd
    infile, flac, decoded = G.load_case(case)
    _min_bs, _max_bs, rate, channels, bps, total, _audio = G.streaminfo(flac)
    src = flac_amd.pcm_parse_header(infile, case["channel_map_none"])
    dst = flac_amd.pcm_parse_header(decoded, case["channel_map_none"])
    # decode.c:346-376: with --channel-map=none no mask is assigned; otherwise the default for the channel count
    mask = 0 if case["channel_map_none"] else G.DEFAULT_MASK.get(channels, 0)
    assert src.channels == channels
    hdr = flac_amd.pcm_write_header(dst.container, channels, bps, rate, mask, total)
    assert hdr == decoded[:dst.data_offset]
    assert len(decoded) >= dst.data_offset + total * channels * ((bps + 7) // 8)


def test_short_and_foreign_input():
    import flac_amd
    infile, _flac, _decoded = G.load_case(CASES[1])
    with pytest.raises(flac_amd.FlacAmdError, match="-5"):   # header cut short: ask for more bytes
        flac_amd.pcm_parse_header(infile[:30])
    with pytest.raises(flac_amd.FlacAmdError, match="-1"):
        flac_amd.pcm_parse_header(b"\0" * 64)
    with pytest.raises(flac_amd.FlacAmdError, match="ID3"):
        flac_amd.pcm_parse_header(b"ID3" + b"\0" * 61)


def test_refusals_follow_the_reference():
    """encode.c:350,380-391,413,435,501,756: what the reference reports as an ERROR is refused here."""
    import struct
    import flac_amd

    def wave(tag, ch, bits, extra=b"", cb=16):
        fmt = struct.pack("<HHIIHH", tag, ch, 44100, 44100 * ch * bits // 8, ch * bits // 8, bits) + extra
        body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", 8) + b"\0" * 8
        return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body

    ok = flac_amd.pcm_parse_header(wave(1, 2, 16))
    assert (ok.channels, ok.bits_per_sample, ok.data_bytes) == (2, 16, 8)
    for bad, what in ((wave(3, 2, 16), "format type"), (wave(1, 2, 20), "bits-per-sample"), (wave(1, 4, 16), "EXTENSIBLE"),
                      (wave(65534, 2, 16, struct.pack("<HHI", 22, 24, 3) + b"\x01\0" + b"\0" * 14), "wValidBitsPerSample"),
                      (wave(65534, 4, 16, struct.pack("<HHI", 22, 16, 0x0f) + b"\x01\0" + b"\0" * 14), "channel mask")):
        with pytest.raises(flac_amd.FlacAmdError, match=what):
            flac_amd.pcm_parse_header(bad)
    # --channel-map=none lets an unknown mask through (encode.c:487)
    fmt = flac_amd.pcm_parse_header(wave(65534, 4, 16, struct.pack("<HHI", 22, 16, 0x0f) + b"\x01\0" + b"\0" * 14), True)
    assert fmt.channels == 4 and fmt.channel_mask == 0x0f
