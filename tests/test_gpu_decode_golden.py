"""GPU parity tests: the HIP decoder, through the C ABI, on streams written by reference libFLAC.

The decoded PCM must equal the PCM the reference encoded (and decoded back) bit for bit.
"""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu

CASES = golden_util.load_index()


def blocksize_of(case):
    if case["blocksize"]:
        return case["blocksize"]
    lpc = case["settings"].get("set_max_lpc_order")
    if lpc is None:
        lpc = 0 if case["level"] in (None, 0, 1, 2) else 1
    return 1152 if lpc == 0 else 4096


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_decode_matches_pcm(case):
    import flac_amd
    pcm, _meta, frames = golden_util.load_case(case)
    blob = b"".join(frames)
    n = pcm.shape[0]
    dec = flac_amd.Decoder(case["channels"], case["bps"], blocksize_of(case), device=0,
                           max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
    try:
        out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=4)
    finally:
        dec.close()
    assert bad == 0
    assert nfr == len(frames)
    assert wide == n
    assert np.array_equal(out, pcm)


def test_decode_many_streams_16bit_output():
    """Several streams in one buffer, decoded in one call into int16 output at given offsets."""
    import flac_amd
    names = ["s16_l8", "s16_l5", "s16_l0", "wasted3_l5", "mixed_l8"]
    blobs, descs, pcms = [], [], []
    off = soff = 0
    for nm in names:
        case = next(c for c in CASES if c["name"] == nm)
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        pad = (-len(blob)) % 4
        descs.append((off, len(blob), soff, pcm.shape[0]))
        blobs.append(blob + b"\0" * pad)
        pcms.append(pcm)
        off += len(blob) + pad
        soff += pcm.shape[0]
    # streams of one decoder share the largest blocksize (4096 here); 1152-blocksize streams have
    # their own frame numbering, so they get their own decoder
    sel = [i for i, nm in enumerate(names) if nm != "s16_l0"]
    buf = b"".join(blobs)
    dec = flac_amd.Decoder(2, 16, 4096, device=0, max_input_bytes=len(buf) + 64, max_frames=256)
    try:
        out, nfr, wide, bad = dec.decode_host(buf, [descs[i] for i in sel], soff, out_sample_bytes=2)
    finally:
        dec.close()
    assert bad == 0
    for i in sel:
        s0 = descs[i][2]
        assert np.array_equal(out[s0:s0 + pcms[i].shape[0]].astype(np.int32), pcms[i]), names[i]


def test_corrupt_frame_is_dropped_not_propagated():
    import flac_amd
    case = next(c for c in CASES if c["name"] == "s16_l8")
    pcm, _meta, frames = golden_util.load_case(case)
    blob = bytearray(b"".join(frames))
    hit = len(frames[0]) + len(frames[1]) // 2
    blob[hit] ^= 0x5a
    dec = flac_amd.Decoder(2, 16, 4096, device=0, max_input_bytes=len(blob) + 64, max_frames=64)
    try:
        out, nfr, wide, bad = dec.decode_host(bytes(blob), [(0, len(blob), 0, pcm.shape[0])], pcm.shape[0])
    finally:
        dec.close()
    assert nfr == len(frames) - 1 and bad >= 1
    assert np.array_equal(out[:4096], pcm[:4096])
    assert not out[4096:8192].any()          # the damaged frame reads as zeros (stream_decoder.c:2105-2113)
    assert np.array_equal(out[8192:], pcm[8192:])
