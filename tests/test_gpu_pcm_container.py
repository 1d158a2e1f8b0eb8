"""GPU parity tests of the PCM container stage (flacamd_pcm_unpack / flacamd_pcm_pack) through the C ABI.

Encode direction: the sample bytes of a WAVE / RF64 / Wave64 / AIFF / AIFF-C file go to the device as they are in
the file, are unpacked there and encoded; every frame must be byte-identical to what the reference `flac` wrote
for the same file.  Decode direction: the reference's .flac is decoded on the device and packed into file bytes,
which must equal the data the reference's decoder wrote.  Fixtures: tests/golden/pcm_*.npz.
"""
import numpy as np
import pytest

import pcm_golden_util as G

pytestmark = pytest.mark.gpu

CASES = G.load_index()
IDS = [c["name"] for c in CASES]


def _unpack_on_device(torch, flac_amd, infile, fmt, dst_bytes, misalign=0):
    dev = torch.device("cuda", 0)
    data = np.frombuffer(infile, dtype=np.uint8)[fmt.data_offset:fmt.data_offset + fmt.data_bytes]
    n = fmt.total_wide_samples
    buf = torch.zeros(len(data) + 16, dtype=torch.uint8, device=dev)
    buf[misalign:misalign + len(data)] = torch.from_numpy(data.copy()).to(dev)
    out = torch.zeros(n * fmt.channels, dtype=torch.int16 if dst_bytes == 2 else torch.int32, device=dev)
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    flac_amd.pcm_unpack(buf.data_ptr() + misalign, fmt, n, out.data_ptr(), dst_bytes, flag.data_ptr())
    torch.cuda.synchronize()
    return out, int(flag[0].item())


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_unpack_then_encode_matches_reference_flac(case):
    import torch
    import flac_amd
    infile, flac, _decoded = G.load_case(case)
    fmt = flac_amd.pcm_parse_header(infile, case["channel_map_none"])
    bps = fmt.bits_per_sample - fmt.shift
    want = G.samples_of(infile[fmt.data_offset:fmt.data_offset + fmt.data_bytes], fmt.channels, fmt.bits_per_sample,
                        fmt.is_big_endian, fmt.is_unsigned, fmt.shift)
    for dst_bytes in ((2, 4) if bps <= 16 else (4,)):
        for misalign in (0, 1, 2):
            out, flag = _unpack_on_device(torch, flac_amd, infile, fmt, dst_bytes, misalign)
            assert flag == 0
            assert np.array_equal(out.cpu().numpy().astype(np.int32).reshape(-1, fmt.channels), want)
    # the unpacked samples, still on the device, through the frame encoder at the reference's settings
    dst_bytes = 2 if bps <= 16 else 4
    out, _ = _unpack_on_device(torch, flac_amd, infile, fmt, dst_bytes)
    _min_bs, max_bs, rate, channels, sbps, total, audio = G.streaminfo(flac)
    assert sbps == bps
    p = flac_amd.preset(8, channels, bps, rate, max_bs)
    nframes = (total + max_bs - 1) // max_bs
    enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
    try:
        cap = enc.max_output_bytes(nframes)
        dev = out.device
        obuf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        fb = torch.zeros(nframes, dtype=torch.int32, device=dev)
        offs = torch.zeros(nframes + 1, dtype=torch.int64, device=dev)
        nf = enc.encode_device(out.data_ptr(), dst_bytes, total, obuf.data_ptr(), cap, fb.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        enc.status()
        assert nf == nframes
        got = obuf[:int(offs[nframes].item())].cpu().numpy().tobytes()
    finally:
        enc.close()
    assert got == flac[audio:]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_decode_then_pack_matches_reference_file(case):
    import torch
    import flac_amd
    infile, flac, decoded = G.load_case(case)
    _min_bs, max_bs, rate, channels, bps, total, audio = G.streaminfo(flac)
    dst = flac_amd.pcm_parse_header(decoded, case["channel_map_none"])
    dev = torch.device("cuda", 0)
    frames = np.frombuffer(flac, dtype=np.uint8)[audio:]
    dec = flac_amd.Decoder(channels, bps, max_bs, device=0, max_input_bytes=len(frames) + 64, max_frames=64)
    fbytes = (bps + 7) // 8
    for sample_bytes in ((2, 4) if bps <= 16 else (4,)):
        fdev = torch.from_numpy(frames.copy()).to(dev)
        pcm = torch.zeros(total * channels, dtype=torch.int16 if sample_bytes == 2 else torch.int32, device=dev)
        dec.decode_device(fdev.data_ptr(), len(frames), [(0, len(frames), 0, total)], pcm.data_ptr(), sample_bytes, total)
        torch.cuda.synchronize()
        nfr, wide, bad = dec.result()
        assert (wide, bad) == (total, 0)
        for misalign in (0, 1):
            outb = torch.zeros(total * channels * fbytes + 16, dtype=torch.uint8, device=dev)
            flac_amd.pcm_pack(pcm.data_ptr(), sample_bytes, channels, bps, dst.container, total, outb.data_ptr() + misalign)
            torch.cuda.synchronize()
            got = outb[misalign:misalign + total * channels * fbytes].cpu().numpy().tobytes()
            assert got == decoded[dst.data_offset:dst.data_offset + total * channels * fbytes]
            assert int(outb[misalign + total * channels * fbytes:].max().item()) == 0   # nothing written past the end
    dec.close()


def test_unpack_flags_nonzero_dead_bits():
    """encode.c:2411-2420: samples whose `shift` low bits are not zero are an error."""
    import torch
    import flac_amd
    case = next(c for c in CASES if c["name"] == "wavext20_stereo")
    infile, _flac, _decoded = G.load_case(case)
    fmt = flac_amd.pcm_parse_header(infile)
    assert fmt.shift == 4
    spoiled = bytearray(infile)
    spoiled[fmt.data_offset + 3 * 777] |= 0x01   # lowest byte of a little-endian 24-bit sample
    _out, flag = _unpack_on_device(torch, flac_amd, bytes(spoiled), fmt, 4)
    assert flag == 1


def test_large_unpack_pack_round_trip():
    """Ten minutes of 24-bit stereo: pack(unpack(x)) == x for both byte orders, at a size that fills the chip."""
    import torch
    import flac_amd
    dev = torch.device("cuda", 0)
    n, ch = 57_600_001, 2
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    data = torch.randint(0, 256, (n * ch * 3,), dtype=torch.uint8, device=dev, generator=g)
    for container, be in ((flac_amd.PCM_WAVE, 0), (flac_amd.PCM_AIFF, 1)):
        fmt = flac_amd.PcmFormat(container=container, channels=ch, sample_rate=96000, bits_per_sample=24, shift=0,
                                 is_big_endian=be, is_unsigned=0, bytes_per_wide_sample=6)
        mid = torch.empty(n * ch, dtype=torch.int32, device=dev)
        flag = torch.zeros(4, dtype=torch.int32, device=dev)
        flac_amd.pcm_unpack(data.data_ptr(), fmt, n, mid.data_ptr(), 4, flag.data_ptr())
        back = torch.zeros_like(data)
        flac_amd.pcm_pack(mid.data_ptr(), 4, ch, 24, container, n, back.data_ptr())
        torch.cuda.synchronize()
        assert int(mid.max().item()) < (1 << 23) and int(mid.min().item()) >= -(1 << 23)
        assert torch.equal(back, data)
