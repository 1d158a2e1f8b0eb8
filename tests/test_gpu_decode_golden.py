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


def _crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def _header_bytes(f):
    """Length of the frame header of f, CRC-8 included, as the header's own codes give it (frame.c layout)."""
    x = f[4]
    n = 4 + (1 if x < 0x80 else 2 if x < 0xe0 else 3 if x < 0xf0 else 4 if x < 0xf8 else 5 if x < 0xfc else 6 if x < 0xfe else 7)
    bs, sr = f[2] >> 4, f[2] & 15
    n += 1 if bs == 6 else 2 if bs == 7 else 0
    n += 1 if sr == 12 else 2 if sr in (13, 14) else 0
    return n + 1


def _damage(frame, rng):
    """One frame with the same length, damaged so that it is no longer a valid frame (the CRC-16 covers every bit)."""
    f = bytearray(frame)
    hb = _header_bytes(f)
    kind = rng.integers(0, 7)
    if kind == 0:      # a few flipped bits after the header, within 2 KiB of each other: two flipped bits 32767 bits
        #               apart, the period of the CRC-16 polynomial, make a frame that is valid for the reference too
        at = int(rng.integers(hb, len(f)))
        for _ in range(int(rng.integers(1, 4))):
            f[int(rng.integers(at, min(at + 2048, len(f))))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # other blocksize / sample rate codes under a header CRC that is right again
        f[2] = int(rng.integers(0x10, 0xff))
        if (f[2] & 15) == 15:
            f[2] &= 0xfe
        hb = min(_header_bytes(f), len(f) - 1)
        f[hb - 1] = _crc8(f[:hb - 1])
    elif kind == 2:    # another channel assignment, same channel count where there is one, header CRC right again
        f[3] = (f[3] & 0x0f) | (int(rng.integers(8, 11)) << 4 if f[3] >> 4 in (1, 8, 9, 10) else f[3] & 0xf0)
        f[hb - 1] = _crc8(f[:hb - 1])
        f[int(rng.integers(hb, len(f)))] ^= 0x10
    elif kind == 3:    # any subframe type byte: reserved types, LPC order 32, wasted-bits flag
        f[hb] = int(rng.integers(0, 256))
        f[hb + 1] ^= 0xff
    elif kind == 4:    # the rest of the frame is zeros: one unary run to the end of the frame and beyond
        at = int(rng.integers(hb, len(f)))
        f[at:] = bytes(len(f) - at)
    elif kind == 5:    # the rest of the frame is ones: empty unary codes, escape codes, 31-bit parameters
        at = int(rng.integers(hb, len(f)))
        f[at:] = b"\xff" * (len(f) - at)
    else:              # the rest of the frame is noise
        at = int(rng.integers(hb, len(f)))
        f[at:] = rng.integers(0, 256, len(f) - at, dtype=np.uint8).tobytes()
    if bytes(f) == bytes(frame):
        f[-1] ^= 1
    return bytes(f)


def _synthetic_stream(channels, bps, rate, bs, level, nframes, seed):
    """PCM (tones plus noise, one silent stretch) and its frames from this library's own encoder, whose output the
    encoder tests pin to the reference's."""
    import flac_amd
    rng = np.random.default_rng(seed)
    n = nframes * bs - bs // 3
    t = np.arange(n)[:, None]
    amp = (1 << (bps - 1)) * 0.4
    pcm = amp * np.sin(t * (0.01 + 0.003 * np.arange(channels)[None, :])) + rng.normal(0, amp / 50, (n, channels))
    pcm[5 * bs:6 * bs + 7] = 0
    pcm = np.clip(np.rint(pcm), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)
    enc = flac_amd.Encoder(flac_amd.preset(level, channels, bps, rate, bs), device=0, max_frames=nframes)
    try:
        frames, _ = enc.encode_host(pcm)
    finally:
        enc.close()
    assert len(frames) == nframes
    return pcm, frames


@pytest.mark.parametrize("channels,bps,rate,bs,level", [(2, 16, 44100, 4096, 8), (2, 24, 96000, 4096, 8), (8, 24, 96000, 1152, 5),
                                                        (1, 8, 22050, 576, 5), (2, 16, 44100, 257, 3), (2, 16, 44100, 4096, 0)])
def test_damaged_frames_leave_their_neighbours_alone(channels, bps, rate, bs, level):
    """Every second frame damaged in a random way, several times over: the damaged frames are dropped (zeros,
    stream_decoder.c:2105-2113), the frames between them come out bit-exact, and nothing reads or writes out of bounds."""
    import flac_amd
    nframes = 48
    pcm, frames = _synthetic_stream(channels, bps, rate, bs, level, nframes, bs + bps)
    n = pcm.shape[0]
    want = pcm.copy()
    for k in range(1, nframes, 2):
        want[k * bs:(k + 1) * bs] = 0
    dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=sum(map(len, frames)) + 64, max_frames=4 * nframes + 64)
    try:
        for seed in range(40):
            rng = np.random.default_rng(seed)
            blob = b"".join(_damage(f, rng) if k % 2 else f for k, f in enumerate(frames))
            out, nfr, _wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=4)
            assert nfr == nframes // 2 and bad >= 1, "seed %d: %d frames, %d bad" % (seed, nfr, bad)
            assert np.array_equal(out, want), "seed %d" % seed
    finally:
        dec.close()


def test_lane_per_frame_form_decodes_every_golden_stream(monkeypatch):
    """Small batches of streams with two or more channels are decoded in the split form (a walk that finds where every
    subframe starts, then a lane per subframe), which is what the other tests therefore run; FLAC_AMD_DEC_MODE=lane keeps
    the lane-per-frame form that large batches use.  It must give back every golden stream exactly as well."""
    import flac_amd
    monkeypatch.setenv("FLAC_AMD_DEC_MODE", "lane")
    for case in CASES:
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        dec = flac_amd.Decoder(case["channels"], case["bps"], blocksize_of(case), device=0, max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
        try:
            got, nfr, nwide, bad = dec.decode_host(blob, [(0, len(blob), 0, pcm.shape[0])], pcm.shape[0], out_sample_bytes=4)
        finally:
            dec.close()
        assert (nfr, nwide, bad) == (len(frames), pcm.shape[0], 0), case["name"]
        assert np.array_equal(got, pcm), case["name"]


@pytest.mark.parametrize("channels,bps,rate,bs,level", [(2, 16, 44100, 4096, 8), (8, 24, 96000, 1152, 5), (3, 16, 44100, 257, 3)])
def test_split_and_lane_forms_agree_on_damaged_streams(monkeypatch, channels, bps, rate, bs, level):
    """The two forms of the frame decoder on streams with every second frame damaged: the same samples, the same
    number of good and bad frames, and the same per-frame records (status, end of frame), whichever channel's
    subframe the damage sits in."""
    import flac_amd
    nframes = 32
    pcm, frames = _synthetic_stream(channels, bps, rate, bs, level, nframes, 7 * bs + bps)
    n = pcm.shape[0]
    results = {}
    for mode in ("lane", "split"):
        monkeypatch.setenv("FLAC_AMD_DEC_MODE", mode)
        dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=sum(map(len, frames)) + 64, max_frames=4 * nframes + 64)
        try:
            res = []
            for seed in range(25):
                rng = np.random.default_rng(1000 + seed)
                blob = b"".join(_damage(f, rng) if k % 2 else f for k, f in enumerate(frames))
                out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=4)
                info = dec.frames()
                res.append((out.copy(), nfr, wide, bad, [(int(r.pos), int(r.status), int(r.end_byte) if r.status in (1, 3) else 0) for r in info]))
            results[mode] = res
        finally:
            dec.close()
    for seed, (a, b) in enumerate(zip(results["lane"], results["split"])):
        assert a[1:4] == b[1:4], seed
        assert np.array_equal(a[0], b[0]), seed
        assert a[4] == b[4], seed


def test_pair_form_decodes_every_stereo_golden_stream(monkeypatch):
    """Large stereo batches in the split form take both channels of a frame into one workgroup, which writes final
    interleaved PCM itself for the frames whose CRC-16 the pass that runs ahead has found right (no planes, no output
    kernel for them).  FLAC_AMD_DEC_PAIR=2 selects that form from two frames up: every stereo golden stream comes back
    exactly, at both output widths."""
    import flac_amd
    monkeypatch.setenv("FLAC_AMD_DEC_PAIR", "2")
    seen = 0
    for case in CASES:
        if case["channels"] != 2:
            continue
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        for width in ((2, 4) if case["bps"] <= 16 else (4,)):
            dec = flac_amd.Decoder(2, case["bps"], blocksize_of(case), device=0, max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
            try:
                got, nfr, nwide, bad = dec.decode_host(blob, [(0, len(blob), 0, pcm.shape[0])], pcm.shape[0], out_sample_bytes=width)
                rows = dec.frames()
            finally:
                dec.close()
            assert (nfr, nwide, bad) == (len(frames), pcm.shape[0], 0), case["name"]
            assert np.array_equal(got, pcm), case["name"]
            assert all(r.reserved & 2 for r in rows), "%s: a frame did not take the direct way out" % case["name"]
        seen += 1
    assert seen > 10


def test_pair_and_plane_forms_agree_on_damaged_streams(monkeypatch):
    """Every second frame damaged: the pair form gives the samples, counts and per-frame records the plane form gives
    (a damaged frame's CRC-16 is not right at its guessed end, so it goes the planes' way and is dropped there)."""
    import flac_amd
    channels, bps, rate, bs, level, nframes = 2, 16, 44100, 4096, 8, 32
    pcm, frames = _synthetic_stream(channels, bps, rate, bs, level, nframes, 7 * bs + bps)
    n = pcm.shape[0]
    results = {}
    for pair in ("0", "2"):
        monkeypatch.setenv("FLAC_AMD_DEC_PAIR", pair)
        dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=sum(map(len, frames)) + 64, max_frames=4 * nframes + 64)
        try:
            res = []
            for seed in range(25):
                rng = np.random.default_rng(2000 + seed)
                blob = b"".join(_damage(f, rng) if k % 2 else f for k, f in enumerate(frames))
                out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=2)
                info = dec.frames()
                res.append((out.copy(), nfr, wide, bad, [(int(r.pos), int(r.status), int(r.end_byte) if r.status in (1, 3) else 0) for r in info]))
            results[pair] = res
        finally:
            dec.close()
    for seed, (a, b) in enumerate(zip(results["0"], results["2"])):
        assert a[1:4] == b[1:4], seed
        assert np.array_equal(a[0], b[0]), seed
        assert a[4] == b[4], seed


def test_planar_output_holds_the_interleaved_samples_frame_by_frame():
    """flacamd_decoder_set_planar_output: every frame's stretch of the output (blocksize x channels samples, where its
    interleaved samples would lie) holds channel 0's samples, then channel 1's, ...; nothing else changes.  Every golden
    stream, both output widths where they apply."""
    import flac_amd
    for case in CASES:
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        ch, n = case["channels"], pcm.shape[0]
        for width in ((2, 4) if case["bps"] <= 16 else (4,)):
            dec = flac_amd.Decoder(ch, case["bps"], blocksize_of(case), device=0, max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
            try:
                dec.set_planar_output(True)
                got, nfr, nwide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=width)
                rows = dec.frames()
            finally:
                dec.close()
            assert (nfr, nwide, bad) == (len(frames), n, 0), case["name"]
            flat = got.reshape(-1)
            for r in rows:
                s0, bs = int(r.sample), int(r.blocksize)
                block = flat[s0 * ch:(s0 + bs) * ch].reshape(ch, bs)
                assert np.array_equal(block.T, pcm[s0:s0 + bs]), "%s: frame at sample %d" % (case["name"], s0)


def test_subframe_records_match_a_parse_of_the_subframe_headers():
    """flacamd_decoder_subframe_info against a host-side parse of the same frames' subframe headers: type, wasted bits,
    order, precision, quantization level, coefficients, warm-up samples, partition order and the first partition's Rice
    parameter, read bit by bit from the stream as the format lays them out (stream_decoder.c:2450-2713)."""
    import flac_amd

    class Bits:
        def __init__(self, data, pos):
            self.d, self.p = data, pos
        def get(self, n):
            v = 0
            for _ in range(n):
                v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
                self.p += 1
            return v
        def signed(self, n):
            v = self.get(n)
            return v - (1 << n) if n and v >> (n - 1) else v

    checked = 0
    for case in CASES:
        if case["bps"] > 24 or checked >= 40:
            continue
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        ch = case["channels"]
        dec = flac_amd.Decoder(ch, case["bps"], blocksize_of(case), device=0, max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
        try:
            dec.enable_subframe_info()
            dec.decode_host(blob, [(0, len(blob), 0, pcm.shape[0])], pcm.shape[0], out_sample_bytes=4)
            rows = dec.frames()
            recs = dec.subframe_info(0, len(rows))
        finally:
            dec.close()
        # the first subframe of each of the first frames: its header follows the frame header directly
        for k, r in enumerate(rows[:6]):
            b = Bits(blob, (int(r.pos) + int(r.hdr_bytes)) * 8)
            rec = recs[k * ch]
            assert b.get(1) == 0
            t = b.get(6)
            wasted = 0
            if b.get(1):
                wasted = 1
                while not b.get(1):
                    wasted += 1
            sbps = case["bps"] - wasted + (1 if r.assignment == 2 else 0)   # (right-side: the first subframe is the side channel)
            assert rec.wasted_bits == wasted, case["name"]
            if t == 0:
                assert rec.type == 0 and rec.constant == b.signed(sbps)
            elif t == 1:
                assert rec.type == 1
            elif 8 <= t <= 12:
                order = t & 7
                assert rec.type == 2 and rec.order == order
                assert [rec.warmup[i] for i in range(order)] == [b.signed(sbps) for _ in range(order)]
                method = b.get(2)
                assert rec.rice2 == method and rec.partition_order == b.get(4)
                assert rec.params[0] == b.get(5 if method else 4)
            else:
                order = (t & 31) + 1
                assert t >= 32 and rec.type == 3 and rec.order == order
                assert [rec.warmup[i] for i in range(order)] == [b.signed(sbps) for _ in range(order)]
                prec = b.get(4) + 1
                assert rec.precision == prec and rec.shift == b.signed(5)
                assert [rec.qlp_coeff[i] for i in range(order)] == [b.signed(prec) for _ in range(order)]
                method = b.get(2)
                assert rec.rice2 == method and rec.partition_order == b.get(4)
                p0 = b.get(5 if method else 4)
                if p0 != (31 if method else 15):
                    assert rec.params[0] == p0
            checked += 1
    assert checked >= 40


@pytest.mark.parametrize("channels", [1, 3, 4, 5, 6, 7, 8])
def test_every_channel_count_and_output_width_comes_back(channels):
    """The pass that turns the decoded planes into interleaved samples takes four frames' values of a sample at a time for
    three to eight channels (whole vectors where the channel count and the output width allow, single values otherwise):
    streams of every channel count, 12 / 16 / 24 bits, into 16- and 32-bit output, with a short last block and block sizes that
    are not a multiple of anything, against the samples that went in."""
    import flac_amd
    for bps, bs, level in ((16, 4096, 5), (24, 1152, 8), (12, 999, 3)):
        nframes = 37
        pcm, frames = _synthetic_stream(channels, bps, 48000, bs, level, nframes, 100 * channels + bps)
        n = pcm.shape[0]
        blob = b"".join(frames)
        dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=len(blob) + 64, max_frames=nframes + 8)
        try:
            for osb in ((2, 4) if bps <= 16 else (4,)):
                out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=osb)
                assert nfr == nframes and wide == n and bad == 0
                assert np.array_equal(np.asarray(out, dtype=np.int32), pcm), (channels, bps, bs, osb)
        finally:
            dec.close()


@pytest.mark.parametrize("channels,bps,bs", [(1, 8, 16), (2, 16, 16), (2, 16, 192), (1, 16, 4096)])
def test_header_scan_with_many_and_with_few_headers_per_block(channels, bps, bs):
    """The scan for frame headers keeps the headers it parses, up to eight per 4 KiB of input, for the pass that puts them in
    order; a stretch of input with more (frames of sixteen samples are a dozen bytes each: hundreds per 4 KiB) is parsed again.
    Streams on both sides of that limit, and one whose stretches fall on either side (silence and noise in turn), come back whole."""
    import flac_amd
    rng = np.random.default_rng(bs + channels)
    nframes = 3000 if bs <= 192 else 64
    n = nframes * bs - bs // 2
    full = 1 << (bps - 1)
    pcm = rng.integers(-full // 4, full // 4, size=(n, channels)).astype(np.int32)
    quarter = n // 4
    pcm[quarter:2 * quarter] = 0          # constant subframes: the smallest frames there are
    pcm[3 * quarter:] = 7
    enc = flac_amd.Encoder(flac_amd.preset(5, channels, bps, 44100, bs), device=0, max_frames=nframes)
    try:
        frames, _ = enc.encode_host(pcm)
    finally:
        enc.close()
    assert len(frames) == nframes
    blob = b"".join(frames)
    dec = flac_amd.Decoder(channels, bps, bs, device=0, max_input_bytes=len(blob) + 64, max_frames=nframes + 64)
    try:
        out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, n)], n, out_sample_bytes=4)
        assert nfr == nframes and wide == n
        assert np.array_equal(out, pcm)
    finally:
        dec.close()
