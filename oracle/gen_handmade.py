#!/usr/bin/env python3
"""Hand-assembled FLAC streams for decoder paths no encoder in this tree (or the reference's)
ever produces, decoded by the REAL reference decoder to make the golden answers.

Test infrastructure only.  Needs oracle/_ref/libFLAC_ref.so (built from /root/reference by
oracle/Makefile); writes tests/golden/handmade.json, which is what travels.

What the streams cover (the reference lines that read them):
  * escape-coded Rice partitions, raw width 0 and > 0, under both residual coding methods
    (4-bit parameters / escape 15, 5-bit parameters / escape 31), next to ordinary partitions
    (stream_decoder.c:2747-2767)
  * a first partition exactly as long as the predictor order: it holds no residual but its parameter
    (plain or escaped) is still in the stream (stream_decoder.c:2734 accepts partition_samples == order)
  * reserved subframe types and a reserved residual coding method: UNPARSEABLE_STREAM, while a set
    padding bit is LOST_SYNC (stream_decoder.c:2476-2503, 2575, 2654)

The bit packing below is written from the format description (doc/html/format.html in the
reference tree): big-endian bit order, CRC-8 poly 0x07 over the header, CRC-16 poly 0x8005 over
the whole frame.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


class Bits:
    def __init__(self):
        self.v = 0
        self.n = 0

    def put(self, nbits, value):
        if nbits:
            self.v = (self.v << nbits) | (value & ((1 << nbits) - 1))
            self.n += nbits

    def unary(self, zeros):
        self.put(zeros, 0)
        self.put(1, 1)

    def pad(self):
        if self.n % 8:
            self.put(8 - self.n % 8, 0)

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big")


def crc(data, poly, width):
    top = 1 << (width - 1)
    mask = (1 << width) - 1
    c = 0
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def utf8_number(n):
    if n < 0x80:
        return bytes([n])
    out = []
    length = 2
    while n >= (1 << (5 * length + 1)):
        length += 1
    for i in range(length - 1):
        out.append(0x80 | (n & 0x3f))
        n >>= 6
    lead = ((0xff << (8 - length)) & 0xff) | n
    return bytes([lead] + out[::-1])


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def wrap32(x):
    x &= 0xffffffff
    return x - (1 << 32) if x & 0x80000000 else x


def restore(warmup, residual, coefs, shift):
    """samples from warm-up + residuals, the decoder's recurrence"""
    d = list(warmup)
    for r in residual:
        pred = sum(q * d[-1 - j] for j, q in enumerate(coefs)) >> shift
        d.append(wrap32(r + pred))
    return d


def put_residual(b, residual, order, blocksize, po, method, parts):
    """parts[p] = ("rice", k) or ("escape", raw_bits); residual covers samples order..blocksize-1"""
    b.put(2, method)
    b.put(4, po)
    plen = 5 if method == 1 else 4
    esc = 31 if method == 1 else 15
    psize = blocksize >> po if po else blocksize - order
    pos = 0
    for p in range(1 << po):
        n = psize if (po == 0 or p > 0) else psize - order
        kind, val = parts[p]
        chunk = residual[pos:pos + n]
        pos += n
        if kind == "rice":
            b.put(plen, val)
            for r in chunk:
                u = (r << 1) ^ (r >> 31) if r >= 0 else ((-r) << 1) - 1
                b.unary(u >> val)
                b.put(val, u & ((1 << val) - 1))
        else:
            b.put(plen, esc)
            b.put(5, val)
            for r in chunk:
                assert val == 0 and r == 0 or -(1 << (val - 1)) <= r < (1 << (val - 1)), (r, val)
                b.put(val, r)
    assert pos == len(residual)


def subframe(b, spec, bps, blocksize):
    """Writes one subframe; returns its samples."""
    kind = spec["kind"]
    wasted = spec.get("wasted", 0)
    sbps = bps - wasted
    if kind == "raw_type":
        # a type byte as given (reserved codes, padding bit), then nothing that matters
        b.put(8, spec["byte"])
        b.put(64, 0)
        return [0] * blocksize
    order = spec.get("order", 0)
    tb = {"constant": 0x00, "verbatim": 0x02, "fixed": 0x10 | (order << 1), "lpc": 0x40 | ((order - 1) << 1)}[kind]
    b.put(8, tb | (1 if wasted else 0))
    if wasted:
        b.unary(wasted - 1)
    if kind == "constant":
        b.put(sbps, spec["value"])
        out = [spec["value"]] * blocksize
    elif kind == "verbatim":
        out = list(spec["samples"])
        for s in out:
            b.put(sbps, s)
    else:
        warm = list(spec["warmup"])
        assert len(warm) == order
        for s in warm:
            b.put(sbps, s)
        if kind == "lpc":
            coefs, shift, prec = spec["coefs"], spec["shift"], spec["precision"]
            b.put(4, prec - 1)
            b.put(5, shift)
            for q in coefs:
                b.put(prec, q)
        else:
            coefs, shift = FIXED[order], 0
        res = list(spec["residual"])
        assert len(res) == blocksize - order
        put_residual(b, res, order, blocksize, spec["po"], spec.get("method", 0), spec["parts"])
        out = restore(warm, res, coefs, shift)
    for s in out:
        assert -(1 << (sbps - 1)) <= s < (1 << (sbps - 1)), "sample %d does not fit %d bits" % (s, sbps)
    return [s << wasted for s in out]


BPS_CODE = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


def frame(number, blocksize, channels, bps, subframes, assignment=None):
    """One fixed-blocksize frame at 44.1 kHz; returns (bytes, samples[channel][i])."""
    b = Bits()
    b.put(14, 0x3ffe)
    b.put(2, 0)
    b.put(4, 6 if blocksize <= 256 else 7)
    b.put(4, 9)
    b.put(4, channels - 1 if assignment is None else assignment)
    b.put(3, BPS_CODE[bps])
    b.put(1, 0)
    for x in utf8_number(number):
        b.put(8, x)
    b.put(8 if blocksize <= 256 else 16, blocksize - 1)
    b.put(8, crc(b.bytes(), 0x07, 8))
    chans = [subframe(b, sp, bps, blocksize) for sp in subframes]
    b.pad()
    b.put(16, crc(b.bytes(), 0x8005, 16))
    return b.bytes(), chans


def stream(frames, blocksize, channels, bps, total):
    si = Bits()
    si.put(16, blocksize)
    si.put(16, blocksize)
    si.put(24, 0)
    si.put(24, 0)
    si.put(20, 44100)
    si.put(3, channels - 1)
    si.put(5, bps - 1)
    si.put(36, total)
    si.put(128, 0)  # no MD5: the decoder then does not check one
    return b"fLaC" + bytes([0x80, 0, 0, 34]) + si.bytes() + b"".join(frames)


def rnd_residual(rng, n, scale):
    return [int(x) for x in rng.integers(-scale, scale + 1, size=n)]


def build_all():
    rng = np.random.default_rng(20240607)
    out = []

    def add(name, blocksize, channels, bps, frame_specs):
        frames = [frame(i, blocksize, channels, bps, sf)[0] for i, sf in enumerate(frame_specs)]
        out.append((name, blocksize, channels, bps, stream(frames, blocksize, channels, bps, blocksize * len(frames)), sum(len(f) for f in frames)))

    # --- escape-coded partitions, 4-bit parameters (method 0), fixed predictor, 16-bit mono ---
    bs = 64
    res = rnd_residual(rng, 14, 5) + [0] * 16 + rnd_residual(rng, 16, 60) + rnd_residual(rng, 16, 20)
    f0 = [dict(kind="fixed", order=2, warmup=[100, 104], residual=res, po=2, parts=[("rice", 3), ("escape", 0), ("escape", 12), ("rice", 5)])]
    f1 = [dict(kind="fixed", order=1, warmup=[-7], residual=rnd_residual(rng, 63, 1000), po=0, parts=[("escape", 15)])]
    res = [0] * 60
    f2 = [dict(kind="fixed", order=4, warmup=[5, 5, 5, 5], residual=res, po=1, parts=[("escape", 0), ("escape", 0)])]
    add("escape_rice_fixed16", bs, 1, 16, [f0, f1, f2])

    # --- escape-coded partitions, 5-bit parameters (method 1), LPC, 24-bit stereo, wasted bits on one channel ---
    bs = 128
    coefs = [100, -50, 6]
    resl = rnd_residual(rng, 29, 40) + rnd_residual(rng, 32, 200000) + [0] * 32 + rnd_residual(rng, 32, 70000)
    left = dict(kind="lpc", order=3, precision=8, shift=6, coefs=coefs, warmup=[1000, 1100, 1180], residual=resl, po=2, method=1,
                parts=[("rice", 6), ("escape", 19), ("escape", 0), ("rice", 16)])
    resr = rnd_residual(rng, 62, 3) + rnd_residual(rng, 64, 50)
    right = dict(kind="fixed", order=2, wasted=2, warmup=[3, 1], residual=resr, po=1, method=1, parts=[("rice", 1), ("escape", 20)])
    g0 = [left, right]
    resl2 = rnd_residual(rng, 96, 100)
    left2 = dict(kind="lpc", order=32, precision=12, shift=10, coefs=[900, -300, 120, -60] + [3, -2] * 14, warmup=[int(500 * np.sin(i / 3.0)) for i in range(32)],
                 residual=resl2, po=0, method=1, parts=[("escape", 8)])
    right2 = dict(kind="verbatim", samples=rnd_residual(rng, bs, 1 << 22))
    g1 = [left2, right2]
    add("escape_rice2_lpc24", bs, 2, 24, [g0, g1])

    # --- first partition as long as the order: its parameter is there, its residuals are not ---
    bs = 16
    h0 = [dict(kind="fixed", order=4, warmup=[10, 12, 15, 19], residual=rnd_residual(rng, 12, 6), po=2,
               parts=[("rice", 2), ("rice", 3), ("rice", 1), ("rice", 4)])]
    h1 = [dict(kind="fixed", order=4, warmup=[-3, -3, -2, 0], residual=rnd_residual(rng, 12, 10), po=2,
               parts=[("escape", 7), ("escape", 8), ("rice", 6), ("rice", 7)])]
    h2 = [dict(kind="lpc", order=2, precision=7, shift=5, coefs=[58, -27], warmup=[40, 44], residual=rnd_residual(rng, 14, 9), po=3,
               parts=[("rice", 0)] + [("rice", 3)] * 7)]
    h3 = [dict(kind="fixed", order=2, warmup=[7, 7], residual=rnd_residual(rng, 14, 9), po=3, method=1,
               parts=[("escape", 0)] + [("rice", 4), ("escape", 5)] * 3 + [("rice", 2)])]
    add("empty_first_partition", bs, 1, 16, [h0, h1, h2, h3])

    # --- reserved codes inside otherwise good frames: frames 1, 3 and 5 are hit ---
    bs = 32
    good = lambda: [dict(kind="fixed", order=1, warmup=[int(rng.integers(-50, 50))], residual=rnd_residual(rng, 31, 12), po=0, parts=[("rice", 3)])]
    bad_type_low = [dict(kind="raw_type", byte=0x0c)]     # type 000110x: reserved
    bad_type_mid = [dict(kind="raw_type", byte=0x3e)]     # type 011111x: reserved
    bad_pad = [dict(kind="raw_type", byte=0x82)]          # padding bit set
    add("reserved_subframe_types", bs, 1, 16, [good(), bad_type_low, good(), bad_type_mid, good(), bad_pad, good()])
    bad_method = [dict(kind="fixed", order=1, warmup=[9], residual=rnd_residual(rng, 31, 12), po=0, method=2, parts=[("rice", 3)])]
    add("reserved_residual_method", bs, 1, 16, [good(), bad_method, good()])
    return out


def main():
    import libflac_ctypes as L
    ref = L.load(os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so"))
    vectors = []
    for name, bs, ch, bps, data, frame_bytes in build_all():
        sd = L.StreamDecoder(ref, data, md5_check=False)
        assert sd.init_status == 0
        ok = bool(sd.call("process_until_end_of_stream"))
        state = sd.state()
        fin = bool(sd.call("finish"))
        sd.close()
        blocks = [[int(n), np.ascontiguousarray(b).astype(np.int32).tolist()] for n, b in sd.blocks]
        vectors.append(dict(name=name, blocksize=bs, channels=ch, bps=bps, stream=data.hex(), stream_sha256=hashlib.sha256(data).hexdigest(),
                            frame_bytes=frame_bytes, process_ok=ok, state=state, finish_ok=fin, errors=sd.errors, blocks=blocks))
        print("%-26s %5d bytes ok=%d state=%d fin=%d errors=%s blocks=%s" % (name, len(data), ok, state, fin, sd.errors, [b[0] for b in blocks]))
    with open(os.path.join(GOLD, "handmade.json"), "w") as f:
        json.dump(vectors, f)
    print("wrote", os.path.join(GOLD, "handmade.json"))


if __name__ == "__main__":
    main()
