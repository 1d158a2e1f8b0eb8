"""CPU tests: bit-level known answers (CRC, UTF-8 numbers, raw fields, Rice blocks, windows, MD5).

The expected bytes were produced by the reference's own bitwriter / crc / window functions
(oracle/gen_golden.py, kats()); the UTF-8 byte patterns below restate the reference's bit-writer
unit test expectations (src/test_libFLAC/bitwriter.c:240-554).
"""
import ctypes as C
import hashlib

import numpy as np

import golden_util
import oracle_lib as O

K = golden_util.load_kats()


def test_crc():
    data = np.frombuffer(bytes.fromhex(K["crc_input"]), dtype=np.uint8)
    p = data.ctypes.data_as(C.c_void_p)
    for n, c8, c16 in zip(K["crc_lens"], K["crc8"], K["crc16"]):
        assert O.lib().fo_crc8(p, C.c_size_t(n)) == c8
        assert O.lib().fo_crc16(p, C.c_size_t(n)) == c16


def _utf8(v):
    buf = (C.c_uint8 * 8)()
    n = O.lib().fo_utf8_encode(C.c_uint64(v), buf)
    return bytes(buf[:n]).hex()


def test_utf8_against_reference():
    for v, hexbytes in K["utf8"].items():
        assert _utf8(int(v)) == hexbytes


def test_utf8_length_classes():
    assert _utf8(0x00) == "00"
    assert _utf8(0x7F) == "7f"
    assert _utf8(0x80) == "c280"
    assert _utf8(0x7FF) == "dfbf"
    assert _utf8(0x800) == "e0a080"
    assert _utf8(0xFFFF) == "efbfbf"
    assert _utf8(0x10000) == "f0908080"
    assert _utf8(0x1FFFFF) == "f7bfbfbf"
    assert _utf8(0x200000) == "f888808080"
    assert _utf8(0x3FFFFFF) == "fbbfbfbfbf"
    assert _utf8(0x4000000) == "fc8480808080"
    assert _utf8(0x7FFFFFFF) == "fdbfbfbfbfbf"
    assert _utf8(0xFFFFFFFFF) == "febfbfbfbfbfbf"


def _pack(pairs):
    vals = (C.c_uint64 * len(pairs))(*[v for v, _ in pairs])
    bits = (C.c_uint32 * len(pairs))(*[n for _, n in pairs])
    out = (C.c_uint8 * 65536)()
    n = O.lib().fo_bitpack(vals, bits, len(pairs), out, C.c_size_t(65536))
    return bytes(out[:n]).hex()


def test_raw_fields():
    assert _pack([tuple(x) for x in K["raw_fields"]["vals"]]) == K["raw_fields"]["bytes"]


def test_rice_blocks():
    for blk in K["rice"]:
        k = blk["k"]
        pairs = [(5, 3)]
        for v in blk["vals"]:
            u = ((v << 1) ^ (v >> 31)) & 0xFFFFFFFF
            msbs = u >> k
            while msbs > 0:
                t = min(msbs, 32)
                pairs.append((0, t))
                msbs -= t
            pairs.append((1, 1))
            if k:
                pairs.append((u & ((1 << k) - 1), k))
        assert _pack(pairs) == blk["bytes"], "rice k=%d" % k


def test_windows():
    shapes = {"tukey_0.5_4096": (0, 0.5, 4096), "tukey_0.25_1000": (0, 0.25, 1000), "hann_576": (1, 0, 576),
              "hamming_100": (3, 0, 100), "blackman_100": (4, 0, 100), "welch_33": (5, 0, 33),
              "triangle_33": (6, 0, 33), "bartlett_32": (7, 0, 32)}
    for name, (code, prm, L) in shapes.items():
        w = np.zeros(L, dtype=np.float32)
        O.lib().fo_window(C.c_uint32(code), C.c_float(prm), C.c_uint32(L), w.ctypes.data_as(C.c_void_p))
        assert hashlib.sha256(w.tobytes()).hexdigest() == K["windows_sha256"][name], name


def test_md5():
    for msg in (b"", b"a", b"abc", b"message digest", b"x" * 55, b"y" * 56, b"z" * 64, b"q" * 1000):
        d = (C.c_uint8 * 16)()
        buf = np.frombuffer(msg, dtype=np.uint8) if msg else np.zeros(1, dtype=np.uint8)
        O.lib().fo_md5(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(msg)), d)
        assert bytes(d) == hashlib.md5(msg).digest()
