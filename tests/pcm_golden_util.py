"""Loading of tests/golden/pcm_*.npz (files written and read by the reference `flac`; see oracle/gen_pcm_golden.py)."""
import json
import os
import struct

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# decode.c:350-374: the channel mask a WAVEFORMATEXTENSIBLE header gets when the stream carries none
DEFAULT_MASK = {1: 0x0001, 2: 0x0003, 3: 0x0007, 4: 0x0033, 5: 0x0607, 6: 0x060f}


def load_index():
    with open(os.path.join(GOLD, "pcm_cases.json")) as f:
        return json.load(f)


def load_case(case):
    z = np.load(os.path.join(GOLD, "pcm_" + case["name"] + ".npz"))
    return z["infile"].tobytes(), z["flac"].tobytes(), z["decoded"].tobytes()


def streaminfo(flac):
    """(min_bs, max_bs, rate, channels, bps, total_samples, audio_offset) of a .flac held in memory"""
    assert flac[:4] == b"fLaC"
    pos, info = 4, None
    while True:
        last, typ = flac[pos] >> 7, flac[pos] & 0x7f
        n = int.from_bytes(flac[pos + 1:pos + 4], "big")
        body = flac[pos + 4:pos + 4 + n]
        if typ == 0:
            min_bs, max_bs = struct.unpack(">HH", body[:4])
            x = int.from_bytes(body[10:18], "big")
            info = (min_bs, max_bs, x >> 44, ((x >> 41) & 7) + 1, ((x >> 36) & 31) + 1, x & ((1 << 36) - 1))
        pos += 4 + n
        if last:
            break
    return info + (pos,)


def samples_of(data, channels, container_bits, big_endian, unsigned, shift):
    """format_input (src/flac/encode.c:2333) in numpy: file bytes -> int32 (n, channels)"""
    nb = container_bits // 8
    raw = np.frombuffer(data, dtype=np.uint8)
    raw = raw[:len(raw) // (nb * channels) * nb * channels].reshape(-1, channels, nb).astype(np.int64)
    v = np.zeros(raw.shape[:2], dtype=np.int64)
    for b in range(nb):
        v |= raw[:, :, (nb - 1 - b) if big_endian else b] << (8 * b)
    if unsigned:
        v -= 1 << (container_bits - 1)
    else:
        v = (v ^ (1 << (container_bits - 1))) - (1 << (container_bits - 1))
    return (v >> shift).astype(np.int32)
