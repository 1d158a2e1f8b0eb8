#!/usr/bin/env python3
"""Test infrastructure: golden vectors for the PCM container stage, made by the reference command line.

For each case a small WAVE / RF64 / Wave64 / AIFF / AIFF-C file is written here (numpy, seeded), encoded by
oracle/_ref/flac_ref (the reference `flac`, compiled from /root/reference by oracle/Makefile) and decoded back by
it into the same container family.  Stored per case in tests/golden/pcm_<name>.npz: the input file, the .flac the
reference wrote and the file its decoder wrote.  Run from the repo root: python oracle/gen_pcm_golden.py
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FLAC = os.path.join(HERE, "_ref", "flac_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

W64_RIFF = bytes.fromhex("726966662E91CF11D6A528DB04C10000")
W64_WAVE = bytes.fromhex("77617665F3ACD311D18C00C04F8EDB8A")
W64_FMT = bytes.fromhex("666D7420F3ACD311D18C00C04F8EDB8A")
W64_DATA = bytes.fromhex("64617461F3ACD311D18C00C04F8EDB8A")
W64_JUNK = bytes.fromhex("6A756E6BF3ACD311D18C00C04F8EDB8A")
PCM_GUID = bytes.fromhex("0100000000001000800000aa00389b71")


def signal(n, ch, bits, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    x = 0.5 * np.sin(2 * np.pi * (0.01 + 0.003 * np.arange(ch)[None, :]) * t) + 0.05 * rng.standard_normal((n, ch))
    full = 1 << (bits - 1)
    return np.clip(np.round(x * 0.8 * full), -full, full - 1).astype(np.int64)


def sample_bytes(pcm, valid_bits, container_bits, big_endian, unsigned):
    """interleaved samples, left-justified in container_bits, as file bytes"""
    v = pcm.astype(np.int64) << (container_bits - valid_bits)
    if unsigned:
        v = v + (1 << (container_bits - 1))
    v = v & ((1 << container_bits) - 1)
    nb = container_bits // 8
    out = np.zeros(v.shape + (nb,), dtype=np.uint8)
    for b in range(nb):
        out[..., (nb - 1 - b) if big_endian else b] = (v >> (8 * b)) & 0xff
    return out.tobytes()


def fmt_chunk_body(ch, rate, container_bits, valid_bits, extensible, mask):
    nb = container_bits // 8
    body = struct.pack("<HHIIHH", 65534 if extensible else 1, ch, rate, rate * ch * nb, ch * nb, container_bits)
    if extensible:
        body += struct.pack("<HHI", 22, valid_bits, mask) + PCM_GUID
    return body


def make_wave(pcm, rate, valid_bits, container_bits, extensible=False, mask=0, junk=False, kind="wav"):
    ch = pcm.shape[1]
    data = sample_bytes(pcm, valid_bits, container_bits, False, container_bits <= 8)
    fmt = fmt_chunk_body(ch, rate, container_bits, valid_bits, extensible, mask)
    if kind == "w64":
        def chunk(guid, body):
            pad = (-len(body)) % 8
            return guid + struct.pack("<Q", 24 + len(body)) + body + b"\0" * pad
        chunks = (chunk(W64_JUNK, b"hello w64") if junk else b"") + chunk(W64_FMT, fmt) + chunk(W64_DATA, data)
        return W64_RIFF + struct.pack("<Q", 40 + len(chunks)) + W64_WAVE + chunks
    def chunk(cid, body, size=None):
        return cid + struct.pack("<I", len(body) if size is None else size) + body + b"\0" * (len(body) & 1)
    chunks = (chunk(b"LIST", b"odd") if junk else b"") + chunk(b"fmt ", fmt)
    if kind == "rf64":
        ds64 = b"ds64" + struct.pack("<IQQQI", 28, 0, len(data), pcm.shape[0], 0)
        body = ds64 + chunks + chunk(b"data", data, 0xffffffff)
        return b"RF64" + struct.pack("<I", 0xffffffff) + b"WAVE" + body
    body = chunks + chunk(b"data", data)
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body


def sane_extended(val):
    shift = 0
    while (val >> (31 - shift)) == 0:
        shift += 1
    return struct.pack(">HII", 63 - (shift + 32) + 0x3FFF, (val << shift) & 0xffffffff, 0)


def make_aiff(pcm, rate, bits, aifc=None, ssnd_offset=0, junk=False):
    ch = pcm.shape[1]
    container_bits = (bits + 7) // 8 * 8
    data = sample_bytes(pcm, bits, container_bits, aifc != b"sowt", False)
    comm = struct.pack(">HIH", ch, pcm.shape[0], bits) + sane_extended(rate)
    if aifc:
        comm += aifc + b"\0\0"  # compression type + empty pascal string, padded
    def chunk(cid, body):
        return cid + struct.pack(">I", len(body)) + body + b"\0" * (len(body) & 1)
    ssnd = struct.pack(">II", ssnd_offset, 0) + b"\xee" * ssnd_offset + data
    body = (chunk(b"ANNO", b"odd") if junk else b"") + chunk(b"COMM", comm) + chunk(b"SSND", ssnd)
    return b"FORM" + struct.pack(">I", 4 + len(body)) + (b"AIFC" if aifc else b"AIFF") + body


CASES = [
    # name, extension, builder kwargs, decode flags
    ("wav8_mono", "wav", lambda: make_wave(signal(1500, 1, 8, 1), 8000, 8, 8), []),
    ("wav16_stereo", "wav", lambda: make_wave(signal(3000, 2, 16, 2), 44100, 16, 16), []),
    ("wav16_junk", "wav", lambda: make_wave(signal(1201, 2, 16, 3), 48000, 16, 16, junk=True), []),
    ("wav24_stereo", "wav", lambda: make_wave(signal(2500, 2, 24, 4), 96000, 24, 24), []),
    ("wavext20_stereo", "wav", lambda: make_wave(signal(2000, 2, 20, 5), 48000, 20, 24, extensible=True, mask=3), []),
    ("wavext12_mono", "wav", lambda: make_wave(signal(1700, 1, 12, 6), 22050, 12, 16, extensible=True, mask=1), []),
    ("wavext16_6ch", "wav", lambda: make_wave(signal(1100, 6, 16, 7), 48000, 16, 16, extensible=True, mask=0x3f), []),
    ("rf64_16_stereo", "rf64", lambda: make_wave(signal(1800, 2, 16, 8), 44100, 16, 16, kind="rf64"), ["--force-rf64-format"]),
    ("w64_24_stereo", "w64", lambda: make_wave(signal(1300, 2, 24, 9), 88200, 24, 24, kind="w64", junk=True), ["--force-wave64-format"]),
    ("aiff8_mono", "aiff", lambda: make_aiff(signal(1500, 1, 8, 10), 11025, 8), ["--force-aiff-format"]),
    ("aiff16_stereo", "aiff", lambda: make_aiff(signal(3000, 2, 16, 11), 44100, 16, junk=True), ["--force-aiff-format"]),
    ("aiff24_stereo", "aiff", lambda: make_aiff(signal(2100, 2, 24, 12), 96000, 24, ssnd_offset=6), ["--force-aiff-format"]),
    ("aiff12_mono", "aiff", lambda: make_aiff(signal(1900, 1, 12, 13), 32000, 12), ["--force-aiff-format"]),
    ("aifc_sowt16_stereo", "aiff", lambda: make_aiff(signal(1600, 2, 16, 14), 44100, 16, aifc=b"sowt"), ["--force-aiff-format"]),
    ("aifc_none16_mono", "aiff", lambda: make_aiff(signal(1400, 1, 16, 15), 16000, 16, aifc=b"NONE"), ["--force-aiff-format"]),
]


def main():
    if not os.path.exists(FLAC):
        sys.exit("build oracle/_ref first (make -C oracle ref)")
    index = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, ext, build, dflags in CASES:
            infile = build()
            src = os.path.join(tmp, name + "." + ext)
            fl = os.path.join(tmp, name + ".flac")
            dec = os.path.join(tmp, name + ".dec." + ext)
            open(src, "wb").write(infile)
            extra = ["--channel-map=none"] if "6ch" in name else []
            subprocess.run([FLAC, "-s", "-f", "-8", "-b", "576", "--no-padding", "--no-seektable"] + extra + ["-o", fl, src], check=True)
            subprocess.run([FLAC, "-s", "-f", "-d"] + dflags + extra + ["-o", dec, fl], check=True)
            np.savez_compressed(os.path.join(GOLD, "pcm_" + name + ".npz"),
                                infile=np.frombuffer(infile, dtype=np.uint8),
                                flac=np.frombuffer(open(fl, "rb").read(), dtype=np.uint8),
                                decoded=np.frombuffer(open(dec, "rb").read(), dtype=np.uint8))
            index.append({"name": name, "ext": ext, "channel_map_none": bool(extra)})
            print(name, len(infile), os.path.getsize(fl), os.path.getsize(dec))
    with open(os.path.join(GOLD, "pcm_cases.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
