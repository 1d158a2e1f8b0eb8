"""The reference's `flac` tool and libFLAC++ running on top of libflac_amd.so on the GPU.

oracle/_ref/flac_on_amd is the reference's command-line tool (its sources, its headers) linked
against this library; oracle/_ref/flac_ref is the same tool on the reference's own libFLAC.  Both
get the same WAVE files and options; the .flac files and the decoded files must be identical byte
for byte, `--verify` and `-t` must pass, `--skip/--until` (decoder seeking) must give the same
samples."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
ON_AMD = os.path.join(REFDIR, "flac_on_amd")
REF = os.path.join(REFDIR, "flac_ref")
PROBE = os.path.join(REFDIR, "flacpp_probe")


def wav_bytes(pcm, bps, rate):
    n, ch = pcm.shape
    width = bps // 8
    if width == 2:
        body = pcm.astype("<i2").tobytes()
    else:
        raw = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        body = raw.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * width, ch * width, bps)
    return hdr + b"data" + struct.pack("<I", len(body)) + body


def run(tool, *args):
    return subprocess.run([tool, "--totally-silent"] + list(args), capture_output=True, text=True, timeout=240)


@pytest.fixture(scope="module")
def tools():
    for t in (ON_AMD, REF):
        if not os.path.exists(t):
            pytest.skip("%s has not been built" % t)
    return ON_AMD, REF


@pytest.mark.parametrize("bps,channels,rate,seconds,opts", [
    (16, 2, 44100, 20, ["-8"]),
    (16, 2, 44100, 7, ["-5"]),
    (16, 2, 44100, 7, ["-1"]),
    (24, 2, 96000, 5, ["-8", "-e", "-p"]),
    (16, 1, 22050, 9, ["-0", "-b", "1152"]),
    (24, 6, 48000, 4, ["-8", "--channel-map=none"]),
], ids=["l8", "l5", "l1_loose", "l8ep_24", "l0_mono", "l8_6ch_24"])
def test_encode_decode_test_match_the_reference_tool(tools, tmp_path, bps, channels, rate, seconds, opts):
    from flac_amd.synth import synth_pcm
    on_amd, ref = tools
    pcm = synth_pcm(seconds=seconds, sample_rate=rate, channels=channels, bits_per_sample=bps, seed=31)
    wav = tmp_path / "in.wav"
    wav.write_bytes(wav_bytes(pcm, bps, rate))
    a, b = tmp_path / "a.flac", tmp_path / "b.flac"
    r = run(on_amd, "--verify", "-f", "-o", str(a), *opts, str(wav))
    assert r.returncode == 0, r.stderr
    r = run(ref, "-f", "-o", str(b), *opts, str(wav))
    assert r.returncode == 0, r.stderr
    assert a.read_bytes() == b.read_bytes(), "the .flac file differs from the reference tool's"
    # test mode and decode, each tool on its own file
    assert run(on_amd, "-t", str(a)).returncode == 0
    aw, bw = tmp_path / "a.wav", tmp_path / "b.wav"
    assert run(on_amd, "-d", "-f", "-o", str(aw), str(a)).returncode == 0
    assert run(ref, "-d", "-f", "-o", str(bw), str(b)).returncode == 0
    assert aw.read_bytes() == bw.read_bytes()
    if channels <= 2:
        assert aw.read_bytes() == wav.read_bytes()   # (more channels come back under a WAVE_FORMAT_EXTENSIBLE header)
    # decoder seeking: a stretch from the middle
    s0, s1 = rate // 3 + 17, rate * 2 + 5
    assert run(on_amd, "-d", "-f", "--skip=%d" % s0, "--until=%d" % s1, "-o", str(aw), str(a)).returncode == 0
    assert run(ref, "-d", "-f", "--skip=%d" % s0, "--until=%d" % s1, "-o", str(bw), str(b)).returncode == 0
    assert aw.read_bytes() == bw.read_bytes()
    assert channels > 2 or len(aw.read_bytes()) == 44 + (s1 - s0) * channels * (bps // 8)


def _analysis_material(kind):
    """Samples that make the encoder use every subframe type, wasted bits, escaped and many partitions."""
    from flac_amd.synth import synth_pcm
    rng = np.random.default_rng(9)
    if kind == "music16":
        return synth_pcm(seconds=6, sample_rate=44100, channels=2, bits_per_sample=16, seed=3), 16, 44100
    if kind == "music24_wasted":
        pcm = synth_pcm(seconds=3, sample_rate=96000, channels=2, bits_per_sample=24, seed=4)
        return (pcm >> 5) << 5, 24, 96000   # five wasted bits in every subframe
    if kind == "mixed16":
        n = 44100 * 3
        pcm = synth_pcm(wide_samples=n, sample_rate=44100, channels=2, bits_per_sample=16, seed=6)
        pcm[20000:40000] = 0                                   # silence: constant subframes
        pcm[50000:70000, 0] = 1234                             # a constant that is not zero, one channel
        pcm[80000:100000] = rng.integers(-32768, 32768, size=(20000, 2))   # full-scale noise: verbatim subframes
        return pcm, 16, 44100
    raise ValueError(kind)


@pytest.mark.parametrize("kind,opts", [
    ("music16", ["-8"]),
    ("music16", ["-2"]),                       # fixed predictors only
    ("music16", ["-8", "-r", "8,8"]),          # 256 partitions per subframe
    ("music16", ["-5", "-b", "1152"]),
    ("music24_wasted", ["-8"]),
    ("mixed16", ["-8"]),
], ids=["l8", "l2_fixed", "l8_po8", "l5_bs1152", "l8_24_wasted", "l8_mixed"])
def test_analysis_output_matches_the_reference_tool(tools, tmp_path, kind, opts):
    """`flac -a` prints what the write callback's FLAC__Frame.subframes[] say (src/flac/analyze.c:63-144): subframe
    type, wasted bits, order, warm-up, coefficient precision, quantization level, coefficients, Rice method, partition
    order and every partition's parameter, besides each frame's offset and size.  Same text from both tools."""
    on_amd, ref = tools
    pcm, bps, rate = _analysis_material(kind)
    wav = tmp_path / "in.wav"
    wav.write_bytes(wav_bytes(pcm, bps, rate))
    f = tmp_path / "in.flac"
    assert run(ref, "-f", "-o", str(f), *opts, str(wav)).returncode == 0
    a, b = tmp_path / "a.ana", tmp_path / "b.ana"
    r = run(on_amd, "-a", "-f", "-o", str(a), str(f))
    assert r.returncode == 0, r.stderr
    assert run(ref, "-a", "-f", "-o", str(b), str(f)).returncode == 0
    ta, tb = a.read_text(), b.read_text()
    assert "type=LPC" in tb or "type=FIXED" in tb
    if kind == "mixed16":
        assert "type=CONSTANT" in tb and "type=VERBATIM" in tb
    if kind == "music24_wasted":
        assert "wasted_bits=5" in tb
    assert ta == tb, "first difference at line %d" % next(i for i, (x, y) in enumerate(zip(ta.splitlines() + [""], tb.splitlines() + [""])) if x != y)


def test_damaged_file_fails_test_mode(tools, tmp_path):
    from flac_amd.synth import synth_pcm
    on_amd, _ref = tools
    pcm = synth_pcm(seconds=3, sample_rate=44100, channels=2, bits_per_sample=16, seed=5)
    wav = tmp_path / "in.wav"
    wav.write_bytes(wav_bytes(pcm, 16, 44100))
    a = tmp_path / "a.flac"
    assert run(on_amd, "-f", "-5", "-o", str(a), str(wav)).returncode == 0
    data = bytearray(a.read_bytes())
    data[len(data) // 2] ^= 0x40
    a.write_bytes(bytes(data))
    assert run(on_amd, "-t", str(a)).returncode != 0


def test_libflacpp_on_top_of_the_library():
    if not os.path.exists(PROBE):
        pytest.skip("oracle/_ref/flacpp_probe has not been built")
    r = subprocess.run([PROBE, "50000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok ")
