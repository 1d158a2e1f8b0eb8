"""GPU tests at the sizes of the benchmark configurations that are not the bench line itself:
`flac -5` on ten minutes of 16-bit stereo, `-8 -e -p` on 24-bit/96 kHz stereo, eight channels of
24-bit/192 kHz, and one GPU's share of the ten-thousand-stream batch decode - each on the signal the
benchmark uses for it (flac_amd.workloads: synth_pcm, seed 1234 + 1000 x stream).

At these sizes the CPU restatement cannot check every byte in seconds, so each case is held to
properties that do not depend on size: the decoder gives back exactly the samples that went in,
every frame is accounted for, the frame sizes add up, and a leading stretch of the stream is
byte-identical to what the oracle (tests/oracle_lib.py) produces for the same samples.  The
ten-minute `-5` case is short enough on the CPU to be compared whole against the reference
library when that has been built (oracle/_ref); tests/test_gpu_reference_accepts.py does the same
for the other configurations.  `synth` below is a second, cheaper signal (tones, a sweep and
uniform noise) that the decoder probes and a few tests with a history of their own keep using."""
import os

import numpy as np
import pytest

import golden_util
from test_gpu_decode_golden import blocksize_of

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")


def synth(seed, wide, channels, bps, rate):
    """Tones, a slow sweep and noise, different in every channel: compressible like music, and
    cheap to make (ten seconds are synthesised, then repeated with fresh low-level noise)."""
    rng = np.random.default_rng(seed)
    base_n = min(wide, rate * 10)
    t = np.arange(base_n, dtype=np.float64) / rate
    full = float(1 << (bps - 1))
    chans = []
    for c in range(channels):
        x = 0.30 * np.sin(2 * np.pi * (220.0 * (c + 1)) * t) + 0.20 * np.sin(2 * np.pi * (1330.0 + 17 * c) * t + c)
        x += 0.15 * np.sin(2 * np.pi * (50.0 + 400.0 * t / t[-1]) * t)
        chans.append(x)
    base = (np.stack(chans, axis=1) * full * 0.8).astype(np.int64)
    reps = (wide + base_n - 1) // base_n
    out = np.empty((reps * base_n, channels), dtype=np.int32)
    noise_amp = max(2, 1 << max(1, bps - 12))
    for k in range(reps):
        seg = base + rng.integers(-noise_amp, noise_amp + 1, size=base.shape)
        out[k * base_n:(k + 1) * base_n] = np.clip(seg, -(1 << (bps - 1)), (1 << (bps - 1)) - 1)
    return out[:wide]


def gpu_encode(params, pcm, sample_bytes):
    """Whole stream in one batch, PCM resident at the given container width.  Returns the device
    tensors (frames back to back, sizes, offsets) and the frame count."""
    import flac_amd
    import torch
    wide, ch = pcm.shape
    host = pcm.astype(np.int16) if sample_bytes == 2 else pcm.astype(np.int32)
    dev = torch.from_numpy(host).to("cuda:0")
    nframes = -(-wide // params.blocksize)
    enc = flac_amd.Encoder(params, device=0, max_frames=nframes)
    try:
        cap = enc.max_output_bytes(nframes)
        out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
        sizes = torch.zeros(nframes, dtype=torch.int32, device="cuda:0")
        offs = torch.zeros(nframes + 1, dtype=torch.int64, device="cuda:0")
        nf = enc.encode_device(dev.data_ptr(), sample_bytes, wide, out.data_ptr(), cap, sizes.data_ptr(), offs.data_ptr())
        torch.cuda.synchronize()
        enc.status()
    finally:
        enc.close()
    assert nf == nframes
    return dev, out, sizes, offs, nframes


def check_round_trip(params, pcm, sample_bytes, oracle_frames=0):
    import flac_amd
    import oracle_lib as O
    import torch
    wide, ch = pcm.shape
    dev, out, sizes, offs, nframes = gpu_encode(params, pcm, sample_bytes)
    total = int(offs[nframes].item())
    szs = sizes.cpu().numpy().astype(np.int64)
    assert szs.min() > 0 and int(szs.sum()) == total
    assert np.array_equal(offs.cpu().numpy()[1:], np.cumsum(szs))
    if oracle_frames:
        n = min(oracle_frames, nframes)
        op = O.preset(*params_key(params))
        for k in ("do_mid_side_stereo", "loose_mid_side_stereo", "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search",
                  "do_exhaustive_model_search", "min_residual_partition_order", "max_residual_partition_order"):
            setattr(op, k, getattr(params, k))
        want, _ = O.encode_frames(op, pcm[:n * params.blocksize])
        head = out[:int(szs[:n].sum())].cpu().numpy().tobytes()
        assert head == b"".join(want[:n]), "the first %d frames differ from the oracle's" % n
    dec = flac_amd.Decoder(ch, params.bits_per_sample, params.blocksize, device=0, max_input_bytes=total + 64, max_frames=nframes + 16)
    try:
        back = torch.zeros((wide, ch), dtype=torch.int16 if sample_bytes == 2 else torch.int32, device="cuda:0")
        dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide, 0)], back.data_ptr(), sample_bytes, wide)
        torch.cuda.synchronize()
        frames, nwide, bad = dec.result()
    finally:
        dec.close()
    assert (frames, nwide, bad) == (nframes, wide, 0)
    assert torch.equal(back, dev)
    return out[:total], szs


def params_key(p):
    # level is not kept in the resolved settings; 8 resolves the same fields the overrides then replace
    return (8, p.channels, p.bits_per_sample, p.sample_rate, p.blocksize)


def test_level5_ten_minutes_against_the_reference_library():
    """configs[0]: `flac -5`, 16-bit/44.1 kHz stereo, ten minutes: every frame byte equals the
    reference library's, and the stream decodes back to the input."""
    import flac_amd
    from flac_amd import workloads as W
    pcm = W.pcm("C1")
    p = W.params("C1", flac_amd)
    frames_dev, szs = check_round_trip(p, pcm, 2, oracle_frames=40)
    if not os.path.exists(REF_LIB):
        pytest.skip("reference library not built: compared with the oracle on the leading frames only")
    import libflac_ctypes as L
    ref = L.load(REF_LIB)
    _meta, ref_frames, ok, _writes = L.encode(ref, pcm, 16, 44100, level=5)
    assert ok and len(ref_frames) == len(szs)
    assert b"".join(ref_frames) == frames_dev.cpu().numpy().tobytes()


def test_exhaustive_precision_search_24bit_96k():
    """configs[2]: `flac -8 -e -p` on 24-bit/96 kHz stereo, ten minutes (the last block is short)."""
    import flac_amd
    from flac_amd import workloads as W
    pcm = W.pcm("C3")
    wide = pcm.shape[0]
    p = W.params("C3", flac_amd)
    assert wide == 96000 * 600 and wide % p.blocksize != 0 and p.do_qlp_coeff_prec_search and p.do_exhaustive_model_search
    check_round_trip(p, pcm, 4, oracle_frames=6)


def test_eight_channels_24bit_192k():
    """configs[3]: eight independent 24-bit/192 kHz channels, one minute, `flac -8`."""
    import flac_amd
    from flac_amd import workloads as W
    pcm = W.pcm("C4")
    p = W.params("C4", flac_amd)
    assert pcm.shape == (192000 * 60, 8)
    check_round_trip(p, pcm, 4, oracle_frames=6)


def test_batch_decode_one_gpus_share_of_ten_thousand_streams():
    """configs[4]: 1250 five-second `-8` streams (25 distinct ones, 50 copies each, in shuffled
    order) decoded in one call; every stream must come back as the PCM it was made from."""
    import flac_amd
    import torch
    from flac_amd import workloads as W
    rate, seconds, distinct, copies = 44100, 5, 25, 50
    wide = rate * seconds
    p = W.params("C5", flac_amd)
    sources, encoded = [], []
    for i in range(distinct):
        pcm = W.pcm("C5", stream_id=i)
        assert pcm.shape == (wide, 2)
        dev, out, sizes, offs, nframes = gpu_encode(p, pcm, 2)
        total = int(offs[nframes].item())
        sources.append(dev)
        encoded.append(out[:total].clone())
    order = np.random.default_rng(1).permutation(distinct * copies) % distinct
    # streams start on 4-byte boundaries, as separately stored files would after a copy
    pieces, descs, pos = [], [], 0
    for k, which in enumerate(order):
        n = int(encoded[which].numel())
        descs.append((pos, n, k * wide, wide, 0))
        pieces.append(encoded[which])
        pad = (-n) % 4
        if pad:
            pieces.append(torch.zeros(pad, dtype=torch.uint8, device="cuda:0"))
        pos += n + pad
    flac = torch.cat(pieces)
    nstreams = len(order)
    frames_per_stream = -(-wide // p.blocksize)
    dec = flac_amd.Decoder(2, 16, p.blocksize, device=0, max_input_bytes=int(flac.numel()) + 64, max_frames=nstreams * frames_per_stream + 64)
    try:
        out = torch.zeros((nstreams * wide, 2), dtype=torch.int16, device="cuda:0")
        dec.decode_device(flac.data_ptr(), int(flac.numel()), descs, out.data_ptr(), 2, nstreams * wide)
        torch.cuda.synchronize()
        frames, nwide, bad = dec.result()
    finally:
        dec.close()
    assert (frames, nwide, bad) == (nstreams * frames_per_stream, nstreams * wide, 0)
    got = out.view(nstreams, wide, 2)
    for which in range(distinct):
        idx = torch.from_numpy(np.nonzero(order == which)[0]).to("cuda:0")
        assert bool((got[idx] == sources[which].unsqueeze(0)).all()), "copies of stream %d" % which


@pytest.mark.parametrize("mode,lw", [("split", 4), ("split", 8), ("split", 16), ("lane", 4), ("lane", 8), ("lane", 16), ("lane", 32)])
def test_decoder_frames_per_wave(mode, lw, monkeypatch):
    """The decoder carries 4 to 32 frames per wave (32 only with a lane per frame, for the largest batches) depending
    on the batch; every choice, in either form, must decode every golden stream exactly."""
    import flac_amd
    monkeypatch.setenv("FLAC_AMD_DEC_MODE", mode)
    monkeypatch.setenv("FLAC_AMD_DEC_LW", str(lw))
    for case in golden_util.load_index():
        pcm, _meta, frames = golden_util.load_case(case)
        blob = b"".join(frames)
        dec = flac_amd.Decoder(case["channels"], case["bps"], blocksize_of(case), device=0,
                               max_input_bytes=len(blob) + 64, max_frames=len(frames) + 8)
        try:
            got, nfr, nwide, bad = dec.decode_host(blob, [(0, len(blob), 0, pcm.shape[0])], pcm.shape[0], out_sample_bytes=4)
        finally:
            dec.close()
        assert (nfr, nwide, bad) == (len(frames), pcm.shape[0], 0), case["name"]
        assert np.array_equal(got, pcm), case["name"]


def test_sync_pattern_inside_a_good_frame_is_not_a_bad_frame():
    """One minute of eight 24-bit channels whose encoded bytes happen to hold, inside a frame, a sync pattern with a valid
    header CRC-8: a reader that walks the stream never looks there (frame_sync_, stream_decoder.c:1941), so the batch decoder
    must give back every sample and report no bad frame, although the pattern is in its table of frame headers."""
    import ctypes as C
    import flac_amd
    import torch
    wide = 192000 * 60
    pcm = synth(99, wide, 8, 24, 192000)
    p = flac_amd.preset(8, 8, 24, 192000)
    dev, out, sizes, offs, nframes = gpu_encode(p, pcm, 4)
    total = int(offs[nframes].item())
    dec = flac_amd.Decoder(8, 24, p.blocksize, device=0, max_input_bytes=total + 64, max_frames=nframes + 64)
    try:
        back = torch.zeros_like(dev)
        dec.decode_device(out.data_ptr(), total, [(0, total, 0, wide)], back.data_ptr(), 4, wide)
        torch.cuda.synchronize()
        assert dec.result() == (nframes, wide, 0)
        assert torch.equal(back, dev)
        rows = (C.c_uint64 * (6 * (nframes + 64)))()
        count, cut = C.c_uint32(0), C.c_int(0)
        assert flac_amd.lib().flacamd_decoder_frames(dec._h, rows, nframes + 64, C.byref(count), C.byref(cut)) == 0
        assert count.value > nframes and cut.value == 0   # the pattern is listed, with a status that is not "good"
    finally:
        dec.close()
