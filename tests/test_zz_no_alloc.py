"""GPU test of flacamd_encode_frames_static, the batch encode that allocates no device memory.

It is called through the C ABI on PCM resident in device memory and must produce exactly what
flacamd_encode_frames (the handle-based entry point it is a variant of) produces for the same
input: frame bytes, per-frame sizes, offsets, and a clean status word.  Two calls in a row on the
same GPU, the second with a smaller input and other settings, so that whatever the first call left
in the shared static buffer would show; then a batch beyond the buffer's size must be refused.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def encode_both_ways(level, channels, bps, rate, wide, seed, first_frame=0, **over):
    import torch

    import flac_amd
    from flac_amd.synth import synth_pcm

    lib = flac_amd.lib()
    dev = torch.device("cuda", 0)
    pcm = synth_pcm(wide_samples=wide, sample_rate=rate, channels=channels, bits_per_sample=bps, seed=seed)
    sample_bytes = 2 if bps <= 16 else 4
    host = pcm.astype(np.int16) if sample_bytes == 2 else pcm.astype(np.int32)
    pcm_dev = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
    p = flac_amd.preset(level, channels, bps, rate, **over)
    nframes = -(-wide // p.blocksize)

    enc = flac_amd.Encoder(p, device=0, max_frames=nframes)
    cap = enc.max_output_bytes(nframes)
    want = {"out": torch.zeros(cap, dtype=torch.uint8, device=dev), "sizes": torch.zeros(nframes, dtype=torch.int32, device=dev),
            "offs": torch.zeros(nframes + 1, dtype=torch.int64, device=dev)}
    n = enc.encode_device(pcm_dev.data_ptr(), sample_bytes, wide, want["out"].data_ptr(), cap, want["sizes"].data_ptr(),
                          want["offs"].data_ptr(), first_frame=first_frame, is_last=True)
    torch.cuda.synchronize()
    enc.status()
    enc.close()
    assert n == nframes

    got = {"out": torch.zeros(cap, dtype=torch.uint8, device=dev), "sizes": torch.zeros(nframes, dtype=torch.int32, device=dev),
           "offs": torch.zeros(nframes + 1, dtype=torch.int64, device=dev)}
    status = torch.full((1,), 0xdead, dtype=torch.int32, device=dev)
    nf = C.c_uint32(0)
    rc = lib.flacamd_encode_frames_static(
        C.byref(p), C.c_int(0), C.c_void_p(pcm_dev.data_ptr()), C.c_uint32(sample_bytes), C.c_uint64(wide), C.c_uint32(first_frame),
        C.c_int(1), None, C.c_void_p(got["out"].data_ptr()), C.c_size_t(cap), C.c_void_p(got["sizes"].data_ptr()),
        C.c_void_p(got["offs"].data_ptr()), C.c_void_p(status.data_ptr()), C.byref(nf), C.c_void_p(0))
    assert rc == 0, lib.flacamd_last_error().decode()
    torch.cuda.synchronize()
    assert nf.value == nframes
    assert int(status.item()) == 0
    assert torch.equal(got["sizes"], want["sizes"])
    assert torch.equal(got["offs"], want["offs"])
    total = int(want["offs"][nframes].item())
    assert total > 0 and torch.equal(got["out"][:total], want["out"][:total])
    return total


def test_static_entry_point_matches_the_handle_based_one_twice_in_a_row():
    # a -8 stereo batch of 40 blocks and a tail, then a smaller 24-bit three-channel one at -5 with another blocksize
    # and apodization, numbered from a later frame: the second call finds the first one's leftovers in the buffer
    a = encode_both_ways(8, 2, 16, 44100, 4096 * 40 + 777, seed=11)
    b = encode_both_ways(5, 3, 24, 96000, 1152 * 7 + 100, seed=12, first_frame=1000, blocksize=1152)
    c = encode_both_ways(8, 2, 16, 44100, 4096 * 3, seed=13)
    assert a > b > 0 and c > 0


def test_window_tables_survive_a_call_without_windows():
    # -8, then -0 at blocksize 1152 (no LPC, so no window table, and smaller buffers), then -8 again: the window table
    # kept from the first call must not have been overwritten by the second call's buffers
    a = encode_both_ways(8, 2, 16, 44100, 4096 * 20 + 5, seed=21)
    b = encode_both_ways(0, 2, 16, 44100, 1152 * 300, seed=22)
    c = encode_both_ways(8, 2, 16, 44100, 4096 * 20 + 5, seed=21)
    assert a == c and b > 0


def test_static_entry_point_refuses_what_does_not_fit():
    import torch

    import flac_amd
    lib = flac_amd.lib()
    p = flac_amd.preset(8, 2, 16, 44100)
    wide = 4096 * 4000  # needs about 100 MB of intermediate buffers
    dummy = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    nf = C.c_uint32(0)
    rc = lib.flacamd_encode_frames_static(
        C.byref(p), C.c_int(0), C.c_void_p(dummy.data_ptr()), C.c_uint32(2), C.c_uint64(wide), C.c_uint32(0), C.c_int(1), None,
        C.c_void_p(dummy.data_ptr()), C.c_size_t(16), None, None, None, C.byref(nf), C.c_void_p(0))
    assert rc == -2, rc  # FLACAMD_E_UNSUPPORTED, before any kernel is launched
    assert b"scratch" in lib.flacamd_last_error()
