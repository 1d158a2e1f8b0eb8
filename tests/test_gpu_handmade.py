"""GPU decoder on the hand-assembled streams: escape-coded Rice partitions (raw width 0 and > 0, both
coding methods), a first partition that holds only warm-up samples, reserved subframe types and
residual coding methods.  The answers are the reference decoder's (tests/golden/handmade.json)."""
import os

import numpy as np
import pytest

import handmade_util
import libflac_ctypes as L

pytestmark = pytest.mark.gpu

VECS = handmade_util.load()


@pytest.fixture(scope="module")
def lib():
    import flac_amd
    return L.load(flac_amd.LIB_PATH)


@pytest.fixture(params=["split", "lane"], autouse=True)
def decoder_form(request, monkeypatch):
    """Every case through both forms of the frame decoder: the split form (small batches of two or more channels take it
    by default) and a lane per frame."""
    monkeypatch.setenv("FLAC_AMD_DEC_MODE", request.param)
    return request.param


@pytest.mark.parametrize("v", VECS, ids=[v["name"] for v in VECS])
def test_batch_decoder(v):
    import flac_amd
    blob = v["frames_blob"]
    nframes = len([1 for i in range(len(blob) - 1) if blob[i] == 0xff and blob[i + 1] == 0xf8])  # upper bound
    total = (max(n for n, _b in v["blocks_np"]) // v["blocksize"] + 1) * v["blocksize"]
    dec = flac_amd.Decoder(v["channels"], v["bps"], v["blocksize"], device=0, max_input_bytes=len(blob) + 64, max_frames=nframes + 8)
    try:
        out, nfr, wide, bad = dec.decode_host(blob, [(0, len(blob), 0, total)], total)
    finally:
        dec.close()
    assert nfr == len(v["blocks_np"])
    assert np.array_equal(out, handmade_util.expected_pcm(v, total))
    assert (bad > 0) == bool(v["errors"])


@pytest.mark.parametrize("v", VECS, ids=[v["name"] for v in VECS])
def test_stream_decoder(lib, v):
    sd = L.StreamDecoder(lib, v["stream_bytes"], md5_check=False)
    try:
        assert sd.init_status == 0
        ok = bool(sd.call("process_until_end_of_stream"))
        state = sd.state()
        fin = bool(sd.call("finish"))
    finally:
        sd.close()
    assert [n for n, _b in sd.blocks] == [n for n, _b in v["blocks_np"]]
    for (n, got), (_n, want) in zip(sd.blocks, v["blocks_np"]):
        assert np.array_equal(np.ascontiguousarray(got).astype(np.int32).reshape(want.shape), want), n
    assert sd.errors == v["errors"]
    assert (ok, state, fin) == (v["process_ok"], v["state"], v["finish_ok"])
