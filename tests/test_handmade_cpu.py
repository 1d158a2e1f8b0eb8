"""The hand-assembled streams (escape-coded Rice partitions, an empty first partition, reserved
codes): the CPU restatement must decode them as the reference's decoder did (tests/golden/handmade.json)."""
import hashlib

import numpy as np
import pytest

import handmade_util
import oracle_lib as O

VECS = handmade_util.load()


@pytest.mark.parametrize("v", VECS, ids=[v["name"] for v in VECS])
def test_oracle_decodes_like_the_reference(v):
    assert hashlib.sha256(v["stream_bytes"]).hexdigest() == v["stream_sha256"]
    nframes_total = v["frame_bytes"] and (max(n for n, _b in v["blocks_np"]) // v["blocksize"] + 1)
    if v["errors"]:
        # the restatement's frame walker has no resynchronisation: it must refuse the stream, not misread it
        n, got, _nf = O.decode_frames(v["frames_blob"], v["channels"], v["bps"], 44100, v["blocksize"] * 64)
        assert n < 0 and got is None
        return
    total = nframes_total * v["blocksize"]
    n, got, nf = O.decode_frames(v["frames_blob"], v["channels"], v["bps"], 44100, total)
    assert n == total and nf == nframes_total
    assert np.array_equal(got, handmade_util.expected_pcm(v, total))
