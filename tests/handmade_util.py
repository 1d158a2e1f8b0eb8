"""Loading of tests/golden/handmade.json: hand-assembled streams (oracle/gen_handmade.py) with what the
reference's stream decoder made of them."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    with open(os.path.join(GOLD, "handmade.json")) as f:
        vecs = json.load(f)
    for v in vecs:
        v["stream_bytes"] = bytes.fromhex(v["stream"])
        v["frames_blob"] = v["stream_bytes"][len(v["stream_bytes"]) - v["frame_bytes"]:]
        v["blocks_np"] = [(n, np.array(b, dtype=np.int32).reshape(-1, v["channels"])) for n, b in v["blocks"]]
    return vecs


def expected_pcm(v, total_wide):
    """The stream's samples as the reference delivers them, zeros where it delivered nothing."""
    out = np.zeros((total_wide, v["channels"]), dtype=np.int32)
    for n, b in v["blocks_np"]:
        out[n:n + b.shape[0]] = b
    return out
