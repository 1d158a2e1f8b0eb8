"""A short run of the mutation fuzzer over the host-side file parsers (metadata levels 0/1/2, the stream decoder's
metadata reader, PCM container headers) built with AddressSanitizer and UBSan: tools/fuzz_host.sh.  Host code only."""
import os
import shutil
import subprocess

import pytest

import test_metadata_iterators_cpu as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_files(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    bases = []
    for k, data in enumerate([T.sample_stream(False), T.sample_stream(True), T.sample_stream(False, last_padding=False)]):
        p = tmp_path / ("base%d.flac" % k)
        p.write_bytes(data)
        bases.append(str(p))
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz_host.sh"), "1500", "7"] + bases, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in r.stderr:
        pytest.skip("the sanitizer runtime is not installed")
    assert r.returncode == 0, r.stderr[-4000:]
    assert "done: 1500 iterations" in r.stdout
