"""The drop-in boundary, proven by the linker: the reference's own `flac` command-line tool and its
libFLAC++ (compiled from the reference's sources and headers by oracle/Makefile `on_amd`) link
against libflac_amd.so in place of the reference libFLAC, with no symbol left unresolved, and ask the
loader for the reference library's soname.  No GPU is needed to check that (the tool prints its
version without touching one); tests/test_gpu_reference_cli.py runs the tool for real."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
LIB = os.path.join(ROOT, "flac_amd", "lib", "libflac_amd.so")
CLI = os.path.join(REFDIR, "flac_on_amd")
CXX = os.path.join(REFDIR, "libFLACpp_on_amd.so")

needs_build = pytest.mark.skipif(not (os.path.exists(CLI) and os.path.exists(CXX)), reason="oracle/_ref/*_on_amd not built (needs the reference sources)")


def dynamic_symbols(path, undefined):
    out = subprocess.run(["nm", "-D", "--undefined-only" if undefined else "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}


def test_soname_is_the_reference_librarys():
    dyn = subprocess.run(["readelf", "-d", LIB], check=True, capture_output=True, text=True).stdout
    assert re.search(r"SONAME.*\[libFLAC\.so\.8\]", dyn)   # -version-info 10:0:2, src/libFLAC/Makefile.am:98
    link = os.path.join(os.path.dirname(LIB), "libFLAC.so.8")
    assert os.path.realpath(link) == os.path.realpath(LIB)


@needs_build
@pytest.mark.parametrize("client", [CLI, CXX], ids=["flac", "libFLAC++"])
def test_every_libflac_symbol_the_client_wants_is_exported(client):
    wanted = {s for s in dynamic_symbols(client, undefined=True) if s.startswith("FLAC__") or s.startswith("FLAC_API")}
    have = dynamic_symbols(LIB, undefined=False)
    assert len(wanted) > 40
    assert not (wanted - have), sorted(wanted - have)
    dyn = subprocess.run(["readelf", "-d", client], check=True, capture_output=True, text=True).stdout
    assert "libFLAC.so.8" in dyn and "libFLAC_ref" not in dyn


@needs_build
def test_the_tool_starts_on_top_of_the_library():
    out = subprocess.run([CLI, "--version"], check=True, capture_output=True, text=True).stdout
    assert out.strip() == "flac 1.2.1"
    ldd = subprocess.run(["ldd", CLI], check=True, capture_output=True, text=True).stdout
    assert "not found" not in ldd
    assert os.path.realpath(LIB) in {os.path.realpath(p) for p in re.findall(r"=> (\S+)", ldd)}
