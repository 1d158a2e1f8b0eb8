"""Metadata level 1 (simple iterator) and level 2 (chain, iterator) against the reference library, call for call.

The same script of edits runs on two copies of one file, one through this repository's library and one through the
reference libFLAC built into oracle/_ref/; after every call the return values, the iterator's view and the statuses
must agree, and after every write the two files must be byte-identical.  Host code only, no GPU.
"""
import ctypes as C
import io
import os
import random
import shutil

import pytest

import test_metadata_object_cpu as M
from test_metadata_object_cpu import Block, block, cuesheet_body, dump, picture_body, streaminfo_body, vorbis_body

P = C.POINTER(Block)
IO_READ = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p)
IO_WRITE = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p)
IO_SEEK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int)
IO_TELL = C.CFUNCTYPE(C.c_int64, C.c_void_p)
IO_EOF = C.CFUNCTYPE(C.c_int, C.c_void_p)
IO_CLOSE = C.CFUNCTYPE(C.c_int, C.c_void_p)


class IOCallbacks(C.Structure):
    _fields_ = [("read", IO_READ), ("write", IO_WRITE), ("seek", IO_SEEK), ("tell", IO_TELL), ("eof", IO_EOF), ("close", IO_CLOSE)]


def bind(path):
    lib = M.bind(path)
    v, i, u = C.c_void_p, C.c_int, C.c_uint
    sig = {
        "FLAC__metadata_simple_iterator_new": (v, []), "FLAC__metadata_simple_iterator_delete": (None, [v]),
        "FLAC__metadata_simple_iterator_status": (i, [v]), "FLAC__metadata_simple_iterator_init": (i, [v, C.c_char_p, i, i]),
        "FLAC__metadata_simple_iterator_is_writable": (i, [v]), "FLAC__metadata_simple_iterator_next": (i, [v]),
        "FLAC__metadata_simple_iterator_prev": (i, [v]), "FLAC__metadata_simple_iterator_is_last": (i, [v]),
        "FLAC__metadata_simple_iterator_get_block_offset": (C.c_int64, [v]), "FLAC__metadata_simple_iterator_get_block_type": (i, [v]),
        "FLAC__metadata_simple_iterator_get_block_length": (u, [v]),
        "FLAC__metadata_simple_iterator_get_application_id": (i, [v, C.c_char_p]),
        "FLAC__metadata_simple_iterator_get_block": (P, [v]), "FLAC__metadata_simple_iterator_set_block": (i, [v, P, i]),
        "FLAC__metadata_simple_iterator_insert_block_after": (i, [v, P, i]), "FLAC__metadata_simple_iterator_delete_block": (i, [v, i]),
        "FLAC__metadata_chain_new": (v, []), "FLAC__metadata_chain_delete": (None, [v]), "FLAC__metadata_chain_status": (i, [v]),
        "FLAC__metadata_chain_read": (i, [v, C.c_char_p]), "FLAC__metadata_chain_read_with_callbacks": (i, [v, v, IOCallbacks]),
        "FLAC__metadata_chain_check_if_tempfile_needed": (i, [v, i]), "FLAC__metadata_chain_write": (i, [v, i, i]),
        "FLAC__metadata_chain_write_with_callbacks": (i, [v, i, v, IOCallbacks]),
        "FLAC__metadata_chain_write_with_callbacks_and_tempfile": (i, [v, i, v, IOCallbacks, v, IOCallbacks]),
        "FLAC__metadata_chain_merge_padding": (None, [v]), "FLAC__metadata_chain_sort_padding": (None, [v]),
        "FLAC__metadata_iterator_new": (v, []), "FLAC__metadata_iterator_delete": (None, [v]), "FLAC__metadata_iterator_init": (None, [v, v]),
        "FLAC__metadata_iterator_next": (i, [v]), "FLAC__metadata_iterator_prev": (i, [v]), "FLAC__metadata_iterator_get_block_type": (i, [v]),
        "FLAC__metadata_iterator_get_block": (P, [v]), "FLAC__metadata_iterator_set_block": (i, [v, P]),
        "FLAC__metadata_iterator_delete_block": (i, [v, i]), "FLAC__metadata_iterator_insert_block_before": (i, [v, P]),
        "FLAC__metadata_iterator_insert_block_after": (i, [v, P]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(M.REF):
        pytest.skip("the reference library has not been built (oracle/Makefile)")
    return bind(M.OURS), bind(M.REF)


AUDIO = b"\xff\xf8\xc9\x18 these bytes stand for the audio frames " * 40


def sample_stream(id3=False, last_padding=True):
    blocks = [block(0, streaminfo_body()), block(4, vorbis_body(b"reference libFLAC 1.2.1 20070917", [b"TITLE=one", b"artist=two"])),
              block(2, b"test" + bytes(range(20))), block(3, b"".join(bytes([0] * 7 + [k]) + bytes(8) + b"\x10\x00" for k in range(3))),
              block(1, bytes(100)), block(5, cuesheet_body()), block(6, picture_body(3, b"image/png", b"front", 10, 10, 24, 0, b"PNGDATA")),
              block(50, b"opaque block of an unknown type"), block(2, b"zero")]
    if last_padding:
        blocks.append(block(1, bytes(64)))
    blocks[-1] = bytes([blocks[-1][0] | 0x80]) + blocks[-1][1:]
    head = b""
    if id3:
        tag = b"some id3 payload" * 3
        n = len(tag)
        head = b"ID3\x03\x00\x00" + bytes([(n >> 21) & 0x7f, (n >> 14) & 0x7f, (n >> 7) & 0x7f, n & 0x7f]) + tag
    return head + b"fLaC" + b"".join(blocks) + AUDIO


def two_files(tmp_path, data, tag):
    paths = []
    for k in ("ours", "ref"):
        p = tmp_path / ("%s_%s.flac" % (tag, k))
        p.write_bytes(data)
        paths.append(str(p).encode())
    return paths


_keep = []


def entry_of(text):
    buf = C.create_string_buffer(text)
    _keep[:] = [buf]   # copied by the callee; alive for the duration of the call
    return M.Entry(len(text), C.cast(buf, C.c_void_p))


def new_block(lib, rng, kind=None):
    """A block made through the object API: (pointer, description); the same rng state gives the same block in both libraries."""
    kind = kind or rng.choice(["padding", "application", "vorbis", "seektable"])
    if kind == "padding":
        b = lib.FLAC__metadata_object_new(1)
        b.contents.length = rng.choice([0, 1, 4, 10, 60, 64, 96, 100, 200])
    elif kind == "application":
        b = lib.FLAC__metadata_object_new(2)
        b.contents.data.application.id[:] = list(b"aPpL")
        data = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 16, 56, 60, 92, 96, 150])))
        assert lib.FLAC__metadata_object_application_set_data(b, data, len(data), 1)
    elif kind == "vorbis":
        b = lib.FLAC__metadata_object_new(4)
        for k in range(rng.randrange(4)):
            text = ("KEY%d=%s" % (k, "v" * rng.randrange(40))).encode()
            e = entry_of(text)
            assert lib.FLAC__metadata_object_vorbiscomment_append_comment(b, e, 1)
    else:
        b = lib.FLAC__metadata_object_new(3)
        assert lib.FLAC__metadata_object_seektable_template_append_placeholders(b, rng.randrange(6))
    return b


# ---------------------------------------------------------------------------------------------
# level 2
# ---------------------------------------------------------------------------------------------
def chain_view(lib, chain):
    it = lib.FLAC__metadata_iterator_new()
    lib.FLAC__metadata_iterator_init(it, chain)
    out = []
    while True:
        assert lib.FLAC__metadata_iterator_get_block_type(it) == lib.FLAC__metadata_iterator_get_block(it).contents.type
        out.append(dump(lib.FLAC__metadata_iterator_get_block(it)))
        if not lib.FLAC__metadata_iterator_next(it):
            break
    back = 0
    while lib.FLAC__metadata_iterator_prev(it):
        back += 1
    assert back == len(out) - 1
    lib.FLAC__metadata_iterator_delete(it)
    return out


@pytest.mark.parametrize("id3", [False, True])
def test_chain_reads_what_the_reference_reads(libs, tmp_path, id3):
    paths = two_files(tmp_path, sample_stream(id3), "read")
    views = []
    for lib, path in zip(libs, paths):
        ch = lib.FLAC__metadata_chain_new()
        assert lib.FLAC__metadata_chain_read(ch, path) == 1
        assert lib.FLAC__metadata_chain_status(ch) == 0
        views.append((chain_view(lib, ch), lib.FLAC__metadata_chain_check_if_tempfile_needed(ch, 0), lib.FLAC__metadata_chain_check_if_tempfile_needed(ch, 1)))
        lib.FLAC__metadata_chain_delete(ch)
    assert views[0] == views[1]
    assert [b["type"] for b in views[0][0]] == [0, 4, 2, 3, 1, 5, 6, 50, 2, 1] and views[0][0][-1]["is_last"] == 1


def test_chain_refuses_what_the_reference_refuses(libs, tmp_path):
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"RIFF this is no FLAC file at all")
    cut = tmp_path / "cut.flac"
    cut.write_bytes(sample_stream()[:60])
    for path in (str(junk).encode(), str(cut).encode(), str(tmp_path / "absent.flac").encode()):
        res = []
        for lib in libs:
            ch = lib.FLAC__metadata_chain_new()
            ok = lib.FLAC__metadata_chain_read(ch, path)
            res.append((ok, lib.FLAC__metadata_chain_status(ch), lib.FLAC__metadata_chain_status(ch)))
            lib.FLAC__metadata_chain_delete(ch)
        assert res[0] == res[1] and res[0][0] == 0 and res[0][1] != 0 and res[0][2] == 0


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("last_padding", [True, False])
def test_chain_edit_scripts(libs, tmp_path, seed, last_padding):
    paths = two_files(tmp_path, sample_stream(id3=seed % 3 == 0, last_padding=last_padding), "edit%d" % seed)
    chains = []
    for lib, path in zip(libs, paths):
        ch = lib.FLAC__metadata_chain_new()
        assert lib.FLAC__metadata_chain_read(ch, path) == 1
        chains.append(ch)
    its = []
    for lib, ch in zip(libs, chains):
        it = lib.FLAC__metadata_iterator_new()
        lib.FLAC__metadata_iterator_init(it, ch)
        its.append(it)
    script = random.Random(seed)
    for step in range(40):
        op = script.choice(["next", "next", "prev", "delete", "delete_pad", "insert_after", "insert_before", "set", "grow_tags", "sort", "merge",
                            "write_pad", "write_nopad"])
        arg = script.randrange(1 << 30)
        res = []
        for lib, ch, it, path in zip(libs, chains, its, paths):
            if op == "next":
                r = lib.FLAC__metadata_iterator_next(it)
            elif op == "prev":
                r = lib.FLAC__metadata_iterator_prev(it)
            elif op in ("delete", "delete_pad"):
                r = lib.FLAC__metadata_iterator_delete_block(it, 1 if op == "delete_pad" else 0)
            elif op in ("insert_after", "insert_before", "set"):
                b = new_block(lib, random.Random(arg))
                if op == "set" and lib.FLAC__metadata_iterator_get_block_type(it) == 0:
                    lib.FLAC__metadata_object_delete(b)   # replacing STREAMINFO trips an assertion in both
                    r = -1
                else:
                    r = getattr(lib, "FLAC__metadata_iterator_" + {"set": "set_block"}.get(op, op.replace("insert", "insert_block")))(it, b)
                    if not r:
                        lib.FLAC__metadata_object_delete(b)
            elif op == "grow_tags":
                cur = lib.FLAC__metadata_iterator_get_block(it)
                r = 0
                if cur.contents.type == 4:
                    text = ("ADDED%d=%s" % (step, "x" * (arg % 90))).encode()
                    e = entry_of(text)
                    r = lib.FLAC__metadata_object_vorbiscomment_append_comment(cur, e, 1)
            elif op == "sort":
                lib.FLAC__metadata_chain_sort_padding(ch)
                lib.FLAC__metadata_iterator_init(it, ch)
                r = 0
            elif op == "merge":
                lib.FLAC__metadata_chain_merge_padding(ch)
                lib.FLAC__metadata_iterator_init(it, ch)
                r = 0
            else:
                pad = 1 if op == "write_pad" else 0
                need = lib.FLAC__metadata_chain_check_if_tempfile_needed(ch, pad)
                r = (need, lib.FLAC__metadata_chain_write(ch, pad, arg & 1), lib.FLAC__metadata_chain_status(ch))
                lib.FLAC__metadata_iterator_init(it, ch)   # a write may append or drop the trailing padding block
            res.append((r, dump(lib.FLAC__metadata_iterator_get_block(it)), chain_view(lib, ch)))
        assert res[0] == res[1], (seed, step, op)
        if op.startswith("write"):
            a, b = open(paths[0], "rb").read(), open(paths[1], "rb").read()
            assert a == b, (seed, step, op)
            assert a.endswith(AUDIO)
    for lib, ch, it in zip(libs, chains, its):
        lib.FLAC__metadata_iterator_delete(it)
        lib.FLAC__metadata_chain_delete(ch)
    assert not [p for p in os.listdir(tmp_path) if p.endswith(".metadata_edit")]


class MemFile:
    """A file in memory behind FLAC__IOCallbacks."""

    def __init__(self, data=b""):
        self.f = io.BytesIO(data)
        self.at_eof = False

        def rd(ptr, size, n, _h):
            b = self.f.read(size * n)
            self.at_eof = len(b) < size * n
            C.memmove(ptr, b, len(b))
            return len(b) // size if size else 0

        def wr(ptr, size, n, _h):
            self.f.write(C.string_at(ptr, size * n))
            return n

        def sk(_h, off, whence):
            self.f.seek(off, whence)
            self.at_eof = False
            return 0
        self.keep = (IO_READ(rd), IO_WRITE(wr), IO_SEEK(sk), IO_TELL(lambda _h: self.f.tell()), IO_EOF(lambda _h: 1 if self.at_eof else 0))
        self.cb = IOCallbacks(self.keep[0], self.keep[1], self.keep[2], self.keep[3], self.keep[4], IO_CLOSE())


@pytest.mark.parametrize("grow", [10, 500])
def test_chain_through_callbacks(libs, grow):
    """Read and write through I/O callbacks: in place when the padding absorbs the change, through a second handle when not;
    the wrong call for the situation is refused (metadata_iterators.c:1728,1769)."""
    out = []
    for lib in libs:
        src = MemFile(sample_stream())
        ch = lib.FLAC__metadata_chain_new()
        assert lib.FLAC__metadata_chain_read_with_callbacks(ch, None, src.cb) == 1
        it = lib.FLAC__metadata_iterator_new()
        lib.FLAC__metadata_iterator_init(it, ch)
        assert lib.FLAC__metadata_iterator_next(it)
        text = b"COMMENT=" + b"y" * grow
        e = entry_of(text)
        assert lib.FLAC__metadata_object_vorbiscomment_append_comment(lib.FLAC__metadata_iterator_get_block(it), e, 1)
        need = lib.FLAC__metadata_chain_check_if_tempfile_needed(ch, 1)
        log = [need, lib.FLAC__metadata_chain_write(ch, 1, 0), lib.FLAC__metadata_chain_status(ch)]   # read with callbacks, written by name
        tmp = MemFile()
        if need:
            log += [lib.FLAC__metadata_chain_write_with_callbacks(ch, 1, None, src.cb), lib.FLAC__metadata_chain_status(ch)]
            log += [lib.FLAC__metadata_chain_write_with_callbacks_and_tempfile(ch, 1, None, src.cb, None, tmp.cb), lib.FLAC__metadata_chain_status(ch)]
            result = tmp.f.getvalue()
        else:
            log += [lib.FLAC__metadata_chain_write_with_callbacks_and_tempfile(ch, 1, None, src.cb, None, tmp.cb), lib.FLAC__metadata_chain_status(ch)]
            log += [lib.FLAC__metadata_chain_write_with_callbacks(ch, 1, None, src.cb), lib.FLAC__metadata_chain_status(ch)]
            result = src.f.getvalue()
        out.append((log, result, chain_view(lib, ch)))
        lib.FLAC__metadata_iterator_delete(it)
        lib.FLAC__metadata_chain_delete(ch)
    assert out[0] == out[1]
    assert out[0][0][0] == (1 if grow == 500 else 0) and out[0][1].endswith(AUDIO) and text in out[0][1]


# ---------------------------------------------------------------------------------------------
# level 1
# ---------------------------------------------------------------------------------------------
def simple_view(lib, it):
    v = {"type": lib.FLAC__metadata_simple_iterator_get_block_type(it), "length": lib.FLAC__metadata_simple_iterator_get_block_length(it),
         "offset": lib.FLAC__metadata_simple_iterator_get_block_offset(it), "is_last": lib.FLAC__metadata_simple_iterator_is_last(it),
         "writable": lib.FLAC__metadata_simple_iterator_is_writable(it)}
    ident = C.create_string_buffer(4)
    v["app"] = (lib.FLAC__metadata_simple_iterator_get_application_id(it, ident), ident.raw, lib.FLAC__metadata_simple_iterator_status(it))
    b = lib.FLAC__metadata_simple_iterator_get_block(it)
    v["block"] = dump(b) if b else None
    if b:
        lib.FLAC__metadata_object_delete(b)
    return v


@pytest.mark.parametrize("id3", [False, True])
def test_simple_iterator_walk(libs, tmp_path, id3):
    paths = two_files(tmp_path, sample_stream(id3), "walk")
    os.chmod(paths[0], 0o444)
    os.chmod(paths[1], 0o444)
    walks = []
    for lib, path in zip(libs, paths):
        it = lib.FLAC__metadata_simple_iterator_new()
        for read_only in (1, 0):
            assert lib.FLAC__metadata_simple_iterator_init(it, path, read_only, 0) == 1
            seen = [simple_view(lib, it)]
            while lib.FLAC__metadata_simple_iterator_next(it):
                seen.append(simple_view(lib, it))
            while lib.FLAC__metadata_simple_iterator_prev(it):
                seen.append(simple_view(lib, it))
            # a file without write permission opens read-only (root may still open it for writing: the same in both)
            pad = lib.FLAC__metadata_object_new(1)
            seen.append((lib.FLAC__metadata_simple_iterator_next(it), lib.FLAC__metadata_simple_iterator_set_block(it, pad, 1) if not seen[0]["writable"] else -1,
                         lib.FLAC__metadata_simple_iterator_status(it)))
            lib.FLAC__metadata_object_delete(pad)
            walks.append(seen)
        lib.FLAC__metadata_simple_iterator_delete(it)
    assert walks[0] == walks[2] and walks[1] == walks[3]
    assert len(walks[0]) == 20 and walks[0][9]["is_last"] == 1 and walks[0][2]["app"][:2] == (1, b"test")


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("last_padding", [True, False])
def test_simple_iterator_edit_scripts(libs, tmp_path, seed, last_padding):
    paths = two_files(tmp_path, sample_stream(id3=seed % 4 == 0, last_padding=last_padding), "simple%d" % seed)
    its = []
    for lib, path in zip(libs, paths):
        it = lib.FLAC__metadata_simple_iterator_new()
        assert lib.FLAC__metadata_simple_iterator_init(it, path, 0, seed & 1) == 1
        its.append(it)
    script = random.Random(1000 + seed)
    for step in range(30):
        op = script.choice(["next", "next", "prev", "set", "set", "insert", "insert", "delete", "delete_pad"])
        pad = script.randrange(2)
        arg = script.randrange(1 << 30)
        res = []
        for lib, it in zip(libs, its):
            if op == "next":
                r = lib.FLAC__metadata_simple_iterator_next(it)
            elif op == "prev":
                r = lib.FLAC__metadata_simple_iterator_prev(it)
            elif op in ("set", "insert"):
                b = new_block(lib, random.Random(arg))
                fn = lib.FLAC__metadata_simple_iterator_set_block if op == "set" else lib.FLAC__metadata_simple_iterator_insert_block_after
                r = fn(it, b, pad)
                lib.FLAC__metadata_object_delete(b)
            else:
                r = lib.FLAC__metadata_simple_iterator_delete_block(it, 1 if op == "delete_pad" else 0)
            res.append((r, lib.FLAC__metadata_simple_iterator_status(it), simple_view(lib, it)))
        assert res[0] == res[1], (seed, step, op, pad)
        a, b = open(paths[0], "rb").read(), open(paths[1], "rb").read()
        assert a == b, (seed, step, op, pad)
        assert a.endswith(AUDIO)
    for lib, it in zip(libs, its):
        lib.FLAC__metadata_simple_iterator_delete(it)
    assert not [p for p in os.listdir(tmp_path) if p.endswith(".metadata_edit")]


def test_file_stats_are_preserved(libs, tmp_path):
    """preserve_file_stats: permissions and modification time survive a rewrite through a temporary file."""
    paths = two_files(tmp_path, sample_stream(last_padding=False), "stats")
    for lib, path in zip(libs, paths):
        os.chmod(path, 0o640)
        os.utime(path, (1_000_000_000, 1_100_000_000))
        ch = lib.FLAC__metadata_chain_new()
        assert lib.FLAC__metadata_chain_read(ch, path) == 1
        it = lib.FLAC__metadata_iterator_new()
        lib.FLAC__metadata_iterator_init(it, ch)
        b = new_block(lib, random.Random(5), "application")
        assert lib.FLAC__metadata_iterator_insert_block_after(it, b)
        assert lib.FLAC__metadata_chain_check_if_tempfile_needed(ch, 1) == 1
        assert lib.FLAC__metadata_chain_write(ch, 1, 1) == 1
        lib.FLAC__metadata_iterator_delete(it)
        lib.FLAC__metadata_chain_delete(ch)
        st = os.stat(path)
        assert (st.st_mode & 0o777, int(st.st_mtime)) == (0o640, 1_100_000_000)
    assert open(paths[0], "rb").read() == open(paths[1], "rb").read()
