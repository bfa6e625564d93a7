"""CPU: the C-ABI library loads, exports every symbol include/sbwt_amd.h declares, and its host-only
entry points behave; GPU entry points fail loudly (no fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sbwt_amd
from sbwt_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(api.lib_path()):
        from sbwt_amd import build
        build.build(verbose=False)
    return api.lib()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sbwt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sbwt_[a-z0-9_]+)\s*\(", hdr)))


def test_exports_every_declared_symbol(L):
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), "libsbwt_amd.so does not export " + s


def test_words_per_read(L):
    assert [api.words_per_read(x) for x in (1, 16, 64, 65, 100, 128, 129)] == [4, 4, 4, 8, 8, 8, 12]


def test_pack_reads_encoding(L):
    packed, lens = api.pack_reads(["ACGTACGTACGTA", "T" * 70])
    assert list(lens) == [13, 70]
    assert packed.shape == (2, 8)
    # base j at bits 2*(j%16) of word j/16, A=0 C=1 G=2 T=3
    w0 = int(packed[0, 0])
    assert [(w0 >> (2 * j)) & 3 for j in range(13)] == [0, 1, 2, 3] * 3 + [0]
    assert int(packed[1, 4]) == (1 << 12) - 1 and int(packed[1, 3]) == 0xFFFFFFFF


def test_pack_reads_every_length_and_every_byte(L):
    """the packer works eight bases at a time: all tail lengths, and every byte value in every lane"""
    rng = np.random.default_rng(5)
    reads = ["".join("ACGT"[c] for c in rng.integers(0, 4, n)) for n in range(12, 131)]
    packed, lens = api.pack_reads(reads)
    for r, s in enumerate(reads):
        got = [(int(packed[r, j >> 4]) >> (2 * (j & 15))) & 3 for j in range(len(s))]
        assert got == ["ACGT".index(ch) for ch in s], "length %d" % len(s)
        tail = [(int(packed[r, j >> 4]) >> (2 * (j & 15))) & 3 for j in range(len(s), packed.shape[1] * 16)]
        assert not any(tail)
    base = "ACGTTGCAACGTGGATCCAT" * 2
    h = C.c_uint64()
    out = np.zeros(4, np.uint32)
    ln = np.zeros(1, np.uint32)
    for pos in range(0, 40):
        for b in range(1, 256):
            if chr(b) in "ACGT":
                continue
            s = bytearray(base.encode())
            s[pos] = b
            off = np.array([0, len(s)], np.uint64)
            rc = L.sbwt_pack_reads(bytes(s), off.ctypes.data_as(C.c_void_p), 1, 4, out.ctypes.data_as(C.c_void_p),
                                   ln.ctypes.data_as(C.c_void_p), C.byref(h))
            assert rc == -5, (pos, b)


def test_pack_reads_rejects_what_upstream_aborts_on(L):
    with pytest.raises(api.SbwtError) as e:
        api.pack_reads(["ACGTACGTACGTACGT", "ACGTNACGTACGTACGT"])
    assert e.value.code == -5 and "read 1" in str(e.value)
    with pytest.raises(api.SbwtError) as e:
        api.pack_reads(["ACGTACGTACG"])
    assert e.value.code == -6
    with pytest.raises(api.SbwtError) as e:   # lowercase is illegal upstream too
        api.pack_reads(["acgtacgtacgtacgt"])
    assert e.value.code == -5
    with pytest.raises(api.SbwtError) as e:   # illegal char wins over too-short (the RC loop runs first)
        api.pack_reads(["ACGN"])
    assert e.value.code == -5


def test_index_load_errors_are_reported_not_thrown(L, tmp_path):
    h = C.c_void_p()
    rc = L.sbwt_index_load(os.fsencode(str(tmp_path / "nothing")), 0, C.byref(h))
    assert rc == -1 and b"cannot open" in L.sbwt_last_error()
    (tmp_path / "bad.t_table.bwt").write_bytes(b"x" * 100)
    rc = L.sbwt_index_load(os.fsencode(str(tmp_path / "bad")), 0, C.byref(h))
    assert rc == -2 and not h.value


def test_no_cpu_fallback_without_gpu(L, golden_indexes):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = C.c_int(-1)
    assert L.sbwt_device_count(C.byref(n)) == -3 and n.value == 0
    with pytest.raises(api.SbwtError) as e:
        sbwt_amd.Index.load(golden_indexes["g1"])
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
