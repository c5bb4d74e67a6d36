"""CPU: the test-side restatement of the reference's on-disk container (tests/refformat.py) is self-consistent:
record round trips, list interleave formulas (ivf_flat.hpp:184-200, ivf_pq.hpp:288-296, ivf_pq_codepacking.cuh)."""
import io

import numpy as np
import pytest

from tests import refformat as rf


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8, np.uint8, np.int64, np.uint32, np.bool_])
def test_record_roundtrip(dtype):
    rng = np.random.default_rng(0)
    a = (rng.integers(0, 2, (5, 3)) if dtype == np.bool_ else rng.integers(0, 100, (5, 3))).astype(dtype)
    f = io.BytesIO()
    rf.write_record(f, a)
    rf.write_scalar(f, 7, np.int32)
    f.seek(0)
    b = rf.read_record(f)
    assert b.dtype == np.dtype(dtype) and (a == b).all()
    assert rf.scalar(f) == 7


def test_raft_half_spelling_is_accepted():
    # RAFT spells float16 "<e2" (c/src/neighbors/ivf_flat.cpp:315); numpy itself writes "<f2"
    a = np.arange(6, dtype=np.float16).reshape(2, 3)
    hdr = "{'descr': '<e2', 'fortran_order': False, 'shape': (2, 3), }"
    hdr = hdr + " " * (64 - (10 + len(hdr) + 1) % 64) + "\n"
    f = io.BytesIO(b"\x93NUMPY\x01\x00" + len(hdr).to_bytes(2, "little") + hdr.encode() + a.tobytes())
    assert (rf.read_record(f) == a).all()


@pytest.mark.parametrize("dtype,dim", [(np.float32, 8), (np.float32, 6), (np.float16, 16), (np.int8, 32), (np.uint8, 20)])
def test_flat_interleave_formula(dtype, dim):
    rng = np.random.default_rng(1)
    n = 45
    rows = rng.integers(-100, 100, (n, dim)).astype(dtype)
    rec = rf.flat_interleave(rows, 64)
    v = rf.flat_veclen(dim, np.dtype(dtype).itemsize)
    flat = rec.reshape(-1)
    for r in (0, 1, 31, 32, 44):
        for d in range(dim):
            off = (r // 32) * 32 * dim + (d // v) * 32 * v + (r % 32) * v + d % v
            assert flat[off] == rows[r, d]
    assert (rf.flat_deinterleave(rec, n) == rows).all()


def test_flat_interleave_doc_example():
    # ivf_flat.hpp:190-199: veclen = 2 would put x[0,0], x[0,1], x[1,0], x[1,1], ... ; with a real dtype the same
    # pattern appears at veclen 4 (fp32, dim 8): x[0,0..3], x[1,0..3], ..., x[31,0..3], x[0,4..7], ...
    rows = np.arange(32 * 8, dtype=np.float32).reshape(32, 8)
    rec = rf.flat_interleave(rows, 32).reshape(-1)
    assert (rec[:8] == [0, 1, 2, 3, 8, 9, 10, 11]).all()
    assert (rec[128:136] == [4, 5, 6, 7, 12, 13, 14, 15]).all()


@pytest.mark.parametrize("pq_bits,pq_dim", [(8, 16), (8, 40), (5, 20), (4, 33), (6, 25), (7, 64)])
def test_pq_chunk_packing(pq_bits, pq_dim):
    rng = np.random.default_rng(pq_bits)
    n = 70
    codes = rng.integers(0, 1 << pq_bits, (n, pq_dim)).astype(np.uint8)
    rec = rf.pq_interleave(codes, pq_bits)
    cpc = 128 // pq_bits
    assert rec.shape == (3, -(-pq_dim // cpc), 32, 16)
    # spot-check the bitfield by hand: code j of row r sits in chunk j // cpc at bit (j % cpc) * pq_bits
    for r in (0, 33, 69):
        for j in (0, pq_dim // 2, pq_dim - 1):
            chunk = int.from_bytes(rec[r // 32, j // cpc, r % 32].tobytes(), "little")
            assert (chunk >> ((j % cpc) * pq_bits)) & ((1 << pq_bits) - 1) == codes[r, j]
    assert (rf.pq_deinterleave(rec, n, pq_dim, pq_bits) == codes).all()
    assert (rf.bitstream_to_codes(rf.codes_to_bitstream(codes, pq_bits), pq_dim, pq_bits) == codes).all()
