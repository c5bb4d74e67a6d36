"""Independent Python restatement of the reference's on-disk index container, used only by the tests.

Every scalar / mdspan is one NumPy ".npy" record (raft::serialize_scalar / serialize_mdspan; call sites
cpp/src/neighbors/ivf_pq/ivf_pq_serialize.cuh:49-85, ivf_flat/ivf_flat_serialize.cuh:42-77, ivf_list.cuh:108-131,
detail/cagra/cagra_serialize.cuh:49-75, detail/dataset_serialize.hpp:38-101, brute_force_serialize.cu:30-45).
Records are written with numpy's own writer (different header spelling than the C++ writer in
cuvs_amd/csrc/npy_io.hpp), and parsed with a small parser that also accepts RAFT's "<e2" spelling of float16.
"""
import ast
import io
import struct

import numpy as np


# ---------------------------------------------------------------------------------------------- records
def read_record(f):
    magic = f.read(6)
    if magic != b"\x93NUMPY":
        raise ValueError("bad numpy magic %r" % (magic,))
    major, _minor = f.read(2)
    hlen = struct.unpack("<H", f.read(2))[0] if major == 1 else struct.unpack("<I", f.read(4))[0]
    start = f.tell() - (10 if major == 1 else 12)
    header = f.read(hlen).decode("latin1")
    assert (f.tell() - start) % 64 == 0, "payload must start on a 64-byte boundary"
    d = ast.literal_eval(header)
    descr = d["descr"]
    if descr[1:] == "e2":
        descr = descr[0] + "f2"
    dt = np.dtype(descr)
    shape = tuple(d["shape"])
    assert not d["fortran_order"]
    count = int(np.prod(shape)) if shape else 1
    a = np.frombuffer(f.read(count * dt.itemsize), dtype=dt)
    return a.reshape(shape).copy()


def scalar(f):
    a = read_record(f)
    assert a.shape == ()
    return a.item()


def write_record(f, a):
    np.lib.format.write_array(f, np.asarray(a, order="C"), version=(1, 0))


def write_scalar(f, v, dtype):
    write_record(f, np.array(v, dtype=dtype))


PREFIX = {np.dtype("float32"): b"<f4\0", np.dtype("float16"): b"<e2\0", np.dtype("int8"): b"|i1\0",
          np.dtype("uint8"): b"|u1\0"}
CUDA_DTYPE = {np.dtype("float32"): 0, np.dtype("float16"): 2, np.dtype("int8"): 3, np.dtype("uint8"): 8}


# ---------------------------------------------------------------------------------------------- layouts
def flat_veclen(dim, itemsize):
    """ivf_flat.hpp:284-294"""
    v = max(1, 16 // itemsize)
    return 1 if dim % v else v


def flat_interleave(rows, rows32):
    """[n, dim] -> the reference's list record [rows32, dim] (ivf_flat.hpp:184-200: groups of 32 rows, chunks of
    veclen components, row-fastest inside a chunk column)."""
    n, dim = rows.shape
    v = flat_veclen(dim, rows.dtype.itemsize)
    padded = np.zeros((rows32, dim), rows.dtype)
    padded[:n] = rows
    g = padded.reshape(rows32 // 32, 32, dim // v, v)      # [group, row, chunk, comp]
    return np.ascontiguousarray(g.transpose(0, 2, 1, 3)).reshape(rows32, dim)


def flat_deinterleave(rec, n):
    rows32, dim = rec.shape
    v = flat_veclen(dim, rec.dtype.itemsize)
    g = rec.reshape(rows32 // 32, dim // v, 32, v).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(g).reshape(rows32, dim)[:n]


def pq_pack_chunks(codes, pq_bits):
    """[n, pq_dim] uint8 codes -> [n, n_chunks, 16] bytes: chunk c holds codes [c*cpc, (c+1)*cpc) as a little-endian
    bitfield starting at bit 0 of the chunk (ivf_pq_codepacking.cuh:22-52,106-137)."""
    n, pq_dim = codes.shape
    cpc = 128 // pq_bits
    n_chunks = -(-pq_dim // cpc)
    bits = np.zeros((n, n_chunks, 128), np.uint8)
    for j in range(pq_dim):
        c, o = divmod(j, cpc)
        for b in range(pq_bits):
            bits[:, c, o * pq_bits + b] = (codes[:, j] >> b) & 1
    return np.packbits(bits, axis=2, bitorder="little")


def pq_unpack_chunks(chunks, pq_dim, pq_bits):
    n, n_chunks, _ = chunks.shape
    cpc = 128 // pq_bits
    bits = np.unpackbits(chunks, axis=2, bitorder="little")
    codes = np.zeros((n, pq_dim), np.uint8)
    for j in range(pq_dim):
        c, o = divmod(j, cpc)
        for b in range(pq_bits):
            codes[:, j] |= bits[:, c, o * pq_bits + b] << b
    return codes


def pq_interleave(codes, pq_bits):
    """codes [n, pq_dim] -> reference list record [ceil(n/32), n_chunks, 32, 16] (ivf_pq.hpp:288-296)."""
    n = codes.shape[0]
    ch = pq_pack_chunks(codes, pq_bits)
    g32 = -(-n // 32)
    padded = np.zeros((g32 * 32,) + ch.shape[1:], np.uint8)
    padded[:n] = ch
    return np.ascontiguousarray(padded.reshape(g32, 32, ch.shape[1], 16).transpose(0, 2, 1, 3))


def pq_deinterleave(rec, n, pq_dim, pq_bits):
    g32, n_chunks, _, _ = rec.shape
    ch = np.ascontiguousarray(rec.transpose(0, 2, 1, 3)).reshape(g32 * 32, n_chunks, 16)[:n]
    return pq_unpack_chunks(ch, pq_dim, pq_bits)


def bitstream_to_codes(packed, pq_dim, pq_bits):
    """contiguous per-row little-endian bitstream [n, ceil(pq_dim*pq_bits/8)] -> [n, pq_dim]"""
    bits = np.unpackbits(packed, axis=1, bitorder="little")
    codes = np.zeros((packed.shape[0], pq_dim), np.uint8)
    for j in range(pq_dim):
        for b in range(pq_bits):
            codes[:, j] |= bits[:, j * pq_bits + b] << b
    return codes


def codes_to_bitstream(codes, pq_bits):
    n, pq_dim = codes.shape
    nbytes = -(-pq_dim * pq_bits // 8)
    bits = np.zeros((n, nbytes * 8), np.uint8)
    for j in range(pq_dim):
        for b in range(pq_bits):
            bits[:, j * pq_bits + b] = (codes[:, j] >> b) & 1
    return np.packbits(bits, axis=1, bitorder="little")


# ---------------------------------------------------------------------------------------------- files
def parse_brute_force(path):
    with open(path, "rb") as f:
        out = {"prefix": f.read(4)}
        out["version"] = scalar(f)
        out["rows"], out["dim"] = scalar(f), scalar(f)
        out["metric"], out["metric_arg"] = scalar(f), scalar(f)
        if scalar(f):
            out["dataset"] = read_record(f)
        if scalar(f):
            out["norms"] = read_record(f)
        assert f.read(1) == b""
    return out


def write_brute_force(path, dataset, metric=0, metric_arg=2.0, norms=None):
    with open(path, "wb") as f:
        f.write(PREFIX[dataset.dtype])
        write_scalar(f, 0, np.int32)
        write_scalar(f, dataset.shape[0], np.uint64)
        write_scalar(f, dataset.shape[1], np.uint64)
        write_scalar(f, metric, np.int32)
        write_scalar(f, metric_arg, np.float32)
        write_scalar(f, True, np.bool_)
        write_record(f, dataset)
        write_scalar(f, norms is not None, np.bool_)
        if norms is not None:
            write_record(f, norms.astype(np.float32))


def parse_ivf_flat(path):
    with open(path, "rb") as f:
        out = {"prefix": f.read(4)}
        out["version"] = scalar(f)
        out["size"], out["dim"], out["n_lists"], out["metric"] = scalar(f), scalar(f), scalar(f), scalar(f)
        out["adaptive_centers"], out["cma"] = scalar(f), scalar(f)
        out["centers"] = read_record(f)
        if scalar(f):
            out["center_norms"] = read_record(f)
        out["list_sizes"] = read_record(f)
        out["rows"], out["ids"] = [], []
        for L in range(out["n_lists"]):
            rows32 = scalar(f)
            n = int(out["list_sizes"][L])
            assert rows32 == -(-n // 32) * 32
            if rows32 == 0:
                out["rows"].append(None); out["ids"].append(np.zeros(0, np.int64))
                continue
            rec, ids = read_record(f), read_record(f)
            assert rec.shape == (rows32, out["dim"]) and ids.shape == (rows32,)
            out["rows"].append(flat_deinterleave(rec, n)); out["ids"].append(ids[:n])
        assert f.read(1) == b""
    return out


def write_ivf_flat(path, centers, rows, ids, metric=0, dtype=np.float32):
    """rows: list of [n_L, dim] arrays (original dtype), ids: list of int64 [n_L]."""
    dtype = np.dtype(dtype)
    n_lists, dim = centers.shape
    sizes = np.array([len(i) for i in ids], np.uint32)
    with open(path, "wb") as f:
        f.write(PREFIX[dtype])
        write_scalar(f, 4, np.int32)
        write_scalar(f, int(sizes.sum()), np.int64)
        write_scalar(f, dim, np.uint32)
        write_scalar(f, n_lists, np.uint32)
        write_scalar(f, metric, np.int32)
        write_scalar(f, False, np.bool_)
        write_scalar(f, True, np.bool_)
        write_record(f, centers.astype(np.float32))
        has_norms = metric != 6
        write_scalar(f, has_norms, np.bool_)
        if has_norms:
            write_record(f, (centers.astype(np.float32) ** 2).sum(1).astype(np.float32))
        write_record(f, sizes)
        for L in range(n_lists):
            n = int(sizes[L])
            rows32 = -(-n // 32) * 32
            write_scalar(f, rows32, np.uint32)
            if rows32 == 0:
                continue
            write_record(f, flat_interleave(np.asarray(rows[L], dtype).reshape(n, dim), rows32))
            padded = np.full(rows32, -1, np.int64)
            padded[:n] = ids[L]
            write_record(f, padded)


def parse_ivf_pq(path):
    with open(path, "rb") as f:
        out = {"version": scalar(f)}
        for k in ("size", "dim", "pq_bits", "pq_dim", "cma", "metric", "codebook_kind", "codes_layout", "n_lists"):
            out[k] = scalar(f)
        for k in ("pq_centers", "centers", "centers_rot", "rotation", "list_sizes"):
            out[k] = read_record(f)
        out["codes"], out["ids"] = [], []
        cpc = 128 // out["pq_bits"]
        n_chunks = -(-out["pq_dim"] // cpc)
        for L in range(out["n_lists"]):
            n = scalar(f)
            assert n == out["list_sizes"][L]
            if n == 0:
                out["codes"].append(np.zeros((0, out["pq_dim"]), np.uint8)); out["ids"].append(np.zeros(0, np.int64))
                continue
            rec, ids = read_record(f), read_record(f)
            assert ids.shape == (n,)
            if out["codes_layout"] == 1:
                assert rec.shape == (-(-n // 32), n_chunks, 32, 16)
                out["codes"].append(pq_deinterleave(rec, n, out["pq_dim"], out["pq_bits"]))
            else:
                out["codes"].append(bitstream_to_codes(rec, out["pq_dim"], out["pq_bits"]))
            out["ids"].append(ids)
        assert f.read(1) == b""
    return out


def write_ivf_pq(path, dim, pq_bits, pq_dim, metric, pq_centers, centers_ext, centers_rot, rotation, codes, ids,
                 layout=1):
    """codes: list of [n_L, pq_dim] uint8 code arrays."""
    n_lists = centers_ext.shape[0]
    sizes = np.array([len(i) for i in ids], np.uint32)
    with open(path, "wb") as f:
        write_scalar(f, 4, np.int32)
        write_scalar(f, int(sizes.sum()), np.int64)
        write_scalar(f, dim, np.uint32)
        write_scalar(f, pq_bits, np.uint32)
        write_scalar(f, pq_dim, np.uint32)
        write_scalar(f, True, np.bool_)
        write_scalar(f, metric, np.int32)
        write_scalar(f, 0, np.int32)
        write_scalar(f, layout, np.int32)
        write_scalar(f, n_lists, np.uint32)
        for a in (pq_centers, centers_ext, centers_rot, rotation):
            write_record(f, a.astype(np.float32))
        write_record(f, sizes)
        for L in range(n_lists):
            n = int(sizes[L])
            write_scalar(f, n, np.uint32)
            if n == 0:
                continue
            c = np.asarray(codes[L], np.uint8).reshape(n, pq_dim)
            write_record(f, pq_interleave(c, pq_bits) if layout == 1 else codes_to_bitstream(c, pq_bits))
            write_record(f, np.asarray(ids[L], np.int64))


def parse_cagra(path):
    with open(path, "rb") as f:
        out = {"prefix": f.read(4)}
        out["version"] = scalar(f)
        out["size"], out["dim"], out["graph_degree"], out["metric"] = scalar(f), scalar(f), scalar(f), scalar(f)
        out["graph"] = read_record(f)
        out["content_map"] = scalar(f)
        if out["content_map"] & 1:
            out["tag"], out["cuda_dtype"] = scalar(f), scalar(f)
            out["n_rows"], out["ds_dim"], out["stride"] = scalar(f), scalar(f), scalar(f)
            out["dataset"] = read_record(f)
        if out["content_map"] & 2:  # cagra_serialize.cuh:83: the source id of every row, after the dataset
            out["source_indices"] = read_record(f)
        assert f.read(1) == b""
    return out


def write_cagra(path, graph, dataset=None, metric=0, dtype=np.float32, source_indices=None):
    dtype = np.dtype(dtype)
    with open(path, "wb") as f:
        f.write(PREFIX[dtype])
        write_scalar(f, 5, np.int32)
        write_scalar(f, graph.shape[0], np.uint32)
        write_scalar(f, dataset.shape[1] if dataset is not None else 0, np.uint32)
        write_scalar(f, graph.shape[1], np.uint32)
        write_scalar(f, metric, np.int32)
        write_record(f, graph.astype(np.uint32))
        write_scalar(f, (1 if dataset is not None else 0) | (2 if source_indices is not None else 0), np.uint32)
        if dataset is not None:
            write_scalar(f, 2, np.uint32)
            write_scalar(f, CUDA_DTYPE[dtype], np.uint32)
            write_scalar(f, dataset.shape[0], np.int64)
            write_scalar(f, dataset.shape[1], np.uint32)
            write_scalar(f, -(-dataset.shape[1] * dtype.itemsize // 16) * 16 // dtype.itemsize, np.uint32)
            write_record(f, dataset.astype(dtype))
        if source_indices is not None:
            write_record(f, np.asarray(source_indices, np.uint32))


def parse_hnswlib(path, dim, dtype):
    """layout of cagra_serialize.cuh:98-258"""
    dtype = np.dtype(dtype)
    b = open(path, "rb").read()
    off0, max_el, cur, per_elem, label_off, off_data = struct.unpack_from("<6Q", b, 0)
    max_level, entry = struct.unpack_from("<2i", b, 48)
    max_m, max_m0, m = struct.unpack_from("<3Q", b, 56)
    mult, = struct.unpack_from("<d", b, 80)
    efc, = struct.unpack_from("<Q", b, 88)
    deg = max_m0
    assert per_elem == deg * 4 + 4 + dim * dtype.itemsize + 8 and label_off == per_elem - 8 and off_data == deg * 4 + 4
    body = np.frombuffer(b, np.uint8, count=cur * per_elem, offset=96).reshape(cur, per_elem)
    degs = body[:, :4].copy().view(np.int32)[:, 0]
    graph = body[:, 4:4 + deg * 4].copy().view(np.uint32)
    data = body[:, off_data:off_data + dim * dtype.itemsize].copy().view(dtype)
    labels = body[:, label_off:].copy().view(np.uint64)[:, 0]
    tail = np.frombuffer(b, np.int32, offset=96 + cur * per_elem)
    return dict(off0=off0, max_elements=max_el, count=cur, max_level=max_level, entry=entry, max_m=max_m, m=m,
                mult=mult, ef_construction=efc, degrees=degs, graph=graph, data=data, labels=labels, tail=tail)
