"""Host entry points of libmecha_hip.so (csrc/hostio.cpp): the scheduler's file traffic.  No GPU involved: these run
wherever the library loads.  They replace the reference's locked read window + clone (streaming.py:151-178) and its
buffered writer (streaming.py:277-308) for the batched path, so the check is byte equality with plain file I/O."""
import ctypes
import os

import numpy as np
import pytest

from sd_mecha_amd import _lib


def _arr(values, dtype):
    return np.ascontiguousarray(values, dtype=dtype)


def test_pread_many_scatters_and_zero_fills(tmp_path):
    L = _lib.lib()
    rng = np.random.default_rng(0)
    blobs = [rng.integers(0, 256, size=n, dtype=np.uint8) for n in (10_000_001, 4096, 123)]      # > 4 MiB: split into pieces
    fds = []
    for i, b in enumerate(blobs):
        p = tmp_path / f"f{i}.bin"
        p.write_bytes(b.tobytes())
        fds.append(os.open(p, os.O_RDONLY))
    try:
        dst = np.full(12_000_000, 0xAB, dtype=np.uint8)
        # (fd index, file offset, nbytes, dst offset); fd -1 = zero fill
        plan = [(0, 0, 10_000_001, 64), (1, 100, 3000, 10_100_000), (2, 3, 120, 11_000_000), (-1, 0, 500, 11_500_000), (1, 0, 0, 5)]
        fd_a = _arr([fds[i] if i >= 0 else -1 for i, *_ in plan], np.int32)
        dst_a = _arr([dst.ctypes.data + d for *_, d in plan], np.uint64)
        size_a = _arr([n for _, _, n, _ in plan], np.uint64)
        off_a = _arr([o for _, o, _, _ in plan], np.int64)
        for threads in (1, 5):
            dst[:] = 0xAB
            _lib.check(L.mecha_host_pread_many(fd_a.ctypes.data, dst_a.ctypes.data, size_a.ctypes.data, off_a.ctypes.data, len(plan), threads), "pread")
            assert np.array_equal(dst[64:64 + 10_000_001], blobs[0])
            assert np.array_equal(dst[10_100_000:10_103_000], blobs[1][100:3100])
            assert np.array_equal(dst[11_000_000:11_000_120], blobs[2][3:123])
            assert not dst[11_500_000:11_500_500].any()
            assert (dst[:64] == 0xAB).all() and (dst[11_500_500:] == 0xAB).all()          # nothing else touched
        # reading past the end of a file is an error, not a short read
        bad_size = _arr([200], np.uint64)
        rc = L.mecha_host_pread_many(_arr([fds[2]], np.int32).ctypes.data, dst_a.ctypes.data, bad_size.ctypes.data, _arr([0], np.int64).ctypes.data, 1, 2)
        assert rc < 0 and b"pread" in L.mecha_last_error()
    finally:
        for fd in fds:
            os.close(fd)


@pytest.mark.parametrize("entry", ["mecha_host_pwrite_many", "mecha_host_mmap_write_many"])
def test_write_many(tmp_path, entry):
    L = _lib.lib()
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, size=9_000_000, dtype=np.uint8)
    path = tmp_path / "out.bin"
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        total = 9_500_000
        os.ftruncate(fd, total)
        # (src offset, nbytes, file offset): unaligned offsets, one piece larger than 4 MiB, one empty
        plan = [(0, 5_000_003, 4097), (5_000_003, 2_999_997, 6_000_001), (8_000_000, 0, 17), (8_000_000, 400_000, 9_000_001), (8_500_000, 4000, 1)]
        src_a = _arr([src.ctypes.data + s for s, _, _ in plan], np.uint64)
        size_a = _arr([n for _, n, _ in plan], np.uint64)
        off_a = _arr([o for _, _, o in plan], np.int64)
        _lib.check(getattr(L, entry)(fd, src_a.ctypes.data, size_a.ctypes.data, off_a.ctypes.data, len(plan), 4), entry)
        got = np.frombuffer(path.read_bytes(), dtype=np.uint8)
        assert got.size == total
        want = np.zeros(total, dtype=np.uint8)
        for s, n, o in plan:
            want[o:o + n] = src[s:s + n]
        assert np.array_equal(got, want)
        if entry == "mecha_host_mmap_write_many":
            # the mapping variant refuses to write past the end of the file instead of faulting
            far = _arr([total - 10], np.int64)
            rc = getattr(L, entry)(fd, src_a.ctypes.data, _arr([100], np.uint64).ctypes.data, far.ctypes.data, 1, 1)
            assert rc < 0 and b"ftruncate" in L.mecha_last_error()
    finally:
        os.close(fd)


def test_writer_put_many_matches_put(tmp_path):
    """OutSafetensorsDict.put_many (native team, preallocated file) writes the same file contents as per-key put()."""
    import torch
    from sd_mecha_amd.streaming import InSafetensorsDict, OutSafetensorsDict, TensorMetadata
    rng = np.random.default_rng(2)
    tensors = {f"k{i}": torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(torch.bfloat16) for i, n in enumerate((5, 70_001, 0, 1 << 21))}
    header = {k: TensorMetadata(t.shape, t.dtype) for k, t in tensors.items()}
    a = OutSafetensorsDict(tmp_path / "a.safetensors", header, "recipe")
    for k, t in tensors.items():
        a[k] = t
    a.close()
    b = OutSafetensorsDict(tmp_path / "b.safetensors", header, "recipe")
    keys = list(tensors)
    b.put_many(keys, [header[k] for k in keys], [tensors[k].data_ptr() for k in keys], threads=3)
    b.close()
    ra, rb = InSafetensorsDict(tmp_path / "a.safetensors"), InSafetensorsDict(tmp_path / "b.safetensors")
    assert os.path.getsize(tmp_path / "a.safetensors") == os.path.getsize(tmp_path / "b.safetensors")
    assert ra.file_metadata() == rb.file_metadata() == {"mecha_recipe": "recipe"}
    for k, t in tensors.items():
        assert torch.equal(ra[k], t) and torch.equal(rb[k], t)


def test_mmap_write_many_with_cache_bypassing_stores(tmp_path):
    """MECHA_HOST_NT=1 (read once per process, hence the child): the writer's copy through non-temporal stores, with pieces
    that start and end off the 32-byte grid."""
    import subprocess
    import sys
    code = f"""
import os, numpy as np
from sd_mecha_amd import _lib
L = _lib.lib()
src = np.random.default_rng(3).integers(0, 256, size=9_000_000, dtype=np.uint8)
fd = os.open({str(tmp_path / 'nt.bin')!r}, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
os.ftruncate(fd, 9_500_000)
plan = [(0, 5_000_003, 17), (5_000_003, 3_999_990, 5_100_001), (8_999_993, 7, 9_400_000)]
srcs = np.ascontiguousarray([src.ctypes.data + s for s, _, _ in plan], dtype=np.uint64)
sizes = np.ascontiguousarray([n for _, n, _ in plan], dtype=np.uint64)
offs = np.ascontiguousarray([o for _, _, o in plan], dtype=np.int64)
_lib.check(L.mecha_host_mmap_write_many(fd, srcs.ctypes.data, sizes.ctypes.data, offs.ctypes.data, len(plan), 4), "mmap write")
os.close(fd)
got = np.fromfile({str(tmp_path / 'nt.bin')!r}, dtype=np.uint8)
want = np.zeros(9_500_000, dtype=np.uint8)
for s, n, o in plan:
    want[o:o + n] = src[s:s + n]
assert np.array_equal(got, want)
"""
    env = dict(os.environ, MECHA_HOST_NT="1", PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=120)


def test_native_safetensors_header_matches_the_json_reader(tmp_path):
    """mecha_host_parse_safetensors_header (the read side's replacement for json.loads, reference streaming.py:141-149): the same
    keys in the same (data offset) order with the same shapes, dtypes and offsets as the JSON path, non-ASCII names and a
    ``__metadata__`` value full of braces and quotes included; a header it does not read (a name with a JSON escape, 9 dimensions,
    an unknown dtype tag, malformed text) falls back to the JSON path."""
    import json
    import struct
    import torch
    import safetensors.torch
    from sd_mecha_amd import streaming
    tensors = {
        "plain.weight": torch.zeros(3, 5, dtype=torch.float16), "scalar": torch.tensor(2.0), "empty": torch.zeros(0, 4, dtype=torch.bfloat16),
        "unicode é中": torch.zeros(2, 2, 2, dtype=torch.int64),
        "bool": torch.zeros(7, dtype=torch.bool), "u8": torch.zeros(9, dtype=torch.uint8), "i32": torch.zeros(1, 1, 1, 1, dtype=torch.int32),
    }
    tensors.update({f"lora_unet_{i}.lora_up.weight": torch.zeros(4, 2, dtype=torch.float16) for i in range(300)})
    path = tmp_path / "h.safetensors"
    safetensors.torch.save_file(tensors, str(path), metadata={"mecha_recipe": 'merge "x" {"a": [1, 2]} \\ }{', "format": "pt"})
    tables = streaming.read_header_tables(path)
    assert tables is not None
    size, meta, file_meta = tables
    size_json, header = streaming.read_header(path)
    assert size == size_json
    want = [(k, v) for k, v in header.items() if k != "__metadata__"]
    assert list(meta) == [k for k, _ in want]
    for k, v in want:
        m = meta[k]
        assert m.shape == tuple(v["shape"]) and m.dtype == streaming.DTYPE_MAPPING[v["dtype"]][0] and [m.start, m.end] == v["data_offsets"], k
    assert file_meta == header["__metadata__"]
    reader = streaming.InSafetensorsDict(path)
    assert set(reader.keys()) == set(tensors) and reader.file_metadata()["format"] == "pt"
    assert torch.equal(reader["plain.weight"], tensors["plain.weight"]) and reader["unicode é中"].shape == (2, 2, 2)
    assert "lora_unet_7.lora_up.weight" in reader and "lora_unet_7.lora_down.weight" not in reader and len(reader) == len(tensors)
    reader.close()
    esc = {'quote"and\\backslash': torch.zeros(2, dtype=torch.float32), "b": torch.ones(3)}
    p1 = tmp_path / "esc.safetensors"
    safetensors.torch.save_file(esc, str(p1))
    assert streaming.read_header_tables(p1) is None
    r1 = streaming.InSafetensorsDict(p1)
    assert r1['quote"and\\backslash'].shape == (2,) and torch.equal(r1["b"], esc["b"])
    r1.close()
    # a header the scanner does not read (9 dimensions; a dtype tag it does not know): None, and the reader still opens the file
    odd = {"nine": {"dtype": "F32", "shape": [1] * 9, "data_offsets": [0, 4]}}
    text = json.dumps(odd).encode()
    p2 = tmp_path / "odd.safetensors"
    p2.write_bytes(struct.pack("<Q", len(text)) + text + b"\0" * 4)
    assert streaming.read_header_tables(p2) is None
    assert streaming.InSafetensorsDict(p2).meta("nine").shape == (1,) * 9
    text = json.dumps({"x": {"dtype": "F8_E4M3", "shape": [4], "data_offsets": [0, 4]}}).encode()
    p3 = tmp_path / "f8.safetensors"
    p3.write_bytes(struct.pack("<Q", len(text)) + text + b"\0" * 4)
    assert streaming.read_header_tables(p3) is None
    for bad in (b"", b"{", b'{"a": {"dtype": "F32", "shape": [1], "data_offsets": [0]}}', b'{"a": 3}', b'{"a": {"dtype": "F32", "shape": [1], "data_offsets": [0, 4]}} x'):
        p4 = tmp_path / "bad.safetensors"
        p4.write_bytes(struct.pack("<Q", len(bad)) + bad)
        assert streaming.read_header_tables(p4) is None
