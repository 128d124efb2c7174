#!/usr/bin/env python3
"""Minimal reader for the JLD2 (HDF5 subset) files Dojo's examples ship -- stdlib + numpy only.

Test infrastructure: it exists to turn the one file of reference-COMPUTED numbers in the reference tree,
    examples/system_identification/data/datasets/synthetic_sphere.jld2
(written by examples/system_identification/synthetic_sphere.jl:16-42 through src/simulation/simulate.jl:16-50 and
src/simulation/storage.jl:50-67), into the committed fixture tests/golden/reference_sphere.npz.

What it understands (all that JLD2 0.4 writes for plain structs and arrays, uncompressed):
  * the HDF5 version-2 superblock JLD2 places at byte 512, version-2 object headers (OHDR) with continuation
    chunks (OCHK), link messages of the root group;
  * dataspace (0x01), datatype (0x03: fixed-point, float, compound, reference; committed = shared datatypes),
    data layout version 4 (0x08: compact and contiguous), attributes are skipped;
  * JLD2's object references: 8-byte offsets relative to the superblock's base address.
Anything else raises.  Usage as a script:  python tools/jld2_reader.py <file.jld2> [out.npz]
"""
import struct
import sys

import numpy as np


class JLD2File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.d = f.read()
        d = self.d
        if not d.startswith(b"HDF5-based Julia Data Format"):
            raise ValueError("not a JLD2 file")
        sb = 512
        if d[sb:sb + 8] != b"\x89HDF\r\n\x1a\n" or d[sb + 8] != 2:
            raise ValueError("expected a version-2 HDF5 superblock at byte 512")
        if d[sb + 9] != 8 or d[sb + 10] != 8:
            raise ValueError("only 8-byte offsets / lengths")
        self.base, _ext, _eof, root = struct.unpack_from("<QQQQ", d, sb + 12)
        self.root = root
        self.header = d[:d.index(b"\x00")].decode()

    # ---- object headers -------------------------------------------------------------------------------------------
    def messages(self, rel):
        """All header messages [(type, flags, body bytes)] of the object at relative offset `rel`."""
        d = self.d
        off = self.base + rel
        if d[off:off + 4] != b"OHDR" or d[off + 4] != 2:
            raise ValueError("no version-2 object header at %d" % off)
        flags = d[off + 5]
        p = off + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        n = 1 << (flags & 3)
        size = int.from_bytes(d[p:p + n], "little")
        p += n
        chunks = [(p, p + size)]
        out = []
        while chunks:
            p, end = chunks.pop(0)
            while p + 4 <= end:
                t = d[p]
                sz = struct.unpack_from("<H", d, p + 1)[0]
                fl = d[p + 3]
                p += 4
                if flags & 4:
                    p += 2
                body = d[p:p + sz]
                if t == 0x10:
                    a, ln = struct.unpack_from("<QQ", body)
                    a += self.base
                    if d[a:a + 4] != b"OCHK":
                        raise ValueError("bad continuation chunk")
                    chunks.append((a + 4, a + ln - 4))
                elif t != 0:
                    out.append((t, fl, body))
                p += sz
        return out

    def links(self, rel=None):
        """name -> relative offset for the link messages of a group (default: the root group)."""
        out = {}
        for t, _fl, b in self.messages(self.root if rel is None else rel):
            if t != 0x06:
                continue
            if b[0] != 1:
                raise ValueError("link message version")
            fl = b[1]
            p = 2
            if fl & 0x08:
                ltype = b[p]
                p += 1
                if ltype != 0:
                    raise ValueError("only hard links")
            if fl & 0x04:
                p += 8
            if fl & 0x10:
                p += 1
            n = 1 << (fl & 3)
            ln = int.from_bytes(b[p:p + n], "little")
            p += n
            name = b[p:p + ln].decode()
            p += ln
            out[name] = struct.unpack_from("<Q", b, p)[0]
        return out

    # ---- datatypes -------------------------------------------------------------------------------------------------
    def _datatype(self, b, p=0):
        """Parse a datatype message at b[p:]; returns (descr, end).  descr: ('f8',) ('i', size, signed) ('ref',)
        ('compound', size, [(name, offset, descr)])."""
        cv = b[p]
        cls, ver = cv & 0x0F, cv >> 4
        bits = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16)
        size = struct.unpack_from("<I", b, p + 4)[0]
        p += 8
        if cls == 0:
            return ("i", size, bool(bits & 0x08)), p + 4
        if cls == 1:
            if size not in (4, 8):
                raise ValueError("float size")
            return ("f%d" % size,), p + 12
        if cls == 7:
            return ("ref",), p
        if cls == 6:
            if ver != 3:
                raise ValueError("compound datatype version %d" % ver)
            nmem = bits & 0xFFFF
            nb = 1 if size < 256 else 2 if size < 65536 else 4
            mem = []
            for _ in range(nmem):
                e = b.index(b"\x00", p)
                name = b[p:e].decode()
                p = e + 1
                moff = int.from_bytes(b[p:p + nb], "little")
                p += nb
                sub, p = self._datatype(b, p)
                mem.append((name, moff, sub))
            return ("compound", size, mem), p
        raise ValueError("datatype class %d not supported" % cls)

    def datatype_of(self, fl, body):
        if fl & 0x02:                       # shared: version, type, address of the committed datatype
            if body[0] != 3 or body[1] != 2:
                raise ValueError("shared message form")
            rel = struct.unpack_from("<Q", body, 2)[0]
            for t, f2, b2 in self.messages(rel):
                if t == 0x03:
                    return self.datatype_of(f2 & ~0x02, b2)
            raise ValueError("committed datatype without a datatype message")
        return self._datatype(body)[0]

    @staticmethod
    def _np_dtype(descr):
        k = descr[0]
        if k == "f8":
            return np.dtype("<f8")
        if k == "f4":
            return np.dtype("<f4")
        if k == "i":
            return np.dtype("<%s%d" % ("i" if descr[2] else "u", descr[1]))
        if k == "ref":
            return np.dtype("<u8")
        if k == "compound":
            return np.dtype({"names": [m[0] for m in descr[2]],
                             "formats": [JLD2File._np_dtype(m[2]) for m in descr[2]],
                             "offsets": [m[1] for m in descr[2]], "itemsize": descr[1]})
        raise ValueError(k)

    # ---- datasets --------------------------------------------------------------------------------------------------
    def read(self, rel):
        """The dataset at relative offset `rel` as (descr, numpy array); dimensions in Julia (column-major) order
        are reversed into numpy's, i.e. a Julia Vector of n elements comes back with shape (n,)."""
        dims, descr, raw = None, None, None
        for t, fl, b in self.messages(rel):
            if t == 0x01:
                if b[0] != 2:
                    raise ValueError("dataspace version")
                rank, typ = b[1], b[3]
                dims = () if typ == 0 else struct.unpack_from("<%dQ" % rank, b, 4)
            elif t == 0x03:
                descr = self.datatype_of(fl, b)
            elif t == 0x08:
                if b[0] != 4:
                    raise ValueError("layout version %d" % b[0])
                if b[1] == 0:
                    n = struct.unpack_from("<H", b, 2)[0]
                    raw = b[4:4 + n]
                elif b[1] == 1:
                    a, n = struct.unpack_from("<QQ", b, 2)
                    raw = self.d[self.base + a:self.base + a + n]
                else:
                    raise ValueError("chunked layout not supported")
            elif t == 0x0B:
                raise ValueError("filtered datasets not supported")
        if descr is None or raw is None or dims is None:
            raise ValueError("object at %d is not a dataset" % rel)
        arr = np.frombuffer(raw, dtype=self._np_dtype(descr), count=int(np.prod(dims, dtype=np.int64)))
        return descr, arr.reshape(dims)


def _flat_f8(arr):
    """A compound of equal float64 leaves (SVector{3}, Quaternion, ...) as an (n, k) float64 array."""
    k = arr.dtype.itemsize // 8
    return np.frombuffer(arr.tobytes(), dtype="<f8").reshape(arr.shape + (k,))


def read_storages(path):
    """`storages::Vector{Storage}` of the system-identification datasets -> dict of float64 arrays
    [n_storages, n_bodies, n_steps, width] for the fields of src/simulation/storage.jl:1-48
    (x 3, q 4 (s, v1, v2, v3), v 3, ω 3, px 3, pq 3, vl 3, ωl 3)."""
    f = JLD2File(path)
    top = f.links()
    _d, refs = f.read(top["storages"])
    fields = None
    out = {}
    for r in refs.ravel():
        descr, st = f.read(int(r))
        if descr[0] != "compound":
            raise ValueError("Storage is expected to be a struct")
        names = [m[0] for m in descr[2]]
        fields = fields or names
        for name in names:
            _d2, bodies = f.read(int(st[name].ravel()[0]))         # Vector{Vector{SVector}}: one ref per body
            rows = [_flat_f8(f.read(int(b))[1]) for b in bodies.ravel()]
            out.setdefault(name, []).append(np.stack(rows))
    return {k: np.stack(v) for k, v in out.items()}, f.header


if __name__ == "__main__":
    data, header = read_storages(sys.argv[1])
    print(header)
    for k, v in data.items():
        print("%-3s %s" % (k, v.shape))
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], **data)
