"""Houdini classic ``.bgeo`` (version 5) particle files -- the on-disk format the reference's drivers exchange through
partio (test_smokegun.py:39-76 read ``id, position, density``; test_smokegun_resim.py:295-319 write ``id, position,
density, Cd, radius``; test_dambreak2d.py:109-124, test_chocolate.py:110-125 write ``position, Cd / radius``).

partio (github.com/wdas/partio; no version is pinned by the reference: its drivers import a local build,
``sys.path.append('E:/partio/build/py/Release')``, test_smokegun.py:13) is not installed in this image.  This module
restates the published layout of its BGEO reader / writer (src/lib/io/BGEO.cpp) for point attributes, which is all the
drivers use, and offers the handful of partio calls they make, so that their I/O blocks run unchanged with
``import io_bgeo as partio``:

    big-endian throughout
    int32  magic 'Bgeo' | char 'V' | int32 version = 5
    int32  nPoints, nPrims, nPointGroups, nPrimGroups, nPointAttrib, nVertexAttrib, nPrimAttrib, nAttrib
    per point attribute (``position`` is implicit and NOT listed):
        uint16 len, name | uint16 size | int32 houdiniType (0 float, 1 int, 5 vector; 4 = indexed string) |
        size x int32 default values            (type 4: int32 count, then count x (uint16 len, chars))
    per point:  float32 x, y, z, w (w = 1)  then the attributes in order, size x 4 bytes each
    [writer] one particle primitive referencing every point (prim attribute ``generator`` = "papi"), then 0x00 0xff

The reader stops after the point data (primitives, groups and detail attributes are not needed here).  Host-side I/O
only; nothing here is on the GPU path.  Round-trip and byte-layout tests: tests/test_io_cpu.py.
"""
from __future__ import annotations

import struct

import numpy as np

INT, FLOAT, VECTOR, INDEXEDSTR = 1, 0, 5, 4           # houdini type codes double as the partio type constants here
_MAGIC = ((((ord("B") << 8) | ord("g")) << 8 | ord("e")) << 8) | ord("o")


class Attribute(object):
    def __init__(self, name, type_, count):
        self.name, self.type, self.count = name, type_, int(count)


class ParticleSet(object):
    """the subset of partio's ParticlesDataMutable the reference drivers touch"""

    def __init__(self):
        self._attrs = []           # in file order; 'position' is kept first
        self._data = {}            # name -> list of tuples while building / ndarray [N,count] once read
        self._n = 0

    # ---- partio surface ----------------------------------------------------------------------------------------------
    def addAttribute(self, name, type_, count):
        a = Attribute(name, type_, count)
        self._attrs.append(a)
        dt = np.int32 if type_ == INT else np.float32
        self._data[name] = np.zeros((self._n, a.count), dt)
        return a

    def attributeInfo(self, name):
        for a in self._attrs:
            if a.name == name:
                return a
        return None

    def numParticles(self):
        return self._n

    def numAttributes(self):
        return len(self._attrs)

    def addParticle(self):
        return self.addParticles(1)

    def addParticles(self, count):
        """Capacity grows geometrically (amortised doubling): the drivers' per-particle loop -- ``addParticle`` + ``set``
        per particle, test_smokegun_resim.py:295-319 -- is O(N), not O(N^2).  ``_n`` counts the particles, the arrays
        may be longer; every reader goes through ``array`` (trimmed view)."""
        first = self._n
        self._n += int(count)
        for a in self._attrs:
            cur = self._data[a.name]
            if cur.shape[0] < self._n:
                grown = np.zeros((max(self._n, 2 * cur.shape[0], 16), a.count), cur.dtype)
                grown[:first] = cur[:first]
                self._data[a.name] = grown
        return first

    def set(self, attr, index, value):
        if not 0 <= index < self._n:
            raise IndexError("particle %d of %d" % (index, self._n))
        v = np.asarray(value)
        if attr.type == INT and v.dtype.kind == "f" and not np.all(v == np.floor(v)):
            raise ValueError("attribute %r is INT: refusing to truncate %r" % (attr.name, value))
        self._data[attr.name][index] = v

    def get(self, attr, index):
        if not 0 <= index < self._n:
            raise IndexError("particle %d of %d" % (index, self._n))
        return tuple(self._data[attr.name][index].tolist())

    # ---- vectorised access (build extension) -------------------------------------------------------------------------
    def array(self, name):
        return self._data[name][:self._n]

    def set_array(self, name, values):
        a = self.attributeInfo(name)
        v = np.asarray(values).reshape(self._n, a.count)
        self._data[name] = v.astype(np.int32 if a.type == INT else np.float32)


def create():
    return ParticleSet()


def from_arrays(arrays, types=None):
    """{'position': [N,2|3], 'id': [N], 'density': [N,k], ...} -> ParticleSet (ints -> INT, 'position' -> VECTOR)"""
    n = len(next(iter(arrays.values())))
    pt = ParticleSet()
    pt.addParticles(n)
    for name, v in arrays.items():
        v = np.asarray(v)
        v = v.reshape(n, -1)
        t = (types or {}).get(name)
        if t is None:
            t = INT if np.issubdtype(v.dtype, np.integer) else (VECTOR if name == "position" else FLOAT)
        pt.addAttribute(name, t, v.shape[1])
        pt.set_array(name, v)
    return pt


def _rd(f, fmt):
    size = struct.calcsize(fmt)
    b = f.read(size)
    if len(b) != size:
        raise ValueError("truncated .bgeo file")
    return struct.unpack(fmt, b)


def read(path):
    """partio.read for .bgeo: header + point attributes + point data"""
    with open(path, "rb") as f:
        (magic,) = _rd(f, ">i")
        (vchar,) = _rd(f, ">c")
        (version,) = _rd(f, ">i")
        if magic != _MAGIC or vchar != b"V":
            raise ValueError("%s: not a classic .bgeo file (magic %08x)" % (path, magic & 0xffffffff))
        if version != 5:
            raise ValueError("%s: .bgeo version %d (only 5 is supported, as in partio)" % (path, version))
        n_points, n_prims, n_pgroups, n_prgroups, n_pattr, n_vattr, n_prattr, n_attr = _rd(f, ">8i")
        defs = []
        for _ in range(n_pattr):
            (ln,) = _rd(f, ">H")
            name = f.read(ln).decode("ascii")
            size, htype = _rd(f, ">Hi")
            if htype in (FLOAT, INT, VECTOR):
                f.read(4 * size)                                    # default values
            elif htype == INDEXEDSTR:
                (cnt,) = _rd(f, ">i")
                for _ in range(cnt):
                    (sl,) = _rd(f, ">H")
                    f.read(sl)
            else:
                raise ValueError("%s: attribute %r has unsupported houdini type %d" % (path, name, htype))
            defs.append((name, htype, size))
        words = 4 + sum(sz for _, _, sz in defs)
        raw = np.frombuffer(f.read(4 * words * n_points), dtype=">u4")
        if raw.size != words * n_points:
            raise ValueError("%s: truncated point data" % path)
        raw = raw.reshape(n_points, words)
    pt = ParticleSet()
    pt._n = n_points
    pt._attrs.append(Attribute("position", VECTOR, 3))
    pt._data["position"] = raw[:, :3].astype(np.uint32).view(np.float32).reshape(n_points, 3).copy()
    col = 4
    for name, htype, size in defs:
        block = np.ascontiguousarray(raw[:, col:col + size].astype(np.uint32))
        if htype in (INT, INDEXEDSTR):
            val = block.view(np.int32)
        else:
            val = block.view(np.float32)
        pt._attrs.append(Attribute(name, htype, size))
        pt._data[name] = val.reshape(n_points, size).copy()
        col += size
    return pt


def write(path, pt):
    """partio.write for .bgeo: every attribute except 'position' is listed; positions shorter than 3 are zero-padded
    (test_dambreak2d.py:112 declares a 2-component position)"""
    n = pt.numParticles()
    attrs = [a for a in pt._attrs if a.name != "position"]
    pos = np.zeros((n, 4), np.float32)
    if pt.attributeInfo("position") is not None:
        p = np.asarray(pt.array("position"), np.float32).reshape(n, -1)
        pos[:, :min(p.shape[1], 3)] = p[:, :3]
    pos[:, 3] = 1.0
    with open(path, "wb") as f:
        f.write(struct.pack(">icI", _MAGIC, b"V", 5))
        f.write(struct.pack(">8i", n, 1, 0, 0, len(attrs), 0, 1, 0))
        cols = [pos.astype(">f4").view(">u4")]
        for a in attrs:
            nm = a.name.encode("ascii")
            f.write(struct.pack(">H", len(nm)) + nm)
            f.write(struct.pack(">Hi", a.count, a.type))
            f.write(struct.pack(">%di" % a.count, *([0] * a.count)))
            v = np.asarray(pt.array(a.name)).reshape(n, a.count)
            cols.append(v.astype(">i4" if a.type == INT else ">f4").view(">u4"))
        f.write(np.ascontiguousarray(np.concatenate(cols, axis=1)).astype(">u4").tobytes())   # (concatenate -> native order)
        # primitive attribute 'generator' (indexed string, one entry "papi"), then one particle primitive over all points
        gen = b"generator"
        f.write(struct.pack(">H", len(gen)) + gen + struct.pack(">Hi", 1, INDEXEDSTR))
        f.write(struct.pack(">i", 1) + struct.pack(">H", 4) + b"papi")
        f.write(struct.pack(">Ii", 0x8000, n))
        idx = np.arange(n)
        f.write(idx.astype(">i4").tobytes() if n > (1 << 16) else idx.astype(">u2").tobytes())
        f.write(struct.pack(">i", 0))
        f.write(b"\x00\xff")
