"""On-disk formats beside the hot path: classic .bgeo particle files with the attributes the reference
drivers read and write through partio (id, position, density, Cd, radius)."""
import struct

import numpy as np
import pytest

from neural_flow_style_amd import io_bgeo as partio


def test_bgeo_round_trip_of_the_driver_attributes(tmp_path):
    rng = np.random.RandomState(0)
    n = 70000                                                        # > 65536: the primitive lists int32 indices
    arrays = {"id": rng.permutation(n).astype(np.int32), "position": rng.rand(n, 3).astype(np.float32) * 200,
              "density": rng.rand(n, 2).astype(np.float32), "Cd": rng.rand(n, 3).astype(np.float32),
              "radius": np.full((n, 1), 0.5, np.float32)}
    path = str(tmp_path / "070.bgeo")
    partio.write(path, partio.from_arrays(arrays, types={"density": partio.VECTOR}))
    pt = partio.read(path)
    assert pt.numParticles() == n and pt.numAttributes() == 5
    for k, v in arrays.items():
        got = pt.array(k)
        assert got.dtype == (np.int32 if k == "id" else np.float32)
        assert np.array_equal(got.reshape(v.shape) if k != "id" else got[:, 0], v), k
    assert pt.attributeInfo("density").type == partio.VECTOR and pt.attributeInfo("density").count == 2
    assert pt.attributeInfo("nope") is None


def test_bgeo_byte_layout_matches_the_published_format(tmp_path):
    """hand-built file (big-endian header, one float and one int attribute, w = 1 positions) is read correctly, and a
    written file starts with exactly those bytes"""
    pos = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], np.float32)
    dens = np.array([0.25, 0.75], np.float32)
    ids = np.array([1, 0], np.int32)
    blob = struct.pack(">icI", 0x4267656F, b"V", 5) + struct.pack(">8i", 2, 0, 0, 0, 2, 0, 0, 0)
    blob += struct.pack(">H", 7) + b"density" + struct.pack(">Hi", 1, 0) + struct.pack(">i", 0)
    blob += struct.pack(">H", 2) + b"id" + struct.pack(">Hi", 1, 1) + struct.pack(">i", 0)
    for i in range(2):
        blob += struct.pack(">4f", pos[i, 0], pos[i, 1], pos[i, 2], 1.0) + struct.pack(">f", dens[i]) + struct.pack(">i", ids[i])
    path = tmp_path / "hand.bgeo"
    path.write_bytes(blob + b"\x00\xff")
    pt = partio.read(str(path))
    assert np.array_equal(pt.array("position"), pos) and np.array_equal(pt.array("density")[:, 0], dens)
    assert np.array_equal(pt.array("id")[:, 0], ids)
    out = tmp_path / "w.bgeo"
    partio.write(str(out), partio.from_arrays({"position": pos, "density": dens, "id": ids}))
    w = out.read_bytes()
    assert w[:9] == blob[:9]                                         # magic, 'V', version
    assert struct.unpack(">8i", w[9:41]) == (2, 1, 0, 0, 2, 0, 1, 0)  # one particle primitive, one prim attribute
    assert w[41:41 + len(blob) - 41] == blob[41:]                    # attribute table + point data identical
    assert w.endswith(b"\x00\xff")
    with pytest.raises(ValueError):
        (tmp_path / "bad.bgeo").write_bytes(b"nope" + blob[4:])
        partio.read(str(tmp_path / "bad.bgeo"))


def test_partio_calls_of_the_reference_drivers(tmp_path):
    """the write block of test_smokegun_resim.py:295-319 and the read block of test_smokegun.py:41-56, call for call"""
    p_ = np.array([[3.0, 10.0, 7.0], [1.0, 2.0, 3.0], [9.0, 8.0, 7.0]], np.float32)
    p_den = np.array([[0.5, 0.1], [0.2, 0.9], [0.7, 0.3]], np.float32)
    p_id = np.array([2, 0, 1])
    pt = partio.create()
    pid = pt.addAttribute("id", partio.INT, 1)
    position = pt.addAttribute("position", partio.VECTOR, 3)
    density = pt.addAttribute("density", partio.VECTOR, p_den.shape[1])
    color = pt.addAttribute("Cd", partio.FLOAT, 3)
    radius = pt.addAttribute("radius", partio.FLOAT, 1)
    for i in range(p_.shape[0]):
        pt_ = pt.addParticle()
        pt.set(pid, pt_, (int(p_id[i]),))
        pt.set(position, pt_, tuple(p_[i].astype(float)))
        pt.set(density, pt_, tuple(p_den[i].astype(float)))
        pt.set(color, pt_, tuple(np.array([p_den[i, 0]] * 3, dtype=float)))
        pt.set(radius, pt_, (0.5,))
    path = str(tmp_path / "000.bgeo")
    partio.write(path, pt)
    rd = partio.read(path)
    a_id, a_pos, a_den = rd.attributeInfo("id"), rd.attributeInfo("position"), rd.attributeInfo("density")
    got_p, got_r = [], []
    for j in range(rd.numParticles()):
        j_id = rd.get(a_id, j)[0]
        got_p.append(rd.get(a_pos, j_id))
        got_r.append(rd.get(a_den, j_id))
    assert np.allclose(got_p, p_[p_id]) and np.allclose(got_r, p_den[p_id])
    assert np.allclose(rd.array("Cd"), np.repeat(p_den[:, :1], 3, 1)) and np.allclose(rd.array("radius"), 0.5)


def test_per_particle_loop_is_linear_and_ints_are_not_truncated(tmp_path):
    """the reference drivers' unchanged per-particle loop (addParticle + set per particle,
    test_smokegun_resim.py:295-319) must not re-allocate every attribute on every call; INT attributes refuse
    fractional values instead of truncating them"""
    import time
    import io_bgeo as partio
    pt = partio.create()
    P = pt.addAttribute("position", partio.VECTOR, 3)
    I = pt.addAttribute("id", partio.INT, 1)
    Dn = pt.addAttribute("density", partio.FLOAT, 1)
    n = 60000
    t0 = time.time()
    for i in range(n):
        k = pt.addParticle()
        pt.set(P, k, (i * 1e-5, 0.5, 0.25))
        pt.set(I, k, [i])
        pt.set(Dn, k, (1.0 + i,))
    dt = time.time() - t0
    assert dt < 8.0, "per-particle loop took %.1f s for %d particles" % (dt, n)     # quadratic growth: minutes
    assert pt.numParticles() == n and pt.array("id").shape == (n, 1)
    assert pt.get(I, n - 1) == (n - 1,)
    with pytest.raises(ValueError):
        pt.set(I, 0, [0.5])
    with pytest.raises(IndexError):
        pt.set(I, n, [1])
    path = str(tmp_path / "loop.bgeo")
    partio.write(path, pt)
    back = partio.read(path)
    assert back.numParticles() == n
    np.testing.assert_array_equal(back.array("id")[:, 0], np.arange(n))
    np.testing.assert_allclose(back.array("density")[:, 0], 1.0 + np.arange(n, dtype=np.float32))
