"""Re-braided TLAS (trx_set_build_rebraid, trx_scene_set_instance_entry_nodes): a BLAS whose box is large against the
scene is referenced through the subtrees under its root, each TLAS primitive carrying the node its walk starts at.  The
reference has no such thing (its TLAS primitives are whole BLASes, src/cwbvh.rs:108-137), so these tests pin the
extension to the same evidence as everything else: the structural validator, the brute-force query and, on the GPU, the
oracle."""
import numpy as np
import pytest

from helpers import assert_hits_equal, bits, random_rays


@pytest.fixture()
def rebraid(trx):
    lib = trx.load()
    yield lib
    lib.trx_set_build_rebraid(1.0 / 4096.0)   # the default


def build(trx, lib, name, n, frac):
    verts, counts = trx.gen_scene(name, n, 1)
    assert lib.trx_set_build_rebraid(frac) == 0
    return trx.flat_build(verts, counts, use_tlas=True), counts


@pytest.mark.parametrize("name,n", [("kitchen", 20000), ("san_miguel", 120000), ("cornell", 0)])
def test_rebraided_scene_is_valid_and_matches_brute_force(trx, orc, rebraid, name, n):
    flat, counts = build(trx, rebraid, name, n, 1.0 / 4096.0)
    plain, _ = build(trx, rebraid, name, n, 0.0)
    assert plain.instance_entry is None                                   # the reference's layout: whole BLASes only
    assert flat.instance_entry is not None and flat.instance_entry.size == flat.instance_offsets.size
    assert flat.instance_offsets.size > plain.instance_offsets.size       # something was opened
    assert (flat.instance_entry != 0).any()
    # the BLAS part of the node buffer and the triangles do not change: only the TLAS and the instance tables do
    assert np.array_equal(flat.nodes[: flat.tlas_start], plain.nodes[: plain.tlas_start]) and flat.tlas_start == plain.tlas_start
    assert np.array_equal(flat.tri_verts, plain.tri_verts)
    # every primitive names a BLAS start and an entry inside that BLAS; instance_source is that BLAS
    starts = np.unique(plain.instance_offsets)
    assert np.isin(flat.instance_offsets, starts).all()
    ends = np.append(starts[1:], flat.tlas_start)
    seg_end = ends[np.searchsorted(starts, flat.instance_offsets)]
    assert (flat.instance_offsets.astype(np.int64) + flat.instance_entry < seg_end).all()
    assert (np.searchsorted(starts, flat.instance_offsets) == flat.instance_source).all()
    osc = orc.Scene.from_flat(flat)
    rc, msg = osc.validate()
    assert rc == 0, msg                                                   # boxes contain subtrees, every triangle reached once
    rays = random_rays(trx, flat, 6000, 21)
    for sem in (0, 3):
        hits, st = osc.trace_rays(rays, sem=sem)
        assert st.overflow == 0
        bf = osc.brute_rays(rays, sem=sem)
        same_t = bits(hits["t"]) == bits(bf["t"])
        assert same_t.mean() > 0.9995                                     # the slab test's rounding cases
        ref, _ = orc.Scene.from_flat(plain).trace_rays(rays, sem=sem)
        assert (bits(hits["t"]) == bits(ref["t"])).mean() > 0.9995        # and the plain TLAS finds the same hits


def test_entry_node_setter_validates(trx, rebraid, has_gpu):
    lib = trx.load()
    assert lib.trx_scene_set_instance_entry_nodes(None, None, 0) == -1 and b"null scene" in lib.trx_last_error()
    if not has_gpu:
        return
    flat, _ = build(trx, rebraid, "kitchen", 6000, 1.0 / 4096.0)
    sc = trx.Scene(flat)
    import ctypes as C
    bad = flat.instance_entry.copy()
    bad[0] = 0x7fffffff
    rc = lib.trx_scene_set_instance_entry_nodes(sc.handle, bad.ctypes.data_as(C.c_void_p), bad.size)
    assert rc != 0 and b"leaves its BLAS" in lib.trx_last_error()
    assert lib.trx_scene_set_instance_entry_nodes(sc.handle, bad.ctypes.data_as(C.c_void_p), bad.size - 1) != 0
    sc.close()


@pytest.mark.gpu
def test_gpu_walks_rebraided_scenes_like_the_oracle(trx, orc, rebraid):
    """Primary + AO frames and explicit rays (with instance ids) over a re-braided TLAS: bit-exact against the oracle,
    and the same hits as the plain TLAS of the same scene."""
    w, h = 320, 180
    for name, n in (("san_miguel", 400000), ("kitchen", 30000)):
        flat, _ = build(trx, rebraid, name, n, 1.0 / 4096.0)
        eye, look, fov = trx.scene_camera(name)
        view = trx.view_from_camera(eye, look, fov, w, h)
        sc = trx.Scene(flat)
        osc = orc.Scene.from_flat(flat)
        ov = orc.view_from_bytes(view)
        for sem in (0, 3):
            prim, ao, _ = sc.trace_primary_ao(view, w, h, sem=sem, frame=1, ao_eps=0.01)
            want, st = osc.trace_primary(ov, w, h, sem=sem)
            assert_hits_equal(prim, want, "%s rebraided primary sem %d" % (name, sem))
            want_ao, _ = osc.trace_ao(ov, w, h, want, sem=sem, frame=1, ao_eps=0.01)
            assert_hits_equal(ao, want_ao, "%s rebraided AO sem %d" % (name, sem))
            cst = sc.count_primary(view, w, h, sem=sem)
            assert (cst.n_node, cst.n_tri) == (st.n_node, st.n_tri)
        rays = random_rays(trx, flat, 20000, 4)
        hits = sc.trace_rays(rays, sem=3)[0]
        want, inst_want, _ = osc.trace_rays_inst(rays, sem=3)
        assert_hits_equal(hits, want, "%s rebraided rays" % name)
        sc.close()
