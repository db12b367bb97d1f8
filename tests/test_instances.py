"""Instance transforms in the TLAS (the reference's own TODOs, query_tlas.hlsl:409,433,484, and
Traversable::get_instance_transform): CPU tests of the oracle's restatement and of the host-side assembly."""
import os

import numpy as np
import pytest

from helpers import F32_MAX, aimed_rays, assert_hits_equal, bits, instanced_scene, random_rays, w2o_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_identity_transforms_reproduce_the_untransformed_goldens(orc):
    """With every instance at identity the transformed walk must be the reference's walk, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "cornell_tlas_48.npz"))
    w, h = int(g["width"]), int(g["height"])
    view = orc.view_from_bytes(g["view"].tobytes())
    n = g["instance_offsets"].size
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32), (n, 1))
    osc = orc.Scene(g["nodes"], g["tri_verts"], g["instance_offsets"], int(g["tlas_start"]), instance_w2o=ident)
    for sem in (0, 3):
        prim, inst, st = osc.trace_primary_inst(view, w, h, sem=sem)
        assert_hits_equal(prim, g["orc_primary_sem%d" % sem], "identity transforms, sem %d" % sem)
        assert [st.n_node, st.n_tri, st.n_hits, st.max_stack] == list(g["orc_counts_sem%d" % sem])
        ao, _, _ = osc.trace_ao_inst(view, w, h, prim, inst, sem=sem, frame=2, ao_eps=0.01)
        assert_hits_equal(ao, g["orc_ao_sem%d" % sem], "identity transforms, AO, sem %d" % sem)
        hit = prim["prim"] != 0xFFFFFFFF
        assert (inst[hit] < n).all() and (inst[~hit] == 0xFFFFFFFF).all()


def test_instanced_build_layout(trx):
    flat, o2w, world, first, blas_of = instanced_scene(trx)
    n = flat.instance_offsets.size
    assert n == 10 and flat.instance_source.size == n and sorted(flat.instance_source.tolist()) == list(range(n))
    assert len(set(flat.instance_offsets.tolist())) == 3          # three BLASes shared by ten instances
    assert flat.instance_transforms.shape == (n, 16)
    assert flat.blas_tri_start.size == 4 and world.shape[0] == first[-1]
    with pytest.raises(trx.TrxError):
        trx.flat_build_instanced(np.zeros((1, 9), np.float32), [1], [3], None)   # object out of range


def test_transformed_instances_match_brute_force_on_world_space_geometry(trx, orc):
    """Rays against transformed instances (ray taken to object space, t kept in world units) must find what a
    brute-force query over the world-space copies of the triangles finds: same instance and triangle, t within the
    rounding of the two transform directions."""
    flat, o2w, world, first, blas_of = instanced_scene(trx)
    w2o = np.stack([w2o_rows(m) for m in o2w])
    osc = orc.Scene(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start, instance_w2o=w2o)
    wflat = type("W", (), {"tri_verts": world})
    rays = np.concatenate([random_rays(trx, wflat, 2000, 11), aimed_rays(trx, world, 4000, 12)])
    bts = flat.blas_tri_start
    for sem in (0, 3):
        hits, inst, st = osc.trace_rays_inst(rays, sem=sem)
        assert st.overflow == 0
        bf = osc.brute_rays_over(world, rays, sem=sem)
        hit = hits["prim"] != 0xFFFFFFFF
        assert (hit == (bf["prim"] != 0xFFFFFFFF)).mean() > 0.999          # razor-edge rays may differ
        both = hit & (bf["prim"] != 0xFFFFFFFF)
        assert both.sum() > 1000
        rel = np.abs(hits["t"][both] - bf["t"][both]) / np.maximum(bf["t"][both], 1e-6)
        # grazing rays amplify the last-ulp differences of the two transform directions: bound the bulk and the tail
        assert np.quantile(rel, 0.995) < 1e-4 and rel.max() < 5e-3 and np.median(rel) < 2e-6
        # (instance, local triangle) -> index into the world triangle list
        b = np.array([blas_of[int(o)] for o in flat.instance_offsets[inst[both]]])
        widx = first[inst[both]] + (hits["prim"][both] - bts[b])
        assert (widx == bf["prim"][both]).mean() > 0.995                   # the rest are near-ties between neighbours


def test_identity_instance_of_a_transformed_scene_is_exact(trx, orc):
    """Instance 0 of the test scene sits at identity: rays that only meet that instance give bit-identical t with
    and without the transform table."""
    flat, o2w, world, first, blas_of = instanced_scene(trx)
    w2o = np.stack([w2o_rows(m) for m in o2w])
    k0 = int(np.flatnonzero(flat.instance_source == 0)[0])
    assert np.array_equal(w2o[k0], np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32))
    osc = orc.Scene(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start, instance_w2o=w2o)
    rays = aimed_rays(trx, world[first[k0]:first[k0 + 1]], 3000, 5)
    hits, inst, _ = osc.trace_rays_inst(rays, sem=3)
    only0 = inst == k0
    bf = osc.brute_rays_over(world[first[k0]:first[k0 + 1]], rays, sem=3)
    assert only0.sum() > 200
    assert (bits(hits["t"][only0]) == bits(bf["t"][only0])).all()
