"""Instance transforms on the GPU: the HIP path through the C-ABI against the oracle, bit for bit (same
world-to-object rows on both sides, same operation order), plus the Traversable surface that goes with them."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import ALL_SEMS, aimed_rays, assert_hits_equal, instanced_scene, random_rays

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scene_and_oracle(trx, orc, **kw):
    flat, o2w, world, first, blas_of = instanced_scene(trx, **kw)
    sc = trx.Scene(flat)                     # uploads flat.instance_transforms
    w2o = sc.instance_world_to_object()      # the rows the kernels use: handed to the oracle as data
    osc = orc.Scene(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start, instance_w2o=w2o)
    return flat, sc, osc, world


def test_transformed_instances_explicit_rays_all_semantics(trx, orc):
    flat, sc, osc, world = _scene_and_oracle(trx, orc)
    wflat = type("W", (), {"tri_verts": world})
    rays = np.concatenate([random_rays(trx, wflat, 3000, 21), aimed_rays(trx, world, 9000, 22)])
    for sem in ALL_SEMS:
        got, ginst, _ = sc.trace_rays_inst(rays, sem=sem)
        want, winst, st = osc.trace_rays_inst(rays, sem=sem)
        assert st.n_hits > 4000
        assert_hits_equal(got, want, "instanced rays, sem %d" % sem)
        assert (ginst == winst).all()
    # the library's inverse agrees with a float64 inverse to rounding
    from helpers import w2o_rows
    ref = np.stack([w2o_rows(m) for m in flat.instance_transforms])
    assert np.allclose(sc.instance_world_to_object(), ref, rtol=1e-6, atol=1e-6)
    sc.close()


def test_transformed_instances_primary_and_ao_frames(trx, orc):
    flat, sc, osc, world = _scene_and_oracle(trx, orc, n_instances=14, tris_per_object=0, kind="cornell", spread=1.0)
    w, h = 200, 120
    lo, hi = world.reshape(-1, 3).min(0), world.reshape(-1, 3).max(0)
    eye = (hi + 0.1 * (hi - lo)).tolist()
    view = trx.view_from_camera(eye, (0.5 * (lo + hi)).tolist(), 80.0, w, h)
    ov = orc.view_from_bytes(bytes(view))
    for sem in (0, 3):
        gp, gpi, gao, gaoi, ms = sc.trace_primary_ao_inst(view, w, h, sem=sem, frame=3, ao_eps=0.01)
        wp, wpi, st = osc.trace_primary_inst(ov, w, h, sem=sem)
        wao, waoi, _ = osc.trace_ao_inst(ov, w, h, wp, wpi, sem=sem, frame=3, ao_eps=0.01)
        assert st.n_hits > 0.1 * w * h and ms > 0
        assert_hits_equal(gp, wp, "instanced primary, sem %d" % sem)
        assert (gpi == wpi).all()
        assert_hits_equal(gao, wao, "instanced AO, sem %d" % sem)
        assert (gaoi == waoi).all()
    # the plain AO entry point cannot orient the normals of transformed instances: it must say so, not guess
    import torch
    d_p = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    d_a = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, d_p.data_ptr(), sem=3)
    with pytest.raises(trx.TrxError) as e:
        sc.trace_ao_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=3)
    assert "instance" in str(e.value)
    torch.cuda.synchronize()
    # the frame entry point (one dispatch in the reference; two launches here for a two-level scene) and the AO batch carry
    # the instance ids themselves - same records and ids as above; without the id buffer the frame refuses as well
    d_pi = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    d_ai = torch.zeros(2 * w * h, dtype=torch.int32, device="cuda")
    d_a2 = torch.zeros(2 * w * h, dtype=torch.int64, device="cuda")
    with pytest.raises(trx.TrxError) as e:
        sc.trace_frame_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=3, frame=3, ao_eps=0.01)
    assert "instance" in str(e.value)
    sc.trace_frame_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=3, frame=3, ao_eps=0.01, d_primary_inst=d_pi.data_ptr(),
                       d_ao_inst=d_ai.data_ptr())
    sc.trace_ao_batch_dev(view, w, h, d_p.data_ptr(), d_a2.data_ptr(), w * h, 2, sem=3, frame0=3, ao_eps=0.01,
                          d_primary_inst=d_pi.data_ptr(), d_ao_inst=d_ai.data_ptr())
    torch.cuda.synchronize()
    sc.check()
    from tray_racing_amd import dist as D
    assert_hits_equal(D.int64_to_hits(d_p), wp, "instanced frame, primary")
    assert_hits_equal(D.int64_to_hits(d_a), wao, "instanced frame, AO")
    assert_hits_equal(D.int64_to_hits(d_a2[: w * h]), wao, "instanced AO batch, frame 0")
    assert (d_pi.cpu().numpy().view(np.uint32) == wpi).all() and (d_ai[: w * h].cpu().numpy().view(np.uint32) == waoi).all()
    wao4, waoi4, _ = osc.trace_ao_inst(ov, w, h, wp, wpi, sem=3, frame=4, ao_eps=0.01)
    assert_hits_equal(D.int64_to_hits(d_a2[w * h:]), wao4, "instanced AO batch, frame 1")
    assert (d_ai[w * h:].cpu().numpy().view(np.uint32) == waoi4).all()
    sc.close()


def test_identity_transforms_reproduce_the_goldens_on_the_gpu(trx):
    from test_gpu_parity import GoldenFlat, load_view
    g = np.load(os.path.join(GOLDEN, "cornell_tlas_48.npz"))
    flat = GoldenFlat(trx, g).flat
    sc = trx.Scene(flat)
    n = flat.instance_offsets.size
    sc.set_instance_transforms(np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1)))
    w, h = int(g["width"]), int(g["height"])
    view = load_view(trx, g["view"])
    for sem in (0, 3):
        prim, pi, ao, ai, _ = sc.trace_primary_ao_inst(view, w, h, sem=sem, frame=2, ao_eps=0.01)
        assert_hits_equal(prim, g["orc_primary_sem%d" % sem], "identity transforms sem %d primary" % sem)
        assert_hits_equal(ao, g["orc_ao_sem%d" % sem], "identity transforms sem %d ao" % sem)
        hit = prim["prim"] != 0xFFFFFFFF
        assert (pi[hit] < n).all() and (pi[~hit] == 0xFFFFFFFF).all()
    sc.set_instance_transforms(None)        # and back to the reference's behaviour
    prim, ao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=2, ao_eps=0.01)
    assert_hits_equal(prim, g["orc_primary_sem3"], "transforms cleared")
    sc.close()


def test_traversable_surface_with_instances(trx, orc):
    """Traversable::traverse / get_instance_transform over a transformed scene: RayHit.instance_id is the TLAS
    primitive, geometry_id the BLAS, primitive_id local to it; get_instance_transform returns what was set."""
    flat, sc, osc, world = _scene_and_oracle(trx, orc)
    rays = aimed_rays(trx, world, 40, 7)
    want, winst, _ = osc.trace_rays_inst(rays, sem=3)
    offs = sorted(set(int(x) for x in flat.instance_offsets))
    for i in range(rays.shape[0]):
        h = sc.traverse(rays["origin"][i], rays["direction"][i], sem=3)
        if want["prim"][i] == 0xFFFFFFFF:
            assert h.primitive_id == 0xFFFFFFFF and h.instance_id == 0xFFFFFFFF
            continue
        assert np.float32(h.t) == want["t"][i] and h.instance_id == winst[i]
        b = offs.index(int(flat.instance_offsets[winst[i]]))
        assert h.geometry_id == b and h.primitive_id == want["prim"][i] - flat.blas_tri_start[b]
    # the batch form answers every ray exactly as the single-ray form does, in one launch
    got, ms = sc.traverse_batch(rays, sem=3)
    assert ms > 0 and got.shape[0] == rays.shape[0]
    for i in range(rays.shape[0]):
        h = sc.traverse(rays["origin"][i], rays["direction"][i], sem=3)
        assert (got["primitive_id"][i], got["geometry_id"][i], got["instance_id"][i]) == (h.primitive_id, h.geometry_id, h.instance_id)
        assert got["t"][i : i + 1].view(np.uint32)[0] == np.float32(h.t).view(np.uint32)
    for k in (0, 3, 9):
        assert np.array_equal(sc.instance_transform(k), flat.instance_transforms[k])
    with pytest.raises(trx.TrxError):
        sc.instance_transform(10)
    with pytest.raises(trx.TrxError):   # not affine
        bad = np.tile(np.eye(4, dtype=np.float32).reshape(16), (10, 1))
        bad[2, 3] = 0.5
        sc.set_instance_transforms(bad)
    sc.close()
