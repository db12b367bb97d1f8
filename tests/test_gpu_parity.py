"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle.

Bar: bit-exact t (IEEE-754 binary32, same operation order on both sides) and
exact primitive indices.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from helpers import (ALL_SEMS, F32_MAX, assert_hits_equal, bits, deep_chain_scene, golden_inputs, make_scene, random_rays)

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def need_gpu(trx):
    lib = trx.load()  # raises if libtrx.so is missing: the HIP extension is mandatory here
    assert lib.trx_device_count() > 0, "no HIP device visible to libtrx.so"
    buf = C.create_string_buffer(64)
    lib.trx_device_name(0, buf, 64)
    assert buf.value.startswith(b"gfx950"), buf.value


class GoldenFlat:
    def __init__(self, trx, g):
        nodes, tri_verts, inst, tlas_start = golden_inputs(trx, g)
        self.flat = trx.FlatScene(nodes, tri_verts, inst, tlas_start, np.arange(tri_verts.shape[0]), [0, tri_verts.shape[0]])


def load_view(trx, raw):
    from tray_racing_amd import _lib
    v = _lib.View()
    C.memmove(C.byref(v), raw.tobytes(), C.sizeof(v))
    return v


# ---- golden fixtures -----------------------------------------------------------------------

@pytest.mark.parametrize("name", ["cornell_64", "cornell_tlas_48", "soup_52x44", "box14_tlas_48", "ref_cornell_box_64", "ref_box_tlas_48"])
def test_golden_images(trx, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sc = trx.Scene(GoldenFlat(trx, g).flat)
    w, h = int(g["width"]), int(g["height"])
    view = load_view(trx, g["view"])
    for sem in (0, 3):
        prim, ao, ms = sc.trace_primary_ao(view, w, h, sem=sem, frame=2, ao_eps=0.01)
        assert ms > 0
        assert_hits_equal(prim, g["orc_primary_sem%d" % sem], "%s sem %d primary" % (name, sem))
        assert_hits_equal(ao, g["orc_ao_sem%d" % sem], "%s sem %d ao" % (name, sem))
        st = sc.count_primary(view, w, h, sem=sem)
        assert [st.n_node, st.n_tri, st.n_hits, st.max_stack] == list(g["orc_counts_sem%d" % sem])
        assert st.n_rays == w * h and st.overflow == 0
        # against the BVH-independent brute force: t is bit-equal everywhere
        assert (bits(prim["t"]) == bits(g["bf_primary_sem%d" % sem]["t"])).all()
    sc.close()


@pytest.mark.parametrize("name,tris", [("hairball", 60000), ("bistro", 150000), ("cornell", 0)])
def test_thin_waves_a_handful_of_rays_eight_lanes_each(trx, orc, name, tris):
    """A dry wave that is down to eight rays or fewer gives every ray eight lanes (one child of the node each, a leaf's
    triangles eight at a time: kernels.hip, thin_walk).  Batches of 1 ... 100 rays run (almost) wholly that way - a
    64-ray wave goes thin once 56 rays are through - and must equal the oracle bit for bit under all eight semantics,
    closest hit and any hit, with zero direction components, ranged rays and exact ties in the mix; the switch off
    (variant bit 28) must give the same records."""
    flat, _view, osc, _ov = make_scene(trx, orc, name, tris, 64, 64)
    sc = trx.Scene(flat)
    lib = trx.load()
    try:
        for n in (1, 2, 3, 7, 8, 9, 13, 64, 65, 100):
            rays = random_rays(trx, flat, n, 1000 + n, zero_dirs=True)
            for sem in ALL_SEMS:
                want, _ = osc.trace_rays(rays, sem=sem)
                got, _ = sc.trace_rays(rays, sem=sem)
                assert_hits_equal(got, want, "%s %d rays sem %d" % (name, n, sem))
                if sem in (0, 3):
                    occ, _ = sc.trace_occluded(rays, sem=sem)
                    assert (occ.astype(bool) == np.isfinite(want["t"])).all()
                    lib.trx_set_kernel_variant(1 << 28)
                    off, _ = sc.trace_rays(rays, sem=sem)
                    lib.trx_set_kernel_variant(0)
                    assert_hits_equal(off, want, "%s %d rays sem %d, thin waves off" % (name, n, sem))
    finally:
        lib.trx_set_kernel_variant(0)
        sc.close()


def test_golden_ties_and_zero_directions(trx):
    g = np.load(os.path.join(GOLDEN, "ties_rays.npz"))
    sc = trx.Scene(GoldenFlat(trx, g).flat)
    for sem in (0, 3):
        hits, _ = sc.trace_rays(g["rays"], sem=sem)
        assert_hits_equal(hits, g["orc_rays_sem%d" % sem], "ties sem %d" % sem)
        assert (bits(hits["t"]) == bits(g["bf_rays_sem%d" % sem]["t"])).all()
    sc.close()


def test_golden_f16_tlas_rays(trx):
    """The reference's 24-byte f16 triangles through a TLAS, AO with the GPU epsilon (1e-4,
    rt_gpu_software.hlsl:115), and explicit rays with tmin / tmax ranges."""
    g = np.load(os.path.join(GOLDEN, "kitchen_tlas_f16_56x40.npz"))
    flat = GoldenFlat(trx, g).flat
    w, h = int(g["width"]), int(g["height"])
    view = load_view(trx, g["view"])
    sc16 = trx.Scene(flat, tri_format=trx.TRI_F16_24, tri_bytes=g["tri_f16"])
    sc = trx.Scene(flat)
    for sem in (0, 3):
        prim, ao, _ = sc16.trace_primary_ao(view, w, h, sem=sem, frame=9, ao_eps=0.0001)
        assert_hits_equal(prim, g["orc_f16_primary_sem%d" % sem], "f16 primary sem %d" % sem)
        assert_hits_equal(ao, g["orc_f16_ao_eps1e-4_sem%d" % sem], "f16 ao sem %d" % sem)
        got, _ = sc.trace_rays(g["rays"], sem=sem)
        assert_hits_equal(got, g["orc_rays_sem%d" % sem], "tlas rays sem %d" % sem)
    sc16.close()
    sc.close()


# ---- live comparison on seeded scenes, every semantics combination ------------------------------

@pytest.mark.parametrize("name,n,w,h,tlas", [("cornell", 0, 96, 64, False), ("kitchen", 20000, 120, 72, False),
                                              ("kitchen", 20000, 72, 48, True), ("bistro", 150000, 160, 90, False),
                                              ("hairball", 120000, 96, 96, False), ("san_miguel", 120000, 120, 68, True),
                                              ("demoscene", 60000, 64, 136, False), ("soup", 2500, 52, 44, False)])
def test_scenes_all_semantics(trx, orc, name, n, w, h, tlas):
    flat, view, osc, ov = make_scene(trx, orc, name, n, w, h, tlas=tlas)
    sc = trx.Scene(flat)
    rays = random_rays(trx, flat, 5000, 11)
    for sem in ALL_SEMS:
        gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=sem, frame=1, ao_eps=0.01)
        op, ost = osc.trace_primary(ov, w, h, sem=sem)
        oao, _ = osc.trace_ao(ov, w, h, op, sem=sem, frame=1, ao_eps=0.01)
        assert_hits_equal(gp, op, "%s sem %d primary" % (name, sem))
        assert_hits_equal(gao, oao, "%s sem %d ao" % (name, sem))
        gr, _ = sc.trace_rays(rays, sem=sem)
        orr, _ = osc.trace_rays(rays, sem=sem)
        assert_hits_equal(gr, orr, "%s sem %d rays" % (name, sem))
        st = sc.count_primary(view, w, h, sem=sem)
        assert (st.n_node, st.n_tri, st.n_hits, st.max_stack) == (ost.n_node, ost.n_tri, ost.n_hits, ost.max_stack)
    # GPU AO epsilon (rt_gpu_software.hlsl:115) and a different frame seed
    gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=0, frame=7, ao_eps=0.0001)
    oao, _ = osc.trace_ao(ov, w, h, gp, sem=0, frame=7, ao_eps=0.0001)
    assert_hits_equal(gao, oao, "%s ao eps 1e-4" % name)
    sc.close()


def test_triangle_formats(trx, orc):
    flat, view, osc, ov = make_scene(trx, orc, "kitchen", 12000, 96, 64)
    want, _ = osc.trace_primary(ov, 96, 64, sem=0)
    # f32 {v0,e1,e2} handed over directly
    edges = osc.tris.copy()
    sc = trx.Scene(flat, tri_format=trx.TRI_EDGES_36, tri_bytes=edges)
    assert_hits_equal(sc.trace_primary(view, 96, 64, sem=0)[0], want, "TRI_EDGES_36")
    sc.close()
    # the reference's 24-byte f16 triangles (RtCompressedTriangle): parity against the oracle
    # decoding the same bytes; against f32 geometry t moves by ~1e-3 relative (f16 edges)
    packed = trx.pack_tris_f16(flat.tri_verts)
    osc16 = orc.Scene(flat.nodes, tri_f16=packed)
    sc = trx.Scene(flat, tri_format=trx.TRI_F16_24, tri_bytes=packed)
    got, _ = sc.trace_primary(view, 96, 64, sem=0)
    assert_hits_equal(got, osc16.trace_primary(ov, 96, 64, sem=0)[0], "TRI_F16_24")
    both = np.isfinite(got["t"]) & np.isfinite(want["t"])
    assert both.mean() > 0.9 and np.median(np.abs(got["t"][both] - want["t"][both]) / want["t"][both]) < 5e-3
    sc.close()


# ---- shards, layouts, odd sizes ------------------------------------------------------------------

def test_frames_per_launch_match_separate_launches(trx, orc):
    """trx_trace_primary_batch_dev: several frames (different cameras) in one launch, both layouts and a
    strided output, must equal the oracle frame by frame; repeated so the tile-order feedback is live."""
    import torch
    from tray_racing_amd import dist as D
    w, h = 200, 120
    flat, _view, osc, _ov = make_scene(trx, orc, "bistro", 60000, w, h)
    eye, look, fov = trx.scene_camera("bistro")
    views = [trx.view_from_camera((eye[0] + 0.7 * f, eye[1] + 0.2 * f, eye[2]), look, fov, w, h) for f in range(5)]
    wants = [osc.trace_primary(orc.view_from_bytes(v), w, h, sem=3)[0] for v in views]
    sc = trx.Scene(flat)
    stride = w * h + 96
    out = torch.full((5 * stride,), -1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        sc.trace_primary_batch_dev(views, w, h, out.data_ptr(), stride, sem=3)
    sc.check()
    for f in range(5):
        assert_hits_equal(D.int64_to_hits(out[f * stride:f * stride + w * h]), wants[f], "image layout, frame %d" % f)
        assert (out[f * stride + w * h:(f + 1) * stride] == -1).all()
    # tile shards of 3 ranks, compact layout, gathered by hand
    world = 3
    fgs = [D.FrameGather(w, h, r, world, "cuda", batch=4) for r in range(world)]
    for r in range(world):
        for _ in range(2):
            sc.trace_primary_batch_dev(views[:4], w, h, fgs[r].slot(0, 4).data_ptr(), fgs[r].records, sem=3,
                                       shard=(r, world, 1))
    sc.check()
    n = 4 * fgs[0].records
    for r in range(world):   # what the in-place all-gather would deliver to rank 0
        fgs[0].flat[r * n:(r + 1) * n].copy_(fgs[r].flat[r * n:(r + 1) * n])
    frames = D.int64_to_hits(fgs[0].assemble(m=4)).reshape(4, w * h)
    for f in range(4):
        assert_hits_equal(frames[f], wants[f], "shard layout, frame %d" % f)
    with pytest.raises(trx.TrxError, match="n_frames"):
        sc.trace_primary_batch_dev(views + views, w, h, out.data_ptr(), stride, sem=3)
    with pytest.raises(trx.TrxError, match="frame_stride"):
        sc.trace_primary_batch_dev(views[:2], w, h, out.data_ptr(), 10, sem=3)
    sc.close()


def test_shards_and_layouts(trx, orc):
    import torch
    from tray_racing_amd import dist as D
    w, h = 100, 52
    flat, view, osc, ov = make_scene(trx, orc, "cornell", 0, w, h)
    want, _ = osc.trace_primary(ov, w, h, sem=3)
    sc = trx.Scene(flat)
    for world in (1, 2, 3, 8):
        img = torch.full((w * h,), -1, dtype=torch.int64, device="cuda")
        gathered = []
        for r in range(world):
            sc.trace_primary_dev(view, w, h, img.data_ptr(), sem=3, shard=(r, world, 0))
            fgr = D.FrameGather(w, h, r, world, "cuda")
            local = fgr.new_local()
            sc.trace_primary_dev(view, w, h, local.data_ptr(), sem=3, shard=(r, world, 1))
            torch.cuda.synchronize()
            assert local.numel() == D.max_shard_tiles(w, h, world) * 64
            gathered.append(local)
        sc.check()
        assert_hits_equal(D.int64_to_hits(img), want, "image layout, %d shards" % world)
        fgr.gathered.copy_(torch.stack(gathered))       # what the all-gather would deliver
        assert_hits_equal(D.int64_to_hits(fgr.assemble()), want, "shard layout, %d shards" % world)
    sc.close()


@pytest.mark.parametrize("name,tris,tlas,w,h", [("kitchen", 20000, False, 100, 52), ("cornell", 0, False, 33, 47),
                                               ("san_miguel", 60000, True, 120, 72), ("bistro", 400000, False, 256, 144)])
def test_ao_batch_is_n_separate_ao_passes(trx, orc, name, tris, tlas, w, h):
    """trx_trace_ao_batch_dev (BASELINE.json configs[3]'s "4 spp" = AO frames with seeds frame0 .. frame0 + n - 1 over
    one view and one primary hit buffer, as ONE launch): every frame of the batch equals the oracle's AO pass with that
    seed, bit for bit; n = 2, 3, 4, 8, image sizes whose tile count is not a multiple of eight (the queues are padded),
    a two-level scene, and a compact tile shard."""
    import torch
    from tray_racing_amd import dist as D
    flat, view, osc, ov = make_scene(trx, orc, name, tris, w, h, tlas=tlas)
    sc = trx.Scene(flat)
    op, _ = osc.trace_primary(ov, w, h, sem=3)
    d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
    for n, frame0 in ((2, 0), (3, 7), (4, 0), (8, 1021)):
        stride = w * h + 5                                           # frames need not be packed
        d_ao = torch.full((n * stride,), -1, dtype=torch.int64, device="cuda")
        sc.trace_ao_batch_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), stride, n, sem=3, frame0=frame0, ao_eps=0.01)
        torch.cuda.synchronize()
        sc.check()
        for f in range(n):
            want, _ = osc.trace_ao(ov, w, h, op, sem=3, frame=frame0 + f, ao_eps=0.01)
            assert_hits_equal(D.int64_to_hits(d_ao[f * stride: f * stride + w * h]), want, "%s ao batch of %d, frame %d" % (name, n, f))
            assert bool((d_ao[f * stride + w * h: (f + 1) * stride] == -1).all())    # nothing written between frames
    # a compact tile shard: rank 1 of 3 traces its tiles of the primary frame and of a 4-frame AO batch
    world, r = 3, 1
    fg = D.FrameGather(w, h, r, world, "cuda")
    rec = D.max_shard_tiles(w, h, world) * 64
    lp = fg.new_local()
    sc.trace_primary_dev(view, w, h, lp.data_ptr(), sem=3, shard=(r, world, 1))
    la = torch.full((4 * rec,), -1, dtype=torch.int64, device="cuda")
    sc.trace_ao_batch_dev(view, w, h, lp.data_ptr(), la.data_ptr(), rec, 4, sem=3, frame0=2, ao_eps=0.0001, shard=(r, world, 1))
    one = torch.full((rec,), -1, dtype=torch.int64, device="cuda")
    for f in range(4):
        one.fill_(-1)
        sc.trace_ao_dev(view, w, h, lp.data_ptr(), one.data_ptr(), sem=3, frame=2 + f, ao_eps=0.0001, shard=(r, world, 1))
        torch.cuda.synchronize()
        assert bool((la[f * rec:(f + 1) * rec] == one).all()), "shard layout, frame %d" % f
    sc.check()
    sc.close()


@pytest.mark.parametrize("name,tris,tlas,w,h", [("kitchen", 20000, False, 100, 52), ("cornell", 0, False, 33, 47),
                                               ("san_miguel", 60000, True, 120, 72), ("bistro", 400000, False, 256, 144),
                                               ("hairball", 100000, False, 160, 96)])
def test_one_launch_frame_is_the_two_pass_frame(trx, orc, name, tris, tlas, w, h):
    """trx_trace_frame_dev (the reference's single dispatch, rt_gpu_software.hlsl:47-144: a lane whose primary ray hits
    goes on as the pixel's AO ray): the primary and the AO records are the oracle's, bit for bit, under both semantics
    presets, for every refill / conversion threshold, in image layout and in a compact tile shard."""
    import torch
    from tray_racing_amd import dist as D
    flat, view, osc, ov = make_scene(trx, orc, name, tris, w, h, tlas=tlas)
    sc = trx.Scene(flat)
    lib = trx.load()
    d_p = torch.empty(w * h, dtype=torch.int64, device="cuda")
    d_a = torch.empty(w * h, dtype=torch.int64, device="cuda")
    try:
        for sem, frame, eps in ((3, 0, 0.01), (0, 5, 0.0001)):
            op, _ = osc.trace_primary(ov, w, h, sem=sem)
            oa, _ = osc.trace_ao(ov, w, h, op, sem=sem, frame=frame, ao_eps=eps)
            for variant in (0, 64, 1, 24 | (1 << 14), 48 | (3 << 14)):
                lib.trx_set_kernel_variant(variant)
                d_p.fill_(-1)
                d_a.fill_(-1)
                sc.trace_frame_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=sem, frame=frame, ao_eps=eps)
                torch.cuda.synchronize()
                sc.check()
                assert_hits_equal(D.int64_to_hits(d_p), op, "%s one-launch frame, primary (sem %d, variant 0x%x)" % (name, sem, variant))
                assert_hits_equal(D.int64_to_hits(d_a), oa, "%s one-launch frame, ao (sem %d, variant 0x%x)" % (name, sem, variant))
        lib.trx_set_kernel_variant(0)
        # a compact tile shard against the two-pass entry points on the same shard
        world, r = 3, 2
        rec = D.max_shard_tiles(w, h, world) * 64
        bufs = [torch.full((rec,), -1, dtype=torch.int64, device="cuda") for _ in range(4)]
        sc.trace_primary_dev(view, w, h, bufs[0].data_ptr(), sem=3, shard=(r, world, 1))
        sc.trace_ao_dev(view, w, h, bufs[0].data_ptr(), bufs[1].data_ptr(), sem=3, frame=3, ao_eps=0.01, shard=(r, world, 1))
        sc.trace_frame_dev(view, w, h, bufs[2].data_ptr(), bufs[3].data_ptr(), sem=3, frame=3, ao_eps=0.01, shard=(r, world, 1))
        torch.cuda.synchronize()
        sc.check()
        assert bool((bufs[0] == bufs[2]).all()) and bool((bufs[1] == bufs[3]).all())
    finally:
        lib.trx_set_kernel_variant(0)
        sc.close()


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (9, 8), (64, 1), (33, 47)])
def test_image_sizes_not_multiple_of_8(trx, orc, w, h):
    flat, view, osc, ov = make_scene(trx, orc, "cornell", 0, w, h)
    sc = trx.Scene(flat)
    gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=3)
    op, _ = osc.trace_primary(ov, w, h, sem=3)
    assert_hits_equal(gp, op, "%dx%d" % (w, h))
    assert_hits_equal(gao, osc.trace_ao(ov, w, h, op, sem=3)[0], "%dx%d ao" % (w, h))
    sc.close()


def test_counting_pass_into_internal_scratch_with_a_shard_layout(trx, orc):
    """trx_count_primary with d_hits == NULL writes into the scene's scratch buffer; in the compact shard layout
    that is whole tiles (local_tile * 64 + k), more records than w * h when the image ends mid-tile (9 x 9:
    4 tiles = 256 records for 81 pixels).  Counters must still equal the oracle's for every shard."""
    w, h = 9, 9
    flat, view, osc, ov = make_scene(trx, orc, "cornell", 0, w, h)
    sc = trx.Scene(flat)
    _, want = osc.trace_primary(ov, w, h, sem=3)
    for world in (1, 2, 3):
        n_node = n_tri = n_rays = 0
        for r in range(world):
            st = sc.count_primary(view, w, h, sem=3, shard=(r, world, 1))
            n_node, n_tri, n_rays = n_node + st.n_node, n_tri + st.n_tri, n_rays + st.n_rays
        assert (n_rays, n_node, n_tri) == (w * h, want.n_node, want.n_tri)
    sc.check()
    sc.close()


# ---- edge cases -------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 2, 3, 5])
def test_tiny_and_empty_scenes(trx, orc, n):
    verts = trx.gen_scene("soup", max(n, 1), 5)[0][:n]
    flat = trx.flat_build(verts)
    eye, look, fov = trx.scene_camera("soup")
    view = trx.view_from_camera(eye, look, fov, 40, 40)
    sc = trx.Scene(flat)
    got, _ = sc.trace_primary(view, 40, 40)
    if n == 0:
        assert np.isinf(got["t"]).all() and (got["prim"] == 0xFFFFFFFF).all()
    else:
        osc = orc.Scene.from_flat(flat)
        assert_hits_equal(got, osc.trace_primary(orc.view_from_bytes(view), 40, 40)[0], "%d tris" % n)
    sc.close()


def test_degenerate_triangles_and_ray_ranges(trx, orc):
    one = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float32)
    verts = np.concatenate([np.repeat(one, 30, axis=0), np.zeros((6, 9), np.float32), one + np.float32(2.0),
                            np.array([[0, 0, -1, 1, 0, -1, 2, 0, -1]], dtype=np.float32)])  # collinear
    flat = trx.flat_build(verts)
    osc = orc.Scene.from_flat(flat)
    rays = np.zeros(9, dtype=trx.RAY_DTYPE)
    rays["origin"] = [(0.25, 0.25, 5)] * 5 + [(2.25, 2.25, 5), (0.25, 0.25, -3), (0.5, 0, 5), (5, 5, 5)]
    rays["direction"] = [(0, 0, -1)] * 6 + [(0, 0, 1), (0, 0, -1), (0, 0, -1)]
    rays["tmin"] = [0, 5.0, 5.5, 0, 0, 0, 0, 0, 0]
    rays["tmax"] = [F32_MAX, F32_MAX, F32_MAX, 4.9, np.inf, F32_MAX, F32_MAX, F32_MAX, F32_MAX]
    sc = trx.Scene(flat)
    for sem in (0, 3):
        got, _ = sc.trace_rays(rays, sem=sem)
        assert_hits_equal(got, osc.trace_rays(rays, sem=sem)[0], "degenerate sem %d" % sem)
    assert got["t"][0] == 5.0 and got["t"][1] == 5.0 and np.isinf(got["t"][2]) and np.isinf(got["t"][3])
    assert np.isinf(got["t"][8]) and got["prim"][8] == 0xFFFFFFFF
    sc.close()


def test_any_hit_query_equals_closest_hit_occupancy(trx, orc):
    """trx_trace_occluded (intersects_bl_bvh, query.hlsl:440-445): stops at the first accepted triangle, and must
    say exactly what the closest-hit query says about hit / no hit — for ranged rays, every semantics, TLAS or not."""
    import torch
    for name, n, tlas in (("kitchen", 30000, False), ("bistro", 60000, True), ("hairball", 50000, False)):
        flat, _view, osc, _ov = make_scene(trx, orc, name, n, 32, 32, tlas=tlas)
        sc = trx.Scene(flat)
        rays = random_rays(trx, flat, 40000, 11)
        for sem in (0, 3, 5, 6):
            want = osc.trace_rays(rays, sem=sem)[0]["prim"] != trx.MISS_PRIM
            flags, ms = sc.trace_occluded(rays, sem=sem)
            assert ms > 0 and set(np.unique(flags)) <= {0, 1}
            assert (flags.astype(bool) == want).all(), (name, sem, int((flags.astype(bool) != want).sum()))
            assert 0.02 < want.mean() < 0.99
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
        d_flags = torch.full((rays.shape[0] + 8,), 7, dtype=torch.uint8, device="cuda")
        sc.trace_occluded_dev(d_rays.data_ptr(), rays.shape[0], d_flags.data_ptr(), sem=3)
        sc.check()
        assert (d_flags[:rays.shape[0]].cpu().numpy() == flags_for(sc, rays, 3)).all() and (d_flags[rays.shape[0]:] == 7).all()
        sc.close()


def flags_for(sc, rays, sem):
    return sc.trace_occluded(rays, sem=sem)[0]


def test_stack_spill_to_hbm_and_overflow_detection(trx, orc):
    """A hand-made tree drives the per-lane stack past the LDS part (12 entries) into the
    HBM spill; past 64 entries the kernel must flag the ray instead of corrupting memory."""
    rays = np.zeros(70, dtype=trx.RAY_DTYPE)
    rays["origin"] = (0.3, 0.3, 1)
    rays["direction"] = (0, 0, -1)
    rays["origin"][1::2] = (0.3, 0.3, -100)   # half the wave looks the other way: mixed stack depths
    rays["direction"][1::2] = (0, 0, 1)
    rays["tmax"] = F32_MAX
    for depth in (10, 13, 40, 64):
        nodes, tris = deep_chain_scene(depth)
        flat = trx.FlatScene(nodes, tris, [], 0, np.arange(depth), [0, depth])
        osc = orc.Scene(nodes, tris)
        want, ost = osc.trace_rays(rays, sem=0)
        assert ost.overflow == 0 and ost.max_stack == depth - 1
        sc = trx.Scene(flat)
        got, _ = sc.trace_rays(rays, sem=0)
        assert_hits_equal(got, want, "depth %d" % depth)
        assert got["t"][0] == 1.0 and got["prim"][0] == 0
        sc.close()
    nodes, tris = deep_chain_scene(70)
    sc = trx.Scene(trx.FlatScene(nodes, tris, [], 0, np.arange(70), [0, 70]))
    with pytest.raises(trx.TrxError) as e:
        sc.trace_rays(rays, sem=0)
    assert e.value.code == -4 and "overflowed" in str(e.value)
    got, _ = sc.trace_rays(rays[1:2], sem=0)  # the scene stays usable; the flag was cleared
    sc.close()


def test_the_ray_service_walks_deep_stacks_and_reports_overflow_per_call(trx, orc):
    """trx_traverse1 over a single-level scene is answered by the resident ray service (round 6), whose walker is the thin
    walk: a stack past its LDS part goes to the wave's HBM area, one past 64 entries comes back as THIS call's
    TRX_ERR_STACK_OVERFLOW (the answer carries it; the launch slot's sticky flag stays clear and the next call is served).
    Eight threads at once, each ray checked against the oracle."""
    import threading
    for depth in (10, 13, 40, 64):
        nodes, tris = deep_chain_scene(depth)
        osc = orc.Scene(nodes, tris)
        sc = trx.Scene(trx.FlatScene(nodes, tris, [], 0, np.arange(depth), [0, depth]))
        rays = np.zeros(64, dtype=trx.RAY_DTYPE)
        rays["origin"] = (0.3, 0.3, 1)
        rays["direction"] = (0, 0, -1)
        rays["origin"][1::2] = (0.3, 0.3, -100)
        rays["direction"][1::2] = (0, 0, 1)
        rays["tmax"] = F32_MAX
        want, _ = osc.trace_rays(rays, sem=0)
        bad = []

        def work(ids):
            for i in ids:
                h = sc.traverse(rays["origin"][i], rays["direction"][i], sem=0)
                if np.float32(h.t).view(np.uint32) != want["t"][i].view(np.uint32) or h.primitive_id != want["prim"][i]:
                    bad.append((depth, i, h.t, h.primitive_id))
        ts = [threading.Thread(target=work, args=(range(k, 64, 8),)) for k in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not bad, bad[:3]
        sc.close()
    nodes, tris = deep_chain_scene(70)
    sc = trx.Scene(trx.FlatScene(nodes, tris, [], 0, np.arange(70), [0, 70]))
    with pytest.raises(trx.TrxError) as e:
        sc.traverse((0.3, 0.3, 1), (0, 0, -1), sem=0)
    assert e.value.code == -4 and "overflowed" in str(e.value)
    h = sc.traverse((0.3, 0.3, -100), (0, 0, 1), sem=0)   # the service goes on answering
    assert h.primitive_id != 0xFFFFFFFF
    sc.check()                                             # ... and no launch slot carries a sticky overflow
    sc.close()


@pytest.mark.parametrize("kind", ["rebraided", "transformed"])
def test_the_ray_service_walks_two_level_scenes(trx, orc, kind):
    """trx_traverse1 over a two-level scene is answered by the resident service as well (late round 6: the thin walk's two
    levels - instances entered and left, re-braided entry nodes, instance transforms, the instance id in a granule of its
    own): sixteen threads' RayHits equal trx_traverse_batch's (the full-wave two-level walk) field for field and the
    oracle's hits bit for bit, under the CPU preset and under the shader's text."""
    from helpers import aimed_rays, instanced_scene
    if kind == "rebraided":
        w, h = 160, 120
        flat, view, osc, ov = make_scene(trx, orc, "san_miguel", 120000, w, h, tlas=True)
        assert flat.instance_entry is not None and (np.asarray(flat.instance_entry) != 0).any()   # the TLAS was re-braided
        sc = trx.Scene(flat)
        rays = np.concatenate([osc.primary_rays(ov, w, h), random_rays(trx, flat, 6000, 5)])
    else:
        flat, _o2w, world, _first, _blas_of = instanced_scene(trx, n_instances=24, tris_per_object=900)
        sc = trx.Scene(flat)
        osc = orc.Scene(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start, instance_w2o=sc.instance_world_to_object())
        rays = np.concatenate([aimed_rays(trx, world, 12000, 7), random_rays(trx, flat, 4000, 9)])
    for sem in (3, 0):
        want, winst, _ = osc.trace_rays_inst(rays, sem=sem)
        batch, _ms = sc.traverse_batch(rays, sem=sem)
        before = sc.service_stats()["rays"]
        got, secs, _starts = sc.traverse_threads(rays, threads=16, sem=sem)
        assert sc.service_stats()["rays"] - before >= rays.shape[0]          # every call went through the service
        hit = want["prim"] != 0xFFFFFFFF
        assert hit.sum() > rays.shape[0] // 4
        assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), sem
        assert (got["instance_id"][hit] == winst[hit]).all() and (got["instance_id"][~hit] == 0xFFFFFFFF).all(), sem
        for f in ("primitive_id", "geometry_id", "instance_id"):
            assert (got[f] == batch[f]).all(), (sem, f)
        assert (got["t"].view(np.uint32) == batch["t"].view(np.uint32)).all(), sem
        print("two-level %s, sem %d: %d rays, %.3f Mrays/s from 16 threads" % (kind, sem, rays.shape[0], rays.shape[0] / secs / 1e6))
    sc.close()


def test_the_ray_service_restarts_after_idling_and_serves_more_callers_than_slots(trx, orc):
    """The resident kernel stops itself 50 ms after the last call (a device-wide synchronisation elsewhere waits no longer
    than that) and the next call starts it again - its request is served all the same; 100 threads on its 64 slots wait
    their turn; two scenes run a service each, side by side."""
    import ctypes as C
    import time
    w, h = 48, 32
    scenes = []
    for name in ("kitchen", "bistro"):
        flat, view, osc, ov = make_scene(trx, orc, name, 20000, w, h)
        rays = osc.primary_rays(ov, w, h)
        scenes.append((trx.Scene(flat), rays, osc.trace_rays(rays, sem=3)[0]))
    lib = trx.load()

    def starts(sc):
        n = C.c_uint64()
        assert lib.trx_debug_traverse1_stats(sc.handle, C.byref(n), None) == 0
        return n.value
    try:
        for sc, rays, want in scenes:
            got, _, _ = sc.traverse_threads(rays[:64], threads=4, sem=3)
            assert (got["t"].view(np.uint32) == want["t"][:64].view(np.uint32)).all()
        s0 = [starts(sc) for sc, _, _ in scenes]
        assert min(s0) >= 1
        time.sleep(0.4)                                   # both services have stopped themselves by now
        import torch
        torch.cuda.synchronize()                          # ... so a device-wide synchronisation returns at once
        for (sc, rays, want), before in zip(scenes, s0):
            got, _, _ = sc.traverse_threads(rays, threads=100, sem=3)   # more callers than slots
            hit = want["prim"] != 0xFFFFFFFF
            assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and (got["primitive_id"][hit] == want["prim"][hit]).all()
            assert starts(sc) == before + 1               # one new start for the whole loop
    finally:
        for sc, _, _ in scenes:
            sc.close()


# ---- the Traversable surface ---------------------------------------------------------------------

def test_traverse_single_ray_and_concurrent_callers(trx, orc):
    flat, view, osc, ov = make_scene(trx, orc, "kitchen", 8000, 32, 32, tlas=True)
    sc = trx.Scene(flat)
    rays = osc.primary_rays(ov, 32, 32)
    want, _ = osc.trace_rays(rays, sem=3)
    errors = []

    def work(ids):
        try:
            for i in ids:
                h = sc.traverse(rays["origin"][i], rays["direction"][i], sem=3)
                if np.isinf(want["t"][i]):
                    ok = h.primitive_id == 0xFFFFFFFF and np.isinf(h.t)
                else:
                    g = int(np.searchsorted(flat.blas_tri_start, want["prim"][i], side="right") - 1)
                    ok = (np.float32(h.t) == want["t"][i] and h.geometry_id == g and
                          h.primitive_id == want["prim"][i] - flat.blas_tri_start[g])
                if not ok:
                    errors.append(i)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(range(k, 1024, 64),)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    sc.close()


def test_device_pointers_streams_and_runner(trx, orc):
    import torch
    w, h = 128, 72
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 60000, w, h)
    want, _ = osc.trace_primary(ov, w, h, sem=3)
    sc = trx.Scene(flat)
    stream = torch.cuda.Stream()
    out = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    ao = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(6):  # more launches than launch slots: slot reuse is stream-ordered
            sc.trace_primary_dev(view, w, h, out.data_ptr(), sem=3, stream=stream.cuda_stream)
        sc.trace_ao_dev(view, w, h, out.data_ptr(), ao.data_ptr(), sem=3, frame=0, ao_eps=0.01,
                        stream=stream.cuda_stream)
    sc.check(stream.cuda_stream)
    from tray_racing_amd import dist as D
    assert_hits_equal(D.int64_to_hits(out), want, "torch stream + tensor")
    assert_hits_equal(D.int64_to_hits(ao), osc.trace_ao(ov, w, h, want, sem=3, frame=0, ao_eps=0.01)[0], "ao dev")
    rays = random_rays(trx, flat, 3000, 2)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1, 32).copy()).cuda()
    d_hits = torch.zeros(rays.shape[0], dtype=torch.int64, device="cuda")
    sc.trace_rays_dev(d_rays.data_ptr(), rays.shape[0], d_hits.data_ptr(), sem=0)
    torch.cuda.synchronize()
    assert_hits_equal(D.int64_to_hits(d_hits), osc.trace_rays(rays, sem=0)[0], "rays dev")
    mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=1, frames=5)
    assert 0 < mn <= mean
    sc.close()
    verts, counts = trx.gen_scene("cornell", 0, 1)
    ms, blas_s, tlas_ms = trx.cwbvh_gpu_runner(verts, counts, 64, 64, trx.scene_camera("cornell"), tlas=True, frames=3)
    assert ms > 0 and blas_s >= 0 and tlas_ms >= 0


def test_traverse_batch_is_traverse_for_every_ray(trx, orc):
    """trx_traverse_batch: obvhs RayHits (geometry_id, primitive_id local to the geometry, RayHit::none() for a miss) for
    a whole frame of rays in one launch, against the oracle - two-level scene with geometry ranges, and a one-level
    scene where geometry_id stays 0 and primitive_id is the global triangle."""
    for tlas in (True, False):
        flat, view, osc, ov = make_scene(trx, orc, "bistro", 12000, 48, 40, tlas=tlas)   # a street: some rays reach the sky
        sc = trx.Scene(flat)
        rays = osc.primary_rays(ov, 48, 40)
        want, _ = osc.trace_rays(rays, sem=3)
        got, ms = sc.traverse_batch(rays, sem=3)
        miss = want["prim"] == 0xFFFFFFFF
        assert miss.any() and (~miss).any() and ms > 0
        assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
        assert (got["primitive_id"][miss] == 0xFFFFFFFF).all() and (got["geometry_id"][miss] == 0xFFFFFFFF).all()
        if tlas:
            g = np.searchsorted(flat.blas_tri_start, want["prim"][~miss], side="right") - 1
            assert (got["geometry_id"][~miss] == g).all()
            assert (got["primitive_id"][~miss] == want["prim"][~miss] - flat.blas_tri_start[g]).all()
        else:
            assert (got["geometry_id"][~miss] == 0).all() and (got["primitive_id"][~miss] == want["prim"][~miss]).all()
        empty, _ = sc.traverse_batch(rays[:0], sem=3)
        assert empty.shape[0] == 0
        sc.close()


def test_bench_primary_times_what_hip_events_around_the_same_launches_see(trx, orc):
    """trx_bench_primary (the reference's timing loop: a warm-up dispatch, then a timestamp pair per frame, min and mean,
    src/rt_gpu/rt_gpu_software.rs:289-302,339-344,376) against an independent measurement: torch events on the stream
    around trx_trace_primary_dev launches of the same frame.  Same kernel, same tile-order state: the two agree."""
    import torch
    w, h = 1920, 1080
    flat, view, osc, ov = make_scene(trx, orc, "kitchen", 56939, w, h)
    sc = trx.Scene(flat)
    try:
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=6, frames=30)
        assert 0 < mn <= mean < 3.0 * mn and 0.05 < mn < 5.0            # a 1080p kitchen-class frame is a fraction of a ms
        out = torch.empty(w * h, dtype=torch.int64, device="cuda")
        evs = []
        for k in range(36):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sc.trace_primary_dev(view, w, h, out.data_ptr(), sem=3)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs[6:])
        # minimum against minimum, median against minimum (means would inherit any one slow frame of either loop)
        assert ts[0] == pytest.approx(mn, rel=0.15) and ts[len(ts) // 2] == pytest.approx(mn, rel=0.3)
        # and fewer frames or no warm-up only change the statistics, not the order of magnitude
        mn1, mean1 = sc.bench_primary(view, w, h, sem=3, warmup=0, frames=3)
        assert mn1 == pytest.approx(mn, rel=0.5) and mn1 <= mean1
    finally:
        sc.close()


# ---- BASELINE.json sizes ------------------------------------------------------------------------------

@pytest.mark.parametrize("name,w,h,tlas", [("kitchen", 1920, 1080, False), ("bistro", 1920, 1080, False)])
def test_full_size_config(trx, orc, name, w, h, tlas):
    """BASELINE.json configs[1] / configs[2] at full size: the whole 1920x1080 frame against the
    oracle, plus size-independent properties (variant idempotence, exact scale covariance)."""
    verts, counts = trx.gen_scene(name, 0, 1)
    flat = trx.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = trx.scene_camera(name)
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    got, _ = sc.trace_primary(view, w, h, sem=3)
    osc = orc.Scene.from_flat(flat)
    want, ost = osc.trace_primary(orc.view_from_bytes(view), w, h, sem=3)
    assert_hits_equal(got, want, "%s %dx%d" % (name, w, h))
    st = sc.count_primary(view, w, h, sem=3)
    assert (st.n_node, st.n_tri, st.n_hits) == (ost.n_node, ost.n_tri, ost.n_hits)
    # idempotence: every kernel variant (refill threshold) writes the same frame
    lib = trx.load()
    for variant in (8, 32):
        lib.trx_set_kernel_variant(variant)
        again, _ = sc.trace_primary(view, w, h, sem=3)
        assert_hits_equal(again, got, "variant %d" % variant)
    lib.trx_set_kernel_variant(0)
    sc.close()
    # covariance: scaling the scene and the ray origins by a power of two is exact in binary
    # floating point for the triangle test, so t scales exactly (bit for bit) wherever the same
    # triangle is found; the node test's absolute 1e-4 clamp is not scale-free, hence "wherever"
    s = 4.0
    rays = random_rays(trx, flat, 200000, 9, zero_dirs=False)
    rays["tmin"] = 0.0
    rays["tmax"] = F32_MAX
    sc = trx.Scene(flat)
    base, _ = sc.trace_rays(rays, sem=3)
    sc.close()
    flat2 = trx.flat_build((verts * np.float32(s)).astype(np.float32), counts, use_tlas=tlas)
    rays2 = rays.copy()
    rays2["origin"] = rays["origin"] * np.float32(s)
    sc2 = trx.Scene(flat2)
    scaled, _ = sc2.trace_rays(rays2, sem=3)
    sc2.close()
    hit = np.isfinite(base["t"]) & np.isfinite(scaled["t"])
    same = hit & (flat.tri_source[np.where(hit, base["prim"], 0)] == flat2.tri_source[np.where(hit, scaled["prim"], 0)])
    assert hit.sum() > 10000 and same.sum() > 0.999 * hit.sum()
    assert (bits(scaled["t"][same]) == bits(base["t"][same] * np.float32(s))).all()


def test_full_size_bistro_one_launch_frame(trx, orc):
    """configs[2] at full size through trx_trace_frame_dev (the reference's single dispatch: primary ray, then the AO ray of
    every hit pixel in place): both hit buffers of the 1920x1080 frame against the oracle, bit for bit."""
    import torch
    from tray_racing_amd import dist as D
    w, h = 1920, 1080
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 0, w, h)
    sc = trx.Scene(flat)
    d_p = torch.full((w * h,), -1, dtype=torch.int64, device="cuda")
    d_a = torch.full((w * h,), -1, dtype=torch.int64, device="cuda")
    sc.trace_frame_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=3, frame=1, ao_eps=0.01)
    torch.cuda.synchronize()
    sc.check()
    op, _ = osc.trace_primary(ov, w, h, sem=3)
    oa, _ = osc.trace_ao(ov, w, h, op, sem=3, frame=1, ao_eps=0.01)
    assert_hits_equal(D.int64_to_hits(d_p), op, "bistro one-launch frame, primary")
    assert_hits_equal(D.int64_to_hits(d_a), oa, "bistro one-launch frame, ao")
    sc.close()


def test_full_size_hairball_4spp_whole_frames(trx, orc):
    """configs[3]: hairball-class at 1920x1080, primary + the four AO frames ("4 spp" = frame seeds 0..3), every pixel
    of every frame against the oracle."""
    verts, counts = trx.gen_scene("hairball", 0, 1)
    assert verts.shape[0] == 2880000
    flat = trx.flat_build(verts, counts)
    eye, look, fov = trx.scene_camera("hairball")
    w, h = 1920, 1080
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    osc = orc.Scene.from_flat(flat)
    ov = orc.view_from_bytes(view)
    op, ost = osc.trace_primary(ov, w, h, sem=3)
    assert ost.n_hits > 0.25 * w * h
    # the four AO frames as ONE launch (trx_trace_ao_batch_dev: they share one drain), and frame 0 once more through
    # the two-pass entry point the reference's frame loop maps to
    import torch
    from tray_racing_amd import dist as D
    d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
    d_ao = torch.full((4 * w * h,), -1, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
    sc.trace_ao_batch_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), w * h, 4, sem=3, frame0=0, ao_eps=0.01)
    torch.cuda.synchronize()
    sc.check()
    assert_hits_equal(D.int64_to_hits(d_prim), op, "hairball primary")
    for frame in range(4):
        oao, _ = osc.trace_ao(ov, w, h, op, sem=3, frame=frame, ao_eps=0.01)
        assert_hits_equal(D.int64_to_hits(d_ao[frame * w * h:(frame + 1) * w * h]), oao, "hairball ao frame %d (batch)" % frame)
        if frame == 0:
            gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
            assert_hits_equal(gp, op, "hairball primary (two-pass entry point)")
            assert_hits_equal(gao, oao, "hairball ao frame 0 (two-pass entry point)")
    sc.close()


def test_full_size_san_miguel_tlas_4k_all_eight_shards(trx, orc):
    """configs[4]: san-miguel-class two-level scene at 3840x2160 tiled over 8 ranks: every rank's shard is traced in
    the compact shard layout, the eight shards are de-interleaved the way the gather does it, and the WHOLE 4K frame
    is compared with the oracle; then one AO frame over the whole image."""
    from tray_racing_amd import dist as D
    verts, counts = trx.gen_scene("san_miguel", 0, 1)
    assert verts.shape[0] == 5075977
    flat = trx.flat_build(verts, counts, use_tlas=True)
    assert flat.has_tlas and flat.instance_offsets.size > 500
    eye, look, fov = trx.scene_camera("san_miguel")
    w, h = 3840, 2160
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    osc = orc.Scene.from_flat(flat)
    ov = orc.view_from_bytes(view)
    want, ost = osc.trace_primary(ov, w, h, sem=3)
    world = 8
    import ctypes as C
    import torch
    from tray_racing_amd import _lib as L
    R = D.max_shard_tiles(w, h, world) * 64
    miss = D.FrameGather(w, h, 0, world, "cuda").new_local()[0]
    buf = torch.full((world * R,), int(miss), dtype=torch.int64, device="cuda")
    for rank in range(world):   # what each of the 8 ranks would trace, into its block of the gather buffer
        sc.trace_primary_dev(view, w, h, buf[rank * R:(rank + 1) * R].data_ptr(), sem=3, shard=(rank, world, 1))
    sc.check()
    out = torch.empty(w * h, dtype=torch.int64, device="cuda")
    L.check(L.load().trx_assemble_frames(C.c_void_p(buf.data_ptr()), R, w, h, world, 1, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    frame = D.int64_to_hits(out)
    assert_hits_equal(frame, want, "san_miguel tlas 4k, 8 shards assembled")
    gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
    assert_hits_equal(gp, want, "san_miguel tlas 4k, one GPU")
    oao, _ = osc.trace_ao(ov, w, h, want, sem=3, frame=0, ao_eps=0.01)
    assert_hits_equal(gao, oao, "san_miguel tlas 4k ao")
    sc.close()


def test_full_size_demoscene_cfg0(trx, orc):
    """configs[0]: the demoscene stand-in at its full size (2 x 2048 x 2048 triangles) and the reference's
    --width 512 frame (512x1080), primary + AO, every pixel against the oracle."""
    verts, counts = trx.gen_scene("demoscene", 0, 1)
    assert verts.shape[0] == 2 * 2048 * 2048
    flat = trx.flat_build(verts, counts)
    eye, look, fov = trx.scene_camera("demoscene")
    w, h = 512, 1080
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
    st = sc.count_primary(view, w, h, sem=3)
    sc.close()
    osc = orc.Scene.from_flat(flat)
    ov = orc.view_from_bytes(view)
    op, ost = osc.trace_primary(ov, w, h, sem=3)
    oao, _ = osc.trace_ao(ov, w, h, op, sem=3, frame=0, ao_eps=0.01)
    assert_hits_equal(gp, op, "demoscene 512x1080 primary")
    assert_hits_equal(gao, oao, "demoscene 512x1080 ao")
    assert (st.n_node, st.n_tri, st.n_hits) == (ost.n_node, ost.n_tri, ost.n_hits)


@pytest.mark.parametrize("name,w,h,tlas", [("kitchen", 1920, 1080, False), ("bistro", 1920, 1080, False),
                                           ("hairball", 1920, 1080, False)])
def test_semantics_exposure_hlsl_vs_cpu_preset(trx, orc, name, w, h, tlas):
    """The measurable part of "parity unpinned": TRX_SEM_CPU (what bench.py runs) is a recollection of obvhs, the HLSL
    text is TRX_SEM_HLSL.  The GPU frames under both presets equal the oracle's under the same preset bit for bit, and
    the two presets differ on fewer than 1 ray in 1000 of a BASELINE frame, all inside the 1e-5 tolerance on t
    (profiles/r02_semantics_exposure.txt has the counts for every config, AO rays included)."""
    verts, counts = trx.gen_scene(name, 0, 1)
    flat = trx.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = trx.scene_camera(name)
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    osc = orc.Scene.from_flat(flat)
    ov = orc.view_from_bytes(view)
    frames = {}
    for sem in (0, 3):
        got, _ = sc.trace_primary(view, w, h, sem=sem)
        want, _ = osc.trace_primary(ov, w, h, sem=sem)
        assert_hits_equal(got, want, "%s sem %d" % (name, sem))
        frames[sem] = got
    sc.close()
    a, b = frames[0], frames[3]
    differ = (bits(a["t"]) != bits(b["t"])) | (a["prim"] != b["prim"])
    n = int(differ.sum())
    print("%s: %d of %d primary rays differ between TRX_SEM_HLSL and TRX_SEM_CPU" % (name, n, w * h))
    assert n < 1e-3 * w * h
    both = differ & np.isfinite(a["t"]) & np.isfinite(b["t"])
    assert (np.isfinite(a["t"]) == np.isfinite(b["t"])).all()   # never hit <-> miss
    if both.any():
        assert np.max(np.abs(a["t"][both] - b["t"][both]) / b["t"][both]) < 1e-5   # north_star's tolerance on t


def test_full_size_shard_union_and_counter_sums(trx, orc):
    """Size-independent properties at configs[4]'s full size (san-miguel-class, TLAS, 3840x2160 over 8 ranks):
    the 8 compact shards, gathered and de-interleaved, ARE the single-launch frame bit for bit, and the
    per-shard traversal counters add up to the full frame's.  Plus, on the bistro-class scene, the explicit-ray
    entry point fed the oracle's primary rays answers like in-kernel ray generation."""
    import torch
    from tray_racing_amd import dist as D
    verts, counts = trx.gen_scene("san_miguel", 0, 1)
    flat = trx.flat_build(verts, counts, use_tlas=True)
    eye, look, fov = trx.scene_camera("san_miguel")
    w, h, world = 3840, 2160, 8
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    full = torch.empty(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, full.data_ptr(), sem=3)
    fg = D.FrameGather(w, h, 0, world, "cuda")
    tot = [0, 0, 0, 0]
    for r in range(world):
        block = fg.flat[r * fg.records:(r + 1) * fg.records]
        sc.trace_primary_dev(view, w, h, block.data_ptr(), sem=3, shard=(r, world, 1))
        st = sc.count_primary(view, w, h, sem=3, shard=(r, world))
        tot = [a + b for a, b in zip(tot, (st.n_rays, st.n_node, st.n_tri, st.n_hits))]
    sc.check()
    assert torch.equal(fg.assemble(), full)
    st = sc.count_primary(view, w, h, sem=3)
    assert tot == [st.n_rays, st.n_node, st.n_tri, st.n_hits] and st.n_rays == w * h
    hits = D.int64_to_hits(full)
    assert int(np.isfinite(hits["t"]).sum()) == st.n_hits > 0.9 * w * h
    sc.close()

    w, h = 640, 360
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 0, w, h)
    sc = trx.Scene(flat)
    by_kernel, _ = sc.trace_primary(view, w, h, sem=3)
    by_rays, _ = sc.trace_rays(osc.primary_rays(ov, w, h), sem=3)
    assert_hits_equal(by_rays, by_kernel, "explicit primary rays vs in-kernel generation")
    sc.close()


def test_c_program_over_the_abi_matches_the_oracle(trx, orc, tmp_path):
    """tests/c_abi/trace_frame.c: a plain C11 host (no Python, no torch) drives scene generation, the build,
    the upload and one primary + AO frame through include/trx.h; its frame checksums equal the oracle's."""
    import subprocess
    from test_abi import build_c_consumer
    exe = build_c_consumer("trace_frame", tmp_path)
    w, h = 160, 96
    out = subprocess.run([exe, "kitchen", "20000", str(w), str(h), "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got_p, got_a, n_nodes = out.stdout.split()
    flat, _view, osc, ov = make_scene(trx, orc, "kitchen", 20000, w, h)
    assert int(n_nodes) == flat.n_nodes
    prim, _ = osc.trace_primary(ov, w, h, sem=0)
    ao, _ = osc.trace_ao(ov, w, h, prim, sem=0, frame=5, ao_eps=0.01)

    def fnv(a):
        hsh = 1469598103934665603
        for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
            hsh = ((hsh ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % hsh
    assert (got_p, got_a) == (fnv(prim), fnv(ao))


def test_sixteen_threads_call_traverse_for_their_pixels(trx, orc, tmp_path):
    """The reference's CPU pixel loop over the LITERAL Traversable::traverse (src/rt_cpu/rt_cpu.rs:35-57,
    traversable/src/lib.rs:13-28): tests/c_abi/traverse_threads.c runs 16 host threads, each calling trx_traverse1 for
    its pixels of the 512x1080 demoscene-class frame (configs[0]'s image; 500 000 triangles), one ray per call.  Every
    RayHit equals the oracle's hit for that ray, and the calls shared launches (the per-scene combiner): far fewer
    launches than rays."""
    import subprocess
    from test_abi import build_c_consumer
    exe = build_c_consumer("traverse_threads", tmp_path)
    w, h, tris, threads = 512, 1080, 500000, 16
    flat, _view, osc, ov = make_scene(trx, orc, "demoscene", tris, w, h)
    rays = osc.primary_rays(ov, w, h)
    rays_path, hits_path = str(tmp_path / "rays.bin"), str(tmp_path / "hits.bin")
    np.ascontiguousarray(rays).tofile(rays_path)
    out = subprocess.run([exe, "demoscene", str(tris), str(threads), "3", rays_path, hits_path], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr
    n, secs, launches = out.stdout.split()
    assert int(n) == w * h
    got = np.fromfile(hits_path, dtype=np.dtype([("primitive_id", "<u4"), ("geometry_id", "<u4"), ("instance_id", "<u4"), ("t", "<f4")]))
    want, _ = osc.trace_rays(rays, sem=3)
    miss = want["prim"] == 0xFFFFFFFF
    assert (~miss).any()
    assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    assert (got["primitive_id"] == want["prim"]).all()
    assert (got["geometry_id"][~miss] == 0).all() and (got["geometry_id"][miss] == 0xFFFFFFFF).all()
    # round 6: a single-level scene is served by the resident ray service - a handful of kernel starts for half a million rays
    # (one per stretch of calls; the service stops itself 50 ms after the last call) - faster than a launch per batch of
    # callers (this frame: 0.26 Mrays/s against 0.11 in round 5; a call is a ray's walk alone through cold caches, 1.7 us a trip)
    assert int(launches) < 64, (launches, secs)
    assert w * h / float(secs) / 1e6 > 0.3, secs
    print("traverse1 x %d threads: %.3f Mrays/s, %d service starts" % (threads, w * h / float(secs) / 1e6, int(launches)))


@pytest.mark.parametrize("tlas", [0, 1])
def test_callers_of_mixed_semantics_share_the_single_ray_path(trx, orc, tmp_path, tlas):
    """Even threads call trx_traverse1 with the CPU preset, odd threads with the shader's text, all at once: a scene runs
    one ray service per semantics word (two-level scenes as well since the thin walk learned its two levels; round 5's launch
    combiner, whose batches are of one semantics each, stays behind TRX_TRAVERSE1_COMBINER=1) - every RayHit equals the
    oracle's under its caller's semantics."""
    import subprocess
    from test_abi import build_c_consumer
    exe = build_c_consumer("traverse_threads", tmp_path)
    w, h, tris, threads = 96, 64, 60000, 8
    name = "san_miguel" if tlas else "kitchen"
    verts, counts = trx.gen_scene(name, tris, 1)
    flat = trx.flat_build(verts, counts, use_tlas=bool(tlas))
    eye, look, fov = trx.scene_camera(name)
    view = trx.view_from_camera(eye, look, fov, w, h)
    osc = orc.Scene.from_flat(flat)
    rays = osc.primary_rays(orc.view_from_bytes(view), w, h)
    rays_path, hits_path = str(tmp_path / "rays.bin"), str(tmp_path / "hits.bin")
    np.ascontiguousarray(rays).tofile(rays_path)
    out = subprocess.run([exe, name, str(tris), str(threads), "3", rays_path, hits_path, "0", str(tlas)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(hits_path, dtype=np.dtype([("primitive_id", "<u4"), ("geometry_id", "<u4"), ("instance_id", "<u4"), ("t", "<f4")]))
    idx = np.arange(w * h)
    for sem, sel in ((3, idx % threads % 2 == 0), (0, idx % threads % 2 == 1)):
        if tlas:
            want, winst, _ = osc.trace_rays_inst(rays, sem=sem)
        else:
            want, _ = osc.trace_rays(rays, sem=sem)
        assert (got["t"].view(np.uint32)[sel] == want["t"].view(np.uint32)[sel]).all(), sem
        hit = sel & (want["prim"] != 0xFFFFFFFF)
        assert hit.any()
        assert (got["primitive_id"][hit] == want["prim"][hit]).all(), sem   # (no geometry ranges were set: global triangle ids)
        if tlas:
            assert (got["instance_id"][hit] == winst[hit]).all(), sem


def test_fixed_seed_slice_of_the_fuzzer(trx, orc):
    """tests/fuzz_gpu.py: random scenes, cameras, sizes, semantics, triangle formats, TLAS, shards, batch launches
    and query kinds against the oracle, bit for bit; a fixed-seed slice here, `python tests/fuzz_gpu.py --minutes N`
    for longer runs."""
    import fuzz_gpu
    cases, failures = fuzz_gpu.run(minutes=3.0, seed=20260, max_cases=40, verbose=False)
    assert cases == 40 and not failures, failures[:3]


@pytest.mark.parametrize("world,streams,batch,extra", [(2, 2, 4, []), (4, 8, 8, ["--gather", "abi"]),
                                                        (2, 2, 4, ["--gather", "abi", "--gather-to", "root"]),
                                                        (4, 8, 8, [])])
def test_bench_two_ranks_share_the_gpu(trx, orc, tmp_path, world, streams, batch, extra):
    """bench.py's N > 1 path end to end on one GPU: `world` ranks (gloo, shard gather staged through
    host memory) trace their tile shards on device 0 in batches; the frame rank 0 ends up with is
    checked against the oracle here.  world = 4 runs bench.py's own defaults for more than two GPUs (8 streams, 8
    frames per gather: what the first 8-GPU run will use), through torch.distributed and through the ABI; 4 ranks is what
    can share one GPU box - its process guard allows 6 processes on the card, and this test process and the launcher are
    two of them (a 5-rank run was killed by it) - the eight-rank set-up itself (rendezvous, one build handed to seven ranks,
    gather geometry) runs on CPUs in tests/test_dist_gloo.py.  Every rank must be through set-up,
    timed region and the same-protocol solo run within 240 s (the driver gives the whole run 600).  With `--gather abi` the frames are de-interleaved by the ABI's own
    kernel (trx_assemble_frames) from the staged shards - RCCL itself needs one GPU per rank - and `--gather-to root`
    sends them to rank 0 only."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dump = str(tmp_path / "frame.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps",
           str(batch + 2), "--warmup", "2", "--tris", "150000", "--width", "256", "--height", "136", "--dist-backend", "gloo",
           "--dump-frame", dump, "--streams", str(streams), "--gather-batch", str(batch)] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["value"] > 0
    assert d["scaling"] == "strong" and d["config"]["frames_in_flight"] == streams
    assert d["config"]["frames_per_gather"] == batch   # batch + 2 timed frames = one full batch + a partial one
    ph = d["phases_ms_per_frame"]                    # per-phase times of the N > 1 loop
    assert ph["collective_world_size"] == world and ph["trace"] > 0 and ph["gather"] > 0 and ph["assemble"] > 0
    assert 0 < ph["trace_ranks_min"] <= ph["trace_ranks_max"] and ph["gather_via"] == (extra[1] if extra else "torch")
    assert ph["gather_to"] == ("root" if "root" in extra else "all")
    # round 5: a scaling factor is quoted against ONE rank under the same protocol (same streams, same frames per launch, no
    # gather), the timed region is repeated, and the collective's world is reported at top level
    assert d["protocol_version"] == 6 and d["rccl_world"] == world and d["build"]["lib_sha16"]
    # round 6: ONE build (rank 0, all cores), the other ranks read its flat buffers; per-rank set-up and wall clock in the line
    assert len(d["setup_seconds"]) == world == len(d["build_seconds"]) == len(d["wall_seconds"])
    assert d["scene_cache"][0] in ("built", "hit") and d["scene_cache"][1:] == ["read"] * (world - 1)
    assert d["build_seconds"][1:] == [0.0] * (world - 1) and max(d["wall_seconds"]) < 240.0
    assert d["n1_same_protocol_mrays"] > 0 and d["scaling_vs_same_protocol"] == pytest.approx(d["value"] / d["n1_same_protocol_mrays"], rel=1e-2, abs=1e-4)
    rep = d["legs"]["timed_region_repeats"]
    assert rep["n"] >= 7 and rep["ms_per_step_min"] <= rep["ms_per_step_median"] <= rep["ms_per_step_max"]
    g = np.load(dump + ".scene.npz")
    osc = orc.Scene(g["nodes"], g["tri_verts"], g["instance_offsets"], int(g["tlas_start"]))
    want, _ = osc.trace_primary(orc.view_from_bytes(g["view"].tobytes()), int(g["width"]), int(g["height"]), sem=3)
    from tray_racing_amd import dist as D
    import torch
    assert_hits_equal(D.int64_to_hits(torch.from_numpy(np.load(dump))), want, "%d-rank bench frame" % world)


def test_bench_launches_its_own_ranks(trx, orc, tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher around it (the way the driver runs the scaling bench): the parent
    starts the ranks itself as fresh children of torch.distributed.run before it has touched the GPU, relays rank 0's
    JSON line and the launcher's return code; the frame is the oracle's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "frame.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--tris", "150000",
           "--width", "256", "--height", "136", "--dist-backend", "gloo", "--dump-frame", dump]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # one JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 6 and d["warmup"] == 2
    assert d["phases_ms_per_frame"]["collective_world_size"] == 2
    g = np.load(dump + ".scene.npz")
    osc = orc.Scene(g["nodes"], g["tri_verts"], g["instance_offsets"], int(g["tlas_start"]))
    want, _ = osc.trace_primary(orc.view_from_bytes(g["view"].tobytes()), int(g["width"]), int(g["height"]), sem=3)
    from tray_racing_amd import dist as D
    import torch
    assert_hits_equal(D.int64_to_hits(torch.from_numpy(np.load(dump))), want, "self-launched 2-rank bench frame")
    # a rank that fails makes the parent fail: the launcher's return code is relayed
    bad = subprocess.run(cmd[:-2] + ["--scene", "no_such_scene"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert bad.returncode != 0


def test_scheduling_variants_and_streams_do_not_change_results(trx, orc):
    """Work-queue layout, tile-order feedback, issue priorities, the cooperative triangle phase and
    frames in flight on several streams only change WHEN work is done: every combination must
    reproduce the oracle's frame bit for bit, frame after frame (the order feedback kicks in from
    the second frame of a launch slot)."""
    import torch
    from tray_racing_amd import dist as D
    w, h = 320, 184
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 120000, w, h)
    want, _ = osc.trace_primary(ov, w, h, sem=3)
    want_ao, _ = osc.trace_ao(ov, w, h, want, sem=3, frame=4, ao_eps=0.01)
    sc = trx.Scene(flat)
    lib = trx.load()
    NO_LPT, ONE_Q = 1 << 20, 1 << 21
    never_compact, compact3 = 7 << 25, 3 << 25
    try:
        NO_THIN = 1 << 28   # dry waves never switch to eight lanes per ray
        for variant in (0, NO_LPT, ONE_Q, NO_LPT | ONE_Q, never_compact, compact3, (1 << 22), 4 << 16, 12 << 8, NO_THIN, NO_THIN | (1 << 16)):
            lib.trx_set_kernel_variant(variant)
            for rep in range(6):  # > kSlots launches: every slot gets to read its own feedback
                got, ao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=4, ao_eps=0.01)
                assert_hits_equal(got, want, "variant 0x%x rep %d" % (variant, rep))
                assert_hits_equal(ao, want_ao, "variant 0x%x rep %d ao" % (variant, rep))
        lib.trx_set_kernel_variant(0)
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = [torch.zeros(w * h, dtype=torch.int64, device="cuda") for _ in range(3)]
        for k in range(30):
            s = streams[k % 3]
            sc.trace_primary_dev(view, w, h, outs[k % 3].data_ptr(), sem=3, stream=s.cuda_stream)
        for s in streams:
            sc.check(s.cuda_stream)
        for o in outs:
            assert_hits_equal(D.int64_to_hits(o), want, "3 frames in flight")
    finally:
        lib.trx_set_kernel_variant(0)
        sc.close()


def test_frame_loop_keeps_one_tile_order_per_pass_kind(trx, orc):
    """The reference's frame loop runs the primary pass and the AO pass on one queue, so both land on one launch slot:
    each kind has its own tile-order lists and schedule-tuner state there (the AO pass uses them only when it runs with
    whole-tile refills - variant bits 0..6 = 64 - its default replaces finished rays mid-tile and files no order).
    60 frames of both passes - past the tuner's first decisions - equal the oracle's, in both AO refill modes."""
    w, h = 328, 200
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 150000, w, h)
    want_p = osc.trace_primary(ov, w, h, sem=3)[0]
    want_ao = osc.trace_ao(ov, w, h, want_p, sem=3, frame=3, ao_eps=0.01)[0]
    sc = trx.Scene(flat)
    lib = trx.load()
    try:
        for variant in (0, 64):
            lib.trx_set_kernel_variant(variant)
            for k in range(60):
                gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=3, ao_eps=0.01)
                if k % 4 == 0 or 22 <= k < 36:
                    assert_hits_equal(gp, want_p, "frame loop primary, variant %d frame %d" % (variant, k))
                    assert_hits_equal(gao, want_ao, "frame loop AO, variant %d frame %d" % (variant, k))
            # round 6: the same loop without the host between its frames (trx_frame_loop), serial on one stream and with
            # frame i's AO pass on a second stream under frame i + 1's primary pass; the noise seed advances per frame
            # (--animate), so the last of n frames carries seed frame0 + n - 1
            for overlap in (False, True):
                for n in (1, 2, 7):
                    _, gp, gao = sc.frame_loop(view, w, h, sem=3, frames=n, frame0=4 - n, animate=True, ao_eps=0.01, overlap=overlap)
                    assert_hits_equal(gp, want_p, "trx_frame_loop primary, variant %d overlap %d, %d frames" % (variant, overlap, n))
                    assert_hits_equal(gao, want_ao, "trx_frame_loop AO, variant %d overlap %d, %d frames" % (variant, overlap, n))
                _, gp, gao = sc.frame_loop(view, w, h, sem=3, frames=40, frame0=3, animate=False, ao_eps=0.01, overlap=overlap)
                assert_hits_equal(gao, want_ao, "trx_frame_loop AO, static seed, variant %d overlap %d" % (variant, overlap))
    finally:
        lib.trx_set_kernel_variant(0)
        sc.close()


def test_frame_loop_over_a_two_level_scene(trx, orc):
    """trx_frame_loop over a TLAS scene (its passes carry instance ids between them; the two-level AO kernel has neither thin
    waves nor the drain hand-over): serial and overlapped, last frame equal to the oracle's primary + AO frame."""
    w, h = 200, 120
    flat, view, osc, ov = make_scene(trx, orc, "san_miguel", 120000, w, h, tlas=True)
    want_p = osc.trace_primary(ov, w, h, sem=3)[0]
    want_ao = osc.trace_ao(ov, w, h, want_p, sem=3, frame=5, ao_eps=0.01)[0]
    sc = trx.Scene(flat)
    try:
        for overlap in (False, True):
            for n in (1, 3, 6):
                _, gp, gao = sc.frame_loop(view, w, h, sem=3, frames=n, frame0=6 - n, animate=True, ao_eps=0.01, overlap=overlap)
                assert_hits_equal(gp, want_p, "two-level frame loop primary, overlap %d, %d frames" % (overlap, n))
                assert_hits_equal(gao, want_ao, "two-level frame loop AO, overlap %d, %d frames" % (overlap, n))
    finally:
        sc.close()


def test_camera_cuts_and_schedule_modes_never_change_the_hits(trx, orc):
    """The tile order is replayed whatever the camera did (a stale order measured no worse than none); a camera cut
    restarts the schedule tuner.  Alternating between two unrelated cameras (every frame a cut), drifting slowly, variant
    bit 7 (every frame runs as a first frame: natural order) and the tuner's own mode switches only change WHEN tiles
    are traced - each frame equals the oracle's."""
    w, h = 328, 200
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 150000, w, h)
    eye, look, fov = trx.scene_camera("bistro")
    far = trx.view_from_camera((eye[0] + 14.0, eye[1] + 6.0, eye[2] - 5.0), (look[0] - 20.0, look[1] - 3.0, look[2] + 9.0), 55.0, w, h)
    want = {0: osc.trace_primary(ov, w, h, sem=3)[0], 1: osc.trace_primary(orc.view_from_bytes(bytes(far)), w, h, sem=3)[0]}
    sc = trx.Scene(flat)
    lib = trx.load()
    try:
        for k in range(8):                       # A B A B ...: a cut every frame
            got = sc.trace_primary(far if k & 1 else view, w, h, sem=3)[0]
            assert_hits_equal(got, want[k & 1], "alternating cameras, frame %d" % k)
        for k in range(4):                       # the same view again: learnt order
            assert_hits_equal(sc.trace_primary(view, w, h, sem=3)[0], want[0], "static camera, frame %d" % k)
        for k in range(6):                       # a slow drift: below the cut thresholds, the stale order is replayed
            v = trx.view_from_camera((eye[0] + 0.01 * k, eye[1], eye[2]), (look[0] + 0.01 * k, look[1], look[2]), fov, w, h)
            assert_hits_equal(sc.trace_primary(v, w, h, sem=3)[0], osc.trace_primary(orc.view_from_bytes(bytes(v)), w, h, sem=3)[0],
                              "drifting camera, frame %d" % k)
        lib.trx_set_kernel_variant(1 << 7)
        for k in range(3):
            assert_hits_equal(sc.trace_primary(view, w, h, sem=3)[0], want[0], "every frame a cut, frame %d" % k)
        # the feedback tunes itself: 128 frames with it (the order of the first one replayed unchanged while the view is),
        # 4 without, 4 with mid-tile refills, then the fastest mode holds - whichever mode a frame runs in, and across the
        # switches, the hits are the oracle's
        lib.trx_set_kernel_variant(0)
        for k in range(150):
            got = sc.trace_primary(view, w, h, sem=3)[0]
            if k % 7 == 0 or 120 <= k < 142:
                assert_hits_equal(got, want[0], "self-tuning feedback, frame %d" % k)
    finally:
        lib.trx_set_kernel_variant(0)
        sc.close()


def test_literal_division_shortcut_is_exact_at_its_edges(trx, orc):
    """TRX_SEM_HLSL divides per node like the shader (query.hlsl:237-243); the kernel computes e / d as e * (1/d) where
    that is the same float (e a power of two, nothing leaves the normal range) and divides otherwise.  Rays whose
    direction components are denormal, enormous, zero, or just inside the allowed range, and a scene so small that its
    exponent bytes fall below 21, take the other path - every hit equals the oracle's, which always divides.  Likewise
    (p - o) / d, computed from the ray's 1/d by one correction step where ray and scene admit it (round 5; all 2^46
    pairs of significands checked against `/` by tools/ubench/div_exhaustive.hip, profiles/r05_div_exhaustive.log)."""
    w, h = 64, 64
    flat, _view, osc, _ov = make_scene(trx, orc, "cornell", 0, w, h)
    rays = random_rays(trx, flat, 4096, 77)
    d = rays["direction"]
    rng = np.random.default_rng(5)
    edge = np.array([1e-39, -1e-39, 1.2e-38, -1.2e-38, 1e-30, 1048576.0, -1048576.0, 1048577.0, 3e6, -2.5e30, 0.0, -0.0,
                     1.17549435e-38, 5e-324], dtype=np.float64).astype(np.float32)
    for i in range(64, 2048):                      # one or two edge components per ray, the rest as drawn
        d[i, rng.integers(0, 3)] = edge[rng.integers(0, edge.size)]
        if i % 3 == 0:
            d[i, rng.integers(0, 3)] = edge[rng.integers(0, edge.size)]
    # ... and the edges of the other shortcut, (p - o) / d from the ray's reciprocal by one correction step (kernels.hip
    # div_by_rcp; admitted per RAY: |d| in [2^-30, 2^20], every origin component 0 or in [2^-36, 2^59]; per SCENE: every
    # node origin component +0 or in that range; a ray outside sends every node step of its wave through `/`):
    # directions on and just outside 2^-30, origins on and outside both ends, zero origins of either sign, origins that
    # coincide with a node's quantisation origin (p - o = +0) or sit one ulp beside it
    two30 = np.float32(2.0 ** -30)
    edge_d = np.array([two30, -two30, np.nextafter(two30, np.float32(0)), -np.nextafter(two30, np.float32(0)),
                       np.nextafter(two30, np.float32(1)), np.float32(2.0 ** -31), np.float32(-2.0 ** -29)], dtype=np.float32)
    lo36, hi59 = np.float32(2.0 ** -36), np.float32(2.0 ** 59)
    edge_o = np.array([0.0, -0.0, lo36, -lo36, np.nextafter(lo36, np.float32(0)), -np.nextafter(lo36, np.float32(0)), 1e-13, 1e-30, 1e-39,
                       hi59, -hi59, np.nextafter(hi59, np.float32(np.inf)), 1e18, -1.2e18, 3e19], dtype=np.float64).astype(np.float32)
    node_p = flat.nodes[:, 0:3].copy().view(np.float32)
    o = rays["origin"]
    for i in range(2048, 4096):
        k = i % 8
        if k < 2:
            d[i, rng.integers(0, 3)] = edge_d[rng.integers(0, edge_d.size)]
        elif k < 5:
            ax = rng.integers(0, 3)
            pv = node_p[rng.integers(0, node_p.shape[0]), ax]
            o[i, ax] = (pv, np.nextafter(pv, np.float32(np.inf)), np.nextafter(pv, np.float32(-np.inf)))[k - 2]
            if i % 16 >= 8:
                o[i, (ax + 1) % 3] = node_p[rng.integers(0, node_p.shape[0]), (ax + 1) % 3]
        elif k < 7:
            o[i, rng.integers(0, 3)] = edge_o[rng.integers(0, edge_o.size)]
        # (k == 7: as drawn)
    rays["direction"] = d
    rays["origin"] = o
    sc = trx.Scene(flat)
    try:
        for sem in (0, 2, 4, 6):                   # every semantics word whose node test divides
            got, _ = sc.trace_rays(rays, sem=sem)
            want, _ = osc.trace_rays(rays, sem=sem)
            assert_hits_equal(got, want, "edge directions, sem %d" % sem)
        # whole waves inside the range (no lane sends a step through `/`): rays as drawn, far from any edge
        plain = random_rays(trx, flat, 4096, 78, zero_dirs=False)
        for sem in (0, 4):
            got, _ = sc.trace_rays(plain, sem=sem)
            want, _ = osc.trace_rays(plain, sem=sem)
            assert_hits_equal(got, want, "plain rays, sem %d" % sem)
    finally:
        sc.close()
    # a tree some of whose node origins are -0, tiny or enormous (the boxes move with them - for the oracle too): the
    # scene's flag allows the power-of-two shortcut only, every (p - o) / d is divided
    import copy
    odd = copy.copy(flat)
    odd.nodes = flat.nodes.copy()
    pf = odd.nodes[:, 0:3].view(np.float32)
    pf[7::23, 0] = np.float32(-0.0)
    pf[11::29, 1] = np.float32(1e-13)
    pf[13::31, 2] = np.float32(-3e18)
    oosc = orc.Scene.from_flat(odd)
    sc = trx.Scene(odd)
    try:
        for sem in (0, 4):
            got, _ = sc.trace_rays(plain, sem=sem)
            want, _ = oosc.trace_rays(plain, sem=sem)
            assert_hits_equal(got, want, "odd node origins, sem %d" % sem)
    finally:
        sc.close()
    # the same tree with every exponent byte lowered by 40 (boxes 2^-40 of their size, some bytes below 21): the shortcut
    # is off for the whole scene, and whatever the collapsed boxes let through is what the oracle lets through
    low = copy.copy(flat)
    low.nodes = flat.nodes.copy()
    eb = low.nodes.view(np.uint8).reshape(-1, 80)[:, 12:15]
    eb[:] = np.maximum(eb.astype(np.int32) - 40, 1).astype(np.uint8)
    assert (eb < 21).any() and (eb >= 21).any()
    losc = orc.Scene.from_flat(low)
    sc = trx.Scene(low)
    try:
        for sem in (0, 3):
            got, _ = sc.trace_rays(rays, sem=sem)
            want, _ = losc.trace_rays(rays, sem=sem)
            assert_hits_equal(got, want, "lowered exponents, sem %d" % sem)
    finally:
        sc.close()


def test_fetch_probe_measures_a_rate_and_honours_the_mix(trx, orc):
    """trx_debug_fetch_rate (bench.py's `fetch_vs_random`): random nodes of the scene's own buffers per second, with the
    asked-for number of triangle records beside them; traces after it are untouched by it."""
    w, h = 64, 48
    flat, view, osc, ov = make_scene(trx, orc, "bistro", 60000, w, h)
    sc = trx.Scene(flat)
    try:
        n0, t0 = sc.fetch_rate(0.0, steps=64)
        n1, t1 = sc.fetch_rate(0.5, steps=64)
        n2, t2 = sc.fetch_rate(1.5, steps=64)
        assert n0 > 1e8 and t0 == 0.0
        assert abs(t1 / n1 - 0.5) < 1e-9 and abs(t2 / n2 - 1.5) < 1e-9
        assert n0 >= 0.8 * n1 and n1 >= 0.8 * n2      # more bytes per step, no more steps per second
        got = sc.trace_primary(view, w, h, sem=3)[0]
        want = osc.trace_primary(ov, w, h, sem=3)[0]
        assert_hits_equal(got, want, "primary frame after the probe")
    finally:
        sc.close()
    with pytest.raises(Exception):
        trx.Scene(flat).fetch_rate(99.0)
