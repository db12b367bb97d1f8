"""The BVH2 stage of the ploc_cwbvh build on the GPU (csrc/ploc_gpu.cpp: Morton codes, radix sort, PLOC merge rounds as
kernels) must return the very tree the host stage returns: same operations in the same order, so the flat buffers that
come out of reinsertion / collapse / encoding are byte-identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu




@pytest.mark.parametrize("name,n", [("bistro", 200000), ("hairball", 120000), ("kitchen", 56939), ("soup", 40000)])
def test_device_ploc_tree_equals_host_ploc_tree(trx, orc, name, n):
    lib = trx.load()
    verts, _counts = trx.gen_scene(name, n, 1)
    counts = np.array([n], dtype=np.uint64)   # one object: large enough for the device stage
    try:
        for dist, thresh, bits, ratio in [(14, 2, 64, 0.0), (14, 2, 128, 0.0), (2, 0, 64, 0.0), (32, 5, 64, 0.02)]:
            bp = trx.build_params(ploc_search_distance=dist, search_depth_threshold=thresh, sort_precision=bits,
                                  reinsertion_batch_ratio=ratio)
            assert lib.trx_set_build_device(-1) == 0
            host = trx.flat_build_params(verts, counts, bp)
            assert lib.trx_set_build_device(0) == 0
            dev = trx.flat_build_params(verts, counts, bp)
            assert (host.nodes == dev.nodes).all(), (name, dist, thresh, bits, ratio)
            assert (host.tri_source == dev.tri_source).all()
        assert orc.Scene.from_flat(dev).validate() == (0, "")
    finally:
        lib.trx_set_build_device(-1)


def test_device_ploc_full_size_bistro_traces_like_the_host_build(trx, orc):
    """3.87 M triangles through the device stage: valid tree, and a traced frame equal to the oracle's on it."""
    import time
    lib = trx.load()
    verts, _c = trx.gen_scene("bistro", 0, 1)
    counts = np.array([verts.shape[0]], dtype=np.uint64)
    bp = trx.build_params(reinsertion_batch_ratio=0.0)
    try:
        assert lib.trx_set_build_device(0) == 0
        t0 = time.time()
        flat = trx.flat_build_params(verts, counts, bp)
        t_dev = time.time() - t0
        lib.trx_set_build_device(-1)
        t0 = time.time()
        host = trx.flat_build_params(verts, counts, bp)
        t_host = time.time() - t0
    finally:
        lib.trx_set_build_device(-1)
    print("bistro-class 3.87 M triangles, PLOC alone: %.2f s with the BVH2 stage on the GPU, %.2f s on the host cores" % (t_dev, t_host))
    assert (flat.nodes == host.nodes).all()
    osc = orc.Scene.from_flat(flat)
    assert osc.validate() == (0, "")
    eye, look, fov = trx.scene_camera("bistro")
    w, h = 480, 270
    view = trx.view_from_camera(eye, look, fov, w, h)
    sc = trx.Scene(flat)
    got, _ = sc.trace_primary(view, w, h, sem=3)
    sc.close()
    want, _ = osc.trace_primary(orc.view_from_bytes(view), w, h, sem=3)
    assert (got["prim"] == want["prim"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()


@pytest.mark.parametrize("name,n,tlas", [("bistro", 300000, False), ("hairball", 150000, False), ("soup", 60000, False)])
def test_device_reinsertion_searches_equal_host_searches(trx, orc, name, n, tlas):
    """Reinsertion with one batch per iteration (trx_set_build_reinsertion_batches) on the host cores - searches on every
    core, moves applied in candidate order - or entirely on the GPU with the tree resident there (csrc/reinsert_gpu.cpp:
    area keys + radix sort, one thread per search, the moves the sequential pass would apply found as a fixed point of
    atomic-min claims, checked for the one rule that needs the order, applied at once, boxes recomputed level by level).
    Same flat buffers, byte for byte - for the PLOC pipeline (its BVH2 stage on the device as well) and for the binned-SAH
    presets; also when some iterations are handed to the host in the middle of the pass (TRX_BUILD_HOST_APPLY), which is
    what happens when the device finds a move only the sequential pass can judge.  And, independently of either builder: the device-built tree is a
    valid CWBVH over the scene's triangles (orc_validate) and a BVH-free brute-force query over the same rays finds the
    hits the oracle finds through it."""
    lib = trx.load()
    verts, _counts = trx.gen_scene(name, n, 1)
    counts = np.array([n], dtype=np.uint64)
    try:
        assert lib.trx_set_build_reinsertion_batches(1) == 0
        for pipeline in ("ploc", "preset"):
            assert lib.trx_set_build_preset(b"medium_build") == 0
            assert lib.trx_set_build_reinsertion(0.05, 6) == 0
            built = []
            for device in (-1, 0):
                assert lib.trx_set_build_device(device) == 0
                built.append(trx.flat_build_params(verts, counts, trx.build_params()) if pipeline == "ploc"
                             else trx.flat_build(verts, counts))
            host, dev = built
            assert (host.nodes == dev.nodes).all() and (host.tri_source == dev.tri_source).all(), (name, pipeline)
            os.environ["TRX_BUILD_HOST_APPLY"] = "0x2a"   # iterations 1, 3 and 5 go to the host and come back
            try:
                mixed = trx.flat_build_params(verts, counts, trx.build_params()) if pipeline == "ploc" else trx.flat_build(verts, counts)
            finally:
                del os.environ["TRX_BUILD_HOST_APPLY"]
            assert (host.nodes == mixed.nodes).all() and (host.tri_source == mixed.tri_source).all(), (name, pipeline, "hand-over")
            osc = orc.Scene.from_flat(dev)
            assert osc.validate() == (0, "")
        # the one-batch pass really moved nodes: not the tree of the same pipeline without reinsertion
        assert lib.trx_set_build_reinsertion(0.0, 0) == 0
        plain = trx.flat_build(verts, counts)
        assert plain.nodes.shape != dev.nodes.shape or not (plain.nodes == dev.nodes).all()
        eye, look, fov = trx.scene_camera(name)
        w, h = 96, 64
        view = trx.view_from_camera(eye, look, fov, w, h)
        ov = orc.view_from_bytes(view)
        through, _ = osc.trace_primary(ov, w, h, sem=3)
        brute = osc.brute_primary(ov, w, h, sem=3)
        assert (through["t"].view(np.uint32) == brute["t"].view(np.uint32)).all()
        sc = trx.Scene(dev)
        got, _ = sc.trace_primary(view, w, h, sem=3)
        sc.close()
        assert (got["prim"] == through["prim"]).all() and (got["t"].view(np.uint32) == through["t"].view(np.uint32)).all()
    finally:
        lib.trx_set_build_device(-1)
        lib.trx_set_build_reinsertion_batches(0)
        lib.trx_set_build_preset(b"medium_build")


@pytest.mark.parametrize("max_prims,cost", [(3, 1.0), (1, 1.0), (2, 0.4), (3, 3.0)])
def test_device_collapse_and_encoding_equal_the_host_stage(trx, orc, max_prims, cost):
    """With a build device set, the BVH2 -> CWBVH stage runs as kernels (csrc/collapse_gpu.cpp: cost table level by level,
    the collapsed tree top-down, subtree sizes, output offsets, encoding) on a tree left in whatever layout the
    reinsertion pass ended with.  Same nodes and triangle order as the host stage (Collapser in builder.cpp, which
    re-lays the tree out first) for every leaf size and traversal cost, behind each of the three reinsertion passes; the
    result validates and traces like the oracle says."""
    lib = trx.load()
    try:
        for name, n in (("bistro", 200000), ("soup", 60000)):
            verts, _counts = trx.gen_scene(name, n, 1)
            counts = np.array([n], dtype=np.uint64)
            for mode in ("one at a time", "batches of 128", "whole iterations"):
                assert lib.trx_set_build_preset(b"medium_build") == 0
                assert lib.trx_set_build_reinsertion_batches(1 if mode == "whole iterations" else 0) == 0
                built = []
                for device in (-1, 0):
                    assert lib.trx_set_build_device(device) == 0
                    if mode == "one at a time":
                        built.append(trx.flat_build(verts, counts, max_prims_per_leaf=max_prims, traversal_cost=cost))
                    else:
                        built.append(trx.flat_build_params(verts, counts, trx.build_params(max_prims_per_leaf=max_prims,
                                                                                         collapse_traversal_cost=cost)))
                host, dev = built
                assert host.nodes.shape == dev.nodes.shape and (host.nodes == dev.nodes).all(), (name, mode)
                assert (host.tri_source == dev.tri_source).all(), (name, mode)
            osc = orc.Scene.from_flat(dev)
            assert osc.validate() == (0, "")
            eye, look, fov = trx.scene_camera(name)
            w, h = 96, 64
            view = trx.view_from_camera(eye, look, fov, w, h)
            want, _ = osc.trace_primary(orc.view_from_bytes(view), w, h, sem=3)
            sc = trx.Scene(dev)
            got, _ = sc.trace_primary(view, w, h, sem=3)
            sc.close()
            assert (got["prim"] == want["prim"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    finally:
        lib.trx_set_build_device(-1)
        lib.trx_set_build_reinsertion_batches(0)
        lib.trx_set_build_costs(1.0, 0.3)
        lib.trx_set_build_preset(b"medium_build")


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["very_fast_build", "medium_build", "slow_build"])
def test_a_presets_build_with_every_stage_on_the_device(trx, orc, preset):
    """Round 6 (review item 8): trx_flat_build_preset_device - the ploc_cwbvh pipeline (Morton sort + PLOC rounds,
    whole-iteration reinsertion, collapse + encoding, all kernels) under a preset's reinsertion budget.  Its bytes are those
    of the same pipeline on the host cores (device = -1), the tree validates against its triangles, and a frame traced
    over it equals the oracle's."""
    for name, n in (("bistro", 150000), ("hairball", 90000)):
        verts, counts = trx.gen_scene(name, n, 1)
        dev = trx.flat_build_preset_device(verts, counts, preset=preset, device=0)
        host = trx.flat_build_preset_device(verts, counts, preset=preset, device=-1)
        assert host.nodes.shape == dev.nodes.shape and (host.nodes == dev.nodes).all(), (name, preset)
        assert (host.tri_source == dev.tri_source).all(), (name, preset)
        osc = orc.Scene.from_flat(dev)
        if preset != "slow_build":   # (pre-split references cover part of their triangle: validated against their build boxes)
            assert osc.validate() == (0, "")
        else:
            assert osc.validate(boxes=dev.tri_boxes) == (0, "")
        eye, look, fov = trx.scene_camera(name)
        w, h = 96, 64
        view = trx.view_from_camera(eye, look, fov, w, h)
        sc = trx.Scene(dev)
        got, _ = sc.trace_primary(view, w, h, sem=3)
        sc.close()
        want, _ = osc.trace_primary(orc.view_from_bytes(view), w, h, sem=3)
        assert (got["prim"] == want["prim"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    with pytest.raises(trx.TrxError, match="unknown preset"):
        trx.flat_build_preset_device(verts, counts, preset="no_such_build", device=0)
