"""Host-side CWBVH construction: structure of the 80-byte nodes and of the flat
buffers cwbvh_gpu_runner assembles (src/rt_gpu/mod.rs:16-112)."""
import numpy as np
import pytest

from helpers import make_scene


def node_fields(nodes):
    b = nodes.view(np.uint8).reshape(-1, 80)
    return dict(e=b[:, 12:15], imask=b[:, 15], child_base=nodes[:, 4], prim_base=nodes[:, 5], meta=b[:, 24:32])


@pytest.mark.parametrize("name,n,tlas", [("cornell", 0, False), ("cornell", 0, True), ("kitchen", 9000, False),
                                          ("kitchen", 9000, True), ("bistro", 40000, False), ("hairball", 8000, False),
                                          ("san_miguel", 40000, True), ("demoscene", 20000, False), ("soup", 3000, False)])
def test_built_bvh_is_structurally_sound(trx, orc, name, n, tlas):
    flat, _v, osc, _ov = make_scene(trx, orc, name, n, 8, 8, tlas=tlas)
    assert osc.validate() == (0, "")
    f = node_fields(flat.nodes)
    meta = f["meta"]
    inner = (meta & 0x18) == 0x18
    # imask mirrors the inner slots; inner meta is 0b001_11sss (embree/src/bvh_embree_to_cwbvh.rs:153-154)
    assert (np.packbits(inner[:, ::-1], axis=1)[:, 0] == f["imask"]).all()
    slots = np.arange(8, dtype=np.uint8)[None, :]
    assert (meta[inner] == ((0x20 | (24 + slots)) * np.ones_like(meta))[inner]).all()
    # leaves: unary count in the top 3 bits, running offset below, at most 24 per node (:159-167)
    leaf = (meta != 0) & ~inner
    cnt = np.select([(meta >> 5) == 1, (meta >> 5) == 3, (meta >> 5) == 7], [1, 2, 3], 0) * leaf
    assert (cnt[leaf] > 0).all() and cnt.sum(1).max() <= 24
    # exponent bytes are biased IEEE exponents of a positive power of two (:104-110)
    assert (f["e"] > 0).all() and (f["e"] < 255).all()
    # triangles were permuted into primitive_indices order; tri_source is that permutation
    assert sorted(flat.tri_source.tolist()) == list(range(flat.n_tris))
    verts, _ = trx.gen_scene(name, n, 1)
    assert (flat.tri_verts == verts[flat.tri_source]).all()
    if tlas:
        assert flat.has_tlas and flat.tlas_start > 0
        assert (flat.instance_offsets < flat.tlas_start).all()
        # a TLAS primitive is a BLAS, or - re-braided (trx_set_build_rebraid, the default) - a subtree of one: the
        # (BLAS, entry node) pairs are distinct, and every BLAS is referenced
        entry = flat.instance_entry if flat.instance_entry is not None else np.zeros_like(flat.instance_offsets)
        pairs = set(zip(flat.instance_offsets.tolist(), entry.tolist()))
        assert len(pairs) == flat.instance_offsets.size
        assert len(set(flat.instance_offsets.tolist())) == flat.blas_tri_start.size - 1
        assert flat.blas_tri_start[0] == 0 and flat.blas_tri_start[-1] == flat.n_tris
        assert (np.diff(flat.blas_tri_start.astype(np.int64)) > 0).all()
    else:
        assert not flat.has_tlas and flat.tlas_start == 0


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 9, 25])
def test_tiny_inputs(trx, orc, n):
    verts, _ = trx.gen_scene("soup", max(n, 1), 5)
    verts = verts[:n]
    flat = trx.flat_build(verts)
    assert flat.n_tris == n and flat.n_nodes >= 1
    osc = orc.Scene.from_flat(flat) if n else None
    if n:
        assert osc.validate() == (0, "")
        eye, look, fov = trx.scene_camera("soup")
        ov = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, 24, 24))
        got, _ = osc.trace_primary(ov, 24, 24)
        bf = osc.brute_primary(ov, 24, 24)
        assert (got["t"].view(np.uint32) == bf["t"].view(np.uint32)).all()
    else:
        f = node_fields(flat.nodes)
        assert (f["meta"] == 0).all() and f["imask"][0] == 0


def test_duplicate_and_degenerate_triangles(trx, orc):
    one = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float32)
    verts = np.concatenate([np.repeat(one, 40, axis=0),            # 40 identical triangles (identical centroids)
                            np.zeros((5, 9), np.float32),          # zero-area triangles at the origin
                            one + np.float32(2.0)])
    flat = trx.flat_build(verts)
    osc = orc.Scene.from_flat(flat)
    assert osc.validate() == (0, "")
    rays = np.zeros(2, dtype=orc.RAY_DTYPE)
    rays["origin"] = [(0.25, 0.25, 5), (2.25, 2.25, 5)]
    rays["direction"] = [(0, 0, -1), (0, 0, -1)]
    rays["tmax"] = 3.4e38
    got, _ = osc.trace_rays(rays)
    assert got["t"][0] == 5.0 and got["t"][1] == 3.0
    assert flat.tri_source[got["prim"][0]] < 40 and flat.tri_source[got["prim"][1]] == 45


@pytest.mark.parametrize("max_prims", [1, 2, 3])
def test_max_prims_per_leaf(trx, max_prims):
    verts, _ = trx.gen_scene("soup", 2000, 2)
    flat = trx.flat_build(verts, max_prims_per_leaf=max_prims)
    meta = node_fields(flat.nodes)["meta"]
    leaf = (meta != 0) & ((meta & 0x18) != 0x18)
    cnt = np.select([(meta >> 5) == 1, (meta >> 5) == 3, (meta >> 5) == 7], [1, 2, 3], 0)[leaf]
    assert cnt.max() <= max_prims and cnt.sum() == 2000
    with pytest.raises(trx.TrxError, match="maximum of 3 primitives"):  # src/main.rs:176-178
        trx.flat_build(verts, max_prims_per_leaf=4)


def test_large_builds_take_the_parallel_paths_and_stay_deterministic(trx, orc):
    """Above 65 536 BVH2 nodes the collapse and the emission run over subtrees concurrently and splice the results,
    and above 65 536 reinsertion candidates the selection is a parallel sample sort (builder.cpp `emit_all`,
    `compute_costs`, `select_candidates`): one thread and several must produce the same bytes, and a valid tree."""
    verts, counts = trx.gen_scene("bistro", 240000, 1)
    for build in (lambda t: trx.flat_build_params(verts, counts, trx.build_params(), threads=t),
                  lambda t: trx.flat_build(verts, counts, preset="medium_build", threads=t),
                  lambda t: trx.flat_build(verts, counts, preset="medium_build", split=0.3, threads=t)):
        a, b = build(1), build(6)
        assert a.n_nodes > 16000 and a.nodes.tobytes() == b.nodes.tobytes()
        assert (a.tri_source == b.tri_source).all() and a.tri_verts.tobytes() == b.tri_verts.tobytes()
        assert orc.Scene.from_flat(b).validate(boxes=b.tri_boxes) == (0, "")   # tri_boxes: entries of the pre-split build are clipped
    trx.flat_build(verts[:4], preset="medium_build", split=0.0)   # process-wide settings back to their defaults


def test_build_is_deterministic_across_thread_counts(trx):
    verts, counts = trx.gen_scene("kitchen", 30000, 1)
    a = trx.flat_build(verts, counts, threads=1)
    b = trx.flat_build(verts, counts, threads=4)
    assert (a.nodes == b.nodes).all() and (a.tri_source == b.tri_source).all()


@pytest.mark.parametrize("setting", [(0.02, 4), (0.5, 2), (1.0, 1)])
def test_reinsertion_keeps_the_tree_valid_and_the_hits_unchanged(trx, orc, setting):
    """The BVH2 reinsertion pass (obvhs `reinsertion_batch_ratio`, src/main.rs:113-118) only moves
    subtrees: the collapsed tree must stay sound, hold every triangle once, and answer like
    brute force."""
    verts, counts = trx.gen_scene("bistro", 20000, 3)
    eye, look, fov = trx.scene_camera("bistro")
    view = trx.view_from_camera(eye, look, fov, 40, 24)
    try:
        off = trx.flat_build(verts, counts, reinsertion=0.0)
        on = trx.flat_build(verts, counts, reinsertion=setting)
        again = trx.flat_build(verts, counts, reinsertion=setting, threads=1)
    finally:
        trx.flat_build(verts[:1], reinsertion=(0.02, 4))  # back to the library default
    assert (on.nodes == again.nodes).all() and (on.tri_source == again.tri_source).all()
    assert not (on.nodes.shape == off.nodes.shape and (on.nodes == off.nodes).all())
    assert sorted(on.tri_source.tolist()) == list(range(verts.shape[0]))
    so, sn = orc.Scene.from_flat(off), orc.Scene.from_flat(on)
    assert sn.validate()[0] == 0
    ov = orc.view_from_bytes(view)
    want = sn.brute_primary(ov, 40, 24, sem=3)
    got, st_on = sn.trace_primary(ov, 40, 24, sem=3)
    _, st_off = so.trace_primary(ov, 40, 24, sem=3)
    assert (got["t"] == want["t"]).all()
    hit = want["prim"] != 0xffffffff
    assert (hit == (got["prim"] != 0xffffffff)).all()
    # the pass exists to cut traversal work; allow noise on a 960-ray view but not a regression
    assert st_on.n_node + st_on.n_tri < 1.1 * (st_off.n_node + st_off.n_tri)


def test_reinsertion_rejects_bad_arguments(trx):
    lib = trx._lib.load()
    assert lib.trx_set_build_reinsertion(-0.1, 2) != 0
    assert lib.trx_set_build_reinsertion(1.5, 2) != 0
    assert lib.trx_set_build_reinsertion(0.1, -1) != 0
    assert lib.trx_set_build_reinsertion(0.02, 4) == 0


def test_build_presets(trx, orc):
    """trx_set_build_preset: the reference's --preset names (src/main.rs:125-131,563-570) select build-time /
    quality settings of this builder; every one must give a sound tree that answers like brute force."""
    verts, counts = trx.gen_scene("kitchen", 12000, 2)
    eye, look, fov = trx.scene_camera("kitchen")
    ov = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, 32, 20))
    work = {}
    try:
        for preset in ("fastest_build", "very_fast_build", "fast_build", "medium_build", "slow_build", "very_slow_build"):
            flat = trx.flat_build(verts, counts, preset=preset)
            osc = orc.Scene.from_flat(flat)
            assert osc.validate(boxes=flat.tri_boxes)[0] == 0, preset   # slow presets pre-split: entries are clipped
            got, st = osc.trace_primary(ov, 32, 20, sem=3)
            assert (got["t"] == osc.brute_primary(ov, 32, 20, sem=3)["t"]).all(), preset
            work[preset] = st.n_node
        default = trx.flat_build(verts, counts, preset="")
        assert (default.nodes == trx.flat_build(verts, counts, preset="medium_build").nodes).all()
        with pytest.raises(trx.TrxError, match="unknown preset"):
            trx.flat_build(verts, counts, preset="warp_speed")
    finally:
        trx.flat_build(verts[:1], preset="")
    assert work["very_slow_build"] < work["fastest_build"]   # the slow end buys fewer node visits


def test_pre_split_references_cover_their_triangles(trx, orc):
    """trx_set_build_split (obvhs pre_split, the reference's --split): big, mostly empty triangle boxes are cut
    into clipped references.  Every reference box must lie inside its leaf box (validator), the references of a
    triangle together must cover it (sampled), every triangle must still be there, and rays must see the same
    geometry as brute force."""
    verts, counts = trx.gen_scene("kitchen", 15000, 4)
    try:
        plain = trx.flat_build(verts, counts, split=0.0)
        flat = trx.flat_build(verts, counts, split=0.5)
        again = trx.flat_build(verts, counts, split=0.5, threads=1)
    finally:
        trx.flat_build(verts[:1], split=0.0)
    n = verts.shape[0]
    assert (flat.nodes == again.nodes).all() and (flat.tri_source == again.tri_source).all()
    assert n < flat.n_tris <= int(1.5 * n) + 1 and plain.n_tris == n
    assert set(flat.tri_source.tolist()) == set(range(n))
    assert (flat.tri_verts == verts.reshape(-1, 9)[flat.tri_source]).all()      # entries repeat whole triangles
    osc = orc.Scene.from_flat(flat)
    assert osc.validate(boxes=flat.tri_boxes)[0] == 0
    assert osc.validate()[0] != 0            # whole triangles no longer fit their leaf boxes: that is the point
    # coverage: sample every split triangle, each sample must fall into one of its references' boxes
    refs = {}
    for k, src in enumerate(flat.tri_source.tolist()):
        refs.setdefault(src, []).append(k)
    split = [t for t, r in refs.items() if len(r) > 1]
    assert len(split) > 100
    g = np.array([(a, b, 1.0 - a - b) for a in np.linspace(0, 1, 9) for b in np.linspace(0, 1 - a, 9)], dtype=np.float64)
    for t in split[:400]:
        tri = verts.reshape(-1, 3, 3)[t].astype(np.float64)
        pts = g @ tri
        boxes = flat.tri_boxes[refs[t]].astype(np.float64)
        inside = ((pts[:, None, :] >= boxes[None, :, 0:3] - 1e-9) & (pts[:, None, :] <= boxes[None, :, 3:6] + 1e-9)).all(2)
        assert inside.any(1).all(), "triangle %d is not covered by its %d references" % (t, len(refs[t]))
    # same geometry for rays: t as brute force finds it (up to the slab-rounding rays of any CWBVH), fewer tests
    eye, look, fov = trx.scene_camera("kitchen")
    ov = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, 96, 54))
    got, st = osc.trace_primary(ov, 96, 54, sem=3)
    bf = osc.brute_primary(ov, 96, 54, sem=3)
    differ = got["t"].view(np.uint32) != bf["t"].view(np.uint32)
    assert differ.sum() <= 2 and (got["t"][differ] >= bf["t"][differ]).all()
    _, st0 = orc.Scene.from_flat(plain).trace_primary(ov, 96, 54, sem=3)
    assert st.n_tri < st0.n_tri


def test_build_params_struct_mirrors_the_reference(trx, orc):
    """trx_build_params = BvhBuildParams (src/main.rs:571-585): one build with the caller's params, the
    process-wide settings untouched afterwards; the reference's own argument errors are kept."""
    verts, counts = trx.gen_scene("kitchen", 8000, 1)
    before = trx.flat_build(verts, counts)
    bp = trx.build_params()
    assert (bp.pre_split, bp.ploc_search_distance, bp.search_depth_threshold, bp.sort_precision, bp.max_prims_per_leaf) == \
        (0, 14, 2, 64, 3)                                                       # src/main.rs:85-124
    assert abs(bp.reinsertion_batch_ratio - 0.15) < 1e-7 and bp.collapse_traversal_cost == 1.0
    a = trx.flat_build_params(verts, counts, bp)
    assert a.n_tris == 8000 and orc.Scene.from_flat(a).validate()[0] == 0
    b = trx.flat_build_params(verts, counts, trx.build_params(pre_split=1, reinsertion_batch_ratio=0.0, max_prims_per_leaf=2,
                                                              collapse_traversal_cost=2.0), use_tlas=True)
    assert b.n_tris > 8000 and b.has_tlas and orc.Scene.from_flat(b).validate(boxes=b.tri_boxes)[0] == 0
    with pytest.raises(trx.TrxError, match="Unsupported sort precision"):         # src/main.rs:576-580
        trx.flat_build_params(verts, counts, trx.build_params(sort_precision=32))
    with pytest.raises(trx.TrxError, match="maximum of 3 primitives"):            # src/main.rs:176-178
        trx.flat_build_params(verts, counts, trx.build_params(max_prims_per_leaf=8))
    assert (trx.flat_build(verts, counts).nodes == before.nodes).all()


@pytest.mark.parametrize("name,n,tlas", [("kitchen", 9000, False), ("bistro", 40000, True), ("hairball", 20000, False),
                                          ("soup", 3000, False), ("demoscene", 20000, False)])
def test_ploc_build_honours_the_reference_parameters(trx, orc, name, n, tlas):
    """The ploc_cwbvh pipeline behind trx_flat_build_params: PLOC (Meister & Bittner 2018) over a Morton order of
    sort_precision bits, merging within ploc_search_distance places (distance 1 for the first
    search_depth_threshold rounds), then reinsertion and the 8-wide collapse.  Every setting yields a valid tree
    whose hits equal the brute-force query; the PLOC parameters really change the tree (src/main.rs:571-585)."""
    verts, counts = trx.gen_scene(name, n, 1)
    eye, look, fov = trx.scene_camera(name)
    view = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, 40, 24))
    want = None
    trees = {}
    for dist, thresh, bits, ratio in [(14, 2, 64, 0.15), (1, 0, 64, 0.0), (2, 0, 64, 0.0), (32, 5, 64, 0.05), (14, 2, 128, 0.0),
                                      (14, 0, 64, 0.0), (6, 1, 128, 1.5)]:
        bp = trx.build_params(ploc_search_distance=dist, search_depth_threshold=thresh, sort_precision=bits,
                              reinsertion_batch_ratio=ratio)
        flat = trx.flat_build_params(verts, counts, bp, use_tlas=tlas)
        assert flat.n_tris == n and sorted(flat.tri_source.tolist()) == list(range(n))
        osc = orc.Scene.from_flat(flat)
        assert osc.validate() == (0, ""), (dist, thresh, bits, ratio)
        got, _ = osc.trace_primary(view, 40, 24)
        if want is None:
            want = osc.brute_primary(view, 40, 24)
        assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), (dist, thresh, bits, ratio)
        trees[(dist, thresh, bits, ratio)] = flat.nodes.tobytes()
    # the search distance, the depth threshold and the sort precision each matter
    assert trees[(1, 0, 64, 0.0)] != trees[(2, 0, 64, 0.0)] != trees[(14, 0, 64, 0.0)]
    assert trees[(14, 0, 64, 0.0)] != trees[(14, 2, 128, 0.0)]
    # and the build is the same whatever the thread count
    bp = trx.build_params(reinsertion_batch_ratio=0.05)
    a = trx.flat_build_params(verts, counts, bp, use_tlas=tlas, threads=1)
    b = trx.flat_build_params(verts, counts, bp, use_tlas=tlas, threads=5)
    assert (a.nodes == b.nodes).all() and (a.tri_source == b.tri_source).all()


def test_ploc_parameter_errors_and_tiny_inputs(trx, orc):
    verts, counts = trx.gen_scene("soup", 64, 3)
    for bad in (0, 33):
        with pytest.raises(trx.TrxError, match="ploc_search_distance"):
            trx.flat_build_params(verts, counts, trx.build_params(ploc_search_distance=bad))
    for n in (1, 2, 3, 5):
        flat = trx.flat_build_params(verts[:n], np.array([n], dtype=np.uint64), trx.build_params(reinsertion_batch_ratio=0.0))
        assert flat.n_tris == n and orc.Scene.from_flat(flat).validate() == (0, "")
    # identical centroids (one Morton cell): PLOC still terminates and the tree is valid
    same = np.repeat(verts[:1], 50, axis=0)
    flat = trx.flat_build_params(same, np.array([50], dtype=np.uint64), trx.build_params())
    assert orc.Scene.from_flat(flat).validate() == (0, "")


def test_batched_reinsertion_improves_the_ploc_tree(trx, orc):
    """The ploc_cwbvh pipeline runs the reinsertion pass in its parallel formulation (batches of candidates search the
    same tree, moves applied in order, stale ones skipped): the tree stays valid, the hits unchanged, and the node
    visits of a frame drop clearly against PLOC alone."""
    verts, counts = trx.gen_scene("bistro", 150000, 1)
    counts = np.array([150000], dtype=np.uint64)
    eye, look, fov = trx.scene_camera("bistro")
    view = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, 160, 90))
    stats, hits = {}, {}
    for ratio in (0.0, 0.15):
        flat = trx.flat_build_params(verts, counts, trx.build_params(reinsertion_batch_ratio=ratio))
        osc = orc.Scene.from_flat(flat)
        assert osc.validate() == (0, "")
        got, st = osc.trace_primary(view, 160, 90)
        stats[ratio] = st.n_node
        hits[ratio] = flat.tri_source[np.where(got["prim"] != 0xFFFFFFFF, got["prim"], 0)], got["t"].copy()
    assert stats[0.15] < 0.9 * stats[0.0], stats
    assert (hits[0.0][1].view(np.uint32) == hits[0.15][1].view(np.uint32)).all()   # same closest hits through either tree


def test_whole_iteration_reinsertion_is_deterministic_and_valid(trx, orc):
    """trx_set_build_reinsertion_batches(1): every candidate of an iteration searches the tree the previous iteration left
    (host searches here; the device searches of tests/test_gpu_builder.py must reproduce these bytes).  The same tree on
    one core and on five, a valid CWBVH, and not the tree of the default batching."""
    lib = trx.load()
    verts, counts = trx.gen_scene("bistro", 60000, 1)
    counts = np.array([verts.shape[0]], dtype=np.uint64)
    try:
        assert lib.trx_set_build_reinsertion_batches(1) == 0
        for build in (lambda t: trx.flat_build_params(verts, counts, trx.build_params(), threads=t),
                      lambda t: trx.flat_build(verts, counts, preset="medium_build", reinsertion=(0.05, 6), threads=t)):
            a, b = build(1), build(5)
            assert (a.nodes == b.nodes).all() and (a.tri_source == b.tri_source).all()
            assert orc.Scene.from_flat(a).validate() == (0, "")
        whole = trx.flat_build_params(verts, counts, trx.build_params(), threads=2)
        assert lib.trx_set_build_reinsertion_batches(0) == 0
        small = trx.flat_build_params(verts, counts, trx.build_params(), threads=2)
        assert whole.nodes.shape != small.nodes.shape or not (whole.nodes == small.nodes).all()
    finally:
        lib.trx_set_build_reinsertion_batches(0)
        lib.trx_set_build_preset(b"medium_build")
