"""CPU tests of the oracle: golden fixtures, brute-force cross-check, and the
individual functions of the path restated independently in numpy / Python."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from helpers import ALL_SEMS, F32_MAX, assert_hits_equal, bits, golden_inputs, make_scene, random_rays

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(O, name):
    import tray_racing_amd as T
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    nodes, tri_verts, inst, tlas_start = golden_inputs(T, g)
    osc = O.Scene(nodes, tri_verts, inst, tlas_start)
    return g, osc


def assert_matches_bruteforce(osc, rays, got, bf, sem):
    """t must be bit-equal everywhere; prim may differ only inside an exact tie
    (another triangle giving the very same t), because the BVH's visit order,
    not the index order, resolves ties."""
    assert (bits(got["t"]) == bits(bf["t"])).all()
    diff = np.flatnonzero(got["prim"] != bf["prim"])
    for i in diff:
        t = osc.tri_t(rays["origin"][i], rays["direction"][i], int(got["prim"][i]), sem=0)
        assert t is not None and np.float32(t) == got["t"][i], "ray %d: prim %d is not a tie" % (i, got["prim"][i])
    return diff.size


@pytest.mark.parametrize("name", ["cornell_64", "cornell_tlas_48", "soup_52x44", "box14_tlas_48", "ref_cornell_box_64", "ref_box_tlas_48"])
def test_golden_images(orc, name):
    g, osc = load_golden(orc, name)
    w, h = int(g["width"]), int(g["height"])
    view = orc.view_from_bytes(g["view"].tobytes())
    rays = osc.primary_rays(view, w, h)
    for sem in (0, 3):
        prim, st = osc.trace_primary(view, w, h, sem=sem)
        assert_hits_equal(prim, g["orc_primary_sem%d" % sem], "%s sem %d primary" % (name, sem))
        assert [st.n_node, st.n_tri, st.n_hits, st.max_stack] == list(g["orc_counts_sem%d" % sem])
        ao, _ = osc.trace_ao(view, w, h, prim, sem=sem, frame=2, ao_eps=0.01)
        assert_hits_equal(ao, g["orc_ao_sem%d" % sem], "%s sem %d ao" % (name, sem))
        assert_matches_bruteforce(osc, rays, prim, g["bf_primary_sem%d" % sem], sem)
        assert st.n_hits > (0.05 if name == "ref_box_tlas_48" else 0.2) * w * h   # box.ron looks past the box


def test_golden_ties(orc):
    g, osc = load_golden(orc, "ties_rays")
    rays = g["rays"]
    n_ties = 0
    for sem in (0, 3):
        hits, _ = osc.trace_rays(rays, sem=sem)
        assert_hits_equal(hits, g["orc_rays_sem%d" % sem], "ties sem %d" % sem)
        bf = osc.brute_rays(rays, sem=sem)
        assert_hits_equal(bf, g["bf_rays_sem%d" % sem], "ties bf sem %d" % sem)
        n_ties += assert_matches_bruteforce(osc, rays, hits, bf, sem)
    # the two tie rules must disagree somewhere on this scene, or the fixture tests nothing
    assert (g["orc_rays_sem0"]["prim"] != g["orc_rays_sem3"]["prim"]).any()
    assert (g["bf_rays_sem0"]["prim"] != g["bf_rays_sem3"]["prim"]).any()


@pytest.mark.parametrize("name,n,w,h,tlas", [("cornell", 0, 40, 40, False), ("kitchen", 6000, 64, 40, False),
                                              ("kitchen", 6000, 40, 32, True), ("bistro", 30000, 64, 36, False),
                                              ("soup", 800, 32, 32, False), ("demoscene", 5000, 32, 64, False)])
def test_oracle_vs_bruteforce(trx, orc, name, n, w, h, tlas):
    flat, _view, osc, ov = make_scene(trx, orc, name, n, w, h, tlas=tlas)
    rays = osc.primary_rays(ov, w, h)
    for sem in ALL_SEMS:
        got, st = osc.trace_primary(ov, w, h, sem=sem)
        assert st.overflow == 0
        assert_matches_bruteforce(osc, rays, got, osc.brute_primary(ov, w, h, sem=sem), sem)
    rr = random_rays(trx, flat, 600, 3)
    for sem in (0, 3):
        got, _ = osc.trace_rays(rr, sem=sem)
        bf = osc.brute_rays(rr, sem=sem)
        # the node test clamps box entry to 1e-4 (query.hlsl:275,288): hits closer than
        # that are invisible to the BVH by design, the brute force sees them
        near = bf["t"] < 2e-4
        assert near.sum() < 5
        assert_matches_bruteforce(osc, rr[~near], got[~near], bf[~near], sem)


def test_shards_partition_the_image(trx, orc):
    _flat, _v, osc, ov = make_scene(trx, orc, "cornell", 0, 52, 44)
    full, _ = osc.trace_primary(ov, 52, 44, sem=3)
    acc = np.zeros(52 * 44, dtype=orc.HIT_DTYPE)
    touched = np.zeros(52 * 44, dtype=np.int32)
    for r in range(3):
        part = np.zeros(52 * 44, dtype=orc.HIT_DTYPE)
        part["prim"] = 12345
        osc.trace_primary(ov, 52, 44, sem=3, shard=(r, 3), out=part)
        m = part["prim"] != 12345
        touched += m
        acc[m] = part[m]
    assert (touched == 1).all()
    assert_hits_equal(acc, full, "union of shards")


# ---- the functions of the path, one by one ------------------------------------------------

def py_uhash(a, b):  # src/rt_gpu/sampling.hlsl:5-15 in Python integers
    M = 0xFFFFFFFF
    x = ((a * 1597334673) & M) ^ ((b * 3812015801) & M)
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M
    x ^= x >> 15
    x = (x * 0x846CA68B) & M
    x ^= x >> 16
    return x


def test_uhash_and_hash_noise(orc):
    lib = orc.load()
    rng = np.random.default_rng(0)
    for a, b in rng.integers(0, 2**32, size=(200, 2), dtype=np.uint64):
        assert lib.orc_uhash(int(a), int(b)) == py_uhash(int(a), int(b))
    for x, y, f in [(0, 0, 0), (1919, 1079, 0), (7, 3, 1024), (512, 999, 3)]:
        want = np.float32(py_uhash(x, ((y << 11) + f) & 0xFFFFFFFF)) * np.float32(1.0 / 4294967296.0)
        got = lib.orc_hash_noise(x, y, f)
        assert np.float32(got) == want and 0.0 <= got <= 1.0


def test_sincos_accuracy(orc):
    lib = orc.load()
    s, c = C.c_float(), C.c_float()
    th = np.linspace(0.0, 2 * np.pi, 4001, dtype=np.float32)
    err = 0.0
    for t in th:
        lib.orc_sincos(float(t), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(t))), abs(c.value - np.cos(np.float64(t))))
    assert err < 6.1e-8   # within 0.51 ulp of binary32 at 1.0 (it was 2.5e-7 for the binary32 polynomial of rounds 1-3)


def test_sincos_is_the_c_librarys_sinf_and_cosf_bit_for_bit(orc):
    """orc_sincos evaluates the binary64 algorithm glibc (>= 2.28) publishes for sinf / cosf, so on a glibc host the AO
    directions are those of the reference's CPU path (Rust's f32::sin / cos call the C library).  Every binary32 in
    [0, 2 pi] - the whole domain of theta = u2 * tau, 1.09e9 values, a few seconds on all cores."""
    import platform
    name, ver = platform.libc_ver()
    if name != "glibc":
        pytest.skip("the C library here is not glibc")
    if tuple(int(x) for x in ver.split(".")[:2]) < (2, 28):
        pytest.skip("glibc %s predates the sinf / cosf algorithm this restates (2.28)" % ver)
    if platform.machine() not in ("x86_64", "AMD64"):
        pytest.skip("measured identical on x86_64 only (glibc picks its sinf variant per architecture)")
    lib = orc.load()
    hi = int(np.float32(6.28318530717958647692).view(np.uint32))
    assert lib.orc_sincos_libm_mismatches(0, hi + 64, 1) == 0


def test_octant(orc):
    lib = orc.load()
    for d, want in [((1, 1, 1), 0x07070707), ((-1, 1, 1), 0x03030303), ((1, -1, -1), 0x04040404),
                    ((-1, -1, -1), 0), ((0.0, -0.0, 1), 0x07070707)]:
        a = np.array(d, dtype=np.float32)
        assert lib.orc_octant_inv4(a.ctypes.data_as(C.c_void_p)) == want


def tri_test(orc, o, d, tri9, tmin=0.0, t0=F32_MAX, sem=0):
    lib = orc.load()
    o = np.array(o, dtype=np.float32)
    d = np.array(d, dtype=np.float32)
    tri = np.array(tri9, dtype=np.float32)
    t = np.array([t0], dtype=np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    ok = lib.orc_intersect_tri(P(o), P(d), P(tri), tmin, P(t), sem)
    return bool(ok), float(t[0])


def test_triangle_test_rules(orc):
    # triangle v0=(0,0,0) v1=(1,0,0) v2=(0,1,0): stored {v0, e1 = v0-v1, e2 = v2-v0}
    tri = [0, 0, 0, -1, 0, 0, 0, 1, 0]
    ok, t = tri_test(orc, (0.25, 0.25, 2), (0, 0, -1), tri)
    assert ok and t == 2.0
    ok, t = tri_test(orc, (0.25, 0.25, -2), (0, 0, 1), tri)  # no back-face culling (query.hlsl:94)
    assert ok and t == 2.0
    assert not tri_test(orc, (0.75, 0.75, 2), (0, 0, -1), tri)[0]       # w < 0
    assert not tri_test(orc, (0.25, 0.25, 2), (0, 0, 1), tri)[0]        # behind: tt < 0
    assert not tri_test(orc, (0.25, 0.25, 2), (1, 0, 0), tri)[0]        # parallel: det == 0
    # the sign-bit test (query.hlsl:111-116): a ray through the edge x = 0 has u = +0 from above
    # (accepted) but u = 0 * inv_det(-1) = -0.0 from below (rejected)
    assert tri_test(orc, (0.0, 0.5, 1), (0, 0, -1), tri)[0]
    assert not tri_test(orc, (0.0, 0.5, -1), (0, 0, 1), tri)[0]
    # range and tie rules
    assert tri_test(orc, (0.25, 0.25, 2), (0, 0, -1), tri, t0=2.0, sem=0)[0]       # tt <= t commits
    assert not tri_test(orc, (0.25, 0.25, 2), (0, 0, -1), tri, t0=2.0, sem=2)[0]   # tt < t does not
    assert not tri_test(orc, (0.25, 0.25, 2), (0, 0, -1), tri, tmin=2.5)[0]
    assert not tri_test(orc, (0.25, 0.25, 2), (0, 0, -1), [0] * 9)[0]              # zero-area triangle


def test_f16_triangles_decode(trx, orc):
    verts, _ = trx.gen_scene("soup", 200, 3)
    packed = trx.pack_tris_f16(verts)
    osc = orc.Scene(np.zeros((1, 20), np.uint32), tri_f16=packed)
    v = verts.reshape(-1, 9)
    e1 = (v[:, 3:6] - v[:, 0:3]).astype(np.float16).astype(np.float32)
    e2 = (v[:, 6:9] - v[:, 0:3]).astype(np.float16).astype(np.float32)
    assert (osc.tris[:, 0:3] == v[:, 0:3]).all()
    assert (osc.tris[:, 3:6] == -e1).all() and (osc.tris[:, 6:9] == e2).all()


def test_primary_rays_match_float64_restatement(trx, orc):
    eye, look, fov = trx.scene_camera("bistro")
    w, h = 48, 27
    view = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, w, h))
    ov2 = orc.view_from_camera(eye, look, fov, w, h)  # the oracle's own camera (src/main.rs:602-616)
    assert np.allclose(np.array(view.view_inv), np.array(ov2.view_inv), atol=2e-6)
    assert np.allclose(np.array(view.proj_inv), np.array(ov2.proj_inv), rtol=1e-5, atol=1e-6)
    rays = orc.Scene(np.zeros((1, 20), np.uint32), np.zeros((1, 9), np.float32)).primary_rays(view, w, h)
    VI = np.array(view.view_inv, dtype=np.float64).reshape(4, 4).T
    PI = np.array(view.proj_inv, dtype=np.float64).reshape(4, 4).T
    f = np.array(look, float) - np.array(eye, float)
    f /= np.linalg.norm(f)
    for i in (0, w - 1, w * h // 2 + 5, w * h - 1):
        px, py = i % w, i // w
        clip = np.array([px / w * 2 - 1, (1 - py / h) * 2 - 1, 1, 1.0])
        vs = PI @ clip
        vs /= vs[3]
        d = (VI @ vs)[:3] - np.array(eye)
        d /= np.linalg.norm(d)
        # the reference's formulation subtracts the eye from a point 0.01 away from it (near
        # plane), so f32 cancellation limits the direction to ~1e-4 of the f64 value
        assert np.allclose(rays["direction"][i], d, atol=2e-4)
        assert abs(np.linalg.norm(rays["direction"][i]) - 1) < 1e-6
    centre = rays["direction"].reshape(h, w, 3)[h // 2, w // 2]
    assert np.dot(centre, f) > 0.99  # image centre looks along the camera axis; +y is up, y grows downwards
    assert rays["direction"].reshape(h, w, 3)[0, w // 2][1] > rays["direction"].reshape(h, w, 3)[h - 1, w // 2][1]


def test_ao_rays(trx, orc):
    flat, _v, osc, ov = make_scene(trx, orc, "cornell", 0, 32, 32)
    prim, _ = osc.trace_primary(ov, 32, 32, sem=0)
    lib = orc.load()
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rays = osc.primary_rays(ov, 32, 32)
    checked = 0
    for i in range(0, 32 * 32, 7):
        h = orc.HitC(float(prim["t"][i]), int(prim["prim"][i]))
        ok = lib.orc_ao_ray(C.byref(osc.c), C.byref(ov), 32, 32, i % 32, i // 32, h, 5, 0.01, P(o), P(d))
        if not np.isfinite(prim["t"][i]):
            assert ok == 0
            continue
        assert ok == 1
        tri = osc.tris[prim["prim"][i]].astype(np.float64)
        n = np.cross(tri[3:6], tri[6:9])
        n /= np.linalg.norm(n)
        rd = rays["direction"][i].astype(np.float64)
        if np.dot(n, -rd) < 0:
            n = -n
        assert abs(np.linalg.norm(d) - 1) < 1e-6
        assert np.dot(d, n) > -1e-6                       # cosine hemisphere around the viewer-facing normal
        hitp = rays["origin"][i] + rd * prim["t"][i]
        assert np.allclose(o, hitp - rd * 0.01, atol=1e-5)  # origin offset, src/rt_cpu/rt_cpu.rs:67
        checked += 1
    assert checked > 50
    ao0, _ = osc.trace_ao(ov, 32, 32, prim, frame=0)
    ao1, _ = osc.trace_ao(ov, 32, 32, prim, frame=1)
    assert (bits(ao0["t"]) != bits(ao1["t"])).any()       # the frame index seeds the sample


def decode_children(node):
    nb = node.view(np.uint8)
    p = node[:3].view(np.float32).astype(np.float64)
    e = np.array([2.0 ** (int(nb[12 + k]) - 127) for k in range(3)])
    q = nb[32:80].reshape(6, 8).astype(np.float64)  # minx maxx miny maxy minz maxz
    lo = p[:, None] + q[0::2] * e[:, None]
    hi = p[:, None] + q[1::2] * e[:, None]
    return nb[24:32], nb[15], lo, hi


def test_node_intersect_against_float64_slabs(trx, orc):
    """hit_mask bits must correspond to children whose decoded boxes the ray enters
    within [1e-4, tmax] (clear-cut cases only: the f32 test may round at razor edges)."""
    flat, _v, _osc, _ov = make_scene(trx, orc, "soup", 600, 8, 8)
    lib = orc.load()
    rng = np.random.default_rng(4)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    checked = 0
    for ni in range(min(flat.n_nodes, 40)):
        node = np.ascontiguousarray(flat.nodes[ni])
        meta, _imask, lo, hi = decode_children(node)
        for _ in range(25):
            o = rng.uniform(-3, 3, 3).astype(np.float32)
            d = rng.normal(size=3).astype(np.float32)
            d /= np.linalg.norm(d)
            inv = (np.float32(1.0) / d).astype(np.float32)
            oct4 = lib.orc_octant_inv4(P(d))
            tmax = np.float32(rng.uniform(0.5, 8))
            for sem in (0, 1, 4, 5):
                mask = lib.orc_node_intersect(P(o), P(d), P(inv), oct4, float(tmax), P(node), sem)
                for s in range(8):
                    m = int(meta[s])
                    if m == 0:
                        continue
                    t0 = (lo[:, s] - o) / d
                    t1 = (hi[:, s] - o) / d
                    tn = max(np.minimum(t0, t1).max(), 1e-4)
                    tf = min(np.maximum(t0, t1).min(), float(tmax))
                    if abs(tn - tf) < 1e-4 * max(1.0, abs(tf)):
                        continue
                    inner = (m & 0x18) == 0x18
                    bit = (24 + (s ^ (oct4 & 7))) if inner else (m & 0x1F)
                    got = (mask >> bit) & 1
                    assert got == (1 if tn <= tf else 0), (ni, s, tn, tf)
                    checked += 1
    assert checked > 2000


def test_validator_catches_corruption(trx, orc):
    flat, _v, osc, _ov = make_scene(trx, orc, "kitchen", 3000, 8, 8)
    assert osc.validate() == (0, "")
    bad = flat.nodes.copy()
    b8 = bad.view(np.uint8).reshape(-1, 80)
    slot = int(np.flatnonzero(b8[0, 24:32])[0])
    b8[0, 32 + 8 + slot] = b8[0, 32 + slot]  # collapse child max_x onto min_x of the root's first child
    rc, msg = orc.Scene(bad, flat.tri_verts).validate()
    assert rc != 0 and "quantised box" in msg
    bad = flat.nodes.copy()
    meta = bad.view(np.uint8).reshape(-1, 80)[:, 24:32]
    leafy = np.flatnonzero(((meta != 0) & ((meta & 0x18) != 0x18)).any(axis=1))
    bad[leafy[0], 5] += 1  # primitive_base_idx of the first node that owns triangles
    rc, msg = orc.Scene(bad, flat.tri_verts).validate()
    assert rc != 0


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/obj/cornell_box.obj"), reason="reference assets absent")
@pytest.mark.parametrize("scene", ["cornell_box", "box"])
def test_reference_assets_oracle_vs_bruteforce(trx, orc, scene):
    """The only real assets the reference ships (assets/obj/{box,cornell_box}.obj), with the
    cameras of assets/scenes/*.ron.  Read in place; nothing is copied into the repo."""
    cams = {"cornell_box": ((0.0, 1.0, 2.1), (0.0, 1.0, 0.0), 90.0),
            "box": ((3.0, 1.5, 1.4), (-3.9438584, 1.5, -1.7303504), 90.0)}
    verts, counts = trx.load_meshs("/root/reference/assets/obj/%s.obj" % scene)
    assert verts.shape[0] == {"cornell_box": 3968, "box": 14}[scene]
    assert len(counts) == {"cornell_box": 5, "box": 2}[scene]
    w, h = 64, 48
    for tlas in (False, True):
        flat = trx.flat_build(verts, counts, use_tlas=tlas)
        osc = orc.Scene.from_flat(flat)
        assert osc.validate() == (0, "")
        eye, look, fov = cams[scene]
        ov = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, w, h))
        rays = osc.primary_rays(ov, w, h)
        for sem in (0, 3):
            got, st = osc.trace_primary(ov, w, h, sem=sem)
            assert st.n_hits > 100
            assert_matches_bruteforce(osc, rays, got, osc.brute_primary(ov, w, h, sem=sem), sem)


def test_golden_f16_tlas_rays(orc):
    g, osc = load_golden(orc, "kitchen_tlas_f16_56x40")
    w, h = int(g["width"]), int(g["height"])
    view = orc.view_from_bytes(g["view"].tobytes())
    osc16 = orc.Scene(g["nodes"], None, g["instance_offsets"], int(g["tlas_start"]), tri_f16=g["tri_f16"])
    for sem in (0, 3):
        prim, _ = osc16.trace_primary(view, w, h, sem=sem)
        assert_hits_equal(prim, g["orc_f16_primary_sem%d" % sem], "f16 primary sem %d" % sem)
        ao, _ = osc16.trace_ao(view, w, h, prim, sem=sem, frame=9, ao_eps=0.0001)
        assert_hits_equal(ao, g["orc_f16_ao_eps1e-4_sem%d" % sem], "f16 ao sem %d" % sem)
        got, _ = osc.trace_rays(g["rays"], sem=sem)
        assert_hits_equal(got, g["orc_rays_sem%d" % sem], "tlas rays sem %d" % sem)
        bf = g["bf_rays_sem%d" % sem]
        near = bf["t"] < 2e-4
        assert_matches_bruteforce(osc, g["rays"][~near], got[~near], bf[~near], sem)


def test_fixed_seed_slice_of_the_oracle_fuzzer():
    """tests/fuzz_oracle.py: random scenes / builds / rays, CWBVH traversal against brute force.  They may
    differ only the way the reference algorithm itself does (slab rounding on box faces): never a CLOSER
    hit from the BVH, a few ulps at most when both hit, and rarely at all."""
    import fuzz_oracle
    st = fuzz_oracle.run(minutes=2.0, seed=5, max_cases=400, n_rays=1500, verbose=False)
    assert st["cases"] == 400 and st["closer"] == 0 and st["worst_rel"] < 1e-6
    assert st["differing"] + st["missed"] < 1e-4 * st["rays"]


def test_simd_node_test_is_bit_identical_to_the_scalar_one(trx, orc):
    """The AVX2 form of the node test (the one bench.py's cpu_baseline leg times, because obvhs' CPU node test is SIMD
    too) against the scalar restatement: the node masks themselves on random and on adversarial inputs (zero / denormal
    / huge direction components, which make 0 x inf planes), every golden frame, random rays under all eight
    semantics, and a whole primary + AO frame with its node / triangle counts."""
    if not orc.set_simd(True):
        orc.set_simd(False)
        pytest.skip("this CPU has no AVX2 + FMA")
    try:
        lib = orc.load()
        flat, view, osc, ov = make_scene(trx, orc, "bistro", 40000, 160, 90)
        nodes = np.ascontiguousarray(flat.nodes).view(np.uint32).reshape(-1, 20)
        rng = np.random.default_rng(11)
        pts = flat.tri_verts.reshape(-1, 3)
        lo, hi = pts.min(0), pts.max(0)
        special = np.array([0.0, -0.0, 1e-42, -1e-42, 1.1920929e-7, 3e38, -3e38, 1.0, -1.0], dtype=np.float32)
        n_checked = 0
        for k in range(6000):
            o = rng.uniform(lo, hi).astype(np.float32)
            d = rng.normal(size=3).astype(np.float32)
            if k % 3 == 0:
                d[rng.integers(3)] = special[rng.integers(len(special))]
            if k % 7 == 0:
                d[:] = special[rng.integers(len(special), size=3)]
            d = np.where(d == 0, np.float32(1.1920929e-7), d).astype(np.float32)   # the zero-direction fix runs before the node test
            with np.errstate(divide="ignore", over="ignore"):
                inv = (np.float32(1.0) / d).astype(np.float32)
            oct4 = lib.orc_octant_inv4(d.ctypes.data_as(C.c_void_p))
            node = np.ascontiguousarray(nodes[rng.integers(len(nodes))])
            tmax = np.float32(rng.choice([3.4028234663852886e38, 1.0, 10.0, 0.5]))
            for sem in ALL_SEMS:
                args = (o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p), oct4,
                        float(tmax), node.ctypes.data_as(C.c_void_p), sem)
                orc.set_simd(False)
                want = lib.orc_node_intersect(*args)
                orc.set_simd(True)
                assert lib.orc_node_intersect(*args) == want, (k, sem, o, d)
                n_checked += 1
        assert n_checked == 6000 * 8
        # goldens
        for name in ["cornell_64", "cornell_tlas_48", "soup_52x44"]:
            g, gsc = load_golden(orc, name)
            w, h = int(g["width"]), int(g["height"])
            gv = orc.view_from_bytes(g["view"].tobytes())
            for sem in (0, 3):
                prim, st = gsc.trace_primary(gv, w, h, sem=sem)
                assert_hits_equal(prim, g["orc_primary_sem%d" % sem], "%s sem %d primary (simd)" % (name, sem))
                assert [st.n_node, st.n_tri, st.n_hits, st.max_stack] == list(g["orc_counts_sem%d" % sem])
                ao, _ = gsc.trace_ao(gv, w, h, prim, sem=sem, frame=2, ao_eps=0.01)
                assert_hits_equal(ao, g["orc_ao_sem%d" % sem], "%s sem %d ao (simd)" % (name, sem))
        g, gsc = load_golden(orc, "ties_rays")
        for sem in (0, 3):
            hits, _ = gsc.trace_rays(g["rays"], sem=sem)
            assert_hits_equal(hits, g["orc_rays_sem%d" % sem], "ties sem %d (simd)" % sem)
        # random rays (zero direction components included) under every semantics, and a whole frame
        rays = random_rays(trx, flat, 4000, 5)
        for sem in ALL_SEMS:
            orc.set_simd(False)
            want, wst = osc.trace_rays(rays, sem=sem)
            orc.set_simd(True)
            got, gst = osc.trace_rays(rays, sem=sem)
            assert_hits_equal(got, want, "random rays sem %d (simd)" % sem)
            assert (gst.n_node, gst.n_tri) == (wst.n_node, wst.n_tri)
        orc.set_simd(False)
        want, wst = osc.trace_primary(ov, 160, 90, sem=3)
        want_ao, _ = osc.trace_ao(ov, 160, 90, want, sem=3, frame=1, ao_eps=0.01)
        orc.set_simd(True)
        got, gst = osc.trace_primary(ov, 160, 90, sem=3)
        got_ao, _ = osc.trace_ao(ov, 160, 90, got, sem=3, frame=1, ao_eps=0.01)
        assert_hits_equal(got, want, "frame (simd)")
        assert_hits_equal(got_ao, want_ao, "AO frame (simd)")
        assert (gst.n_node, gst.n_tri, gst.n_hits) == (wst.n_node, wst.n_tri, wst.n_hits)
    finally:
        orc.set_simd(False)
