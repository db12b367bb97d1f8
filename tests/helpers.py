"""Shared helpers for the test-suite (scene construction, comparisons)."""
import numpy as np

F32_MAX = 3.4028234663852886e38
ALL_SEMS = (0, 1, 2, 3, 4, 5, 6, 7)


def make_scene(T, O, name, n, w, h, tlas=False, seed=1):
    verts, counts = T.gen_scene(name, n, seed)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    osc = O.Scene.from_flat(flat)
    return flat, view, osc, O.view_from_bytes(view)


REF_ASSETS = "/root/reference/assets"


def golden_inputs(T, g):
    """(nodes, tri_verts, instance_offsets, tlas_start) of a golden fixture.  Fixtures made from the reference's own
    OBJ assets carry no geometry ("MIT & Apache 2.0 Licenses don't apply to assets", README.md:99): only the asset's
    name, the expected outputs and SHA-256 digests of the buffers; their inputs are rebuilt from the mounted
    reference checkout (deterministic loader + builder) and checked against the digests, or the test is skipped."""
    import hashlib
    import os

    import pytest
    if "nodes" in g.files:
        return g["nodes"], g["tri_verts"], g["instance_offsets"], int(g["tlas_start"])
    path = os.path.join(REF_ASSETS, "obj", str(g["asset"]))
    if not os.path.exists(path):
        pytest.skip("fixture built from %s: the reference checkout is not mounted here" % g["asset"])
    verts, counts = T.load_meshs(path)
    flat = T.flat_build(verts, counts, use_tlas=bool(g["use_tlas"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(flat.tri_verts) == str(g["tri_sha256"]), "loader / builder no longer reproduce the fixture's triangles"
    assert sha(flat.nodes) == str(g["nodes_sha256"]), "builder output changed: regenerate tests/golden (make_golden.py ref_)"
    return flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_hits_equal(got, want, what=""):
    """Bit-exact on t (covers +inf) and exact on prim."""
    bt = np.flatnonzero(bits(got["t"]) != bits(want["t"]))
    bp = np.flatnonzero(got["prim"] != want["prim"])
    assert bt.size == 0 and bp.size == 0, "%s: %d t mismatches (first %s), %d prim mismatches (first %s)" % (
        what, bt.size, bt[:5], bp.size, bp[:5])


def random_rays(T, flat, n, seed, zero_dirs=True):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=T.RAY_DTYPE)
    pts = flat.tri_verts.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    pad = 0.1 * (hi - lo) + 1e-3
    rays["origin"] = rng.uniform(lo - pad, hi + pad, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    if zero_dirs and n >= 64:
        d[0:8, 0] = 0.0
        d[8:16, 1] = 0.0
        d[16:24, 2] = 0.0
        d[24:28, 0:2] = 0.0
        d[28:32] = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [-1, 0, 0]], dtype=np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"] = d
    rays["tmin"] = 0.0
    rays["tmax"] = F32_MAX
    rays["tmax"][::7] = np.float32(0.5 * float(np.linalg.norm(hi - lo)))
    rays["tmin"][::11] = np.float32(0.05 * float(np.linalg.norm(hi - lo)))
    return rays


# ---- hand-made CWBVH nodes (an independent restatement of the encoder,
# embree/src/bvh_embree_to_cwbvh.rs:85-186, used to craft adversarial trees) -------------

def encode_node(box_min, box_max, children, child_base, prim_base):
    """children: list of 8 entries, each None | ("inner", lo, hi) | ("leaf", lo, hi, n_tris)."""
    import math
    node = np.zeros(20, dtype=np.uint32)
    b = node.view(np.uint8)
    p = np.asarray(box_min, dtype=np.float32)
    node[:3] = p.view(np.uint32)
    e = np.zeros(3)
    for k in range(3):
        ext = max(float(box_max[k]) - float(box_min[k]), 1e-20) / 255.0
        ex = math.ceil(math.log2(ext))
        while math.ceil((float(box_max[k]) - float(p[k])) / 2.0 ** ex) > 255:
            ex += 1
        e[k] = 2.0 ** ex
        b[12 + k] = ex + 127
    imask, tri_total = 0, 0
    for s, c in enumerate(children):
        if c is None:
            continue
        lo, hi = np.asarray(c[1], float), np.asarray(c[2], float)
        for k in range(3):
            b[32 + 16 * k + s] = int(min(max(math.floor((lo[k] - float(p[k])) / e[k]), 0), 255))
            b[32 + 16 * k + 8 + s] = int(min(max(math.ceil((hi[k] - float(p[k])) / e[k]), 0), 255))
        if c[0] == "inner":
            imask |= 1 << s
            b[24 + s] = 0x20 | (24 + s)
        else:
            b[24 + s] = tri_total | {1: 0x20, 2: 0x60, 3: 0xE0}[c[3]]
            tri_total += c[3]
    b[15] = imask
    node[4] = child_base
    node[5] = prim_base
    return node


def deep_chain_scene(depth):
    """A CWBVH whose traversal stack reaches `depth` entries for the ray (0.3,0.3,1)->(0,0,-1):
    node k has two inner children, the rest of the chain (visited first) and a terminal node
    holding triangle k at z = -k (left pending on the stack).  The closest hit is triangle 0 at t = 1."""
    tris, nodes = [], []
    lo_xy, hi_xy = (0.0, 0.0), (1.0, 1.0)
    # the fixed-up direction (eps, eps, -1) has oct_inv = 6: slot s goes to bit 24 + (s ^ 6), the
    # highest bit is visited first, so slot 1 (bit 31) holds the chain and slot 0 (bit 30) the terminal
    for k in range(depth):
        tris.append([0, 0, -k, 1, 0, -k, 0, 1, -k])
    # layout: node 2k = chain node k, node 2k+1 = terminal node k; last chain node is a plain leaf node
    for k in range(depth):
        zt = -float(k)
        term_box = ((*lo_xy, zt), (*hi_xy, zt))
        if k < depth - 1:
            chain_box = ((*lo_xy, -float(depth - 1)), (*hi_xy, zt - 1.0))
            kids = [None] * 8
            kids[1] = ("inner", *chain_box)
            kids[0] = ("inner", *term_box)
            # children stored in slot order: slot 0 (terminal) first, then slot 1 (chain)
            nodes.append(("chain", k, (*lo_xy, -float(depth - 1)), (*hi_xy, zt), kids))
        else:
            nodes.append(("last", k, (*lo_xy, zt), (*hi_xy, zt), None))
    out = []
    # index plan: chain node k at index 2k, its terminal at 2k+1, next chain node at 2k+2;
    # child_base of chain node k = 2k+1 (slot 0 -> 2k+1, slot 1 -> 2k+2)
    for kind, k, bmin, bmax, kids in nodes:
        if kind == "chain":
            out.append(encode_node(bmin, bmax, kids, 2 * k + 1, 0))
            leaf = [None] * 8
            leaf[0] = ("leaf", (*lo_xy, -float(k)), (*hi_xy, -float(k)), 1)
            out.append(encode_node((*lo_xy, -float(k)), (*hi_xy, -float(k)), leaf, 0, k))
        else:
            leaf = [None] * 8
            leaf[0] = ("leaf", bmin, bmax, 1)
            out.append(encode_node(bmin, bmax, leaf, 0, k))
    return np.array(out, dtype=np.uint32), np.array(tris, dtype=np.float32)


# ---- instanced scenes (TLAS primitives with transforms) ----------------------------------------------------

def random_affine(rng, scale_lo=0.4, scale_hi=1.6, spread=4.0):
    """A column-major 4x4 object-to-world matrix: rotation x non-uniform scale, then a translation."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    A = R @ np.diag(rng.uniform(scale_lo, scale_hi, size=3))
    M = np.eye(4)
    M[:3, :3] = A
    M[:3, 3] = rng.uniform(-spread, spread, size=3)
    return M.T.reshape(16).astype(np.float32)   # column-major: m[c*4 + r]


def w2o_rows(o2w16):
    """World-to-object rows {m0 m1 m2 t} x 3 (12 floats) of a column-major affine 4x4, inverse taken in float64."""
    M = np.asarray(o2w16, dtype=np.float64).reshape(4, 4).T
    return np.linalg.inv(M)[:3, :].reshape(12).astype(np.float32)


def instanced_scene(T, seed=3, n_objects=3, n_instances=10, tris_per_object=400, kind="soup", spread=4.0):
    """A few small triangle-soup objects and `n_instances` transformed instances of them (several per object).
    Returns (flat, object_to_world in TLAS-primitive order, world-space triangles [n, 9] instance-major in
    TLAS-primitive order, first world triangle of every TLAS primitive)."""
    rng = np.random.default_rng(seed)
    if kind == "soup":
        verts, counts = [], []
        for o in range(n_objects):
            v, _ = T.gen_scene("soup", tris_per_object + 37 * o, seed + o)
            verts.append(v)
            counts.append(v.shape[0])
        verts = np.concatenate(verts)
    else:   # the objects of a procedural scene (walls, boxes ...: large triangles, dense images)
        verts, counts = T.gen_scene(kind, tris_per_object, seed)
        counts = [int(c) for c in counts if c]
        n_objects = len(counts)
    inst_obj = np.array([k % n_objects for k in range(n_instances)], dtype=np.uint32)
    o2w = np.stack([random_affine(rng, spread=spread) for _ in range(n_instances)])
    o2w[0] = np.eye(4, dtype=np.float32).reshape(16)        # one identity instance among them
    flat = T.flat_build_instanced(verts, counts, inst_obj, o2w)
    # world-space geometry, in float64 then f32, for the BVH-independent brute-force query
    world, first = [], [0]
    bts = flat.blas_tri_start
    blas_of_offset = {int(off): b for b, off in enumerate(sorted(set(int(x) for x in flat.instance_offsets)))}
    for k in range(flat.instance_offsets.size):
        b = blas_of_offset[int(flat.instance_offsets[k])]
        tv = flat.tri_verts[bts[b]:bts[b + 1]].astype(np.float64).reshape(-1, 3)
        M = flat.instance_transforms[k].astype(np.float64).reshape(4, 4).T
        world.append((tv @ M[:3, :3].T + M[:3, 3]).reshape(-1, 9).astype(np.float32))
        first.append(first[-1] + world[-1].shape[0])
    return flat, flat.instance_transforms, np.concatenate(world), np.array(first), blas_of_offset


def aimed_rays(T, tri_verts, n, seed):
    """Rays from around the geometry towards random points of random triangles (so most of them hit something)."""
    rng = np.random.default_rng(seed)
    v = np.asarray(tri_verts, dtype=np.float32).reshape(-1, 3, 3)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    rays = np.zeros(n, dtype=T.RAY_DTYPE)
    o = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(n, 3))
    tri = v[rng.integers(0, v.shape[0], size=n)]
    bc = rng.dirichlet([1, 1, 1], size=n)
    target = (tri * bc[:, :, None]).sum(1)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["origin"] = o.astype(np.float32)
    rays["direction"] = d.astype(np.float32)
    rays["tmin"] = 0.0
    rays["tmax"] = F32_MAX
    return rays
