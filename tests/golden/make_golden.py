"""Generates the committed golden fixtures in this directory.

PARITY UNPINNED: the reference holds no golden vectors for this path and cannot
be executed, so these fixtures are produced by this repo's own
CPU oracle: `bf_*` arrays come from the BVH-independent brute-force query
(every ray against every triangle), `orc_*` arrays from the CWBVH restatement.
Each .npz carries its INPUTS (80-byte nodes, triangles, instance table, view)
and the expected outputs, so the tests do not depend on the builder or on
/root/reference being present.

The `ref_*` fixtures are built from the two small OBJ assets the reference ships
(assets/obj/cornell_box.obj, assets/obj/box.obj) with the cameras of their .ron scene
files.  The reference's licences do not cover its assets (README.md:99), so these
fixtures hold NO geometry: only the asset's name, SHA-256 digests of the rebuilt
buffers, the view and the expected hit buffers.  Tests rebuild the inputs from the
mounted reference checkout and skip where it is absent.

Run from the repo root:  python tests/golden/make_golden.py [name-substring ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import tray_racing_amd as T  # noqa: E402
from oracle import binding as O  # noqa: E402

SEMS = (0, 3)  # TRX_SEM_HLSL, TRX_SEM_CPU


def view_bytes(view):
    import ctypes
    return np.frombuffer(ctypes.string_at(ctypes.byref(view), ctypes.sizeof(view)), dtype=np.uint8).copy()


def tie_scene():
    """Axis-aligned unit-quad grids: rays through grid vertices / along edges hit
    up to six triangles at exactly the same t (and have zero direction components)."""
    tris = []
    for z in (0.0, -1.5):
        for i in range(-4, 4):
            for j in range(-4, 4):
                a, b, c, d = (i, j, z), (i + 1, j, z), (i + 1, j + 1, z), (i, j + 1, z)
                tris.append(a + b + c)
                tris.append(a + c + d)
    verts = np.array(tris, dtype=np.float32)
    rays = []
    for x in np.arange(-4, 4.5, 0.5):
        for y in np.arange(-4, 4.5, 0.5):
            rays.append(((x, y, 3.0), (0.0, 0.0, -1.0)))       # straight down: two zero components
            rays.append(((x, y, 3.0), (0.0, 0.6, -0.8)))       # one zero component, crosses edges
    rays.append(((0.25, 0.25, -0.75), (0.0, 0.0, 1.0)))        # from between the sheets, upwards
    r = np.zeros(len(rays), dtype=T.RAY_DTYPE)
    r["origin"] = np.array([o for o, _ in rays], dtype=np.float32)
    r["direction"] = np.array([d for _, d in rays], dtype=np.float32)
    r["tmin"] = 0.0
    r["tmax"] = O.F32_MAX
    return verts, r


ONLY = sys.argv[1:]
REF_ASSETS = "/root/reference/assets"


def ron_camera(path):
    """eye / look_at / fov of a tray_racing scene file (assets/scenes/*.ron)."""
    import re
    txt = open(path).read()
    vec = lambda key: [float(x) for x in re.search(key + r":\s*\(([^)]*)\)", txt).group(1).split(",")]
    return vec("eye"), vec("look_at"), float(re.search(r"fov:\s*([-0-9.]+)", txt).group(1))


def save(name, flat, view, w, h, extra, asset=None, use_tlas=False):
    if ONLY and not any(o in name for o in ONLY):
        return
    osc = O.Scene.from_flat(flat)
    if asset is None:
        out = dict(nodes=flat.nodes, tri_verts=flat.tri_verts, instance_offsets=flat.instance_offsets,
                   tlas_start=np.uint32(flat.tlas_start), width=np.uint32(w), height=np.uint32(h))
    else:
        # built from one of the reference's OBJ assets: the licences do not cover assets (README.md:99), so the
        # geometry stays out of the repository; the fixture names the asset and pins the rebuilt buffers by digest
        import hashlib
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        out = dict(asset=asset, use_tlas=np.uint32(use_tlas), n_tris=np.uint64(flat.n_tris), n_nodes=np.uint64(flat.n_nodes),
                   tri_sha256=sha(flat.tri_verts), nodes_sha256=sha(flat.nodes), width=np.uint32(w), height=np.uint32(h))
    if view is not None:
        ov = O.view_from_bytes(view)
        out["view"] = view_bytes(view)
        for sem in SEMS:
            prim, st = osc.trace_primary(ov, w, h, sem=sem)
            out["orc_primary_sem%d" % sem] = prim
            out["orc_counts_sem%d" % sem] = np.array([st.n_node, st.n_tri, st.n_hits, st.max_stack], dtype=np.uint64)
            out["bf_primary_sem%d" % sem] = osc.brute_primary(ov, w, h, sem=sem)
            ao, _ = osc.trace_ao(ov, w, h, prim, sem=sem, frame=2, ao_eps=0.01)
            out["orc_ao_sem%d" % sem] = ao
    out.update(extra(osc) if extra else {})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}, os.path.getsize(path), "bytes")


def main():
    # TLAS fixtures are in the REFERENCE's layout - one TLAS primitive per whole BLAS (src/cwbvh.rs:108-137), no entry-node
    # table, which buffers from the reference's host never carry; this library's re-braided TLAS has its own tests
    # (tests/test_rebraid.py and the full-size two-level frame of the GPU suite)
    from tray_racing_amd import _lib
    _lib.check(_lib.load().trx_set_build_rebraid(0.0))
    # 1. Cornell-class box (5 objects), camera of assets/scenes/cornell_box.ron
    verts, counts = T.gen_scene("cornell", 0, 1)
    eye, look, fov = T.scene_camera("cornell")
    flat = T.flat_build(verts, counts, use_tlas=False)
    save("cornell_64", flat, T.view_from_camera(eye, look, fov, 64, 64), 64, 64, None)
    # 2. same geometry, one BLAS per object + TLAS
    flat = T.flat_build(verts, counts, use_tlas=True)
    save("cornell_tlas_48", flat, T.view_from_camera(eye, look, fov, 48, 48), 48, 48, None)
    # 3. random triangle soup, image size not a multiple of 8
    verts, counts = T.gen_scene("soup", 1500, 7)
    eye, look, fov = T.scene_camera("soup")
    flat = T.flat_build(verts, counts)
    save("soup_52x44", flat, T.view_from_camera(eye, look, fov, 52, 44), 52, 44, None)
    # 4. exact ties and zero direction components, explicit rays
    verts, rays = tie_scene()
    flat = T.flat_build(verts)

    def tie_extra(osc):
        e = {"rays": rays}
        for sem in SEMS:
            e["orc_rays_sem%d" % sem] = osc.trace_rays(rays, sem=sem)[0]
            e["bf_rays_sem%d" % sem] = osc.brute_rays(rays, sem=sem)
        return e

    save("ties_rays", flat, None, 0, 0, tie_extra)
    # 5. the reference's 24-byte f16 triangle format + explicit rays through a TLAS, GPU AO epsilon
    verts, counts = T.gen_scene("kitchen", 4000, 3)
    eye, look, fov = T.scene_camera("kitchen")
    flat = T.flat_build(verts, counts, use_tlas=True)
    view = T.view_from_camera(eye, look, fov, 56, 40)
    packed = T.pack_tris_f16(flat.tri_verts)
    rng = np.random.default_rng(42)
    rr = np.zeros(700, dtype=T.RAY_DTYPE)
    pts = flat.tri_verts.reshape(-1, 3)
    rr["origin"] = rng.uniform(pts.min(0), pts.max(0), size=(700, 3)).astype(np.float32)
    d = rng.normal(size=(700, 3)).astype(np.float32)
    d[:20, 1] = 0.0
    rr["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rr["tmin"] = np.where(np.arange(700) % 5 == 0, 0.3, 0.0).astype(np.float32)
    rr["tmax"] = np.where(np.arange(700) % 3 == 0, 2.5, O.F32_MAX).astype(np.float32)

    def f16_extra(osc):
        osc16 = O.Scene(flat.nodes, None, flat.instance_offsets, flat.tlas_start, tri_f16=packed)
        ov = O.view_from_bytes(view)
        e = {"tri_f16": packed, "rays": rr}
        for sem in SEMS:
            prim, _ = osc16.trace_primary(ov, 56, 40, sem=sem)
            e["orc_f16_primary_sem%d" % sem] = prim
            e["orc_f16_ao_eps1e-4_sem%d" % sem] = osc16.trace_ao(ov, 56, 40, prim, sem=sem, frame=9, ao_eps=0.0001)[0]
            e["orc_rays_sem%d" % sem] = osc.trace_rays(rr, sem=sem)[0]
            e["bf_rays_sem%d" % sem] = osc.brute_rays(rr, sem=sem)
        return e

    save("kitchen_tlas_f16_56x40", flat, view, 56, 40, f16_extra)
    # 5b. stand-ins for the two ref_* fixtures below that need no reference checkout (so they also run on the GPU box):
    #     a 14-triangle, two-object scene - one box and a ground quad, the triangle count and object split of
    #     assets/obj/box.obj - through a TLAS with the camera of assets/scenes/box.ron; the Cornell-class box with the
    #     camera of assets/scenes/cornell_box.ron is fixture 1 above
    def quad(a, b, c, d):
        return [a + b + c, a + c + d]
    bx = [(-1.0, 0.0, -1.0), (1.0, 0.0, -1.0), (1.0, 0.0, 1.0), (-1.0, 0.0, 1.0),
          (-1.0, 2.0, -1.0), (1.0, 2.0, -1.0), (1.0, 2.0, 1.0), (-1.0, 2.0, 1.0)]
    box_tris = (quad(bx[0], bx[1], bx[2], bx[3]) + quad(bx[4], bx[7], bx[6], bx[5]) + quad(bx[0], bx[4], bx[5], bx[1]) +
                quad(bx[1], bx[5], bx[6], bx[2]) + quad(bx[2], bx[6], bx[7], bx[3]) + quad(bx[3], bx[7], bx[4], bx[0]))
    ground = quad((-6.0, -0.001, -6.0), (6.0, -0.001, -6.0), (6.0, -0.001, 6.0), (-6.0, -0.001, 6.0))
    verts = np.array(box_tris + ground, dtype=np.float32)
    counts = np.array([12, 2], dtype=np.uint64)
    flat = T.flat_build(verts, counts, use_tlas=True)
    save("box14_tlas_48", flat, T.view_from_camera((3.0, 1.5, 1.4), (-3.9438584, 1.5, -1.7303504), 90.0, 48, 48), 48, 48, None)
    # 6./7. the reference's own small assets (the only geometry it ships), its loader rules and its cameras
    if os.path.exists(os.path.join(REF_ASSETS, "obj", "cornell_box.obj")):
        verts, counts = T.load_meshs(os.path.join(REF_ASSETS, "obj", "cornell_box.obj"))
        eye, look, fov = ron_camera(os.path.join(REF_ASSETS, "scenes", "cornell_box.ron"))
        flat = T.flat_build(verts, counts)
        save("ref_cornell_box_64", flat, T.view_from_camera(eye, look, fov, 64, 64), 64, 64, None, asset="cornell_box.obj")
        verts, counts = T.load_meshs(os.path.join(REF_ASSETS, "obj", "box.obj"))
        eye, look, fov = ron_camera(os.path.join(REF_ASSETS, "scenes", "box.ron"))
        flat = T.flat_build(verts, counts, use_tlas=True)
        save("ref_box_tlas_48", flat, T.view_from_camera(eye, look, fov, 48, 48), 48, 48, None, asset="box.obj", use_tlas=True)
    else:
        print("reference assets not mounted: ref_* fixtures left as they are")


if __name__ == "__main__":
    main()
