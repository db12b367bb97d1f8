"""Ad-hoc GPU check used during development: HIP path vs oracle on a few scenes,
then a kernel-variant timing sweep on the bistro-class scene.  Not a test and
not the bench: tests/ and bench.py are the judged artefacts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from oracle import binding as O  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402


def compare(tag, g, o):
    bt = (g["t"].view(np.uint32) != o["t"].view(np.uint32)).sum()
    bp = (g["prim"] != o["prim"]).sum()
    print("  %-34s n=%d  t-bit-mismatch=%d  prim-mismatch=%d" % (tag, g.shape[0], bt, bp), flush=True)
    return bt == 0 and bp == 0


def parity(name, n, w, h, tlas):
    verts, counts = T.gen_scene(name, n, 1)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    osc = O.Scene.from_flat(flat)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    ov = O.view_from_bytes(view)
    sc = T.Scene(flat)
    ok = True
    print("%s tris=%d nodes=%d tlas=%s %dx%d" % (name, flat.n_tris, flat.n_nodes, tlas, w, h), flush=True)
    for sem in (0, 1, 2, 3, 4, 7):
        gp, gao, _ = sc.trace_primary_ao(view, w, h, sem=sem, frame=3, ao_eps=0.01)
        op, _ = osc.trace_primary(ov, w, h, sem=sem)
        oao, _ = osc.trace_ao(ov, w, h, op, sem=sem, frame=3, ao_eps=0.01)
        ok &= compare("sem=%d primary" % sem, gp, op)
        ok &= compare("sem=%d ao" % sem, gao, oao)
    rng = np.random.default_rng(5)
    rays = np.zeros(20000, dtype=T.RAY_DTYPE)
    lo, hi = flat.tri_verts.reshape(-1, 3).min(0), flat.tri_verts.reshape(-1, 3).max(0)
    rays["origin"] = rng.uniform(lo, hi, size=(rays.shape[0], 3)).astype(np.float32)
    d = rng.normal(size=(rays.shape[0], 3)).astype(np.float32)
    d[:100, 0] = 0.0  # zero direction components
    d[100:200, 1] = 0.0
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmin"] = 0.0
    rays["tmax"] = 3.4028234663852886e38
    rays["tmax"][::7] = 0.5
    for sem in (0, 3):
        g, _ = sc.trace_rays(rays, sem=sem)
        o, _ = osc.trace_rays(rays, sem=sem)
        ok &= compare("sem=%d random rays" % sem, g, o)
    st = sc.count_primary(view, w, h, sem=3)
    _, ost = osc.trace_primary(ov, w, h, sem=3)
    print("  counts gpu node=%d tri=%d hits=%d maxsp=%d | oracle node=%d tri=%d hits=%d maxsp=%d" % (
        st.n_node, st.n_tri, st.n_hits, st.max_stack, ost.n_node, ost.n_tri, ost.n_hits, ost.max_stack), flush=True)
    ok &= (st.n_node == ost.n_node and st.n_tri == ost.n_tri and st.n_hits == ost.n_hits)
    sc.close()
    return ok


def sweep(name, n, w, h, variants, sems=(3,), frames=10):
    verts, counts = T.gen_scene(name, n, 1)
    t0 = time.time()
    flat = T.flat_build(verts, counts, use_tlas=False)
    print("%s tris=%d nodes=%d build %.1fs" % (name, flat.n_tris, flat.n_nodes, time.time() - t0), flush=True)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    lib = L.load()
    st = sc.count_primary(view, w, h, sem=3)
    rays = w * h
    bytes_ray = (80 * st.n_node + 48 * st.n_tri + 8 * st.n_rays) / st.n_rays
    print("  node/ray %.2f tri/ray %.2f hits %d maxsp %d  algorithmic B/ray %.0f" % (
        st.n_node / st.n_rays, st.n_tri / st.n_rays, st.n_hits, st.max_stack, bytes_ray), flush=True)
    for sem in sems:
        for v in variants:
            lib.trx_set_kernel_variant(v)
            mn, mean = sc.bench_primary(view, w, h, sem=sem, warmup=2, frames=frames)
            print("  sem=%d variant=%3d  min %.3f ms mean %.3f ms  -> %.1f Mrays/s  (%.2f TB/s algorithmic)" % (
                sem, v, mn, mean, rays / mn / 1e3, bytes_ray * rays / mn / 1e9), flush=True)
    lib.trx_set_kernel_variant(0)
    sc.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    lib = L.load()
    import ctypes as C
    buf = C.create_string_buffer(128)
    print("devices", lib.trx_device_count(), flush=True)
    lib.trx_device_name(0, buf, 128)
    print("device0", buf.value.decode(), flush=True)
    ok = True
    if what in ("all", "parity"):
        ok &= parity("cornell", 0, 128, 96, False)
        ok &= parity("soup", 3000, 64, 64, False)
        ok &= parity("kitchen", 20000, 160, 96, False)
        ok &= parity("kitchen", 20000, 100, 60, True)
        ok &= parity("bistro", 200000, 240, 136, False)
        ok &= parity("san_miguel", 150000, 160, 90, True)
        print("PARITY", "OK" if ok else "FAILED", flush=True)
    if what in ("all", "sweep"):
        sweep("kitchen", 0, 1920, 1080, [64, 32, 16, 8], sems=(3,))
        sweep("bistro", 0, 1920, 1080, [64, 48, 32, 24, 16, 8, 4], sems=(3, 0, 7))
    sys.exit(0 if ok else 1)
