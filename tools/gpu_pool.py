"""The drain's orphan pool (experiment builds, -DTRX_POOL=1; TRX_LIB=tuning_libs/pool.so): per scene one AO pass and one
explicit-ray batch at 1920x1080 - checked against the oracle bit for bit, the launch's error flag read - then timed in
batches of back-to-back launches, with the pool's counters per launch (rays parked / taken / turned away / refused by the
record check / reservations / slots reserved).
usage: TRX_LIB=tuning_libs/pool.so python tools/gpu_pool.py [scene ...]"""
import ctypes as C
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import tray_racing_amd as T
    from oracle import binding as O
    from tray_racing_amd import _lib as L
    from tray_racing_amd import dist as D
    from tools.prof_config import hemisphere_rays
    lib = L.load()
    w, h = 1920, 1080
    bad = 0
    for name in (sys.argv[1:] or ["hairball", "bistro"]):
        verts, counts = T.gen_scene(name, 0, 1)
        flat = T.flat_build(verts, counts, use_tlas=False)
        eye, look, fov = T.scene_camera(name)
        view = T.view_from_camera(eye, look, fov, w, h)
        sc = T.Scene(flat)
        osc = O.Scene.from_flat(flat)
        ov = O.view_from_bytes(view)
        prim = torch.zeros(w * h, dtype=torch.int64, device="cuda")
        ao = torch.zeros(w * h, dtype=torch.int64, device="cuda")
        sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=3)
        torch.cuda.synchronize()
        op = D.int64_to_hits(prim)
        stats = (C.c_uint32 * 8)()

        def pool_stats():
            if not hasattr(lib, "trx_debug_pool_stats"):
                return [0] * 8
            lib.trx_debug_pool_stats(sc._h, stats)
            return list(stats)
        pool_stats()
        # ---- correctness: three AO passes (different seeds) and a ray batch against the oracle
        for frame in (0, 1, 2):
            sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=3, frame=frame, ao_eps=0.01)
            torch.cuda.synchronize()
            try:
                sc.check()
                err = ""
            except Exception as e:  # noqa: BLE001
                err = str(e)
            want, _ = osc.trace_ao(ov, w, h, op, sem=3, frame=frame, ao_eps=0.01)
            got = D.int64_to_hits(ao)
            same = bool((got["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and (got["prim"] == want["prim"]).all())
            st = pool_stats()
            print("%-9s AO frame %d: equals oracle %s%s | parked %d taken %d turned away %d refused %d | %d reservations, %d slots | never whole %d, check code %d" % (
                name, frame, same, (" ERROR " + err) if err else "", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]), flush=True)
            bad += (not same) or bool(err)
        rays = hemisphere_rays(flat, None, eye, 1 << 20, 7)
        d_rays = torch.from_numpy(rays.view("u1").copy()).cuda()
        hits = torch.zeros(len(rays), dtype=torch.int64, device="cuda")
        sc.trace_rays_dev(d_rays.data_ptr(), len(rays), hits.data_ptr(), sem=3)
        torch.cuda.synchronize()
        want, _ = osc.trace_rays(rays, sem=3)
        got = D.int64_to_hits(hits)
        same = bool((got["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and (got["prim"] == want["prim"]).all())
        st = pool_stats()
        print("%-9s 1 M rays: equals oracle %s | parked %d taken %d turned away %d refused %d | %d reservations, %d slots | never whole %d, check code %d" % (
            name, same, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]), flush=True)
        bad += not same
        if bad:
            print("stopping: results differ", flush=True)
            return 1

        def batches(fn, n_batches=6, per=10, warm=20):
            for i in range(warm):
                fn(i)
            ts = []
            for b in range(n_batches):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(per):
                    fn(b * per + i)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / per)
            return min(ts), statistics.median(ts)
        a = batches(lambda i: sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=3, frame=i, ao_eps=0.01))
        st = pool_stats()
        r = batches(lambda i: sc.trace_rays_dev(d_rays.data_ptr(), len(rays), hits.data_ptr(), sem=3))
        print("%-9s AO pass %.4f ms (min %.4f), 1 M rays %.4f ms (min %.4f) | per AO launch: parked %.0f taken %.0f turned away %.0f" % (
            name, a[1], a[0], r[1], r[0], st[0] / 80.0, st[1] / 80.0, st[2] / 80.0), flush=True)
        sc.check()
        sc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
