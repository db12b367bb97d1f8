"""trx_traverse1 under concurrent callers (development aid): rays per second and rays per launch of the single-ray
Traversable::traverse at several host thread counts, through tests/c_abi/traverse_threads.c (pthreads, one ray per call),
and - with TRX_LIB pointing at an older build - the same thread loop through ctypes (the round-1..4 path: one copy +
launch + copy + synchronise per ray and thread).
usage: python tools/gpu_traverse1.py [--tris 500000] [--rays 200000] [--threads 1,16,64,256,1024]"""
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    args = dict(zip(sys.argv[1::2], sys.argv[2::2]))
    tris = int(args.get("--tris", 500000))
    n_rays = int(args.get("--rays", 200000))
    counts = [int(x) for x in args.get("--threads", "1,16,64,256,1024").split(",")]
    import tray_racing_amd as T
    from oracle import binding as O
    from helpers import make_scene
    w, h = 512, 1080
    flat, _view, osc, ov = make_scene(T, O, "demoscene", tris, w, h)
    rays = osc.primary_rays(ov, w, h)
    rng = np.random.default_rng(5)
    rays = np.ascontiguousarray(rays[rng.permutation(w * h)[:n_rays]])
    if os.environ.get("TRX_LIB"):
        # an older library: its trx_traverse1 from Python threads (ctypes releases the GIL around the call)
        import ctypes as C
        from tray_racing_amd import _lib as L
        sc = T.Scene(flat)
        lib = L.load()
        n = min(n_rays, 20000)
        for threads in counts:
            if threads > 64:
                continue
            outs = (L.RayHit * n)()
            rr = (L.Ray * n).from_buffer_copy(rays[:n].tobytes())

            def work(k):
                for i in range(k, n, threads):
                    lib.trx_traverse1(sc._h, C.byref(rr[i]), 3, C.byref(outs[i]))
            ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            print("%-28s threads %4d: %8.4f Mrays/s (%d rays, %.2f s)" % (os.path.basename(os.environ["TRX_LIB"]), threads, n / dt / 1e6, n, dt), flush=True)
        sc.close()
        return 0
    tmp = tempfile.mkdtemp(prefix="trx_t1_", dir="/tmp")
    exe = os.path.join(tmp, "traverse_threads")
    libdir = os.path.join(ROOT, "tray_racing_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "traverse_threads.c"),
                           "-o", exe, "-pthread", "-L", libdir, "-ltrx", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    rp, hp = os.path.join(tmp, "rays.bin"), os.path.join(tmp, "hits.bin")
    rays.tofile(rp)
    want, _ = osc.trace_rays(rays, sem=3)
    for threads in counts:
        out = subprocess.run([exe, "demoscene", str(tris), str(threads), "3", rp, hp], capture_output=True, text=True, timeout=900)
        if out.returncode:
            print("threads %d failed: %s" % (threads, out.stderr[-300:]))
            return 1
        n, secs, launches = out.stdout.split()
        got = np.fromfile(hp, dtype=np.dtype([("primitive_id", "<u4"), ("geometry_id", "<u4"), ("instance_id", "<u4"), ("t", "<f4")]))
        ok = bool((got["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and (got["primitive_id"] == want["prim"]).all())
        print("combiner  threads %4d: %8.4f Mrays/s, %7.1f rays per launch, %6.1f us per launch, equals oracle: %s" % (
            threads, int(n) / float(secs) / 1e6, int(n) / max(int(launches), 1), float(secs) / max(int(launches), 1) * 1e6, ok), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
