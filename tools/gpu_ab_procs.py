"""A/B of library builds over many PROCESSES (development aid).  One process's AO pass time depends on where its scene landed
in memory and on the clock state it found - +-3 % from process to process, more than most changes are worth - so each
library named on the command line is measured in ROUNDS fresh child processes, alternating, and the table gives the
median and the spread over processes.  A child times batches of back-to-back launches (one hipEvent pair per batch: no
idle gaps, clocks stay up) of the AO pass and of an explicit-ray pass, per scene.
usage: python tools/gpu_ab_procs.py [--rounds 5] [--scenes hairball,bistro] [--variant 0] lib [lib ...]   (names under tuning_libs/, or `product`; lib:0x... = under that kernel-variant word)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(scenes, variant):
    import numpy as np  # noqa: F401
    import torch
    sys.path.insert(0, ROOT)
    import tray_racing_amd as T
    from tray_racing_amd import _lib as L
    from tools.prof_config import hemisphere_rays
    lib = L.load()
    w, h = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))   # (W=3840 H=2160 TLAS=1: configs[4])
    SEM = int(os.environ.get("SEM", "3"))   # (SEM=0: the shader's literal arithmetic, TRX_SEM_HLSL)
    out = {}
    for name in scenes:
        verts, counts = T.gen_scene(name, 0, 1)
        flat = T.flat_build(verts, counts, use_tlas=os.environ.get("TLAS", "0") == "1")   # TLAS=1: the two-level kernels
        eye, look, fov = T.scene_camera(name)
        view = T.view_from_camera(eye, look, fov, w, h)
        sc = T.Scene(flat)
        prim = torch.zeros(w * h, dtype=torch.int64, device="cuda")
        ao = torch.zeros(w * h, dtype=torch.int64, device="cuda")
        sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=SEM)
        torch.cuda.synchronize()
        rays = hemisphere_rays(flat, None, eye, 1 << 20, 7)
        d_rays = torch.from_numpy(rays.view("u1").copy()).cuda()
        hits = torch.zeros(len(rays), dtype=torch.int64, device="cuda")
        lib.trx_set_kernel_variant(variant)

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

        a_min, a_med = batches(lambda i: sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=SEM, frame=i, ao_eps=0.01))
        r_min, r_med = batches(lambda i: sc.trace_rays_dev(d_rays.data_ptr(), len(rays), hits.data_ptr(), sem=SEM))
        p_min, p_med = batches(lambda i: sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=SEM), warm=140)
        out[name] = {"ao": a_med, "ao_min": a_min, "rays": r_med, "rays_min": r_min, "primary": p_med, "primary_min": p_min}
        if hasattr(lib, "trx_trace_frame_dev"):   # the reference-style frame: two launches / one launch; 4 spp in one launch
            prim1 = torch.zeros(w * h, dtype=torch.int64, device="cuda")
            ao4 = torch.zeros(4 * w * h, dtype=torch.int64, device="cuda")

            def two(i):
                sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=SEM)
                sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=SEM, frame=i, ao_eps=0.01)
            out[name]["frame2"] = batches(two)[1]
            out[name]["frame1"] = batches(lambda i: sc.trace_frame_dev(view, w, h, prim1.data_ptr(), ao.data_ptr(), sem=SEM, frame=i, ao_eps=0.01))[1]
            out[name]["ao4"] = batches(lambda i: sc.trace_ao_batch_dev(view, w, h, prim.data_ptr(), ao4.data_ptr(), w * h, 4, sem=SEM, frame0=4 * i, ao_eps=0.01), per=4)[1]
        lib.trx_set_kernel_variant(0)
        sc.close()
    print("AB_CHILD " + json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child(args[1].split(","), int(args[2], 0))
    rounds, scenes, variant = 5, "hairball,bistro", "0"
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1])
        elif args[0] == "--scenes":
            scenes = args[1]
        elif args[0] == "--variant":
            variant = args[1]
        args = args[2:]
    libs = args
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ)
            name, _, var = l.partition(":")   # lib[:variant word]
            if name != "product":
                env["TRX_LIB"] = os.path.join(ROOT, "tuning_libs", name + ".so")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scenes, var or variant], env=env, capture_output=True, text=True, timeout=600)
            line = [x for x in p.stdout.splitlines() if x.startswith("AB_CHILD ")]
            if p.returncode or not line:
                print("child failed (%s): %s" % (l, p.stderr[-400:]), flush=True)
                return 1
            res[l].append(json.loads(line[0][9:]))
            print("round %d %-8s " % (r, l) + "  ".join("%s ao %.3f rays %.3f prim %.4f" % (s, v["ao"], v["rays"], v["primary"]) for s, v in res[l][-1].items()), flush=True)
    keys = ["ao", "rays", "primary"] + [k for k in ("frame2", "frame1", "ao4") if all(k in x[s] for l in libs for x in res[l] for s in x)]
    print("\nmedian over %d processes [min .. max] of each process's median batch (ms per launch)" % rounds)
    for s in scenes.split(","):
        for key in keys:
            row = []
            for l in libs:
                v = [x[s][key] for x in res[l]]
                row.append("%s %.4f [%.4f .. %.4f]" % (l, statistics.median(v), min(v), max(v)))
            print("%-9s %-8s " % (s, key) + "   ".join(row))
    return 0


if __name__ == "__main__":
    sys.exit(main())
