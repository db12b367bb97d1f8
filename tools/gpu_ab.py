"""A/B timing of tuning builds (development aid): every tuning_libs/*.so runs the same frames in its own process, twice
round robin; prints min / mean primary ms, the primary + AO frame ms and a checksum of both hit buffers per scene.
usage: python tools/gpu_ab.py [scene,scene,...] [passes]"""
import glob
import os
import subprocess
import sys
import zlib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

if os.environ.get("TRX_AB_CHILD"):
    import numpy as np
    import tray_racing_amd as T
    w, h = 1920, 1080
    for name in sys.argv[1].split(","):
        verts, counts = T.gen_scene(name, 0, 1)
        flat = T.flat_build(verts, counts)
        eye, look, fov = T.scene_camera(name)
        view = T.view_from_camera(eye, look, fov, w, h)
        sc = T.Scene(flat)
        lib = T.load()
        import torch
        mbuf = torch.empty(w * h, dtype=torch.int64, device="cuda")
        for variant in [int(x, 0) for x in os.environ.get("TRX_AB_VARIANTS", "0").split(",")]:
            # a camera that moves (bench.py's leg): the tile order is learnt from the previous view
            lib.trx_set_kernel_variant(variant)
            for step in [float(x) for x in os.environ.get("TRX_AB_STEPS", "0.05").split(",")]:
                mv = []
                for f in range(72):
                    off = step * f
                    v = T.view_from_camera((eye[0] + off, eye[1], eye[2]), (look[0] + off, look[1], look[2]), fov, w, h)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    sc.trace_primary_dev(v, w, h, mbuf.data_ptr(), sem=3)
                    b.record()
                    mv.append((a, b))
                torch.cuda.synchronize()
                mt = [a.elapsed_time(b) for a, b in mv][8:]
                mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=10, frames=40)
                print("AB %-14s variant 0x%x camera step %.2f m: moving min %.4f mean %.4f | static min %.4f mean %.4f" % (
                    name, variant, step, min(mt), sum(mt) / len(mt), mn, mean), flush=True)
        lib.trx_set_kernel_variant(0)
        mns, means = [], []
        for rep in range(3):
            mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=10, frames=40)
            mns.append(mn)
            means.append(mean)
        ao_ms = 1e9
        for f in range(8):
            prim, ao, ms = sc.trace_primary_ao(view, w, h, sem=3, frame=f % 4, ao_eps=0.01)
            if f >= 3:
                ao_ms = min(ao_ms, ms)
        prim, ao, _ = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
        crc = zlib.crc32(ao.tobytes(), zlib.crc32(prim.tobytes()))
        print("AB %-14s primary min %.4f mean %.4f | primary+AO %.4f ms | crc %08x" % (name, min(mns), sum(means) / 3, ao_ms, crc), flush=True)
        sc.close()
    sys.exit(0)

scenes = sys.argv[1] if len(sys.argv) > 1 else "bistro,bistro_dense,hairball"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
libs = sorted(glob.glob(os.path.join(root, "tuning_libs", "*.so")))
for p in range(passes):
    for lib in libs:
        env = dict(os.environ, TRX_LIB=lib, TRX_AB_CHILD="1")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), scenes], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("AB ")]
        if not lines:
            print(os.path.basename(lib), "FAILED", out.stderr[-400:], flush=True)
        for l in lines:
            print("pass %d %-24s %s" % (p, os.path.basename(lib), l[3:]), flush=True)
