"""Round 4: how a launch slot's frame time settles from its first frame on (the driver's bench protocol times frames
7..26 of a slot): per-frame hipEvent times of N frames on a fresh stream (= a fresh launch slot), R repetitions, per
kernel-variant word.
usage: python tools/gpu_settle.py [scene] [frames] [reps] [variant ...]   (variants as for trx_set_kernel_variant)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 48
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
variants = [int(v, 0) for v in sys.argv[4:]] or [0]
w, h = 1920, 1080
lib = T.load()
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
chain = os.environ.get("CHAIN", "0") == "1"   # one scene for all repetitions: the GPU never idles between them; a new stream = a new slot
shared = T.Scene(flat) if chain else None
# development builds (TRX_LIB=tuning_libs/dev.so): SETTINGS="tune:freeze,..." runs every variant under each TRX_TUNE word /
# TRX_FREEZE_AFTER count (the library reads both from the environment at every launch)
settings = [x.split(":") for x in os.environ.get("SETTINGS", "").split(",") if x] or [None]
for setting in settings:
  if setting:
      os.environ["TRX_TUNE"] = setting[0]
      os.environ["TRX_FREEZE_AFTER"] = setting[1] if len(setting) > 1 else "0"
      print("== TRX_TUNE %s TRX_FREEZE_AFTER %s" % (os.environ["TRX_TUNE"], os.environ["TRX_FREEZE_AFTER"]), flush=True)
  for variant in variants:
    for rep in range(reps):
          sc = shared if chain else T.Scene(flat)      # a fresh scene: fresh launch slots, nothing learnt
          lib.trx_set_kernel_variant(variant)
          out = torch.empty(w * h, dtype=torch.int64, device="cuda")
          warm = int(os.environ.get("WARM", "0"))   # frames on ANOTHER stream (another slot) first: the GPU's clocks are up
          if warm:
              ws = torch.cuda.Stream()
              with torch.cuda.stream(ws):
                  for f in range(warm):
                      sc.trace_primary_dev(view, w, h, out.data_ptr(), sem=3, stream=ws.cuda_stream)
              torch.cuda.synchronize()
          s = torch.cuda.Stream()
          evs = []
          with torch.cuda.stream(s):
              for f in range(n_frames):
                  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                  a.record(s)
                  sc.trace_primary_dev(view, w, h, out.data_ptr(), sem=3, stream=s.cuda_stream)
                  b.record(s)
                  evs.append((a, b))
          torch.cuda.synchronize()
          ts = [a.elapsed_time(b) for a, b in evs]
          lib.trx_set_kernel_variant(0)
          drv = ts[6:26]
          print("%s variant 0x%x rep %d: frames 7..26 mean %.4f min %.4f | last 16 mean %.4f" % (
              name, variant, rep, sum(drv) / len(drv), min(drv), sum(ts[-16:]) / 16), flush=True)
          print("   " + " ".join("%.3f" % t for t in ts), flush=True)
          if not chain:
              sc.close()
