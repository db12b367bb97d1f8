"""[needs a development build: make -C tray_racing_amd/csrc KFLAGS=-DTRX_DEV_TUNE OUT=... and TRX_LIB pointing at it]
Ablation timing (development aid; results of the ablated runs are wrong by design): frame time and wave-level trip
counts with parts of the trip switched off by TRX_TUNE bits, to price the parts in situ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
for name in sys.argv[1:] or ["bistro"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    for tune, label in ((0, "full trip"), (2, "no triangle phase (rays never shorten)")):
        os.environ["TRX_TUNE"] = str(tune)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        trips = st.n_wave_node
        print("%s | %-40s | %.3f ms | %8d wave-level node steps, %8d triangle rounds | %.0f wave-cycles per node step at 4096 waves, 2.4 GHz" % (
            name, label, mn, trips, st.n_wave_tri, mn * 1e-3 * 2.4e9 * 4096 / max(trips, 1)), flush=True)
    os.environ["TRX_TUNE"] = "0"
    sc.close()
