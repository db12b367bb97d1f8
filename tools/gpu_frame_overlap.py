import sys, os
sys.path.insert(0, os.getcwd())
import tray_racing_amd as T
lib = T.load()
for name in ("bistro", "hairball"):
    v, c = T.gen_scene(name, 0, 1)
    flat = T.flat_build(v, c, preset="medium_build")
    eye, look, fov = T.scene_camera(name)
    w, h = 1920, 1080
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    for variant in (0, 8 << 8, 12 << 8):
        lib.trx_set_kernel_variant(variant)
        sc.frame_loop(view, w, h, sem=3, frames=8, overlap=False, fetch=False)
        sc.frame_loop(view, w, h, sem=3, frames=8, overlap=True, fetch=False)
        ser = min(sc.frame_loop(view, w, h, sem=3, frames=48, overlap=False, fetch=False)[0] for _ in range(3)) / 48
        ovl = min(sc.frame_loop(view, w, h, sem=3, frames=48, overlap=True, fetch=False)[0] for _ in range(3)) / 48
        print(name, "variant 0x%x: serial %.4f overlapped %.4f ms per frame" % (variant, ser, ovl), flush=True)
    lib.trx_set_kernel_variant(0)
    sc.close()
