"""One 15360x8640 frame (132.7 M primary rays, 1 GB of hit records) on the kitchen-class scene, a sample of tiles
checked against the oracle: index arithmetic at sizes far beyond the bench (development aid)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import tray_racing_amd as T
from tray_racing_amd import dist as D
from oracle import binding as O
w, h = 15360, 8640   # 132.7 M rays, 1 GB of hits
verts, counts = T.gen_scene("kitchen", 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera("kitchen")
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
out = torch.empty(w * h, dtype=torch.int64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(3):
    ev0.record(); sc.trace_primary_dev(view, w, h, out.data_ptr(), sem=3); ev1.record(); torch.cuda.synchronize()
sc.check()
print("16K frame: %.2f ms = %.0f Mrays/s" % (ev0.elapsed_time(ev1), w * h / ev0.elapsed_time(ev1) / 1e3))
# oracle on one tile in 4096
osc = O.Scene.from_flat(flat); ov = O.view_from_bytes(view)
full = np.zeros(w * h, dtype=O.HIT_DTYPE)
osc.trace_primary(ov, w, h, sem=3, shard=(7, 4096), out=full)
tx = w // 8
tiles = np.arange(7, tx * (h // 8), 4096)
ys, xs = np.divmod(np.arange(64), 8)
px = (tiles[:, None] % tx) * 8 + xs[None, :]; py = (tiles[:, None] // tx) * 8 + ys[None, :]
idx = (py.astype(np.int64) * w + px).ravel()
got = D.int64_to_hits(out[torch.from_numpy(idx).cuda()])
want = full[idx]
print("sample of %d rays: t equal %s prim equal %s" % (idx.size, (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), (got["prim"] == want["prim"]).all()))
