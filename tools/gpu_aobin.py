"""Experiment: how much would binning AO rays by direction octant buy?  (development aid)

Generates cosine-hemisphere rays from the primary hits with torch (not bit-identical to the
kernel's generator; only the distribution matters here) and times trx_trace_rays_dev on
  a) tile order (what the in-kernel AO pass sees, hit pixels only),
  b) octant bins inside groups of G tiles,
  c) frame-wide octant bins, tile order inside a bin.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
dev = "cuda"


def make_rays(flat, view, w, h, prim_t, prim_id):
    m = np.frombuffer(bytes(view), dtype=np.float32)
    view_inv = torch.tensor(m[0:16].reshape(4, 4).T.copy(), device=dev)   # column-major storage
    proj_inv = torch.tensor(m[16:32].reshape(4, 4).T.copy(), device=dev)
    eye = torch.tensor(m[32:35].copy(), device=dev)
    py, px = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    cx = px.float() / w * 2 - 1
    cy = (1 - py.float() / h) * 2 - 1
    clip = torch.stack([cx, cy, torch.ones_like(cx), torch.ones_like(cx)], -1).reshape(-1, 4)
    v = clip @ proj_inv.T
    v = v / v[:, 3:4]
    wv = v @ view_inv.T
    d = wv[:, :3] - eye
    d = d / d.norm(dim=1, keepdim=True)
    hit = (prim_id != 0xffffffff)
    tv = torch.tensor(flat.tri_verts.reshape(-1, 3, 3), device=dev)
    pid = prim_id.clamp(max=tv.shape[0] - 1).long()
    e1 = tv[pid, 1] - tv[pid, 0]
    e2 = tv[pid, 2] - tv[pid, 0]
    n = torch.linalg.cross(e1, e2)
    n = n / n.norm(dim=1, keepdim=True).clamp(min=1e-30)
    n = n * torch.sign((n * -d).sum(1, keepdim=True))
    o = eye + d * prim_t[:, None] - d * 0.01
    g = torch.Generator(device=dev).manual_seed(1)
    u1 = torch.rand(w * h, device=dev, generator=g)
    u2 = torch.rand(w * h, device=dev, generator=g) * 6.2831853
    r = u1.sqrt()
    loc = torch.stack([r * u2.cos(), r * u2.sin(), (1 - u1).clamp(min=0).sqrt()], -1)
    # orthonormal basis around n (Duff et al.)
    s = torch.where(n[:, 2] >= 0, 1.0, -1.0)
    a = -1.0 / (s + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    t1 = torch.stack([1 + s * n[:, 0] * n[:, 0] * a, s * b, -s * n[:, 0]], -1)
    t2 = torch.stack([b, s + n[:, 1] * n[:, 1] * a, -n[:, 1]], -1)
    ad = t1 * loc[:, 0:1] + t2 * loc[:, 1:2] + n * loc[:, 2:3]
    ad = ad / ad.norm(dim=1, keepdim=True)
    rays = torch.zeros(w * h, 8, device=dev)
    rays[:, 0:3] = o
    rays[:, 3] = 0.0
    rays[:, 4:7] = ad
    rays[:, 7] = 3.4028234663852886e38
    return rays, hit


def time_rays(sc, rays, reps=10):
    n = rays.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for i in range(reps):
        ev0.record()
        sc.trace_rays_dev(rays.data_ptr(), n, out.data_ptr(), sem=3)
        ev1.record()
        torch.cuda.synchronize()
        if i >= 3:
            best = min(best, ev0.elapsed_time(ev1))
    return best, out


for name in sys.argv[1:] or ["bistro"]:
    w, h = 1920, 1080
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    prim = torch.zeros(w * h, dtype=torch.int64, device=dev)
    sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=3)
    torch.cuda.synchronize()
    pv = prim.view(torch.int32).reshape(-1, 2)
    prim_t = pv[:, 0].contiguous().view(torch.float32)
    prim_id = pv[:, 1].long() & 0xffffffff
    rays, hit = make_rays(flat, view, w, h, prim_t, prim_id)
    # tile order: 8x8 tiles row-major, pixels row-major inside a tile
    idx = torch.arange(w * h, device=dev).reshape(h // 8, 8, w // 8, 8).permute(0, 2, 1, 3).reshape(-1)
    idx = idx[hit[idx]]
    base = rays[idx].contiguous()
    n = base.shape[0]
    t_tile, ref = time_rays(sc, base)
    print("%s: %d AO rays; tile order %.3f ms (%.0f Mrays/s)" % (name, n, t_tile, n / t_tile / 1e3), flush=True)
    d = base[:, 4:7]
    octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
    pos = torch.arange(n, device=dev)
    for group in (8, 32, 128, 512, 1 << 30):
        key = (pos // (64 * group)) * 8 + octant
        order = torch.sort(key, stable=True).indices
        t, out = time_rays(sc, base[order].contiguous())
        same = bool((out == ref[order]).all())
        print("   octant bins inside %s tiles: %.3f ms (x%.2f) same=%s" % (group if group < 1 << 30 else "all", t, t_tile / t, same),
              flush=True)
    # finer direction bins: 8 octants x dominant axis (24 bins)
    dom = d.abs().argmax(1)
    for group in (128, 1 << 30):
        key = (pos // (64 * group)) * 24 + octant * 3 + dom
        order = torch.sort(key, stable=True).indices
        t, out = time_rays(sc, base[order].contiguous())
        print("   24 direction bins inside %s tiles: %.3f ms (x%.2f)" % (group if group < 1 << 30 else "all", t, t_tile / t), flush=True)
    # origin sort: Morton code of the ray origin (10 bits per axis), alone and inside direction octants
    o = base[:, 0:3]
    lo, hi = o.min(0).values, o.max(0).values
    q = ((o - lo) / (hi - lo).clamp(min=1e-20) * 1023.0).long().clamp(0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    for label, key in (("Morton order of the origins", morton), ("octant, then Morton order of the origins", (octant << 30) | morton)):
        order = torch.sort(key, stable=True).indices
        t, out = time_rays(sc, base[order].contiguous())
        same = bool((out == ref[order]).all())
        print("   %s: %.3f ms (x%.2f) same=%s; the sort itself (torch.sort of %d keys): " % (label, t, t_tile / t, same, n), end="", flush=True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            o2 = torch.sort(key, stable=True).indices
            _ = base[o2]
        ev1.record()
        torch.cuda.synchronize()
        print("%.3f ms with the gather" % (ev0.elapsed_time(ev1) / 5), flush=True)
    sc.close()
