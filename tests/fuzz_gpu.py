"""Randomised differential test: the HIP path against the oracle, bit for bit, over random scenes,
cameras, image sizes, semantics, triangle formats, TLAS on/off, shards, batch launches and all three
query kinds (primary, AO, explicit rays).  Test infrastructure (it drives the oracle); not collected by
pytest — tests/test_gpu_parity.py runs a short fixed-seed slice of it.

    python tests/fuzz_gpu.py --minutes 10 [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

SCENES = ("cornell", "kitchen", "bistro", "bistro_dense", "hairball", "san_miguel", "soup", "demoscene")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def differs(got, want):
    bt = np.flatnonzero(bits(got["t"]) != bits(want["t"]))
    bp = np.flatnonzero(got["prim"] != want["prim"])
    return (bt.size, bp.size, bt[:3].tolist(), bp[:3].tolist()) if bt.size or bp.size else None


BIG = False   # --big: few large cases (up to 1 M triangles, up to 1920x1080) instead of many small ones


def instanced_case(T, O, rng, case):
    """Random instanced scene (several transformed instances per object, shear and non-uniform scale included): explicit
    rays, or a primary + AO frame, with the instance ids beside the hits."""
    from helpers import aimed_rays, instanced_scene, random_rays
    seed = int(rng.integers(1, 1 << 30))
    kind = str(rng.choice(["soup", "soup", "cornell", "kitchen"]))
    n_inst = int(rng.integers(1, 40))
    tris = int(rng.choice([40, 400, 3000])) if kind == "soup" else (0 if kind == "cornell" else int(rng.choice([3000, 20000])))
    sem = int(rng.integers(8))
    flat, _o2w, world, _first, _b = instanced_scene(T, seed=seed % 100000, n_objects=int(rng.integers(1, 6)), n_instances=n_inst,
                                                   tris_per_object=tris, kind=kind, spread=float(rng.uniform(0.5, 6.0)))
    sc = T.Scene(flat)
    osc = O.Scene(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start, instance_w2o=sc.instance_world_to_object())
    desc = "case %d: instanced %s x%d seed=%d sem=%d" % (case, kind, n_inst, seed, sem)
    bad = []
    try:
        if rng.integers(2):
            wflat = type("W", (), {"tri_verts": world})
            rays = np.concatenate([random_rays(T, wflat, int(rng.integers(1, 4000)), seed), aimed_rays(T, world, int(rng.integers(1, 12000)), seed + 1)])
            got, gi, _ = sc.trace_rays_inst(rays, sem=sem)
            want, wi, _ = osc.trace_rays_inst(rays, sem=sem)
            bad += [("inst rays", differs(got, want)), ("inst ids", (int((gi != wi).sum()),) if (gi != wi).any() else None)]
            if rng.integers(2):   # round 6: the same rays one by one through the ray service (two levels, instance transforms)
                sub = rays[: int(rng.integers(1, 2500))]
                one, _secs, _starts = sc.traverse_threads(sub, threads=int(rng.integers(1, 13)), sem=sem)
                hit = want["prim"][: sub.shape[0]] != 0xFFFFFFFF
                n_b = int((bits(one["t"]) != bits(want["t"][: sub.shape[0]])).sum()) + int((one["instance_id"][hit] != wi[: sub.shape[0]][hit]).sum())
                bad += [("inst traverse1", (n_b,) if n_b else None)]
        else:
            w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
            lo, hi = world.reshape(-1, 3).min(0), world.reshape(-1, 3).max(0)
            eye = (lo + rng.uniform(-0.3, 1.3, 3) * (hi - lo + 1e-3)).tolist()
            view = T.view_from_camera(eye, (0.5 * (lo + hi)).tolist(), float(rng.uniform(30, 110)), w, h)
            ov = O.view_from_bytes(bytes(view))
            frame = int(rng.integers(0, 5000))
            gp, gpi, gao, gaoi, _ = sc.trace_primary_ao_inst(view, w, h, sem=sem, frame=frame, ao_eps=0.01)
            wp, wpi, _ = osc.trace_primary_inst(ov, w, h, sem=sem)
            wao, waoi, _ = osc.trace_ao_inst(ov, w, h, wp, wpi, sem=sem, frame=frame, ao_eps=0.01)
            bad += [("inst primary", differs(gp, wp)), ("inst ao", differs(gao, wao)),
                    ("inst ids", (int((gpi != wpi).sum()) + int((gaoi != waoi).sum()),) if ((gpi != wpi).any() or (gaoi != waoi).any()) else None)]
    finally:
        sc.close()
    return desc, [(k, v) for k, v in bad if v]


def one_case(T, O, rng, case):
    import torch
    from helpers import random_rays
    from tray_racing_amd import dist as D
    if not BIG and rng.integers(8) == 0:
        return instanced_case(T, O, rng, case)
    name = SCENES[int(rng.integers(len(SCENES)))]
    n = int(rng.choice([200000, 500000, 1000000] if BIG else [1, 2, 3, 17, 300, 2500, 20000, 90000]))
    if name in ("cornell",):
        n = 0
    seed = int(rng.integers(1, 1 << 30))
    tlas = bool(rng.integers(2))
    sem = int(rng.integers(8))
    fmt = int(rng.choice([T.TRI_VERTS_36, T.TRI_VERTS_36, T.TRI_EDGES_36, T.TRI_F16_24]))
    w, h = (int(rng.integers(300, 1921)), int(rng.integers(200, 1081))) if BIG else (int(rng.integers(1, 200)), int(rng.integers(1, 120)))
    verts, counts = T.gen_scene(name, n, seed)
    leaf = int(rng.integers(1, 4))
    split = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
    if rng.integers(4) == 0:   # the ploc_cwbvh pipeline with random BvhBuildParams
        bp = T.build_params(ploc_search_distance=int(rng.integers(1, 33)), search_depth_threshold=int(rng.integers(0, 6)),
                            sort_precision=int(rng.choice([64, 128])), reinsertion_batch_ratio=float(rng.choice([0.0, 0.05, 0.15, 1.5])),
                            max_prims_per_leaf=leaf, pre_split=int(split > 0))
        flat = T.flat_build_params(verts, counts, bp, use_tlas=tlas)
    else:
        flat = T.flat_build(verts, counts, use_tlas=tlas, max_prims_per_leaf=leaf, split=split)
    T.flat_build(verts[:1], split=0.0)
    eye, look, fov = T.scene_camera(name)
    pts = flat.tri_verts.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    if rng.integers(3):
        eye = tuple((lo + rng.uniform(-0.2, 1.2, 3) * (hi - lo + 1e-3)).tolist())
        look = tuple((lo + rng.uniform(0, 1, 3) * (hi - lo + 1e-3)).tolist())
        fov = float(rng.uniform(20, 120))
    view = T.view_from_camera(eye, look, fov, w, h)
    ov = O.view_from_bytes(view)
    desc = "case %d: %s n=%d seed=%d tlas=%s sem=%d fmt=%d %dx%d leaf<=%s" % (case, name, n, seed, tlas, sem, fmt, w, h, leaf) + " split=%.1f" % split
    if fmt == T.TRI_F16_24:
        packed = T.pack_tris_f16(flat.tri_verts)
        sc = T.Scene(flat, tri_format=fmt, tri_bytes=packed)
        osc = O.Scene(flat.nodes, None, flat.instance_offsets, int(flat.tlas_start), tri_f16=packed,
                      instance_entry=flat.instance_entry)
    elif fmt == T.TRI_EDGES_36:   # {v0, e1, e2} handed over as the oracle derives them
        osc = O.Scene.from_flat(flat)
        sc = T.Scene(flat, tri_format=fmt, tri_bytes=osc.tris.copy())
    else:
        sc = T.Scene(flat)
        osc = O.Scene.from_flat(flat)
    bad = []
    kind = int(rng.integers(7))
    try:
        if kind == 6:      # round 6: the literal single-ray traverse from a few host threads (the resident ray service, both levels)
            rays = random_rays(T, flat, int(rng.integers(1, 3000)), seed)
            got, _secs, _starts = sc.traverse_threads(rays, threads=int(rng.integers(1, 13)), sem=sem)
            batch, _ms = sc.traverse_batch(rays, sem=sem)
            want = osc.trace_rays(rays, sem=sem)[0]
            n_t = int((bits(got["t"]) != bits(want["t"])).sum())
            n_f = sum(int((got[f] != batch[f]).sum()) for f in ("primitive_id", "geometry_id", "instance_id")) + int((bits(got["t"]) != bits(batch["t"])).sum())
            bad += [("traverse1 t vs oracle", (n_t,) if n_t else None), ("traverse1 vs traverse_batch", (n_f,) if n_f else None)]
        elif kind == 4:      # round 4: n AO frames in one launch (trx_trace_ao_batch_dev), odd frame stride
            m, frame0, eps = int(rng.integers(2, 9)), int(rng.integers(0, 5000)), float(rng.choice([0.01, 0.0001]))
            stride = w * h + int(rng.integers(0, 7))
            d_p = torch.empty(w * h, dtype=torch.int64, device="cuda")
            d_a = torch.full((m * stride,), -1, dtype=torch.int64, device="cuda")
            sc.trace_primary_dev(view, w, h, d_p.data_ptr(), sem=sem)
            sc.trace_ao_batch_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), stride, m, sem=sem, frame0=frame0, ao_eps=eps)
            sc.check()
            wp, _ = osc.trace_primary(ov, w, h, sem=sem)
            bad += [("primary", differs(D.int64_to_hits(d_p), wp))]
            for f in range(m):
                wa, _ = osc.trace_ao(ov, w, h, wp, sem=sem, frame=frame0 + f, ao_eps=eps)
                bad += [("ao batch frame %d/%d" % (f, m), differs(D.int64_to_hits(d_a[f * stride:f * stride + w * h]), wa))]
        elif kind == 5:    # round 4: the one-launch frame (trx_trace_frame_dev; two launches inside for two-level scenes)
            frame, eps = int(rng.integers(0, 5000)), float(rng.choice([0.01, 0.0001]))
            d_p = torch.full((w * h,), -1, dtype=torch.int64, device="cuda")
            d_a = torch.full((w * h,), -1, dtype=torch.int64, device="cuda")
            T.load().trx_set_kernel_variant(int(rng.choice([0, 1, 24, 64, 8 | (1 << 14), 32 | (3 << 14)])))
            try:
                sc.trace_frame_dev(view, w, h, d_p.data_ptr(), d_a.data_ptr(), sem=sem, frame=frame, ao_eps=eps)
                sc.check()
            finally:
                T.load().trx_set_kernel_variant(0)
            wp, _ = osc.trace_primary(ov, w, h, sem=sem)
            wa, _ = osc.trace_ao(ov, w, h, wp, sem=sem, frame=frame, ao_eps=eps)
            bad += [("frame primary", differs(D.int64_to_hits(d_p), wp)), ("frame ao", differs(D.int64_to_hits(d_a), wa))]
        elif kind == 0:      # primary + AO, host entry point
            frame, eps = int(rng.integers(0, 5000)), float(rng.choice([0.01, 0.0001]))
            prim, ao, _ = sc.trace_primary_ao(view, w, h, sem=sem, frame=frame, ao_eps=eps)
            wp, _ = osc.trace_primary(ov, w, h, sem=sem)
            wa, _ = osc.trace_ao(ov, w, h, wp, sem=sem, frame=frame, ao_eps=eps)
            bad += [("primary", differs(prim, wp)), ("ao", differs(ao, wa))]
        elif kind == 1:    # explicit rays
            rays = random_rays(T, flat, int(rng.integers(1, 1500000 if BIG else 30000)), seed)
            got, _ = sc.trace_rays(rays, sem=sem)
            want = osc.trace_rays(rays, sem=sem)[0]
            bad += [("rays", differs(got, want))]
            occ = sc.trace_occluded(rays, sem=sem)[0].astype(bool)   # any-hit: same hit / no hit
            n_occ = int((occ != (want["prim"] != 0xffffffff)).sum())
            bad += [("occluded", (n_occ,) if n_occ else None)]
        elif kind == 2:    # tile shards, compact layout, gathered by hand
            world = int(rng.integers(2, 9))
            fg = D.FrameGather(w, h, 0, world, "cuda")
            for r in range(world):
                blk = fg.flat[r * fg.records:(r + 1) * fg.records]
                sc.trace_primary_dev(view, w, h, blk.data_ptr(), sem=sem, shard=(r, world, 1))
            sc.check()
            bad += [("shards/%d" % world, differs(D.int64_to_hits(fg.assemble()), osc.trace_primary(ov, w, h, sem=sem)[0]))]
        else:              # several frames in one launch, different cameras
            m = int(rng.integers(2, 9))
            views = [T.view_from_camera(tuple((np.array(eye) + 0.05 * f * (hi - lo)).tolist()), look, fov, w, h) for f in range(m)]
            out = torch.empty(m * w * h, dtype=torch.int64, device="cuda")
            for _ in range(2):
                sc.trace_primary_batch_dev(views, w, h, out.data_ptr(), w * h, sem=sem)
            sc.check()
            for f in range(m):
                want, _ = osc.trace_primary(O.view_from_bytes(views[f]), w, h, sem=sem)
                bad += [("batch frame %d/%d" % (f, m), differs(D.int64_to_hits(out[f * w * h:(f + 1) * w * h]), want))]
    finally:
        sc.close()
    bad = [(k, v) for k, v in bad if v]
    return desc + " kind=%d" % kind, bad


def run(minutes=1.0, seed=1, max_cases=1 << 30, verbose=True):
    import tray_racing_amd as T
    from oracle import binding as O
    rng = np.random.default_rng(seed)
    t_end = time.time() + 60.0 * minutes
    case, failures = 0, []
    while time.time() < t_end and case < max_cases:
        desc, bad = one_case(T, O, rng, case)
        if bad:
            failures.append((desc, bad))
            print("MISMATCH", desc, bad, flush=True)
        elif verbose and case % 20 == 0:
            print("ok  ", desc, flush=True)
        case += 1
    print("%d cases, %d with mismatches" % (case, len(failures)), flush=True)
    return case, failures


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    BIG = a.big
    _, fails = run(a.minutes, a.seed)
    sys.exit(1 if fails else 0)
