"""The stand-in scenes must be as hard as their namesakes (scenes.cpp states the targets): decisions about the
incoherent-ray machinery are taken on them, so a hairball that is lighter than a street scene decides nothing.
Counts come from the oracle's PROFILE_RT counters (query.hlsl:377-379,407-409) at the reference's cameras
(assets/scenes/*.ron), full triangle counts (README.md:27-34), a 480x270 frame."""
import numpy as np

W, H = 480, 270


def _counts(trx, orc, name):
    verts, counts = trx.gen_scene(name, 0, 1)
    flat = trx.flat_build(verts, counts)
    eye, look, fov = trx.scene_camera(name)
    view = orc.view_from_bytes(trx.view_from_camera(eye, look, fov, W, H))
    osc = orc.Scene.from_flat(flat)
    hits, st = osc.trace_primary(view, W, H, sem=orc.SEM_CPU)
    nn, nt = osc.count_per_ray(view, W, H, sem=orc.SEM_CPU)
    assert int(nn.sum()) == st.n_node and int(nt.sum()) == st.n_tri
    return verts, view, hits, nn.astype(np.float64), nt.astype(np.float64)


def test_hairball_class_is_a_worst_case(trx, orc):
    verts, view, hits, nn, nt = _counts(trx, orc, "hairball")
    assert verts.shape[0] == 2880000
    # the ball's silhouette from hairball.ron's camera: eye (0,0,7), fov 90, radius 4.5
    ys, xs = np.mgrid[0:H, 0:W]
    ndc_x = (xs + 0.0) / W * 2 - 1
    ndc_y = 1 - (ys + 0.0) / H * 2
    tan_half = np.tan(np.radians(90.0) / 2)
    d = np.stack([ndc_x * tan_half * W / H, ndc_y * tan_half, -np.ones_like(ndc_x)], axis=-1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    eye = np.array([0.0, 0.0, 7.0])
    closest = np.linalg.norm(np.cross(np.broadcast_to(eye, d.shape), d), axis=-1)  # distance of the ray to the centre
    inside = (closest < 4.4).reshape(-1)
    frac = inside.mean()
    assert 0.28 < frac < 0.34, frac                         # 31 % of a 16:9 frame
    hit = hits["prim"] != 0xFFFFFFFF
    assert hit[inside].mean() > 0.90, hit[inside].mean()    # strands fill the ball
    assert hit[~inside & (closest.reshape(-1) > 4.6)].mean() == 0.0
    # bistro-class costs 17.8 node visits per primary ray on the same builder
    assert nn[inside].mean() > 1.5 * 17.8, nn[inside].mean()
    assert nn.mean() > 17.8 and nt.mean() > 2 * 8.3, (nn.mean(), nt.mean())


def test_bistro_dense_matches_the_profile_rt_legend(trx, orc):
    verts, view, hits, nn, nt = _counts(trx, orc, "bistro_dense")
    assert verts.shape[0] == 3872303
    assert 25.0 < nn.mean() < 40.0, nn.mean()   # about 30 node visits per primary ray
    assert 10.0 < nt.mean() < 25.0, nt.mean()   # about 15 triangle tests
    assert (hits["prim"] != 0xFFFFFFFF).mean() > 0.95
