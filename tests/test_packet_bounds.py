"""The packet test of a wave-uniform node step (kernels.hip node_intersect_kept, trace_walk_plain.inc) restated in numpy
float32 and checked as a PROPERTY: a child the packet test leaves out is missed by every ray of the packet.  The bounds
are the per-ray test's own multiplications and additions evaluated at the ends of the rays' 1/d interval; what makes them
bounds is that round-to-nearest products and sums are monotone in each operand - this test does not assume it, it looks.
Also the literal-division variants: b = RN(c / d) per ray against the interval of RN(c RN(1/d)) moved out by 2^-21."""
import numpy as np

F = np.float32


def per_ray_slabs(o, d, p, e, q_lo, q_hi, literal):
    """tmin / tmax of every (ray, child) as node_intersect computes them (query.hlsl:237-300), without the t clamp.
    o, d: [R, 3]; p: [3]; e: [3] (powers of two); q_lo, q_hi: [8, 3] bytes.  literal: b = (p - o) / d, else (p - o) * (1/d)."""
    ix = F(1.0) / d
    a = e[None, :] * ix
    c = p[None, :] - o
    b = (c / d) if literal else (c * ix)
    neg = d < 0
    qn = np.where(neg[:, None, :], q_hi[None], q_lo[None]).astype(F)   # near plane: the max plane where d < 0
    qf = np.where(neg[:, None, :], q_lo[None], q_hi[None]).astype(F)
    tn = qn * a[:, None, :] + b[:, None, :]
    tf = qf * a[:, None, :] + b[:, None, :]
    tmin = np.maximum(tn.max(axis=2), F(0.0001))
    tmax = tf.min(axis=2)
    return tmin, tmax


def packet_keep(o, d, p, e, q_lo, q_hi, literal):
    """The packet test: [8] bools, False = no ray of the packet can enter the child."""
    ix = F(1.0) / d
    lo, hi = ix.min(axis=0), ix.max(axis=0)            # per axis; one sign per axis (the caller's packets share an octant)
    neg = lo < 0
    c = p - o[0]
    alo, ahi = e * lo, e * hi
    b0, b1 = c * lo, c * hi
    blo, bhi = np.minimum(b0, b1), np.maximum(b0, b1)
    if literal:
        blo = (blo - np.abs(blo) * F(2.0 ** -21)).astype(F)
        bhi = (bhi + np.abs(bhi) * F(2.0 ** -21)).astype(F)
    qn = np.where(neg[None, :], q_hi, q_lo).astype(F)
    qf = np.where(neg[None, :], q_lo, q_hi).astype(F)
    lb = qn * alo[None, :] + blo[None, :]
    ub = qf * ahi[None, :] + bhi[None, :]
    return ~(np.maximum(lb.max(axis=1), F(0.0001)) > ub.min(axis=1))


def random_packet(rng, spread):
    """64 rays from one origin inside a narrow cone (an 8x8 tile), all in one octant, and a node frame in front of them."""
    o = np.tile(rng.uniform(-50, 50, 3).astype(F), (64, 1))
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    d = (axis[None, :] + spread * rng.uniform(-1, 1, (64, 3))).astype(F)
    d[np.abs(d) < 1e-6] = F(1e-6)
    same = (np.sign(d) == np.sign(d[0])).all()
    e_exp = rng.integers(-8, 6, 3)
    e = (F(2.0) ** e_exp.astype(F)).astype(F)
    centre = o[0].astype(np.float64) + axis * rng.uniform(0.5, 400) + rng.normal(size=3) * rng.uniform(0, 60)
    p = (centre - 128.0 * e.astype(np.float64) * rng.uniform(0, 1.5, 3)).astype(F)
    q_lo = rng.integers(0, 250, (8, 3))
    q_hi = np.minimum(q_lo + rng.integers(0, 60, (8, 3)), 255)
    return same, o, d, p, e, q_lo.astype(np.uint8), q_hi.astype(np.uint8)


def test_a_child_the_packet_test_leaves_out_is_missed_by_every_ray():
    rng = np.random.default_rng(11)
    culled = kept = packets = 0
    for literal in (False, True):
        for k in range(4000):
            same, o, d, p, e, q_lo, q_hi = random_packet(rng, spread=rng.choice([0.002, 0.01, 0.05]))
            if not same:
                continue   # (the kernel does not run the packet test on a wave whose rays span octants)
            packets += 1
            tmin, tmax = per_ray_slabs(o, d, p, e, q_lo, q_hi, literal)
            keep = packet_keep(o, d, p, e, q_lo, q_hi, literal)
            enters = (tmin <= tmax).any(axis=0)          # per child: some ray enters it
            assert not (enters & ~keep).any(), (literal, k)
            culled += int((~keep).sum())
            kept += int(keep.sum())
    # ... and the test is worth running: it leaves out most children, and what it keeps is mostly entered by some ray
    assert packets > 6000 and culled > 2 * kept


def test_the_bounds_bound_plane_by_plane():
    """Finer than the mask: for every child and axis the packet's near bound is <= every ray's near parameter and its far
    bound >= every ray's far parameter."""
    rng = np.random.default_rng(12)
    for literal in (False, True):
        for k in range(1500):
            same, o, d, p, e, q_lo, q_hi = random_packet(rng, spread=0.02)
            if not same:
                continue
            ix = F(1.0) / d
            a = e[None, :] * ix
            c = p[None, :] - o
            b = (c / d) if literal else (c * ix)
            neg = d[0] < 0
            qn = np.where(neg[None, :], q_hi, q_lo).astype(F)
            qf = np.where(neg[None, :], q_lo, q_hi).astype(F)
            tn = qn[None] * a[:, None, :] + b[:, None, :]
            tf = qf[None] * a[:, None, :] + b[:, None, :]
            lo, hi = ix.min(axis=0), ix.max(axis=0)
            alo, ahi = e * lo, e * hi
            b0, b1 = c[0] * lo, c[0] * hi
            blo, bhi = np.minimum(b0, b1), np.maximum(b0, b1)
            if literal:
                blo = (blo - np.abs(blo) * F(2.0 ** -21)).astype(F)
                bhi = (bhi + np.abs(bhi) * F(2.0 ** -21)).astype(F)
            assert (qn * alo[None, :] + blo[None, :] <= tn.min(axis=0)).all(), (literal, k)
            assert (qf * ahi[None, :] + bhi[None, :] >= tf.max(axis=0)).all(), (literal, k)
