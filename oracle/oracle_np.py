"""Second, independent restatement of the deterministic part of the path in plain Python floats (IEEE
binary64, no contraction), written from the Rust source without looking at oracle.cpp's structure.
TEST INFRASTRUCTURE ONLY.  It exists to catch transcription errors in oracle.cpp: two restatements that
agree bit for bit on hit ids, t, hit records and quantised bytes are unlikely to share a mistake.
Slow (pure Python): use on small images only.
"""
import math

INF = float("inf")


def sphere_hit(center0, center1, moving, r, o, d, tm, tmin, tmax):
    """src/scene/sphere.rs:79-113.  Returns None or (t, p, outward_normal, front, u, v)."""
    r = max(r, 0.0)
    if moving:  # sphere.rs:61-70
        c = tuple(center0[i] + tm * (center1[i] - center0[i]) for i in range(3))
    else:
        c = tuple(center0)
    oc = (c[0] - o[0], c[1] - o[1], c[2] - o[2])
    a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    h = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]
    cc = (oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2]) - r * r
    disc = h * h - a * cc
    if disc < 0.0:
        return None
    sq = math.sqrt(disc)
    t = (h - sq) / a
    if not (tmin < t < tmax):
        t = (h + sq) / a
    if not (tmin < t < tmax):
        return None
    p = (o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t)
    n = ((p[0] - c[0]) / r, (p[1] - c[1]) / r, (p[2] - c[2]) / r)
    front = (d[0] * n[0] + d[1] * n[1] + d[2] * n[2]) < 0.0
    theta = -math.acos(n[1])
    phi = -math.atan2(n[2], n[0]) + math.pi
    u = 0.5 * phi * 0.3183098861837907  # FRAC_1_PI
    v = theta * 0.3183098861837907
    return t, p, n, front, u, v


def primary_hits(spheres, cam):
    """camera.rs:247-254 + closest hit in (0.001, inf).  spheres: list of (c0, c1, moving, r).
    cam: dict with lookfrom, px_du, px_dv, vp_upper_left, W, H.  Returns (ids, ts) as nested lists."""
    ids, ts = [], []
    lf, du, dv, ul = cam["lookfrom"], cam["px_du"], cam["px_dv"], cam["vp_upper_left"]
    for hn in range(cam["H"]):
        row_i, row_t = [], []
        for wn in range(cam["W"]):
            pc = tuple((ul[i] + (du[i] * float(wn))) + (dv[i] * float(hn)) for i in range(3))
            d = (pc[0] - lf[0], pc[1] - lf[1], pc[2] - lf[2])
            best, bid = INF, 0xFFFFFFFF
            for k, (c0, c1, mv, r) in enumerate(spheres):
                h = sphere_hit(c0, c1, mv, r, lf, d, 0.0, 0.001, best)
                if h is not None:
                    best, bid = h[0], k
            row_i.append(bid)
            row_t.append(best)
        ids.append(row_i)
        ts.append(row_t)
    return ids, ts


def quantize(x):
    """core/rgb.rs:20-36 + utils/math.rs:13-21.  None for NaN (the reference panics)."""
    g = math.sqrt(x) if x > 0.0 else 0.0
    v = g * 254.999
    if v != v:
        return None
    c = min(max(v, 0.0), 255.0)
    f = math.floor(c)  # c >= 0 here; f64::round = half away from zero
    return int(f) + (1 if c - f >= 0.5 else 0)
