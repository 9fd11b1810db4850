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


# ---- the stochastic part of the path (round 2): an independent f64 render in plain Python floats ----------------------
# Written from the Rust source and from DESIGN.md's RNG contract (Philox4x32-10 keyed (seed; pixel, sample, bounce, block);
# f64 dimension d = 64-bit word pair d % 2 of block d // 2, top 53 bits), NOT from oracle.cpp.  tests/test_oracle.py demands
# bit-equal linear radiance and bytes from the two restatements on small frames - lens, moving spheres, every material, all
# three textures included (both sides call the same libm for tan/acos/atan2/sin/cos).

M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Philox-4x32 with 10 rounds."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for r in range(10):
        if r:
            k0 = (k0 + 0x9E3779B9) & M32
            k1 = (k1 + 0xBB67AE85) & M32
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
    return c0, c1, c2, c3


def draw(seed, pixel, sample, bounce, dim):
    o = philox4x32_10((pixel, sample, bounce, dim // 2), (seed & M32, (seed >> 32) & M32))
    w = (o[2] | (o[3] << 32)) if dim % 2 else (o[0] | (o[1] << 32))
    return float(w >> 11) * 2.0 ** -53


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _mul(a, l):
    return (a[0] * l, a[1] * l, a[2] * l)


def _div(a, l):
    return (a[0] / l, a[1] / l, a[2] / l)


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _unit(a):  # point3.rs:39-46
    return _div(a, math.sqrt(_dot(a, a)))


def _as_i32(f):  # Rust `as i32`: saturating, NaN -> 0
    if f != f:
        return 0
    return max(-2147483648, min(2147483647, int(f)))


def _as_u32(x):  # Rust `as u32`
    if not x > 0.0:
        return 0
    return min(0xFFFFFFFF, int(x))


def texture_color(tex, u, v, p):
    """scene/texture.rs.  tex: ("solid", rgb) | ("checker", inv_scale, even, odd) | ("image", array(H, W, 3) or None)."""
    if tex[0] == "solid":
        return tex[1]
    if tex[0] == "checker":  # texture.rs:46-59
        s = 0
        for e in p:
            s += _as_i32(math.floor(e * tex[1]))
            s = ((s + 2 ** 31) % 2 ** 32) - 2 ** 31  # wrapping i32 add
        r = int(math.fmod(s, 2))  # Rust remainder: sign of the dividend
        return tex[2] if r == 0 else tex[3]
    img = tex[1]  # texture.rs:71-94
    if img is None or img.shape[0] == 0:
        return (0.0, 1.0, 1.0)
    H, W = img.shape[0], img.shape[1]
    u = min(max(u, 0.0), 1.0)
    v = min(max(abs(v), 0.0), 1.0)
    i = min(_as_u32(u * float(W)), W - 1)
    j = min(_as_u32(v * float(H)), H - 1)
    px = img[j, i]
    return (float(px[0]) / 255.0, float(px[1]) / 255.0, float(px[2]) / 255.0)


def scatter(mat, rd, tm, p, n, front, u, v, dr):
    """scene/material.rs.  mat: ("lambertian", tex, reflectance) | ("metal", albedo, fuzz or None) | ("dielectric", ior).
    dr(dim) -> uniform.  Returns None (absorbed) or (attenuation, origin, direction, time)."""
    if mat[0] == "lambertian":  # material.rs:59-84
        if not dr(0) < mat[2]:
            return None
        d = _add(n, _unit((dr(1), dr(2), dr(3))))  # positive-octant sampler, point3.rs:88-93
        if abs(d[0]) < 1e-8 and abs(d[1]) < 1e-8 and abs(d[2]) < 1e-8:
            d = n
        return _div(texture_color(mat[1], u, v, p), mat[2]), p, d, tm
    if mat[0] == "metal":  # material.rs:97-114
        refl = _sub(rd, _mul(_mul(n, 2.0), _dot(rd, n)))
        if mat[2] is not None:
            refl = _add(_unit(refl), _mul(_unit((dr(0), dr(1), dr(2))), mat[2]))
            if not _dot(refl, n) > 0.0:
                return None
        return mat[1], p, refl, tm
    ri = 1.0 / mat[1] if front else mat[1]  # material.rs:126-154
    ud = _unit(rd)
    cos_t = min(_dot((-ud[0], -ud[1], -ud[2]), n), 1.0)
    sin_t = math.sqrt(1.0 - cos_t * cos_t)
    reflect = ri * sin_t > 1.0
    if not reflect:
        r0 = (1.0 - ri) / (1.0 + ri)
        r0 = r0 * r0
        x = 1.0 - cos_t
        reflect = r0 + (1.0 - r0) * (x * ((x * x) * (x * x))) > dr(0)
    if reflect:
        d = _sub(ud, _mul(_mul(n, 2.0), _dot(ud, n)))
    else:  # refract, material.rs:157-166
        c2 = min(_dot((-ud[0], -ud[1], -ud[2]), n), 1.0)
        perp = _mul(_add(ud, _mul(n, c2)), ri)
        par = _mul(n, -math.sqrt(abs(1.0 - _dot(perp, perp))))
        d = _add(perp, par)
    return (1.0, 1.0, 1.0), p, d, tm


def closest_hit(spheres, o, d, tm):
    best, bid, rec = INF, None, None
    for k, (c0, c1, mv, r, _m) in enumerate(spheres):
        h = sphere_hit(c0, c1, mv, r, o, d, tm, 0.001, best)
        if h is not None:
            best, bid, rec = h[0], k, h
    return bid, rec


def color(spheres, o, d, tm, depth, seed, pixel, sample, bounce):
    """camera.rs:268-286, recursion kept."""
    if depth == 0:
        return (0.0, 0.0, 0.0), 0
    bid, rec = closest_hit(spheres, o, d, tm)
    if bid is not None:
        t, p, n_out, front, u, v = rec
        n = n_out if front else (-n_out[0], -n_out[1], -n_out[2])
        b = bounce + 1
        s = scatter(spheres[bid][4], d, tm, p, n, front, u, v, lambda dim: draw(seed, pixel, sample, b, dim))
        if s is None:
            return (0.0, 0.0, 0.0), 1
        att, so, sd, stm = s
        c, nseg = color(spheres, so, sd, stm, depth - 1, seed, pixel, sample, b)
        return (att[0] * c[0], att[1] * c[1], att[2] * c[2]), nseg + 1
    ud = _unit(d)
    a = 0.5 * (ud[1] + 1.0)
    return (1.0 * (1.0 - a) + 0.5 * a, 1.0 * (1.0 - a) + 0.7 * a, 1.0 * (1.0 - a) + 1.0 * a), 1


def render(spheres, cam, seed):
    """camera.rs:232-259.  spheres: list of (c0, c1, moving, r, material); cam: dict of the Camera POD (lookfrom, px_du, px_dv,
    vp_upper_left, px00_loc, disk_u_r, disk_v_r, defocus_angle, W, H, spp, depth).  Returns (linear [H][W][3], bytes, segments)."""
    lin, q8, segs = [], [], 0
    lf = cam["lookfrom"]
    for hn in range(cam["H"]):
        row, rowq = [], []
        for wn in range(cam["W"]):
            pix = hn * cam["W"] + wn
            if cam["spp"] > 0:
                px = (0.0, 0.0, 0.0)
                for s in range(cam["spp"]):
                    ox, oy = draw(seed, pix, s, 0, 0) - 0.5, draw(seed, pix, s, 0, 1) - 0.5
                    ps = _add(_add(cam["px00_loc"], _mul(cam["px_du"], float(wn) + ox)), _mul(cam["px_dv"], float(hn) + oy))
                    orig = lf
                    if not cam["defocus_angle"] <= 0.0:  # camera.rs:220-224, point3.rs:106-110
                        r = math.sqrt(draw(seed, pix, s, 0, 2))
                        th = draw(seed, pix, s, 0, 3) * 6.283185307179586
                        orig = _add(_add(lf, _mul(cam["disk_u_r"], r * math.cos(th))), _mul(cam["disk_v_r"], r * math.sin(th)))
                    c, n = color(spheres, orig, _sub(ps, orig), draw(seed, pix, s, 0, 4), cam["depth"], seed, pix, s, 0)
                    segs += n
                    px = _add(px, c)
                px = _mul(px, 1.0 / float(cam["spp"]))
            else:  # camera.rs:247-254: pixel corner, lookfrom, time 0
                pc = _add(_add(cam["vp_upper_left"], _mul(cam["px_du"], float(wn))), _mul(cam["px_dv"], float(hn)))
                px, n = color(spheres, lf, _sub(pc, lf), 0.0, cam["depth"], seed, pix, 0, 0)
                segs += n
            row.append(px)
            rowq.append(tuple(quantize(x) for x in px))
        lin.append(row)
        q8.append(rowq)
    return lin, q8, segs
