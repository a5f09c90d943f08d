"""pyoracle.py -- second, independent CPU restatement of the OpenFDCM hot path (numpy float32).

TEST INFRASTRUCTURE ONLY (same rules as oracle/fdcm_oracle.cpp).  The reference cannot be built in
this image, so nothing can be checked against the reference binary; instead two restatements were
written separately from the cited reference lines -- the C++ one (threaded, used as the CPU
baseline) and this slow numpy one -- and tests/test_oracle_cross.py requires them to agree bit for
bit on random inputs.  This file also generates the golden fixtures (tests/golden/make_golden.py).

libm calls (atanf, cosf, sinf) go to the C library through ctypes so that they are the same
functions the C++ code uses, not numpy's own SIMD implementations.
Paths cited below are relative to /root/reference.
"""
import ctypes
import math

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL("libm.so.6")
for _n in ("atanf", "cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
FLT_MAX = f32(np.finfo(np.float32).max)
PIF = f32(3.14159265358979323846)
PI2F = f32(1.57079632679489661923)
L2, L2_SQUARED, L1 = 0, 1, 2


def atanf(x): return f32(_libm.atanf(float(x)))
def cosf(x): return f32(_libm.cosf(float(x)))
def sinf(x): return f32(_libm.sinf(float(x)))


def _div(a, b):
    with np.errstate(all="ignore"):
        return f32(a) / f32(b)


# ---------------------------------------------------------------- drawing.h
def rasterize_vector(vx, vy):
    """drawing.h:57-67."""
    t = _div(vy, vx)
    if t >= -1.0 and t < 1:
        c = 1 if vx < 0 else 0
        return f32(1 - 2 * c), f32(float(t) - 2.0 * c * float(t))
    c = 1 if vy < 0 else 0
    inv = _div(f32(1), t)
    return f32(float(inv) - 2.0 * c * float(inv)), f32(1 - 2 * c)


def lin_spaced(n, lo, hi):
    """Eigen 3.4.0 LinSpaced<float> (NullaryFunctors.h linspaced_op_impl, scalar path)."""
    lo, hi = f32(lo), f32(hi)
    if n == 1:
        lo = hi
    size1 = 1 if n == 1 else n - 1
    step = f32(0) if n == 1 else f32((hi - lo) / f32(n - 1))
    flip = abs(hi) < abs(lo)
    out = np.zeros(n, dtype=np.float32)
    for i in range(n):
        if flip:
            out[i] = lo if i == 0 else f32(hi - f32(f32(size1 - i) * step))
        else:
            out[i] = hi if i == size1 else f32(lo + f32(f32(i) * step))
    return out


def _round_half_away(a):
    a = np.asarray(a, dtype=np.float32)
    return (np.sign(a) * np.floor(np.abs(a) + f32(0.5))).astype(np.int64)


def rasterize_line(l):
    """drawing.h:74-102 -> (xs, ys) int64."""
    p1x, p1y, p2x, p2y = (f32(v) for v in l)
    if abs(f32(p2x - p1x)) <= f32(1e-5) and abs(f32(p2y - p1y)) <= f32(1e-5):
        return _round_half_away([p1x]), _round_half_away([p1y])
    vx, vy = f32(p2x - p1x), f32(p2y - p1y)
    rx, ry = rasterize_vector(vx, vy)
    eps = float(np.finfo(np.float32).eps)
    req = lambda a: abs(float(a)) <= eps + 1e-10 * abs(float(a))
    if req(rx):
        n = int(_div(vy, ry)) + 1
        xs, ys = np.full(n, p1x, dtype=np.float32), lin_spaced(n, p1y, p2y)
    elif req(ry):
        n = int(_div(vx, rx)) + 1
        xs, ys = lin_spaced(n, p1x, p2x), np.full(n, p1y, dtype=np.float32)
    else:
        n = int(max(_div(vx, rx), _div(vy, ry))) + 1
        xs, ys = lin_spaced(n, p1x, p2x), lin_spaced(n, p1y, p2y)
    return _round_half_away(xs), _round_half_away(ys)


def clip_line(l, xmax, ymax):
    """drawing.cpp:29-112 with the box [0,xmax]x[0,ymax]; returns None when purged."""
    p = [f32(v) for v in l]
    xmax, ymax = f32(xmax), f32(ymax)

    def code(x, y):
        c = 0
        if x < 0: c |= 1
        elif x > xmax: c |= 2
        if y < 0: c |= 4
        elif y > ymax: c |= 8
        return c

    def clip_y(a, b, yc):
        p[a] = f32(p[a] + f32(f32(f32(p[b] - p[a]) * f32(yc - p[a + 1])) / f32(p[b + 1] - p[a + 1])))
        p[a + 1] = f32(yc)

    def clip_x(a, b, xc):
        p[a + 1] = f32(p[a + 1] + f32(f32(f32(p[b + 1] - p[a + 1]) * f32(xc - p[a])) / f32(p[b] - p[a])))
        p[a] = f32(xc)

    c1, c2 = code(p[0], p[1]), code(p[2], p[3])
    for _ in range(1000):
        if c1 == 0 and c2 == 0:
            return p
        if c1 & c2:
            return None
        with np.errstate(all="ignore"):
            if c1:
                if c1 & 8: clip_y(0, 2, ymax)
                elif c1 & 4: clip_y(0, 2, 0)
                elif c1 & 2: clip_x(0, 2, xmax)
                else: clip_x(0, 2, 0)
                c1 = code(p[0], p[1])
                continue
            if c2 & 8: clip_y(2, 0, ymax)
            elif c2 & 4: clip_y(2, 0, 0)
            elif c2 & 2: clip_x(2, 0, xmax)
            else: clip_x(2, 0, 0)
            c2 = code(p[2], p[3])
    return None


# ---------------------------------------------------------------- imgproc.h
def column_pass_l2(img):
    """_distanceTransformColumnPassL2, imgproc.h:91-130, on img[rows, cols] (each column in place)."""
    R, Ccols = img.shape
    sq = [f32(i * i) for i in range(R)]
    for i in range(Ccols):
        f = img[:, i]
        v = [0] * R
        z = [f32(0)] * (R + 1)
        k = 0
        z[0], z[1] = f32(-np.inf), f32(np.inf)
        for q in range(1, R):
            while True:
                vk = v[k]
                with np.errstate(all="ignore"):
                    s = f32(f32(f32(f32(f[q] + sq[q]) - f[vk]) - sq[vk]) / f32(2 * q - 2 * vk))
                if s > z[k]:
                    k += 1
                    v[k] = q
                    z[k] = s
                    z[k + 1] = f32(np.inf)
                    break
                k -= 1
        k = 0
        for q in range(R):
            while z[k + 1] < f32(q):
                k += 1
            vk = v[k]
            f[q] = f32(f[vk] + sq[abs(q - vk)])


def column_pass_l1(img):
    """imgproc.h:137-146 on img[rows, cols]: sweeps across columns."""
    for q in range(1, img.shape[1]):
        img[:, q] = np.minimum(img[:, q], img[:, q - 1] + f32(1))
    for q in range(img.shape[1] - 2, -1, -1):
        img[:, q] = np.minimum(img[:, q], img[:, q + 1] + f32(1))


def distance_transform(lines, W, H, dist):
    """imgproc.h:169-194; lines = list of 4-float lines; returns img[H, W] float32."""
    img = np.full((H, W), FLT_MAX, dtype=np.float32)
    for l in lines:
        c = clip_line(l, W - 1, H - 1)
        if c is None:
            continue
        xs, ys = rasterize_line(c)
        img[ys, xs] = 0
    if dist == L1:
        column_pass_l1(img)
        t = np.ascontiguousarray(img.T)
        column_pass_l1(t)
        return np.ascontiguousarray(t.T)
    column_pass_l2(img)
    t = np.ascontiguousarray(img.T)
    column_pass_l2(t)
    out = np.ascontiguousarray(t.T)
    return np.sqrt(out) if dist == L2 else out


def line_integral(img, angle):
    """imgproc.h:38-84 in place on img[H, W]."""
    rx, ry = rasterize_vector(cosf(angle), sinf(angle))
    R, C = img.shape
    p0x = C - 1 if rx < 0 else 0
    p0y = R - 1 if ry < 0 else 0
    rnd = lambda v: int(math.copysign(math.floor(abs(float(v)) + 0.5), float(v)))
    if abs(rx) == 1:
        prev = p0x
        for i in range(1, C):
            x = p0x + i * int(rx)
            dy = rnd(f32(f32(i) * ry)) - rnd(f32(f32(i - 1) * ry))
            y1, y2, n = max(dy, 0), max(-dy, 0), R - abs(dy)
            img[y1:y1 + n, x] += img[y2:y2 + n, prev]
            prev = x
    elif abs(ry) == 1:
        prev = p0y
        for i in range(1, R):
            dx = rnd(f32(f32(i) * rx)) - rnd(f32(f32(i - 1) * rx))
            y = p0y + i * int(ry)
            x1, x2, n = max(dx, 0), max(-dx, 0), C - abs(dx)
            img[y, x1:x1 + n] += img[prev, x2:x2 + n]
            prev = y


# ---------------------------------------------------------------- dt3cpu.h / dt3cpu.cpp
def closest_orientation(keys, l):
    """dt3cpu.h:93-114."""
    with np.errstate(all="ignore"):
        ang = atanf(f32(f32(l[3]) - f32(l[1])) / f32(f32(l[2]) - f32(l[0])))
    it = int(np.searchsorted(keys, ang, side="right")) if not np.isnan(ang) else len(keys)
    if it != len(keys) and it != 0:
        up, lo = abs(f32(ang - keys[it])), abs(f32(ang - keys[it - 1]))
        return it - 1 if lo < up else it
    it = len(keys) - 1
    a1, a2 = f32(ang - keys[0]), f32(ang - keys[it])
    m1 = min(a1, abs(f32(a1 - PIF))) if not np.isnan(a1) else a1
    m2 = min(a2, abs(f32(a2 - PIF))) if not np.isnan(a2) else a2
    return 0 if m1 < m2 else it


def build(scene, depth=30, coeff=5.0, padding=2.2, dist=L2, stop_after=3):
    """buildCpuFeaturemap, dt3cpu.h:174-234.  scene (4, N).  Returns dict(keys, vol[k][x][y], t, W, H)."""
    scene = np.asarray(scene, dtype=np.float32)
    if scene.shape[1] == 0:
        return dict(keys=np.zeros(0, np.float32), vol=np.zeros((0, 0, 0), np.float32), t=np.zeros(2, np.float32), W=0, H=0)
    xs = np.concatenate([scene[0], scene[2]]); ys = np.concatenate([scene[1], scene[3]])
    mn = np.array([xs.min(), ys.min()], np.float32); mx = np.array([xs.max(), ys.max()], np.float32)
    d = mx - mn
    req = f32(f32(max(f32(1), f32(padding))) * max(d[0], d[1])) * f32(1)
    t = np.array([f32(req / f32(2)) - f32(f32(mx[0] + mn[0]) / f32(2)), f32(req / f32(2)) - f32(f32(mx[1] + mn[1]) / f32(2))], np.float32)
    S = int(math.ceil(float(f32(req + f32(1)))))
    keys = np.unique(np.array([f32(f32(f32(i) * PIF) / f32(depth)) - PI2F for i in range(depth)], np.float32))
    m = len(keys)
    tl = scene.copy()
    tl[0] += t[0]; tl[2] += t[0]; tl[1] += t[1]; tl[3] += t[1]
    cls = [[] for _ in range(m)]
    for i in range(tl.shape[1]):
        cls[closest_orientation(keys, tl[:, i])].append(tl[:, i])
    imgs = [distance_transform(cls[k], S, S, dist) for k in range(m)]
    if stop_after >= 2:  # propagateOrientation, dt3cpu.cpp:77-107
        fwd, bwd = int(math.ceil(1.5 * m)), -int(math.floor(1.5 * m))
        def prop(start, end, step):
            c = start
            while c != end:
                c1 = (m + int(math.fmod(c - step, m))) % m
                c2 = (m + int(math.fmod(c, m))) % m
                h = abs(f32(keys[c1] - keys[c2]))
                w = f32(f32(coeff) * min(h, abs(f32(h - PIF))))
                imgs[c2] = np.minimum(imgs[c2], imgs[c1] + w)
                c += step
        prop(0, fwd, 1)
        prop(m, bwd, -1)
    if stop_after >= 3:
        for k in range(m):
            with np.errstate(over="ignore"):
                line_integral(imgs[k], keys[k])
    vol = np.stack([np.ascontiguousarray(im.T) for im in imgs]).astype(np.float32)
    return dict(keys=keys, vol=vol, t=t, W=S, H=S)


# ---------------------------------------------------------------- search
def eigen_sum(v):
    """VectorXf::sum(), Eigen 3.4.0 Redux.h (Packet4f, aligned data)."""
    v = np.asarray(v, dtype=np.float32)
    n = len(v)
    if n == 0:
        return f32(0)
    a2, a1 = (n // 8) * 8, (n // 4) * 4
    if a1:
        p0 = v[0:4].copy()
        if a1 > 4:
            p1 = v[4:8].copy()
            for i in range(8, a2, 8):
                p0 = p0 + v[i:i + 4]
                p1 = p1 + v[i + 4:i + 8]
            p0 = p0 + p1
            if a1 > a2:
                p0 = p0 + v[a2:a2 + 4]
        res = f32(f32(p0[0] + p0[2]) + f32(p0[1] + p0[3]))
        for i in range(a1, n):
            res = f32(res + v[i])
        return res
    res = v[0]
    for i in range(1, n):
        res = f32(res + v[i])
    return res


def transform(lines, T):
    """math.h:341-344 on (4, N) lines with T (2,3)."""
    out = np.zeros_like(lines)
    for r in (0, 2):
        x, y = lines[r], lines[r + 1]
        out[r] = (T[0, 0] * x + T[0, 1] * y) + T[0, 2]
        out[r + 1] = (T[1, 0] * x + T[1, 1] * y) + T[1, 2]
    return out


def align(tl, rl):
    """math.h:387-406."""
    def norm(l):
        dx, dy = f32(l[2] - l[0]), f32(l[3] - l[1])
        with np.errstate(all="ignore"):
            n = np.sqrt(f32(f32(dx * dx) + f32(dy * dy)))
            return f32(dx / n), f32(dy / n)
    tdx, tdy = norm(tl)
    adx, ady = norm(rl)
    c = f32(f32(adx * tdx) + f32(ady * tdy))
    s = f32(f32(ady * tdx) - f32(adx * tdy))
    rc = (f32(f32(rl[2] + rl[0]) / f32(2)), f32(f32(rl[3] + rl[1]) / f32(2)))
    res = []
    for (r00, r01, r10, r11) in ((c, f32(-s), s, c), (f32(-c), s, f32(-s), f32(-c))):
        x1, y1 = f32(f32(r00 * tl[0]) + f32(r01 * tl[1])), f32(f32(r10 * tl[0]) + f32(r11 * tl[1]))
        x2, y2 = f32(f32(r00 * tl[2]) + f32(r01 * tl[3])), f32(f32(r10 * tl[2]) + f32(r11 * tl[3]))
        res.append(np.array([[r00, r01, f32(rc[0] - f32(f32(x2 + x1) / f32(2)))],
                             [r10, r11, f32(rc[1] - f32(f32(y2 + y1) / f32(2)))]], np.float32))
    return res


def minmax_translation(tmpl, av, W, H, extra):
    """dt3cpu.cpp:30-75."""
    inf = f32(np.inf)
    if abs(av[0]) <= f32(1e-5) and abs(av[1]) <= f32(1e-5):
        return inf, inf
    size = (f32(W), f32(H))
    mn = (f32(min(tmpl[0].min(), tmpl[2].min()) + extra[0]), f32(min(tmpl[1].min(), tmpl[3].min()) + extra[1]))
    mx = (f32(max(tmpl[0].max(), tmpl[2].max()) + extra[0]), f32(max(tmpl[1].max(), tmpl[3].max()) + extra[1]))
    nan = f32(np.nan)
    if any(f32(f32(size[r] - f32(1)) - mx[r]) < 0 for r in (0, 1)) or any(mn[r] < 0 for r in (0, 1)):
        return nan, nan
    ext = [[None, None], [None, None]]
    for r in (0, 1):
        mult = [f32(-mx[r]), f32(-mn[r]), f32(f32(size[r] - mx[r]) - f32(1)), f32(f32(size[r] - mn[r]) - f32(1))]
        with np.errstate(all="ignore"):
            q = [f32(v / f32(av[r])) for v in mult]
        pos = [inf if np.signbit(v) else v for v in q]
        neg = [v if np.signbit(v) else f32(-inf) for v in q]
        ext[0][r] = nan if any(np.isnan(v) for v in neg) else max(neg)
        ext[1][r] = nan if any(np.isnan(v) for v in pos) else min(pos)
    fin = np.isfinite
    if all(fin(ext[a][b]) for a in (0, 1) for b in (0, 1)):
        return max(ext[0][0], ext[0][1]), min(ext[1][0], ext[1][1])
    if fin(ext[0][0]) and fin(ext[1][0]):
        return ext[0][0], ext[1][0]
    return ext[0][1], ext[1][1]


def evaluate(fm, tmpl, bins, tr):
    """dt3cpu.cpp:153-175 for one translation."""
    ox, oy = f32(fm["t"][0] + tr[0]), f32(fm["t"][1] + tr[1])
    vals = np.zeros(tmpl.shape[1], np.float32)
    for i in range(tmpl.shape[1]):
        x1, y1 = int(f32(tmpl[0, i] + ox)), int(f32(tmpl[1, i] + oy))
        x2, y2 = int(f32(tmpl[2, i] + ox)), int(f32(tmpl[3, i] + oy))
        v = fm["vol"][bins[i]]
        vals[i] = abs(f32(v[x1, y1] - v[x2, y2]))
    return eigen_sum(vals)


def optimize(fm, tmpl, av, kind, B):
    """batchoptimize.cpp:15-99 (kind 1) / defaultoptimize.cpp:13-66 (kind 0)."""
    eps = float(np.finfo(np.float32).eps)
    ssum = f32(abs(av[0]) + abs(av[1]))
    if abs(float(ssum)) <= eps + 1e-10 * abs(float(ssum)):
        return None
    sx, sy = rasterize_vector(av[0], av[1])
    lo, hi = minmax_translation(tmpl, (sx, sy), fm["W"], fm["H"], fm["t"])
    if not (np.isfinite(lo) and np.isfinite(hi)):
        return None
    bins = [closest_orientation(fm["keys"], tmpl[:, i]) for i in range(tmpl.shape[1])]
    scores = [evaluate(fm, tmpl, bins, (f32(0), f32(0)))]
    trs = [(f32(0), f32(0))]
    if kind == 0:
        B = 1
    for d in (1, -1):
        lim = int(hi) if d > 0 else int(lo)
        k0 = d
        while (k0 <= lim) if d > 0 else (k0 >= lim):
            ks = [k for k in range(k0, k0 + d * B, d) if ((k <= lim) if d > 0 else (k >= lim))]
            bt = [(f32(f32(k) * sx), f32(f32(k) * sy)) for k in ks]
            bs = [evaluate(fm, tmpl, bins, t) for t in bt]
            a = int(np.argmin(np.array(bs, np.float32)))
            if bs[a] > scores[-1]:
                break
            trs.append(bt[a]); scores.append(bs[a])
            if kind == 1 and bs[a] < bs[-1]:
                break
            k0 += d * B
    b = int(np.argmin(np.array(scores, np.float32)))
    return scores[b], trs[b]


def search(fm, templates, scene, maxT, maxS, kind=1, B=10):
    """search<DefaultMatch>, defaultmatch.cpp:32-89 with DefaultSearch (defaultsearch.cpp:29-49).
    Ties in the length sort follow numpy's stable sort; fixtures avoid equal lengths."""
    scene = np.asarray(scene, np.float32)
    out = []
    if not templates or scene.shape[1] == 0 or fm["W"] == 0:
        return out
    def lengths(l):
        dx, dy = l[2] - l[0], l[3] - l[1]
        return np.sqrt(dx * dx + dy * dy).astype(np.float32)
    sl = lengths(scene)
    ssi = np.argsort(-sl, kind="stable")
    ssl = sl[ssi]
    for ti, tm in enumerate(templates):
        tm = np.asarray(tm, np.float32)
        if tm.shape[1] == 0:
            continue
        tlen = lengths(tm)
        sti = np.argsort(-tlen, kind="stable")
        for j in range(min(tm.shape[1], maxT)):
            val = tlen[sti[j]]
            it = int(np.searchsorted(-ssl, -val, side="left"))  # lower_bound with greater
            if it == 0: c = 0
            elif it == len(ssl): c = it - 1
            else: c = it if abs(f32(val - ssl[it])) < abs(f32(val - ssl[it - 1])) else it - 1
            b = max(0, c - maxS // 2); e = min(b + maxS, len(ssl)); b = max(0, e - maxS)
            for i in range(b, e):
                sline = scene[:, ssi[i]]
                dx, dy = f32(sline[2] - sline[0]), f32(sline[3] - sline[1])
                with np.errstate(all="ignore"):
                    n = np.sqrt(f32(f32(dx * dx) + f32(dy * dy)))
                    av = (f32(dx / n), f32(dy / n))
                for T in align(tm[:, sti[j]], sline):
                    r = optimize(fm, transform(tm, T), av, kind, B)
                    if r is not None:
                        Tm = T.copy()
                        Tm[0, 2] = f32(T[0, 2] + r[1][0]); Tm[1, 2] = f32(T[1, 2] + r[1][1])
                        out.append((ti, r[0], Tm))
    return out
