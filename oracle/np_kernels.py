"""np_kernels.py -- TEST INFRASTRUCTURE.  Second, independent restatement (vectorised numpy
float32) of three pieces of the reference's device code, used only to pin the C oracle:

    xorshift_random        trace.metal:87-93
    generate_camera_rays   trace.metal:1020-1067   (K2)
    generate_light_rays    trace.metal:1070-1124   (K1; with numpy statements of detmath sin/cos)
    orthonormal, random_hemisphere_cosine / _uniform   trace.metal:200-224
    traverse_bvh           trace.metal:106-176     (closest hit, per-ray stack)

Every operation is an IEEE binary32 numpy ufunc applied in the reference's order (numpy's
float32 +,-,*,/ and sqrt are correctly rounded), so agreement with bdpt_oracle.c is expected to be
exact, which tests/test_oracle_pinning.py asserts.  Two restatements written separately from the
same Metal source agreeing bit for bit is the strongest pin available: the reference ships no
test vectors.
"""
import numpy as np

f32 = np.float32
DELTA = f32(0.0001)


def xorshift(state):
    """One xorshift32 step on a uint32 array; returns (new_state, float in [0,1])."""
    s = state.astype(np.uint32).copy()
    s ^= (s << np.uint32(13))
    s ^= (s >> np.uint32(17))
    s ^= (s << np.uint32(5))
    return s, s.astype(f32) / f32(4294967296.0)     # (float)0xFFFFFFFF rounds to 2^32


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _normalize(a):
    inv = f32(1.0) / np.sqrt(_dot(a, a))
    return a * inv[..., None]


def _min(x, y):   # MSL: y < x ? y : x
    return np.where(y < x, y, x)


def _max(x, y):   # MSL: x < y ? y : x
    return np.where(x < y, y, x)


def generate_camera_rays(camera, seeds):
    """K2.  camera: 1-element Camera record; seeds: (B,2) uint32.  Returns (origin, direction,
    c_importance, new_seeds)."""
    c = camera.reshape(-1)[0]
    W, H = int(c["pixel_width"]), int(c["pixel_height"])
    B = W * H
    s0, xo = xorshift(seeds[:, 0])
    s1, yo = xorshift(seeds[:, 1])
    idx = np.arange(B)
    px, py = (idx % W).astype(f32), (idx // W).astype(f32)
    Wf, Hf = f32(W), f32(H)
    xn = ((px + xo) - f32(0.5) * Wf) / Wf
    yn = ((py + yo) - f32(0.5) * Hf) / Hf
    dx, dy = c["dx"][:3].astype(f32), c["dy"][:3].astype(f32)
    xv = (xn[:, None] * dx[None, :]) * f32(c["phys_width"])
    yv = (yn[:, None] * dy[None, :]) * f32(c["phys_height"])
    origin = (c["center"][:3].astype(f32)[None, :] + xv) + yv
    direction = _normalize(c["focal_point"][:3].astype(f32)[None, :] - origin)
    c_imp = f32(1.0) / (f32(c["phys_width"]) * f32(c["phys_height"]))
    return origin.astype(f32), direction.astype(f32), c_imp, np.stack([s0, s1], axis=1)


def traverse(origin, direction, boxes, triangles):
    """Closest hit per ray with the reference's per-ray stack walk.  Returns best_i, best_t, u, v."""
    n = len(origin)
    o, d = origin.astype(f32), direction.astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = f32(1.0) / d
        bmin, bmax = boxes["min"][:, :3].astype(f32), boxes["max"][:, :3].astype(f32)
        left, right = boxes["left"].astype(np.int64), boxes["right"].astype(np.int64)
        v0 = triangles["v0"][:, :3].astype(f32)
        e1 = triangles["v1"][:, :3].astype(f32) - v0
        e2 = triangles["v2"][:, :3].astype(f32) - v0
        best_i = np.full(n, -1, np.int32)
        best_t = np.full(n, np.inf, f32)
        bu, bv = np.zeros(n, f32), np.zeros(n, f32)
        stack = np.zeros((n, 64), np.int64)
        sp = np.ones(n, np.int64)            # stack[0] = root
        while True:
            act = np.flatnonzero((sp > 0) & (sp < 64))
            if len(act) == 0:
                break
            sp[act] -= 1
            node = stack[act, sp[act]]
            t0 = (bmin[node] - o[act]) * inv[act]
            t1 = (bmax[node] - o[act]) * inv[act]
            tmn, tmx = _min(t0, t1), _max(t0, t1)
            tmin = _max(_max(tmn[:, 0], tmn[:, 1]), _max(tmn[:, 2], f32(0.0)))
            tmax = _min(_min(tmx[:, 0], tmx[:, 1]), _min(tmx[:, 2], f32(np.inf)))
            go = (tmin <= tmax) & (tmin < best_t[act])
            inner = go & (right[node] == 0)
            ia = act[inner]
            stack[ia, sp[ia]] = left[node[inner]]
            stack[ia, sp[ia] + 1] = left[node[inner]] + 1
            sp[ia] += 2
            leaf = go & (right[node] != 0)
            la, ln = act[leaf], node[leaf]
            if len(la):
                for k in range(int((right[ln] - left[ln]).max())):
                    tri = left[ln] + k
                    m = tri < right[ln]
                    r, ti = la[m], tri[m]
                    h = _cross(d[r], e2[ti])
                    a = _dot(e1[ti], h)
                    f = f32(1.0) / a
                    s = o[r] - v0[ti]
                    u = f * _dot(s, h)
                    ok = ~((u < 0) | (u > 1))
                    q = _cross(s, e1[ti])
                    v = f * _dot(d[r], q)
                    ok &= ~((v < 0) | (u + v > 1))
                    t = f * _dot(e2[ti], q)
                    ok &= (t > DELTA) & (t < best_t[r])
                    rr = r[ok]
                    best_i[rr] = ti[ok]
                    best_t[rr] = t[ok]
                    bu[rr] = u[ok]
                    bv[rr] = v[ok]
    return best_i, best_t, bu, bv


# ---- detmath (oracle/detmath.h) restated with numpy float32 ufuncs, same operation order ----
_FOPI, _DP1, _DP2, _DP3 = f32(1.27323954473516), f32(0.78515625), f32(2.4187564849853515625e-4), f32(3.77489497744594108e-8)
_PIO2 = f32(1.5707963267948966192)


def _sin_poly(x, z):
    y = ((f32(-1.9515295891E-4) * z + f32(8.3321608736E-3)) * z - f32(1.6666654611E-1)) * z * x
    return y + x


def _cos_poly(z):
    y = ((f32(2.443315711809948E-005) * z - f32(1.388731625493765E-003)) * z + f32(4.166664568298827E-002)) * z * z
    y = y - f32(0.5) * z
    return y + f32(1.0)


def det_sincos(xx):
    """(sin, cos) of float32 arrays in [0, 8192]: octant reduction + minimax polynomials."""
    xx = np.asarray(xx, dtype=f32)
    x = np.abs(xx)
    j = (_FOPI * x).astype(np.int32)
    y = j.astype(f32)
    odd = (j & 1) == 1
    j = np.where(odd, j + 1, j)
    y = np.where(odd, y + f32(1.0), y).astype(f32)
    j = j & 7
    hi = j > 3
    sneg = (xx < 0) ^ hi
    cneg = hi.copy()
    j = np.where(hi, j - 4, j)
    cneg ^= (j > 1)
    x = ((x - y * _DP1) - y * _DP2) - y * _DP3
    z = x * x
    ps, pc = _sin_poly(x, z), _cos_poly(z)
    swap = (j == 1) | (j == 2)
    s = np.where(swap, pc, ps)
    c = np.where(swap, ps, pc)
    return np.where(sneg, -s, s).astype(f32), np.where(cneg, -c, c).astype(f32)


def det_asin(xx):
    xx = np.asarray(xx, dtype=f32)
    a = np.abs(xx)
    big = a > f32(0.5)
    zb = f32(0.5) * (f32(1.0) - a)
    x = np.where(big, np.sqrt(zb), a).astype(f32)
    z = np.where(big, zb, x * x).astype(f32)
    p = ((((f32(4.2163199048E-2) * z + f32(2.4181311049E-2)) * z + f32(4.5470025998E-2)) * z + f32(7.4953002686E-2)) * z
         + f32(1.6666752422E-1)) * z * x + x
    p = np.where(big, _PIO2 - (p + p), p).astype(f32)
    p = np.where(a < f32(1.0e-4), a, p)
    return np.where(xx < 0, -p, p).astype(f32)


def det_acos(x):
    x = np.asarray(x, dtype=f32)
    hi = x > f32(0.5)
    lo = x < f32(-0.5)
    r_hi = f32(2.0) * det_asin(np.sqrt(f32(0.5) * (f32(1.0) - np.where(hi, x, f32(1.0)))))
    r_lo = f32(3.14159265358979323846) - f32(2.0) * det_asin(np.sqrt(f32(0.5) * (f32(1.0) + np.where(lo, x, f32(-1.0)))))
    r_mid = _PIO2 - det_asin(np.where(hi | lo, f32(0.0), x))
    return np.where(hi, r_hi, np.where(lo, r_lo, r_mid)).astype(f32)


PI = f32(3.14159265359)


def orthonormal(n):
    ax, ay, az = np.abs(n[:, 0]), np.abs(n[:, 1]), np.abs(n[:, 2])
    pick_x = (ax <= ay) & (ax <= az)
    pick_y = ~pick_x & (ay <= az)
    v = np.zeros_like(n)
    v[pick_x, 0] = 1
    v[pick_y, 1] = 1
    v[~pick_x & ~pick_y, 2] = 1
    x = _normalize(v - _dot(v, n)[:, None] * n)
    y = _normalize(_cross(n, x))
    return x, y


def random_hemisphere_uniform(xa, ya, za, rx, ry):
    z = rx
    r = np.sqrt(_max(f32(0.0), f32(1.0) - z * z))
    phi = (f32(2) * PI) * ry
    sp, cp = det_sincos(phi)
    return _normalize(((r * cp)[:, None] * xa + (r * sp)[:, None] * ya) + z[:, None] * za)


def random_hemisphere_cosine(xa, ya, za, rx, ry):
    theta = det_acos(np.sqrt(rx))
    phi = (f32(2) * PI) * ry
    st, ct = det_sincos(theta)
    sp, cp = det_sincos(phi)
    return _normalize(((st * cp)[:, None] * xa + (st * sp)[:, None] * ya) + ct[:, None] * za)


def generate_light_rays(light_triangles, areas, seeds):
    """K1.  Returns (origin, direction, l_importance, light_index, new_seeds)."""
    count = len(light_triangles)
    s0, s1 = seeds[:, 0].copy(), seeds[:, 1].copy()
    s0, r = xorshift(s0)
    li = np.minimum((r * f32(count)).astype(np.int32), count - 1)
    s0, u = xorshift(s0)
    s1, v = xorshift(s1)
    flip = (u + v) > f32(1.0)
    u = np.where(flip, f32(1.0) - u, u).astype(f32)
    v = np.where(flip, f32(1.0) - v, v).astype(f32)
    w = f32(1.0) - u - v
    T = light_triangles[li]
    n = T["normal"][:, :3].astype(f32)
    origin = ((T["v0"][:, :3] * u[:, None] + T["v1"][:, :3] * v[:, None]) + T["v2"][:, :3] * w[:, None]) + DELTA * n
    x, y = orthonormal(n)
    s0, rx = xorshift(s0)
    s1, ry = xorshift(s1)
    d = random_hemisphere_uniform(x, y, n, rx, ry)
    l_imp = f32(1.0) / (f32(count) * areas[li].astype(f32))
    return origin.astype(f32), d.astype(f32), l_imp.astype(f32), li, np.stack([s0, s1], axis=1)
