"""np_kernels.py -- TEST INFRASTRUCTURE.  Second, independent restatement (vectorised numpy
float32) of three pieces of the reference's device code, used only to pin the C oracle:

    xorshift_random        trace.metal:87-93
    generate_camera_rays   trace.metal:1020-1067   (K2)
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
