"""py_kernels.py -- TEST INFRASTRUCTURE.  A second restatement of the two megakernels of the reference,

    generate_paths   src/trace.metal:381-532   (K3: the path loop, importance hand-over, the three colour cases)
    connect_paths    src/trace.metal:620-869   (K5: strategy loop, p-ratio chain, specular zeroing, t == 1 projection, filter weights)

and of the three small kernels behind them -- adaptive_finalize_samples (K6, :981-1018), the light_sort network with its
driver loop (K7, :872-934 + renderer.py:212-231; not a stable sort, so the order it leaves inside a pixel's run decides float
sums) and light_image_gather (K8, :937-964) --

written from the Metal text statement by statement -- NOT from oracle/bdpt_oracle.c -- as scalar Python over numpy
float32 scalars and numpy records in the reference's own Ray / Path / WeightAggregator layouts.  Purpose: the C oracle, the
HIP kernels and oracle/np_kernels.py were written by one reader of trace.metal; np_kernels.py restates the leaf routines
(K1, K2, traverse_bvh, samplers, detmath) but not the LOOP BOOKKEEPING of K3 and K5 -- which record holds which pdf when,
what `new_ray = next_ray` carries over, which ray of the t == 1 strategy is overwritten and what it keeps.  Here that
bookkeeping is restated a second time with the Ray records copied around exactly as the Metal code copies them, and
tests/test_oracle_pinning.py requires the C oracle to agree with it bit for bit on small frames (pure-Python loops:
seconds at 16x16).  A misreading common to both would still pass; two independent ones would not.

Pinned reading of what the Metal source leaves open (the same decisions DESIGN.md 2.1 / 2.2 lists, restated, not imported):
thread-local records start zeroed (Q3 / Q4 / Q5); dot = (x*x + y*y) + z*z; normalize(v) = v * (1 / sqrt(dot(v, v)));
min / max in the MSL wording; unsuffixed literals are float; sin / cos / acos / atan / exp are the deterministic float32
definitions of oracle/detmath.h (restated below and in np_kernels.py); `round` is half away from zero; float3 is 16 bytes
with a zero fourth word.  Closest hits come from np_kernels.traverse (itself pinned against the C oracle), called in batches.

Only tests/ may import this.  Nothing here touches /root/reference at run time.
"""
import numpy as np

from . import np_kernels as npk

f32 = np.float32
PI = f32(3.14159265359)          # trace.metal:4
DELTA = f32(0.0001)              # trace.metal:5
ZERO, ONE, TWO, HALF = f32(0.0), f32(1.0), f32(2.0), f32(0.5)
INF = f32(np.inf)


# ---- float3 algebra on 3-element float32 arrays ----
def V(x, y, z):
    return np.array([x, y, z], dtype=f32)


def v3(field):
    return np.array(field[:3], dtype=f32)


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cross(a, b):
    return V(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def length(a):
    return np.sqrt(dot(a, a))


def normalize(a):
    with np.errstate(divide="ignore", invalid="ignore"):
        return a * (ONE / np.sqrt(dot(a, a)))


def put3(rec, name, v):
    rec[name][:3] = v
    rec[name][3] = 0


# ---- deterministic elementary functions (oracle/detmath.h): scalar restatements ----
def d_sin(x):
    return f32(npk.det_sincos(np.array([x], f32))[0][0])


def d_cos(x):
    return f32(npk.det_sincos(np.array([x], f32))[1][0])


def d_acos(x):
    if not (x >= f32(-1.0) and x <= ONE):
        return f32(np.nan)
    return f32(npk.det_acos(np.array([x], f32))[0])


def d_atan(xx):
    """Cephes atanf: range reduction at tan(3pi/8) and tan(pi/8), degree-4 polynomial in z = x*x."""
    x = f32(xx)
    neg = False
    if x < ZERO:
        neg, x = True, -x
    with np.errstate(divide="ignore", invalid="ignore"):
        if x > f32(2.414213562373095):
            y, x = f32(1.5707963267948966192), -(ONE / x)
        elif x > f32(0.4142135623730950):
            y, x = f32(0.7853981633974483096), (x - ONE) / (x + ONE)
        else:
            y = ZERO
        z = x * x
        y = y + ((((f32(8.05374449538e-2) * z - f32(1.38776856032E-1)) * z + f32(1.99777106478E-1)) * z
                  - f32(3.33329491539E-1)) * z * x + x)
    return -y if neg else y


def d_exp(xx):
    """Cephes expf: x = n ln2 + r (ln2 in two parts), degree-5 polynomial, scale by 2^n; 0 below -87, inf above 88."""
    x = f32(xx)
    if x != x:
        return x
    if x > f32(88.0):
        return INF
    if x < f32(-87.0):
        return ZERO
    fz = np.floor(f32(1.44269504088896341) * x + HALF)
    x = x - fz * f32(0.693359375)
    x = x - fz * f32(-2.12194440e-4)
    n = int(fz)
    z = x * x
    z = (((((f32(1.9875691500E-4) * x + f32(1.3981999507E-3)) * x + f32(8.3334519073E-3)) * x
           + f32(4.1665795894E-2)) * x + f32(1.6666665459E-1)) * x + f32(5.0000001201E-1)) * z + x + ONE
    scale = np.array([(n + 127) << 23], dtype=np.uint32).view(f32)[0]
    return z * scale


def msl_max(x, y):
    return y if x < y else x


def xorshift(seed):
    """trace.metal:87-93 on a Python int (uint32); returns (new seed, float32)."""
    seed ^= (seed << 13) & 0xFFFFFFFF
    seed ^= seed >> 17
    seed ^= (seed << 5) & 0xFFFFFFFF
    return seed, f32(seed) / f32(4294967296.0)       # (float)0xFFFFFFFF rounds to 2^32


# ---- trace.metal:200-379 ----
def orthonormal(n):
    if abs(n[0]) <= abs(n[1]) and abs(n[0]) <= abs(n[2]):
        v = V(1, 0, 0)
    elif abs(n[1]) <= abs(n[2]):
        v = V(0, 1, 0)
    else:
        v = V(0, 0, 1)
    x = normalize(v - dot(v, n) * n)
    y = normalize(cross(n, x))
    return x, y


def random_hemisphere_cosine(xa, ya, za, rx, ry):
    theta = d_acos(np.sqrt(rx))
    phi = (TWO * PI) * ry
    return normalize(((d_sin(theta) * d_cos(phi)) * xa + (d_sin(theta) * d_sin(phi)) * ya) + d_cos(theta) * za)


def GGX_sample(n, rx, ry, alpha):
    x, y = orthonormal(n)
    theta = (TWO * PI) * rx
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = d_atan(alpha * np.sqrt(ry) / np.sqrt(ONE - ry))
    return normalize(((d_sin(phi) * d_cos(theta)) * x + (d_sin(phi) * d_sin(theta)) * y) + d_cos(phi) * n)


def specular_reflection(i, m):
    return normalize((TWO * dot(i, m)) * m - i)


def GGX_transmit(i, m, ni, no):
    ci = dot(i, m)
    eta = ni / no
    with np.errstate(invalid="ignore"):
        ct = np.sqrt(ONE + eta * (ci * ci - ONE))
    return normalize((eta * ci - ct) * m - eta * i)


def transmit_half_direction(i, o, ni, no):
    return normalize(no * o + ni * i)


def degreve_fresnel(i, m, ni, nt):
    ci = abs(dot(i, m))
    eta = ni / nt
    st2 = eta * eta * (ONE - ci * ci)
    if st2 >= ONE:
        return ONE
    ct = np.sqrt(ONE - st2)
    with np.errstate(divide="ignore", invalid="ignore"):
        rpar = (nt * ci - ni * ct) / (nt * ci + ni * ct)
        rper = (ni * ci - nt * ct) / (ni * ci + nt * ct)
    return HALF * (rpar * rpar + rper * rper)


def GGX_G1(v, m, n, alpha):
    mv = dot(m, v)
    sin2 = ONE - mv * mv
    with np.errstate(divide="ignore", invalid="ignore"):
        tan2 = sin2 / (mv * mv)
        return TWO / (ONE + np.sqrt(ONE + alpha * alpha * tan2))


def GGX_G(i, o, m, n, alpha):
    if dot(i, m) * dot(i, n) <= ZERO:
        return ZERO
    if dot(o, m) * dot(o, n) <= ZERO:
        return ZERO
    return GGX_G1(i, m, n, alpha) * GGX_G1(o, m, n, alpha)


def GGX_D(m, n, alpha):
    if alpha == ZERO:
        return ONE
    a2 = alpha * alpha
    c = dot(m, n)
    c2 = c * c
    den = c2 * (a2 - ONE) + ONE
    with np.errstate(divide="ignore", invalid="ignore"):
        return a2 / (PI * den * den)


def reflect_jacobian(m, o):
    with np.errstate(divide="ignore"):
        return ONE / (f32(4.0) * abs(dot(m, o)))


def transmit_jacobian(i, o, ni, no):
    h = transmit_half_direction(i, o, ni, no)
    ci, co = dot(i, h), dot(o, h)
    num = no * no * abs(co)
    den = (ni * ci + no * co) * (ni * ci + no * co)
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den


def GGX_BRDF_reflect(i, o, m, n, ni, no, alpha):
    D, G, F = GGX_D(m, n, alpha), GGX_G(i, o, m, n, alpha), degreve_fresnel(i, m, ni, no)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (D * G * F) / (f32(4.0) * abs(dot(i, m)))


def GGX_BRDF_transmit(i, o, m, n, ni, no, alpha):
    h = transmit_half_direction(i, o, ni, no)
    D, G, F = GGX_D(m, n, alpha), GGX_G(i, o, m, n, alpha), degreve_fresnel(i, m, ni, no)
    im, om, in_, on = dot(i, h), dot(o, h), dot(i, n), dot(o, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        coeff = (im * om) / (in_ * on)
        num = no * no * D * G * (ONE - F)
        den = (ni * im + no * om) * (ni * im + no * om)
        return coeff * num / den


def diffuse_bounce(wi, n, from_camera, rx, ry):
    x, y = orthonormal(n)
    wo = random_hemisphere_cosine(x, y, n, rx, ry)
    f = abs(dot(n, wo)) / PI
    if from_camera:
        c_p, l_p = abs(dot(n, wo)) / PI, abs(dot(n, wi)) / PI
    else:
        c_p, l_p = abs(dot(n, wi)) / PI, abs(dot(n, wo)) / PI
    return wo, f, c_p, l_p


def reflect_bounce(wi, n, m, ni, no, alpha, from_camera):
    wo = specular_reflection(wi, m)
    f = GGX_BRDF_reflect(wi, wo, m, n, ni, no, alpha)
    pf = degreve_fresnel(wi, m, ni, no)
    pm = abs(dot(m, n)) * GGX_D(m, n, alpha)
    if from_camera:
        c_p, l_p = pf * pm * reflect_jacobian(m, wo), pf * pm * reflect_jacobian(m, wi)
    else:
        c_p, l_p = pf * pm * reflect_jacobian(m, wi), pf * pm * reflect_jacobian(m, wo)
    return wo, f, c_p, l_p


def transmit_bounce(wi, n, m, ni, no, alpha, from_camera):
    wo = GGX_transmit(wi, m, ni, no)
    f = GGX_BRDF_transmit(wi, wo, m, n, ni, no, alpha)
    pf = ONE - degreve_fresnel(wi, m, ni, no)
    pm = abs(dot(m, n)) * GGX_D(m, n, alpha)
    if from_camera:
        c_p, l_p = pf * pm * transmit_jacobian(wi, wo, ni, no), pf * pm * transmit_jacobian(wo, wi, no, ni)
    else:
        c_p, l_p = pf * pm * transmit_jacobian(wo, wi, no, ni), pf * pm * transmit_jacobian(wi, wo, ni, no)
    return wo, f, c_p, l_p


def sample_normal(tri, u, v):
    return normalize((v3(tri["n0"]) * (ONE - u - v) + v3(tri["n1"]) * u) + v3(tri["n2"]) * v)


# ---------------------------------------------------------------- K3: generate_paths, trace.metal:381-532
def generate_paths(rays, boxes, triangles, materials, random_buffer, Ray, Path):
    """rays: Ray[B] (first vertices from K1 / K2); random_buffer: (B, 2) uint32, updated in place.
    Returns (out float32[B,4], Path[B])."""
    B = len(rays)
    out = np.zeros((B, 4), f32)
    paths = np.zeros(B, Path)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        st = []
        for id_ in range(B):
            ray = rays[id_].copy()
            new_ray, next_ray = np.zeros((), Ray), np.zeros((), Ray)
            paths[id_]["from_camera"] = ray["from_camera"]
            if int(ray["from_camera"]) == 0:
                new_ray["l_importance"] = ONE / (TWO * PI)
            else:
                new_ray["c_importance"] = ray["c_importance"]
            st.append(dict(ray=ray, new_ray=new_ray, next_ray=next_ray, alive=True,
                           seed0=int(random_buffer[id_, 0]), seed1=int(random_buffer[id_, 1])))
        for i in range(6):
            live = [k for k in range(B) if st[k]["alive"]]
            if not live:
                break
            o = np.array([st[k]["ray"]["origin"][:3] for k in live], f32)
            d = np.array([st[k]["ray"]["direction"][:3] for k in live], f32)
            bi, bt, bu, bv = npk.traverse(o, d, boxes, triangles)
            for j, k in enumerate(live):
                _path_step(st[k], paths[k], i, int(bi[j]), f32(bt[j]), f32(bu[j]), f32(bv[j]), triangles, materials)
        for id_ in range(B):
            p = paths[id_]
            for i in range(int(p["length"])):
                if int(p["rays"][i]["hit_light"]) >= 0:
                    c = v3(p["rays"][i - 1]["color"]) / f32(p["rays"][i]["tot_importance"])
                    out[id_] = (c[0], c[1], c[2], 1.0)
                    break
            random_buffer[id_, 0], random_buffer[id_, 1] = st[id_]["seed0"], st[id_]["seed1"]
    return out, paths


def _path_step(s, path, i, best_i, best_t, u, v, triangles, materials):
    """One iteration of the path loop (trace.metal:407-516) after traverse_bvh; `s` holds ray / new_ray / next_ray."""
    ray, new_ray, next_ray = s["ray"], s["new_ray"], s["next_ray"]
    from_camera = int(path["from_camera"]) != 0
    if best_i == -1:
        s["alive"] = False
        return
    tri = triangles[best_i]
    mat = materials[int(tri["material"])]
    alpha = f32(mat["alpha"])
    rd, tn = v3(ray["direction"]), v3(tri["normal"])
    sn = sample_normal(tri, u, v)
    facing = dot(-rd, tn)
    if facing > ZERO:
        n, ni, no = sn, ONE, f32(mat["ior"])
    elif facing < ZERO:
        n, ni, no = -sn, f32(mat["ior"]), ONE
    else:
        s["alive"] = False
        return
    put3(new_ray, "origin", v3(ray["origin"]) + rd * best_t)
    new_ray["material"] = tri["material"]
    new_ray["triangle"] = best_i
    new_ray["hit_light"] = best_i if (int(tri["is_light"]) != 0 and dot(rd, tn) < ZERO) else -1
    new_ray["hit_camera"] = best_i if int(tri["is_camera"]) != 0 else -1
    wi = -rd
    s["seed0"], rxa = xorshift(s["seed0"])
    s["seed1"], rya = xorshift(s["seed1"])
    s["seed0"], rxb = xorshift(s["seed0"])
    s["seed1"], ryb = xorshift(s["seed1"])
    f, c_p, l_p = ONE, ONE, ONE
    m = GGX_sample(n, rxa, rya, alpha)
    if dot(wi, m) < ZERO or dot(m, n) < ZERO:
        s["alive"] = False
        return
    put3(new_ray, "normal", n)
    fresnel = degreve_fresnel(wi, m, ni, no)
    mtype = int(mat["type"])
    if mtype == 0:
        wo, f, c_p, l_p = diffuse_bounce(wi, n, from_camera, rxb, ryb)
    elif mtype == 1:
        if rxb <= fresnel:
            wo, f, c_p, l_p = reflect_bounce(wi, n, m, ni, no, alpha, from_camera)
        else:
            wo, f, c_p, l_p = transmit_bounce(wi, n, m, ni, no, alpha, from_camera)
    elif mtype == 2:
        if rxb <= fresnel:
            wo, f, c_p, l_p = reflect_bounce(wi, n, m, ni, no, alpha, from_camera)
        else:
            wo, f, c_p, l_p = diffuse_bounce(wi, n, from_camera, rxb, ryb)
    else:
        wo, f, c_p, l_p = reflect_bounce(wi, n, m, ni, no, alpha, from_camera)
    rc, mc = v3(ray["color"]), v3(mat["color"])
    if dot(wi, tn) > ZERO and dot(wo, tn) > ZERO:
        col = (f * rc) * mc                    # external reflection
    elif dot(wi, tn) < ZERO and dot(wo, tn) > ZERO:
        col = (f * rc) * mc                    # egress
    else:
        col = f * rc                           # internal reflection, ingress
    put3(new_ray, "color", col)
    put3(new_ray, "direction", wo)
    put3(new_ray, "inv_direction", ONE / wo)
    if from_camera:
        next_ray["c_importance"] = c_p
        ray["l_importance"] = l_p
        new_ray["tot_importance"] = f32(ray["tot_importance"]) * f32(new_ray["c_importance"])
    else:
        next_ray["l_importance"] = l_p
        ray["c_importance"] = c_p
        new_ray["tot_importance"] = f32(ray["tot_importance"]) * f32(new_ray["l_importance"])
    if f == ZERO:
        s["alive"] = False
        return
    path["rays"][i] = ray
    path["length"] = i + 1
    s["ray"] = new_ray.copy()
    s["new_ray"] = next_ray.copy()


# ---------------------------------------------------------------- K5: connect_paths, trace.metal:620-869
def cosine_geometry_term(a, b):
    dist = length(v3(b["origin"]) - v3(a["origin"]))
    cos_a = abs(dot(v3(a["direction"]), v3(a["normal"])))
    cos_b = abs(dot(v3(b["direction"]), v3(b["normal"])))
    return cos_a * cos_b / (dist * dist)


def _round_half_away(x):
    return np.floor(x + HALF) if x >= ZERO else -np.floor(-x + HALF)


class _Tracer:
    """Closest-hit queries of the strategy loop in two passes: the first pass runs the loop with every query answered
    'no hit' and records the rays it asks for (which rays are asked for depends on the culls only, not on the answers:
    a pair whose query misses is skipped, trace.metal:193 / :593); they are traversed in one batch; the second pass runs
    the loop again with the real answers."""
    def __init__(self):
        self.rays, self.answers, self.cursor = [], None, 0

    def trace(self, o, d):
        if self.answers is None:
            self.rays.append((o.copy(), d.copy()))
            return -1, INF
        k = self.cursor
        self.cursor += 1
        return int(self.answers[0][k]), f32(self.answers[1][k])

    def resolve(self, boxes, triangles):
        if self.rays:
            o = np.array([r[0] for r in self.rays], f32)
            d = np.array([r[1] for r in self.rays], f32)
            bi, bt, _, _ = npk.traverse(o, d, boxes, triangles)
        else:
            bi, bt = np.zeros(0, np.int32), np.zeros(0, f32)
        self.answers, self.cursor = (bi, bt), 0


def connect_paths(camera_paths, light_paths, triangles, materials, boxes, camera, Ray, WeightAggregator, n_light):
    """Returns dict(aggregators, out, light_pixel_indices, light_path_indices, light_ray_indices, light_weights,
    light_shade); the five light arrays have length n_light and start as reset_light_indices leaves them
    (indices -1, the others 0: trace.metal:967-979 / renderer.py)."""
    B = len(camera_paths)
    res = None
    tracer = _Tracer()
    for second in (False, True):
        if second:
            tracer.resolve(boxes, triangles)
        res = dict(aggregators=np.zeros(B, WeightAggregator), out=np.zeros((B, 4), f32),
                   light_pixel_indices=np.full(n_light, -1, np.int32), light_path_indices=np.zeros(n_light, np.int32),
                   light_ray_indices=np.zeros(n_light, np.int32), light_weights=np.zeros(n_light, f32),
                   light_shade=np.zeros(n_light, f32))
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            for id_ in range(B):
                _connect_pixel(id_, camera_paths[id_], light_paths[id_], triangles, materials, camera.reshape(-1)[0], Ray, tracer, res)
    return res


def _world_ray_to_camera_ray(triangles, materials, c, world_ray, camera_ray, tracer):
    """trace.metal:570-617; returns pixel_idx (or None when it stays untouched) and updates camera_ray in place."""
    if int(materials[int(triangles[int(world_ray["triangle"])]["material"])]["type"]) > 0:
        return None
    wo = v3(world_ray["origin"])
    focal, cdir = v3(c["focal_point"]), v3(c["direction"])
    tdir = normalize(focal - wo)
    if dot(tdir, cdir) > ZERO:
        return None
    best_i, best_t = tracer.trace(wo, tdir)
    if best_i == -1:
        return None
    if int(triangles[best_i]["is_camera"]) == 0:
        return None
    camera_point = wo + best_t * tdir
    center = v3(c["center"])
    x = dot(camera_point - center, v3(c["dx"]))
    y = dot(camera_point - center, v3(c["dy"]))
    W, H = int(c["pixel_width"]), int(c["pixel_height"])
    pixel_x = int(_round_half_away((x / f32(c["phys_width"]) + HALF) * f32(W)))
    pixel_y = int(_round_half_away((y / f32(c["phys_height"]) + HALF) * f32(H)))
    pixel_idx = pixel_y * W + pixel_x
    put3(camera_ray, "origin", camera_point)
    cd = normalize(focal - camera_point)
    put3(camera_ray, "direction", cd)
    put3(camera_ray, "inv_direction", ONE / cd)
    put3(camera_ray, "normal", cdir)
    camera_ray["material"] = 7
    put3(camera_ray, "color", V(1, 1, 1))
    camera_ray["triangle"] = best_i
    camera_ray["tot_importance"] = ONE
    camera_ray["hit_light"] = -1
    camera_ray["hit_camera"] = best_i
    return pixel_idx


def _visibility_test(a, b, tracer):
    ao = v3(a["origin"])
    direction = normalize(v3(b["origin"]) - ao)
    best_i, _ = tracer.trace(ao, direction)
    if best_i == -1:
        return False
    if best_i == int(a["triangle"]):
        return False
    return best_i == int(b["triangle"])


def _connect_pixel(id_, camera_path_in, light_path, triangles, materials, c, Ray, tracer, res):
    camera_path = camera_path_in.copy()
    cached_camera_zero = camera_path["rays"][0].copy()
    total = V(0, 0, 0)
    pixel_idx = int(cached_camera_zero["pixel_idx"])
    W, H = int(c["pixel_width"]), int(c["pixel_height"])
    total_pixels = W * H
    contrib_weight_sum = ZERO
    Lc, Ll = int(camera_path["length"]), int(light_path["length"])

    def mtype(ray):
        return int(materials[int(ray["material"])]["type"])

    for t in range(1, Lc + 1):
        for s in range(0, Ll + 1):
            if t + s < 2:
                continue
            light_ray = np.zeros((), Ray)
            light_ray["triangle"] = -1
            camera_ray = np.zeros((), Ray)
            camera_ray["triangle"] = -1
            camera_path["rays"][0] = cached_camera_zero
            dir_l_to_c = V(0, 0, 0)
            light_pixel_idx = -1
            if s == 0:
                camera_ray = camera_path["rays"][t - 1].copy()
                if int(camera_ray["hit_light"]) < 0:
                    continue
            elif t == 1:
                light_ray = light_path["rays"][s - 1].copy()
                cam0 = camera_path["rays"][0].copy()
                px = _world_ray_to_camera_ray(triangles, materials, c, light_ray, cam0, tracer)
                camera_path["rays"][0] = cam0
                if px is not None:
                    light_pixel_idx = px
                if light_pixel_idx == -1:
                    continue
                camera_ray = camera_path["rays"][0].copy()
                dir_l_to_c = normalize(v3(camera_ray["origin"]) - v3(light_ray["origin"]))
            else:
                camera_ray = camera_path["rays"][t - 1].copy()
                light_ray = light_path["rays"][s - 1].copy()
                if mtype(light_ray) > 0:
                    continue
                if mtype(camera_ray) > 0:
                    continue
                dir_l_to_c = normalize(v3(camera_ray["origin"]) - v3(light_ray["origin"]))
                if dot(v3(light_ray["normal"]), dir_l_to_c) < DELTA:
                    continue
                if dot(v3(camera_ray["normal"]), -dir_l_to_c) < DELTA:
                    continue
                if not _visibility_test(light_ray, camera_ray, tracer):
                    continue

            def get_ray(i):
                return light_path["rays"][i] if i < s else camera_path["rays"][t + s - i - 1]

            p_ratios = np.zeros(32, f32)
            p_values = np.zeros(32, f32)
            for i in range(s + t):
                if i == 0:
                    a, b = get_ray(0), get_ray(1)
                    num = f32(a["l_importance"])
                    den = f32(a["c_importance"]) * cosine_geometry_term(a, b)
                elif i == s + t - 1:
                    a, b = get_ray(s + t - 1), get_ray(s + t - 2)
                    num = f32(a["l_importance"]) * cosine_geometry_term(a, b)
                    den = f32(a["c_importance"])
                else:
                    a, b, cc = get_ray(i - 1), get_ray(i), get_ray(i + 1)
                    num = f32(b["l_importance"]) * cosine_geometry_term(a, b)
                    den = f32(b["c_importance"]) * cosine_geometry_term(b, cc)
                p_ratios[i] = num / den
            prior_camera_importance = f32(camera_ray["tot_importance"])
            prior_light_importance = ONE if s == 0 else f32(light_ray["tot_importance"])
            p_s = prior_camera_importance * prior_light_importance
            p_i = p_s
            for i in range(s, s + t + 1):
                p_values[i + 1] = p_ratios[i] * p_i
                p_i = p_values[i + 1]
            p_i = p_s
            for i in range(s - 1, -1, -1):
                p_values[i] = p_i / p_ratios[i]
                p_i = p_values[i]
            p_values[s] = p_s
            for i in range(s + t):
                if mtype(get_ray(i)) > 0:
                    p_values[i] = ZERO
                    p_values[i + 1] = ZERO
            p_values[s + t] = ZERO
            total_p = ZERO
            for i in range(s + t + 1):
                total_p = total_p + p_values[i]
            if p_values[s] > ZERO and total_p > ZERO:
                w = p_values[s] / total_p
            else:
                continue
            color = V(1, 1, 1)
            g = ONE
            new_light_f = ONE
            if s == 0:
                prior_color = v3(camera_path["rays"][t - 2]["color"])
                emission = v3(materials[int(camera_ray["material"])]["emission"])
                color = prior_color * emission
            elif t == 1:
                prior_light_ind = max(0, s - 2)
                prior_color = v3(light_path["rays"][prior_light_ind]["color"])
                if s > 1:
                    new_light_f = abs(dot(dir_l_to_c, v3(light_ray["normal"]))) / PI
                color = (prior_color * new_light_f) * v3(materials[int(light_ray["material"])]["color"])
                g = cosine_geometry_term(light_ray, camera_ray)
            else:
                prior_camera_color = v3(camera_path["rays"][t - 2]["color"])
                camera_material = materials[int(camera_ray["material"])]
                new_camera_f = abs(dot(-dir_l_to_c, v3(camera_ray["normal"]))) / PI
                camera_color = (prior_camera_color * new_camera_f) * v3(camera_material["color"])
                if s == 1:
                    light_color = v3(materials[int(light_ray["material"])]["emission"])
                else:
                    prior_light_color = v3(light_path["rays"][s - 2]["color"])
                    light_material = materials[int(light_ray["material"])]
                    new_light_f = abs(dot(dir_l_to_c, v3(light_ray["normal"]))) / PI
                    light_color = (prior_light_color * new_light_f) * v3(light_material["color"])
                color = camera_color * light_color
                g = cosine_geometry_term(camera_ray, light_ray)
            if t != 1:
                total = total + ((w * g) * color) / p_s
                contrib_weight_sum = contrib_weight_sum + w
            else:
                k = id_ + s * total_pixels
                res["light_pixel_indices"][k] = light_pixel_idx
                res["light_path_indices"][k] = id_
                res["light_ray_indices"][k] = s - 1
                res["light_weights"][k] = w
                res["light_shade"][k] = new_light_f * g / p_s

    # reconstruction-filter weights, trace.metal:827-862
    agg = res["aggregators"][id_]
    weight_sum = ZERO
    ppw = f32(c["phys_width"]) / f32(W)
    pph = f32(c["phys_height"]) / f32(H)
    sigma = HALF * np.sqrt(ppw * ppw + pph * pph)
    weights = np.zeros((3, 3), f32)
    film = v3(camera_path_in["rays"][0]["origin"])
    center, dx, dy = v3(c["center"]), v3(c["dx"]), v3(c["dy"])
    for i in range(-1, 2):
        for j in range(-1, 2):
            nx = (pixel_idx % W) + i
            ny = (pixel_idx // W) + j
            if nx < 0 or nx >= W or ny < 0 or ny >= H:
                continue
            idx = ny * W + nx
            if idx < 0 or idx >= W * H:
                continue
            xn = (f32(nx) - HALF * f32(W)) / f32(W)
            yn = (f32(ny) - HALF * f32(H)) / f32(H)
            pc = (center + (xn * f32(c["phys_width"])) * dx) + (yn * f32(c["phys_height"])) * dy
            dist = length(pc - film)
            wgt = d_exp(-dist * dist / (TWO * sigma * sigma))
            weights[i + 1][j + 1] = wgt
            weight_sum = weight_sum + wgt
    if weight_sum != ZERO:
        weights = (weights / weight_sum).astype(f32)
    agg["weights"] = weights
    agg["total_contribution"][:3] = total
    agg["total_contribution"][3] = 0
    agg["contrib_weight_sum"] = contrib_weight_sum
    res["out"][id_] = (total[0], total[1], total[2], 1.0)


# ---------------------------------------------------------------- K6, K7 (+ its driver loop), K8
def adaptive_finalize_samples(aggregators, camera, sample_bin_offsets):
    """trace.metal:981-1018.  Returns (out float32[B,4], sample_counts uint32[B], sample_weights float32[B])."""
    c = camera.reshape(-1)[0]
    W, H = int(c["pixel_width"]), int(c["pixel_height"])
    B = len(aggregators)
    out = np.zeros((B, 4), f32)
    counts = np.zeros(B, np.uint32)
    sw = np.zeros(B, f32)
    for id_ in range(B):
        total = V(0, 0, 0)
        weight_sum = ZERO
        for i in range(-1, 2):
            for j in range(-1, 2):
                sx, sy = (id_ % W) + i, (id_ // W) + j
                if sx < 0 or sx >= W or sy < 0 or sy >= H:
                    continue
                idx = sy * W + sx
                if idx < 0 or idx >= W * H:
                    continue
                for k in range(int(sample_bin_offsets[idx]), int(sample_bin_offsets[idx + 1])):
                    wa = aggregators[k]
                    weight = f32(wa["weights"][1 - i][1 - j])
                    total = total + weight * v3(wa["total_contribution"])
                    weight_sum = weight_sum + weight * f32(wa["contrib_weight_sum"])
        counts[id_] = int(sample_bin_offsets[id_ + 1]) - int(sample_bin_offsets[id_])
        out[id_] = (total[0], total[1], total[2], 1.0)
        sw[id_] = weight_sum
    return out, counts, sw


def light_sort_all(pix, path, ray, wgt, shd):
    """The bitonic network of renderer.py:212-231 driving trace.metal:872-934, all (stage, passOfStage) launches; sorts the five
    arrays IN PLACE by pixel index.  The pairs of one launch are disjoint, so a launch is one vectorised step."""
    n = len(pix)
    log_n = int(np.log2(n))
    g = np.arange(n // 2, dtype=np.int64)                  # global_pair_id = id * pairs_per_thread + p, id < n / 8, p < 4
    for stage in range(1, log_n + 1):
        for pass_of_stage in range(stage, 0, -1):
            pd, bw = 1 << (pass_of_stage - 1), 1 << stage
            left = (g // pd) * pd * 2 + (g % pd)
            right = left + pd
            ok = (right < n) & (left < n)
            asc = (g & (bw >> 1)) == 0
            lp, rp = pix[np.where(ok, left, 0)], pix[np.where(ok, right, 0)]
            swap = ok & ((asc & (lp > rp)) | (~asc & (lp < rp)))
            a, b = left[swap], right[swap]
            for arr in (pix, path, ray, wgt, shd):
                arr[a], arr[b] = arr[b].copy(), arr[a].copy()


def light_image_gather(light_paths, materials, path_indices, ray_indices, bins, offset, weights, shades, sum_weights):
    """trace.metal:937-964.  Returns light_image float32[B,4]; adds to sum_weights in place."""
    B = len(light_paths)
    img = np.zeros((B, 4), f32)
    with np.errstate(over="ignore", invalid="ignore"):
        for id_ in range(B):
            total = V(0, 0, 0)
            weight_sum = ZERO
            for i in range(int(bins[id_]) + int(offset), int(bins[id_ + 1]) + int(offset)):
                path = light_paths[int(path_indices[i])]
                ray_idx = int(ray_indices[i])
                ray = path["rays"][ray_idx]
                prior = path["rays"][max(0, ray_idx - 1)]
                mat = materials[int(ray["material"])]
                total = total + ((f32(weights[i]) * f32(shades[i])) * v3(prior["color"])) * v3(mat["color"])
                weight_sum = weight_sum + f32(weights[i])
            img[id_] = (total[0], total[1], total[2], 1.0)
            sum_weights[id_] = f32(sum_weights[id_]) + weight_sum
    return img
