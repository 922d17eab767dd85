"""mirt_oracle_pt_np.py — a SECOND, independent CPU restatement of the path-traced mode: the reference's WGSL fragment
shader (src/raytracer/raytracer.wgsl) transcribed to numpy float32, statement by statement.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/test_oracle_pt_numpy.py, never by the product.
PARITY UNPINNED: the reference holds no fixtures for this path and its shader cannot run here (no wgpu / naga).

Why it exists.  The C oracle (oracle/mirt_oracle_pt.c) and the HIP kernels share a set of spec decisions where WGSL
leaves the arithmetic open (DESIGN.md "PT spec decisions": explicit fma placement, one 1/a per ray, the "mirt-math v1"
polynomial sin/cos/acos/atan2/pow, exact integer accumulation), so their bit-for-bit agreement proves kernel == oracle,
not oracle == shader.  This file makes NONE of those choices: it was written from the shader text alone, without looking
at the C restatement; every WGSL expression is one numpy float32 expression in the shader's own association, the
elementary functions are numpy's, samples are summed in float32 frame by frame exactly as the fragment shader does
(wgsl:61-80), and the RNG is the shader's one stream per pixel and frame (wgsl:105-122, 498-502) -- what the library
reproduces with MirtParams.frame_spp = num_samples_per_pixel.  The two restatements therefore differ by rounding only
(a few ulp per operation); a path tracer amplifies that on the rare paths where it flips a branch, so the comparison is
statistical: tests/test_oracle_pt_numpy.py states the tolerance.

Reference text followed (raytracer.wgsl unless noted):
  53-81    fsMain            pixel index, initRng, accumulate, 1/N, uncharted2
  83-103   uncharted2 / uncharted2Tonemap
  105-122  samplePixel       jitter, u, 1 - v, sum of rayColor
  124-172  rayColor          bounce loop, nearest hit scan (closestT shrinks), sky on a miss, throughput * color
  174-204  scatterRay        switch material.id: 0 lambertian, 1 metal, 2 dielectric, 3 checkerboard, default missing
  198-242  lambertian        sample (r1, r2), eval / pdf, pixarOnb
  244-248  scatterMetal      reflect + fuzz * unit-sphere sample
  250-298  scatterDielectric refract / schlick (the drawn reflection is discarded) / reflect
  300-314  checkerboard, missing material
  316-343  radiance          Hosek-Wilkie evaluation from the 144-byte state
  377-387  textureLookup
  407-444  rayIntersectSphere, sphereIntersection, rayPointAtParameter
  456-491  cameraMakeRay, rngNextVec3InUnitDisk, rngNextVec3InUnitSphere
  493-521  rngNextFloat, initRng, rngNextInt, jenkinsHash
  mod.rs:284, 323-350        frame_number starts at 1 and advances by one per frame
  main.rs:465                the Bgra8UnormSrgb surface applies the sRGB OETF and the unorm conversion
Two things the shader does not define are taken from the library's specification: the sky without a Hosek state is the
RTIOW gradient, and a texel index past the table is clamped to its last entry (S8).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
u32 = np.uint32

EPSILON = f32(0.001)
PI = f32(3.1415927)
FRAC_1_PI = f32(0.31830987)
MIN_T = f32(0.001)
MAX_T = f32(1000.0)

FLAG_SKY_HOSEK, FLAG_NO_TONEMAP, FLAG_NO_SRGB = 1, 2, 4


# ---------------------------------------------------------------- RNG (wgsl:493-521)
def jenkins_hash(x):
    x = np.asarray(x, dtype=u32).copy()
    with np.errstate(over="ignore"):
        x += x << u32(10)
        x ^= x >> u32(6)
        x += x << u32(3)
        x ^= x >> u32(11)
        x += x << u32(15)
    return x


def _rng_next_int(state):
    with np.errstate(over="ignore"):
        old = state + u32(747796405) + u32(2891336453)
        word = ((old >> ((old >> u32(28)) + u32(4))) ^ old) * u32(277803737)
    return (word >> u32(22)) ^ word


class Rng:
    """One u32 state per pixel; `next_float(mask)` advances the pixels in `mask` only."""

    def __init__(self, state):
        self.state = state.astype(u32)

    def next_float(self, mask):
        self.state = np.where(mask, _rng_next_int(self.state), self.state)
        return self.state.astype(f32) / f32(0xFFFFFFFF)        # f32(0xffffffffu) == 2^32


def init_rng(x, y, width, frame):
    with np.errstate(over="ignore"):
        seed = (x.astype(u32) * u32(1) + y.astype(u32) * u32(width)) ^ jenkins_hash(np.full(x.shape, frame, dtype=u32))
    return Rng(jenkins_hash(seed))


# ---------------------------------------------------------------- small vector helpers (vec3 = list of 3 arrays)
def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _normalize(v):
    ln = np.sqrt(_dot(v, v))
    return [v[0] / ln, v[1] / ln, v[2] / ln]


def _reflect(e1, e2):
    """WGSL reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2."""
    k = f32(2.0) * _dot(e2, e1)
    return [e1[i] - k * e2[i] for i in range(3)]


def _where3(m, a, b):
    return [np.where(m, a[i], b[i]) for i in range(3)]


def _as_u32(x):
    """WGSL u32(f32): truncation; out-of-range values are clamped (the shader only feeds it [0, 2^31))."""
    y = np.where(np.isnan(x), f32(0), x)
    return np.trunc(np.clip(y, f32(0), f32(4294967040.0))).astype(np.uint64)


def _unit_disk(rng, mask):
    r = np.sqrt(rng.next_float(mask))
    alpha = f32(2.0) * PI * rng.next_float(mask)
    return r * np.cos(alpha), r * np.sin(alpha)


def _unit_sphere(rng, mask):
    r = np.power(rng.next_float(mask), f32(0.33333))
    theta = PI * rng.next_float(mask)
    phi = f32(2.0) * PI * rng.next_float(mask)
    return [r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)]


# ---------------------------------------------------------------- scene access
class Scene:
    """camera: dict of float32 vectors eye, horizontal, vertical, u, v, lower_left_corner + lens_radius
    spheres: float32 [n, 4] (centre, radius) + uint32 [n] material index; materials: list of
    (id, (w1, h1, off1), (w2, h2, off2), x); texels: float32 [T, 3]; sky: None or (params[27], radiances[3], sun[3])."""

    def __init__(self, camera, centres_radii, material_idx, materials, texels, sky=None):
        self.cam = {k: (np.asarray(v, dtype=f32) if k != "lens_radius" else f32(v)) for k, v in camera.items()}
        self.spheres = np.asarray(centres_radii, dtype=f32).reshape(-1, 4)
        self.material_idx = np.asarray(material_idx, dtype=np.int64)
        self.mat_id = np.array([m[0] for m in materials], dtype=np.int64)
        self.mat_desc = np.array([[m[1], m[2]] for m in materials], dtype=np.int64).reshape(-1, 2, 3)
        self.mat_x = np.array([m[3] for m in materials], dtype=f32)
        self.texels = np.asarray(texels, dtype=f32).reshape(-1, 3)
        self.sky = sky


def _texture_lookup(sc, desc, u, v):
    """wgsl:377-387; desc = int64 [N, 3] (width, height, offset) per pixel."""
    uc = np.clip(u, f32(0), f32(1))
    vc = f32(1.0) - np.clip(v, f32(0), f32(1))
    w, h, off = desc[:, 0], desc[:, 1], desc[:, 2]
    j = _as_u32(uc * w.astype(f32))
    i = _as_u32(vc * h.astype(f32))
    idx = (i * w.astype(np.uint64) + j) & np.uint64(0xFFFFFFFF)
    g = np.minimum(off.astype(np.uint64) + idx, np.uint64(max(1, sc.texels.shape[0]) - 1)).astype(np.int64)
    t = sc.texels[g]
    return [t[:, 0], t[:, 1], t[:, 2]]


def _pixar_onb(n):
    s = np.where(n[2] >= f32(0), f32(1.0), f32(-1.0))
    a = f32(-1.0) / (s + n[2])
    b = n[0] * n[1] * a
    uu = [f32(1.0) + s * n[0] * n[0] * a, s * b, -s * n[0]]
    vv = [b, s + n[1] * n[1] * a, -n[1]]
    return uu, vv


def _scatter_lambertian(sc, hit, desc, rng, mask):
    # sampleLambertian
    r1 = rng.next_float(mask)
    r2 = rng.next_float(mask)
    sqrt_r2 = np.sqrt(r2)
    z = np.sqrt(f32(1.0) - r2)
    phi = f32(2.0) * PI * r1
    x = np.cos(phi) * sqrt_r2
    y = np.sin(phi) * sqrt_r2
    uu, vv = _pixar_onb(hit["n"])
    wi = [uu[k] * x + vv[k] * y + hit["n"][k] * z for k in range(3)]          # mat3x3(u, v, n) * vec3(x, y, z)
    tex = _texture_lookup(sc, desc, hit["u"], hit["v"])
    dn = _dot(hit["n"], wi)
    ev = [tex[k] * FRAC_1_PI * np.maximum(EPSILON, dn) for k in range(3)]      # evalLambertian
    pdf = np.maximum(EPSILON, dn * FRAC_1_PI)                                  # pdfLambertian
    return wi, [ev[k] / pdf for k in range(3)]


def _scatter(sc, ray_d, hit, mat, rng, mask):
    """scatterRay (wgsl:174-314) for the pixels in `mask` (whose nearest hit has material index `mat`).
    Returns (new direction, albedo); the new origin is hit.p."""
    n_px = mask.shape[0]
    mid = sc.mat_id[mat]
    new_d = [np.zeros(n_px, f32) for _ in range(3)]
    alb = [np.ones(n_px, f32) for _ in range(3)]
    with np.errstate(all="ignore"):
        m0 = mask & (mid == 0)
        if m0.any():
            wi, thr = _scatter_lambertian(sc, hit, sc.mat_desc[mat, 0], rng, m0)
            new_d, alb = _where3(m0, wi, new_d), _where3(m0, thr, alb)
        m1 = mask & (mid == 1)
        if m1.any():
            fuzz = sc.mat_x[mat]
            refl = _reflect(ray_d, hit["n"])
            s = _unit_sphere(rng, m1)
            d = [refl[k] + fuzz * s[k] for k in range(3)]
            tex = _texture_lookup(sc, sc.mat_desc[mat, 0], hit["u"], hit["v"])
            new_d, alb = _where3(m1, d, new_d), _where3(m1, tex, alb)
        m2 = mask & (mid == 2)
        if m2.any():
            ri = sc.mat_x[mat]
            wo = ray_d
            inside = _dot(wo, hit["n"]) > f32(0)
            outward = _where3(inside, [-hit["n"][k] for k in range(3)], hit["n"])
            ni_over_nt = np.where(inside, ri, f32(1.0) / ri)
            # (cosine / schlick feed a comparison whose outcome the shader discards: only the draw matters)
            uv = _normalize(wo)
            dt = _dot(uv, outward)
            disc = f32(1.0) - ni_over_nt * ni_over_nt * (f32(1.0) - dt * dt)
            refracts = disc > f32(0)
            sq = np.sqrt(np.where(refracts, disc, f32(0)))
            refr = _normalize([ni_over_nt * (uv[k] - dt * outward[k]) - sq * outward[k] for k in range(3)])
            rng.next_float(m2 & refracts)                    # `if rngNextFloat(rngState) < reflectionProb { reflect(...); }`
            refl = _reflect(wo, hit["n"])
            d = _where3(refracts, refr, refl)
            new_d, alb = _where3(m2, d, new_d), _where3(m2, [np.ones(n_px, f32)] * 3, alb)
        m3 = mask & (mid == 3)
        if m3.any():
            p = hit["p"]
            sines = np.sin(f32(5.0) * p[0]) * np.sin(f32(5.0) * p[1]) * np.sin(f32(5.0) * p[2])
            desc = np.where((sines < f32(0))[:, None], sc.mat_desc[mat, 0], sc.mat_desc[mat, 1])
            wi, thr = _scatter_lambertian(sc, hit, desc, rng, m3)
            new_d, alb = _where3(m3, wi, new_d), _where3(m3, thr, alb)
        m4 = mask & (mid > 3)
        if m4.any():
            s = _unit_sphere(rng, m4)
            d = [hit["n"][k] + s[k] for k in range(3)]
            pink = [np.full(n_px, c, f32) for c in (0.9921, 0.24705, 0.57254)]
            new_d, alb = _where3(m4, d, new_d), _where3(m4, pink, alb)
    return new_d, alb


def _hosek_radiance(sky, theta, gamma, ch):
    params, radiances, _ = sky
    p = [f32(params[9 * ch + k]) for k in range(9)]
    r = f32(radiances[ch])
    cos_gamma = np.cos(gamma)
    cos_gamma2 = cos_gamma * cos_gamma
    cos_theta = np.abs(np.cos(theta))
    exp_m = np.exp(p[4] * gamma)
    ray_m = cos_gamma2
    mie_lhs = f32(1.0) + cos_gamma2
    mie_rhs = np.power(f32(1.0) + p[8] * p[8] - f32(2.0) * p[8] * cos_gamma, f32(1.5))
    mie_m = mie_lhs / mie_rhs
    zenith = np.sqrt(cos_theta)
    lhs = f32(1.0) + p[0] * np.exp(p[1] / (cos_theta + f32(0.01)))
    rhs = p[2] + p[3] * exp_m + p[5] * ray_m + p[6] * mie_m + p[7] * zenith
    return r * (lhs * rhs)


def _sky(sc, d, hosek):
    v = _normalize(d)
    if hosek:
        s = [f32(c) for c in sc.sky[2][:3]]
        theta = np.arccos(v[1])
        gamma = np.arccos(np.clip(_dot(v, s), f32(-1), f32(1)))
        return [_hosek_radiance(sc.sky, theta, gamma, ch) for ch in range(3)]
    t = f32(0.5) * (v[1] + f32(1.0))                        # the library's default sky: RTIOW gradient
    return [(f32(1.0) - t) * f32(1.0) + t * f32(c) for c in (0.5, 0.7, 1.0)]


def _ray_color(sc, o, d, rng, num_bounces, hosek):
    """rayColor (wgsl:124-172) for all pixels at once; returns throughput * color."""
    n_px = o[0].shape[0]
    color = [np.zeros(n_px, f32) for _ in range(3)]
    thr = [np.ones(n_px, f32) for _ in range(3)]
    alive = np.ones(n_px, dtype=bool)
    with np.errstate(all="ignore"):
        for _bounce in range(num_bounces):
            if not alive.any():
                break
            closest = np.full(n_px, MAX_T, f32)
            hit = {"p": [np.zeros(n_px, f32) for _ in range(3)], "n": [np.zeros(n_px, f32) for _ in range(3)],
                   "u": np.zeros(n_px, f32), "v": np.zeros(n_px, f32)}
            mat = np.zeros(n_px, dtype=np.int64)
            a = _dot(d, d)
            for si in range(sc.spheres.shape[0]):
                c = [f32(sc.spheres[si, k]) for k in range(3)]
                radius = f32(sc.spheres[si, 3])
                oc = [o[k] - c[k] for k in range(3)]
                b = _dot(oc, d)
                cc = _dot(oc, oc) - radius * radius
                disc = b * b - a * cc
                pos = disc > f32(0)
                sq = np.sqrt(np.where(pos, b * b - a * cc, f32(0)))
                t0 = (-b - sq) / a
                ok0 = pos & (t0 < closest) & (t0 > MIN_T)
                t1 = (-b + sq) / a
                ok1 = pos & ~ok0 & (t1 < closest) & (t1 > MIN_T)
                t = np.where(ok0, t0, t1)
                h = alive & (ok0 | ok1)
                if not h.any():
                    continue
                # sphereIntersection
                p = [o[k] + t * d[k] for k in range(3)]
                n = [(f32(1.0) / radius) * (p[k] - c[k]) for k in range(3)]
                theta = np.arccos(-n[1])
                phi = np.arctan2(-n[2], n[0]) + PI
                uu = f32(0.5) * FRAC_1_PI * phi
                vv = FRAC_1_PI * theta
                closest = np.where(h, t, closest)
                hit["p"], hit["n"] = _where3(h, p, hit["p"]), _where3(h, n, hit["n"])
                hit["u"], hit["v"] = np.where(h, uu, hit["u"]), np.where(h, vv, hit["v"])
                mat = np.where(h, sc.material_idx[si], mat)
            hit_any = alive & (closest < MAX_T)
            miss = alive & ~hit_any
            if miss.any():
                c = _sky(sc, d, hosek)
                color = _where3(miss, c, color)
            if hit_any.any():
                new_d, alb = _scatter(sc, d, hit, mat, rng, hit_any)
                o = _where3(hit_any, hit["p"], o)
                d = _where3(hit_any, new_d, d)
                thr = _where3(hit_any, [thr[k] * alb[k] for k in range(3)], thr)
            alive = hit_any                                   # `break` after a miss
        return [thr[k] * color[k] for k in range(3)]


def _uncharted2_tonemap(x):
    a, b, c, d, e, f = (f32(v) for v in (0.15, 0.50, 0.10, 0.20, 0.02, 0.30))
    return ((x * (a * x + c * b) + d * e) / (x * (a * x + b) + d * f)) - e / f


def _uncharted2(x):
    curr = _uncharted2_tonemap(f32(0.246) * x)
    white_scale = f32(1.0) / _uncharted2_tonemap(f32(11.2))
    return white_scale * curr


def _srgb_oetf(x):
    x = np.clip(x, f32(0), f32(1))
    return np.where(x <= f32(0.0031308), f32(12.92) * x, f32(1.055) * np.power(x, f32(1.0 / 2.4)) - f32(0.055))


def render_pt(sc: Scene, width: int, height: int, frames: int, samples_per_frame: int, num_bounces: int = 8, flags: int = 0,
              rows=None, frames_before: int = 0):
    """`frames` calls of the fragment shader over the rows `rows` (default all), `samples_per_frame` samples each, after
    `frames_before` earlier calls (frame_number is never reset, mod.rs:284, 350, 385: an accumulation that starts after a camera
    move begins at a later frame).
    Returns (rgba8 uint8 [len(rows), width, 4], accumulated float32 [len(rows), width, 3] = the shader's imageBuffer)."""
    ys = np.arange(height, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    X, Y = np.meshgrid(np.arange(width, dtype=np.int64), ys)
    X, Y = X.ravel(), Y.ravel()
    n_px = X.size
    inv_w, inv_h = f32(1.0) / f32(width), f32(1.0) / f32(height)
    cam = sc.cam
    every = np.ones(n_px, dtype=bool)
    pixel = [np.zeros(n_px, f32) for _ in range(3)]           # imageBuffer[idx], cleared on the first frame
    hosek = bool(flags & FLAG_SKY_HOSEK)
    with np.errstate(all="ignore"):
        for frame in range(frames_before + 1, frames_before + frames + 1):      # frame_number starts at 1 (mod.rs:284) and never resets
            rng = init_rng(X, Y, width, frame)
            color = [np.zeros(n_px, f32) for _ in range(3)]
            for _s in range(samples_per_frame):               # samplePixel
                u = (X.astype(f32) + rng.next_float(every)) * inv_w
                v = (Y.astype(f32) + rng.next_float(every)) * inv_h
                # cameraMakeRay(camera, rngState, u, 1 - v)
                vv = f32(1.0) - v
                dx, dy = _unit_disk(rng, every)
                lx, ly = cam["lens_radius"] * dx, cam["lens_radius"] * dy
                lens = [lx * cam["u"][k] + ly * cam["v"][k] for k in range(3)]
                origin = [cam["eye"][k] + lens[k] for k in range(3)]
                direction = [cam["lower_left_corner"][k] + u * cam["horizontal"][k] + vv * cam["vertical"][k] - origin[k] for k in range(3)]
                c = _ray_color(sc, origin, direction, rng, num_bounces, hosek)
                color = [color[k] + c[k] for k in range(3)]
            pixel = [pixel[k] + color[k] for k in range(3)]
        inv_n = f32(1.0) / f32(frames * samples_per_frame)
        out = np.zeros((n_px, 4), dtype=np.uint8)
        out[:, 3] = 255
        for k in range(3):
            m = inv_n * pixel[k]
            if not flags & FLAG_NO_TONEMAP:
                m = _uncharted2(m)
            if not flags & FLAG_NO_SRGB:
                m = _srgb_oetf(m)
            q = np.clip(np.where(np.isnan(m), f32(0), m), f32(0), f32(1)) * f32(255.0) + f32(0.5)
            out[:, k] = np.trunc(q).astype(np.uint8)
    acc = np.stack(pixel, axis=1).reshape(len(ys), width, 3)
    return out.reshape(len(ys), width, 4), acc
