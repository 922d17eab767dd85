"""mirt_oracle_parity_np.py — a SECOND, independent CPU restatement of the reference's CPU pixel loop (parity mode).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/test_oracle_numpy.py, never by the
product.  PARITY UNPINNED: the reference holds no fixtures for this path and cannot be built here (Rust, no
toolchain); this file does not change that.  What it removes is the single-author risk of the C oracle
(oracle/mirt_oracle_parity.c): it is written in another language (numpy, float32 arrays, one IEEE operation per numpy
call, no fused multiply-add) directly from the reference's source text, file by file, without looking at the C
restatement, and the two must agree pixel for pixel (tests/test_oracle_numpy.py).

Reference text followed (all under /root/reference/src/raytracer/):
  layer.rs:264-282   Layer::set_data            the row-major pixel loop
  layer.rs:304-381   ray_color_per_pixel        sample loop, depth = 20 per PIXEL, material index 2, early returns
  layer.rs:413-444   ray_hit_world_raw          `closest_hit = old_hit` -> the LAST sphere hit in list order wins
  mod.rs:745-754     GpuCamera::make_ray        ((llc + u*horizontal) + v*vertical) - eye, no lens offset
  mod.rs:1000-1019   texture_lookup             nearest texel; called with the SCREEN-space (uu, vv)
  mod.rs:1121-1157   Sphere::closest_hit_raw    `discriminant < 0.0` rejects; first root, else second; [tmin, tmax]
  mod.rs:1217-1243   update_ray_hit_info        p, n = (1/r)(p - c), face flip (t is never written)
  mod.rs:1095-1110   set_face_normal
  mod.rs:1292-1315   scatter_metal              reflect(unit_vertor(d), n); accepted if dot(reflected, n) > 0
  math.rs:4-9        coord_to_color             x as f32 / w
  math.rs:15-17      vec3_to_rgb8               Rust `as u8`: truncate, saturate, NaN -> 0
  math.rs:94-110     clamp, random_f32          random_f32() = rand / (f32::MAX + 1.0) <= 2.94e-39
  math.rs:147-159    unit_vertor (v / 3: `.len()` is the element count), reflect (v - (2 dot(v,n)) n)
nalgebra semantics used: dot of 3-vectors = (a0*b0 + a1*b1) + a2*b2; normalize = v / sqrt(dot(v, v)).

One documented deviation, shared with the C oracle and the kernel: the jitter `random_f32()` is at most
2.94e-39 and is taken as exactly +0.0, and the texel index is clamped to the table where the reference would panic.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
F32_MAX = np.finfo(np.float32).max


def _dot(a, b):
    """nalgebra 3-vector dot, unfused: (a0*b0 + a1*b1) + a2*b2."""
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _as_u8(x: np.ndarray) -> np.ndarray:
    """Rust `f32 as u8` (math.rs:15-17): truncate toward zero, saturate to [0, 255], NaN -> 0."""
    y = np.where(np.isnan(x), f32(0), x)
    y = np.clip(y, f32(0), f32(255))
    return np.trunc(y).astype(np.uint8)


def _as_u32(x: np.ndarray) -> np.ndarray:
    """Rust `f32 as u32`: truncate, saturate, NaN -> 0."""
    y = np.where(np.isnan(x), f32(0), x)
    y = np.clip(y, f32(0), f32(4294967040.0))          # largest f32 below 2^32; values above saturate to u32::MAX
    out = np.trunc(y).astype(np.uint64)
    out = np.where(x >= f32(4294967296.0), np.uint64(0xffffffff), out)
    return out


def _world_hit(spheres, ro, rd, tmin: np.float32, tmax: np.float32, rec_t: np.ndarray):
    """`Layer::ray_hit_world_raw` over `Sphere::closest_hit_raw` for a batch of rays (arrays of shape [3][N]).

    Returns (hit_anything [N] bool, p [3][N], n [3][N]) = the fields of `rec` the caller reads afterwards."""
    n_rays = ro[0].shape[0]
    hit_anything = np.zeros(n_rays, dtype=bool)
    closest_hit = np.full(n_rays, tmax, dtype=f32)
    old_hit = rec_t                                     # (*rec).t at entry; nobody ever writes it
    rec_p = [np.zeros(n_rays, f32) for _ in range(3)]
    rec_n = [np.zeros(n_rays, f32) for _ in range(3)]
    a = _dot(rd, rd)                                    # the same for every sphere
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        for (cx, cy, cz, radius) in spheres:
            c = (f32(cx), f32(cy), f32(cz))
            r = f32(radius)
            oc = [ro[k] - c[k] for k in range(3)]
            half_b = _dot(oc, rd)
            cc = _dot(oc, oc) - r * r
            disc = half_b * half_b - a * cc
            alive = ~(disc < f32(0))                    # `if discriminant < 0.0 { return }`: 0 and NaN go on
            sq = np.sqrt(np.where(alive, disc, f32(0)))
            sq = np.where(np.isnan(disc), f32(np.nan), sq)
            t = (-half_b - sq) / a
            first_bad = (t < tmin) | (closest_hit < t)
            t2 = (-half_b + sq) / a
            second_bad = (t2 < tmin) | (closest_hit < t2)
            t = np.where(first_bad, t2, t)
            hit = alive & ~(first_bad & second_bad)
            # update_ray_hit_info (mod.rs:1217-1243); `if t < 0.0 return` cannot trigger for t >= tmin > 0
            p = [ro[k] + rd[k] * t for k in range(3)]
            inv_r = f32(1.0) / r
            n = [inv_r * (p[k] - c[k]) for k in range(3)]
            front = _dot(rd, n) < f32(0)                # set_face_normal
            n = [np.where(front, n[k], -n[k]) for k in range(3)]
            for k in range(3):
                rec_p[k] = np.where(hit, p[k], rec_p[k])
                rec_n[k] = np.where(hit, n[k], rec_n[k])
            hit_anything |= hit
            closest_hit = np.where(hit, old_hit, closest_hit)       # `closest_hit = old_hit` (= f32::MAX)
    return hit_anything, rec_p, rec_n


def render_parity(camera, spheres, material2, texels, width: int, height: int, spp: int, rows=None) -> np.ndarray:
    """`Layer::set_data` for the rows `rows` (default all) -> uint8 [len(rows), width, 4] (RGBA8, A = 255).

    camera     dict of float32 3-vectors: eye, horizontal, vertical, lower_left_corner (GpuCamera, mod.rs:681-697)
    spheres    sequence of (cx, cy, cz, radius) in list order
    material2  (tex_width, tex_height, tex_offset, x) of material_data[2] (layer.rs:345-349); None for an empty world
    texels     float32 [T, 3], the flattened texel table (layer.rs:44)"""
    ys = np.arange(height, dtype=np.uint32) if rows is None else np.asarray(rows, dtype=np.uint32)
    X, Y = np.meshgrid(np.arange(width, dtype=np.uint32), ys)
    X, Y = X.ravel(), Y.ravel()
    n_px = X.size
    wf, hf = f32(width), f32(height)
    u = X.astype(f32) / wf                               # coord_to_color
    v = Y.astype(f32) / hf
    uu, vv = u + f32(0.0), v + f32(0.0)                  # jitter == +0.0 (see the header)
    eye = [f32(camera["eye"][k]) for k in range(3)]
    hor = [f32(camera["horizontal"][k]) for k in range(3)]
    ver = [f32(camera["vertical"][k]) for k in range(3)]
    llc = [f32(camera["lower_left_corner"][k]) for k in range(3)]
    ro = [np.full(n_px, eye[k], f32) for k in range(3)]
    rd = [((llc[k] + uu * hor[k]) + vv * ver[k]) - eye[k] for k in range(3)]      # make_ray
    rec_t = np.full(n_px, F32_MAX, f32)

    out = np.zeros((n_px, 4), dtype=np.uint8)
    out[:, 3] = 255
    grad = np.stack([_as_u8(v * f32(255.0)), _as_u8(u * f32(255.0)), _as_u8(np.full(n_px, 255.0, f32))], axis=1)

    prim, p1, n1 = _world_hit(spheres, ro, rd, f32(0.001), F32_MAX, rec_t)
    # every sample of a pixel is identical (jitter 0), so the sample loop has four outcomes:
    #   primary miss                                   -> the loop runs out: gradient
    #   primary hit, scatter_metal rejects             -> black at sample 0
    #   primary hit, scattered ray hits                -> colour at sample 0
    #   primary hit, scattered ray misses              -> nothing returned; depth drops by one per sample:
    #                                                     sample 20 finds depth == 0 -> black iff spp >= 21, else gradient
    colour = np.zeros((n_px, 3), dtype=np.uint8)
    scat_ok = np.zeros(n_px, dtype=bool)
    sec = np.zeros(n_px, dtype=bool)
    if prim.any():
        tw, th, toff, fuzzy = material2
        # texture_lookup(desc1 of material 2, texels, uu, vv)
        uc = np.clip(uu, f32(0), f32(1))
        vc = f32(1.0) - np.clip(vv, f32(0), f32(1))
        j = _as_u32(uc * f32(tw))
        i = _as_u32(vc * f32(th))
        idx = (i * np.uint64(tw) + j) & np.uint64(0xffffffff)          # u32 arithmetic
        g = np.minimum(np.uint64(toff) + idx, np.uint64(texels.shape[0] - 1))
        albedo = texels[g.astype(np.int64)].astype(f32)
        # scatter_metal: reflect(unit_vertor(d), n)
        unit = [rd[k] / f32(3.0) for k in range(3)]
        kk = f32(2.0) * _dot(unit, n1)
        sd = [unit[k] - kk * n1[k] for k in range(3)]
        scat_ok = prim & (_dot(sd, n1) > f32(0))
        sec_hit, _p2, n2 = _world_hit(spheres, p1, sd, f32(0.001), F32_MAX, rec_t)
        sec = scat_ok & sec_hit
        with np.errstate(invalid="ignore", divide="ignore"):
            norm = np.sqrt(_dot(n2, n2))
            c = [((n2[k] / norm) * f32(255.0)) / f32(2.0) for k in range(3)]
        fz = f32(fuzzy)
        c = [c[k] * (albedo[:, k] * fz) for k in range(3)]
        c = [f32(0.0) + c[k] for k in range(3)]                        # pixel_color += sampled_color
        colour = np.stack([_as_u8(c[0]), _as_u8(c[1]), _as_u8(c[2])], axis=1)
    black = prim & ~scat_ok
    starved = prim & scat_ok & ~sec & (spp >= 21)
    out[:, :3] = grad
    out[sec, :3] = colour[sec]
    out[black | starved, :3] = 0
    return out.reshape(len(ys), width, 4)
