#!/usr/bin/env python3
"""Derive the polynomial coefficients of the "mirt-math v1" elementary functions.

The path-traced mode needs sin/cos/acos/atan2/log2/exp2 whose results are bit-identical on the
host CPU and on gfx950.  libm and OCML differ in the last bits, so both sides evaluate the SAME
fixed polynomials with explicit fma Horner steps.  This script derives those polynomials
(Chebyshev interpolation of the reduced function in float64, converted to the monomial basis,
rounded to float32) and prints them as C initialisers.  The printed values are pasted into
oracle/mirt_oracle_math.h and weekend-raytracer-wgpu_amd/csrc/mirt_device_math.h; the unit test
tests/test_math_spec.py re-runs the derivation and checks that both files hold exactly these
values.

Run:  python tools/fit_poly.py
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import chebyshev as C
from numpy.polynomial import polynomial as P


def cheb_fit(fn, lo: float, hi: float, deg: int) -> np.ndarray:
    """Monomial coefficients (ascending) of the degree-`deg` Chebyshev interpolant of fn on [lo,hi]."""
    k = np.arange(deg + 1)
    nodes = np.cos(np.pi * (k + 0.5) / (deg + 1))           # Chebyshev nodes on [-1,1]
    x = 0.5 * (hi - lo) * nodes + 0.5 * (hi + lo)
    c = C.chebfit(nodes, fn(x), deg)
    mono_t = C.cheb2poly(c)                                    # in t = (2x-(hi+lo))/(hi-lo)
    a = 2.0 / (hi - lo)
    b = -(hi + lo) / (hi - lo)
    out = np.zeros(1)
    lin = np.array([b, a])
    for i, ci in enumerate(mono_t):                            # compose with the affine map
        out = P.polyadd(out, ci * P.polypow(lin, i))
    return out


def safe(fn_small, fn, eps):
    def g(x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(np.abs(x) < eps, fn_small(x), fn(np.where(np.abs(x) < eps, 1.0, x)))
    return g


def derive() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    q = (np.pi / 4) ** 2
    # sin(r) = r + r*z*S(z), z = r*r in [0,(pi/4)^2]
    out["SIN"] = cheb_fit(
        safe(lambda z: -1 / 6 + z / 120, lambda z: (np.sin(np.sqrt(z)) / np.sqrt(z) - 1) / z, 1e-6),
        0.0, q, 3)
    # cos(r) = 1 - z/2 + z*z*Cc(z)
    out["COS"] = cheb_fit(
        safe(lambda z: 1 / 24 - z / 720, lambda z: (np.cos(np.sqrt(z)) - 1 + z / 2) / (z * z), 1e-4),
        0.0, q, 3)
    # asin(x) = x + x*z*A(z), z = x*x in [0,0.25]
    out["ASIN"] = cheb_fit(
        safe(lambda z: 1 / 6 + 3 * z / 40, lambda z: (np.arcsin(np.sqrt(z)) / np.sqrt(z) - 1) / z, 1e-6),
        0.0, 0.25, 5)
    # atan(a) = a + a*z*T(z), z = a*a in [0,1]
    out["ATAN"] = cheb_fit(
        safe(lambda z: -1 / 3 + z / 5, lambda z: (np.arctan(np.sqrt(z)) / np.sqrt(z) - 1) / z, 1e-6),
        0.0, 1.0, 8)
    # log2(1+f) = f*L(f), f in [sqrt(1/2)-1, sqrt(2)-1]
    out["LOG2"] = cheb_fit(
        safe(lambda f: (1 - f / 2) / np.log(2), lambda f: np.log2(1 + f) / f, 1e-9),
        np.sqrt(0.5) - 1, np.sqrt(2.0) - 1, 9)
    # exp2(f) = 1 + f*E(f), f in [-0.5,0.5]
    out["EXP2"] = cheb_fit(
        safe(lambda f: np.log(2) + f * np.log(2) ** 2 / 2, lambda f: (np.exp2(f) - 1) / f, 1e-9),
        -0.5, 0.5, 6)
    return {k: v.astype(np.float32) for k, v in out.items()}


def c_literal(v: np.float32) -> str:
    return float(v).hex() + "f"


def main() -> None:
    for name, coeffs in derive().items():
        body = ", ".join(c_literal(c) for c in coeffs)
        print(f"/* {name}: {len(coeffs)} coefficients, ascending powers */")
        print(f"#define MIRT_POLY_{name} {{ {body} }}")
        print("/*   = " + ", ".join(repr(float(c)) for c in coeffs) + " */")


if __name__ == "__main__":
    main()
