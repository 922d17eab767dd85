"""mirt-math v1: the polynomial tables in the oracle header and in the device header are exactly
what tools/fit_poly.py derives, and the oracle's elementary functions are accurate."""
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import fit_poly  # noqa: E402

ORACLE_H = (ROOT / "oracle" / "mirt_oracle_math.h").read_text()
DEVICE_H = (ROOT / "weekend-raytracer-wgpu_amd" / "csrc" / "mirt_device_math.h").read_text()


def _table(text: str, pattern: str):
    body = re.search(pattern + r"\s*=\s*\{([^}]*)\}", text).group(1)
    return [float.fromhex(tok.strip().rstrip("f")) for tok in body.split(",")]


def test_coefficient_tables_match_the_derivation():
    derived = fit_poly.derive()
    names = {"SIN": ("OM_SIN\\[4\\]", "kSin\\[4\\]"), "COS": ("OM_COS\\[4\\]", "kCos\\[4\\]"),
             "ASIN": ("OM_ASIN\\[6\\]", "kAsin\\[6\\]"), "ATAN": ("OM_ATAN\\[9\\]", "kAtan\\[9\\]"),
             "LOG2": ("OM_LOG2\\[10\\]", "kLog2\\[10\\]"), "EXP2": ("OM_EXP2\\[7\\]", "kExp2\\[7\\]")}
    for key, (o_pat, d_pat) in names.items():
        want = [float(np.float32(c)) for c in derived[key]]
        assert _table(ORACLE_H, o_pat) == want, key
        assert _table(DEVICE_H, d_pat) == want, key


def test_sincos_accuracy(oracle):
    x = np.linspace(-30.0, 30.0, 400001).astype(np.float32)
    s, c = oracle.sincos(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2.5e-7
    big = np.array([2500.0, -2500.0, 12345.678, 1.0e5], np.float32)
    s, c = oracle.sincos(big)
    assert np.abs(s - np.sin(big.astype(np.float64))).max() < 1e-3      # reduction degrades gracefully
    s, c = oracle.sincos(np.array([np.nan, np.inf, 1.0e7], np.float32))  # outside the domain -> sin 0, cos 1
    assert (s == 0).all() and (c == 1).all()


def test_acos_atan2_accuracy(oracle):
    x = np.linspace(-1.0, 1.0, 200001).astype(np.float32)
    assert np.abs(oracle.acos(x) - np.arccos(x.astype(np.float64))).max() < 6e-7
    assert oracle.acos(np.array([1.5, -1.5, np.nan], np.float32)).tolist() == [0.0, float(np.float32(3.1415927)), 0.0]
    rng = np.random.default_rng(1)
    y, xx = rng.normal(size=200000).astype(np.float32), rng.normal(size=200000).astype(np.float32)
    assert np.abs(oracle.atan2(y, xx) - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 6e-7
    z = np.zeros(1, np.float32)
    assert oracle.atan2(z, z)[0] == 0.0
    assert abs(oracle.atan2(z, -np.ones(1, np.float32))[0] - np.pi) < 1e-6
    assert abs(oracle.atan2(np.ones(1, np.float32), z)[0] - np.pi / 2) < 1e-6


def test_log2_exp2_pow_accuracy(oracle):
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(-80, 80, 200000)).astype(np.float32)
    ref = np.log2(x.astype(np.float64))
    assert (np.abs(oracle.log2(x) - ref) / np.maximum(1.0, np.abs(ref))).max() < 2e-7
    y = rng.uniform(-125, 125, 200000).astype(np.float32)
    assert np.abs(oracle.exp2(y) / np.exp2(y.astype(np.float64)) - 1).max() < 3e-7
    assert oracle.exp2(np.array([-200.0, 130.0], np.float32)).tolist() == [0.0, float("inf")]
    xi = rng.uniform(0, 1, 200000).astype(np.float32)
    e = np.float32(0.33333)
    got = oracle.pow_pos(xi, np.full_like(xi, e))
    assert np.abs(got / np.power(xi.astype(np.float64), float(e)) - 1).max() < 1e-6
    assert oracle.pow_pos(np.zeros(1, np.float32), np.full(1, e, np.float32))[0] == 0.0
