"""The committed golden images (tests/golden/images_v1.npz, written by tools/make_goldens.py) pin
the oracle's output: an edit that changes what the oracle computes fails here."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import make_goldens  # noqa: E402
from helpers import GOLDEN, assert_images_equal  # noqa: E402

GOLD = np.load(GOLDEN / "images_v1.npz")


def test_golden_file_lists_every_case():
    assert set(GOLD.files) == set(make_goldens.CASES)


@pytest.mark.parametrize("name", sorted(make_goldens.CASES))
def test_oracle_reproduces_golden(name, oracle):
    sd, p = make_goldens.case_inputs(name)
    assert_images_equal(oracle.render(sd, p), GOLD[name], name)


def test_texture_fixture_hashes():
    """sha256 of the decoded RGB8 texels (decoder-unpinned, see tools/make_texture_fixtures.py)."""
    import hashlib
    import weekend_raytracer_wgpu_amd as m
    earth = np.load(m.asset_path("assets/earthmap.jpeg"))["rgb8"]
    moon = np.load(m.asset_path("assets/moon.jpeg"))["rgb8"]
    assert earth.shape == moon.shape == (512, 1024, 3)
    assert earth[0, 0].tolist() == [255, 255, 255]
    assert hashlib.sha256(earth.tobytes()).hexdigest().startswith("a8cdc92a168d554d")
    assert hashlib.sha256(moon.tobytes()).hexdigest().startswith("e68208cf243e7858")
