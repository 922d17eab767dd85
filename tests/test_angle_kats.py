"""K9: the reference's only tests — the five `Angle` unit tests of src/raytracer/angle.rs:52-93 —
re-expressed against the host mirror (libmirt host arithmetic) and against the oracle."""
import numpy as np

import weekend_raytracer_wgpu_amd as m

FRAC_PI_2 = float(np.float32(np.pi / 2))


def test_angle_to_radians():            # angle.rs:57-60
    assert m.Angle.degrees(90.0).as_radians() == FRAC_PI_2


def test_angle_to_degrees():            # angle.rs:63-66
    assert m.Angle.radians(FRAC_PI_2).as_degrees() == 90.0


def test_angle_add():                   # angle.rs:69-74
    assert (m.Angle.degrees(90.0) + m.Angle.degrees(90.0)).as_degrees() == 180.0


def test_angle_clamp_max():             # angle.rs:77-83
    r = m.Angle.degrees(90.0).clamp(m.Angle.degrees(0.0), m.Angle.degrees(45.0))
    assert r.as_degrees() == 45.0


def test_angle_clamp_min():             # angle.rs:86-92
    r = m.Angle.degrees(-90.0).clamp(m.Angle.degrees(0.0), m.Angle.degrees(45.0))
    assert r.as_degrees() == 0.0


def test_oracle_angle_agrees(oracle):
    for d in (0.0, 25.0, -10.0, 30.0, 45.0, 60.0, 90.0, 180.0, 33.3):
        assert oracle.LIB.mirt_oracle_degrees_to_radians(d) == m.Angle.degrees(d).as_radians()
        r = m.Angle.degrees(d).as_radians()
        assert oracle.LIB.mirt_oracle_radians_to_degrees(r) == m.Angle.radians(r).as_degrees()
    assert oracle.LIB.mirt_oracle_degrees_to_radians(90.0) == FRAC_PI_2
