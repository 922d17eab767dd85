"""The library's own JPEG decoder (csrc/mirt_jpeg.cpp; `Texture::new_from_image`, texture.rs:21-46) — no GPU involved.

Pins: (1) bit-identical RGB8 against Pillow (libjpeg-turbo: the IJG slow-integer IDCT, fixed-point YCbCr tables and
triangle upsampling this decoder restates) on generated files of every supported kind — baseline / progressive, 4:4:4 /
4:2:2 / 4:2:0, grey, optimised tables, restart intervals, sizes that are not multiples of the MCU; (2) the reference's two
assets, when the reference tree is present (it is in the build container, not on the GPU box), against the decodes
committed under weekend-raytracer-wgpu_amd/assets/ — those are what every texture test and bench.py --config 4 use.
Against the `image` crate the decode stays decoder-unpinned."""
import ctypes as C
import io
from pathlib import Path

import numpy as np
import pytest

import weekend_raytracer_wgpu_amd as m

Image = pytest.importorskip("PIL.Image")
REF_ASSETS = Path("/root/reference/assets")


def _gen(w, h, kind, seed=1):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return (rng.random((h, w, 3)) * 255).astype(np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([x * 255 / max(1, w - 1), y * 255 / max(1, h - 1), (x + y) * 255 / max(1, w + h - 2)], -1) + 20 * np.sin(x / 3.0)[..., None]
    return np.clip(a, 0, 255).astype(np.uint8)


def _encode(arr, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, "JPEG", **kw)
    return buf.getvalue()


def _pillow(b):
    return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (8, 8), (17, 33), (64, 48), (129, 65)])
@pytest.mark.parametrize("progressive", [False, True])
def test_bit_identical_to_libjpeg_turbo(size, progressive):
    w, h = size
    for kind in ("smooth", "noise"):
        for sub in (0, 1, 2):                              # 4:4:4, 4:2:2, 4:2:0
            for q in (30, 90):
                b = _encode(_gen(w, h, kind), quality=q, subsampling=sub, progressive=progressive)
                assert m.jpeg_info(b) == (w, h)
                assert np.array_equal(m.decode_jpeg(b), _pillow(b)), (size, kind, sub, q, progressive)


def test_grey_optimised_tables_and_restart_intervals():
    grey = _gen(75, 50, "smooth")[..., 0]
    for prog in (False, True):
        b = _encode(grey, quality=80, progressive=prog)
        assert np.array_equal(m.decode_jpeg(b), _pillow(b))
    rgb = _gen(90, 61, "noise", seed=5)
    for kw in (dict(optimize=True), dict(restart_marker_blocks=3), dict(restart_marker_rows=1, subsampling=2),
               dict(restart_marker_blocks=1, progressive=True), dict(quality=100, subsampling=0), dict(quality=1)):
        b = _encode(rgb, **kw)
        assert np.array_equal(m.decode_jpeg(b), _pillow(b)), kw


def test_texels_follow_texture_rs():
    """`inv_255 * (p as f32)` (texture.rs:28-41), not p / 255."""
    b = _encode(_gen(16, 8, "noise"), quality=90, subsampling=0)
    tex = m.Texture.new_from_jpeg_bytes(b)
    rgb = m.decode_jpeg(b)
    assert tex.dimensions() == (16, 8)
    want = (np.float32(1.0) / np.float32(255.0)) * rgb.astype(np.float32)
    assert tex.as_slice().dtype == np.float32 and np.array_equal(tex.as_slice().reshape(8, 16, 3), want)


def test_refused_files_report_an_error():
    with pytest.raises(m.MirtError) as e:
        m.decode_jpeg(b"\x89PNG\r\n\x1a\n" + bytes(32))
    assert e.value.status == m._abi.MIRT_ERR_IMAGE_DECODE and "SOI" in str(e.value)
    good = _encode(_gen(32, 32, "smooth"), quality=80)
    with pytest.raises(m.MirtError):
        m.decode_jpeg(good[: len(good) // 8])              # cut inside the tables
    cmyk = io.BytesIO()
    Image.fromarray(_gen(16, 16, "noise")).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(m.MirtError) as e:
        m.decode_jpeg(cmyk.getvalue())
    assert "component" in str(e.value)
    # a truncated scan decodes like libjpeg does: zeros are supplied after the end of the data
    big = _encode(_gen(160, 120, "noise"), quality=90)
    cut = big[: len(big) - len(big) // 3] + b"\xff\xd9"
    got = m.decode_jpeg(cut)
    assert got.shape == (120, 160, 3) and np.array_equal(got[:32], _pillow(big)[:32])       # the rows before the cut are intact


@pytest.mark.skipif(not REF_ASSETS.is_dir(), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("name,fixture", [("earthmap.jpeg", "earthmap_1024x512_rgb8.npz"), ("moon.jpeg", "moon_1024x512_rgb8.npz")])
def test_reference_assets_decode_to_the_committed_texel_fixtures(name, fixture):
    b = (REF_ASSETS / name).read_bytes()
    got = m.decode_jpeg(b)
    want = np.load(Path(m.__file__).parent / "assets" / fixture)["rgb8"]
    assert got.shape == want.shape == (512, 1024, 3)
    assert np.array_equal(got, want)
    assert np.array_equal(got, _pillow(b))
    # and the Texture the reference's Layer::scene would build from the file
    t = m.Texture.new_from_image(str(REF_ASSETS / name))
    assert t.dimensions() == (1024, 512)
    assert np.array_equal(t.as_slice(), m.Texture.new_from_rgb8(want).as_slice())


def test_a_header_that_claims_more_pixels_than_the_file_can_code_is_refused():
    """Untrusted input: the decoder allocates its coefficient planes from the SOF header, so a short file must not be able to
    claim a huge frame (16384 x 16384 from a few hundred bytes would be 1.5 GB).  The bound is 4096 pixels per byte of file;
    the header-only query still answers."""
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, format="JPEG", quality=50)
    data = bytearray(buf.getvalue())
    i = data.index(b"\xff\xc0")                      # SOF0: length(2) precision(1) height(2) width(2)
    data[i + 5:i + 9] = bytes([0x40, 0x00, 0x40, 0x00])     # 16384 x 16384
    assert m.jpeg_info(bytes(data)) == (16384, 16384)
    with pytest.raises(m.MirtError) as e:
        m.decode_jpeg(bytes(data))
    assert e.value.status_name == "MIRT_ERR_IMAGE_DECODE" and "too large" in str(e.value)
