"""The step after the path: `canvas.to_image().save("image.png")` of examples/*.rs.
include/raytracer_amd/png.hpp writes 8-bit RGB PNGs with zlib only; decoded here by hand (no image
library in this environment): signature, IHDR, IDAT inflate, filter type 0 rows."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import helpers as H
from raytracer_amd import RtOptions, scenes_lib


def decode_png(data: bytes) -> np.ndarray:
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) & 0xFFFFFFFF == crc, typ
        if typ == b"IHDR":
            w, h, depth, ctype, comp, filt, lace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, comp, filt, lace) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()  # filter type 0 on every row
    return raw[:, 1:].reshape(h, w, 3)


def encode(rgb: np.ndarray) -> bytes:
    S = scenes_lib()
    h, w, _ = rgb.shape
    n = C.c_size_t()
    assert S.rtx_png_encode(rgb.ctypes.data, w, h, None, 0, C.byref(n)) == 0
    out = np.zeros(n.value, dtype=np.uint8)
    assert S.rtx_png_encode(rgb.ctypes.data, w, h, out.ctypes.data, out.size, C.byref(n)) == 0
    return out.tobytes()


@pytest.mark.parametrize("shape", [(1, 1), (7, 13), (54, 96)])
def test_png_encoder_round_trip(shape):
    rgb = np.random.default_rng(1).integers(0, 256, size=(*shape, 3), dtype=np.uint8)
    assert np.array_equal(decode_png(encode(rgb)), rgb)


def test_png_encoder_rejects_empty_images():
    S = scenes_lib()
    n = C.c_size_t()
    assert S.rtx_png_encode(None, 4, 4, None, 0, C.byref(n)) != 0
    px = np.zeros(3, dtype=np.uint8)
    assert S.rtx_png_encode(px.ctypes.data, 0, 1, None, 0, C.byref(n)) != 0


@pytest.mark.gpu
def test_rendered_png_is_within_one_lsb_of_the_reference_bytes(tmp_path):
    """North star: per-channel f32 within 1e-4 '(<= 1 LSB in the PNG)'."""
    S = scenes_lib()
    for name, w, h in (("test_world", 11, 11), ("cover_image", 192, 108)):
        path = str(tmp_path / f"{name}.png")
        assert S.rtx_render_example_png(name.encode(), w, h, None, 0, path.encode()) == 0
        img = decode_png(open(path, "rb").read())
        ref, _, _ = H.oracle_render(name, w, h, threads=8)
        want = H.oracle_quantize(ref)
        d = np.abs(img.astype(int) - want.astype(int))
        assert img.shape == (h, w, 3) and (d > 1).mean() <= 1e-3 and (d > 0).mean() <= 2e-2
    assert tuple(img[0, 0]) == tuple(want[0, 0])
