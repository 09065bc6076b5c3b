"""`Texture::from_file` (texture.rs:27-62) for the host side of the boundary: the app's sky image
(`Texture::from_file("assets/textures/default.exr")`, app.rs:121) and any other file texture, as the (h, w, 3) f32
linear-RGB array that `SceneBuilder.set_sky_image` / `add_texture` take.

What the reference does, restated:
  * extension `hdr` / `exr` (case-insensitive): `image::open(path)?.into_rgb32f()` taken as linear RGB;
  * anything else: the same call (8-bit -> v/255, 16-bit -> v/65535, grey replicated), then `srgb_to_linear`
    per channel (texture.rs:14-20; `powf` is glibc's, as in gltf.py's table).
Decoding lives in the `image` crate (a Cargo dependency, not under /root/reference), so its published behaviour is
restated and this loader is "parity unpinned": Radiance RGBE pixels become `c * 2^(e - 136)` with `e == 0 -> 0`
(`Rgbe8Pixel::to_hdr`), which is exact in f32.  OpenEXR needs a decoder this image does not have: `.exr` raises
`TextureError` naming the way out (convert the file to `.hdr`); nothing is guessed.
"""
import os

import numpy as np

from .gltf import _srgb_lut


class TextureError(ValueError):
    pass


def _read_hdr(path):
    data = open(path, "rb").read()
    end = data.find(b"\n\n")
    if end < 0 or not (data.startswith(b"#?RADIANCE") or data.startswith(b"#?RGBE")):
        raise TextureError(f"{path}: not a Radiance HDR file")
    header = data[:end].decode("latin-1").split("\n")
    if not any(line.strip() == "FORMAT=32-bit_rle_rgbe" for line in header):
        raise TextureError(f"{path}: only FORMAT=32-bit_rle_rgbe is supported")
    pos = end + 2
    eol = data.find(b"\n", pos)
    dims = data[pos:eol].decode("latin-1").split()
    if len(dims) != 4 or dims[0] != "-Y" or dims[2] != "+X":
        raise TextureError(f"{path}: resolution line {' '.join(dims)!r}: only '-Y H +X W' is supported")
    h, w = int(dims[1]), int(dims[3])
    pos = eol + 1
    out = np.empty((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w <= 0x7FFF and data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] & 0x80) == 0:
            if (data[pos + 2] << 8 | data[pos + 3]) != w:
                raise TextureError(f"{path}: scanline {y} length mismatch")
            pos += 4
            for c in range(4):                     # the four components are run-length coded one after the other
                x = 0
                while x < w:
                    n = data[pos]
                    if n > 128:                    # a run of n - 128 copies of the next byte
                        n -= 128
                        if n == 0 or x + n > w:
                            raise TextureError(f"{path}: bad run in scanline {y}")
                        out[y, x:x + n, c] = data[pos + 1]
                        pos += 2
                    else:                          # n literal bytes
                        if n == 0 or x + n > w:
                            raise TextureError(f"{path}: bad literal count in scanline {y}")
                        out[y, x:x + n, c] = np.frombuffer(data, dtype=np.uint8, count=n, offset=pos + 1)
                        pos += 1 + n
                    x += n
        else:                                      # flat RGBE quadruples (old-style runs are not produced by any current writer)
            row = np.frombuffer(data, dtype=np.uint8, count=4 * w, offset=pos).reshape(w, 4)
            if w and ((row[:, 0] == 1) & (row[:, 1] == 1) & (row[:, 2] == 1)).any():
                raise TextureError(f"{path}: old-style run-length scanlines are not supported")
            out[y] = row
            pos += 4 * w
    e = out[..., 3].astype(np.int32)
    scale = np.where(e == 0, np.float32(0.0), np.exp2((e - 136).astype(np.float32))).astype(np.float32)
    return (out[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


def _read_ldr(path):
    from PIL import Image
    im = Image.open(path)
    im.load()
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):          # 16-bit grey: v / 65535, replicated
        g = np.asarray(im).astype(np.float32) / np.float32(65535.0)
        c = np.repeat(g[..., None], 3, axis=2)
        lut16 = None
    else:
        if im.mode != "RGB":
            im = im.convert("RGB")                          # palettes, grey, alpha: what into_rgb32f keeps of them
        px = np.asarray(im, dtype=np.uint8)
        return _srgb_lut()[px]                              # srgb_to_linear(v / 255) for the 256 possible v
    # srgb_to_linear on arbitrary f32 (only reached by 16-bit sources): same formula, same powf
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    vals, inv = np.unique(c, return_inverse=True)
    lin = np.array([v / np.float32(12.92) if v <= np.float32(0.04045)
                    else np.float32(libm.powf(float((v + np.float32(0.055)) / np.float32(1.055)), 2.4)) for v in vals], dtype=np.float32)
    return lin[inv].reshape(c.shape)


def texture_from_file(path):
    """(h, w, 3) float32 linear RGB, row 0 at the top — `Texture::from_file`'s `data`, `width`, `height`."""
    ext = os.path.splitext(path)[1].lower().lstrip(".")
    if ext == "hdr":
        return _read_hdr(path)
    if ext == "exr":
        raise TextureError(f"{path}: OpenEXR decoding is not available in this environment (no EXR library); "
                           "convert the file to Radiance .hdr, which the reference treats identically (texture.rs:29-35)")
    return _read_ldr(path)
