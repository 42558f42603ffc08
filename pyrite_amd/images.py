"""Texture ingest for the project front-end: a PNG reader and the linearisation `Texture::from_path` performs
(pyrite/src/texture.rs:25-85, :174-295).

The reference decodes with the `image` crate and converts with `palette` 0.7.2; neither is vendored, so the conversions
are restated from their published definitions: the sRGB transfer function (IEC 61966-2-1) for non-`linear` textures,
`component / max` for linear ones and for alpha, Rec. 709 luma weights for colour -> mono. PNG is read in Python, baseline
JPEG through the small native reader `csrc/jpeg.c` (libpyrite_images.so, built by `pyrite_amd.build`); the container has no
image library. Textures may also be handed over as arrays (see `compiler.FlatScene.texture_id`)."""
import ctypes as C
import os
import struct
import zlib

import numpy as np

f32 = np.float32
IMAGES_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpyrite_images.so")
_images_lib = None


def read_jpeg(path):
    """-> uint8 array [height, width, 3] (grayscale files come back as three equal channels)."""
    global _images_lib
    if _images_lib is None:
        if not os.path.exists(IMAGES_LIB):
            raise OSError("%s is missing: run `python -m pyrite_amd.build`" % IMAGES_LIB)
        lib = C.CDLL(IMAGES_LIB)
        lib.pyr_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p,
                                        C.c_size_t]
        lib.pyr_image_free.argtypes = [C.POINTER(C.c_uint8)]
        _images_lib = lib
    with open(path, "rb") as f:
        data = f.read()
    w, h, rgb, err = C.c_int(), C.c_int(), C.POINTER(C.c_uint8)(), C.create_string_buffer(256)
    if _images_lib.pyr_jpeg_decode(data, len(data), C.byref(w), C.byref(h), C.byref(rgb), err, len(err)) != 0:
        raise ValueError("%s: %s" % (path, err.value.decode()))
    try:
        return np.ctypeslib.as_array(rgb, shape=(h.value, w.value, 3)).copy()
    finally:
        _images_lib.pyr_image_free(rgb)


def read_image(path):
    """The image formats project files use (image::ImageFormat::from_path, texture.rs:32-35): by extension."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".png":
        return read_png(path)
    if ext in (".jpg", ".jpeg"):
        return read_jpeg(path)
    raise ValueError("%s: unsupported image format (PNG and baseline JPEG are read)" % path)


def read_png(path):
    """-> uint8 or uint16 array [height, width, channels] with channels in (1: luma, 2: luma+alpha, 3: rgb, 4: rgba)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG file" % path)
    pos, chunks, idat, palette, trns = 8, None, [], None, None
    while pos < len(data):
        length, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            chunks = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, dtype=np.uint8)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    width, height, depth, color_type, _, _, interlace = chunks
    if interlace:
        raise ValueError("%s: interlaced PNGs are not supported" % path)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    if depth not in (8, 16) and not (color_type in (0, 3) and depth in (1, 2, 4)):
        raise ValueError("%s: unsupported bit depth %d" % (path, depth))
    bits_per_pixel = channels * depth
    bpp = max(1, bits_per_pixel // 8)
    stride = (width * bits_per_pixel + 7) // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(height):
        ftype = int(raw[y * (stride + 1)])
        line = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)].astype(np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 0xFF
        else:
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                else:  # Paeth
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 0xFF
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(height, width, channels, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    elif depth == 8:
        img = out.reshape(height, width, channels)
    else:  # packed 1/2/4-bit gray or palette indices
        bits = np.unpackbits(out, axis=1)[:, :width * depth].reshape(height, width, depth)
        img = np.zeros((height, width), dtype=np.uint8)
        for k in range(depth):
            img = (img << 1) | bits[..., k]
        if color_type == 0:
            img = (img.astype(np.uint16) * 255 // ((1 << depth) - 1)).astype(np.uint8)
        img = img[..., None]
    if color_type == 3:
        rgb = palette[img[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, dtype=np.uint8)
            alpha[:len(trns)] = trns
            img = np.concatenate([rgb, alpha[img[..., 0]][..., None]], axis=-1)
        else:
            img = rgb
    return np.ascontiguousarray(img)


def srgb_to_linear(c):
    """IEC 61966-2-1 decoding of components in [0, 1], evaluated in f64 and rounded to f32."""
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(f32)


LUMA_WEIGHTS = np.array([0.2126729, 0.7151522, 0.0721750], dtype=f32)  # the Y row of the sRGB (D65) RGB -> XYZ matrix


def linearise(image, linear, mono):
    """Texture::from_path's conversion (texture.rs:37-78 + convert_pixels :174-199): integer or float image
    [h, w, c] -> float32 texels, [h, w, 4] (LinSrgba) or [h, w] (LinLuma)."""
    image = np.asarray(image)
    if image.ndim == 2:
        image = image[..., None]
    if image.dtype == np.uint8:
        unit = image.astype(f32) / f32(255.0)
    elif image.dtype == np.uint16:
        unit = image.astype(f32) / f32(65535.0)
    else:
        unit = image.astype(f32)
    channels = unit.shape[-1]
    has_alpha = channels in (2, 4)
    color = unit[..., :channels - 1] if has_alpha else unit
    alpha = unit[..., -1] if has_alpha else np.ones(unit.shape[:2], dtype=f32)
    if not linear:
        color = srgb_to_linear(color)
    if mono:
        if color.shape[-1] == 1:
            return np.ascontiguousarray(color[..., 0], dtype=f32)
        luma = color[..., 0] * LUMA_WEIGHTS[0] + color[..., 1] * LUMA_WEIGHTS[1] + color[..., 2] * LUMA_WEIGHTS[2]
        return np.ascontiguousarray(luma, dtype=f32)
    if color.shape[-1] == 1:
        color = np.repeat(color, 3, axis=-1)
    return np.ascontiguousarray(np.concatenate([color, alpha[..., None]], axis=-1), dtype=f32)


def load_texture(path, linear, mono):
    return linearise(read_image(path), linear, mono)
