#!/usr/bin/env python3
"""Turn the rendered example images the reference keeps next to its test projects into small numeric fixtures.

    python tests/golden/make_reference_fixtures.py        (needs /root/reference; the fixtures are committed)

pyrite/test/spheres/hq_example.png and pyrite/test/diamonds/hq_example.png were rendered by the reference itself with the
`simple` renderer (spheres.lua: 512x256, 600 spp; diamonds.lua: 512x300, 200 spp, 256 bounces, thin lens, dispersion).
They are the only outputs of the reference that exist for this path. Each is reduced to linear-light RGB block means
(8 x 8 pixels, sRGB decoded) -- data, a few KB -- against which tests/test_reference_images.py holds the oracle with a
loose tolerance: the images predate the current film-development code (their colour rendition differs slightly from what
main.rs:315-418 produces today), so only luminance structure and level are compared."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pyrite_amd import images  # noqa: E402

REFERENCE = "/root/reference/pyrite/test"
BLOCK = 8


def block_means(rgb8, block=BLOCK):
    lin = images.srgb_to_linear(rgb8[..., :3].astype(np.float64) / 255.0).astype(np.float64)
    h, w = lin.shape[0] // block * block, lin.shape[1] // block * block
    return lin[:h, :w].reshape(h // block, block, w // block, block, 3).mean((1, 3)).astype(np.float32)


def decode_jpeg(path):
    return images.read_jpeg(path)


def shrink(img, factor):
    """Box filter on the 8-bit values (what an image editor's resize does), rounded back to 8 bits."""
    h, w = img.shape[0] // factor * factor, img.shape[1] // factor * factor
    small = img[:h, :w].reshape(h // factor, factor, w // factor, factor, 3).astype(np.float64).mean((1, 3))
    return np.clip(np.floor(small + 0.5), 0, 255).astype(np.uint8)


def texture_fixtures():
    """test/textures: the texture images shrunk to a few tens of KB each (the render is compared in 8 x 8 pixel blocks, far
    coarser than the texture detail that is lost) and the two small meshes re-emitted by this script's own writer. The
    `fabric` textures of the cube are not in the reference checkout (.MISSING_LARGE_BLOBS)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from extract_reference_data import reemit_obj
    from pyrite_amd import develop

    out_dir = os.path.join(HERE, "textures")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(REFERENCE, "textures")
    for rel, factor in (("color_checker.jpg", 4), ("tiles/color.jpg", 16), ("tiles/normal.jpg", 16), ("tactile_paving/color.jpg", 16),
                        ("tactile_paving/normal.jpg", 16)):
        img = shrink(decode_jpeg(os.path.join(src, rel)), factor)
        name = rel.replace("/", "_").replace(".jpg", ".png")
        develop.save_png(os.path.join(out_dir, name), img)
        print(rel, "->", name, img.shape, os.path.getsize(os.path.join(out_dir, name)), "bytes")
    for name in ("color_checker.obj", "cube.obj"):
        reemit_obj(os.path.join(src, name), os.path.join(out_dir, name), "test/textures " + name)


def main():
    out = {}
    texture_fixtures()
    for name in ("spheres", "diamonds", "textures"):
        img = images.read_png(os.path.join(REFERENCE, name, "hq_example.png"))
        out[name] = block_means(img)
        out[name + "_size"] = np.array(img.shape[:2][::-1])
        if name != "textures":  # round 4: the 8-bit values themselves, so that the transfer function that wrote them can be fitted
            out[name + "_u8"] = np.ascontiguousarray(img[..., :3], dtype=np.uint8)  # (tests/reference_pin.py); expected outputs, data
        print(name, img.shape, "->", out[name].shape)
    np.savez_compressed(os.path.join(HERE, "reference_example_images.npz"), block=BLOCK, **out)


if __name__ == "__main__":
    main()
