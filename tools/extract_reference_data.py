#!/usr/bin/env python3
"""Re-encode the DATA the reference ships (never its code) into this repo's own fixture formats.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):

    python tools/extract_reference_data.py

Inputs  -> outputs (all under pyrite_amd/data/):
  pyrite/data/d65.csv, a.csv            -> tables.npz: d65, a            (A divided by 100: build.rs:160-161)
  pyrite/data/srgb_cie1931.csv          -> tables.npz: rgb_basis [471,3] (build.rs:18-59)
  pyrite/data/ciexyz65_1.csv            -> tables.npz: xyz [471,3], xyz_min, xyz_max (build.rs:68-121)
  pyrite/test/cornell/colors.lua        -> cornell_spectra.json: white / green / red arrays (numbers only)
  pyrite/test/cornell/lamp.lua          -> cornell_spectra.json: lamp array (numbers only)
  pyrite/test/cornell/box.obj           -> cornell_box.obj    (re-emitted by this script's own writer)
  pyrite/test/diamonds/diamonds.obj     -> diamonds.obj       (re-emitted by this script's own writer)
"""
import csv
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/pyrite"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyrite_amd", "data")


def read_csv(path):
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    header, body = rows[0], rows[1:]
    return header, np.array([[float(x) for x in r] for r in body if r], dtype=np.float64)


def lua_number_array(text, name):
    """Numbers of `name = spectrum { format="array", min=.., max=.., points={...} }`."""
    m = re.search(name + r"\s*=\s*spectrum\s*\{(.*?)\n\s{4}\},?\n", text, re.S)
    if not m:
        raise SystemExit("spectrum %s not found" % name)
    body = m.group(1)
    mn = float(re.search(r"min\s*=\s*([-0-9.eE]+)", body).group(1))
    mx = float(re.search(r"max\s*=\s*([-0-9.eE]+)", body).group(1))
    pts = re.search(r"points\s*=\s*\{(.*?)\}", body, re.S).group(1)
    values = [float(x) for x in re.findall(r"[-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?", pts)]
    return {"format": "array", "min": mn, "max": mx, "points": values}


def reemit_obj(src, dst, title):
    """Parse v / vt / vn / o / f records and write them back in a normalised layout."""
    out = ["# %s -- geometry data re-encoded by tools/extract_reference_data.py" % title]
    with open(src) as f:
        for line in f:
            parts = line.split()
            if not parts or parts[0] not in ("v", "vt", "vn", "o", "g", "f"):
                continue
            if parts[0] in ("v", "vn", "vt"):
                out.append(parts[0] + " " + " ".join(repr(float(x)) for x in parts[1:]))
            else:
                out.append(" ".join(parts))
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are already committed")
    os.makedirs(OUT, exist_ok=True)

    _, d65 = read_csv(os.path.join(REF, "data/d65.csv"))
    _, a = read_csv(os.path.join(REF, "data/a.csv"))
    _, rgb = read_csv(os.path.join(REF, "data/srgb_cie1931.csv"))
    _, xyz = read_csv(os.path.join(REF, "data/ciexyz65_1.csv"))
    assert np.array_equal(d65[:, 0], a[:, 0])
    np.savez(
        os.path.join(OUT, "tables.npz"),
        light_min=np.float32(min(d65[:, 0].min(), a[:, 0].min())),
        light_max=np.float32(max(d65[:, 0].max(), a[:, 0].max())),
        d65=d65[:, 1].astype(np.float32),
        # build.rs:160: `intensity / 100.0` is evaluated in f32
        a=(a[:, 1].astype(np.float32) / np.float32(100.0)).astype(np.float32),
        rgb_basis=rgb.astype(np.float32),
        rgb_min=np.float32(360.0),
        rgb_max=np.float32(360.0 + rgb.shape[0]),  # build.rs:37-38 (sic: one past the data)
        xyz=xyz[:, 1:4].astype(np.float32),
        xyz_min=np.float32(xyz[:, 0].min()),
        xyz_max=np.float32(xyz[:, 0].max()),
    )

    colors = open(os.path.join(REF, "test/cornell/colors.lua")).read()
    lamp = open(os.path.join(REF, "test/cornell/lamp.lua")).read()
    spectra = {name: lua_number_array(colors, name) for name in ("white", "green", "red")}
    spectra["lamp"] = lua_number_array(lamp, "color")
    with open(os.path.join(OUT, "cornell_spectra.json"), "w") as f:
        json.dump(spectra, f)

    reemit_obj(os.path.join(REF, "test/cornell/box.obj"), os.path.join(OUT, "cornell_box.obj"), "Cornell box")
    reemit_obj(os.path.join(REF, "test/diamonds/diamonds.obj"), os.path.join(OUT, "diamonds.obj"), "diamonds scene")
    for k, v in spectra.items():
        print(k, v["min"], v["max"], len(v["points"]))


if __name__ == "__main__":
    main()
