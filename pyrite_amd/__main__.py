"""`python -m pyrite_amd project.lua` -- what `pyrite project.lua` does for a project with the `simple` renderer
(pyrite/src/main.rs:46-330): load the project file, render it on the GPU, develop the film with the project's `image.filter`
/ `image.white`, and write `render.png` next to the project file (main.rs:180-184).

    python -m pyrite_amd path/to/project.lua [-o out.png] [--seed N] [--device D] [--spp N] [--size WxH]"""
import argparse
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m pyrite_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("project", help="project file (*.lua)")
    ap.add_argument("-o", "--output", default=None, help="image to write (default: render.png next to the project file)")
    ap.add_argument("--seed", type=int, default=None, help="RNG seed (default: from the clock; the reference seeds from OS entropy)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--spp", type=int, default=None, help="override renderer.pixel_samples")
    ap.add_argument("--size", default=None, help="override image size, WIDTHxHEIGHT")
    args = ap.parse_args(argv)

    from . import lua_project, scenes
    from .develop import develop, save_png

    project, base_dir = lua_project.load_project(args.project)
    if args.size:
        w, h = (int(x) for x in args.size.lower().split("x"))
        project.setdefault("image", {}).update(width=w, height=h)
    if args.spp:
        project["renderer"] = project["renderer"].with_(pixel_samples=args.spp)
    seed = args.seed if args.seed is not None else int(time.time_ns() & 0x7FFFFFFFFFFFFFFF)
    world, cam, r, film = scenes.build(project, seed=seed, base_dir=base_dir)
    print("The scene contains %d objects." % (len(world.flat.tri_material) + len(world.flat.spheres) + len(world.flat.planes)))  # world.rs:251-254

    def on_status(percent, message):
        print("\r%s... %3d %%" % (message, percent), end="", flush=True)

    t = time.time()
    r.render(film, cam, world, on_status=on_status, device=args.device)
    print("\rRendering... done in %.2f s (%.1f Msamples/s)" % (time.time() - t, film.width * film.height * r.pixel_samples / (time.time() - t) / 1e6))
    print("Saving final result...")  # main.rs:313
    image = project.get("image") or {}
    rgb = develop(film, filter=image.get("filter"), white=image.get("white"), device=args.device)
    out = args.output or os.path.join(base_dir, "render.png")
    save_png(out, rgb)
    print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
