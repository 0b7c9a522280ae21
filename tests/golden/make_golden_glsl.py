"""Frames computed by the reference's GLSL shader TEXT compiled as C++ under its vendored glm (oracle/glsl_driver.cpp,
`make -C oracle glsl`: needs /root/reference) -> tests/golden/glsl_images_small.npz.

kind "first":   the live shader, 453-skeleton/RayTracerBVH.cpp:221-355 (first accepted leaf in LIFO order, 512-pop cap)
kind "closest": the earlier shader kept block-commented in the same file, :46-166 (closest hit, no cap)

The vectors are DATA (pixels); the shader text itself never enters the repo.  Run from the repo root:
    make -C oracle glsl && python tests/golden/make_golden_glsl.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc  # noqa: E402

# scene -> (W, H, cameras (theta, phi, radius, target or None, fov))
CASES = {
    "sphere32": (96, 64, [(0.5, 0.7, 1.8, None, 45.0), (2.0, 0.3, 0.2, None, 45.0), (0.0, 1.5707964, 2.0, None, 45.0)]),
    "sphere64": (128, 80, [(0.5, 0.7, 1.8, None, 45.0), (1.0, 0.9, 0.25, None, 70.0)]),
    "odd": (96, 64, [(0.4, 0.9, 9.0, None, 45.0), (2.2, 0.4, 2.0, None, 60.0)]),
    "calgary": (128, 72, [(0.6, 0.5, 3500.0, None, 45.0), (1.2, 0.1, 900.0, None, 60.0)]),
}


def scene(name):
    if name.startswith("sphere"):
        g = orc.test_sphere_grid(int(name[6:]))
    elif name == "calgary":
        z = np.load(os.path.join(HERE, "ref_scene_cache.npz"))
        dims = tuple(int(x) for x in z["dims"])
        data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
        g = orc.Grid(dims, z["min"].astype(np.float32), np.float32(z["voxel"]), data)
    else:
        z = np.load(os.path.join(HERE, "ref_octrees_small.npz"))
        d = z["odd_grid"]
        g = orc.Grid((d.shape[2], d.shape[1], d.shape[0]), z["odd_min"], np.float32(z["odd_voxel"]), d)
    return g, orc.build_flat_octree(g)


def camera(spec):
    t, p, r, tgt, fov = spec
    cam = orc.Camera(t, p, r)
    if tgt is not None:
        cam.set_target(*[float(x) for x in tgt])
    return cam.get_view(), cam.get_pos(), fov


def main():
    if not orc.glsl_available():
        raise SystemExit("oracle/_ref/libglsl_*.so missing: run `make -C oracle glsl` (needs /root/reference)")
    out = {}
    for name, (W, H, cams) in CASES.items():
        g, nodes = scene(name)
        for i, spec in enumerate(cams):
            view, pos, fov = camera(spec)
            for kind in ("first", "closest"):
                img = orc.glsl_render(kind, nodes, g.min, g.voxel_size, view, pos, W / H, fov, W, H)
                out[f"{name}_cam{i}_{kind}"] = img
                print(name, i, kind, "hit pixels", int((img[..., 0] != 0).sum()))
    np.savez_compressed(os.path.join(HERE, "glsl_images_small.npz"), **out)
    print("wrote glsl_images_small.npz:", len(out), "frames")


if __name__ == "__main__":
    main()
