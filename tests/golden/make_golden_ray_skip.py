"""Generates tests/golden/ref_ray_skip.npz.  Run ONLY where /root/reference exists:

    make -C oracle ref refvr && python tests/golden/make_golden_ray_skip.py

ref_* vectors: outputs of the REFERENCE'S OWN compiled octreeRaySkip (453-skeleton/VolumeRaycastRenderer.cpp:50-155,
reached through oracle/ref_shim_vr.cpp, which #includes that file to get at the static function) on the reference's own
createOctreeFromVoxelGrid trees.  For each scene: the 7x7 probe rays of drawRaycast (:1602-1630, computed with the
reference's glm calls), random directions from outside and inside the grid, axis-parallel and almost-axis-parallel
directions (the |d| < 1e-10 clamp, :83-87), a narrowed [tMin, tMax], and random visibility maps (:64-67).
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402


def scene(name):
    if name.startswith("sphere"):
        g = orc.test_sphere_grid(int(name[6:]))
        cam = (0.5, 0.7, 1.8)
    elif name == "odd":
        z = np.load(os.path.join(HERE, "ref_octrees_small.npz"))
        d = z["odd_grid"]
        g = orc.Grid((d.shape[2], d.shape[1], d.shape[0]), z["odd_min"], np.float32(z["odd_voxel"]), d)
        cam = (0.4, 0.9, 9.0)
    else:
        z = np.load(os.path.join(HERE, "ref_scene_cache.npz"))
        dims = tuple(int(x) for x in z["dims"])
        data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
        g = orc.Grid(dims, z["min"].astype(np.float32), np.float32(z["voxel"]), data)
        cam = (0.6, 0.5, 3500.0)
    return g, cam


def main():
    assert orc.ref_available() and orc.refvr_available(), "make -C oracle ref refvr first"
    out = {}
    rng = np.random.default_rng(20251004)
    for name in ("sphere32", "odd", "calgary"):
        g, (theta, phi, radius) = scene(name)
        view, eye, _ = orc.ref_camera(theta, phi, radius)
        nodes = orc.ref_build_flat_octree(g)
        ext = np.float32(max(g.dims)) * g.voxel_size
        centre = g.min + 0.5 * np.array(g.dims, np.float32) * g.voxel_size
        cases = []
        # (1) the probe grid of drawRaycast, as the reference computes it
        cases.append(("probe", eye, orc.ref_probe_rays(view, eye, np.float32(1920 / 1080).item()), 0.0, 1e30))
        # (2) random directions from the camera, towards the grid
        tgt = centre + (rng.random((192, 3)).astype(np.float32) - 0.5) * ext
        d = tgt - eye
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        cases.append(("random", eye, d.astype(np.float32), 0.0, 1e30))
        # (3) from inside the grid, any direction
        inside = (centre + (rng.random(3).astype(np.float32) - 0.5) * 0.3 * ext).astype(np.float32)
        d = rng.normal(size=(128, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        cases.append(("inside", inside, d, 0.0, 1e30))
        # (4) axis-parallel and almost axis-parallel directions: exact zeros, +-1e-11, +-1e-9 (either side of the 1e-10 clamp), -0.0
        ax = []
        for a in range(3):
            for sgn in (1.0, -1.0):
                for eps in (0.0, -0.0, 1e-11, -1e-11, 1e-9, -1e-9):
                    v = np.full(3, eps, np.float32)
                    v[a] = sgn
                    ax.append(v)
        o_ax = (centre + np.array([0.37, -0.21, 0.11], np.float32) * g.voxel_size - np.float32(1.3) * ext * np.array([1, 1, 1], np.float32) * 0).astype(np.float32)
        cases.append(("axis_inside", o_ax, np.array(ax, np.float32), 0.0, 1e30))
        for a in range(3):      # from outside, along each axis through the grid
            o = centre.copy(); o[a] -= np.float32(1.5) * ext
            cases.append((f"axis_outside{a}", o.astype(np.float32), np.array(ax, np.float32), 0.0, 1e30))
        # (5) a narrowed parameter interval
        cases.append(("narrow", eye, cases[1][2], float(np.float32(0.6) * np.float32(radius)), float(np.float32(1.05) * np.float32(radius))))
        for tag, ro, rds, tmin, tmax in cases:
            key = f"{name}_{tag}"
            out[key + "_ro"] = np.asarray(ro, np.float32)
            out[key + "_rd"] = np.ascontiguousarray(rds, np.float32)
            out[key + "_t"] = np.array([tmin, tmax], np.float32)
            out[key + "_out"] = orc.ref_octree_ray_skip(g, ro, rds, tmin, tmax)
        # (6) visibility maps: random flags (root kept visible), applied to the probe + random rays
        for k in range(2):
            flags = (rng.random(len(nodes)) < (0.9 if k == 0 else 0.6)).astype(np.uint8)
            flags[0] = 1
            rds = np.concatenate([cases[0][2], cases[1][2][:64]])
            key = f"{name}_vis{k}"
            out[key + "_flags"] = np.packbits(flags)
            out[key + "_ro"] = np.asarray(eye, np.float32)
            out[key + "_rd"] = rds
            out[key + "_t"] = np.array([0.0, 1e30], np.float32)
            out[key + "_out"] = orc.ref_octree_ray_skip(g, eye, rds, 0.0, 1e30, visible=flags)
        # (7) the nearest-hit RENDER MODE: every pixel of a 96 x 64 frame.  Directions: generateRay (the GLSL of
        #     S/RayTracerBVH.cpp:338-355 cannot run here: the oracle's restatement makes them); distances: the reference's compiled
        #     octreeRaySkip for exactly those directions, without and with a visibility map.
        PW, PH = 96, 64
        cams = [(theta, phi, radius)] + ([(1.3, 0.2, 0.55)] if name == "sphere32" else [])        # sphere32: also an eye inside the shell's hollow
        for ci, (t_, p_, r_) in enumerate(cams):
            v_, e_, _ = orc.ref_camera(t_, p_, r_)
            rd_px = orc.generate_rays(v_, e_, PW / PH, 45.0, PW, PH).reshape(-1, 3)
            key = f"{name}_pixels{ci}"
            out[key + "_cam"] = np.array([t_, p_, r_], np.float32)
            out[key + "_out"] = orc.ref_octree_ray_skip(g, e_, rd_px, 0.0, 1e30)
            flags = np.unpackbits(out[f"{name}_vis1_flags"])[: len(nodes)]
            out[key + "_vis_out"] = orc.ref_octree_ray_skip(g, e_, rd_px, 0.0, 1e30, visible=flags)
        # (8) octreeRaySkip's consumer (S/VR:1640-1663) on the probe distances of (1): valid = (t < 1e30 && t > 0), sorted,
        #     index int(n * 0.15f), x 0.75f, blended 0.4 old + 0.6 new -- three consecutive frames starting from 0
        F = np.float32
        seq, last = [], F(0.0)
        for flags_key in (None, f"{name}_vis0_flags"):
            t_probe = out[f"{name}_probe_out"] if flags_key is None else out[f"{name}_vis0_out"][:49]
            for _ in range(3):
                valid = np.sort(t_probe[(t_probe < F(1e30)) & (t_probe > F(0.0))])
                skip = F(0.0)
                if len(valid):
                    skip = F(valid[max(0, int(F(len(valid)) * F(0.15)))] * F(0.75))
                last = F(F(last * F(0.4)) + F(skip * F(F(1.0) - F(0.4))))
                seq.append(last)
        out[f"{name}_probe_skip_seq"] = np.array(seq, np.float32)      # 3 updates without, then 3 with the visibility map vis0
        hits = {k: int((v < 1e30).sum()) for k, v in out.items() if k.startswith(name) and k.endswith("_out")}
        print(name, len(nodes), "nodes; finite results:", hits)
    np.savez_compressed(os.path.join(HERE, "ref_ray_skip.npz"), **out)
    print("wrote ref_ray_skip.npz", os.path.getsize(os.path.join(HERE, "ref_ray_skip.npz")), "bytes")


if __name__ == "__main__":
    main()
