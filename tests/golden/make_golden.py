"""Generates tests/golden/*.npz + golden.json.  Run ONLY where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

Two kinds of vectors, kept apart in the files and labelled in golden.json:

  ref_*   outputs of the REFERENCE'S OWN compiled code (oracle/_ref/libref.so: OctreeVoxel.cpp,
          Camera.cpp, Frustum.cpp, CacheUtils.cpp + header-only glm), i.e. real golden vectors:
          flat octree arrays, node counts/hashes, camera matrices, glm results, frustum verdicts,
          localMC triangles, and the decoded contents of the reference's data file sceneCache.bin.
  orc_*   outputs of this repo's CPU oracle for the GLSL kernel, which cannot run anywhere here
          (no GL context): images, step maps and frame counters.  These pin the oracle against
          accidental change; they are NOT reference outputs ("parity unpinned" for that part).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402

SPHERE_CAM = (0.5, 0.7, 1.8)                      # BASELINE.md configs 1-3,5
CALGARY_DEFAULT = (np.float32(np.pi / 2).item(), 0.0, 500.0)   # glm::radians(90.0f) == float(pi/2); main.cpp:509
CALGARY_OBLIQUE = (0.6, 0.5, 3500.0)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert orc.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    meta: dict = {"generator": "tests/golden/make_golden.py", "octrees": {}, "cameras": {}, "images": {}}

    # ---- ref: flat octrees ------------------------------------------------------------------
    small = {}
    for dim in (16, 32):
        g = orc.test_sphere_grid(dim)
        small[f"sphere{dim}"] = orc.ref_build_flat_octree(g)
        small[f"sphere{dim}_min"] = g.min
        small[f"sphere{dim}_voxel"] = np.float32(g.voxel_size)
    # a non-cubic, non-power-of-two grid (out-of-range voxels read EMPTY, OctreeVoxel.cpp:692-701)
    rng = np.random.default_rng(20250404)
    odd = (rng.random((5, 11, 19)) < 0.35).astype(np.uint8)
    godd = orc.Grid((19, 11, 5), np.array([-3.0, 2.0, 0.5], np.float32), np.float32(0.25), odd)
    small["odd_grid"] = odd
    small["odd_min"] = godd.min
    small["odd_voxel"] = np.float32(godd.voxel_size)
    small["odd"] = orc.ref_build_flat_octree(godd)
    np.savez_compressed(os.path.join(HERE, "ref_octrees_small.npz"), **small)

    for dim in (16, 32, 64, 128, 256, 512):
        g = orc.test_sphere_grid(dim)
        a = orc.ref_build_flat_octree(g)
        meta["octrees"][f"sphere{dim}"] = {
            "source": "reference createOctreeFromVoxelGrid + setOctree BFS numbering",
            "nodes": int(len(a)), "leaves": int((a["isLeaf"] == 1).sum()), "solid_leaves": int(((a["isLeaf"] == 1) & (a["isSolid"] == 1)).sum()),
            "root_size": int(a["size"][0]), "sha256": sha(a), "grid_min": [float(x) for x in g.min], "filled": int(g.data.sum())}
        print("octree", dim, meta["octrees"][f"sphere{dim}"]["nodes"])

    # ---- ref: sceneCache.bin ------------------------------------------------------------------
    cal = orc.ref_load_voxel_grid("/root/reference/sceneCache.bin")
    np.savez_compressed(os.path.join(HERE, "ref_scene_cache.npz"),
                        dims=np.array(cal.dims, np.int32), min=cal.min, voxel=np.float32(cal.voxel_size),
                        packed=np.packbits(cal.data.reshape(-1)))
    a = orc.ref_build_flat_octree(cal)
    meta["octrees"]["calgary"] = {
        "source": "reference loadVoxelGrid(sceneCache.bin) + createOctreeFromVoxelGrid + BFS numbering",
        "dims": list(cal.dims), "grid_min": [float(x) for x in cal.min], "voxel": float(cal.voxel_size),
        "filled": int(cal.data.sum()), "nodes": int(len(a)), "leaves": int((a["isLeaf"] == 1).sum()),
        "solid_leaves": int(((a["isLeaf"] == 1) & (a["isSolid"] == 1)).sum()), "root_size": int(a["size"][0]), "sha256": sha(a)}
    print("calgary", len(a))
    # the file itself (S/CacheUtils.cpp:5-30 wrote it): 3 x int32 dims, 4 x float32 min / voxel size, uint64 count, count bytes.
    # Its SHA-256 and 36-byte header pin the product's saveVoxelGrid / loadVoxelGrid to the reference's BYTES, not just its size.
    raw = open("/root/reference/sceneCache.bin", "rb").read()
    meta["scene_cache_file"] = {"source": "/root/reference/sceneCache.bin as shipped", "bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(),
                                "header_hex": raw[:36].hex(), "voxel_byte_values": sorted(int(v) for v in np.unique(np.frombuffer(raw[36:], np.uint8)))}

    # ---- ref: cameras + glm ---------------------------------------------------------------------
    cams = {"sphere": (SPHERE_CAM, None), "calgary_default": (CALGARY_DEFAULT, (0.0, 100.0)),
            "calgary_oblique": (CALGARY_OBLIQUE, None), "panned": ((0.1, 2.0, 3.0), (3.0, -2.0))}
    camz = {}
    R = orc.ref()
    for name, ((t, p, r), pan) in cams.items():
        view, pos, tgt = orc.ref_camera(t, p, r, pan)
        inv = np.zeros(16, np.float32); R.ref_glm_inverse(view, inv)
        persp = np.zeros(16, np.float32); R.ref_glm_perspective(R.ref_glm_radians(45.0), np.float32(1920 / 1080).item(), 0.01, 5000.0, persp)
        vp = np.zeros(16, np.float32); R.ref_glm_mul(persp, view, vp)
        camz[name + "_params"] = np.array([t, p, r, 1.0 if pan else 0.0, pan[0] if pan else 0.0, pan[1] if pan else 0.0], np.float32)
        camz[name + "_view"] = view; camz[name + "_pos"] = pos; camz[name + "_target"] = tgt
        camz[name + "_inv"] = inv; camz[name + "_persp"] = persp; camz[name + "_vp"] = vp
        meta["cameras"][name] = {"theta": t, "phi": p, "radius": r, "pan": pan}
    # frustum verdicts for boxes around the calgary scene (testAABB, margin 150 and 0)
    boxes_min = (rng.random((512, 3)).astype(np.float32) - 0.5) * np.float32(12000.0)
    boxes_ext = (rng.random((512, 1)).astype(np.float32) ** 3) * np.float32(2560.0) + np.float32(10.0)
    boxes_max = boxes_min + boxes_ext
    camz["frustum_min"] = boxes_min; camz["frustum_max"] = boxes_max
    for name in ("calgary_default", "calgary_oblique", "sphere"):
        for margin in (150.0, 0.0):
            camz[f"frustum_{name}_m{int(margin)}"] = orc.ref_frustum_test(camz[name + "_vp"], boxes_min, boxes_max, margin)
    # glm normalize / mat*vec probes
    vecs = (rng.random((64, 4)).astype(np.float32) - 0.5) * 4
    n3 = np.zeros((64, 3), np.float32); n4 = np.zeros((64, 4), np.float32); mv = np.zeros((64, 4), np.float32)
    for i in range(64):
        o3 = np.zeros(3, np.float32); R.ref_glm_normalize3(np.ascontiguousarray(vecs[i, :3]), o3); n3[i] = o3
        o4 = np.zeros(4, np.float32); R.ref_glm_normalize4(np.ascontiguousarray(vecs[i]), o4); n4[i] = o4
        o4 = np.zeros(4, np.float32); R.ref_glm_mat_vec(camz["sphere_inv"], np.ascontiguousarray(vecs[i]), o4); mv[i] = o4
    camz["probe_vecs"] = vecs; camz["probe_normalize3"] = n3; camz["probe_normalize4"] = n4; camz["probe_matvec_sphere_inv"] = mv
    np.savez_compressed(os.path.join(HERE, "ref_cameras.npz"), **camz)

    # ---- ref: localMC (config 5 input) ------------------------------------------------------------
    g16 = orc.test_sphere_grid(16)
    np.savez_compressed(os.path.join(HERE, "ref_localmc_sphere16.npz"),
                        whole=orc.ref_local_mc(g16, 0, 0, 0, 16), cell_4_4_4_s4=orc.ref_local_mc(g16, 4, 4, 4, 4))

    # ---- orc: kernel outputs of the oracle (own restatement) --------------------------------------
    imgs = {}
    for dim, (W, H) in ((16, (64, 64)), (32, (96, 64))):
        g = orc.test_sphere_grid(dim)
        nodes = orc.build_flat_octree(g)
        cam = orc.Camera(*SPHERE_CAM)
        img, st = orc.render(nodes, g.min, g.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        steps = orc.render_steps(nodes, g.min, g.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        imgs[f"sphere{dim}_{W}x{H}_rgba"] = img
        imgs[f"sphere{dim}_{W}x{H}_steps"] = steps
        meta["images"][f"sphere{dim}_{W}x{H}"] = {"source": "oracle (own restatement of the GLSL)", **{k: int(v) for k, v in st.items()}}
    np.savez_compressed(os.path.join(HERE, "orc_images_small.npz"), **imgs)
    for dim, (W, H) in ((64, (512, 512)), (256, (1920, 1080))):
        g = orc.test_sphere_grid(dim)
        nodes = orc.build_flat_octree(g)
        cam = orc.Camera(*SPHERE_CAM)
        out = np.zeros((H, W, 4), np.float32)
        img, st = orc.render(nodes, g.min, g.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H, nthreads=8, out=out)
        meta["images"][f"sphere{dim}_{W}x{H}"] = {"source": "oracle (own restatement of the GLSL)", "sha256": sha(img),
                                                   **{k: int(v) for k, v in st.items()}}
        print("image", dim, st)
    calnodes = orc.build_flat_octree(cal)
    for name, W, H in (("calgary_default", 1300, 1300), ("calgary_oblique", 1920, 1080)):
        (t, p, r), pan = cams[name]
        cam = orc.Camera(t, p, r)
        if pan:
            cam.pan(*pan)
        out = np.zeros((H, W, 4), np.float32)
        img, st = orc.render(calnodes, cal.min, cal.voxel_size, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H, nthreads=8, out=out)
        meta["images"][f"{name}_{W}x{H}"] = {"source": "oracle (own restatement of the GLSL)", "sha256": sha(img),
                                             **{k: int(v) for k, v in st.items()}}
        print("image", name, st)

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
