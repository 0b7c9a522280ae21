"""GPU parity tests proper: the HIP path, called through the C ABI (ray_tracing_octrees_amd.hip.Context
is a 1:1 ctypes binding of include/rto_hip.h), against the CPU oracle on the same inputs.

Bar: BIT-EXACT RGBA32F pixels (stricter than BASELINE.json's 1e-4 per channel -- the tolerance written in
the north star is 1e-4; every comparison below asserts equality of the float bit patterns) and exact
per-pixel traversal step counts (the reference's first-hit-in-DFS-order + 512-pop-cap semantics)."""
import ctypes as C
import os

import numpy as np
import pytest

import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip
from conftest import Scene, assert_bit_exact, partition_row_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TOL = 1e-4   # north-star tolerance; never reached: see assert_bit_exact

KERNELS = [("packed", rto.KERNEL_PACKED), ("packed_persistent", rto.KERNEL_PACKED_PERSISTENT), ("packed_v3", rto.KERNEL_PACKED_V3),
           ("packed_v1", rto.KERNEL_PACKED_V1), ("generic", rto.KERNEL_GENERIC)]


def oracle_frame(orc, s, view, pos, W, H, aspect=None, fov=45.0):
    out = np.zeros((H, W, 4), np.float32)
    return orc.render(s.nodes, s.min, s.voxel, view, pos, (W / H) if aspect is None else aspect, fov, W, H,
                      nthreads=min(16, orc.max_threads()), out=out)


def upload(ctx, s):
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.upload_octree(s.nodes, s.min, s.voxel)


@pytest.mark.parametrize("kname,kernel", KERNELS)
@pytest.mark.parametrize("scene,W,H", [("sphere16", 64, 64), ("sphere32", 96, 64), ("sphere64", 512, 512),
                                       ("odd", 200, 120), ("sphere32", 1, 1), ("sphere32", 7, 13), ("sphere32", 250, 9)])
def test_pixels_and_steps_match_oracle(ctx, orc, scenes, camera, scene, W, H, kname, kernel):
    s = scenes(scene)
    if scene == "odd":
        cam = orc.Camera(0.4, 0.9, 9.0)
        view, pos = cam.get_view(), cam.get_pos()
    else:
        view, pos = camera("sphere")
    upload(ctx, s)
    ctx.set_kernel(kernel)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    got = ctx.render_host(f)
    assert np.abs(got - want).max() <= TOL
    assert_bit_exact(got, want, f"{scene} {W}x{H} {kname}")
    np.testing.assert_array_equal(ctx.render_steps(f), orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H))
    gs = ctx.frame_stats(f)
    assert (gs["rays"], gs["pops"], gs["hits"], gs["capped"]) == (W * H, st["pops"], st["hits"], st["capped"])


def _random_grid(orc, rng):
    """Random occupancy with structure at several scales (blobs, slabs, noise), random dims / origin / voxel size."""
    dims = tuple(int(rng.integers(3, 41)) for _ in range(3))            # (dimX, dimY, dimZ), mostly not powers of two
    dz, dy, dx = dims[2], dims[1], dims[0]
    data = np.zeros((dz, dy, dx), np.uint8)
    zz, yy, xx = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    for _ in range(int(rng.integers(1, 5))):                            # solid blobs -> large uniform leaves
        c = rng.uniform(0, 1, 3) * (dz, dy, dx)
        r = rng.uniform(1.5, 0.45 * max(dims))
        data[(zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r] = 1
    if rng.random() < 0.5:                                              # a thin slab
        data[:, int(rng.integers(0, dy)), :] = 1
    noise = rng.random(data.shape) < rng.choice([0.0, 0.02, 0.3])
    data ^= noise.astype(np.uint8)                                      # salt and pepper -> depth to min-leaf 1
    gmin = rng.uniform(-50, 50, 3).astype(np.float32)
    voxel = np.float32(rng.choice([0.03125, 0.7, 1.0, 3.3]))
    if rng.random() < 0.5:                                              # a "round" origin: with 0.03125 / 1.0 every node plane is then exact in float
        gmin = (np.round(gmin / voxel) * voxel).astype(np.float32)      # (the lean kernels' 9-plane child test, rto_api.hip grid_is_exact)
    return orc.Grid(dims, gmin, voxel, data)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTO_FUZZ_START", "0")),          # a larger sweep: RTO_FUZZ_SEEDS=400 [RTO_FUZZ_START=2500]
                                        int(os.environ.get("RTO_FUZZ_START", "0")) + int(os.environ.get("RTO_FUZZ_SEEDS", "12"))))
def test_fuzz_random_scenes_and_cameras(ctx, orc, seed):
    """Seeded fuzz: random grids, cameras outside / inside / grazing, random image sizes, fov and aspect; every
    kernel must reproduce the oracle's pixel bits, per-pixel step counts and frame counters; the triangle path too."""
    rng = np.random.default_rng(1000 + seed)
    g = _random_grid(orc, rng)
    nodes = orc.build_flat_octree(g)
    s = Scene(g, nodes)
    upload(ctx, s)
    ext = np.float32(max(g.dims)) * g.voxel_size
    centre = g.min + 0.5 * np.array(g.dims, np.float32) * g.voxel_size
    tris = off = None
    shots = []
    for shot in range(3):
        kind = ("outside", "inside", "grazing")[shot]
        radius = float(ext * {"outside": rng.uniform(1.2, 4.0), "inside": rng.uniform(0.05, 0.45), "grazing": rng.uniform(0.7, 1.0)}[kind])
        cam = orc.Camera(float(rng.uniform(0, 6.28)), float(rng.uniform(-1.4, 1.4)), radius)
        cam.set_target(*[float(x) for x in centre + rng.uniform(-0.2, 0.2, 3).astype(np.float32) * ext])
        view, pos = cam.get_view(), cam.get_pos()
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        fov = float(rng.choice([20.0, 45.0, 90.0, 120.0]))
        aspect = float(rng.choice([W / H, 1.0, 2.5]))
        f = rto.make_frame(view, pos, aspect, fov, W, H)
        want, st = oracle_frame(orc, s, view, pos, W, H, aspect=aspect, fov=fov)
        steps = orc.render_steps(s.nodes, s.min, s.voxel, view, pos, aspect, fov, W, H)
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            what = f"seed {seed} {kind} {g.dims} {W}x{H} fov {fov} {kname}"
            assert_bit_exact(ctx.render_host(f), want, what)
            np.testing.assert_array_equal(ctx.render_steps(f), steps, err_msg=what)
            gs = ctx.frame_stats(f)
            assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), what
        # a grid whose planes are exact runs the 9-plane child test: the general form must give the same frame and the same step counts
        ctx.set_kernel(rto.KERNEL_AUTO)
        if ctx.debug_set_exact_grid(True)[1]:
            ctx.debug_set_exact_grid(False)
            try:
                assert_bit_exact(ctx.render_host(f), want, f"seed {seed} {kind}: general child test on an exact grid")
                np.testing.assert_array_equal(ctx.render_steps(f), steps)
            finally:
                ctx.debug_set_exact_grid(True)
        # the nearest-hit render mode (octreeRaySkip per pixel) on the same frame
        nrgba, nt = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, aspect, fov, W, H)
        for _ in range(2):
            grgba, gt_ = ctx.render_skip_host(f)
        assert gt_.tobytes() == nt.tobytes(), f"seed {seed} {kind} {g.dims} {W}x{H} fov {fov}: nearest-hit distances"
        assert_bit_exact(grgba, nrgba, f"seed {seed} {kind} nearest-hit colours")
        # the closest-hit mode (the reference's earlier shader): near-first, pop-order and node-by-node kernels; twice each (the second
        # frame runs under the launch order and the rim the first one left)
        cwant, cst = orc.render_closest(s.nodes, s.min, s.voxel, view, pos, aspect, fov, W, H)
        for kname, kernel in (("near first", rto.KERNEL_AUTO), ("pop order", rto.KERNEL_PACKED_V1), ("node by node", rto.KERNEL_GENERIC)):
            ctx.set_kernel(kernel)
            for rep in range(2):
                assert_bit_exact(ctx.render_closest_host(f), cwant, f"seed {seed} {kind} {g.dims} {W}x{H} fov {fov} closest hit, {kname}, frame {rep}")
        ctx.set_kernel(rto.KERNEL_AUTO)
        _, gcs = ctx.render_closest_host(f, stats=True)
        assert (gcs["pops"], gcs["hits"]) == (cst["pops"], cst["hits"]), f"seed {seed} {kind} closest-hit counters"
        if shot == 0:
            tris, off = orc.build_leaf_triangles(s.grid, s.nodes)
            ctx.build_leaf_triangles(g.data)                            # GPU builder == oracle's buffer, then render from it
            gt, go = ctx.download_leaf_triangles()
            assert go.tobytes() == np.asarray(off, np.int32).tobytes(), f"seed {seed}: triOffset"
            assert gt.tobytes() == np.ascontiguousarray(tris, np.float32).reshape(-1, 12).tobytes(), f"seed {seed}: triangles"
        if len(tris):
            wt, wst = orc.render_triangles(s.nodes, tris, off, s.min, s.voxel, view, pos, aspect, fov, W, H, shadow=True)
            for kname, kernel in (("lean", rto.KERNEL_AUTO), ("packed", rto.KERNEL_PACKED_V3), ("generic", rto.KERNEL_GENERIC)):
                ctx.set_kernel(kernel)
                got, gs = ctx.render_triangles_host(f, shadow=True, stats=True)
                assert_bit_exact(got, wt, f"seed {seed} {kind} triangles {kname}")
                assert (gs["pops"], gs["hits"]) == (wst["pops"], wst["hits"]), f"seed {seed} {kind} triangles {kname}"
            shots.append((f, W, H, want, wt))
    ctx.set_kernel(rto.KERNEL_AUTO)
    # the same frames through the several-frames-per-launch kernels (frames of one launch share width and height)
    torch = pytest.importorskip("torch")
    for f, W, H, want, wt in shots:
        arr = hip.Context.frame_array([f, f, f])
        out = torch.full((3, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
        ctx.render_batch_device(arr, out.data_ptr(), out.stride(0) * 4, None, False, 0)
        torch.cuda.synchronize()
        for i in range(3):
            assert_bit_exact(out[i].cpu().numpy(), want, f"seed {seed} batched frame {i}")
        out.fill_(7.0)
        ctx.render_triangles_batch_device(arr, out.data_ptr(), out.stride(0) * 4, True, None, False, 0)
        torch.cuda.synchronize()
        for i in range(3):
            assert_bit_exact(out[i].cpu().numpy(), wt, f"seed {seed} batched triangle frame {i}")


@pytest.mark.parametrize("kname,kernel", KERNELS)
def test_golden_images(ctx, scenes, camera, golden, kname, kernel):
    z = golden("orc_images_small.npz")
    view, pos = camera("sphere")
    for dim, (W, H) in ((16, (64, 64)), (32, (96, 64))):
        upload(ctx, scenes(f"sphere{dim}"))
        ctx.set_kernel(kernel)
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        assert_bit_exact(ctx.render_host(f), z[f"sphere{dim}_{W}x{H}_rgba"], f"golden sphere{dim} {kname}")
        np.testing.assert_array_equal(ctx.render_steps(f), z[f"sphere{dim}_{W}x{H}_steps"])


@pytest.mark.parametrize("kname,kernel", KERNELS)
def test_config2_full_size_256_1080p(ctx, orc, scenes, camera, golden_meta, kname, kernel):
    """BASELINE config 2 at full size: 256^3 shell sphere, 1920x1080, including the 434 capped rays."""
    s = scenes("sphere256")
    view, pos = camera("sphere")
    W, H = 1920, 1080
    upload(ctx, s)
    ctx.set_kernel(kernel)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    got = ctx.render_host(f)
    assert_bit_exact(got, want, f"config2 {kname}")
    gs = ctx.frame_stats(f)
    m = golden_meta["images"]["sphere256_1920x1080"]
    assert (gs["pops"], gs["hits"], gs["capped"]) == (m["pops"], m["hits"], m["capped"]) == (st["pops"], st["hits"], st["capped"])
    # size-independent properties of any frame of this path
    assert (got[..., 3] == 1.0).all()
    miss = (got[..., :3] == 0).all(axis=-1)
    assert int((~miss).sum()) == st["hits"]
    lit = got[~miss][:, :3]
    assert lit.min() >= np.float32(0.1) and lit.max() <= np.float32(1.1)


@pytest.mark.parametrize("scene", ["sphere256", "calgary",
                                   pytest.param("sphere512", marks=pytest.mark.skipif(not os.environ.get("RTO_CAMERA_BIG"),
                                                                                    reason="config 5's size (512^3, 4K): RTO_CAMERA_BIG=1 sweeps it"))])
def test_random_cameras_at_full_size(ctx, orc, scenes, scene):
    """The launch geometry (solid rectangle, occupancy mask, cost-sorted order, 4 / 6 resident waves) is chosen per frame from the
    camera: seeded random cameras at 1920x1080 -- far, near, grazing, inside the volume, looking past it -- give the oracle's
    pixels in the first-hit mode (three frames each: the table of the previous camera, then its own) and in the nearest-hit mode,
    and the instrumented frame's counters agree.  RTO_CAMERA_SEEDS widens the sweep (default 4 per scene)."""
    s = scenes(scene)
    upload(ctx, s)
    tris = off = None
    if scene != "calgary":                                     # the triangle path (config 5's kernels) from the same cameras
        tris, off = orc.build_leaf_triangles(s.grid, s.nodes)
        ctx.upload_leaf_triangles(tris, off)
    W, H = (3840, 2160) if scene == "sphere512" else (1920, 1080)
    n = int(os.environ.get("RTO_CAMERA_SEEDS", "4"))
    first = int(os.environ.get("RTO_CAMERA_START", "0"))
    dims = np.array(s.grid.dims, np.float32)
    ext = float(dims.max() * s.voxel)
    centre = np.asarray(s.min, np.float32) + 0.5 * dims * np.float32(s.voxel)
    for seed in range(first, first + n):
        rng = np.random.default_rng(77000 + seed)
        kind = ("far", "near", "inside", "past")[seed % 4]
        radius = ext * {"far": rng.uniform(1.5, 5.0), "near": rng.uniform(0.55, 1.0), "inside": rng.uniform(0.05, 0.45), "past": rng.uniform(0.8, 2.0)}[kind]
        cam = orc.Camera(float(rng.uniform(0, 6.28)), float(rng.uniform(-1.4, 1.4)), float(radius))
        aim = rng.uniform(-0.15, 0.15, 3) if kind != "past" else rng.uniform(0.6, 1.2, 3) * rng.choice([-1.0, 1.0], 3)
        cam.set_target(*[float(x) for x in centre + aim.astype(np.float32) * ext])
        view, pos = cam.get_view(), cam.get_pos()
        fov = float(rng.choice([30.0, 45.0, 70.0]))
        f = rto.make_frame(view, pos, W / H, fov, W, H)
        want, st = oracle_frame(orc, s, view, pos, W, H, fov=fov)
        for k in range(3):
            assert_bit_exact(ctx.render_host(f), want, f"{scene} seed {seed} ({kind}) frame {k}")
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), (scene, seed, kind)
        nrgba, nt = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H, nthreads=min(16, orc.max_threads()))
        for k in range(2):
            grgba, gt = ctx.render_skip_host(f)
        assert gt.tobytes() == nt.tobytes() and grgba.tobytes() == nrgba.tobytes(), f"{scene} seed {seed} ({kind}): nearest-hit mode"
        cw, cst = orc.render_closest(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H, nthreads=min(16, orc.max_threads()))
        cg, cgs = ctx.render_closest_host(f, stats=True)     # the reference's closest-hit shader (round 4)
        assert_bit_exact(cg, cw, f"{scene} seed {seed} ({kind}): closest-hit mode")
        assert (cgs["pops"], cgs["hits"]) == (cst["pops"], cst["hits"]), (scene, seed, kind, "closest-hit counters")
        if tris is not None:
            wt, wst = orc.render_triangles(s.nodes, tris, off, s.min, s.voxel, view, pos, W / H, fov, W, H, shadow=True, nthreads=min(16, orc.max_threads()))
            for k in range(2):                                  # the colour kernel (mask, launch order), then the instrumented one
                assert_bit_exact(ctx.render_triangles_host(f, shadow=True), wt, f"{scene} seed {seed} ({kind}) triangles + shadow, frame {k}")
            got, gs = ctx.render_triangles_host(f, shadow=True, stats=True)
            assert_bit_exact(got, wt, f"{scene} seed {seed} ({kind}) triangles + shadow, instrumented frame")
            assert (gs["pops"], gs["hits"]) == (wst["pops"], wst["hits"]), (scene, seed, kind)
    assert ctx.debug_sort_violations() == 0


@pytest.mark.parametrize("cam_name,W,H", [("calgary_default", 1300, 1300), ("calgary_oblique", 1920, 1080)])
def test_config4_calgary(ctx, orc, scenes, camera, golden_meta, cam_name, W, H):
    """BASELINE config 4: the shipped sceneCache.bin grid (425x243x29, root 512); default app camera (eye
    inside a solid leaf: every ray hits at t=0) and the oblique, divergent camera."""
    s = scenes("calgary")
    view, pos = camera(cam_name)
    upload(ctx, s)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    m = golden_meta["images"][f"{cam_name}_{W}x{H}"]
    assert (st["pops"], st["hits"]) == (m["pops"], m["hits"])
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f), want, f"calgary {cam_name} {kname}")
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])


def test_config4_resampled_to_512x512x128(ctx, orc, scenes, camera):
    """BASELINE.json names config 4 as "voxelized at 512x512x128"; that grid cannot be regenerated (its source data is
    absent upstream).  SURVEY 8d's stand-in: a nearest-neighbour resample of the shipped 425x243x29 grid to
    512x512x128 -- SYNTHETIC, labelled as such -- at 1920x1080 with the oblique (divergent) camera."""
    src = scenes("calgary")
    d = src.grid.data                                   # (dimZ, dimY, dimX)
    tz, ty, tx = 128, 512, 512
    iz = ((np.arange(tz) + 0.5) * d.shape[0] / tz).astype(np.int64)
    iy = ((np.arange(ty) + 0.5) * d.shape[1] / ty).astype(np.int64)
    ix = ((np.arange(tx) + 0.5) * d.shape[2] / tx).astype(np.int64)
    data = np.ascontiguousarray(d[iz][:, iy][:, :, ix])
    g = orc.Grid((tx, ty, tz), src.min, src.voxel, data)
    s = Scene(g, orc.build_flat_octree(g))
    assert s.nodes[0]["size"] == 512
    upload(ctx, s)
    W, H = 1920, 1080
    cam = orc.Camera(0.6, 0.5, 4500.0)
    cam.set_target(*[float(x) for x in (g.min + 0.5 * np.array(g.dims, np.float32) * g.voxel_size)])
    view, pos = cam.get_view(), cam.get_pos()
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    assert st["hits"] > 100_000
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f), want, f"config 4 resampled {kname}")
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), kname
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.render_host(f); ctx.render_host(f)
    print(f"config 4 (synthetic 512x512x128 resample, {len(s.nodes)} nodes): {st['hits']} hits, {st['pops'] / (W * H):.1f} pops/ray, "
          f"{st['capped']} capped, kernel {ctx.last_kernel_ms() * 1e3:.1f} us")
    # the GPU builder makes the same array from these voxels
    ctx.build_octree(data, g.min, g.voxel_size)
    assert ctx.download_nodes().tobytes() == s.nodes.tobytes()


def test_depth10_grid_1024(ctx, orc):
    """A 1024^3 grid (1 GiB of voxels, root edge 1024, depth 10 -- one level beyond every BASELINE config): the GPU
    builder reproduces the oracle's array and every kernel its pixels, steps and counters."""
    N = 1024
    d = np.zeros((N, N, N), np.uint8)
    for (cx, cy, cz, r) in ((300, 400, 500, 120), (700, 650, 300, 90), (512, 512, 800, 60)):
        z, y, x = np.ogrid[cz - r:cz + r + 1, cy - r:cy + r + 1, cx - r:cx + r + 1]
        d[cz - r:cz + r + 1, cy - r:cy + r + 1, cx - r:cx + r + 1] |= ((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2 <= r * r).astype(np.uint8)
    mn = np.array([-0.5, -0.5, -0.5], np.float32)
    vs = np.float32(1.0 / N)
    s = Scene(orc.Grid((N, N, N), mn, vs, d), None)
    s.nodes = orc.build_flat_octree(s.grid)
    assert s.nodes[0]["size"] == N
    ctx.build_octree(d, mn, vs)
    assert ctx.info().num_nodes == len(s.nodes)
    assert ctx.download_nodes().tobytes() == s.nodes.tobytes()
    cam = orc.Camera(0.4, 0.8, 1.6)
    view, pos = cam.get_view(), cam.get_pos()
    W, H = 1280, 720
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    steps = orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H)
    assert st["hits"] > 10_000
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f), want, f"1024^3 {kname}")
        np.testing.assert_array_equal(ctx.render_steps(f), steps, err_msg=kname)
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), kname
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.build_leaf_triangles(None)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    gt, go = ctx.download_leaf_triangles()
    assert go.tobytes() == np.asarray(wo, np.int32).tobytes() and gt.tobytes() == np.ascontiguousarray(wt, np.float32).tobytes()
    wtri, _ = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=True, nthreads=min(16, orc.max_threads()))
    assert_bit_exact(ctx.render_triangles_host(f, shadow=True), wtri, "1024^3 triangles + shadow")


def test_config5_primary_rays_512_4k(ctx, orc, scenes, camera):
    """BASELINE config 5, primary rays only (the triangle/shadow extension has no reference): 512^3, 3840x2160."""
    s = scenes("sphere512")
    view, pos = camera("sphere")
    W, H = 3840, 2160
    upload(ctx, s)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = oracle_frame(orc, s, view, pos, W, H)
    assert_bit_exact(ctx.render_host(f), want, "config5 primary")
    gs = ctx.frame_stats(f)
    assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])
    assert st["capped"] > 10000      # ~15.7k rays run into the 512-pop cap here


def test_axis_aligned_rays_take_the_exact_nan_path(ctx, orc, scenes):
    """Rays with zero direction components give 1/0 = inf and 0*inf = NaN in the slab test (S/RT:228-230).
    An axis-aligned camera at odd resolution puts such rays through grid-plane-aligned origins."""
    s = scenes("sphere32")
    upload(ctx, s)
    W, H = 65, 33
    for (t, p, r, target) in ((0.0, 0.0, 1.5, (0.0, 0.0, 0.0)), (0.0, np.float32(np.pi / 2), 2.0, (0.0, 0.0, 0.0)),
                              (0.0, 0.0, 1.0, (float(s.min[0] + 8 * s.voxel), float(s.min[1] + 16 * s.voxel), 0.0))):
        cam = orc.Camera(t, p, r)
        cam._c.target[0], cam._c.target[1], cam._c.target[2] = target
        view, pos = cam.get_view(), cam.get_pos()
        want, st = oracle_frame(orc, s, view, pos, W, H)
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            assert_bit_exact(ctx.render_host(f), want, f"axis-aligned {kname}")
            np.testing.assert_array_equal(ctx.render_steps(f), orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H))
        assert st["hits"] > 0


def test_eye_inside_solid_and_far_away(ctx, orc, scenes):
    s = scenes("sphere32")
    upload(ctx, s)
    W, H = 96, 64
    for radius in (0.3, 0.05, 40.0):     # inside the shell material (tHit = 0), inside the hollow core, sub-pixel sphere
        cam = orc.Camera(0.2, 0.3, radius)
        view, pos = cam.get_view(), cam.get_pos()
        want, _ = oracle_frame(orc, s, view, pos, W, H)
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            assert_bit_exact(ctx.render_host(f), want, f"r={radius} {kname}")


def test_aspect_and_fov_are_the_callers(ctx, orc, scenes, camera):
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 120, 50
    for aspect, fov in ((1.0, 45.0), (2.4, 45.0), (W / H, 20.0), (0.5, 100.0)):
        want, _ = oracle_frame(orc, s, view, pos, W, H, aspect=aspect, fov=fov)
        assert_bit_exact(ctx.render_host(rto.make_frame(view, pos, aspect, fov, W, H)), want, f"aspect {aspect} fov {fov}")


def test_single_node_and_noncanonical_arrays_fall_back_to_generic(ctx, orc, scenes, camera):
    view, pos = camera("sphere")
    W, H = 64, 48
    # (a) a one-node tree: the whole grid is one solid leaf
    for fill in (1, 0):
        g = orc.Grid((4, 4, 4), np.array([-0.5, -0.5, -0.5], np.float32), np.float32(0.25), np.full((4, 4, 4), fill, np.uint8))
        nodes = orc.build_flat_octree(g)
        assert len(nodes) == 1
        ctx.set_kernel(rto.KERNEL_AUTO)
        ctx.upload_octree(nodes, g.min, g.voxel_size)
        assert ctx.info().canonical == 0
        want, _ = orc.render(nodes, g.min, g.voxel_size, view, pos, W / H, 45.0, W, H)
        assert_bit_exact(ctx.render_host(rto.make_frame(view, pos, W / H, 45.0, W, H)), want, "single node")
        with pytest.raises(rto.RtoError) as e:
            ctx.set_kernel(rto.KERNEL_PACKED)
        assert e.value.code == hip.RTO_E_UNSUPPORTED
    # (b) a frustum-compacted array uploaded as-is: children no longer consecutive, some are -1
    cal = scenes("calgary")
    cview, cpos = camera("calgary_oblique")
    compact, _ = orc.cull_compact(cal.nodes, cal.min, cal.voxel, cview, 45.0, W / H)
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.upload_octree(compact, cal.min, cal.voxel)
    assert ctx.info().canonical == 0
    want, _ = orc.render(compact, cal.min, cal.voxel, cview, cpos, W / H, 45.0, W, H)
    assert_bit_exact(ctx.render_host(rto.make_frame(cview, cpos, W / H, 45.0, W, H)), want, "compacted upload")


@pytest.mark.parametrize("cam_name", ["calgary_oblique", "calgary_default"])
def test_frustum_update_equals_reference_compaction(ctx, orc, scenes, camera, cam_name):
    """renderSceneComputeWithCulling(updateFrustum=true): GPU test + compaction == S/RT:731-802 on the CPU."""
    cal = scenes("calgary")
    view, pos = camera(cam_name)
    W, H = 480, 270
    aspect = W / H
    upload(ctx, cal)
    want_nodes, vis = orc.cull_compact(cal.nodes, cal.min, cal.voxel, view, 45.0, aspect)
    ctx.update_frustum(view, 45.0, aspect, enable=True)
    got_nodes = ctx.download_visible_nodes()
    assert ctx.info().culling_active == 1 and ctx.info().visible_nodes == len(want_nodes) == int(vis.sum())
    assert got_nodes.tobytes() == want_nodes.tobytes()
    want, st = orc.render(want_nodes, cal.min, cal.voxel, view, pos, aspect, 45.0, W, H)
    f = rto.make_frame(view, pos, aspect, 45.0, W, H)
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f), want, f"culled {cam_name} {kname}")
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])
    # a different camera through the culled tree (what the reference shows between frustum updates)
    cam2 = orc.Camera(0.9, 2.5, 3000.0)
    v2, p2 = cam2.get_view(), cam2.get_pos()
    want2, _ = orc.render(want_nodes, cal.min, cal.voxel, v2, p2, aspect, 45.0, W, H)
    f2 = rto.make_frame(v2, p2, aspect, 45.0, W, H)
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f2), want2, f"culled, other camera {kname}")
    # disabling restores the full tree
    ctx.update_frustum(view, 45.0, aspect, enable=False)
    full, _ = orc.render(cal.nodes, cal.min, cal.voxel, v2, p2, aspect, 45.0, W, H)
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(f2), full, f"culling off {kname}")


def test_frustum_update_proven_on_the_host_equals_the_kernel(ctx, orc, scenes):
    """rto_update_frustum asks the host first whether the planes can cull any node at all (a lower bound of the positive-vertex
    value over every box inside the root's, a thousandfold above the float evaluation's error); when not, it writes the "every node
    visible" state once and launches nothing.  Same outcome as the kernel, which the debug switch forces: survivors (all nodes,
    == the oracle's compaction), frames, counters -- for cameras around a small scene (the reference's 150-unit margin: proven),
    across an interleaved culling update with planes that DO cull (custom planes, margin 0: not proven), and back."""
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 320, 200
    aspect = W / H
    n = len(s.nodes)
    try:
        for i, cam in enumerate((orc.Camera(0.5, 0.7, 1.8), orc.Camera(2.1, -0.4, 0.9), orc.Camera(0.0, 1.5, 6.0))):
            view, pos = cam.get_view(), cam.get_pos()
            want_nodes, vis = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, aspect)
            assert len(want_nodes) == n                           # the reference's loop passes every node here
            f = rto.make_frame(view, pos, aspect, 45.0, W, H)
            want, st = orc.render(want_nodes, s.min, s.voxel, view, pos, aspect, 45.0, W, H)
            outs = []
            for shortcut in (True, False, True):
                ctx.debug_set_frustum_shortcut(shortcut)
                ctx.update_frustum(view, 45.0, aspect, enable=True)
                ctx.update_frustum(view, 45.0, aspect, enable=True)          # a second update in the same state
                assert ctx.debug_last_frustum_update_proven() == shortcut
                assert ctx.info().culling_active == 1 and ctx.info().visible_nodes == n
                assert ctx.download_visible_nodes().tobytes() == want_nodes.tobytes()
                assert_bit_exact(ctx.render_host(f), want, f"camera {i}, host-side proof {'on' if shortcut else 'off'}")
                gs = ctx.frame_stats(f)
                assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])
                t = ctx.octree_ray_skip(pos, np.array([[0, 0, -1], [0.2, -0.3, -0.9], [-0.5, 0.1, -0.8], [0.1, 0.1, -1]], np.float32),
                                        use_visibility=True)
                outs.append(t.tobytes())
            assert outs[0] == outs[1] == outs[2]
            # an update that does cull, in between: a plane through the middle of the grid, no margin
            planes = np.array([[1, 0, 0, 0.1], [-1, 0, 0, 50], [0, 1, 0, 50], [0, -1, 0, 50], [0, 0, 1, 50], [0, 0, -1, 50]], np.float32)
            ctx.debug_set_frustum_shortcut(True)
            ctx.debug_update_frustum_planes(planes, 0.0)
            assert 0 < ctx.info().visible_nodes < n
        # a scene thousands of units across: the margin no longer covers it, whatever the proof says both paths equal the reference's loop
        cal = scenes("calgary")
        upload(ctx, cal)
        for cam in (orc.Camera(0.6, 0.5, 3500.0), orc.Camera(1.2, 0.1, 900.0), orc.Camera(2.5, 0.8, 200.0)):
            view = cam.get_view()
            want_nodes, vis = orc.cull_compact(cal.nodes, cal.min, cal.voxel, view, 45.0, aspect)
            for shortcut in (True, False):
                ctx.debug_set_frustum_shortcut(shortcut)
                ctx.update_frustum(view, 45.0, aspect, enable=True)
                assert ctx.info().visible_nodes == len(want_nodes)
                assert ctx.download_visible_nodes().tobytes() == want_nodes.tobytes()
    finally:
        ctx.debug_set_frustum_shortcut(True)
        ctx.update_frustum(np.eye(4, dtype=np.float32).reshape(16), 45.0, aspect, enable=False)


def test_frustum_proof_never_claims_more_than_the_kernel_finds(ctx, orc):
    """Property: over random plane sets and margins around a small grid -- planes far away, planes grazing the root box, planes
    cutting it -- the update with the host-side proof and the update with the kernel forced leave the same survivors.  (If the
    proof ever fired for planes that cull a node, the counts would differ.)  Some sets must be proven, some not."""
    rng = np.random.default_rng(99)
    dims = (16, 16, 16)
    data = (rng.random((16, 16, 16)) < 0.3).astype(np.uint8)
    gmin = np.array([-3.0, 10.0, 0.5], np.float32)
    g = orc.Grid(dims, gmin, np.float32(0.25), data)
    nodes = orc.build_flat_octree(g)
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.upload_octree(nodes, g.min, g.voxel_size)
    n = len(nodes)
    centre = gmin + 2.0
    all_visible = some_culled = proven = 0
    try:
        for trial in range(120):
            planes = np.zeros((6, 4), np.float32)
            margin = float(rng.choice([0.0, 0.1, 1.0, 5.0, 150.0]))
            for i in range(6):
                nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
                # signed distance of the grid's centre from the plane: from well inside (+8) to cutting the box (0) to outside (-3);
                # every fifth trial: every plane a hair outside the widened root box (all nodes pass, too close to be proven)
                reach = 2.0 * float(np.abs(nrm).sum()) + margin * float(np.abs(nrm).sum())
                dist = reach + 0.003 if trial % 5 == 4 else float(rng.choice([8.0 + 2 * margin, 3.5, 2.0 * 1.7320508 + margin, 1.0, 0.0, -1.0, -3.0 - margin]))
                planes[i, :3] = nrm
                planes[i, 3] = dist - float(np.dot(nrm, centre))
            counts = []
            for shortcut in (True, False):
                ctx.debug_set_frustum_shortcut(shortcut)
                ctx.debug_update_frustum_planes(planes, margin)
                counts.append(ctx.info().visible_nodes)
                if shortcut:
                    was_proven = ctx.debug_last_frustum_update_proven()
                else:
                    assert not ctx.debug_last_frustum_update_proven()
            assert counts[0] == counts[1], f"trial {trial}: {counts} survivors with / without the host-side proof (margin {margin})"
            assert not was_proven or counts[1] == n
            proven += was_proven
            all_visible += counts[1] == n
            some_culled += counts[1] < n
        assert proven >= 3 and all_visible > proven and some_culled >= 5, (proven, all_visible, some_culled)
    finally:
        ctx.debug_set_frustum_shortcut(True)
        ctx.update_frustum(np.eye(4, dtype=np.float32).reshape(16), 45.0, 1.0, enable=False)


@pytest.mark.parametrize("first_survivor", ["solid leaf", "internal node"])
def test_culled_root_with_surviving_descendants_follows_the_reference(ctx, orc, first_survivor):
    """A7, literally (RayTracerBVH.cpp:765-812): when the frustum test drops the ROOT but keeps descendants, the reference's
    compacted array starts with whatever visible node comes first, and its traversal starts there.  Nested boxes make that
    impossible in exact arithmetic; in float it needs a plane through a corner where the root's nodeMax and a child's differ
    by an ulp (gridMin + 0*vs + size*vs vs (gridMin + x*vs) + half*vs).  The debug hook injects such a plane; every
    kernel choice must then render what the oracle renders from the compacted array: the default kernel starts its traversal at
    the node the update recorded on the device (StartState: a leaf -> one pop; an internal node -> its descriptor), the A/B
    kernels hand the frame to the generic kernel over the compacted array."""
    rng = np.random.default_rng(4)
    found = None
    for _ in range(4000):
        dim = 16
        gmin = rng.uniform(-60, 60, 3).astype(np.float32)
        vs = np.float32(rng.choice([0.7, 3.3, 0.37, 1.9])) if first_survivor == "solid leaf" else np.float32(rng.uniform(0.2, 4.0))
        for a in range(3):
            root_mx = np.float32(np.float32(gmin[a] + np.float32(0) * vs) + np.float32(dim) * vs)
            child_mx = np.float32(np.float32(gmin[a] + np.float32(dim // 2) * vs) + np.float32(dim // 2) * vs)
            # the internal-node variant also needs the far quarter (x = 12, size 4) to stick out of the root: its solid leaves must survive too
            quarter_mx = np.float32(np.float32(gmin[a] + np.float32(12) * vs) + np.float32(4) * vs)
            if child_mx > root_mx and (first_survivor == "solid leaf" or quarter_mx > root_mx):
                found = (gmin, vs, a, root_mx, child_mx if first_survivor == "solid leaf" else min(child_mx, quarter_mx))
                break
        if found:
            break
    assert found, "no grid origin / voxel size with the rounding difference found"
    gmin, vs, a, root_mx, child_mx = found
    data = (rng.random((16, 16, 16)) < 0.35).astype(np.uint8)
    # the survivors are the root's children on the far side of `a`; the first of them in BFS order (octant 1 << a) becomes
    # index 0 of the compacted array, where the reference's traversal starts: make it a solid leaf so that there is something to see
    sl = [slice(0, 8)] * 3                               # data is [z][y][x]
    sl[2 - a] = slice(8, 16)
    if first_survivor == "solid leaf":
        data[tuple(sl)] = 1
    else:                                                # ... or a mixed cell (near half noise, far half solid): the traversal starts at an internal node
        data[tuple(sl)] = (rng.random((8, 8, 8)) < 0.5).astype(np.uint8)
        sl[2 - a] = slice(12, 16)
        data[tuple(sl)] = 1
    g = orc.Grid((16, 16, 16), gmin, vs, data)
    nodes = orc.build_flat_octree(g)
    s = Scene(g, nodes)
    upload(ctx, s)
    planes = np.zeros((6, 4), np.float32)
    planes[:, 3] = 1.0                                  # five planes that keep everything: 0*x + 0*y + 0*z + 1 >= 0
    planes[0, :3] = 0.0
    planes[0, a] = 1.0
    planes[0, 3] = -child_mx                            # n = +axis: the positive vertex is nodeMax[a] (+ margin 0)
    want_nodes, vis = orc.cull_compact_planes(nodes, gmin, vs, planes, 0.0)
    assert not vis[0] and vis.any(), "the construction must cull the root and keep descendants"
    assert (want_nodes[0]["isLeaf"] == 1) == (first_survivor == "solid leaf")
    ctx.debug_update_frustum_planes(planes, 0.0)
    assert ctx.info().visible_nodes == len(want_nodes)
    assert ctx.download_visible_nodes().tobytes() == want_nodes.tobytes()
    ext = np.float32(16) * vs
    cam = orc.Camera(0.6, 0.8, float(2.2 * ext))
    cam.set_target(*[float(x) for x in (gmin + np.float32(0.5) * ext)])
    W, H = 200, 150
    view, pos = cam.get_view(), cam.get_pos()
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = orc.render(want_nodes, gmin, vs, view, pos, W / H, 45.0, W, H)
    assert st["hits"] > 0, "the surviving subtree must be visible from this camera"
    try:
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            assert_bit_exact(ctx.render_host(f), want, f"culled root, {kname}")
            gs = ctx.frame_stats(f)
            assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), kname
    finally:
        ctx.set_kernel(rto.KERNEL_AUTO)
        ctx.update_frustum(view, 45.0, W / H, enable=False)


def test_frustum_update_reads_nothing_back_and_can_be_captured(ctx, orc, scenes, camera):
    """rto_update_frustum on a canonical tree is ONE kernel on the context's stream and reads nothing back: where traversals
    start and how many nodes survived stay on the device (the traversal kernels read them there; rto_octree_info_get fetches
    the count when asked).  So `update + render` -- the call main.cpp:1357-1363 makes every frame -- can be stream-captured
    and replayed; plain updates in between do not confuse a replay, and the count follows the state the replay left."""
    torch = pytest.importorskip("torch")
    cal = scenes("calgary")
    W, H = 480, 270
    aspect = W / H
    upload(ctx, cal)
    cams = [camera("calgary_oblique"), (orc.Camera(0.9, 2.5, 3000.0).get_view(), orc.Camera(0.9, 2.5, 3000.0).get_pos())]
    wants, counts, frames = [], [], []
    for view, pos in cams:
        nodes_c, vis = orc.cull_compact(cal.nodes, cal.min, cal.voxel, view, 45.0, aspect)
        wants.append(orc.render(nodes_c, cal.min, cal.voxel, view, pos, aspect, 45.0, W, H)[0])
        counts.append(len(nodes_c))
        frames.append(rto.make_frame(view, pos, aspect, 45.0, W, H))
    try:
        ctx.update_frustum(cams[0][0], 45.0, aspect, enable=True)          # the first update of an octree allocates: outside the capture
        ctx.render_resident(frames[0])
        assert_bit_exact(ctx.download_resident(), wants[0], "update + resident render, camera 0")
        assert ctx.info().visible_nodes == counts[0]
        s = torch.cuda.ExternalStream(ctx.stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.update_frustum(cams[1][0], 45.0, aspect, enable=True)
            ctx.render_resident(frames[1])
        for _ in range(2):
            ctx.update_frustum(cams[0][0], 45.0, aspect, enable=True)      # a plain update + frame in between
            ctx.render_resident(frames[0])
            assert_bit_exact(ctx.download_resident(), wants[0], "plain update between replays")
            assert ctx.info().visible_nodes == counts[0]
            with torch.cuda.stream(s):
                g.replay()
            assert_bit_exact(ctx.download_resident(), wants[1], "replayed update + render, camera 1")
            assert ctx.info().visible_nodes == counts[1], "the count is fetched from the device when asked for"
        del g
        # the A/B kernels take the update's result from the host's copy (fetched on demand): same frame
        ctx.update_frustum(cams[0][0], 45.0, aspect, enable=True)
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            assert_bit_exact(ctx.render_host(frames[0]), wants[0], f"after an asynchronous update, {kname}")
    finally:
        ctx.set_kernel(rto.KERNEL_AUTO)
        ctx.update_frustum(cams[0][0], 45.0, aspect, enable=False)
    # a context whose first update would have to allocate refuses inside a capture instead of breaking it
    fresh = rto.Context(0)
    try:
        fresh.upload_octree(scenes("sphere16").nodes, scenes("sphere16").min, scenes("sphere16").voxel)
        s2 = torch.cuda.ExternalStream(fresh.stream)
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(rto.RtoError) as e:
            with torch.cuda.graph(g2, stream=s2):
                fresh.update_frustum(cams[0][0], 45.0, aspect, enable=True)
        assert e.value.code == hip.RTO_E_UNSUPPORTED
    finally:
        fresh.close()


def test_frames_on_other_streams_wait_for_the_frustum_update(ctx, orc, scenes, camera):
    """rto_update_frustum returns as soon as its kernel is queued on the CONTEXT's stream.  A frame launched right afterwards
    on another stream -- a caller's, or rto_comm's render stream: what RayTracerBVH::renderSceneComputeWithCulling does on
    several GPUs -- must still see the new visibility state, not the previous update's (an event behind the update, a stream
    wait in front of every such launch).  The update is held back here behind a queue of large frames on the context's
    stream, so that a launch without the wait would provably run first."""
    torch = pytest.importorskip("torch")
    cal = scenes("calgary")
    W, H = 480, 270
    aspect = W / H
    upload(ctx, cal)
    camA = camera("calgary_oblique")
    cB = orc.Camera(0.9, 2.5, 3000.0)
    camB = (cB.get_view(), cB.get_pos())
    nodesB, _ = orc.cull_compact(cal.nodes, cal.min, cal.voxel, camB[0], 45.0, aspect)
    nodesA, _ = orc.cull_compact(cal.nodes, cal.min, cal.voxel, camA[0], 45.0, aspect)
    assert len(nodesA) != len(nodesB)
    fB = rto.make_frame(camB[0], camB[1], aspect, 45.0, W, H)
    fA = rto.make_frame(camA[0], camA[1], aspect, 45.0, W, H)
    wantB = orc.render(nodesB, cal.min, cal.voxel, camB[0], camB[1], aspect, 45.0, W, H)[0]
    wantA_in_B = orc.render(nodesB, cal.min, cal.voxel, camA[0], camA[1], aspect, 45.0, W, H)[0]
    assert wantA_in_B.tobytes() != orc.render(nodesA, cal.min, cal.voxel, camA[0], camA[1], aspect, 45.0, W, H)[0].tobytes(), \
        "the two visibility states must give different pictures for this test to mean anything"
    big = rto.make_frame(camA[0], camA[1], 3840 / 2160, 45.0, 3840, 2160)
    other = torch.cuda.Stream()
    out = torch.full((2, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
    comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        ctx.update_frustum(camA[0], 45.0, aspect, enable=True)
        ctx.render_resident(big)
        ctx.synchronize()
        for _ in range(12):                                     # ~1 ms of work queued in front of the update
            ctx.render_resident(big)
        ctx.update_frustum(camB[0], 45.0, aspect, enable=True)
        ctx.render_device(fB, out[0].data_ptr(), None, other.cuda_stream)
        ctx.render_device(fA, out[1].data_ptr(), None, other.cuda_stream)
        torch.cuda.synchronize()
        assert_bit_exact(out[0].cpu().numpy(), wantB, "foreign stream, the update's own camera")
        assert_bit_exact(out[1].cpu().numpy(), wantA_in_B, "foreign stream, another camera through the new visibility state")
        # the same through rto_comm's render stream (one-rank communicator, and rehearsing rank 1 of 2)
        for world, rank in ((0, 0), (2, 1)):
            ctx.update_frustum(camA[0], 45.0, aspect, enable=True)
            ctx.synchronize()
            comm.debug_rehearse(world, rank)
            for _ in range(12):
                ctx.render_resident(big)
            ctx.update_frustum(camB[0], 45.0, aspect, enable=True)
            out.fill_(7.0)
            comm.submit(hip.Context.frame_array([fB, fA]), out.data_ptr(), out.stride(0) * 4)
            comm.flush()
            rows = np.ones(H, bool) if world == 0 else ((np.arange(H) // 16) % world == rank)
            assert_bit_exact(out[0].cpu().numpy()[rows], wantB[rows], f"rto_comm (rehearsing {rank} of {world}), the update's camera")
            assert_bit_exact(out[1].cpu().numpy()[rows], wantA_in_B[rows], f"rto_comm (rehearsing {rank} of {world}), another camera")
        # switching culling off rewrites the descriptors as well
        for _ in range(12):
            ctx.render_resident(big)
        ctx.update_frustum(camB[0], 45.0, aspect, enable=False)
        ctx.render_device(fA, out[0].data_ptr(), None, other.cuda_stream)
        torch.cuda.synchronize()
        assert_bit_exact(out[0].cpu().numpy(), oracle_frame(orc, cal, camA[0], camA[1], W, H)[0], "foreign stream after culling was switched off")
    finally:
        comm.debug_rehearse(0)
        comm.close()
        ctx.update_frustum(camA[0], 45.0, aspect, enable=False)


def test_occupancy_mask_never_changes_pixels_and_removes_work(ctx, orc, scenes):
    """The occupancy mask (mask_block: built by the first workgroups of every colour / shade launch of the default kernels) is a scheduling device:
    frames with it, without it and the oracle's are bit-identical -- cameras outside, grazing, far away, with the scene partly
    or wholly off the screen, and inside the geometry (cells around the eye: the "whole frame" word) -- while the number of
    tiles whose wave walks the tree drops to little more than the tiles that contain a hit."""
    torch = pytest.importorskip("torch")
    for scene, cams in (("sphere64", [(0.5, 0.7, 1.8, (0, 0, 0), 45.0), (0.5, 0.7, 1.8, (0.8, 0.3, 0.0), 45.0), (0.5, 0.7, 6.0, (0, 0, 0), 10.0),
                                      (0.5, 0.7, 0.3, (0, 0, 0), 45.0), (0.1, 0.2, 0.9, (0.0, 0.9, 0.0), 70.0), (0.5, 0.7, 1.8, (5.0, 5.0, 9.0), 45.0),
                                      (-1.2, 4.0, 0.75, (0.2, -0.1, 0.1), 120.0)]),
                        ("calgary", [(0.6, 0.5, 3500.0, None, 45.0), (1.2, 0.1, 900.0, None, 60.0)]),
                        ("odd", [(0.4, 0.9, 9.0, None, 45.0)])):
        s = scenes(scene)
        upload(ctx, s)
        level, ncells = ctx.debug_tile_mask_info()
        assert level >= 1 and 0 < ncells <= 8192, (scene, level, ncells)
        W, H = 640, 360
        for (t, p, r, tgt, fov) in cams:
            cam = orc.Camera(t, p, r)
            if tgt is not None:
                cam.set_target(*[float(x) for x in tgt])
            view, pos = cam.get_view(), cam.get_pos()
            f = rto.make_frame(view, pos, W / H, fov, W, H)
            want, st = oracle_frame(orc, s, view, pos, W, H, fov=fov)
            near_rgba, near_t = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H, nthreads=min(8, orc.max_threads()))
            for on in (2, 1, 0):
                ctx.debug_set_tile_mask(on)
                for _ in range(2):
                    got = ctx.render_host(f)
                assert_bit_exact(got, want, f"{scene} cam {(t, p, r, tgt, fov)} mask mode {on}")
                for _ in range(3):                               # the nearest-hit mode shares the launch geometry, the mask and the order tables
                    nrgba, nt = ctx.render_skip_host(f)
                assert nt.tobytes() == near_t.tobytes(), f"{scene} cam {(t, p, r, tgt, fov)} nearest-hit distances, mask mode {on}"
                assert_bit_exact(nrgba, near_rgba, f"{scene} cam {(t, p, r, tgt, fov)} nearest-hit colours, mask mode {on}")
                out = torch.full((2, H, W, 4), 7.0, dtype=torch.float32, device="cuda")          # the batch kernel and a 3-way partition too
                ctx.render_batch_device(hip.Context.frame_array([f, f]), out.data_ptr(), out.stride(0) * 4, None, False, 0)
                torch.cuda.synchronize()
                assert_bit_exact(out[1].cpu().numpy(), want, f"{scene} batched, mask mode {on}")
                part = hip.Partition(3, 1, 16)
                rows = partition_row_map(H, 3, 1, 16)
                pb = torch.full((len(rows), W, 4), 7.0, dtype=torch.float32, device="cuda")
                ctx.render_device(f, pb.data_ptr(), part)
                ctx.synchronize()
                assert_bit_exact(pb.cpu().numpy(), want[rows], f"{scene} part 1/3, mask mode {on}")
            ctx.debug_set_tile_mask(2)
            gs = ctx.frame_stats(f)                              # instrumented frames never use the mask: exact pops
            assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])
            if scene == "sphere64" and tgt == (0, 0, 0) and r == 1.8:
                ctx.render_host(f)
                cost = ctx.debug_tile_cost().reshape((H + 7) // 8, (W + 7) // 8)
                hit_tiles = (want[..., 0] != 0)[: H // 8 * 8].reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
                assert (cost[: H // 8, : W // 8][hit_tiles] > 0).all(), "a tile with a hit is never masked out"
                ys, xs = np.nonzero(hit_tiles)                    # inside the hit tiles' bounding box every entry was written by this frame
                inner = cost[ys.min(): ys.max() + 1, xs.min(): xs.max() + 1]
                n_walk, n_hit, n_box = int((inner > 0).sum()), int(hit_tiles.sum()), inner.size
                assert n_hit <= n_walk <= 1.35 * n_hit + 40 and n_walk < n_box, (n_walk, n_hit, n_box)
                # the cost the launch order sorts by is that of the tile's BUSIEST ray: a loop trip pops at most 8 nodes, so the
                # tile's trip count bounds every pixel's pop count (lane 0's count alone, recorded until round 3, did not)
                pops = np.abs(ctx.render_steps(f))[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
                ctx.render_host(f)
                cost = ctx.debug_tile_cost().reshape((H + 7) // 8, (W + 7) // 8)[: H // 8, : W // 8]
                walked = hit_tiles & (cost > 0)
                assert (8 * cost[walked] + 1 >= pops[walked]).all(), "a tile's recorded cost is below what its busiest ray needs"
    ctx.debug_set_tile_mask(1)


def test_work_less_tiles_next_to_work_are_marked_for_the_launch_order(ctx, orc, scenes):
    """The rim (tile_may_hit): a tile outside the occupancy mask records cost -1 instead of 0 when the mask covers a tile within two
    tiles of it, and k_order_build starts such tiles ahead of the certainly empty ones -- where a camera in motion finds its new
    work.  Mask mode 2 (complete before the frame): every tile within two tiles of a tile whose wave walked has a non-zero cost,
    far tiles keep 0, the table stays a permutation, and frames of a camera that moves on stay bit-exact."""
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 1280, 720
    tX, tY = W // 8, H // 8
    try:
        ctx.debug_set_tile_mask(2)
        cams = [orc.Camera(0.5 + 0.02 * i, 0.7, 2.2) for i in range(6)]
        for i, cam in enumerate(cams):
            view, pos = cam.get_view(), cam.get_pos()
            f = rto.make_frame(view, pos, W / H, 45.0, W, H)
            want, _ = oracle_frame(orc, s, view, pos, W, H)
            assert_bit_exact(ctx.render_host(f), want, f"camera {i} of a moving sequence")
            assert ctx.debug_sort_violations() == 0
            cost = ctx.debug_tile_cost().reshape(tY, tX)
            walked = cost > 0
            assert walked.any()
            near = np.zeros_like(walked)
            ys, xs = np.nonzero(walked)
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    yy, xx = np.clip(ys + dy, 0, tY - 1), np.clip(xs + dx, 0, tX - 1)
                    near[yy, xx] = True
            # inside the launch box every tile's cost was written by this frame; the box contains the walked tiles' neighbourhood
            # except where it ends: compare inside the walked tiles' own bounding box
            y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
            inner = np.zeros_like(walked); inner[y0:y1 + 1, x0:x1 + 1] = True
            assert (cost[near & inner] != 0).all(), f"camera {i}: a tile next to work recorded cost 0"
            assert ((cost == -1) & inner).any(), f"camera {i}: no rim tile at all"
    finally:
        ctx.debug_set_tile_mask(1)


def test_persistent_kernel_in_a_graph_with_an_odd_frame_count(ctx, orc, scenes):
    """The persistent-threads kernel takes launch slots from a global counter; every launch zeroes its own counter with a
    memset node, so a captured sequence of ANY length replays exactly (an odd number of frames used to leave the next
    replay an exhausted counter).  Buffers are poisoned between replays."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 640, 360
    cams = [orc.Camera(0.5 + 0.25 * i, 0.7, 1.8) for i in range(3)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
    stream = torch.cuda.Stream()
    try:
        ctx.set_kernel(rto.KERNEL_PACKED_PERSISTENT)
        for _ in range(3):
            ctx.render_device(frames[0], bufs[0].data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                for k in range(7):                                   # odd
                    ctx.render_device(frames[k % 3], bufs[k % 3].data_ptr(), None, stream.cuda_stream)
            for rep in range(3):
                for b in bufs:
                    b.fill_(7.0)
                g.replay()
                torch.cuda.synchronize()
                for i in range(3):
                    assert_bit_exact(bufs[i].cpu().numpy(), wants[i], f"persistent kernel, replay {rep}, camera {i}")
        assert ctx.debug_sort_violations() == 0
    finally:
        ctx.set_kernel(rto.KERNEL_AUTO)


def test_frustum_update_that_culls_nothing_and_everything(ctx, orc, scenes, camera):
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    ctx.update_frustum(view, 45.0, 1.0, enable=True)           # unit scene, 150-unit margin: all visible
    assert ctx.info().visible_nodes == len(s.nodes)
    assert ctx.download_visible_nodes().tobytes() == s.nodes.tobytes()
    # looking away from a far-off scene culls every node: the reference then dispatches over an empty SSBO
    far = orc.Camera(0.0, 0.0, 10.0)
    far._c.target[2] = 9000.0
    ctx.update_frustum(far.get_view(), 45.0, 1.0, enable=True)
    assert ctx.info().visible_nodes == 0
    img = ctx.render_host(rto.make_frame(view, pos, 1.0, 45.0, 32, 32))
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    ctx.update_frustum(view, 45.0, 1.0, enable=False)


def test_partitions_assemble_to_the_full_frame(ctx, orc, scenes, camera):
    """Multi-GPU decomposition on one GPU: every part renders its bands into a compact buffer, the buffers
    are laid end to end as a gather would, rto_assemble_device rebuilds the frame."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    view, pos = camera("sphere")
    upload(ctx, s)
    for (W, H) in ((640, 360), (333, 250)):
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        want = ctx.render_host(f)
        for nparts, band in ((2, 16), (3, 8), (4, 16), (8, 8), (8, 64)):
            rows0 = ctx.partition_rows(f, hip.Partition(nparts, 0, band))
            gathered = torch.zeros((nparts, rows0, W, 4), dtype=torch.float32, device="cuda")
            total = 0
            for p in range(nparts):
                part = hip.Partition(nparts, p, band)
                total += ctx.partition_rows(f, part)
                ctx.render_device(f, gathered[p].data_ptr(), part)
            assert total == H
            frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
            ctx.assemble_device(f, hip.Partition(nparts, 0, band), gathered.data_ptr(), frame.data_ptr())
            ctx.synchronize()      # everything above ran on the default stream (stream handle 0)
            assert_bit_exact(frame.cpu().numpy(), want, f"{W}x{H} parts={nparts} band={band}")


def test_shade_payload_assembles_to_the_same_frame(ctx, orc, scenes, camera):
    """The 4-byte multi-GPU payload: every kernel writes the Lambert term (-1 = miss) per pixel of its part,
    rto_assemble_shade_device finishes the colour expression while re-interleaving -- bit-identical to the
    RGBA32F path and to the oracle, whole-frame and split 2/3/8 ways."""
    torch = pytest.importorskip("torch")

    for scene, cam in (("sphere64", "sphere"), ("calgary", "calgary_oblique")):
        s = scenes(scene)
        view, pos = camera(cam)
        upload(ctx, s)
        W, H = 417, 250
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        want, _ = oracle_frame(orc, s, view, pos, W, H)
        for kname, kernel in KERNELS:
            ctx.set_kernel(kernel)
            for nparts, band in ((1, 16), (2, 16), (3, 8), (8, 24)):
                rows0 = ctx.partition_rows(f, hip.Partition(nparts, 0, band)) if nparts > 1 else H
                gathered = torch.full((nparts, rows0, W), 7.0, dtype=torch.float32, device="cuda")
                for p in range(nparts):
                    ctx.render_shade_device(f, gathered[p].data_ptr(), hip.Partition(nparts, p, band) if nparts > 1 else None)
                frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                ctx.assemble_shade_device(f, hip.Partition(nparts, 0, band), gathered.data_ptr(), frame.data_ptr())
                ctx.synchronize()
                assert_bit_exact(frame.cpu().numpy(), want, f"{scene} {kname} shade payload, parts={nparts}")
                if nparts == 1:
                    sh = gathered[0].cpu().numpy()
                    hit = want[..., 0] != 0
                    assert ((sh == -1.0) == ~hit).all() and (sh[hit] >= 0).all()
        ctx.set_kernel(rto.KERNEL_AUTO)


@pytest.mark.parametrize("scene,cam0,dtheta", [("sphere64", (0.5, 0.7, 1.8), 0.21), ("calgary", (0.6, 0.5, 3500.0), 0.35)])
def test_temporal_launch_order_is_only_a_schedule(ctx, orc, scenes, scene, cam0, dtheta):
    """The packed kernel launches its tiles in the order of an EARLIER frame's trip counts.  Whatever that table
    holds -- stale (camera moved a lot), rebuilt every frame or every 4th, built for another image size or another
    partition, or a hostile caller-supplied permutation -- every tile is rendered exactly once and the pixels are
    the oracle's.  Frames go into a buffer pre-filled with a sentinel so that a skipped tile cannot hide."""
    torch = pytest.importorskip("torch")
    s = scenes(scene)
    upload(ctx, s)
    try:
        for period in (1, 2, 4):
            ctx.set_launch_order(1, period)
            frame_no = 0
            for (W, H) in ((400, 240), (333, 250), (400, 240)):
                buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                for k in range(6):
                    c = orc.Camera(cam0[0] + dtheta * frame_no, cam0[1] + 0.03 * frame_no, cam0[2])
                    frame_no += 1
                    view, pos = c.get_view(), c.get_pos()
                    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
                    want, _ = oracle_frame(orc, s, view, pos, W, H)
                    buf.fill_(7.0)
                    ctx.render_device(f, buf.data_ptr())
                    ctx.synchronize()
                    assert_bit_exact(buf.cpu().numpy(), want, f"{scene} period {period} {W}x{H} frame {k}")
                    if k == 3:      # a partition render in between shares (and invalidates) the table
                        part = hip.Partition(2, 1, 16)
                        rows = ctx.partition_rows(f, part)
                        pb = torch.full((rows, W, 4), 7.0, dtype=torch.float32, device="cuda")
                        ctx.render_device(f, pb.data_ptr(), part)
                        ctx.synchronize()
                        assert_bit_exact(pb.cpu().numpy(), want[partition_row_map(H, 2, 1, 16)], f"{scene} part 1/2 mid-sequence")
        # caller-supplied tables: reversed and random permutations
        W, H = 400, 240
        c = orc.Camera(*cam0)
        view, pos = c.get_view(), c.get_pos()
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        want, _ = oracle_frame(orc, s, view, pos, W, H)
        ctx.set_launch_order(1, 1)
        ctx.render_host(f)
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        assert len(ctx.debug_tile_cost()) == tiles
        rng = np.random.default_rng(5)
        for order in (np.arange(tiles)[::-1], rng.permutation(tiles)):
            ctx.debug_set_tile_order(order)
            buf = torch.full((H, W, 4), 7.0, dtype=torch.float32, device="cuda")
            ctx.render_device(f, buf.data_ptr())
            ctx.render_device(f, buf.data_ptr())
            ctx.synchronize()
            assert_bit_exact(buf.cpu().numpy(), want, f"{scene} caller-supplied order")
        ctx.debug_set_tile_order(None)
        assert_bit_exact(ctx.render_host(f), want, f"{scene} automatic order restored")
        # centre-out policy
        ctx.set_launch_order(0)
        assert_bit_exact(ctx.render_host(f), want, f"{scene} centre-out")
    finally:
        ctx.debug_set_tile_order(None)
        ctx.set_launch_order(1, 8)


def test_8k_frame_with_temporal_order(ctx, orc, scenes, camera):
    """7680x4320 = 518,400 tiles: the launch-order sort runs with 507 blocks (its table still fits LDS); three frames
    so that the third one is launched through a rebuilt table.  A 16K frame exceeds that and must fall back silently."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    view, pos = camera("sphere")
    upload(ctx, s)
    try:
        ctx.set_launch_order(1, 1)
        W, H = 7680, 4320
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        want, st = oracle_frame(orc, s, view, pos, W, H)
        buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        for k in range(3):
            buf.fill_(7.0)
            ctx.render_device(f, buf.data_ptr())
            ctx.synchronize()
            assert_bit_exact(buf.cpu().numpy(), want, f"8K frame {k}")
        assert len(ctx.debug_tile_cost()) == (W // 8) * (H // 8)
        del buf
        W, H = 15360, 8640                                   # 2,073,600 tiles: no table, centre-out order
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        big = torch.full((H, W, 4), 7.0, dtype=torch.float32, device="cuda")
        ctx.render_device(f, big.data_ptr())
        ctx.render_device(f, big.data_ptr())
        ctx.synchronize()
        # compare a band through the sphere (the oracle renders rows on request)
        out = np.zeros((H, W, 4), np.float32)
        orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=min(16, orc.max_threads()), rows=(4000, 4640), out=out)
        got = big[4000:4640].cpu().numpy()
        assert_bit_exact(got, out[4000:4640], "16K frame, rows 4000..4639")
        assert float(big.min()) == 0.0 and float(big[..., 3].min()) == 1.0     # every pixel was written
    finally:
        ctx.set_launch_order(1, 8)


def test_frames_in_flight_on_several_streams_of_one_context(ctx, orc, scenes):
    """Three HIP streams render interleaved frames of one context without any host sync in between; every stream
    owns its launch-order tables (rebuilt after every frame here), so nothing is shared between frames in flight."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 640, 360
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    cams = [orc.Camera(0.5 + 0.3 * i, 0.7, 1.8 + 0.1 * i) for i in range(3)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    try:
        for period in (1, 3):
            ctx.set_launch_order(1, period)
            for b in bufs:
                b.fill_(7.0)
            torch.cuda.synchronize()
            for k in range(60):
                i = k % 3
                ctx.render_device(frames[i], bufs[i].data_ptr(), None, streams[i].cuda_stream)
            torch.cuda.synchronize()
            for i in range(3):
                assert_bit_exact(bufs[i].cpu().numpy(), wants[i], f"stream {i}, period {period}")
            assert ctx.debug_sort_violations() == 0
    finally:
        ctx.set_launch_order(1, 8)


def test_mask_handoff_under_uneven_concurrent_frames(ctx, orc, scenes):
    """The occupancy mask is handed from the first workgroups of a launch to the other waves of the SAME launch, across CUs and
    XCDs (mask_block -> tile_may_hit), and a reader that saw "complete" but a stale tile word would paint a black tile over
    geometry.  The shape /opt/skills/guides/MI355X_MICROARCH.md asks such a hand-off to be tested in: a busy, unevenly loaded
    chip and warm consumers -- frames of DIFFERENT cameras (cheap, costly, the eye inside the shell, the sphere half off the
    screen) in flight on three streams of one context, mask mode 1 (built inside the launch), every stream changing camera from
    frame to frame so that the words a wave reads were last written for another view -- and EVERY pixel of EVERY frame compared
    with the oracle, not just the last frame per stream."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere256")
    upload(ctx, s)
    ctx.debug_set_tile_mask(1)
    W, H = 1920, 1080
    specs = [(0.5, 0.7, 1.8, None, 45.0), (2.1, 0.4, 1.1, None, 45.0), (0.3, 1.2, 3.5, None, 30.0), (1.0, 0.9, 0.25, None, 70.0),
             (0.5, 0.7, 1.8, (0.9, 0.2, 0.0), 45.0), (4.0, 0.5, 0.9, (0.0, 0.6, 0.0), 60.0)]
    frames, wants = [], []
    for (t, p, r, tgt, fov) in specs:
        cam = orc.Camera(t, p, r)
        if tgt is not None:
            cam.set_target(*[float(x) for x in tgt])
        frames.append(rto.make_frame(cam.get_view(), cam.get_pos(), W / H, fov, W, H))
        wants.append(oracle_frame(orc, s, cam.get_view(), cam.get_pos(), W, H, fov=fov)[0])
    rounds, nstreams = 8, 3
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    bufs = torch.full((nstreams, rounds, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for j in range(rounds):                      # no host wait anywhere: 24 frames queued on three streams
        for i in range(nstreams):
            ctx.render_device(frames[(i * 2 + j) % len(frames)], bufs[i, j].data_ptr(), None, streams[i].cuda_stream)
    torch.cuda.synchronize()
    for i in range(nstreams):
        for j in range(rounds):
            k = (i * 2 + j) % len(frames)
            assert_bit_exact(bufs[i, j].cpu().numpy(), wants[k], f"stream {i}, frame {j} (camera {specs[k]})")
    assert ctx.debug_sort_violations() == 0


def test_several_frames_in_one_launch(ctx, orc, scenes):
    """rto_render_batch_device: up to 8 frames per kernel launch (k_trace_lean_batch), every frame with its own camera --
    hence its own rectangle, box and launch order.  11 frames = a launch of 8 and one of 3; whole frames in colour, one
    part of a 3-way split as the 4-byte payload; repeated so that the shared launch-order table is rebuilt in between;
    the other kernel choices take the same call as a sequence of launches."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 417, 250
    cams = [orc.Camera(0.2 + 0.45 * i, 0.3 + 0.11 * i, 1.5 + 0.07 * i) for i in range(11)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    arr = hip.Context.frame_array(frames)
    stream = torch.cuda.Stream()
    out = torch.empty((11, H, W, 4), dtype=torch.float32, device="cuda")
    try:
        for kernel in (rto.KERNEL_AUTO, rto.KERNEL_PACKED_V3, rto.KERNEL_GENERIC):
            ctx.set_kernel(kernel)
            for rep in range(3 if kernel == rto.KERNEL_AUTO else 1):
                ctx.set_launch_order(1, 1 + rep)
                out.fill_(7.0)
                torch.cuda.synchronize()
                ctx.render_batch_device(arr, out.data_ptr(), out.stride(0) * 4, None, False, stream.cuda_stream)
                torch.cuda.synchronize()
                for i in range(11):
                    assert_bit_exact(out[i].cpu().numpy(), wants[i], f"batched frame {i}, kernel {kernel}, repeat {rep}")
        ctx.set_kernel(rto.KERNEL_AUTO)
        part = hip.Partition(3, 1, 16)
        rows = ctx.partition_rows(frames[0], part)
        shade = torch.full((11, rows, W), 7.0, dtype=torch.float32, device="cuda")
        ctx.render_batch_device(arr, shade.data_ptr(), shade.stride(0) * 4, part, True, stream.cuda_stream)
        torch.cuda.synchronize()
        owner = (np.arange(H) // 16) % 3
        for i in range(11):
            one = torch.empty((rows, W), dtype=torch.float32, device="cuda")
            ctx.render_shade_device(frames[i], one.data_ptr(), part, stream.cuda_stream)
            torch.cuda.synchronize()
            assert one.cpu().numpy().tobytes() == shade[i].cpu().numpy().tobytes(), f"batched part, frame {i}"
        assert int((owner == 1).sum()) == rows
        assert ctx.debug_sort_violations() == 0
        # config 5's kernel takes batches the same way (rto_render_triangles_batch_device): 5 cameras, with and without shadow
        wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
        ctx.build_leaf_triangles(s.grid.data)
        arr5 = hip.Context.frame_array(frames[:5])
        for shadow in (True, False):
            out[:5].fill_(7.0)
            torch.cuda.synchronize()
            ctx.render_triangles_batch_device(arr5, out.data_ptr(), out.stride(0) * 4, shadow, None, False, stream.cuda_stream)
            torch.cuda.synchronize()
            for i in range(5):
                want, _ = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, cams[i].get_view(), cams[i].get_pos(), W / H, 45.0, W, H, shadow=shadow)
                assert_bit_exact(out[i].cpu().numpy(), want, f"batched triangle frame {i}, shadow {shadow}")
    finally:
        ctx.set_kernel(rto.KERNEL_AUTO)
        ctx.set_launch_order(1, 8)


def test_resident_frame_stays_on_the_gpu(ctx, orc, scenes, camera):
    """rto_render_resident leaves the frame in the context's device buffer (the reference's texture is never read
    back either); rto_download_resident / the device pointer give the oracle's pixels, for the octree and triangle paths."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 320, 200
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    for _ in range(3):
        ctx.render_resident(f)
    assert_bit_exact(ctx.download_resident(), want, "resident octree frame")
    ptr, w, h = ctx.resident_frame()
    assert (w, h) == (W, H) and ptr
    # the device pointer is usable by other device work: re-interleave it as a 1-part "gather"
    frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    ctx.assemble_device(f, hip.Partition(1, 0, 16), ptr, frame.data_ptr(), ctx.stream)
    ctx.synchronize()
    assert_bit_exact(frame.cpu().numpy(), want, "resident frame through its device pointer")
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    ctx.build_leaf_triangles(s.grid.data)
    for mode, shadow in ((1, False), (2, True)):
        wtri, _ = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=shadow)
        ctx.render_resident(f, mode)
        assert_bit_exact(ctx.download_resident(), wtri, f"resident triangle frame, mode {mode}")
    ctx.render_host(f)                       # any other user of the buffer invalidates the resident frame
    with pytest.raises(rto.RtoError):
        ctx.resident_frame()
    with pytest.raises(rto.RtoError):
        ctx.render_resident(f, 7)


def test_frames_captured_in_a_hip_graph_replay_exactly(ctx, orc, scenes):
    """rto_render_device is stream-capturable after a warm-up frame (no allocation, no synchronisation): 24 frames of 6
    cameras captured once (more frames than a launch-order period, an odd number of would-be rebuilds) and replayed
    three times give the oracle's pixels every time.  Captured launches keep a launch-order table of their own and every
    capture starts with a rebuild node, so a replay depends on nothing outside its graph: plain frames whose tile box has
    ANOTHER size (a camera much closer / much further away), a second graph and the graph's own tail in between replays
    change nothing -- buffers are poisoned before every replay so that a skipped or duplicated tile cannot hide."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 640, 360
    cams = [orc.Camera(0.5 + 0.25 * i, 0.7 + 0.05 * i, 1.8) for i in range(6)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
    stream = torch.cuda.Stream()
    try:
        ctx.set_launch_order(1, 3)
        for _ in range(7):                                   # warm-up on the capture stream: tables, a valid order
            ctx.render_device(frames[0], bufs[0].data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                for k in range(24):
                    ctx.render_device(frames[k % 6], bufs[k % 6].data_ptr(), None, stream.cuda_stream)
            for rep in range(3):
                for b in bufs:
                    b.fill_(7.0)
                g.replay()
                torch.cuda.synchronize()
                for i in range(6):
                    assert_bit_exact(bufs[i].cpu().numpy(), wants[i], f"graph replay {rep}, camera {i}")
        # plain launches afterwards still work and rebuild their table
        for _ in range(5):
            ctx.render_device(frames[1], bufs[1].data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        assert_bit_exact(bufs[1].cpu().numpy(), wants[1], "plain launches after the graph")
        # plain frames with tile boxes of other sizes on the capture stream (each rebuilds the PLAIN table), a second graph of
        # one far-away camera, then the first graph again: every replay must still render its own cameras exactly
        near, far = orc.Camera(0.5, 0.7, 1.1), orc.Camera(0.5, 0.7, 4.0)
        others = []
        for c in (near, far):
            fr = rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H)
            others.append((fr, oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0]))
        scratch = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            for _ in range(4):
                ctx.render_device(others[1][0], scratch.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
            with torch.cuda.graph(g2, stream=stream):
                for _ in range(5):
                    ctx.render_device(others[1][0], scratch.data_ptr(), None, stream.cuda_stream)
            for rep in range(3):
                for fr, want in others:                      # plain launches: boxes larger and smaller than the graphs' frames
                    for _ in range(4):
                        scratch.fill_(7.0)
                        ctx.render_device(fr, scratch.data_ptr(), None, stream.cuda_stream)
                    torch.cuda.synchronize()
                    assert_bit_exact(scratch.cpu().numpy(), want, f"plain frame with another box, round {rep}")
                scratch.fill_(7.0)
                g2.replay()
                torch.cuda.synchronize()
                assert_bit_exact(scratch.cpu().numpy(), others[1][1], f"second graph, round {rep}")
                for b in bufs:
                    b.fill_(7.0)
                g.replay()
                torch.cuda.synchronize()
                for i in range(6):
                    assert_bit_exact(bufs[i].cpu().numpy(), wants[i], f"first graph after foreign frames, round {rep}, camera {i}")
        assert ctx.debug_sort_violations() == 0
    finally:
        ctx.set_launch_order(1, 8)


def test_batched_gather_layout_assembles_every_frame(ctx, orc, scenes):
    """Several frames per collective: every part renders `batch` frames into [batch][rows][width], the gather delivers
    [rank][batch][rows][width], rto_assemble_batch_device rebuilds frame `index` -- both payloads, 3 parts, 4 cameras."""
    torch = pytest.importorskip("torch")

    s = scenes("sphere64")
    upload(ctx, s)
    W, H, nparts, band, batch = 417, 250, 3, 16, 4
    cams = [orc.Camera(0.5 + 0.3 * i, 0.7 - 0.1 * i, 1.8) for i in range(batch)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    rows0 = ctx.partition_rows(frames[0], hip.Partition(nparts, 0, band))
    for shade in (True, False):
        shape = (nparts, batch, rows0, W) if shade else (nparts, batch, rows0, W, 4)
        gathered = torch.full(shape, 7.0, dtype=torch.float32, device="cuda")
        for p in range(nparts):
            for f in range(batch):
                (ctx.render_shade_device if shade else ctx.render_device)(frames[f], gathered[p, f].data_ptr(), hip.Partition(nparts, p, band))
        out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        for f in range(batch):
            ctx.assemble_batch_device(frames[f], hip.Partition(nparts, 0, band), gathered.data_ptr(), batch, f, shade, out.data_ptr())
            ctx.synchronize()
            assert_bit_exact(out.cpu().numpy(), wants[f], f"batched gather, frame {f}, shade={shade}")
    with pytest.raises(rto.RtoError):
        ctx.assemble_batch_device(frames[0], hip.Partition(nparts, 0, band), gathered.data_ptr(), batch, batch, False, out.data_ptr())


def test_render_device_on_a_caller_stream(ctx, scenes, camera):
    torch = pytest.importorskip("torch")
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 256, 128
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want = ctx.render_host(f)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx.render_device(f, out.data_ptr(), None, stream.cuda_stream)
    stream.synchronize()
    assert_bit_exact(out.cpu().numpy(), want, "caller stream")
    assert ctx.last_kernel_ms() > 0


def test_error_codes(ctx, scenes, camera):
    fresh = rto.Context(0)
    view, pos = camera("sphere")
    f = rto.make_frame(view, pos, 1.0, 45.0, 16, 16)
    with pytest.raises(rto.RtoError) as e:
        fresh.render_host(f)
    assert e.value.code == hip.RTO_E_NO_OCTREE and "setOctree" in str(e.value)
    with pytest.raises(rto.RtoError) as e:
        fresh.update_frustum(view, 45.0, 1.0)
    assert e.value.code == hip.RTO_E_NO_OCTREE
    with pytest.raises(rto.RtoError) as e:
        fresh.upload_octree(np.zeros(0, rto.NODE_DTYPE), [0, 0, 0], 1.0)
    assert e.value.code == hip.RTO_E_INVALID
    fresh.upload_octree(scenes("sphere16").nodes, scenes("sphere16").min, scenes("sphere16").voxel)
    with pytest.raises(rto.RtoError) as e:
        fresh.render_host(rto.make_frame(view, pos, 1.0, 45.0, 0, 16))
    assert e.value.code == hip.RTO_E_INVALID
    with pytest.raises(rto.RtoError) as e:
        fresh.render_device(f, 0)
    assert e.value.code == hip.RTO_E_INVALID
    with pytest.raises(rto.RtoError) as e:
        fresh._check(fresh._L.rto_render_device(fresh._h, C.byref(f), C.byref(hip.Partition(2, 0, 12)), C.c_void_p(8), None))
    assert e.value.code == hip.RTO_E_INVALID        # band_rows must be a multiple of 8
    with pytest.raises(rto.RtoError):
        rto.Context(9999)
    fresh.close()


def test_cpp_dropin_class_end_to_end(orc, scenes, camera):
    """The C++ RayTracerBVH (reference call sequence, S/main.cpp:1127-1131, 1357-1363) over the product's
    own VoxelGrid / createOctreeFromVoxelGrid / Camera, checked against the oracle."""
    W, H = 320, 200
    g = rto.VoxelGrid.test_sphere(64)
    root = rto.createOctreeFromVoxelGrid(g)
    rt = rto.RayTracerBVH()
    rt.ensureComputeInitialized()
    rt.setOctree(root, g)
    assert rt.numNodes == 23561
    cam = rto.Camera(0.5, 0.7, 1.8)
    rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, True)
    got = rt.framebuffer()
    s = scenes("sphere64")
    view, pos = camera("sphere")
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    assert_bit_exact(got, want, "C++ drop-in, with culling")
    rt.renderSceneCompute(cam, W, H, W / H, 45.0)
    assert_bit_exact(rt.framebuffer(), want, "C++ drop-in")
    # setOctree before ensureComputeInitialized also works (upload is deferred)
    rt2 = rto.RayTracerBVH()
    rt2.setOctree(root, g)
    rt2.ensureComputeInitialized()
    rt2.renderSceneCompute(cam, W, H, W / H, 45.0)
    assert_bit_exact(rt2.framebuffer(), want, "C++ drop-in, deferred upload")
    # addition: octree built on the GPU from the grid, no pointer tree at all
    rt4 = rto.RayTracerBVH()
    rt4.setOctreeFromGrid(g)
    assert rt4.numNodes == 23561
    rt4.renderSceneCompute(cam, W, H, W / H, 45.0)
    assert_bit_exact(rt4.framebuffer(), want, "C++ drop-in, GPU-built octree")
    # null root: silently nothing (S/RT:439, :625)
    rt3 = rto.RayTracerBVH()
    rt3.ensureComputeInitialized()
    rt3.setOctree(None, g)
    rt3.renderSceneCompute(cam, W, H, W / H, 45.0)
    assert rt3.framebuffer() is None
    rto.freeOctree(root)


def test_plain_cpp_example_renders_config1(tmp_path):
    """examples/render_sphere: main.cpp's call sequence in plain C++ on the product's host layer (no Python in the
    process).  BASELINE config 1 -- 64^3 sphere, 512x512 -- must light the 65,009 pixels SURVEY.md records."""
    import subprocess

    exe = os.path.join(ROOT, "examples", "render_sphere")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    out = tmp_path / "frame.ppm"
    env = dict(os.environ, RTO_HIP_LIB=os.path.join(ROOT, "ray_tracing_octrees_amd", "librto_hip.so"))
    p = subprocess.run([exe, "64", "512", "512", str(out)], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "23561 nodes" in p.stdout and "65009 lit pixels" in p.stdout, p.stdout
    assert out.stat().st_size == len(b"P6\n512 512\n255\n") + 512 * 512 * 3


def test_reference_host_stack_drives_the_hip_path(orc, scenes, camera, tmp_path):
    """oracle/_ref/dropin_test = the REFERENCE's compiled OctreeVoxel.cpp/Camera.cpp (+ its headers and glm) linked
    with this repo's RayTracerBVH.cpp built with -DRTO_REFERENCE_HEADERS; it follows main.cpp's call sequence.
    Built only where /root/reference exists (make -C oracle dropin); the binary travels to the GPU box."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "dropin_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_test not built (needs the reference checkout)")
    W, H = 400, 240
    out = tmp_path / "frame.raw"
    env = dict(os.environ, RTO_HIP_LIB=hip.lib_path())
    p = subprocess.run([exe, "64", str(W), str(H), str(out)], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "23561 nodes" in p.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(H, W, 4)
    s = scenes("sphere64")
    view, pos = camera("sphere")
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    assert_bit_exact(got, want, "reference host stack + HIP path")


@pytest.mark.parametrize("scene", ["sphere32", "odd", "calgary"])
def test_octree_ray_skip_equals_the_reference_vectors(ctx, scenes, golden, scene):
    """N1 on the GPU against the reference's own compiled octreeRaySkip (tests/golden/ref_ray_skip.npz, see
    test_oracle_golden.py): probe grid, random, inside, axis-parallel / clamp cases and a narrowed interval, both kernels."""
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    upload(ctx, s)
    for tag in ("probe", "random", "inside", "axis_inside", "axis_outside0", "axis_outside1", "axis_outside2", "narrow"):
        key = f"{scene}_{tag}"
        ro, rd, (tmin, tmax), want = z[key + "_ro"], z[key + "_rd"], z[key + "_t"], z[key + "_out"]
        for kname, kernel in (("descriptors", rto.KERNEL_AUTO), ("nodes", rto.KERNEL_GENERIC)):
            ctx.set_kernel(kernel)
            got = ctx.octree_ray_skip(ro, rd, float(tmin), float(tmax))
            assert got.tobytes() == want.tobytes(), f"{key} {kname}: {int((got.view(np.uint32) != want.view(np.uint32)).sum())} of {len(want)} differ"
    ctx.set_kernel(rto.KERNEL_AUTO)


def test_octree_ray_skip_matches_oracle(ctx, orc, scenes, camera):
    """N1: the GPU form of octreeRaySkip, bit-exact distances vs the oracle's restatement of the recursion,
    on random rays, axis-aligned rays (the 1e-10 clamp, S/VR:83-87) and the reference's 7x7 probe pattern."""
    rng = np.random.default_rng(11)
    for name, ro in (("sphere32", (0.9, 0.7, 1.3)), ("calgary", (300.0, 900.0, 2500.0)), ("odd", (9.0, 6.0, 7.0))):
        s = scenes(name)
        upload(ctx, s)
        ro = np.array(ro, np.float32)
        centre = s.min + np.array(s.grid.dims, np.float32) * np.float32(0.5) * s.voxel
        tg = centre[None, :] + (rng.random((400, 3)).astype(np.float32) - 0.5) * (np.array(s.grid.dims, np.float32) * s.voxel)[None, :]
        rd = tg - ro[None, :]
        rd = (rd / np.sqrt((rd * rd).sum(axis=1, keepdims=True))).astype(np.float32)
        axis = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, -0.6, -0.8], [1e-12, -1, 0]], np.float32)
        rd = np.concatenate([rd, axis])
        want = np.array([orc.octree_ray_skip(s.nodes, s.min, s.voxel, ro, d) for d in rd], np.float32)
        assert (want < 1e30).sum() > 20
        # narrower [tMin, tMax] window
        want2 = np.array([orc.octree_ray_skip(s.nodes, s.min, s.voxel, ro, d, 0.5, float(np.median(want[want < 1e30]))) for d in rd], np.float32)
        # AUTO = the descriptor form (canonical trees), GENERIC = the 60-byte node form
        for kname, kernel in (("descriptors", rto.KERNEL_AUTO), ("nodes", rto.KERNEL_GENERIC)):
            ctx.set_kernel(kernel)
            got = ctx.octree_ray_skip(ro, rd)
            assert got.tobytes() == want.tobytes(), (name, kname)
            got2 = ctx.octree_ray_skip(ro, rd, 0.5, float(np.median(want[want < 1e30])))
            assert got2.tobytes() == want2.tobytes(), (name, kname)
        ctx.set_kernel(rto.KERNEL_AUTO)
    # the probe pattern of drawRaycast (S/VR:1602-1647): 7x7 NDC samples within +-0.2, unprojected with inverse(P), inverse(V)
    s = scenes("calgary")
    upload(ctx, s)
    view, pos = camera("calgary_oblique")
    P = orc.perspective(orc.radians(45.0), 16 / 9, 0.1, 5000.0)
    invV, invP = orc.mat4_inverse(view).reshape(4, 4).T, orc.mat4_inverse(P).reshape(4, 4).T   # row-major views
    dirs = []
    for y in range(7):
        for x in range(7):
            ndc = np.array([(x / 6 - 0.5) * 2 * 0.2, (y / 6 - 0.5) * 2 * 0.2, 1.0, 1.0], np.float32)
            v = invP @ ndc
            v = v / v[3]
            w = invV @ v
            d = w[:3] - pos
            dirs.append(d / np.sqrt((d * d).sum()))
    dirs = np.array(dirs, np.float32)
    want = np.array([orc.octree_ray_skip(s.nodes, s.min, s.voxel, pos, d) for d in dirs], np.float32)
    assert ctx.octree_ray_skip(pos, dirs).tobytes() == want.tobytes()
    assert (want < 1e30).any()
    # visibility map: nodes culled by the last frustum update return 1e30
    ctx.update_frustum(view, 45.0, 16 / 9, enable=True)
    _, vis = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, 16 / 9)
    assert ctx.octree_ray_skip(pos, dirs, use_visibility=False).tobytes() == want.tobytes()
    got_v = ctx.octree_ray_skip(pos, dirs, use_visibility=True)
    assert (got_v >= want).all()
    ctx.set_kernel(rto.KERNEL_GENERIC)                      # both forms apply the same visibility map
    assert ctx.octree_ray_skip(pos, dirs, use_visibility=True).tobytes() == got_v.tobytes()
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.update_frustum(view, 45.0, 16 / 9, enable=False)


@pytest.mark.parametrize("scene", ["sphere32", "odd", "calgary"])
def test_nearest_hit_render_mode_equals_the_reference_vectors(ctx, orc, scenes, golden, scene):
    """N1 as a render mode (rto_render_skip_*): every pixel's distance is the one the reference's COMPILED octreeRaySkip returned
    for that pixel's ray (tests/golden/ref_ray_skip.npz "pixels": 96 x 64, from outside and -- sphere32 -- from inside the shell's
    hollow); colours and the visibility-map variant (flags of a real frustum update) equal the oracle's, which the same vectors pin
    with random flags; a partition renders its rows; the outputs may stay on the device."""
    torch = pytest.importorskip("torch")
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    upload(ctx, s)
    W, H = 96, 64
    ncam = 2 if scene == "sphere32" else 1
    for ci in range(ncam):
        t, p, r = [float(x) for x in z[f"{scene}_pixels{ci}_cam"]]
        cam = orc.Camera(t, p, r)
        view, pos = cam.get_view(), cam.get_pos()
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        rgba, dist = ctx.render_skip_host(f)
        want = z[f"{scene}_pixels{ci}_out"].reshape(H, W)
        assert dist.tobytes() == want.tobytes(), f"{scene} camera {ci}: {int((dist.view(np.uint32) != want.view(np.uint32)).sum())} distances differ from the reference's"
        orgba, odist = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H)
        assert_bit_exact(rgba, orgba, f"{scene} camera {ci}: nearest-hit colours")
        # device-resident outputs, one part of three
        part = hip.Partition(3, 2, 8)
        rows = partition_row_map(H, 3, 2, 8)
        d_rgba = torch.full((len(rows), W, 4), 7.0, dtype=torch.float32, device="cuda")
        d_dist = torch.full((len(rows), W), 7.0, dtype=torch.float32, device="cuda")
        ctx.render_skip_device(f, d_rgba.data_ptr(), d_dist.data_ptr(), False, part)
        ctx.synchronize()
        assert d_dist.cpu().numpy().tobytes() == want[rows].tobytes() and d_rgba.cpu().numpy().tobytes() == orgba[rows].tobytes()
        # a frustum update's flags as the visibility map (S/VR:64-67)
        aspect = float(np.float32(16 / 9))
        ctx.update_frustum(view, 45.0, aspect, enable=True)
        _, vis = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, aspect)
        vrgba, vdist = ctx.render_skip_host(f, use_visibility=True)
        worgba, wodist = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, visible=vis)
        assert vdist.tobytes() == wodist.tobytes() and vrgba.tobytes() == worgba.tobytes(), f"{scene} camera {ci}: with the visibility map"
        ctx.update_frustum(view, 45.0, aspect, enable=False)


@pytest.mark.parametrize("scene", ["sphere32", "sphere64", "odd", "calgary"])
def test_closest_hit_mode_equals_the_earlier_shader(ctx, orc, scenes, golden, scene):
    """rto_render_closest_*: the reference's closest-hit traversal (the shader it keeps block-commented, RayTracerBVH.cpp:46-166).
    Expected pixels: that shader's TEXT compiled as C++ under the reference's glm (tests/golden/glsl_images_small.npz, made by
    tests/golden/make_golden_glsl.py) -- and the oracle's restatement, which the CPU suite pins to the same frames.  Outside,
    inside-the-shell and axis-aligned cameras; counters too."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_glsl", os.path.join(ROOT, "tests", "golden", "make_golden_glsl.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    z = golden("glsl_images_small.npz")
    s = scenes(scene)
    upload(ctx, s)
    W, H, cams = m.CASES[scene]
    for i, cspec in enumerate(cams):
        view, pos, fov = m.camera(cspec)
        f = rto.make_frame(view, pos, W / H, fov, W, H)
        got, st = ctx.render_closest_host(f, stats=True)
        assert got.tobytes() == z[f"{scene}_cam{i}_closest"].tobytes(), f"{scene} camera {i}: closest-hit frame vs the shader text under glm"
        want, wst = orc.render_closest(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H)
        assert_bit_exact(got, want, f"{scene} camera {i}: closest-hit frame vs the oracle")
        assert (st["pops"], st["hits"]) == (wst["pops"], wst["hits"])
        assert_bit_exact(ctx.render_closest_host(f), want, f"{scene} camera {i}: without counters")
        # the live shader's frame through the same context, against ITS text
        assert ctx.render_host(f).tobytes() == z[f"{scene}_cam{i}_first"].tobytes(), f"{scene} camera {i}: first-hit frame vs the live shader's text"


def test_closest_hit_mode_at_config2_size_and_with_culling(ctx, orc, scenes, camera):
    """The closest-hit mode at BASELINE config 2's size (256^3, 1920x1080) against the oracle, a 3-way partition on a caller's
    stream, and through a frustum update (the reference's culled render walks the compacted array)."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere256")
    upload(ctx, s)
    W, H = 1920, 1080
    nthr = min(16, orc.max_threads())
    for cam in (orc.Camera(0.5, 0.7, 1.8), orc.Camera(1.0, 0.9, 0.25)):
        view, pos = cam.get_view(), cam.get_pos()
        f = rto.make_frame(view, pos, W / H, 45.0, W, H)
        want, wst = orc.render_closest(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=nthr)
        got, st = ctx.render_closest_host(f, stats=True)
        assert_bit_exact(got, want, "closest-hit mode, 256^3 at 1080p")
        assert (st["pops"], st["hits"]) == (wst["pops"], wst["hits"])
    part = hip.Partition(3, 1, 16)
    rows = partition_row_map(H, 3, 1, 16)
    pb = torch.full((len(rows), W, 4), 7.0, dtype=torch.float32, device="cuda")
    other = torch.cuda.Stream()
    ctx.render_closest_device(f, pb.data_ptr(), part, other.cuda_stream)
    torch.cuda.synchronize()
    assert_bit_exact(pb.cpu().numpy(), want[rows], "closest-hit mode, part 1 of 3 on a caller's stream")
    cal = scenes("calgary")
    upload(ctx, cal)
    view, pos = camera("calgary_oblique")
    Wc, Hc = 480, 270
    nodes_c, _ = orc.cull_compact(cal.nodes, cal.min, cal.voxel, view, 45.0, Wc / Hc)
    fc = rto.make_frame(view, pos, Wc / Hc, 45.0, Wc, Hc)
    try:
        ctx.update_frustum(view, 45.0, Wc / Hc, enable=True)
        wantc, _ = orc.render_closest(nodes_c, cal.min, cal.voxel, view, pos, Wc / Hc, 45.0, Wc, Hc)
        assert_bit_exact(ctx.render_closest_host(fc), wantc, "closest-hit mode through a frustum update")
    finally:
        ctx.update_frustum(view, 45.0, Wc / Hc, enable=False)


def test_closest_hit_ties_go_to_the_leaf_popped_first(ctx, orc):
    """Equal tHit at two leaves: the reference's walk (children pop 7 .. 0, strictly nearer replaces) keeps the one popped first.
    The near-first kernel (default) resolves the tie by that order without walking it; the pop-order kernel and the node-by-node
    kernel walk it.  Rays made to tie: unit voxels on integer planes, cameras on integer points looking along axes, face
    diagonals and body diagonals (an odd frame size puts the centre ray exactly on an edge shared by four leaves, a diagonal
    ray through corner after corner), a random half-full grid with one merged block so the tied leaves differ in size and state."""
    rng = np.random.default_rng(2024)
    n = 16
    data = (rng.random((n, n, n)) < 0.5).astype(np.uint8)
    data[4:8, 4:8, 8:12] = 1                                            # one 4^3 leaf
    data[8:10, 8:10, 8:10] = 0                                          # and a hole at the centre
    g = orc.Grid((n, n, n), np.zeros(3, np.float32), np.float32(1.0), data)
    nodes = orc.build_flat_octree(g)
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.upload_octree(nodes, g.min, g.voxel_size)
    W = H = 33
    centre = np.array([8.0, 8.0, 8.0], np.float32)
    dirs = [d for d in np.ndindex(3, 3, 3) if d != (1, 1, 1)]           # 26 directions: axes, face diagonals, body diagonals
    differ = 0
    for d in dirs:
        off = (np.array(d, np.float32) - 1.0) * np.float32(24.0)
        pos = centre + off
        fwd = -off / np.linalg.norm(off)
        up = np.array([0.0, 1.0, 0.0], np.float32) if abs(fwd[1]) < 0.9 else np.array([0.0, 0.0, 1.0], np.float32)
        sx = np.cross(fwd, up); sx = sx / np.linalg.norm(sx)
        uy = np.cross(sx, fwd)
        view = np.eye(4, dtype=np.float32)                              # column-major, as glm::lookAt lays it out
        view[0, :3], view[1, :3], view[2, :3] = sx, uy, -fwd
        view[0, 3], view[1, 3], view[2, 3] = -np.dot(sx, pos), -np.dot(uy, pos), np.dot(fwd, pos)
        view = np.ascontiguousarray(view.T.astype(np.float32)).reshape(16)
        for fov in (90.0, 53.13010235415598):                           # tan(fov/2) = 1 and 1/2: pixel directions are small rationals
            f = rto.make_frame(view, pos, 1.0, fov, W, H)
            want, wst = orc.render_closest(nodes, g.min, g.voxel_size, view, pos, 1.0, fov, W, H)
            first, _ = orc.render(nodes, g.min, g.voxel_size, view, pos, 1.0, fov, W, H)
            differ += int((first.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
            for kname, kernel in (("near first", rto.KERNEL_AUTO), ("pop order", rto.KERNEL_PACKED_V1), ("node by node", rto.KERNEL_GENERIC)):
                ctx.set_kernel(kernel)
                assert_bit_exact(ctx.render_closest_host(f), want, f"closest hit from {tuple(int(x) for x in off)}, fov {fov:.0f}, {kname}")
            ctx.set_kernel(rto.KERNEL_AUTO)
            got, st = ctx.render_closest_host(f, stats=True)
            assert (st["pops"], st["hits"]) == (wst["pops"], wst["hits"])
    assert differ > 0                                                   # the two shaders do disagree on this grid: the test can see a wrong winner


def test_nearest_hit_render_mode_at_config2_size(ctx, orc, scenes, camera):
    """The same at BASELINE config 2's size (256^3 sphere, 1920x1080) against the oracle; where both modes hit, the nearest-hit
    distance never exceeds the distance of the reference's first-in-DFS-order hit (the reason SURVEY.md section 8f wants the mode)."""
    s = scenes("sphere256")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 1920, 1080
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    rgba, dist = ctx.render_skip_host(f)
    orgba, odist = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=min(16, orc.max_threads()))
    assert dist.tobytes() == odist.tobytes()
    assert_bit_exact(rgba, orgba, "nearest-hit frame, config 2 size")
    dfs, _ = oracle_frame(orc, s, view, pos, W, H)
    hit_dfs, hit_near = dfs[..., 0] != 0, dist < 1e30
    assert hit_near.sum() >= hit_dfs.sum() > 100000                # the DFS mode loses rays to its 512-pop cap; this mode has none
    # both kinds of frame in turn on one stream: each keeps its own launch-order history (stream, kind), neither disturbs the other
    for k in range(3):
        assert_bit_exact(ctx.render_host(f), dfs, f"first-hit frame between nearest-hit frames, round {k}")
        r2, d2 = ctx.render_skip_host(f)
        assert d2.tobytes() == odist.tobytes() and r2.tobytes() == orgba.tobytes(), f"nearest-hit frame between first-hit frames, round {k}"
    assert ctx.debug_sort_violations() == 0


@pytest.mark.parametrize("scene,cam", [("sphere32", (0.5, 0.7, 1.8)), ("odd", (0.4, 0.9, 9.0)), ("calgary", (0.6, 0.5, 3500.0))])
def test_probe_consumer_in_one_launch_equals_the_reference(ctx, orc, scenes, golden, scene, cam):
    """octreeRaySkip's consumer (S/VolumeRaycastRenderer.cpp:1602-1663) in ONE launch: the 49 probe distances are the compiled
    reference's, and three consecutive updates give the sequence computed from them (15th percentile x 0.75, blend 0.4 / 0.6);
    with a frustum update's flags as the visibility map the value follows the oracle; the device form keeps the float on the GPU."""
    torch = pytest.importorskip("torch")
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    upload(ctx, s)
    c = orc.Camera(*cam)
    view, eye = c.get_view(), c.get_pos()
    aspect = float(np.float32(1920 / 1080))
    want = z[f"{scene}_probe_skip_seq"][:3]
    last, got = np.float32(0.0), []
    for k in range(3):
        last, probes = ctx.probe_skip_host(view, eye, aspect, last, with_probes=True)
        got.append(last)
        assert probes.tobytes() == z[f"{scene}_probe_out"].tobytes(), "the 49 probe distances"
    assert np.array(got, np.float32).tobytes() == want.tobytes(), (got, want)
    d_skip = torch.zeros(1, dtype=torch.float32, device="cuda")     # the caller's lastSkipDistance lives on the device
    for k in range(3):
        ctx.probe_skip_device(view, eye, aspect, d_skip.data_ptr())
    ctx.synchronize()
    assert d_skip.cpu().numpy().tobytes() == want[2:3].tobytes()
    ctx.update_frustum(view, 45.0, aspect, enable=True)
    _, vis = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, aspect)
    a = ctx.probe_skip_host(view, eye, aspect, float(want[2]), use_visibility=True)
    b = orc.probe_skip_distance(s.nodes, s.min, s.voxel, view, eye, aspect, float(want[2]), visible=vis)
    assert np.float32(a).tobytes() == np.float32(b).tobytes()
    ctx.update_frustum(view, 45.0, aspect, enable=False)
    import time
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.probe_skip_device(view, eye, aspect, d_skip.data_ptr())
    ctx.synchronize()
    print(f"probe update, {scene}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (200 back-to-back device calls, no copies)")


@pytest.fixture(params=["morton", "level_by_level"])
def build_path(request, ctx):
    """rto_build_octree's two forms (seven launches with Morton-ordered pyramid levels / level by level): same arrays."""
    ctx.debug_set_build_path(request.param == "level_by_level")
    yield request.param
    ctx.debug_set_build_path(False)


@pytest.mark.parametrize("name", ["sphere16", "sphere64", "sphere256", "calgary", "odd"])
def test_gpu_octree_build_equals_reference_arrays(ctx, orc, scenes, camera, golden_meta, name, build_path):
    """N4: the flat array built on the GPU from the voxel grid is byte-identical to createOctreeFromVoxelGrid +
    setOctree (oracle == reference goldens), and renders identically through the packed kernel."""
    import hashlib

    s = scenes(name)
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.build_octree(s.grid.data, s.min, s.voxel)
    got = ctx.download_nodes()
    assert len(got) == len(s.nodes)
    assert got.tobytes() == s.nodes.tobytes()
    if name in golden_meta["octrees"]:
        assert hashlib.sha256(got.tobytes()).hexdigest() == golden_meta["octrees"][name]["sha256"]
    info = ctx.info()
    assert info.canonical == 1 and info.num_internal == int((s.nodes["isLeaf"] == 0).sum())
    if name == "odd":
        cam = orc.Camera(0.4, 0.9, 9.0)
        view, pos = cam.get_view(), cam.get_pos()
    else:
        view, pos = camera("calgary_oblique" if name == "calgary" else "sphere")
    W, H = 320, 200
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    for kname, kernel in KERNELS:
        ctx.set_kernel(kernel)
        assert_bit_exact(ctx.render_host(rto.make_frame(view, pos, W / H, 45.0, W, H)), want, f"gpu-built {name} {kname}")
    # culling works on the GPU-built tree too
    if name == "calgary":
        ctx.set_kernel(rto.KERNEL_AUTO)
        ctx.update_frustum(view, 45.0, W / H, enable=True)
        cn, _ = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, W / H)
        assert ctx.download_visible_nodes().tobytes() == cn.tobytes()
        wantc, _ = orc.render(cn, s.min, s.voxel, view, pos, W / H, 45.0, W, H)
        assert_bit_exact(ctx.render_host(rto.make_frame(view, pos, W / H, 45.0, W, H)), wantc, "gpu-built culled")
        ctx.update_frustum(view, 45.0, W / H, enable=False)


def test_gpu_octree_build_degenerate_and_random_grids(ctx, orc, build_path):
    rng = np.random.default_rng(3)
    cases = [((1, 1, 1), 1.0), ((1, 1, 1), 0.0), ((4, 4, 4), 1.0), ((4, 4, 4), 0.0), ((3, 3, 3), 1.0), ((7, 5, 3), 0.5),
             ((33, 9, 20), 0.9), ((64, 1, 1), 0.5), ((2, 3, 1), 0.0), ((17, 17, 17), 0.02),
             # dimX % 16 == 0 takes the 16-byte level-1 pyramid kernel: odd / single rows and slices, all states
             ((16, 5, 7), 0.5), ((32, 33, 2), 0.9), ((48, 1, 1), 0.5), ((16, 16, 16), 1.0), ((16, 2, 3), 0.0), ((80, 37, 11), 0.03),
             # several 32^3 bricks, some of them outside the grid (their cells must read EMPTY), sparse and dense
             ((65, 40, 33), 0.002), ((96, 64, 32), 0.4), ((33, 33, 33), 0.0), ((64, 64, 64), 1.0), ((100, 3, 70), 0.1), ((2, 2, 2), 0.5)]
    for dims, p in cases:
        data = (rng.random((dims[2], dims[1], dims[0])) < p).astype(np.uint8)
        mn = np.array([0.5, -2.0, 3.0], np.float32)
        want = orc.build_flat_octree(orc.Grid(dims, mn, np.float32(0.3), data))
        ctx.build_octree(data, mn, 0.3)
        assert ctx.download_nodes().tobytes() == want.tobytes(), (dims, p)
        assert ctx.info().canonical == (1 if len(want) > 1 else 0)
    with pytest.raises(rto.RtoError) as e:
        ctx.build_octree(np.zeros((0, 4, 4), np.uint8), [0, 0, 0], 1.0)
    assert e.value.code == hip.RTO_E_INVALID


def test_gpu_octree_build_512(ctx, orc, scenes, build_path):
    s = scenes("sphere512")
    ctx.build_octree(s.grid.data, s.min, s.voxel)
    assert ctx.download_nodes().tobytes() == s.nodes.tobytes()
    ctx.build_octree(s.grid.data, s.min, s.voxel)                # second build: scratch comes from the pool
    k_ms, up_ms = ctx.last_build_ms()
    print(f"rto_build_octree 512^3 [{build_path}]: kernels {k_ms:.3f} ms, upload {up_ms:.3f} ms")
    assert 0 < k_ms < 1000


@pytest.mark.parametrize("scene,W,H,cam", [("sphere32", 160, 120, (0.5, 0.7, 1.8)), ("sphere64", 320, 200, (0.5, 0.7, 1.8)),
                                           ("sphere64", 200, 200, (-0.6, 3.9, 1.4)), ("odd", 200, 120, (0.4, 0.9, 9.0)),
                                           ("calgary", 320, 180, (0.6, 0.5, 3500.0))])
def test_config5_leaf_triangles_and_shadow_ray(ctx, orc, scenes, scene, W, H, cam):
    """N2 / BASELINE config 5 (extension, own oracle -- the reference has no ray/triangle code): Marching-Cubes
    leaf triangles (pinned by the reference's localMC) + Moeller-Trumbore + one shadow ray, bit-exact vs the oracle."""
    s = scenes(scene)
    upload(ctx, s)
    g = rto.VoxelGrid.from_array(s.grid.data, s.min, s.voxel)
    tris, off = rto.buildLeafTriangles(g, s.nodes)          # product builder (C++), equal to the oracle's (CPU test)
    ctx.upload_leaf_triangles(tris, off)
    c = orc.Camera(*cam)
    view, pos = c.get_view(), c.get_pos()
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    for shadow in (False, True):
        want, st = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=shadow,
                                        nthreads=min(16, orc.max_threads()))
        assert st["hits"] > 0
        # AUTO = packed descriptors + "leaf owns triangles" mask; GENERIC = the 60-byte node skeleton
        for kname, kernel in (("packed", rto.KERNEL_AUTO), ("generic", rto.KERNEL_GENERIC)):
            ctx.set_kernel(kernel)
            got, gs = ctx.render_triangles_host(f, shadow=shadow, stats=True)
            assert_bit_exact(got, want, f"{scene} triangles shadow={shadow} {kname}")
            assert (gs["pops"], gs["hits"]) == (st["pops"], st["hits"]), kname
            got2 = ctx.render_triangles_host(f, shadow=shadow)              # colour-only instantiation
            assert_bit_exact(got2, want, f"{scene} triangles shadow={shadow} {kname} (no counters)")


@pytest.mark.parametrize("scene", ["sphere16", "sphere32", "sphere64", "odd", "calgary"])
def test_gpu_leaf_triangle_build_equals_the_host_builders(ctx, orc, scenes, scene):
    """rto_build_leaf_triangles: the buffer MarchingCubesRenderer/localMC would emit per leaf, built in HBM -- the same
    bytes as the oracle's builder (itself pinned by the reference's localMC triangles) and as the product's C++ one;
    for an uploaded octree + voxels and for rto_build_octree + the voxels it kept."""
    s = scenes(scene)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    upload(ctx, s)
    ctx.build_leaf_triangles(s.grid.data)
    gt, go = ctx.download_leaf_triangles()
    assert go.tobytes() == np.asarray(wo, np.int32).tobytes(), f"{scene}: triOffset"
    assert gt.shape == np.asarray(wt).reshape(-1, 12).shape
    assert gt.tobytes() == np.ascontiguousarray(wt, np.float32).tobytes(), f"{scene}: triangles"
    k_ms, _ = ctx.last_build_ms()
    assert 0 < k_ms < 1000
    # GPU-built octree, voxels already resident
    ctx.build_octree(s.grid.data, s.min, s.voxel)
    ctx.build_leaf_triangles(None)
    gt2, go2 = ctx.download_leaf_triangles()
    assert go2.tobytes() == go.tobytes() and gt2.tobytes() == gt.tobytes()
    # and it renders: same frame as with the uploaded buffer
    c = orc.Camera(0.5, 0.7, 1.8) if scene != "calgary" else orc.Camera(0.6, 0.5, 3500.0)
    if scene == "odd":
        c = orc.Camera(0.4, 0.9, 9.0)
    W, H = 200, 120
    f = rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H)
    a = ctx.render_triangles_host(f, shadow=True)
    ctx.upload_leaf_triangles(wt, wo)
    assert_bit_exact(a, ctx.render_triangles_host(f, shadow=True), f"{scene}: GPU-built vs uploaded triangles")
    # error paths
    with pytest.raises(rto.RtoError):
        rto.Context(0).build_leaf_triangles(s.grid.data)            # no octree
    upload(ctx, s)
    with pytest.raises(rto.RtoError):
        ctx.build_leaf_triangles(None)                              # no voxels kept by an upload


def test_gpu_leaf_triangle_build_512(ctx, orc, scenes):
    s = scenes("sphere512")
    ctx.build_octree(s.grid.data, s.min, s.voxel)
    ctx.build_leaf_triangles(None)
    k_ms, _ = ctx.last_build_ms()
    gt, go = ctx.download_leaf_triangles()
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    assert go.tobytes() == np.asarray(wo, np.int32).tobytes()
    assert gt.tobytes() == np.ascontiguousarray(wt, np.float32).tobytes()
    print(f"GPU leaf-triangle build 512^3: {len(gt)} triangles in {k_ms:.3f} ms")


def test_config5_shadows_exist_and_partitions_agree(ctx, orc, scenes):
    torch = pytest.importorskip("torch")
    s = scenes("calgary")
    upload(ctx, s)
    g = rto.VoxelGrid.from_array(s.grid.data, s.min, s.voxel)
    ctx.upload_leaf_triangles(*rto.buildLeafTriangles(g, s.nodes))
    c = orc.Camera(0.9, 2.4, 2600.0)
    W, H = 384, 216
    f = rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H)
    lit = ctx.render_triangles_host(f, shadow=False)
    shd = ctx.render_triangles_host(f, shadow=True)
    changed = (lit != shd).any(axis=-1)
    assert changed.sum() > 50                                   # buildings do cast shadows
    assert (shd[changed][:, 0] == np.float32(0.1)).all()        # shadowed pixels keep the ambient term only
    # band partition of the triangle path == whole frame
    nparts, band = 3, 16
    rows0 = ctx.partition_rows(f, hip.Partition(nparts, 0, band))
    gathered = torch.zeros((nparts, rows0, W, 4), dtype=torch.float32, device="cuda")
    for p in range(nparts):
        ctx.render_triangles_device(f, gathered[p].data_ptr(), True, hip.Partition(nparts, p, band))
    frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    ctx.assemble_device(f, hip.Partition(nparts, 0, band), gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert_bit_exact(frame.cpu().numpy(), shd, "triangle path, 3 parts")
    # ... and with the 4-byte shade payload, on both triangle kernels
    for kname, kernel in (("packed", rto.KERNEL_AUTO), ("generic", rto.KERNEL_GENERIC)):
        ctx.set_kernel(kernel)
        gs = torch.zeros((nparts, rows0, W), dtype=torch.float32, device="cuda")
        for p in range(nparts):
            ctx.render_triangles_shade_device(f, gs[p].data_ptr(), True, hip.Partition(nparts, p, band))
        ctx.assemble_shade_device(f, hip.Partition(nparts, 0, band), gs.data_ptr(), frame.data_ptr())
        ctx.synchronize()
        assert_bit_exact(frame.cpu().numpy(), shd, f"triangle path, 3 parts, shade payload, {kname}")
    ctx.set_kernel(rto.KERNEL_AUTO)
    # error paths
    fresh = rto.Context(0)
    fresh.upload_octree(scenes("sphere16").nodes, scenes("sphere16").min, scenes("sphere16").voxel)
    with pytest.raises(rto.RtoError) as e:
        fresh.render_triangles_host(rto.make_frame(c.get_view(), c.get_pos(), 1.0, 45.0, 16, 16))
    assert e.value.code == hip.RTO_E_NO_OCTREE
    with pytest.raises(rto.RtoError) as e:
        fresh.upload_leaf_triangles(np.zeros((3, 12), np.float32), np.zeros(5, np.int32))
    assert e.value.code == hip.RTO_E_INVALID
    fresh.close()


def test_config5_full_size_512_4k_triangles_shadow(ctx, orc, scenes, camera):
    """BASELINE config 5 at full size on one GPU: 512^3 sphere, 3840x2160, MC leaf triangles + 1 shadow ray per hit."""
    s = scenes("sphere512")
    view, pos = camera("sphere")
    W, H = 3840, 2160
    upload(ctx, s)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    assert len(wt) > 1_000_000
    ctx.upload_leaf_triangles(wt, wo)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=True,
                                    nthreads=min(16, orc.max_threads()))
    for kname, kernel in (("packed", rto.KERNEL_AUTO), ("generic", rto.KERNEL_GENERIC)):
        ctx.set_kernel(kernel)
        got, gs = ctx.render_triangles_host(f, shadow=True, stats=True)
        assert_bit_exact(got, want, f"config 5 full size {kname}")
        assert (gs["pops"], gs["hits"]) == (st["pops"], st["hits"]), kname
        ctx.render_triangles_host(f, shadow=True)
        print(f"config5 4K triangles+shadow [{kname}]: {st['hits']} hits, {len(wt)} triangles, kernel {ctx.last_kernel_ms():.3f} ms")


def test_cpp_dropin_class_triangle_path(orc, scenes):
    W, H = 256, 160
    g = rto.VoxelGrid.test_sphere(32)
    root = rto.createOctreeFromVoxelGrid(g)
    rt = rto.RayTracerBVH()
    rt.ensureComputeInitialized()
    rt.setOctree(root, g)
    rt.buildLeafTriangles()
    cam = rto.Camera(0.5, 0.7, 1.8)
    rt.renderSceneTriangles(cam, W, H, W / H, 45.0, True)
    s = scenes("sphere32")
    oc = orc.Camera(0.5, 0.7, 1.8)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    want, _ = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, oc.get_view(), oc.get_pos(), W / H, 45.0, W, H, shadow=True)
    assert_bit_exact(rt.framebuffer(), want, "C++ class, triangle path (buffer built on the GPU)")
    rt.buildLeafTrianglesOnHost()
    rt.renderSceneTriangles(cam, W, H, W / H, 45.0, True)
    assert_bit_exact(rt.framebuffer(), want, "C++ class, triangle path (buffer built on the host)")
    rto.freeOctree(root)
    rt2 = rto.RayTracerBVH()                     # everything on the GPU: octree and triangles from the voxels
    rt2.ensureComputeInitialized()
    rt2.setOctreeFromGrid(g)
    rt2.buildLeafTriangles()
    rt2.renderSceneTriangles(cam, W, H, W / H, 45.0, True)
    assert_bit_exact(rt2.framebuffer(), want, "C++ class, GPU-built octree + GPU-built triangles")


def test_root_screen_rectangle_never_changes_pixels(ctx, orc, scenes):
    """The packed kernel skips ray setup for pixels outside a conservative screen rectangle of the root box.
    Cameras that put the box partly off-screen, behind the eye, around the eye and far away must all stay exact."""
    s = scenes("sphere32")
    upload(ctx, s)
    W, H = 200, 120
    for (t, p, r, tgt, fov) in ((0.5, 0.7, 1.8, (0, 0, 0), 45.0), (0.5, 0.7, 1.8, (0.8, 0.3, 0.0), 45.0), (0.1, 0.2, 0.9, (0.0, 0.9, 0.0), 70.0),
                                (0.5, 0.7, 0.3, (0, 0, 0), 45.0), (0.5, 0.7, 6.0, (0, 0, 0), 10.0), (0.5, 0.7, 1.8, (5.0, 5.0, 9.0), 45.0),
                                (-1.2, 4.0, 0.75, (0.2, -0.1, 0.1), 120.0), (0.5, 0.7, 1.2, (0, 0, 2.0), 45.0)):
        cam = orc.Camera(t, p, r)
        cam._c.target[0], cam._c.target[1], cam._c.target[2] = tgt
        view, pos = cam.get_view(), cam.get_pos()
        want, st = oracle_frame(orc, s, view, pos, W, H, fov=fov)
        f = rto.make_frame(view, pos, W / H, fov, W, H)
        ctx.set_kernel(rto.KERNEL_PACKED)
        assert_bit_exact(ctx.render_host(f), want, f"root rectangle cam {(t, p, r, tgt, fov)}")
        np.testing.assert_array_equal(ctx.render_steps(f), orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H))


def test_comm_one_rank_through_the_c_abi(ctx, orc, scenes):
    """The multi-GPU path below the C boundary (rto_comm_*: render part -> ONE grouped ncclSend/ncclRecv -> assemble on rank 0),
    driven with a one-rank RCCL communicator -- everything except traffic between GPUs: batches of 1 and 4 frames, six
    submits in flight before the flush (the two buffer sets alternate), octree and triangle modes, bit-exact frames."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere64")
    upload(ctx, s)
    W, H = 417, 250
    cams = [orc.Camera(0.5 + 0.3 * i, 0.7 + 0.05 * i, 1.8) for i in range(4)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants = [oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H)[0] for c in cams]
    comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        out = torch.full((6, 4, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
        arr4 = hip.Context.frame_array(frames)
        for k in range(6):                                              # six batches of four frames, nothing waited for in between
            comm.submit(arr4, out[k].data_ptr(), out.stride(1) * 4)
        comm.flush()
        for k in range(6):
            for i in range(4):
                assert_bit_exact(out[k, i].cpu().numpy(), wants[i], f"comm batch {k} frame {i}")
        sent, whole = comm.debug_last_payload()
        assert 0 < sent < whole, "only the columns of the geometry's rectangle travel"
        one = torch.full((H, W, 4), 7.0, dtype=torch.float32, device="cuda")
        # a camera that looks away from the scene: nothing but background, a token window travels
        away = orc.Camera(0.5, 0.7, 1.8)
        away.set_target(40.0, 0.0, 40.0)
        fa = rto.make_frame(away.get_view(), away.get_pos(), W / H, 45.0, W, H)
        wa = oracle_frame(orc, s, away.get_view(), away.get_pos(), W, H)[0]
        comm.submit(hip.Context.frame_array([fa, frames[1]]), out[0].data_ptr(), out.stride(1) * 4)
        comm.flush()
        assert_bit_exact(out[0, 0].cpu().numpy(), wa, "comm: frame without geometry")
        assert_bit_exact(out[0, 1].cpu().numpy(), wants[1], "comm: frame after the empty one")
        for i in (2, 0, 3):                                             # batch size changes: buffers are re-made after a drain
            comm.submit(hip.Context.frame_array([frames[i]]), one.data_ptr(), one.stride(0) * 4 * H)
            comm.flush()
            assert_bit_exact(one.cpu().numpy(), wants[i], f"comm single frame {i}")
        # config 5's path through the same pipe
        wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
        ctx.build_leaf_triangles(s.grid.data)
        for mode, shadow in ((hip.RESIDENT_TRIANGLES, False), (hip.RESIDENT_TRIANGLES_SHADOW, True)):
            wtri, _ = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, cams[1].get_view(), cams[1].get_pos(), W / H, 45.0, W, H, shadow=shadow)
            one.fill_(7.0)
            comm.submit(hip.Context.frame_array([frames[1]]), one.data_ptr(), 0, mode)
            comm.flush()
            assert_bit_exact(one.cpu().numpy(), wtri, f"comm triangles mode {mode}")
        with pytest.raises(rto.RtoError):
            comm.submit(arr4, 0, 0)                                     # rank 0 without a frame buffer
        # every rank of a 2-, 3-, 4- and 8-GPU split, played in turn by this GPU (rto_comm_debug_rehearse): the bands rank r
        # renders, ships and assembles must be the oracle's rows of the bands it owns -- together the whole frame
        upload(ctx, s)
        for world in (2, 3, 4, 8):
            got = np.zeros((H, W, 4), np.float32)
            band = np.arange(H) // 16
            # from 4 GPUs on rank 0 only gathers and assembles; the frame is split over ranks 1 .. world-1
            owner = band % world if world < 4 else 1 + band % (world - 1)
            for r in range(world):
                comm.debug_rehearse(world, r)
                two = torch.full((2, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
                comm.submit(hip.Context.frame_array([frames[0], frames[3]]), two.data_ptr(), two.stride(0) * 4)
                comm.flush()
                a, b = two[0].cpu().numpy(), two[1].cpu().numpy()
                got[owner == r] = a[owner == r]
                assert_bit_exact(b[owner == r], wants[3][owner == r], f"rehearsal world {world} rank {r}, second frame of the batch")
            assert_bit_exact(got, wants[0], f"rehearsal world {world}: the ranks' bands together")
        # the same for config 5's path: the parts of two frames in one launch of the triangle kernel
        ctx.build_leaf_triangles(s.grid.data)
        wtris = [orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, cams[i].get_view(), cams[i].get_pos(), W / H, 45.0, W, H, shadow=True)[0] for i in (1, 2)]
        for world in (3, 5):
            owner = (np.arange(H) // 16) % 3 if world == 3 else 1 + (np.arange(H) // 16) % 4
            for r in range(world):
                comm.debug_rehearse(world, r)
                two = torch.full((2, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
                comm.submit(hip.Context.frame_array([frames[1], frames[2]]), two.data_ptr(), two.stride(0) * 4, hip.RESIDENT_TRIANGLES_SHADOW)
                comm.flush()
                for q in range(2):
                    assert_bit_exact(two[q].cpu().numpy()[owner == r], wtris[q][owner == r], f"triangle rehearsal rank {r} of {world}, frame {q}")
        comm.debug_rehearse(0)
        with pytest.raises(rto.RtoError):
            comm.debug_rehearse(4, 4)
    finally:
        comm.close()


def test_comm_ranks_exchange_through_a_loopback_transport(tmp_path):
    """Row (e) above one rank: every rank of a 2 / 3 / 4 / 5 / 8-GPU split as its own rto_context + rto_comm (rto_comm_create, the
    multi-process entry point, NOT the rehearsal hook) in one child process on the one GPU, the product's submit / pack / comm_exchange
    / assemble unchanged -- rank 0's receive loop over the peers, the self-send of a rendering rank 0 at world 2 and 3, cropped and
    uncropped plans, octree and triangle + shadow modes, both buffer sets -- with tests/rccl_shim (a device-to-device copy per matched
    ncclSend / ncclRecv pair) in the place of librccl, which refuses two ranks on one device.  What stays unexercised is RCCL's own
    transport; every byte rank 0 assembles here was packed by ANOTHER rank's context.  The child never imports torch (it maps the real
    librccl under the same SONAME)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc to build the loopback transport with")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "librccl.so.1"
    subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-std=c++17", os.path.join(root, "tests", "rccl_shim", "rccl_shim.cpp"), "-o", str(so)],
                   check=True, capture_output=True)
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_comm_loopback_worker.py"), "serial"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "assembled frames equal the oracle's" in p.stdout, p.stdout[-2000:]
    # one host thread per rank, six batches in flight before the flush (the pipelined form bench.py times at N > 1): a send / receive of
    # the stand-in returns when it has met its counterpart, as NCCL's kernels complete
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_comm_loopback_worker.py"), "threads"], env=dict(env, RTO_RCCL_SHIM_RENDEZVOUS="1"),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "one thread per rank, pipelined" in p.stdout, p.stdout[-2000:]


def test_comm_timeout_and_dead_communicator(ctx, orc, scenes):
    """The failure path of the collective (nothing upstream to mirror: the reference is single-GPU).  A flush with a limit returns in
    time on a healthy communicator; an aborted one -- what a flush timeout or an asynchronous RCCL error leaves behind -- is DEAD: it
    refuses submits and flushes with RTO_E_INVALID, reports so, and rto_comm_destroy still returns; the context renders on, and a new
    communicator on the same context works."""
    torch = pytest.importorskip("torch")
    s = scenes("sphere32")
    upload(ctx, s)
    W, H = 160, 96
    cam = orc.Camera(0.5, 0.7, 1.8)
    f = rto.make_frame(cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
    want = oracle_frame(orc, s, cam.get_view(), cam.get_pos(), W, H)[0]
    arr = hip.Context.frame_array([f])
    out = torch.full((H, W, 4), 7.0, dtype=torch.float32, device="cuda")
    comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        comm.submit(arr, out.data_ptr(), 0)
        comm.flush(timeout_ms=60000)
        assert not comm.is_dead()
        assert_bit_exact(out.cpu().numpy(), want, "healthy communicator, flush with a limit")
        comm.submit(arr, out.data_ptr(), 0)
        comm.debug_abort()                                      # what the expiry of a flush timeout does
        assert comm.is_dead()
        for call in (lambda: comm.submit(arr, out.data_ptr(), 0), lambda: comm.flush(), lambda: comm.flush(timeout_ms=10)):
            with pytest.raises(rto.RtoError) as e:
                call()
            assert e.value.code == hip.RTO_E_INVALID and "dead" in str(e.value)
    finally:
        comm.close()                                            # must return
    assert_bit_exact(ctx.render_host(f), want, "the context renders on after its communicator died")
    again = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        out.fill_(7.0)
        again.submit(arr, out.data_ptr(), 0)
        again.flush(timeout_ms=60000)
        assert_bit_exact(out.cpu().numpy(), want, "a new communicator on the same context")
    finally:
        again.close()


def _rehearse_every_rank(ctx, comm, frames, wants, worlds, mode, what):
    """Every rank r of an N-GPU split, played in turn by this GPU through the whole pipeline (render its bands, pack the plan's
    windows, grouped send/recv, assembly): the rows the PLAN gives to rank r must be the oracle's, and together they are the
    whole frame.  The plan comes from the exported planner, i.e. this also checks that rto_comm_* does what rto_split_plan_make says."""
    import torch
    import tilesplit

    H, W = wants[0].shape[:2]
    n = len(frames)
    arr = hip.Context.frame_array(frames)
    for world in worlds:
        plan = tilesplit.make_plan(ctx.scene_bounds(), frames, world, 16)
        src_part, _ = tilesplit.row_sources(plan)
        owner = src_part + plan.first_render_rank
        got = [np.zeros((H, W, 4), np.float32) for _ in range(n)]
        for r in range(world):
            comm.debug_rehearse(world, r)
            out = torch.full((n, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
            comm.submit(arr, out.data_ptr(), out.stride(0) * 4, mode)
            comm.flush()
            sent, whole = comm.debug_last_payload()
            assert (sent, whole) == (plan.pack_floats, plan.full_floats), f"{what} world {world}: the communicator ships what the plan says"
            res = out.cpu().numpy()
            mine = owner == r
            if tilesplit.part_of_rank(plan, r) < 0:
                assert not mine.any(), "the gatherer owns no rows"
            for i in range(n):
                assert_bit_exact(res[i][mine], wants[i][mine], f"{what}: world {world} rank {r} frame {i}: its bands")
                got[i][mine] = res[i][mine]
        for i in range(n):
            assert_bit_exact(got[i], wants[i], f"{what}: world {world} frame {i}: the ranks' bands together")
            assert int((got[i][..., 0] != 0).sum()) == int((wants[i][..., 0] != 0).sum())
    comm.debug_rehearse(0)


def test_config3_every_rank_of_2_4_8_gpus_at_full_size(ctx, orc, scenes, camera):
    """BASELINE config 3 at its own workload: 256^3 sphere, 1920x1080, the screen split over 2, 4 and 8 GPUs -- each rank's
    share rendered, shipped and assembled by this GPU (rto_comm_debug_rehearse), two cameras per batch; the union of the
    ranks' bands is the oracle's frame bit for bit, and the frame's pop / hit counters are the oracle's."""
    pytest.importorskip("torch")
    s = scenes("sphere256")
    upload(ctx, s)
    W, H = 1920, 1080
    cams = [orc.Camera(0.5, 0.7, 1.8), orc.Camera(1.1, 0.55, 2.1)]
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    wants, stats = zip(*[oracle_frame(orc, s, c.get_view(), c.get_pos(), W, H) for c in cams])
    for f, st in zip(frames, stats):
        gs = ctx.frame_stats(f)
        assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"])
    comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        _rehearse_every_rank(ctx, comm, frames, wants, (2, 4, 8), hip.RESIDENT_OCTREE, "config 3")
    finally:
        comm.close()


def test_config5_every_rank_of_8_gpus_at_full_size(ctx, orc, scenes, camera):
    """BASELINE config 5's split at its own workload: 512^3 sphere, 3840x2160, leaf triangles + shadow ray, 8 GPUs (rank 0
    gathers, ranks 1..7 render), every rank played by this GPU."""
    pytest.importorskip("torch")
    s = scenes("sphere512")
    view, pos = camera("sphere")
    W, H = 3840, 2160
    ctx.set_kernel(rto.KERNEL_AUTO)
    ctx.build_octree(s.grid.data, s.min, s.voxel)
    ctx.build_leaf_triangles(None)
    wt, wo = orc.build_leaf_triangles(s.grid, s.nodes)
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, st = orc.render_triangles(s.nodes, wt, wo, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=True, nthreads=min(16, orc.max_threads()))
    comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id(), band_rows=16)
    try:
        _rehearse_every_rank(ctx, comm, [f], [want], (8,), hip.RESIDENT_TRIANGLES_SHADOW, "config 5")
    finally:
        comm.close()


@pytest.mark.parametrize("what,allocs", [("frustum", 5), ("comm", 6)])
def test_partial_allocation_failures_leave_nothing_behind(what, allocs):
    """Fault injection (rto_debug_fault_alloc(k): the k-th buffer allocation fails): whichever of its allocations fails, a frustum
    update / a communicator submit reports the failure, keeps no half-allocated state, and the repeated call works."""
    import subprocess
    import sys

    for k in range(1, allocs + 1):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_alloc_worker.py"), what, str(k)], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, f"{what}, allocation {k}: " + p.stdout[-1500:] + p.stderr[-2500:]


def test_comm_group_of_one_gpu_renders_into_the_resident_frame(ctx, orc, scenes, camera):
    """rto_comm_create_all / rto_comm_render_resident_all (what RayTracerBVH::setDevices(n) drives) with the one GPU of
    this box: the frame lands in rank 0's resident framebuffer; a second context on the same device is refused."""
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 320, 200
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    grp = hip.CommGroup([ctx], band_rows=8)
    try:
        for _ in range(3):
            grp.render_resident(f)
        assert_bit_exact(ctx.download_resident(), want, "comm group, resident frame")
    finally:
        grp.close()
    other = rto.Context(0)
    try:
        with pytest.raises(rto.RtoError):
            hip.CommGroup([ctx, other])                                 # one context per GPU
    finally:
        other.close()


def test_comm_group_abort_with_a_batch_in_flight_then_destroy(ctx, orc, scenes, camera):
    """a rto_comm_create_all group shares its fate.  A batch is submitted, a member is aborted as a flush timeout would
    (ncclCommAbort), and then: every member reports dead, rto_comm_submit_all refuses BEFORE any member starts the batch (the
    group's slots stay in step), rto_comm_destroy of every member returns (it polls, never sits in hipStreamSynchronize behind
    a collective that cannot finish), and the context renders on with a fresh group.  ranks_seen is ncclCommCount's answer."""
    import torch
    s = scenes("sphere32")
    view, pos = camera("sphere")
    upload(ctx, s)
    W, H = 320, 200
    f = rto.make_frame(view, pos, W / H, 45.0, W, H)
    want, _ = oracle_frame(orc, s, view, pos, W, H)
    arr = hip.Context.frame_array([f, f])
    out = torch.full((2, H, W, 4), 7.0, dtype=torch.float32, device="cuda")
    grp = hip.CommGroup([ctx], band_rows=8)
    try:
        assert grp.ranks_seen() == 1
        grp.submit(arr, out.data_ptr(), H * W * 16)
        grp.flush()
        assert_bit_exact(out[1].cpu().numpy(), want, "group of one, batched submit")
        grp.submit(arr, out.data_ptr(), H * W * 16)                    # in flight
        grp.debug_abort(0)
        assert grp.is_dead() == [True]
        with pytest.raises(rto.RtoError) as e:
            grp.submit(arr, out.data_ptr(), H * W * 16)
        assert e.value.code == hip.RTO_E_INVALID and "dead" in str(e.value)
        with pytest.raises(rto.RtoError):
            grp.ranks_seen()
    finally:
        grp.close()                                                    # must return
    assert_bit_exact(ctx.render_host(f), want, "the context renders on after its group died")
    again = hip.CommGroup([ctx], band_rows=8)
    try:
        again.render_resident(f)
        assert_bit_exact(ctx.download_resident(), want, "a fresh group on the same context")
    finally:
        again.close()


def test_cpp_class_set_devices(orc, scenes):
    """RayTracerBVH::setDevices: with one device the reference call sequence is unchanged; asking for more GPUs than the
    box has fails loudly (no silent single-GPU fallback)."""
    W, H = 200, 120
    g = rto.VoxelGrid.test_sphere(32)
    root = rto.createOctreeFromVoxelGrid(g)
    cam = rto.Camera(0.5, 0.7, 1.8)
    s = scenes("sphere32")
    oc = orc.Camera(0.5, 0.7, 1.8)
    want, _ = oracle_frame(orc, s, oc.get_view(), oc.get_pos(), W, H)
    rt = rto.RayTracerBVH()
    rt.setDevices(1)
    rt.ensureComputeInitialized()
    rt.setOctree(root, g)
    rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, True)
    assert_bit_exact(rt.framebuffer(), want, "C++ class, setDevices(1)")
    import torch
    if torch.cuda.device_count() < 2:
        rt2 = rto.RayTracerBVH()
        rt2.setDevices(2)
        rt2.ensureComputeInitialized()
        rt2.setOctree(root, g)
        rt2.renderSceneCompute(cam, W, H, W / H, 45.0)
        assert rt2.framebuffer() is None or len(rt2.framebuffer()) == 0
    rto.freeOctree(root)


def test_bench_falls_back_to_replicas_when_the_split_cannot_run():
    """bench.py at N > 1 drives its frames through rto_comm_* (RCCL).  If the communicator cannot be made or its first batch fails on
    any rank, every rank renders whole frames on its own GPU and the line says so (`split_error`, `split_fallback`) instead of there
    being no line: exercised here on one GPU with a one-rank communicator whose probe batch is declared failed; and the healthy
    one-rank communicator still gives the split's own line."""
    import json
    import subprocess
    import sys

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--dim", "64", "--width", "320", "--height", "200",
            "--cpu-frames", "0", "--ramp-ms", "0", "--orbit-frames", "0", "--dropin-frames", "0", "--no-extras", "--force-comm"]
    for extra, failed in ((["--inject-split-failure"], True), ([], False)):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 6 and d["verified_against_oracle"] is True and d["value"] > 0
        if failed:
            assert "injected" in d["split_error"] and "WHOLE" in d["split_fallback"] and "one kernel launch per frame" in d["config"]["parallelism"]
            assert "single_frame_latency" not in d
        else:
            assert "split_error" not in d and "screen split over 1 GPUs" in d["config"]["parallelism"] and d["ranks_seen"] == 1


def test_bench_line_contract_small_run():
    """bench.py end to end on a small scene: one JSON line with the keys the driver reads, a verified frame, a roofline object
    and a CPU baseline (the default sizes are exercised by the driver itself; this guards the contract, not the numbers)."""
    import json
    import subprocess
    import sys

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--dim", "64", "--width", "320", "--height", "200",
           "--cpu-frames", "1", "--ramp-ms", "0", "--orbit-frames", "8", "--dropin-frames", "10"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["verified_against_oracle"] is True and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and d["roofline"]["bound"] == "valu_issue" and d["roofline"]["hbm_algorithmic"]["achieved"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["frames_per_launch"]["ms_per_frame"] > 0 and d["frames_per_launch"]["frames_equal_the_timed_frame"] is True and d["orbit"]["frames"] == 8
    assert d["dropin_call"]["ms_per_call"] > 0 and d["dropin_call"]["frame_equals_timed_frame"] is True
    assert len(d["config"]["workload"]) <= 120 and d["config"]["workload"].startswith("cfg2:") and len(d["config"]["clock_ramp"]) <= 120
    assert "one kernel launch per frame" in d["config"]["parallelism"] and d["roofline"]["kernel_ms_event_pair_median"] > 0
    # the legs behind the headline (round 5): configs 4 and 5, a cold context, camera jumps, the split rehearsed -- reduced sizes here
    for name in ("4", "5"):
        c = d["configs"][name]
        assert c["verified_against_oracle"] is True and c["ms_per_frame"] > 0 and c["Mrays_per_s"] > 0 and c["steps"] == 20 and c["hit_rays"] > 0
        assert c["workload"].startswith(f"cfg{name}:") and "roofline" in c
    assert d["configs"]["5"]["kernel"] == "k_trace_lean_triangles"
    assert d["cold"]["frames"] == 20 and len(d["cold"]["kernel_ms_each"]) == 20 and d["cold"]["frame_equals_timed_frame"] is True
    assert d["cold"]["first_frame_kernel_ms"] > 0 and d["cold"]["ms_per_frame"] > 0
    rc = d["random_cameras"]
    assert rc["cameras"] == 8 and rc["kernel_ms_max"] >= rc["kernel_ms_mean"] > 0 and rc["frames_verified_against_oracle"] == [0, 1, 2, 3]
    sr = d["split_rehearsal"]
    assert "skipped" not in sr, sr
    for k in ("N2_rank0", "N2_rank1"):
        assert sr[k]["ms_per_frame"] > 0 and sr[k]["render_ms_per_frame"] > 0 and sr[k]["gather_assemble_ms_per_frame"] > 0


@pytest.mark.parametrize("scene", ["sphere64", "calgary", "odd"])
def test_exact_grid_child_test_equals_the_general_one(ctx, orc, scenes, camera, scene):
    """Round 5: where the host proves that every node plane gridMin + k * voxelSize is computed without rounding (the test spheres,
    sceneCache.bin) the lean kernels compute 9 plane parameters per trip (fma, two adds) instead of 12 -- the same floats by
    construction.  Both forms against the oracle: pixels, per-pixel step counts, counters, the triangle path with shadow rays, inside
    and outside cameras; and a grid that is NOT exact (voxel 0.7) must be refused by the proof."""
    s = scenes(scene)
    upload(ctx, s)
    exact, used = ctx.debug_set_exact_grid(True)
    assert exact == used
    if scene in ("sphere64", "calgary"):
        assert exact, "the benchmark scenes' grids are exact: origin -0.5 / voxel 2^-k, origin (-2125, -1215, -150) / voxel 10"
    W, H = 320, 200
    cams = [camera("calgary_oblique" if scene == "calgary" else "sphere")]
    dims = np.array(s.grid.dims, np.float32)
    centre = np.asarray(s.min, np.float32) + 0.5 * dims * np.float32(s.voxel)
    inside = orc.Camera(1.1, 0.4, float(0.2 * dims.max() * s.voxel))
    inside.set_target(*[float(x) for x in centre])
    cams.append((inside.get_view(), inside.get_pos()))
    # eyes exactly ON a plane of the grid (phi = 0 puts the eye at x = target.x, theta = 0 at y = target.y; the target on a node plane):
    # plane parameters that are +0 / -0, the corner the sign-of-a-difference comparisons have to survive (fails_le)
    ext = float(dims.max() * s.voxel)
    plane = np.asarray(s.min, np.float32) + np.float32(s.voxel) * np.floor(dims / 2)          # a node plane on every axis
    for th, ph in ((0.4, 0.0), (0.0, 0.8), (0.0, 0.0)):
        c = orc.Camera(th, ph, 1.6 * ext)
        c.set_target(*[float(x) for x in plane])
        cams.append((c.get_view(), c.get_pos()))
    tris, off = orc.build_leaf_triangles(s.grid, s.nodes)
    ctx.upload_leaf_triangles(tris, off)
    try:
        for view, pos in cams:
            f = rto.make_frame(view, pos, W / H, 45.0, W, H)
            want, st = oracle_frame(orc, s, view, pos, W, H)
            steps = orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H)
            wt, _ = orc.render_triangles(s.nodes, tris, off, s.min, s.voxel, view, pos, W / H, 45.0, W, H, shadow=True)
            for on in (True, False):
                ctx.debug_set_exact_grid(on)
                what = f"{scene}, exact-grid form {'on' if on else 'off'}"
                for _ in range(2):
                    assert_bit_exact(ctx.render_host(f), want, what)
                np.testing.assert_array_equal(ctx.render_steps(f), steps, err_msg=what)
                gs = ctx.frame_stats(f)
                assert (gs["pops"], gs["hits"], gs["capped"]) == (st["pops"], st["hits"], st["capped"]), what
                if len(tris):
                    for _ in range(2):
                        assert_bit_exact(ctx.render_triangles_host(f, shadow=True), wt, what + ", triangles + shadow")
    finally:
        ctx.debug_set_exact_grid(True)
    # the proof refuses a grid whose products k * voxel round
    rng = np.random.default_rng(5)
    data = (rng.random((8, 8, 8)) < 0.4).astype(np.uint8)
    for gmin, voxel, expect in (((0.0, 0.0, 0.0), 0.7, False), ((-0.5, -0.5, -0.5), 0.125, True), ((0.1, 0.0, 0.0), 0.125, False),
                                ((-2125.0, -1215.0, -150.0), 10.0, True), ((1e8, 0.0, 0.0), 1.0, False)):
        g = orc.Grid((8, 8, 8), np.array(gmin, np.float32), np.float32(voxel), data)
        upload(ctx, Scene(g, orc.build_flat_octree(g)))
        assert ctx.debug_set_exact_grid(True) == (expect, expect), (gmin, voxel)


def test_camera_a_hair_outside_the_root_box(ctx, orc):
    """Ray origins a few ulps outside the root box, near its edges and corners, on a grid far from the world origin (coordinates
    ~1000, voxels of 500 ulps): child boxes may stick out of the root's by an ulp, so their planes can pass such an origin.
    The fold-free child test (waves whose rays all start outside the root box) must not be chosen then -- every kernel has to
    reproduce the oracle's pixels, step counts and counters."""
    rng = np.random.default_rng(77)
    dims = (32, 32, 32)
    data = (rng.random((32, 32, 32)) < 0.25).astype(np.uint8)
    data[8:24, 8:24, 8:24] = 1
    gmin = np.array([1000.0, -2000.0, 500.0], np.float32)
    voxel = np.float32(0.03125)
    g = orc.Grid(dims, gmin, voxel, data)
    s = Scene(g, orc.build_flat_octree(g))
    upload(ctx, s)
    W, H = 96, 64
    lo, hi = gmin, gmin + np.float32(32) * voxel
    centre = 0.5 * (lo + hi)
    for k in (1, 2, 3, 8, 40, 1000):
        for corner in ((0, 0, 0), (1, 0, 1), (1, 1, 1), (0, 1, 0)):
            pos = np.array([hi[a] if corner[a] else lo[a] for a in range(3)], np.float32)
            for a in range(3):                                           # k ulps outside on one axis, a little inside the faces of the others
                step = np.spacing(pos[a]) * np.float32(k)
                pos[a] = pos[a] + step if corner[a] else pos[a] - step
                if a != k % 3:
                    pos[a] = pos[a] - np.float32(0.2) if corner[a] else pos[a] + np.float32(0.2)
            cam = orc.Camera(0.0, 0.0, 1.0)
            cam.set_target(*[float(x) for x in centre])
            view = orc.look_at(pos, centre) if hasattr(orc, "look_at") else None
            if view is None:
                # a view matrix looking from pos towards the centre: reuse the oracle's camera at the same direction
                d = centre - pos
                r = float(np.linalg.norm(d))
                cam = orc.Camera(float(np.arcsin(np.clip(-d[1] / r, -1, 1))), float(np.arctan2(-d[0], -d[2])), r)
                cam.set_target(*[float(x) for x in centre])
                view = cam.get_view()
            f = rto.make_frame(view, pos, W / H, 60.0, W, H)
            want, st = oracle_frame(orc, s, view, pos, W, H, fov=60.0)
            steps = orc.render_steps(s.nodes, s.min, s.voxel, view, pos, W / H, 60.0, W, H)
            for kname, kernel in KERNELS:
                ctx.set_kernel(kernel)
                what = f"{k} ulps outside corner {corner}, {kname}"
                assert_bit_exact(ctx.render_host(f), want, what)
                np.testing.assert_array_equal(ctx.render_steps(f), steps, err_msg=what)
    ctx.set_kernel(rto.KERNEL_AUTO)
