"""The oracle against the golden vectors (CPU only).

ref_* fixtures were produced by the reference's own compiled code (tests/golden/make_golden.py);
orc_* fixtures pin the oracle's kernel restatement against accidental change.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import SPHERE_CAM


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["sphere16", "sphere32", "odd"])
def test_flat_octree_matches_reference_arrays(scenes, golden, name):
    want = golden("ref_octrees_small.npz")[name]
    got = scenes(name).nodes
    assert got.dtype.itemsize == 60
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("name", ["sphere16", "sphere32", "sphere64", "sphere128", "sphere256", "calgary"])
def test_flat_octree_counts_and_hash(scenes, golden_meta, name):
    m = golden_meta["octrees"][name]
    a = scenes(name).nodes
    assert len(a) == m["nodes"]
    assert int((a["isLeaf"] == 1).sum()) == m["leaves"]
    assert int(((a["isLeaf"] == 1) & (a["isSolid"] == 1)).sum()) == m["solid_leaves"]
    assert int(a["size"][0]) == m["root_size"]
    assert sha(a) == m["sha256"]
    np.testing.assert_array_equal(scenes(name).min, np.array(m["grid_min"], np.float32))


def test_survey_node_counts(golden_meta):
    # SURVEY.md section 6 / BASELINE.md: the figures every later report quotes
    assert golden_meta["octrees"]["sphere64"]["nodes"] == 23561
    assert golden_meta["octrees"]["sphere256"]["nodes"] == 374921
    assert golden_meta["octrees"]["sphere512"]["nodes"] == 1500361
    assert golden_meta["octrees"]["calgary"]["nodes"] == 348409
    assert golden_meta["octrees"]["calgary"]["filled"] == 141000


def test_octree_structure_invariants(scenes):
    a = scenes("sphere64").nodes
    internal = a["isLeaf"] == 0
    assert (a["isLeaf"] == a["isUniform"]).all()                       # leaf <=> uniform (OctreeVoxel.cpp:716-745)
    assert (a["child"][~internal] == -1).all()
    c0 = a["child"][internal][:, 0]
    assert (a["child"][internal] == c0[:, None] + np.arange(8)).all()  # 8 consecutive children (RayTracerBVH.cpp:473-487)
    assert (c0 % 8 == 1).all()
    k = np.arange(8)
    for name, bit in (("x", 1), ("y", 2), ("z", 4)):                   # child order bit0=x, bit1=y, bit2=z
        child = a[name][a["child"][internal]]
        half = (a["size"][internal] // 2)[:, None]
        assert (child == a[name][internal][:, None] + np.where(k & bit, half, 0)).all()


def test_camera_and_glm_match_reference(orc, golden):
    z = golden("ref_cameras.npz")
    for name in ("sphere", "calgary_default", "calgary_oblique", "panned"):
        t, p, r, do_pan, dx, dy = [float(v) for v in z[name + "_params"]]
        cam = orc.Camera(t, p, r)
        if do_pan:
            cam.pan(dx, dy)
        assert cam.get_view().tobytes() == z[name + "_view"].tobytes(), name
        assert cam.get_pos().tobytes() == z[name + "_pos"].tobytes(), name
        assert cam.target.tobytes() == z[name + "_target"].tobytes(), name
        assert orc.mat4_inverse(z[name + "_view"]).tobytes() == z[name + "_inv"].tobytes(), name
        persp = orc.perspective(orc.radians(45.0), float(np.float32(1920 / 1080)), 0.01, 5000.0)
        assert persp.tobytes() == z[name + "_persp"].tobytes(), name
        assert orc.mat4_mul(persp, z[name + "_view"]).tobytes() == z[name + "_vp"].tobytes(), name


def test_frustum_matches_reference(orc, golden):
    z = golden("ref_cameras.npz")
    mins, maxs = z["frustum_min"], z["frustum_max"]
    seen = set()
    for cam in ("calgary_default", "calgary_oblique", "sphere"):
        planes = orc.frustum_planes(z[cam + "_vp"])
        for margin in (150.0, 0.0):
            want = z[f"frustum_{cam}_m{int(margin)}"]
            got = np.array([orc.frustum_test(planes, mins[i], maxs[i], margin) for i in range(len(mins))], np.int32)
            np.testing.assert_array_equal(got, want)
            seen |= set(want.tolist())
    assert seen == {-1, 0, 1}      # the vectors exercise all three verdicts


def test_oracle_kernel_pinned_small_images(orc, scenes, golden, golden_meta):
    z = golden("orc_images_small.npz")
    for dim, (W, H) in ((16, (64, 64)), (32, (96, 64))):
        s = scenes(f"sphere{dim}")
        cam = orc.Camera(*SPHERE_CAM)
        img, st = orc.render(s.nodes, s.min, s.voxel, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        assert img.tobytes() == z[f"sphere{dim}_{W}x{H}_rgba"].tobytes()
        steps = orc.render_steps(s.nodes, s.min, s.voxel, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
        np.testing.assert_array_equal(steps, z[f"sphere{dim}_{W}x{H}_steps"])
        m = golden_meta["images"][f"sphere{dim}_{W}x{H}"]
        assert {k: int(st[k]) for k in ("pops", "hits", "capped", "max_stack")} == {k: m[k] for k in ("pops", "hits", "capped", "max_stack")}


@pytest.mark.parametrize("dim,W,H", [(64, 512, 512), (256, 1920, 1080)])
def test_oracle_kernel_pinned_config_images(orc, scenes, camera, golden_meta, dim, W, H):
    s = scenes(f"sphere{dim}")
    view, pos = camera("sphere")
    out = np.zeros((H, W, 4), np.float32)
    img, st = orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=min(8, orc.max_threads()), out=out)
    m = golden_meta["images"][f"sphere{dim}_{W}x{H}"]
    assert sha(img) == m["sha256"]
    assert (st["pops"], st["hits"], st["capped"]) == (m["pops"], m["hits"], m["capped"])


def test_survey_render_statistics(golden_meta):
    # 65,009 hits at config 1; 280,670 hits and 434 capped rays at config 2
    assert golden_meta["images"]["sphere64_512x512"]["hits"] == 65009
    assert golden_meta["images"]["sphere256_1920x1080"]["hits"] == 280670
    assert golden_meta["images"]["sphere256_1920x1080"]["capped"] == 434
    assert golden_meta["images"]["calgary_default_1300x1300"]["hits"] == 1300 * 1300   # eye inside a solid leaf


def test_oracle_threads_and_row_ranges_agree(orc, scenes, camera):
    s = scenes("sphere32")
    view, pos = camera("sphere")
    W, H = 80, 60
    a, _ = orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=1)
    b, _ = orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, nthreads=4)
    assert a.tobytes() == b.tobytes()
    c = np.zeros_like(a)
    for y0, y1 in ((0, 7), (7, 33), (33, 60)):
        orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, rows=(y0, y1), out=c)
    assert a.tobytes() == c.tobytes()


def test_cull_compact_properties(orc, scenes, camera):
    s = scenes("calgary")
    view, _ = camera("calgary_oblique")
    out, vis = orc.cull_compact(s.nodes, s.min, s.voxel, view, 45.0, 1920 / 1080)
    assert 0 < len(out) < len(s.nodes)                     # the oblique camera does cull something
    assert len(out) == int(vis.sum())
    kept = s.nodes[vis]
    for f in ("x", "y", "z", "size", "isLeaf", "isSolid", "isUniform"):
        np.testing.assert_array_equal(out[f], kept[f])
    ch = out["child"]
    assert ((ch >= -1) & (ch < len(out))).all()
    # with the unit sphere everything lies within the 150-unit margin: nothing is culled
    sp = scenes("sphere32")
    v2, _ = camera("sphere")
    out2, vis2 = orc.cull_compact(sp.nodes, sp.min, sp.voxel, v2, 45.0, 1.0)
    assert vis2.all() and out2.tobytes() == sp.nodes.tobytes()


def test_octree_ray_skip_returns_a_solid_leaf_entry(orc, scenes):
    """N1 (VolumeRaycastRenderer.cpp:50-155).  The reference orders children by Hamming distance from the
    octant of the ray's POSITIVE direction bits (:114-152), which is not front-to-back for positive
    components, so the result is the entry distance of the first solid leaf met in THAT order: some
    solid leaf the ray really crosses (never closer than the nearest one), or 1e30 when there is none."""
    s = scenes("sphere16")
    rng = np.random.default_rng(7)
    ro = np.array([0.9, 0.7, 1.3], np.float32)
    leaves = s.nodes[(s.nodes["isLeaf"] == 1) & (s.nodes["isSolid"] == 1)]
    mn = s.min[None, :] + np.stack([leaves["x"], leaves["y"], leaves["z"]], 1).astype(np.float32) * s.voxel
    mx = mn + (leaves["size"].astype(np.float32) * s.voxel)[:, None]
    hits = nearest = 0
    for _ in range(200):
        tgt = (rng.random(3).astype(np.float32) - 0.5) * np.float32(0.9)
        rd = tgt - ro
        rd = (rd / np.sqrt((rd * rd).sum())).astype(np.float32)
        t = orc.octree_ray_skip(s.nodes, s.min, s.voxel, ro, rd)
        inv = np.float32(1.0) / rd
        t1, t2 = (mn - ro) * inv, (mx - ro) * inv
        tn = np.minimum(t1, t2).max(axis=1)
        tf = np.maximum(t1, t2).min(axis=1)
        ok = (np.maximum(tn, 0) <= tf)
        if ok.any():
            hits += 1
            entries = np.maximum(tn[ok], 0)
            assert np.abs(entries - t).min() <= 1e-5 * max(1.0, t)      # it is one of the crossed leaves
            assert t >= float(entries.min()) - 1e-5
            nearest += int(abs(t - float(entries.min())) <= 1e-5)
        else:
            assert t >= 1e30
    assert hits > 50


RAY_SKIP_CASES = ("probe", "random", "inside", "axis_inside", "axis_outside0", "axis_outside1", "axis_outside2", "narrow")


@pytest.mark.parametrize("scene", ["sphere32", "odd", "calgary"])
def test_octree_ray_skip_equals_the_reference(orc, scenes, golden, scene):
    """N1 pinned: tests/golden/ref_ray_skip.npz holds the distances the reference's OWN compiled octreeRaySkip returned
    (453-skeleton/VolumeRaycastRenderer.cpp:50-155 through oracle/ref_shim_vr.cpp): the 7x7 probe rays of drawRaycast,
    random rays from outside and inside, axis-parallel and nearly axis-parallel rays (the 1e-10 clamp, :83-87, and -0.0),
    a narrowed [tMin, tMax] and two random visibility maps (:64-67).  The oracle must return the same float bits."""
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    finite = 0
    for tag in RAY_SKIP_CASES + ("vis0", "vis1"):
        key = f"{scene}_{tag}"
        ro, rd, (tmin, tmax), want = z[key + "_ro"], z[key + "_rd"], z[key + "_t"], z[key + "_out"]
        vis = None
        if tag.startswith("vis"):
            vis = np.unpackbits(z[key + "_flags"])[: len(s.nodes)]
        got = orc.octree_ray_skip_many(s.nodes, s.min, s.voxel, ro, rd, float(tmin), float(tmax), visible=vis)
        assert got.tobytes() == want.tobytes(), f"{key}: {int((got.view(np.uint32) != want.view(np.uint32)).sum())} of {len(want)} distances differ"
        finite += int((want < 1e30).sum())
    assert finite > 100


PIXEL_CAMS = {"sphere32": 2, "odd": 1, "calgary": 1}


@pytest.mark.parametrize("scene", ["sphere32", "odd", "calgary"])
def test_nearest_hit_render_mode_equals_the_reference_distances(orc, scenes, golden, scene):
    """N1 as a render mode: for every pixel of a 96 x 64 frame, the reference's compiled octreeRaySkip was run on the ray of
    generateRay (tests/golden/ref_ray_skip.npz "pixels": make_golden_ray_skip.py); the oracle's render_skip must return those
    distances bit for bit -- without and with a visibility map, from outside and (sphere32) from inside the shell's hollow --
    and its colours are the reference's shade applied at exactly those hits."""
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    W, H = 96, 64
    flags = np.unpackbits(z[f"{scene}_vis1_flags"])[: len(s.nodes)]
    for ci in range(PIXEL_CAMS[scene]):
        t, p, r = [float(x) for x in z[f"{scene}_pixels{ci}_cam"]]
        cam = orc.Camera(t, p, r)
        view, pos = cam.get_view(), cam.get_pos()
        for vis, key in ((None, "_out"), (flags, "_vis_out")):
            rgba, dist = orc.render_skip(s.nodes, s.min, s.voxel, view, pos, W / H, 45.0, W, H, visible=vis)
            want = z[f"{scene}_pixels{ci}{key}"].reshape(H, W)
            assert dist.tobytes() == want.tobytes(), f"{scene} camera {ci} {key}: {int((dist.view(np.uint32) != want.view(np.uint32)).sum())} distances differ"
            hit = want < 1e30
            assert ((rgba[..., :3] == 0).all(axis=-1) == ~hit).all() and (rgba[..., 3] == 1).all()
            assert (rgba[hit][:, 0] >= np.float32(0.1)).all() and (rgba[hit][:, 0] <= np.float32(1.1)).all()
        # the same rays one by one: the render mode is octreeRaySkip on generateRay's directions, nothing else
        rd = orc.generate_rays(view, pos, W / H, 45.0, W, H).reshape(-1, 3)
        some = np.arange(0, W * H, 97)
        one = orc.octree_ray_skip_many(s.nodes, s.min, s.voxel, pos, rd[some])
        assert one.tobytes() == z[f"{scene}_pixels{ci}_out"][some].tobytes()


@pytest.mark.parametrize("scene,cam", [("sphere32", (0.5, 0.7, 1.8)), ("odd", (0.4, 0.9, 9.0)), ("calgary", (0.6, 0.5, 3500.0))])
def test_probe_consumer_equals_the_reference(orc, scenes, golden, scene, cam):
    """octreeRaySkip's consumer in drawRaycast (S/VolumeRaycastRenderer.cpp:1602-1663): the oracle's 7 x 7 probe directions are
    the ones the reference's own glm calls produce (compiled: refvr_probe_rays), and its skip distance -- 15th percentile of the
    valid distances x 0.75, blended 0.4 old + 0.6 new -- follows, over three consecutive frames without and three with a
    visibility map, the sequence computed from the compiled function's distances."""
    z = golden("ref_ray_skip.npz")
    s = scenes(scene)
    c = orc.Camera(*cam)
    view, eye = c.get_view(), c.get_pos()
    aspect = float(np.float32(1920 / 1080))
    assert orc.probe_rays(view, eye, aspect).tobytes() == z[f"{scene}_probe_rd"].tobytes(), "probe directions"
    want = z[f"{scene}_probe_skip_seq"]
    flags = np.unpackbits(z[f"{scene}_vis0_flags"])[: len(s.nodes)]
    last, got = np.float32(0.0), []
    for k in range(6):
        last = orc.probe_skip_distance(s.nodes, s.min, s.voxel, view, eye, aspect, last, visible=None if k < 3 else flags)
        got.append(last)
    assert np.array(got, np.float32).tobytes() == want.tobytes(), (got, want)
    if scene != "odd":
        assert want[2] > 0 and abs(want[2] - want[1]) < abs(want[1] - want[0])        # converging towards 0.75 x the percentile


@pytest.mark.skipif(not os.path.exists("/root/reference/453-skeleton"), reason="reference checkout not present")
def test_octree_ray_skip_against_live_reference_random_grids(orc):
    """Where the reference sources exist: the oracle against the compiled octreeRaySkip itself, on random grids and rays."""
    if not orc.refvr_available():
        pytest.skip("oracle/_ref/libref_vr.so not built (make -C oracle refvr)")
    rng = np.random.default_rng(321)
    for dims, p in (((7, 5, 3), 0.5), ((16, 16, 16), 0.1), ((33, 9, 20), 0.7), ((2, 3, 1), 1.0)):
        data = (rng.random((dims[2], dims[1], dims[0])) < p).astype(np.uint8)
        g = orc.Grid(dims, np.array([0.5, -2.0, 3.0], np.float32), np.float32(0.3), data)
        nodes = orc.build_flat_octree(g)
        ro = np.array([4.0, 1.5, -3.0], np.float32)
        rd = rng.normal(size=(200, 3)).astype(np.float32)
        rd /= np.linalg.norm(rd, axis=1, keepdims=True).astype(np.float32)
        rd[:6] = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [1e-11, 1, 0], [-0.0, 0, -1], [1, 1e-9, -1e-11]], np.float32)
        flags = (rng.random(len(nodes)) < 0.8).astype(np.uint8)
        flags[0] = 1
        for vis in (None, flags):
            want = orc.ref_octree_ray_skip(g, ro, rd, 0.0, 1e30, visible=vis)
            got = orc.octree_ray_skip_many(nodes, g.min, g.voxel_size, ro, rd, 0.0, 1e30, visible=vis)
            assert got.tobytes() == want.tobytes(), (dims, vis is not None)


@pytest.mark.skipif(not os.path.exists("/root/reference/453-skeleton"), reason="reference checkout not present")
def test_oracle_against_live_reference_random_grids(orc):
    """Where the reference sources exist, compare against their compiled code directly on random grids."""
    assert orc.ref_available()
    rng = np.random.default_rng(99)
    for dims, p in (((7, 5, 3), 0.5), ((16, 16, 16), 0.1), ((33, 9, 20), 0.9), ((1, 1, 1), 1.0), ((2, 3, 1), 0.0)):
        data = (rng.random((dims[2], dims[1], dims[0])) < p).astype(np.uint8)
        g = orc.Grid(dims, np.array([0.5, -2.0, 3.0], np.float32), np.float32(0.3), data)
        assert orc.build_flat_octree(g).tobytes() == orc.ref_build_flat_octree(g).tobytes(), dims


def test_render_triangles_refuses_arrays_of_the_wrong_shape(orc, scenes):
    """The oracle's C side indexes tris / tri_offset unchecked: a camera-aim array passed as tri_offset was a core dump on the GPU
    box in round 3 (gpurun_out/r3_t33.log).  The wrapper now raises instead."""
    s = scenes("sphere16")
    tris, off = orc.build_leaf_triangles(s.grid, s.nodes)
    cam = orc.Camera(0.5, 0.7, 1.8)
    args = (s.min, s.voxel, cam.get_view(), cam.get_pos(), 1.0, 45.0, 16, 16)
    orc.render_triangles(s.nodes, tris, off, *args)
    for bad_off in (np.zeros(3, np.float32), off[:-1], off + 1, off[::-1].copy()):
        with pytest.raises(ValueError):
            orc.render_triangles(s.nodes, tris, bad_off, *args)
    with pytest.raises(ValueError):
        orc.render_triangles(s.nodes, tris[:, :9], off, *args)
    with pytest.raises(ValueError):
        orc.render_triangles(s.nodes, tris, off, s.min, s.voxel, cam.get_view()[:3], cam.get_pos(), 1.0, 45.0, 16, 16)


def _glsl_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_glsl", os.path.join(os.path.dirname(__file__), "golden", "make_golden_glsl.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("scene", ["sphere32", "sphere64", "odd", "calgary"])
def test_oracle_equals_the_reference_glsl_text_under_glm(orc, scenes, golden, scene):
    """The mechanical cross-check of rows A3-A6.  The reference's GLSL functions (RayTracerBVH.cpp:221-355, and the earlier
    block-commented closest-hit shader at :46-166) were cut out of the reference file by `make -C oracle glsl`, compiled as C++
    against the reference's vendored glm (`out T x` -> `T& x` the only edit) and run over whole frames: outside, inside-the-shell
    and axis-aligned cameras, the odd grid, the shipped Calgary grid.  The oracle -- the hand restatement every GPU test compares
    with -- must produce the same float bits for every pixel: against the recorded frames always, against the compiled text
    itself where oracle/_ref/libglsl_*.so exists.  (glm stands in for a GLSL compiler: corroboration, not an execution of the
    reference's kernel.)"""
    m = _glsl_cases()
    z = golden("glsl_images_small.npz")
    s = scenes(scene)
    W, H, cams = m.CASES[scene]
    for i, spec in enumerate(cams):
        view, pos, fov = m.camera(spec)
        first, _ = orc.render(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H)
        closest, st = orc.render_closest(s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H)
        assert first.tobytes() == z[f"{scene}_cam{i}_first"].tobytes(), f"{scene} camera {i}: first-hit oracle vs the shader text"
        assert closest.tobytes() == z[f"{scene}_cam{i}_closest"].tobytes(), f"{scene} camera {i}: closest-hit oracle vs the earlier shader's text"
        assert st["capped"] == 0 and st["hits"] == int((closest[..., 0] != 0).sum())
        if orc.glsl_available():
            assert orc.glsl_render("first", s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H).tobytes() == first.tobytes()
            assert orc.glsl_render("closest", s.nodes, s.min, s.voxel, view, pos, W / H, fov, W, H).tobytes() == closest.tobytes()


def test_closest_hit_is_never_farther_than_first_hit(orc, scenes):
    """What distinguishes the two traversal rules of the reference: the live shader stops at the first accepted solid leaf in LIFO
    order (and after 512 pops), the earlier one goes on until nothing nearer is left.  So every first-hit pixel is a closest-hit
    pixel too, and where they differ the closest-hit leaf is at least as near."""
    s = scenes("sphere32")
    cam = orc.Camera(2.0, 0.3, 0.2)                       # inside the shell: the two rules disagree on thousands of pixels
    W, H = 96, 64
    first, st1 = orc.render(s.nodes, s.min, s.voxel, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
    closest, st2 = orc.render_closest(s.nodes, s.min, s.voxel, cam.get_view(), cam.get_pos(), W / H, 45.0, W, H)
    assert st2["hits"] >= st1["hits"] and ((first[..., 0] != 0) <= (closest[..., 0] != 0)).all()
    assert (first.view(np.uint32) != closest.view(np.uint32)).any()
