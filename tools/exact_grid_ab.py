"""A/B of the exact-grid child test (9 plane parameters per trip where the host proves the grid's planes exact) against the general
12-plane form on the same library: configs 2, 4, 5, kernel time by event pairs, interleaved.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ray_tracing_octrees_amd as rto

def scene(cfg):
    if cfg == "4":
        z = np.load(os.path.join(ROOT, "tests", "golden", "ref_scene_cache.npz"))
        dims = tuple(int(x) for x in z["dims"])
        data = np.unpackbits(z["packed"])[: dims[0] * dims[1] * dims[2]].reshape(dims[2], dims[1], dims[0])
        return rto.VoxelGrid.from_array(data, z["min"].astype(np.float32), np.float32(z["voxel"])), rto.Camera(0.6, 0.5, 3500.0), 1920, 1080, False
    dim = 256 if cfg == "2" else 512
    return rto.VoxelGrid.test_sphere(dim), rto.Camera(0.5, 0.7, 1.8), *((1920, 1080, False) if cfg == "2" else (3840, 2160, True))

s = torch.cuda.Stream(); torch.cuda.set_stream(s)
for cfg in ("2", "4", "5"):
    g, cam, W, H, tri = scene(cfg)
    ctx = rto.Context(0)
    ctx.build_octree(g.data, g.min, g.voxelSize)
    if tri: ctx.build_leaf_triangles(None)
    f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
    fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    render = (lambda: ctx.render_triangles_device(f, fb.data_ptr(), True, None, s.cuda_stream)) if tri else (lambda: ctx.render_device(f, fb.data_ptr(), None, s.cuda_stream))
    for _ in range(2000 if not tri else 300): render()
    torch.cuda.synchronize()
    res = {True: [], False: []}
    frames = {}
    for rep in range(6):
        for on in (True, False):
            exact, used = ctx.debug_set_exact_grid(on)
            for _ in range(20): render()
            torch.cuda.synchronize()
            n = 200 if not tri else 40
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(s)
            for _ in range(n): render()
            eb.record(s); torch.cuda.synchronize()
            res[on].append(ea.elapsed_time(eb) / n * 1e3)
            frames[on] = fb.clone()
    same = bool(torch.equal(frames[True], frames[False]))
    a, b = sorted(res[True])[len(res[True]) // 2], sorted(res[False])[len(res[False]) // 2]
    print(f"config {cfg}: grid exact {exact}; 9-plane form {a:.2f} us per frame, general form {b:.2f} us ({100 * (a / b - 1):+.1f} %); frames identical: {same}")
    ctx.close()
