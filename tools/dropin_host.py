"""tools/dropin_host.py -- is the drop-in loop bound by the host?  Time to ISSUE 200 renderSceneComputeWithCulling calls (no wait) and time until
the GPU has finished them, with and without the frustum update, with the update's host-side proof on and off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_octrees_amd as rto
from ray_tracing_octrees_amd import hip

g = rto.VoxelGrid.test_sphere(256)
cam = rto.Camera(0.5, 0.7, 1.8)
W, H = 1920, 1080
rt = rto.RayTracerBVH()
rt.ensureComputeInitialized()
rt.setOctreeFromGrid(g)
rt.setFrustumCullingEnabled(True)
L = hip.load()
WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(WARM):
    rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, True)
rt.finish()
for name, update, proof in (("update, proven on the host", True, 1), ("update, kernel forced", True, 0), ("no update", False, 1), ("update, proven on the host", True, 1), ("no update", False, 1)):
    L.rto_debug_set_frustum_shortcut(rt.context_handle, proof)
    for _ in range(20):
        rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, update)
    best = None
    runs = []
    for _ in range(5):
        rt.finish()
        t0 = time.perf_counter()
        for _ in range(200):
            rt.renderSceneComputeWithCulling(cam, W, H, W / H, 45.0, update)
        t1 = time.perf_counter()
        rt.finish()
        t2 = time.perf_counter()
        r = ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6)
        runs.append(r[1])
        best = r if best is None or r[1] < best[1] else best
    print(f"{name:28s}: issued in {best[0]:5.1f} us per call, finished in {best[1]:5.1f} us per call; all runs: " + " ".join(f"{x:.1f}" for x in runs), flush=True)
