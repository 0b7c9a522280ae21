"""Where config 5's frame spends its instructions: an A/B build of the library with -DRTO_TRI_PROFILE counts, per frame, the
loop trips of the node phase (of each wave's longest lane), the lanes walking in them, the triangle rounds, chunks and pairs.
Build first:  tools/build_variants.sh triprof "-DRTO_TRI_PROFILE"   then on the GPU box:
    RTO_HIP_LIB=build/variants/librto_hip_triprof.so python tools/tri_profile.py [dim W H]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_octrees_amd as rto

dim, W, H = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 3840, 2160)
g = rto.VoxelGrid.test_sphere(dim)
ctx = rto.Context(0)
ctx.build_octree(g.data, g.min, g.voxelSize)
ctx.build_leaf_triangles(None)
cam = rto.Camera(0.5, 0.7, 1.8)
f = rto.make_frame(cam.getView(), cam.getPos(), W / H, 45.0, W, H)
import torch
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(3):
    ctx.render_triangles_device(f, buf.data_ptr(), True)
ctx.synchronize()
L = rto.hip.load()
# zero the counters through the instrumented frame path, then one colour frame
_, st0 = ctx.render_triangles_host(f, shadow=True, stats=True)
out = (C.c_uint64 * 3)()
L.rto_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
L.rto_debug_counters(ctx._h, out, 1)        # zero
ctx.render_triangles_device(f, buf.data_ptr(), True)
ctx.synchronize()
L.rto_debug_counters(ctx._h, out, 0)
trips, rounds = out[0] >> 32, out[0] & 0xffffffff
lanes = out[1]
chunks, pairs = out[2] >> 32, out[2] & 0xffffffff
print(f"stats frame: pops {st0['pops']} hits {st0['hits']}")
print(f"wave trips (longest lane) {trips}  lanes walking beside it {lanes} -> node-phase lane utilisation {lanes / max(1, trips) / 64:.3f}")
print(f"triangle rounds {rounds}  chunks {chunks}  pairs {pairs} -> pairs per chunk {pairs / max(1, chunks):.1f} ({pairs / max(1, chunks) / 64:.2f} of 64), chunks per round {chunks / max(1, rounds):.2f}")
print(f"estimate: node phase {trips * 135 / 1e6:.0f} M instr, triangle chunks {chunks * 110 / 1e6:.0f} M, round overhead {rounds * 60 / 1e6:.0f} M")
