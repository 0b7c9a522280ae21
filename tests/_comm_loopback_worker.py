"""Child process of test_gpu_parity.py::test_comm_ranks_exchange_through_a_loopback_transport (run with the directory of the built
tests/rccl_shim in front of LD_LIBRARY_PATH; never imports torch, which would map the real librccl under the same SONAME).

Every rank of a 2 / 3 / 4 / 5 / 8-GPU split as its OWN rto_context + rto_comm (rto_comm_create: the multi-process entry point) in this
one process on the one GPU; the product's submit / pack / comm_exchange / assemble run unchanged, only the transport under ncclSend /
ncclRecv is the shim's device-to-device copy.  Rank 0's assembled frames must be the oracle's whole frames, bit for bit."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ray_tracing_octrees_amd as rto  # noqa: E402
from oracle import orc  # noqa: E402
from ray_tracing_octrees_amd import hip  # noqa: E402


def bits_differ(a, b):
    return int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum())


def main():
    shim = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)          # the same object rto_comm.inc's dlopen will get
    assert hasattr(shim, "rccl_shim_pending"), "the real librccl was found instead of tests/rccl_shim: LD_LIBRARY_PATH not set by the test?"
    shim.rccl_shim_pending.restype = C.c_int
    shim.rccl_shim_size_mismatches.restype = C.c_int
    hipl = C.CDLL("libamdhip64.so")
    hipl.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hipl.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hipl.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hipl.hipFree.argtypes = [C.c_void_p]

    g = orc.test_sphere_grid(64)
    nodes = orc.build_flat_octree(g)
    tris, off = orc.build_leaf_triangles(g, nodes)
    W, H = 417, 250                                            # 16 bands, the last one 10 rows; a width that is no multiple of anything
    cams = [orc.Camera(0.5 + 0.3 * i, 0.7 + 0.05 * i, 1.8) for i in range(3)]
    near = orc.Camera(0.9, 0.4, 0.75)                          # the sphere fills the frame: nothing to crop
    cams.append(near)
    frames = [rto.make_frame(c.get_view(), c.get_pos(), W / H, 45.0, W, H) for c in cams]
    threads = min(16, orc.max_threads())
    want_oct = [orc.render(nodes, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H, nthreads=threads, out=np.zeros((H, W, 4), np.float32))[0]
                for c in cams]
    want_tri = [orc.render_triangles(nodes, tris, off, g.min, g.voxel_size, c.get_view(), c.get_pos(), W / H, 45.0, W, H, shadow=True)[0] for c in cams]

    nf = len(frames)
    d_out = C.c_void_p()
    assert hipl.hipMalloc(C.byref(d_out), nf * H * W * 16) == 0
    host = np.empty((nf, H, W, 4), np.float32)
    checked = 0
    for world in (2, 3, 4, 5, 8):
        ctxs = [rto.Context(0) for _ in range(world)]
        for c in ctxs:
            c.upload_octree(nodes, g.min, g.voxel_size)
            c.upload_leaf_triangles(tris, off)
        uid = hip.comm_unique_id()
        comms = [hip.Comm(ctxs[r], world, r, uid, band_rows=16) for r in range(world)]
        try:
            assert all(cm.ranks_seen() == world for cm in comms)
            for mode, wants, what in ((hip.RESIDENT_OCTREE, want_oct, "octree"), (hip.RESIDENT_TRIANGLES_SHADOW, want_tri, "triangles + shadow")):
                for batch in ([0, 1, 2, 3], [3], [1, 3]):      # a batch of four (windows cropped to the geometry), the uncroppable frame alone, a mixed pair
                    arr = hip.Context.frame_array([frames[i] for i in batch])
                    for rep in range(2):                       # both buffer sets of every rank
                        assert hipl.hipMemset(d_out, 0x55, nf * H * W * 16) == 0
                        for r in range(world - 1, 0, -1):      # the senders first: their sends wait in the shim for rank 0's receives
                            comms[r].submit(arr, 0, 0, mode)
                        comms[0].submit(arr, d_out.value, H * W * 16, mode)
                        for cm in comms:
                            cm.flush(20000)
                        assert shim.rccl_shim_pending() == 0, "a send or a receive found no counterpart"
                        assert hipl.hipMemcpy(host.ctypes.data_as(C.c_void_p), d_out, len(batch) * H * W * 16, 2) == 0
                        for k, i in enumerate(batch):
                            bad = bits_differ(host[k], wants[i])
                            assert bad == 0, f"world {world}, {what}, batch {batch} frame {i} (pass {rep}): {bad} pixels differ from the oracle's whole frame"
                            checked += 1
            assert shim.rccl_shim_size_mismatches() == 0, "a sender and rank 0 disagreed on the element count of a part"
        finally:
            for cm in comms:
                cm.close()
            for c in ctxs:
                c.close()
    hipl.hipFree(d_out)
    print(f"loopback transport: {checked} assembled frames equal the oracle's, worlds 2 3 4 5 8")


if __name__ == "__main__":
    main()
