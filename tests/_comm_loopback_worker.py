"""Child process of test_gpu_parity.py::test_comm_ranks_exchange_through_a_loopback_transport (run with the directory of the built
tests/rccl_shim in front of LD_LIBRARY_PATH; never imports torch, which would map the real librccl under the same SONAME).
argv[1] = "serial" (one host thread drives all ranks, a flush after every batch) or "threads" (one host thread per rank, six batches
in flight before the flush, RTO_RCCL_SHIM_RENDEZVOUS=1: a send / receive returns when it has met its counterpart).

Every rank of a 2 / 3 / 4 / 5 / 8-GPU split as its OWN rto_context + rto_comm (rto_comm_create: the multi-process entry point) in this
one process on the one GPU; the product's submit / pack / comm_exchange / assemble run unchanged, only the transport under ncclSend /
ncclRecv is the shim's device-to-device copy.  Rank 0's assembled frames must be the oracle's whole frames, bit for bit."""
import ctypes as C
import os
import sys
import threading

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
    if len(sys.argv) > 1 and sys.argv[1] == "threads":
        assert os.environ.get("RTO_RCCL_SHIM_RENDEZVOUS") == "1"
        return threaded(shim, hipl, g, nodes, tris, off, frames, want_oct, want_tri, W, H)
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


def threaded(shim, hipl, g, nodes, tris, off, frames, want_oct, want_tri, W, H):
    """One host thread per rank (what N processes do, as far as one process can): every rank submits six batches of four frames back to
    back -- the two buffer sets of every rank alternate, batch k's gather overlaps batch k + 1's render -- and flushes once."""
    nb, nf = 6, len(frames)
    d_out = C.c_void_p()
    assert hipl.hipMalloc(C.byref(d_out), nb * nf * H * W * 16) == 0
    host = np.empty((nb, nf, H, W, 4), np.float32)
    arr = hip.Context.frame_array(frames)
    checked = 0
    for world in (2, 3, 5, 8):
        uid = hip.comm_unique_id()
        ready = threading.Barrier(world)
        errors = []

        def rank_main(r, mode):
            ctx = comm = None
            try:
                ctx = rto.Context(0)
                ctx.upload_octree(nodes, g.min, g.voxel_size)
                ctx.upload_leaf_triangles(tris, off)
                comm = hip.Comm(ctx, world, r, uid, band_rows=16)
                ready.wait(60)
                for k in range(nb):
                    comm.submit(arr, d_out.value + k * nf * H * W * 16 if r == 0 else 0, H * W * 16, mode)
                comm.flush(60000)
            except BaseException as e:      # noqa: BLE001 -- reported by the main thread
                errors.append(f"rank {r}: {type(e).__name__}: {e}")
                try:
                    ready.abort()
                except Exception:
                    pass
            finally:
                if comm is not None:
                    comm.close()
                if ctx is not None:
                    ctx.close()

        for mode, wants, what in ((hip.RESIDENT_OCTREE, want_oct, "octree"), (hip.RESIDENT_TRIANGLES_SHADOW, want_tri, "triangles + shadow")):
            assert hipl.hipMemset(d_out, 0x55, nb * nf * H * W * 16) == 0
            ready.reset()
            ts = [threading.Thread(target=rank_main, args=(r, mode)) for r in range(world)]
            for t in ts:
                t.start()
            for t in ts:
                t.join(300)
            assert not errors, errors
            assert not any(t.is_alive() for t in ts), "a rank's thread did not come back"
            assert shim.rccl_shim_pending() == 0 and shim.rccl_shim_size_mismatches() == 0
            assert hipl.hipMemcpy(host.ctypes.data_as(C.c_void_p), d_out, nb * nf * H * W * 16, 2) == 0
            for k in range(nb):
                for i in range(nf):
                    bad = bits_differ(host[k, i], wants[i])
                    assert bad == 0, f"threads: world {world}, {what}, batch {k} frame {i}: {bad} pixels differ from the oracle's whole frame"
                    checked += 1
    hipl.hipFree(d_out)
    print(f"loopback transport, one thread per rank, pipelined: {checked} assembled frames equal the oracle's, worlds 2 3 5 8")


if __name__ == "__main__":
    main()
