/*
 * rto_hip.h -- C ABI of librto_hip.so: the MI355X (gfx950) replacement for the
 * device side of the reference's RayTracerBVH.
 *
 * The reference (abodthedude25/Ray_Tracing_Octrees, 453-skeleton/ = S/) has no
 * FFI: its "device boundary" is the OpenGL driver.  Each entry point below
 * replaces the GL calls named next to it; the C++ class in
 * ray_tracing_octrees_amd/host/RayTracerBVH.h keeps the reference's own
 * signatures (S/RayTracerBVH.h:28-80) on top of this ABI, and INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * Plain C types only: pointers, sizes, floats.  Matrices are column-major
 * float[16] exactly as glm::mat4 lays them out (&view[0][0],
 * S/RayTracerBVH.cpp:671).  All functions return RTO_OK (0) or a negative
 * RTO_E* code; rto_last_error() gives the message.  A context is bound to one
 * GPU; calls on one context must be serialised by the caller (the reference is
 * single-threaded, S/main.cpp), different contexts may be driven concurrently.
 */
#ifndef RTO_HIP_H
#define RTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTO_OK            0
#define RTO_E_INVALID    -1   /* bad argument                                  */
#define RTO_E_NO_OCTREE  -2   /* render before rto_upload_octree               */
#define RTO_E_HIP        -3   /* HIP runtime error (message has the hipError)  */
#define RTO_E_NO_DEVICE  -4   /* no usable gfx950 device / ordinal out of range */
#define RTO_E_UNSUPPORTED -5  /* e.g. packed kernel requested for a non-canonical array */
#define RTO_E_TIMEOUT    -6   /* rto_comm_flush_timeout: the collective did not complete; the communicator was aborted and is dead */

/* == struct GPUNodes, S/RayTracerBVH.h:21-26 / GLSL OctreeNodeGPUStruct S/RayTracerBVH.cpp:195-204 */
typedef struct rto_node {
    int32_t x, y, z, size;
    int32_t isLeaf, isSolid, isUniform;
    int32_t child[8];              /* -1 = none */
} rto_node;                        /* 60 bytes, stride 60 */

/* The uniforms of S/RayTracerBVH.cpp:652-680 that the shader actually reads
 * (numNodes and invVP are set by the reference but unused by the GLSL). */
typedef struct rto_frame {
    float   view[16];              /* camera.getView(), column-major           :668,:671 */
    float   cam_pos[3];            /* camera.getPos()                          :673-674 */
    float   aspect;                /*                                          :676     */
    float   fov_deg;               /* degrees                                  :677     */
    int32_t width, height;         /*                                          :679-680 */
} rto_frame;

/* Screen-space partition for multi-GPU rendering (no reference counterpart,
 * SURVEY.md section 8e): the image is cut into bands of band_rows rows; band b
 * belongs to part (b % num_parts).  A part's bands are stored back to back in
 * its compact buffer.  {1, 0, any} = the whole frame. */
typedef struct rto_partition {
    int32_t num_parts;
    int32_t part;
    int32_t band_rows;             /* multiple of 8 */
} rto_partition;

/* Which traversal kernel runs. AUTO = PACKED when the uploaded array is a
 * canonical BFS octree (what setOctree produces), else GENERIC. */
#define RTO_KERNEL_AUTO    0
#define RTO_KERNEL_GENERIC 1       /* 60-byte nodes, explicit child indices, per-thread stack[128] */
#define RTO_KERNEL_PACKED  2       /* 8-byte child descriptors, LDS level stack, branch-free O(1)-ascent loop with the pop
                                    * count rebuilt once per ray after the loop (k_trace_lean) */
#define RTO_KERNEL_PACKED_V1 3     /* first form of the packed kernel (level-by-level ascent); kept for A/B runs */
#define RTO_KERNEL_PACKED_PERSISTENT 4  /* the default kernel as persistent threads: a machine-filling grid whose waves
                                         * take launch slots from a global counter; kept for A/B runs */
#define RTO_KERNEL_PACKED_V3 5     /* round-1 default (k_trace_packed3: pops counted inside the loop); kept for A/B runs */

typedef struct rto_stats {         /* per-frame counters, same meaning as the oracle's */
    uint64_t rays, pops, hits, capped;
} rto_stats;

typedef struct rto_octree_info {
    int64_t num_nodes;             /* as uploaded                                         */
    int64_t num_internal;          /* nodes that push children                            */
    int32_t root_size;
    int32_t depth;                 /* log2(root_size)                                     */
    int32_t canonical;             /* 1 if the packed kernel can be used                  */
    int32_t culling_active;        /* 1 after rto_update_frustum(enable=1)                */
    int64_t visible_nodes;         /* count kept by the last frustum update (== num_nodes if none) */
} rto_octree_info;

typedef struct rto_context rto_context;

/* ---- lifetime -------------------------------------------------------------
 * replaces: GL context + RayTracerBVH::ensureComputeInitialized (S/RayTracerBVH.cpp:508-612). */
int  rto_create(int device_ordinal, rto_context** out);
void rto_destroy(rto_context* ctx);
const char* rto_last_error(const rto_context* ctx);     /* ctx may be NULL: error of the last failed rto_create */
int  rto_device_name(const rto_context* ctx, char* buf, size_t buflen);

/* ---- octree upload --------------------------------------------------------
 * replaces: glBufferData(GL_SHADER_STORAGE_BUFFER, numNodes*sizeof(GPUNodes), ...) in
 * RayTracerBVH::setOctree (S/RayTracerBVH.cpp:495-504).  The array is copied; the
 * library additionally repacks canonical arrays into child descriptors. */
int  rto_upload_octree(rto_context* ctx, const rto_node* nodes, int64_t num_nodes,
                       const float grid_min[3], float voxel_size);
/* N4 -- replaces createOctreeFromVoxelGrid + the BFS of setOctree + the upload (S/OctreeVoxel.cpp:704-778,
 * S/RayTracerBVH.cpp:443-504) in one call: the voxel grid (VoxelGrid.data layout: 1 byte per voxel, 0 EMPTY /
 * 1 FILLED, x fastest) is copied to the GPU and the flat GPUNodes array + child descriptors are built there
 * (occupancy pyramid, level-order emission).  The resident array is byte-identical to what the reference's two
 * functions produce; rto_download_nodes returns it. */
int  rto_build_octree(rto_context* ctx, const uint8_t* voxels, int dim_x, int dim_y, int dim_z,
                      const float grid_min[3], float voxel_size);
int  rto_download_nodes(rto_context* ctx, rto_node* out, int64_t capacity, int64_t* count);   /* out may be NULL */
/* Developer aid: rto_build_octree has two forms that produce the same arrays -- four launches for any depth (pyramid
 * levels kept in Morton order, every tree level ranked and emitted at once; grids up to 1024^3) and a level-by-level
 * form (~5 launches per level; any size).  level_by_level != 0 forces the second, 0 restores the automatic choice. */
int  rto_debug_set_build_path(rto_context* ctx, int level_by_level);
/* Device time of the last rto_build_octree: kernels (pyramid + emission) and the host-to-device voxel copy. */
int  rto_last_build_ms(const rto_context* ctx, float* kernels_ms, float* upload_ms);
int  rto_octree_info_get(const rto_context* ctx, rto_octree_info* out);
int  rto_set_kernel(rto_context* ctx, int kernel /* RTO_KERNEL_* */);
/* Launch order of the 8x8-pixel tiles in the packed kernel (a scheduling hint; pixels never depend on it).
 * Only the tiles of the root box's screen rectangle get a wave; the frame ends when its deepest rays end, so the
 * waves that will run longest should start first.
 * CENTRE_OUT: outwards from the projection of the solid geometry's centre.  TEMPORAL (default): by the per-tile
 * trip counts earlier frames of the same size recorded (CENTRE_OUT until one exists); the table is rebuilt by one
 * small kernel in front of a frame whenever the rectangle's tile box changed its size, and otherwise after refresh_period
 * frames (default 8; 0 keeps the current period) -- an interval that doubles, up to 8 x refresh_period, for as long as the
 * box keeps its size (a camera that stands still or pans). */
#define RTO_ORDER_CENTRE_OUT 0
#define RTO_ORDER_TEMPORAL   1
int  rto_set_launch_order(rto_context* ctx, int policy, int refresh_period);
/* The launch-order tables belong to the launch stream (hipStream_t as void*).  A context keeps them for the 16 most
 * recently used streams; call this before destroying a stream so that a later stream that happens to get the same
 * handle does not inherit its table (harmless to pixels, a stale schedule).  Synchronises the device. */
int  rto_forget_stream(rto_context* ctx, void* hip_stream);

/* ---- frustum culling ------------------------------------------------------
 * replaces: the CPU loop + compaction + SSBO re-upload of
 * renderSceneComputeWithCulling(updateFrustum=true) (S/RayTracerBVH.cpp:725-813).
 * The test runs on the GPU over the resident array; rendering afterwards behaves as
 * if the compacted array had been uploaded.  enable=0 restores the full array.
 * For canonical BFS octrees (what setOctree / rto_build_octree produce) the update is ONE kernel on rto_stream(ctx) and reads
 * nothing back: the visibility flags, the descriptors' visibility bits, the number of surviving nodes and the node traversals
 * start at (the root; the first visible node when the root itself was culled, S/RT:765-812) stay on the device, where the
 * traversal kernels read them; rto_octree_info_get / rto_download_visible_nodes fetch the count when asked (they wait for the
 * device).  The compacted array itself is made when somebody needs it (rto_download_visible_nodes, the generic kernel).
 * Ordering: frames on the context's own stream are ordered around the update by the stream; if frames were launched on other
 * streams since the last update (or ever captured on one), the device is waited for first.  Other arrays: per-node test,
 * one read-back (synchronous).  Before any of that the host asks whether the planes can cull a node at all (see
 * rto_debug_set_frustum_shortcut): when provably not, nothing is launched. */
int  rto_update_frustum(rto_context* ctx, const float view[16], float fov_deg, float aspect, int enable);
/* Developer aid: the same update with caller-supplied planes (LEFT, RIGHT, TOP, BOTTOM, NEAR, FAR as nx, ny, nz, d;
 * normalised) and margin instead of the ones S/RT:731-755 derives -- lets a test build situations real cameras cannot,
 * e.g. a culled root with surviving descendants (S/RT:765-812 then starts at whatever lands at compacted index 0). */
int  rto_debug_update_frustum_planes(rto_context* ctx, const float planes[24], float margin);
/* rto_update_frustum first asks, on the host, whether the planes can cull ANY node: every box lies in the root's, so a lower
 * bound of the positive-vertex value over all nodes is one expression per plane; when it exceeds the float evaluation's
 * possible error a thousandfold for all six planes -- with the reference's margin of 150 units: every camera near a scene a
 * few units across -- every node is visible, the device state that says so is written once and further such updates launch
 * NOTHING (identical results, S/RT:743-802's loop passes every node as well).  enabled = 0 forces the kernel (A/B, tests). */
int  rto_debug_set_frustum_shortcut(rto_context* ctx, int enabled);
/* *proven = 1 when the last frustum update was answered by that proof (no kernel), else 0. */
int  rto_debug_last_frustum_update_proven(const rto_context* ctx, int* proven);
/* Copies the compacted array (== m_visibleNodes, S/RayTracerBVH.cpp:775-802) to the host for parity
 * checks.  out may be NULL to query the count only. */
int  rto_download_visible_nodes(rto_context* ctx, rto_node* out, int64_t capacity, int64_t* count);

/* ---- render ---------------------------------------------------------------
 * replaces: glTexImage2D(RGBA32F) + uniforms + glDispatchCompute + glMemoryBarrier
 * (S/RayTracerBVH.cpp:630-688).  Output: RGBA32F, row-major, row 0 = top, alpha 1. */

/* Asynchronous: renders this part's bands into d_out (device pointer,
 * rto_partition_rows()*width*16 bytes) on hip_stream, a hipStream_t passed as void*.
 * As everywhere in HIP, NULL is the default (null) stream; rto_stream() returns the
 * context's own non-blocking stream for callers that want one. */
int  rto_render_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part /* NULL = whole frame */,
                       void* d_out, void* hip_stream);
/* HIP graphs: once a frame of the same width/height/aspect/fov has been rendered on a stream, rto_render_device (and the
 * _shade / _triangles variants) allocate nothing and never synchronise on it, so a sequence of frames may be captured
 * with hipStreamBeginCapture and replayed; a captured launch that WOULD have to allocate or synchronise (first frame of
 * a new size on that stream) fails with RTO_E_UNSUPPORTED instead of invalidating the capture (the runtime needs ~9 us between dependent plain launches but ~1 us between
 * graph nodes: 70 -> 61 us per frame at BASELINE config 2).  Captured launches use a launch-order table of their own (plain
 * launches never read it), and the first frame a capture records on a stream always carries a table-rebuild node, so a replay
 * never depends on what plain launches, other graphs or its own last frame left behind.  Replay a graph on the stream it was
 * captured on: the tables belong to that stream.  rto_update_frustum (canonical octrees, after one plain call) and
 * rto_render_resident can be captured on rto_stream(ctx) the same way: the update reads nothing back. */
/* Synchronous convenience: whole frame into host memory (the one API addition the
 * reference lacks: its texture is never read back). */
int  rto_render_host(rto_context* ctx, const rto_frame* frame, float* host_rgba);
/* The reference renders into a GL texture that stays on the GPU (S/RayTracerBVH.cpp:630-646, 684-688) and never reads
 * it back.  Same here: rto_render_resident renders the whole frame into the context's own device framebuffer
 * (asynchronous, on rto_stream(ctx)); mode RTO_RESIDENT_OCTREE = the octree path, RTO_RESIDENT_TRIANGLES /
 * RTO_RESIDENT_TRIANGLES_SHADOW = the leaf-triangle path of config 5.  rto_resident_frame returns the device pointer
 * (RGBA32F, width*height*16 bytes; valid until a render of another size or rto_destroy) for display interop or further
 * device work; rto_download_resident copies it to the host (synchronises). */
#define RTO_RESIDENT_OCTREE 0
#define RTO_RESIDENT_TRIANGLES 1
#define RTO_RESIDENT_TRIANGLES_SHADOW 2
int  rto_render_resident(rto_context* ctx, const rto_frame* frame, int mode);
int  rto_resident_frame(rto_context* ctx, void** d_rgba, int* width, int* height);
int  rto_download_resident(rto_context* ctx, float* host_rgba);
/* Number of rows part `part` owns. */
int  rto_partition_rows(const rto_frame* frame, const rto_partition* part);
/* Reassembles num_parts compact buffers laid end to end (as a gather delivers them;
 * every part padded to rto_partition_rows(part 0) rows) into a row-major frame. */
int  rto_assemble_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part,
                         const void* d_gathered, void* d_frame, void* hip_stream);
/* The same pair with a 4-byte payload per pixel, for the xGMI gather (no reference counterpart: the reference is
 * single-GPU).  Every pixel of S/RayTracerBVH.cpp:331-336 / :363 is a function of ONE float -- the Lambert term
 * max(dot(n, -L), 0) of the hit, or "no hit" -- so a part ships that float (-1.0f = no hit: rows*width*4 bytes
 * instead of *16) and the gathering GPU finishes `vec3(1,.8,.6) * term + .1` while it re-interleaves.  Same float
 * operations in the same order: the assembled frame is bit-identical to rto_render_device's. */
int  rto_render_shade_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part /* NULL = whole frame */,
                             void* d_shade, void* hip_stream);
int  rto_assemble_shade_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part,
                               const void* d_gathered_shade, void* d_frame, void* hip_stream);
/* Fewer, larger collectives: when every rank ships `batch` consecutive frames of its part in ONE gather
 * ([rank][batch][part-0 rows][width], either payload), this rebuilds frame `index` of the batch. */
int  rto_assemble_batch_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part, const void* d_gathered,
                               int batch, int index, int shade_payload, void* d_frame, void* hip_stream);
/* Several frames at once: this part (NULL: the whole frame) of frames[0..n-1] (any cameras, same width/height) into
 * d_out + i*frame_stride_bytes, either payload, asynchronously on hip_stream, up to 8 frames per kernel launch.  One frame's
 * kernel cannot be shorter than the ~36 us its deepest tile needs for its chain of dependent node visits, with most of the
 * GPU idle meanwhile; frames rendered together fill it (config 2: 46 -> 36 us per frame; a rank of an 8-GPU split: 36 -> 6 us
 * per part).  Pixels are those of n rto_render_device calls.  This is the throughput form: a caller that must show frame i
 * before it knows frame i+1's camera keeps using rto_render_device.
 * rto_assemble_batch_all_device: all `batch` frames of a gather into d_frames + i*frame_stride_bytes. */
int  rto_render_batch_device(rto_context* ctx, const rto_frame* frames, int n, const rto_partition* part, int shade_payload,
                             void* d_out, size_t frame_stride_bytes, void* hip_stream);
int  rto_assemble_batch_all_device(rto_context* ctx, const rto_frame* frames, int batch, const rto_partition* part,
                                   const void* d_gathered, int shade_payload, void* d_frames, size_t frame_stride_bytes, void* hip_stream);

/* ---- multi-GPU: screen split + ONE gather per batch, below the C boundary ---------------------------------
 * No reference counterpart (the reference is single-GPU).  A communicator binds one context
 * (= one GPU, holding the whole octree) to a rank of a world of GPUs of one node.  rto_comm_submit renders this rank's
 * bands (band b of band_rows rows belongs to rank b % world; from 4 ranks on rank 0 only gathers and assembles and band b
 * belongs to rank 1 + b % (world - 1)) of n consecutive frames as the 4-byte payload of
 * rto_render_shade_device, ships them to rank 0 with ONE grouped ncclSend/ncclRecv (RCCL over xGMI: a direct gather into
 * the root, never a ring) and, on rank 0, re-interleaves and finishes the colours into d_frames + i*frame_stride_bytes
 * (RGBA32F, bit-identical to rto_render_device).  Asynchronous and pipelined: the call returns when the work is
 * enqueued on the communicator's own two HIP streams, the gather of batch k overlaps the render of batch k+1 (two
 * sets of buffers alternate); d_frames must stay valid until rto_comm_flush (or a wait on rto_comm_stream) returns.
 * Every rank makes the same sequence of calls.  mode: RTO_RESIDENT_OCTREE / _TRIANGLES / _TRIANGLES_SHADOW.
 *
 * One process per GPU: rank 0 calls rto_comm_unique_id, hands the RTO_COMM_ID_BYTES bytes to the other ranks by any
 * means (MPI, a TCP store, a file), every rank calls rto_comm_create.  One process, several GPUs (what
 * RayTracerBVH::setDevices uses): rto_comm_create_all over n contexts on n different devices, then
 * rto_comm_submit_all issues every rank's part of a batch from the calling thread.  librccl is loaded on first use. */
typedef struct rto_comm rto_comm;
#define RTO_COMM_ID_BYTES 128
int  rto_comm_unique_id(void* id /* RTO_COMM_ID_BYTES */);
int  rto_comm_create(rto_context* ctx, int world, int rank, const void* id, int band_rows, rto_comm** out);
int  rto_comm_create_all(rto_context* const* ctxs, int n, int band_rows, rto_comm** out /* n handles */);
void rto_comm_destroy(rto_comm* comm);
const char* rto_comm_last_error(const rto_comm* comm);
int  rto_comm_submit(rto_comm* comm, const rto_frame* frames, int n, int mode, void* d_frames /* rank 0 */, size_t frame_stride_bytes);
int  rto_comm_submit_all(rto_comm* const* comms, int n_comms, const rto_frame* frames, int n, int mode, void* d_frames, size_t frame_stride_bytes);
/* single-process group: one frame through all ranks into rank 0's resident framebuffer (cf. rto_render_resident);
 * asynchronous, rto_download_resident / rto_resident_frame of rank 0's context give the assembled frame */
int  rto_comm_render_resident_all(rto_comm* const* comms, int n_comms, const rto_frame* frame, int mode);
int  rto_comm_flush(rto_comm* comm);              /* waits until every submitted batch is complete on this rank; reports an asynchronous
                                                     RCCL error of the communicator (ncclCommGetAsyncError) as RTO_E_HIP and marks it dead */
/* The same with a limit: polls the communicator's two streams (and its asynchronous error state) for at most timeout_ms.  On expiry
 * -- a peer died before its ncclSend, a link is down: without a limit rank 0 would sit in hipStreamSynchronize for ever -- or on an
 * asynchronous RCCL error the communicator is aborted (ncclCommAbort), marked DEAD and RTO_E_TIMEOUT / RTO_E_HIP is returned.  A dead
 * communicator refuses every further submit / flush (RTO_E_INVALID, "communicator is dead"); rto_comm_destroy still returns (it does
 * not wait for streams of a dead communicator's aborted collectives beyond their completion by the abort).  timeout_ms <= 0: no limit
 * (as rto_comm_flush). */
int  rto_comm_flush_timeout(rto_comm* comm, int timeout_ms);
/* 1 once the communicator has been aborted (timeout or asynchronous error), else 0.  The communicators of ONE
 * rto_comm_create_all group die together: every batch queues a Send / Recv on each of them, so an abort of one member aborts
 * (and marks dead) all of them, and rto_comm_submit_all checks every member before any of them starts the batch.
 * rto_comm_destroy never waits on a collective that cannot finish: it polls the streams (10 s) and aborts a live communicator
 * whose streams stay busy (a peer process that died) before it releases anything. */
int  rto_comm_is_dead(const rto_comm* comm);
/* ncclCommCount of the live communicator: the ranks RCCL itself says take part (what an N-GPU bench line reports as ranks_seen). */
int  rto_comm_ranks_seen(const rto_comm* comm, int* ranks);
/* Test hook: marks the communicator as timed out exactly as rto_comm_flush_timeout does on expiry (ncclCommAbort, dead). */
int  rto_comm_debug_abort(rto_comm* comm);
/* Developer aid: a ONE-rank communicator renders, ships and assembles as rank as_rank of as_world GPUs (every per-rank cost
 * of an N-GPU split except the other GPUs' traffic); the assembled frames hold that rank's bands only.  as_world = 0: off. */
int  rto_comm_debug_rehearse(rto_comm* comm, int as_world, int as_rank);
/* Developer aid: floats this rank shipped for the batch submitted last -- only the columns of the geometry's screen rectangle
 * travel, rank 0 paints the background itself -- and what whole rows would have been. */
int  rto_comm_debug_set_rehearsal_clear(rto_comm* comm, int enabled);   /* rehearsals: per-batch clear of the absent ranks' rows (default on; timing runs switch it off) */
int  rto_comm_debug_last_payload(const rto_comm* comm, int64_t* packed_floats, int64_t* full_floats);
/* Bench aid: with timing on, every batch records four timed events; after its flush rto_comm_debug_last_timing gives the GPU
 * milliseconds of the batch submitted last: ms[0] this rank's render (+ pack), ms[1] its grouped send / recv (+ rank 0's assembly). */
int  rto_comm_debug_set_timing(rto_comm* comm, int enabled);
int  rto_comm_debug_last_timing(rto_comm* comm, float ms[2]);
void* rto_comm_stream(rto_comm* comm);            /* hipStream_t the gathers and (rank 0) the assembled frames are ordered on */

/* ---- multi-GPU: the split plan (pure host arithmetic, no GPU needed) -----------------------------------------
 * Everything the ranks of a screen split must agree on BEFORE they post sends and receives -- who renders, which part a
 * rank owns, the rows every part's buffer is padded to, the column window of each frame of the batch that travels (only
 * the columns of the geometry's screen rectangle do), the offsets inside a packed part and the float count per rank --
 * is derived by ONE function from the frames and the scene's bounds.  rto_comm_submit calls it on every rank; tests call
 * the same function from N processes and compare the plans byte for byte (tests/_tilesplit_worker.py), so a mismatch
 * cannot first appear as a hang inside RCCL.  No reference counterpart (the reference is single-GPU). */
typedef struct rto_scene_bounds {
    float   grid_min[3];
    float   voxel_size;
    int32_t root_size;             /* octree root edge in voxels (power of two)                      */
    int32_t solid_lo[3], solid_hi[3];  /* bounding box of the solid leaves, voxel units; lo > hi: nothing solid */
} rto_scene_bounds;
int  rto_scene_bounds_get(const rto_context* ctx, rto_scene_bounds* out);
/* the same from a GPUNodes array on the host (what rto_upload_octree derives) */
int  rto_scene_bounds_of_nodes(const rto_node* nodes, int64_t num_nodes, const float grid_min[3], float voxel_size, rto_scene_bounds* out);

#define RTO_SPLIT_MAX_FRAMES 32    /* frames of one batch whose windows the plan can hold; larger batches ship whole rows */
typedef struct rto_split_plan {
    int32_t world, band_rows, width, height, n_frames;
    int32_t render_parts;          /* parts a frame is cut into: world, or world - 1 when rank 0 only gathers (world >= 4) */
    int32_t first_render_rank;     /* 0, or 1 when rank 0 only gathers and assembles                 */
    int32_t rows_part0;            /* rows of part 0 = rows every part's buffer is padded to          */
    int32_t cropped;               /* 1: only the window columns of each frame travel                 */
    int32_t win_x0[RTO_SPLIT_MAX_FRAMES], win_w[RTO_SPLIT_MAX_FRAMES];   /* window of frame i: columns [x0, x0 + w) */
    int64_t win_off[RTO_SPLIT_MAX_FRAMES];   /* float offset of frame i inside a rank's packed part (rows_part0 x win_w each) */
    int64_t frame_floats;          /* rows_part0 * width: one frame of a part, unpacked               */
    int64_t full_floats;           /* frame_floats * n_frames                                         */
    int64_t pack_floats;           /* floats every rendering rank ships == stride between the parts on rank 0 */
} rto_split_plan;
int  rto_split_plan_make(const rto_scene_bounds* scene, const rto_frame* frames, int n, int world, int band_rows, rto_split_plan* out);
int  rto_split_part_of_rank(const rto_split_plan* plan, int rank);      /* part that rank renders, -1: it renders nothing */
int  rto_split_rows_of_part(const rto_split_plan* plan, int part);      /* rows part owns (<= rows_part0)                */
/* Where row `row` of an assembled frame comes from: *part and the row inside that part's compact buffer. */
int  rto_split_row_source(const rto_split_plan* plan, int row, int* part, int* local_row);

/* ---- N2: leaf triangles + shadow ray (BASELINE config 5) -------------------------
 * No upstream counterpart: the reference has no ray/triangle code (its MC triangles are rasterised).  The
 * triangles are those MarchingCubesRenderer emits per leaf (localMC, S/OctreeVoxel.cpp:780-879;
 * host/OctreeVoxel.h buildLeafTriangles): 12 floats each (v0, v1, v2, face normal), ordered by node;
 * tri_offset[i]..tri_offset[i+1] (numNodes+1 entries) is node i's range.  Rendering follows the reference's
 * traversal order and 512-pop cap; a leaf is hit when one of its triangles is (Moeller-Trumbore, nearest
 * t > 0 in the leaf); shading = the reference's Lambert term on the ray-facing face normal; shadow != 0 adds one
 * ray towards the light (any hit => ambient only).  stats may be NULL (pops counts both traversals). */
int  rto_upload_leaf_triangles(rto_context* ctx, const float* tris, int64_t num_tris, const int32_t* tri_offset);
/* The same buffer built in HBM (replaces the CPU loop MarchingCubesRenderer::render -> localMC per leaf,
 * S/Renderer.cpp:14-36, S/OctreeVoxel.cpp:780-879): identical triangles in identical order, for the resident octree.
 * voxels: dimX*dimY*dimZ bytes (x fastest, 0 EMPTY / 1 FILLED) of the grid the octree was made from; NULL (dims
 * ignored) reuses the voxels rto_build_octree kept in HBM.  rto_last_build_ms() then reports this build.
 * rto_download_leaf_triangles copies the result out for parity checks (tris / tri_offset may be NULL). */
int  rto_build_leaf_triangles(rto_context* ctx, const uint8_t* voxels, int dimX, int dimY, int dimZ);
int  rto_download_leaf_triangles(rto_context* ctx, float* tris, int64_t tri_capacity, int32_t* tri_offset /* numNodes+1 */,
                                 int64_t* num_tris);
int  rto_render_triangles_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part, int shadow,
                                 void* d_out, void* hip_stream);
/* Several frames per kernel launch, as rto_render_batch_device: this part (NULL: whole frames) of frames[0..n-1] into
 * d_out + i*frame_stride_bytes, RGBA32F or (shade_payload) the 4-byte Lambert term.  A part's kernel lasts as long as its
 * deepest tile (0.24 ms at config 5 whatever its share of the pixels); parts launched together fill the GPU. */
int  rto_render_triangles_batch_device(rto_context* ctx, const rto_frame* frames, int n, const rto_partition* part, int shadow, int shade_payload,
                                       void* d_out, size_t frame_stride_bytes, void* hip_stream);
int  rto_render_triangles_host(rto_context* ctx, const rto_frame* frame, int shadow, float* host_rgba, rto_stats* stats);
/* 4-byte payload variant (see rto_render_shade_device); reassemble with rto_assemble_shade_device. */
int  rto_render_triangles_shade_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part, int shadow,
                                       void* d_shade, void* hip_stream);

/* ---- N1: octreeRaySkip --------------------------------------------------------
 * replaces: the CPU recursion octreeRaySkip(root, ro, rd, tMin, tMax, grid, &visibility)
 * (S/VolumeRaycastRenderer.cpp:50-155; called for a 7x7 probe grid per frame at :1602-1647).
 * n rays share the origin ro; rd holds n directions (x,y,z).  out_t[i] = entry distance of the first solid
 * leaf in the reference's child order, or 1e30.  use_visibility != 0 applies the flags of the last
 * rto_update_frustum the way the reference applies its visibility map (:64-67).  Synchronous, host buffers. */
int  rto_octree_ray_skip(rto_context* ctx, const float ro[3], const float* rd, int64_t n, float t_min, float t_max,
                         int use_visibility, float* out_t);

/* The reference's CLOSEST-hit traversal -- the earlier compute shader it keeps block-commented in the same file
 * (453-skeleton/RayTracerBVH.cpp:46-166; traversal :63-138): the rays, slab test, LIFO order and shading of rto_render_device, but no
 * break on the first accepted leaf and no 512-pop cap: every node whose tNear lies below the best hit so far is visited, the solid
 * leaf with the smallest tHit = max(0, tNear) wins, a later leaf replacing it only when strictly nearer.  Dead code upstream
 * (replaced by the "workload limiting" shader rto_render_device implements), the only traversal rule of the reference this library
 * could not render until round 4.  Pixels are bit-identical to that shader's text compiled under the reference's glm
 * (oracle/glsl_driver.cpp, tests/golden/glsl_images_small.npz).  RGBA32F, row 0 = top, asynchronous on hip_stream; part as in
 * rto_render_device; a frustum update in force is honoured the way the reference's culled render is (compacted array).  Three
 * kernels, one frame: canonical trees walk the descriptor tree near children first (k_closest_near_first: the winner of the
 * exhaustive walk is the leaf of least tHit among those whose ancestors all pass their slab tests, ties to the leaf popped
 * first, whatever the order of the walk); the counters come from the same pops made in the reference's order
 * (k_closest_lean); arbitrary arrays, culled frames and RTO_KERNEL_GENERIC go node by node over the 60-byte array
 * (k_trace_closest).  _host: synchronous; stats (may be NULL): rays, popped nodes (uncapped), hit pixels. */
int  rto_render_closest_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part /* NULL = whole frame */, void* d_rgba, void* hip_stream);
int  rto_render_closest_host(rto_context* ctx, const rto_frame* frame, float* host_rgba, rto_stats* stats /* may be NULL */);

/* The same search as a RENDER MODE (octreeRaySkip "as a second kernel mode = nearest hit"): for every
 * pixel, the ray of generateRay (S/RayTracerBVH.cpp:338-355, origin = cam_pos) goes through
 * octreeRaySkip(root, ro, rd, 0, 1e30, grid, visibility) (S/VolumeRaycastRenderer.cpp:50-155).  d_dist (width*height floats,
 * row 0 = top, may be NULL): the distance it returns, 1e30 = nothing -- bit-identical to the reference's compiled function on
 * those rays (tests/golden/ref_ray_skip.npz "pixels").  d_rgba (RGBA32F, may be NULL): the reference's shade
 * (S/RayTracerBVH.cpp:283-285, 331-336: box-centre pseudo-normal, Lambert + 0.1) of the leaf that distance belongs to, at
 * tHit = that distance; background (0,0,0,1).  Asynchronous on hip_stream, outputs stay on the device; part as in
 * rto_render_device.  Canonical BFS octrees only.  _host: synchronous convenience (either pointer may be NULL). */
int  rto_render_skip_device(rto_context* ctx, const rto_frame* frame, const rto_partition* part /* NULL = whole frame */, int use_visibility,
                            void* d_rgba, void* d_dist, void* hip_stream);
int  rto_render_skip_host(rto_context* ctx, const rto_frame* frame, int use_visibility, float* host_rgba, float* host_dist);
/* octreeRaySkip's CONSUMER in drawRaycast (S/VolumeRaycastRenderer.cpp:1602-1663) in one launch, nothing copied: the 7x7 probe
 * directions through inverse(perspective(45 deg, aspect, 0.1, 5000)) and inverse(view), the 49 traversals, the value std::sort
 * would leave at index int(n * 0.15f) among the valid distances (0 < t < 1e30) x 0.75, and the temporal blend
 * *d_skip = *d_skip * 0.4f + that * 0.6f (`static float lastSkipDistance` of :1658 = the float the caller keeps on the device:
 * set it to 0 before the first frame).  Asynchronous on hip_stream.  _host: the same with a host float (synchronous); probe_t
 * (may be NULL) receives the 49 distances. */
int  rto_probe_skip_device(rto_context* ctx, const float view[16], const float cam_pos[3], float aspect, int use_visibility,
                           void* d_skip /* one float on the device: in = previous value, out = new */, void* hip_stream);
int  rto_probe_skip_host(rto_context* ctx, const float view[16], const float cam_pos[3], float aspect, int use_visibility,
                         float* io_skip, float* probe_t /* 49 floats or NULL */);

/* ---- instrumentation ------------------------------------------------------*/
/* Renders the frame once with counting enabled (synchronous). */
int  rto_frame_stats(rto_context* ctx, const rto_frame* frame, rto_stats* out);
/* Per-pixel traversalSteps: +steps for a hit, -steps for a miss (width*height int32, host). */
int  rto_render_steps_host(rto_context* ctx, const rto_frame* frame, int32_t* host_steps);
/* Developer aid: per-wave timeline of one frame of the packed kernel.  8 int32 per 8x8 tile (row-major
 * tiles): start lo/hi, end lo/hi (100 MHz wall clock), loop iterations, HW_ID, XCC_ID, lanes that entered
 * the tree | launch slot << 8; all zero for tiles outside the root rectangle's tile box, which get no wave (their pixels are written
 * by the waves of the box as wide stores).  host_records may be NULL to query *num_tiles. */
int  rto_debug_timeline(rto_context* ctx, const rto_frame* frame, int32_t* host_records, int64_t capacity_tiles,
                        int64_t* num_tiles);
/* Developer aids for the launch-order study: per-tile trip counts of the last frame, and a caller-supplied
 * slot -> tile table (NULL restores the automatic one). */
int  rto_debug_tile_cost(rto_context* ctx, int32_t* host_cost, int64_t capacity, int64_t* count);
int  rto_debug_set_tile_order(rto_context* ctx, const int32_t* host_order, int64_t n);
/* The occupancy mask: the first few workgroups of every colour / shade frame of the default kernels
 * project the octree's coarse cells (its internal nodes at one depth + the solid leaves above it) onto the screen and stamp
 * the 8x8 tiles they can touch; once the mask is complete, waves of unstamped tiles write black without setting up a single
 * ray (about two thirds of the rays inside the geometry's screen rectangle miss everything).  Nobody waits for the mask: the
 * previous frame's costliest tiles never look, later waves look only if it is complete -- tiles without work sort to the end
 * of the launch order, where it always is.  A scheduling device like the launch order: pixels never depend on it.
 * rto_debug_set_tile_mask: 0 = off, 1 = on (default), 2 = on and built by a launch of its own in front of the frame, every
 * wave consulting it (tests: the mask decides for every tile).  _info: the depth the cells are taken from and their number
 * (0: no mask for this octree). */
int  rto_debug_set_tile_mask(rto_context* ctx, int enabled);
int  rto_debug_tile_mask_info(const rto_context* ctx, int* level, int* num_cells);
/* Test hook: from now on the k-th (1-based) allocation of the frustum-update / rto_comm buffers of this PROCESS fails (k <= 0: never).
 * An explicit call -- the library reads no environment variable (A/B knobs exist only in -DRTO_DEV_KNOBS builds). */
int  rto_debug_fault_alloc(long k);
/* Writes the launch-order sort refused because they fell outside the table (must be 0; synchronises). */
int  rto_debug_sort_violations(rto_context* ctx, int* count);
/* Device time in ms of the most recent traversal kernel launched by this context
 * (hipEvent pair on the launch stream; synchronises on that event). */
int  rto_last_kernel_ms(rto_context* ctx, float* ms);
/* Per-launch timing without synchronising inside a timed loop: after rto_timing_begin(ctx, n) the next n
 * traversal-kernel launches are bracketed by their own hipEvent pair on the launch stream (just that kernel: the
 * launch-order kernel that may follow is outside the pair); rto_timing_read synchronises the device and returns the
 * durations in ms.  rto_timing_begin(ctx, 0) switches it off (rto_last_kernel_ms keeps working: one event pair per launch).
 * rto_timing_begin(ctx, -1) records no events at all until the next rto_timing_begin(ctx, >= 0): an event costs ~2 us
 * between two dependent launches, which matters when frames of ~50 us are issued back to back without a HIP graph. */
int  rto_timing_begin(rto_context* ctx, int capacity);
int  rto_timing_read(rto_context* ctx, float* ms, int capacity, int* count);   /* ms may be NULL to query count */
/* The context's own hipStream_t (used by the synchronous entry points). */
void* rto_stream(rto_context* ctx);
/* Waits for all work on the context's device (hipDeviceSynchronize). */
int  rto_synchronize(rto_context* ctx);

#ifdef __cplusplus
}
#endif
#endif
