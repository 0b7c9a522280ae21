// rto_device.hip.h -- gfx950 device code of librto_hip.so.
//
// Replaces the GLSL compute shader of the reference's RayTracerBVH
// (453-skeleton/RayTracerBVH.cpp:182-369, "S/RT" below).  Two traversal kernels
// produce bit-identical pixels:
//
//   k_trace_generic  walks the uploaded 60-byte GPUNodes array with explicit child
//                    indices and a private stack[128] -- a direct statement of
//                    S/RT:239-327, used for arbitrary (e.g. user-compacted) arrays.
//
//   k_trace_packed   the MI355X path.  For canonical BFS octrees (what
//                    setOctree produces, S/RT:443-490) the tree is re-encoded as one
//                    8-byte child descriptor per INTERNAL node:
//                        .x = solidMask | internalMask<<8 | visibleMask<<16
//                        .y = descriptor index of the first internal child
//                    Leaves need no storage (a 256^3 scene shrinks from 22.5 MB to
//                    375 KB and lives in every XCD's L2).  One loop iteration = one
//                    internal node: the 8 child slab tests share 12 per-axis
//                    half-plane terms, empty leaves are never tested (they only
//                    advance the step counter, which is all S/RT:257-274 does for
//                    them), and the per-ray stack is one 8-byte entry per tree LEVEL
//                    in LDS ([level][lane], bank-conflict free) instead of
//                    stack[128] per thread.
//
// Exactness contract (DESIGN.md "Numerics"): every float expression keeps the
// operation order of the GLSL/glm source, the TU is compiled with
// -ffp-contract=off and HIP's default correctly-rounded fp32 divide/sqrt, so the
// pixels equal the CPU oracle's bit for bit.  min/max follow glm's
// (y<x)?y:x convention; rays whose reciprocal direction is not finite (the only
// way a NaN can enter the slab test) take an exact compare/select path, all other
// rays use v_min_f32/v_max3_f32, which agree with that convention on non-NaN data
// up to the sign of zero, which no later operation can observe.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rto_hip.h"

namespace rto {

constexpr int kMaxTraversalSteps = 512;   // S/RT:192
constexpr int kWave = 64;
#ifndef RTO_BLOCK
#define RTO_BLOCK 256             // A/B builds (-DRTO_BLOCK=n), traversal kernel at config 2: 64 -> 67.4 us, 128 -> 69.0, 256 -> 66.3, 512 -> 71.3
#endif
constexpr int kBlock = RTO_BLOCK;         // 4 waves, each owns one 8x8 pixel tile
#ifndef RTO_PACKED3_WAVES
#define RTO_PACKED3_WAVES 6       // waves per SIMD for the default traversal kernel (8 forces spills and measured 6 % slower)
#endif
#ifndef RTO_PERSIST_CHUNK
#define RTO_PERSIST_CHUNK 8       // launch slots a persistent wave takes per atomic (A/B builds: -DRTO_PERSIST_CHUNK=n)
#endif
constexpr int kMaxDepth = 20;             // log2(root size) supported by the packed kernel

// Where a traversal starts while a frustum update is active -- kept ON THE DEVICE (written by k_cull_desc, read by the traversal
// kernels), so that rto_update_frustum needs no read-back and can be stream-captured.  Normally the root.  If the update
// culled the root but not all of its descendants, the reference's compacted array begins with the first visible node in BFS
// order and its traversal starts THERE (S/RT:765-812): that node is recorded here.
struct StartState {
    int visible;            // 0: no node survived the update -> nothing is traversed, the frame is black
    unsigned desc;          // descriptor index of the start node (when it is internal)
    int x, y, z;            // its position, voxel units
    int shift;              // log2 of its edge
    int leaf, solid;        // a terminal start node: the traversal is one pop
    int rootVisible;        // the root's own flag
    int firstVisible;       // index of the start node in the uploaded array (0x7fffffff: none)
    long long visibleCount; // nodes the update kept (== m_visibleNodes.size(), S/RT:775)
    unsigned ticket;        // k_cull_desc: blocks finished (returns to 0 at the end of every launch)
};

struct RenderParams {
    float invView[16];     // glm::inverse(view), hoisted from S/RT:348 (pixel independent)
    float camPos[3];
    float aspect;
    float tanHalfFov;      // tan(radians(fov) * 0.5), hoisted from S/RT:340-344
    float gridMin[3];
    float voxelSize;
    float lightNeg[3];     // -normalize(vec3(-1)), S/RT:333-334
    float lightInv[3];     // 1.0f / lightNeg (IEEE division on the host: the same float the device's correctly rounded division gives): every shadow ray's reciprocal direction
    int W, H;
    int rootSize, depth;
    int exactGrid;                  // host-proven (grid_is_exact): every node plane is computed without rounding -> child_axis_terms_exact
    int numParts, part, bandRows;   // rto_partition
    int localRows;                  // rows this part owns
    int tilesX, tilesY;             // 8x8 tiles over W x localRows
    int rootVisible;                // 0 => frustum update culled the root: black frame
    int orderCx, orderCy;           // tile nearest the projected scene centre: tiles launch centre-out (heavy first)
    int rootX0, rootY0, rootX1, rootY1;   // pixel rectangle (inclusive, GLOBAL rows) outside of which no ray can meet the root box
    int solidX0, solidY0, solidX1, solidY1;   // the same for the bounding box of the solid leaves (host side: replaces root* for colour / shade frames)
    const int* tileOrder;           // launch slot -> tile as x | y << 16 relative to the box below: its tiles, costliest tiles of an EARLIER
                                    // frame first (null: centre-out over the box)
    int* tileCost;                  // tile -> loop trip count of its wave in THIS frame (null: not recorded)
    // Launch geometry of the packed kernels.  Only tiles of the box [boxX0, boxX0+boxW) x [boxY0, boxY0+boxH) -- the tile
    // bounding box of the root rectangle, rounded outwards -- get a wave (traceWaves = boxW*boxH).  With skipOutside
    // the pixels OUTSIDE the root rectangle (black for certain, one pop) are not stored by their tile's wave: the
    // outside region is cut into 64-pixel chunks of contiguous memory and wave `slot` stores the chunks
    // slot, slot + launchWaves, ... after its own tile (1 KB per store instruction instead of 8 x 128 B, and no wave
    // at all for the ~80 % of the tiles of a typical frame that lie wholly outside).
    int boxX0, boxY0, boxW, boxH;
    int traceWaves, launchWaves;
    int skipOutside;
    int fillChunks;                 // chunks of the outside region: top band, bottom band, then per middle row left + right strip
    int fillTopRows, fillBotRow0;   // local rows [0, fillTopRows) and [fillBotRow0, localRows) lie wholly outside
    int fillLeftW, fillRightX0;     // in the rows between: pixels [0, fillLeftW) and [fillRightX0, W)
    int fillTopChunks, fillBotChunks, fillLeftPer, fillRightPer;
    const float* rayX;              // [W]  ((px+.5)/W*2-1)*aspect*tanHalfFov, the separable part of S/RT:341-346 (host-computed)
    const float* rayY;              // [H]  (1-(py+.5)/H*2)*tanHalfFov
    const StartState* start;        // non-null while a frustum update is active on a canonical tree: replaces rootVisible / the root as start node (lean kernels)
    // Screen-space occupancy mask (mask_block, run by the first workgroups of a colour / shade launch of the lean kernels): one word per
    // 8x8 tile in GLOBAL image rows ([strip = row / 8][tx]); a tile whose word differs from maskStamp -- and with the "whole frame"
    // word at maskAllIndex also different -- cannot contain a ray that meets a solid leaf: its wave stores black and does nothing else.
    unsigned* tileMask;             // null: no mask (instrumented frames, A/B kernels, non-canonical arrays)
    unsigned maskStamp;
    int maskAllIndex;               // words of the tile array; behind it: the "whole frame" word, the "mask complete" word, the ticket
    int maskBlocks;                 // workgroups at the FRONT of the grid that build the mask instead of tracing (0: none)
    int maskTrustSlots;             // launch slots below this never consult the mask (kMaskTrustSlots; 0 in the tests' forced mode)
    const int4* maskCells;          // the coarse cells (x, y, z, edge)
    int maskNumCells;
    int maskLdsBytes;               // dynamic LDS of the launch: a mask workgroup gathers its stamps there (mask_block)
    float maskInvAspTan, maskInvTanH;   // 1 / (aspect * tan(fov/2)), 1 / tan(fov/2)
    float viewRows[12];             // rows 0..2 of the view matrix
};
constexpr int kRimTiles = 2;            // see the cost record of trace_tile_lean
constexpr int kMaskTrustSlots = 1024;   // the first launch slots (the costliest tiles of the previous frame) never consult the mask

// ---------------------------------------------------------------- scalar helpers
// glm/detail/func_common.inl: min(x,y) = (y<x)?y:x ; max(x,y) = (x<y)?y:x
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }

template <bool EXACT> struct MinMax;
template <> struct MinMax<true> {
    static __device__ __forceinline__ float mn(float x, float y) { return gmin(x, y); }
    static __device__ __forceinline__ float mx(float x, float y) { return gmax(x, y); }
};
template <> struct MinMax<false> {
    static __device__ __forceinline__ float mn(float x, float y) { return __builtin_fminf(x, y); }
    static __device__ __forceinline__ float mx(float x, float y) { return __builtin_fmaxf(x, y); }
};

// sqrtf -> llvm.sqrt.f32 = v_sqrt_f32 + 2 fma fix-ups (correctly rounded); __fsqrt_rn is the bare 1-ulp v_sqrt_f32.
__device__ __forceinline__ float inversesqrt(float x) { return 1.0f / __builtin_sqrtf(x); }

__device__ __forceinline__ int global_row(const RenderParams& P, int ly) {
    if (P.numParts == 1) return ly;
    int band = ly / P.bandRows;
    int r = ly - band * P.bandRows;
    return (band * P.numParts + P.part) * P.bandRows + r;
}

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;   // 1.0 / d  (S/RT:228, hoisted: it never changes along a ray)
};

// S/RT:338-355 generateRay
__device__ __forceinline__ Ray generate_ray(const RenderParams& P, int px, int py) {
    float nx = ((float)px + 0.5f) / (float)P.W * 2.0f - 1.0f;
    float ny = 1.0f - ((float)py + 0.5f) / (float)P.H * 2.0f;
    nx *= P.aspect;
    nx *= P.tanHalfFov;
    ny *= P.tanHalfFov;
    // normalize(vec4(nx, ny, -1, 0)): glm vec4 dot = (x*x + y*y) + (z*z + w*w)
    float d4 = (nx * nx + ny * ny) + ((-1.0f) * (-1.0f) + 0.0f * 0.0f);
    float inv4 = inversesqrt(d4);
    float vx = nx * inv4, vy = ny * inv4, vz = (-1.0f) * inv4, vw = 0.0f * inv4;
    // invView * v: glm mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
    const float* m = P.invView;
    float wx = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * vw);
    float wy = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * vw);
    float wz = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * vw);
    // normalize(vec3): dot = x*x + y*y + z*z
    float d3 = wx * wx + wy * wy + wz * wz;
    float inv3 = inversesqrt(d3);
    Ray r;
    r.ox = P.camPos[0]; r.oy = P.camPos[1]; r.oz = P.camPos[2];
    r.dx = wx * inv3; r.dy = wy * inv3; r.dz = wz * inv3;
    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    return r;
}

// Same, with the separable screen terms taken from the host tables (identical float operations, done once
// per column / row instead of once per pixel).
__device__ __forceinline__ Ray generate_ray_tab(const RenderParams& P, int px, int py) {
    const float nx = P.rayX[px], ny = P.rayY[py];
    float d4 = (nx * nx + ny * ny) + ((-1.0f) * (-1.0f) + 0.0f * 0.0f);
    float inv4 = inversesqrt(d4);
    float vx = nx * inv4, vy = ny * inv4, vz = (-1.0f) * inv4, vw = 0.0f * inv4;
    const float* m = P.invView;
    float wx = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * vw);
    float wy = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * vw);
    float wz = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * vw);
    float d3 = wx * wx + wy * wy + wz * wz;
    float inv3 = inversesqrt(d3);
    Ray r;
    r.ox = P.camPos[0]; r.oy = P.camPos[1]; r.oz = P.camPos[2];
    r.dx = wx * inv3; r.dy = wy * inv3; r.dz = wz * inv3;
    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    return r;
}

// The grid's origin and voxel size as four scalars.  Read straight from the kernel-argument struct, clang's vectoriser
// merges the loads of gridMin[2] and voxelSize into <2 x float> pieces that overlap the gridMin array, SROA then cannot
// split the by-value copy of RenderParams, and the four floats live in SCRATCH (a store at wave start, reloads at every
// use: 13,800 waves x 64 lanes x 16 B of memory traffic per frame).  The empty asm keeps them four separate SGPR values.
struct Geo { float gx, gy, gz, vs; };
__device__ __forceinline__ Geo geo_of(const RenderParams& P) {
    Geo g = { P.gridMin[0], P.gridMin[1], P.gridMin[2], P.voxelSize };
    asm volatile("" : "+s"(g.gx), "+s"(g.gy), "+s"(g.gz), "+s"(g.vs));
    return g;
}

// S/RT:226-236 intersectAABB + S/RT:265-266 node box, for one node given by integer coords.
__device__ __forceinline__ bool slab_exact(const Geo& G, const Ray& r, int x, int y, int z, int size,
                                           float& tNear, float& tFar,
                                           float& mnx, float& mny, float& mnz, float& mxx, float& mxy, float& mxz) {
    float vs = G.vs;
    mnx = G.gx + (float)x * vs;
    mny = G.gy + (float)y * vs;
    mnz = G.gz + (float)z * vs;
    float ext = (float)size * vs;
    mxx = mnx + ext; mxy = mny + ext; mxz = mnz + ext;
    float t1x = (mnx - r.ox) * r.ix, t1y = (mny - r.oy) * r.iy, t1z = (mnz - r.oz) * r.iz;
    float t2x = (mxx - r.ox) * r.ix, t2y = (mxy - r.oy) * r.iy, t2z = (mxz - r.oz) * r.iz;
    float tminx = gmin(t1x, t2x), tminy = gmin(t1y, t2y), tminz = gmin(t1z, t2z);
    float tmaxx = gmax(t1x, t2x), tmaxy = gmax(t1y, t2y), tmaxz = gmax(t1z, t2z);
    tNear = gmax(gmax(tminx, tminy), tminz);
    tFar = gmin(gmin(tmaxx, tmaxy), tmaxz);
    return (tNear <= tFar && tFar > 0.0f);
}
__device__ __forceinline__ bool slab_exact(const RenderParams& P, const Ray& r, int x, int y, int z, int size,
                                           float& tNear, float& tFar,
                                           float& mnx, float& mny, float& mnz, float& mxx, float& mxy, float& mxz) {
    return slab_exact(geo_of(P), r, x, y, z, size, tNear, tFar, mnx, mny, mnz, mxx, mxy, mxz);
}

// Hit epilogue: S/RT:279-285 (tHit, centre pseudo-normal) + S/RT:331-336 (Lambert).
// Split in two so that the multi-GPU path can ship the 4-byte Lambert term instead of the 16-byte pixel:
// shade_term() is everything up to max(dot(n, -L), 0); shade_color() is the final colour expression, evaluated
// either in the traversal kernel (RGBA32F output) or in k_assemble_shade on the gathering GPU.  Same float
// operations in the same order either way, so the pixel bits do not depend on where the second half runs.
__device__ __forceinline__ float shade_term(const RenderParams& P, const Geo& G, const Ray& r, int x, int y, int z, int size) {
    float tNear, tFar, mnx, mny, mnz, mxx, mxy, mxz;
    slab_exact(G, r, x, y, z, size, tNear, tFar, mnx, mny, mnz, mxx, mxy, mxz);
    float tHit = gmax(0.0f, tNear);
    float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy), cz = 0.5f * (mnz + mxz);
    float px = r.ox + r.dx * tHit, py = r.oy + r.dy * tHit, pz = r.oz + r.dz * tHit;
    float qx = px - cx, qy = py - cy, qz = pz - cz;
    float inv = inversesqrt(qx * qx + qy * qy + qz * qz);
    float nx = qx * inv, ny = qy * inv, nz = qz * inv;
    return gmax(0.0f, nx * P.lightNeg[0] + ny * P.lightNeg[1] + nz * P.lightNeg[2]);   // >= +0, never NaN (glm max keeps 0)
}

constexpr float kShadeMiss = -1.0f;   // shade-buffer code of a ray without a hit (the Lambert term is never negative)

__device__ __forceinline__ float4 shade_color(float ndotl) {
    if (ndotl < 0.0f) return make_float4(0.f, 0.f, 0.f, 1.f);                             // S/RT:363 background
    return make_float4(1.0f * ndotl + 0.1f, 0.8f * ndotl + 0.1f, 0.6f * ndotl + 0.1f, 1.0f);
}

__device__ __forceinline__ float shade_term(const RenderParams& P, const Ray& r, int x, int y, int z, int size) {
    return shade_term(P, geo_of(P), r, x, y, z, size);
}

__device__ __forceinline__ float4 shade_hit(const RenderParams& P, const Ray& r, int x, int y, int z, int size) {
    return shade_color(shade_term(P, r, x, y, z, size));
}

// The framebuffer is written once and not read again by the frame: streaming (non-temporal) stores keep its 33 MB
// out of the way of the end-of-kernel cache write-back (-1.1 us per frame at config 2).  One global_store_dwordx4 nt.
__device__ __forceinline__ void store_pixel(float4* dst, const float4& c) {
    float* o = reinterpret_cast<float*>(dst);
    __builtin_nontemporal_store(c.x, o); __builtin_nontemporal_store(c.y, o + 1);
    __builtin_nontemporal_store(c.z, o + 2); __builtin_nontemporal_store(c.w, o + 3);
}

// Output modes of the traversal kernels.
constexpr int kModeColor = 0;   // RGBA32F framebuffer
constexpr int kModeSteps = 1;   // per-pixel +/-steps and frame counters (instrumentation)
constexpr int kModeTimeline = 2; // per-wave {start, end (100 MHz wall clock), loop iterations, HW_ID} in stepsOut (8 ints / tile)
constexpr int kPersistChunk = RTO_PERSIST_CHUNK;   // launch slots a persistent wave takes per atomic
constexpr int kModeShade = 3;    // one float per pixel: the Lambert term of the hit, kShadeMiss for a miss (multi-GPU payload)

struct Counters { unsigned long long pops, hits, capped; };

__device__ __forceinline__ void wave_accumulate(Counters* c, int steps, bool hit, bool valid) {
    // per-wave reduction, then one atomic per counter per wave
    unsigned long long pops = valid ? (unsigned long long)steps : 0ull;
    unsigned long long hits = (valid && hit) ? 1ull : 0ull;
    unsigned long long capped = (valid && !hit && steps >= kMaxTraversalSteps) ? 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
        pops += __shfl_down(pops, off);
        hits += __shfl_down(hits, off);
        capped += __shfl_down(capped, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&c->pops, pops);
        atomicAdd(&c->hits, hits);
        atomicAdd(&c->capped, capped);
    }
}

// ================================================================ generic kernel
// Direct statement of S/RT:239-327 over the 60-byte array.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_trace_generic(RenderParams P, const rto_node* __restrict__ nodes,
                                                           float4* __restrict__ out, int* __restrict__ stepsOut,
                                                           Counters* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int tx = tile % P.tilesX, ty = tile / P.tilesX;
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);

    bool hit = false;
    int steps = 0;
    float shade = kShadeMiss;
    if (inImage && P.rootVisible) {
        Ray r = generate_ray(P, px, py);
        int stack[128];                 // S/RT:247
        int sp = 0;
        stack[sp++] = 0;
        const float closestT = 1e30f;   // S/RT:242; only ever written together with the break
        while (sp > 0 && steps < kMaxTraversalSteps) {
            sp--;
            int nodeIdx = stack[sp];
            if (nodeIdx < 0) continue;
            steps++;
            const rto_node nd = nodes[nodeIdx];
            float tNear, tFar, a0, a1, a2, a3, a4, a5;
            if (!slab_exact(P, r, nd.x, nd.y, nd.z, nd.size, tNear, tFar, a0, a1, a2, a3, a4, a5)) continue;
            if (tNear >= closestT) continue;
            if (nd.isUniform == 1 || nd.isLeaf == 1) {   // S/RT:277-311, both branches have one body
                if (nd.isSolid == 1) {
                    float tHit = gmax(0.0f, tNear);
                    if (tHit < closestT && tHit <= tFar) {
                        hit = true;
                        shade = shade_term(P, r, nd.x, nd.y, nd.z, nd.size);
                        break;
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int c = nd.child[i];
                if (c >= 0) stack[sp++] = c;
            }
        }
    }
    if (MODE == kModeColor) {
        if (valid) out[(size_t)ly * P.W + px] = shade_color(shade);
    } else if (MODE == kModeShade) {
        if (valid) reinterpret_cast<float*>(out)[(size_t)ly * P.W + px] = shade;
    } else {
        if (inImage) stepsOut[(size_t)py * P.W + px] = hit ? steps : -steps;
        wave_accumulate(counters, steps, hit, inImage);
    }
}

// ================================================================ closest-hit kernel
// The reference's OTHER traversal rule for this shader: the earlier version kept block-commented in the same file
// (S/RT:63-138).  Same LIFO order, same slab test and the same `tNear >= closestT` pruning as above, but no break on the first
// accepted leaf and no step cap: the leaf with the smallest tHit wins, and a later leaf replaces it only when STRICTLY nearer (ties
// go to the leaf popped first).  Stated node by node over the 60-byte array, in the reference's pop order, so that the winner --
// which depends on that order through the pruning whenever two boxes' float planes disagree by an ulp -- is the reference's by
// construction.  Dead code upstream: built for completeness (rto_render_closest_*), not tuned.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_trace_closest(RenderParams P, const rto_node* __restrict__ nodes,
                                                           float4* __restrict__ out, Counters* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int tx = tile % P.tilesX, ty = tile / P.tilesX;
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);

    bool hit = false;
    int steps = 0;
    float shade = kShadeMiss;
    if (inImage && P.rootVisible) {
        Ray r = generate_ray(P, px, py);
        int stack[128];                 // S/RT:71
        int sp = 0;
        stack[sp++] = 0;
        float closestT = 1e30f;         // S/RT:66
        int bx = 0, by = 0, bz = 0, bs = 0;
        while (sp > 0) {                // S/RT:75
            sp--;
            int nodeIdx = stack[sp];
            if (nodeIdx < 0) continue;
            steps++;
            const rto_node nd = nodes[nodeIdx];
            float tNear, tFar, a0, a1, a2, a3, a4, a5;
            if (!slab_exact(P, r, nd.x, nd.y, nd.z, nd.size, tNear, tFar, a0, a1, a2, a3, a4, a5)) continue;     // S/RT:87-88
            if (tNear >= closestT) continue;                                                                 // S/RT:91-92
            if (nd.isUniform == 1 || nd.isLeaf == 1) {                                                       // S/RT:94-122
                if (nd.isSolid == 1) {
                    float tHit = gmax(0.0f, tNear);
                    if (tHit < closestT && tHit <= tFar) { closestT = tHit; hit = true; bx = nd.x; by = nd.y; bz = nd.z; bs = nd.size; }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {                                                                    // S/RT:125-129
                int c = nd.child[i];
                if (c >= 0 && sp < 128) stack[sp++] = c;       // (a LIFO walk of an octree of depth d holds at most 7 d + 1 entries)
            }
        }
        if (hit) shade = shade_term(P, r, bx, by, bz, bs);     // the normal of S/RT:101-103 at the winning leaf: p = o + d * tHit, tHit = max(0, tNear) of ITS box
    }
    if (MODE == kModeColor) {
        if (valid) out[(size_t)ly * P.W + px] = shade_color(shade);
    } else {
        unsigned long long pops = inImage ? (unsigned long long)steps : 0ull, hits = (inImage && hit) ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { pops += __shfl_down(pops, off); hits += __shfl_down(hits, off); }
        if (lane == 0) { atomicAdd(&counters->pops, pops); atomicAdd(&counters->hits, hits); }
        if (valid) out[(size_t)ly * P.W + px] = shade_color(shade);
    }
}

// ================================================================ packed kernel
// 8 child slab tests of the internal node at integer position (cx,cy,cz) whose children
// have edge `half`.  Returns the 8-bit mask of children k (bit0=+x, bit1=+y, bit2=+z,
// S/OctreeVoxel.cpp:751-754) with   tNear <= tFar && tFar > 0 && !(tNear >= 1e30)
// i.e. the ones that survive S/RT:269-274.
template <bool EXACT>
__device__ __forceinline__ unsigned child_pass_mask(const RenderParams& P, const Ray& r, int cx, int cy, int cz, int half) {
    using M = MinMax<EXACT>;
    const float vs = P.voxelSize;
    const float sv = (float)half * vs;                       // vec3(node.size) * voxelSize
    float tmn[3][2], tmx[3][2];
    const int c[3] = { cx, cy, cz };
    const float o[3] = { r.ox, r.oy, r.oz };
    const float inv[3] = { r.ix, r.iy, r.iz };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float flo = P.gridMin[a] + (float)c[a] * vs;          // nodeMin of the low children
        float fhi = P.gridMin[a] + (float)(c[a] + half) * vs; // nodeMin of the high children
        float mlo = flo + sv, mhi = fhi + sv;                 // their nodeMax
        float t1 = (flo - o[a]) * inv[a], t2 = (mlo - o[a]) * inv[a];
        tmn[a][0] = M::mn(t1, t2); tmx[a][0] = M::mx(t1, t2);
        float u1 = (fhi - o[a]) * inv[a], u2 = (mhi - o[a]) * inv[a];
        tmn[a][1] = M::mn(u1, u2); tmx[a][1] = M::mx(u1, u2);
    }
    unsigned pass = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        float tn = M::mx(M::mx(tmn[0][k & 1], tmn[1][(k >> 1) & 1]), tmn[2][k >> 2]);
        float tf = M::mn(M::mn(tmx[0][k & 1], tmx[1][(k >> 1) & 1]), tmx[2][k >> 2]);
        bool ok = (tn <= tf) && (tf > 0.0f) && !(tn >= 1e30f);
        pass |= ok ? (1u << k) : 0u;
    }
    return pass;
}

// LDS stack entry (one per tree level per lane):
//   .x = pending (8) | internalMask, unmasked (8) << 8 | visibleMask (8) << 16      .y = first-internal-child descriptor
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_trace_packed(RenderParams P, const uint2* __restrict__ desc,
                                                          float4* __restrict__ out, int* __restrict__ stepsOut,
                                                          Counters* __restrict__ counters) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * P.depth * kWave + lane;   // entry(level) = stk[level * 64]

    const int tile = blockIdx.x * (kBlock / kWave) + wave;
    const int tx = tile % P.tilesX, ty = tile / P.tilesX;
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);

    unsigned long long tl0 = 0;
    int tlIters = 0;
    if (MODE == kModeTimeline) tl0 = wall_clock64();
    bool hit = false;
    int steps = 0;
    int hx = 0, hy = 0, hz = 0, hs = 0;   // the solid leaf that was hit
    Ray r;
    bool alive = false;
    if (inImage && P.rootVisible) {
        r = generate_ray(P, px, py);
        // pop of the root: S/RT:254-274 with nodeIdx 0
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        steps = 1;
        alive = slab_exact(P, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
    }
    // A NaN can only enter a slab test through a non-finite reciprocal direction or origin.
    const bool risky = alive && !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                                  __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));

    int cur = 0;                 // descriptor of the node whose children are being popped
    int cx = 0, cy = 0, cz = 0;  // its integer position
    int half = P.rootSize >> 1;  // its children's edge
    int sp = 0;                  // == its level

    while (alive) {
        if (MODE == kModeTimeline) tlIters++;
        // ---- the node was popped and is internal: S/RT:313-318 pushes child[0..7], so they pop 7..0
        const uint2 d = desc[cur];
        const unsigned vis = (d.x >> 16) & 0xffu;
        const unsigned imaskAll = (d.x >> 8) & 0xffu;   // every internal child owns a descriptor, visible or not
        const unsigned imask = imaskAll & vis;
        const unsigned smask = d.x & vis;
        unsigned passMask;
        if (__builtin_amdgcn_ballot_w64(risky) != 0ull) passMask = child_pass_mask<true>(P, r, cx, cy, cz, half);
        else passMask = child_pass_mask<false>(P, r, cx, cy, cz, half);
        // children that do more than count a step: internal ones that pass the slab test, and solid
        // leaves that pass it (for those S/RT:279-280 always accepts: tHit=max(0,tNear)<=tFar, <1e30)
        unsigned pending = (imask | smask) & passMask;
        const unsigned solidHit = smask & passMask;
        if (solidHit) pending &= ~((1u << (31 - __builtin_clz(solidHit))) - 1u);   // nothing below the first hit is reached
        unsigned im = imaskAll, vm = vis, base = d.y;   // im ranks descriptors and tells internal from solid
        int prev = 8;   // children prev..7 of this node have been popped already
        // ---- pop children / climb until the next internal node is entered or the ray ends
        for (;;) {
            if (pending == 0) {
                steps += __builtin_popcount(vm & ((1u << prev) - 1u));   // the rest only count steps
                if (sp == 0 || steps >= kMaxTraversalSteps) { alive = false; break; }
                sp--;
                const uint2 e = stk[sp * kWave];
                pending = e.x & 0xffu; im = (e.x >> 8) & 0xffu; vm = (e.x >> 16) & 0xffu; base = e.y;
                half <<= 1;
                prev = ((cx & half) ? 1 : 0) | ((cy & half) ? 2 : 0) | ((cz & half) ? 4 : 0);
                const int keep = ~((half << 1) - 1);
                cx &= keep; cy &= keep; cz &= keep;
                continue;
            }
            const int j = 31 - __builtin_clz(pending);
            const int skipped = __builtin_popcount(vm & ((1u << prev) - 1u) & ~((2u << j) - 1u));
            if (steps + skipped >= kMaxTraversalSteps) { steps = kMaxTraversalSteps; alive = false; break; }   // S/RT:254 cap
            steps += skipped + 1;
            pending &= ~(1u << j);
            prev = j;
            const int nx = cx + ((j & 1) ? half : 0), ny = cy + ((j & 2) ? half : 0), nz = cz + ((j & 4) ? half : 0);
            if (!((im >> j) & 1u)) {   // solid leaf: S/RT:278-288 hit + break
                hit = true; hx = nx; hy = ny; hz = nz; hs = half;
                alive = false;
                break;
            }
            // internal child that passed its slab test: enter it
            stk[sp * kWave] = make_uint2(pending | (im << 8) | (vm << 16), base);
            sp++;
            cur = (int)(base + (unsigned)__builtin_popcount(im & ((1u << j) - 1u)));
            cx = nx; cy = ny; cz = nz;
            half >>= 1;
            break;
        }
    }
    if (!hit && steps > kMaxTraversalSteps) steps = kMaxTraversalSteps;

    if (MODE == kModeShade) {
        if (valid) reinterpret_cast<float*>(out)[(size_t)ly * P.W + px] = hit ? shade_term(P, r, hx, hy, hz, hs) : kShadeMiss;
    } else if (MODE == kModeColor || MODE == kModeTimeline) {
        if (valid) {
            float4 color = make_float4(0.f, 0.f, 0.f, 1.f);
            if (hit) color = shade_hit(P, r, hx, hy, hz, hs);
            out[(size_t)ly * P.W + px] = color;
        }
        if (MODE == kModeTimeline) {
            // wave-max of the per-lane iteration counts = iterations the wave executed
            int it = tlIters;
            for (int off = 32; off > 0; off >>= 1) it = max(it, __shfl_down(it, off));
            int act = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tlIters > 0));
            if (lane == 0 && ty < P.tilesY) {
                const unsigned long long tl1 = wall_clock64();
                unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
                unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID[3:0]
                int* rec = stepsOut + (size_t)tile * 8;
                rec[0] = (int)(tl0 & 0xffffffffu); rec[1] = (int)(tl0 >> 32);
                rec[2] = (int)(tl1 & 0xffffffffu); rec[3] = (int)(tl1 >> 32);
                rec[4] = it; rec[5] = (int)hwid; rec[6] = (int)xcc; rec[7] = act;
            }
        }
    } else {
        if (inImage) stepsOut[(size_t)py * P.W + px] = hit ? steps : -steps;
        wave_accumulate(counters, steps, hit, inImage);
    }
}

// ================================================================ packed kernel, low-latency form (helpers)
// Same traversal as k_trace_packed, restructured so that ONE loop iteration (= one internal node per
// lane) is a short straight-line block -- the frame time of this path is set by the deepest rays
// (~65 internal nodes) times the latency of one iteration, not by bandwidth:
//   * O(1) ascent: `lvlPending` has a bit per tree level that still holds unpopped interesting
//     children, so a finished subtree jumps straight to the deepest such level (one LDS read) instead
//     of climbing level by level.  The step counts of the exhausted levels that are skipped over are
//     pre-summed on the way down (`tailRun`; each stack entry remembers the sum above it).
//   * the three slab conditions  tNear<=tFar, tFar>0, tNear<1e30  are folded into the per-axis
//     half-plane terms:  max(tNear, FLT_TRUE_MIN) <= min(tFar, prev(1e30))  -- one compare per child.
//   * v_max3/v_min3 per child, and the 8 pass bits are shifted into the mask through the carry chain
//     (v_addc_co_u32), one VALU op per child.
// Descriptor/stack word layout (bits): [7:0] solid | pending, [15:8] internal (unmasked), [23:16] visible,
// [31:24] tail count above (stack entries only).

__device__ __forceinline__ int unrank_centre_out(int k, int c, int n) {
    // k-th element of 0..n-1 enumerated outwards from c: c, c+1, c-1, c+2, c-2, ... clipped to the range
    const int lo = c, hi = n - 1 - c;
    const int m = lo < hi ? lo : hi;
    if (k <= 2 * m) return c + ((k & 1) ? ((k + 1) >> 1) : -(k >> 1));
    return lo < hi ? k : n - 1 - k;
}

__device__ __forceinline__ void tile_of(const RenderParams& P, int t, int& tx, int& ty) {
    // t-th tile of the box, enumerated outwards from the tile nearest the projected scene centre
    const int rowRank = t / P.boxW, colRank = t - rowRank * P.boxW;
    ty = P.boxY0 + unrank_centre_out(rowRank, min(max(P.orderCy - P.boxY0, 0), P.boxH - 1), P.boxH);
    tx = P.boxX0 + unrank_centre_out(colRank, min(max(P.orderCx - P.boxX0, 0), P.boxW - 1), P.boxW);
}

// Launch slot -> tile.  Slots >= traceWaves exist only to share the fill duty (tiny boxes): no tile.  `slot` must be
// wave-uniform in a scalar register (the kernels pass it through readfirstlane): the table entry is then a scalar load and
// none of this costs VALU issue (as a per-lane value the old `tile / tilesX` was a 30-instruction VALU division per wave).
__device__ __forceinline__ bool resolve_slot(const RenderParams& P, int slot, int& tx, int& ty, int& tile) {
    tx = 0; ty = P.tilesY; tile = 0;
    if (slot >= P.traceWaves) return false;
    if (P.tileOrder) {
        // entries: tile inside the box, x | y << 16.  Clamped to the box (scalar arithmetic: `slot` is wave-uniform): a table is
        // only ever a schedule, and whatever it holds no wave may leave the frame's tile arrays
        const unsigned e = (unsigned)P.tileOrder[slot];
        tx = P.boxX0 + min((int)(e & 0xffffu), P.boxW - 1); ty = P.boxY0 + min((int)(e >> 16), P.boxH - 1);
    }
    else tile_of(P, slot, tx, ty);
    tile = ty * P.tilesX + tx;
    return true;
}

// Occupancy mask look-up for the tile (tx, ty) of this part (ty in local tile rows; a local tile lies inside one band, bands
// being multiples of 8 rows).  tx, ty, slot are wave-uniform.  The mask is built by the first workgroups of the SAME launch
// (mask_block): a wave never waits for it -- the first kMaskTrustSlots launch slots (the previous frame's costliest tiles: live
// anyway) do not even look, a later wave looks only once the "complete" word carries this frame's stamp and otherwise walks
// its tile as if there were no mask.  Tiles without work sort to the END of the launch order, and by the time their waves
// start the mask is long complete.  Agent-scope loads: the words were written by other workgroups, possibly on another XCD.
//
// The rim.  A tile the mask does not cover records its cost for the launch order right here: 0, or -1 when the mask covers a
// tile within kRimTiles tiles of it -- the rim of the silhouette, where a camera in motion finds work a frame or a few later;
// k_order_build ranks such tiles ahead of the certainly empty ones.  Without the rim the tiles a moving mask newly covers
// (12-14 trips, 6 us each) start among the ~8,000 work-less tiles at the very end of the launch and end the frame alone
// (config 2, a camera 0.01 rad from the one the costs come from: 36.8 us against 33.8; tools/timeline_simd.py learn=).
// It costs no extra round trip: the look-up is ONE load either way -- lane k reads the k-th word of the (2 kRimTiles + 1)^2
// neighbourhood, every other lane the "whole frame" word -- where it used to be two loads of one word each.
__device__ __forceinline__ bool tile_may_hit(const RenderParams& P, int tx, int ty, int slot, int lane = -1, int tile = 0) {
    if (!P.tileMask || ty >= P.tilesY || slot < P.maskTrustSlots) return true;
    if (__hip_atomic_load(P.tileMask + P.maskAllIndex + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.maskStamp) return true;   // not complete (yet)
    asm volatile("" ::: "memory");                             // the loads below are issued after this poll has matched, never hoisted above it
    // No acquire fence here: it would invalidate the whole CU's L1 (the descriptors!) once per wave.  The loads below are
    // agent-scope loads themselves (they do not read this CU's or this XCD's stale lines) and are issued after the branch on
    // the load above has resolved; the words they read were stored, and released, before the "complete" word.
    const int strip = global_row(P, ty * 8) >> 3;
    if (lane < 0)
        return __hip_atomic_load(P.tileMask + strip * P.tilesX + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.maskStamp ||
               __hip_atomic_load(P.tileMask + P.maskAllIndex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.maskStamp;
    constexpr int kSide = 2 * kRimTiles + 1, kCentre = kRimTiles * kSide + kRimTiles;
    const int ns = strip + lane / kSide - kRimTiles, nx = tx + lane % kSide - kRimTiles;
    const bool inside = lane < kSide * kSide && nx >= 0 && nx < P.tilesX && ns >= 0 && ns < ((P.H + 7) >> 3);
    const bool on = __hip_atomic_load(P.tileMask + (inside ? ns * P.tilesX + nx : P.maskAllIndex), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.maskStamp;
    const unsigned long long seen = __builtin_amdgcn_ballot_w64(on);
    const bool live = ((seen >> kCentre) & 1ull) != 0ull || (seen >> 63) != 0ull;      // lane 63 always reads the "whole frame" word
    if (!live && P.tileCost && lane == 0) P.tileCost[tile] = (seen & ((1ull << (kSide * kSide)) - 1ull)) ? -1 : 0;
    return live;
}

// Wave64 inclusive scans in registers (gfx9 DPP: shifts inside the rows of 16 lanes, then the two row broadcasts), no LDS
// round trips: lanes without a source, and rows outside the row mask, take the identity 0.
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);      // row_shr:1, 2, 4, 8
    v += dpp0<0x142, 0xa>(v);                                                                                    // row_bcast:15 into rows 1, 3
    v += dpp0<0x143, 0xc>(v);                                                                                    // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_scan_max_nonneg(int v) {       // values >= 0
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v));
    v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}

// ================================================================ launch order
// Launch order of the box's tiles by descending cost (trip count an earlier frame recorded; 64 buckets), built by G workgroups
// that never talk to each other: workgroup w owns the tiles i = w, w + G, w + 2G, ... of the box (raster order: a strided
// sample, so every workgroup sees the same cost distribution), sorts ITS tiles by a counting sort in LDS, and writes its p-th
// costliest tile to table position p * G + w.  The table is then the interleave of G sorted samples -- globally sorted up to the
// sampling noise, which is all a launch order needs -- and a permutation of the box's tiles BY CONSTRUCTION, whatever the cost
// array holds (stale values, tiles that were outside the box when it was written): every cost is read ONCE (its bucket is
// staged in LDS), and positions p * G + w, p < |sample w|, are exactly 0 .. n-1.  No state between calls, nothing shared between
// workgroups: safe as a node of a captured graph.  Entries: x | y << 16 relative to the box's corner.
// 4.8 us at config 2 against the 13 us of round 3's single-workgroup sort: cheap enough to run in front of EVERY frame of a camera
// in motion, whose costs are stale after one frame (the rays that wander along the silhouette are in other tiles every frame).
// (Tried and dropped in round 4: the same sort run by workgroups INSIDE the frame's own launch, for the next frame -- no launch, no
//  gap, but the tiles that matter, the costliest, finish last and had not recorded their cost yet: orbit 38.8 us per frame against
//  38.0 with the explicit rebuild, a static camera 34.4 against 33.5.)
constexpr int kOrderBuckets = 64;
constexpr int kOrderBlock = 1024;
constexpr int kOrderGroups = 16;

// sample w of G; `stage`: LDS, one byte per tile of the sample; `cnt`: LDS, kOrderBuckets ints
__device__ __forceinline__ void order_sort_sample(const int* __restrict__ tileCost, int tilesX, int boxX0, int boxY0, int boxW, int boxH, int w, int G,
                                                  int* __restrict__ order, int* __restrict__ violations, unsigned char* stage, int* cnt) {
    const int n = boxW * boxH;
    const int cw = (n - w + G - 1) / G;
    const int lane = threadIdx.x & 63, bd = (int)blockDim.x;
    auto tile_of_elem = [&](int e, int& rx, int& ry) { const int i = e * G + w; ry = i / boxW; rx = i - ry * boxW; };
    if ((int)threadIdx.x < kOrderBuckets) cnt[threadIdx.x] = 0;
    __syncthreads();
    // Bucket 0 -- tiles whose wave did not walk: with the occupancy mask more than half of a box -- is counted and placed by
    // ballot (one LDS atomic per wave): 64 lanes adding to ONE counter serialise.
    // bucket 0: nothing there; 1: no work, but inside the occupancy mask's margin (cost -1); 2..63: 1 + min(cost, 62)
    for (int e0 = (int)(threadIdx.x & ~63u); e0 < cw; e0 += bd) {           // wave-uniform trip count: ballots inside
        const int e = e0 + lane;
        int b = -1;
        if (e < cw) {
            int rx, ry;
            tile_of_elem(e, rx, ry);
            const int c = tileCost[(boxY0 + ry) * tilesX + boxX0 + rx];
            b = c > 0 ? min(c, kOrderBuckets - 2) + 1 : (c < 0 ? 1 : 0);
            stage[e] = (unsigned char)b;
        }
        const unsigned long long zeros = __builtin_amdgcn_ballot_w64(b == 0);
        if (b > 0) atomicAdd(&cnt[b], 1);
        if (zeros && lane == (int)__builtin_ctzll(zeros)) atomicAdd(&cnt[0], (int)__builtin_popcountll(zeros));
    }
    __syncthreads();
    if ((int)threadIdx.x < kWave) {                 // write cursors: costlier buckets first.  Lane b owns bucket 63 - b.
        const int b = kOrderBuckets - 1 - lane;
        const int total = cnt[b];
        cnt[b] = wave_scan_add(total) - total;      // exclusive prefix sum in registers (DPP), no LDS round trips
    }
    __syncthreads();
    for (int e0 = (int)(threadIdx.x & ~63u); e0 < cw; e0 += bd) {
        const int e = e0 + lane;
        const int b = e < cw ? (int)stage[e] : -1;
        const unsigned long long zeros = __builtin_amdgcn_ballot_w64(b == 0);
        int r = -1;
        if (b > 0) r = atomicAdd(&cnt[b], 1);
        if (zeros) {
            const int leader = (int)__builtin_ctzll(zeros);
            int base = 0;
            if (lane == leader) base = atomicAdd(&cnt[0], (int)__builtin_popcountll(zeros));
            base = __builtin_amdgcn_readlane(base, leader);
            if (b == 0) r = base + (int)__builtin_popcountll(zeros & ((1ull << lane) - 1ull));
        }
        if (b < 0) continue;
        int rx, ry;
        tile_of_elem(e, rx, ry);
        // cannot fall outside by construction (both passes see the same staged buckets); refused writes are counted and tests assert 0
        const int pos = r * G + w;
        if (r >= 0 && r < cw && pos < n) order[pos] = rx | (ry << 16);
        else atomicAdd(violations, 1);
    }
}

__global__ __launch_bounds__(kOrderBlock) void k_order_build(const int* __restrict__ tileCost, int tilesX, int boxX0, int boxY0, int boxW, int boxH,
                                                              int* __restrict__ order, int* __restrict__ violations) {
    extern __shared__ unsigned char order_stage[];  // one byte per tile of the sample
    __shared__ int cnt[kOrderBuckets];
    order_sort_sample(tileCost, tilesX, boxX0, boxY0, boxW, boxH, (int)blockIdx.x, (int)gridDim.x, order, violations, order_stage, cnt);   // G = the launch's own group count (kOrderGroups; dev builds: 16 / 32 / 64)
}

// ================================================================ screen-space occupancy mask
// About two thirds of the rays inside the solid geometry's screen rectangle miss everything (config 2: the corners of the
// sphere's bounding box, the ring around its silhouette) -- and each still costs its wave the ray set-up (two inversesqrt,
// three reciprocals, the root test: ~330 VALU instructions, as much as three loop trips).  The mask removes most of them
// before any ray exists: the "coarse cells" of the tree at one level L -- its internal nodes at depth L and every solid leaf
// at depth <= L: together they contain every solid leaf -- are projected onto the screen (conservatively: widened by a voxel,
// which also covers config 5's Marching-Cubes triangles, and by 3 pixels), and every tile a projection touches is stamped.
// A tile without the frame's stamp cannot contain a hit: its wave writes black (exactly what S/RT:363 gives rays that meet
// nothing) and exits.  Stamps instead of bits: every launch brings a fresh stamp, so the array is never cleared and all
// writers of a frame store the same value (no atomics; safe under graph replay: a replayed launch re-stamps its own tiles).
constexpr int kMaskMaxRectTiles = 2048;    // a cell that covers more tiles than this stamps the "whole frame" word instead

// A word of the mask leaves as an sc1 (agent-scope) store: it bypasses nothing on the way to memory that a reader on another XCD
// could miss, and its acknowledgement -- what s_waitcnt vmcnt(0) waits for -- means "visible to every CU of the device".
__device__ __forceinline__ void mask_put(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// four consecutive words at once (16-byte aligned): one fabric write instead of four
__device__ __forceinline__ void mask_put4(unsigned* p, unsigned v) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u q = { v, v, v, v };
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
}
// s_waitcnt vmcnt(0) as inline asm: invisible to the compiler's wait-count insertion, so it is never merged away.
__device__ __forceinline__ void drain_vector_memory() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One workgroup's share of a frame's mask: cell blockInFrame * blockDim + thread.  The workgroup that finishes last (ticket)
// publishes the "complete" word.
//
// Stamps leave the CU as sc1 stores (the hand-off below needs that), and a scalar sc1 store is one fabric write: the ~20 tiles
// a cell's rectangle covers, written cell by cell (112 k stores per config-2 frame, most of them the same few thousand words
// over and over), made the mask complete after the frame instead of 2 us into it (69 us per frame instead of 33).  So a
// workgroup first ORs its cells' rectangles into a bitmap of ITS tile bounding box in LDS -- its cells are neighbours in the
// tree's level order, the box is small -- and then stores one stamp per set bit, four at a time where four neighbours are set
// (mask_put4): ~7 k words per frame in ~2.5 k stores.  Measured at config 2, us per frame: plain stores (round 3's invalid
// form, with this dedupe) 33.2; this form 33.6; without the x4 stores 34.7; sc1 stores cell by cell 69.  `bits` / `capWords`:
// the launch's dynamic LDS (the traversal stacks of a tracing workgroup; a mask workgroup has no other use for it).  A
// workgroup whose box does not fit stores per cell as before (correct, slow: scattered cells in a very large frame).
__device__ __forceinline__ void mask_block(const RenderParams& P, int blockInFrame, unsigned* bits, int capWords) {
    unsigned* mask = P.tileMask;
    const unsigned stamp = P.maskStamp;
    const int all = P.maskAllIndex;
    const int i = blockInFrame * (int)blockDim.x + (int)threadIdx.x;
    int kind = 0, tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;          // 0: stamps nothing, 1: the tile rectangle, 2: the "whole frame" word
    // The first 8 words of the dynamic LDS hold the workgroup's tile bounding box and its "whole frame" flag (a static __shared__
    // array would add to every workgroup's LDS request, which the launchers size for a residency of 5 or 6 waves per
    // SIMD); the bitmap follows.  When a bitmap of the WHOLE frame fits (1080p: 4 KB) there is no bounding box to agree on: the
    // bitmap is cleared before the projection and two barriers do (with the box: four) -- the mask is complete ~1 us earlier,
    // and every wave that starts before that sets up rays for nothing.
    int* box = reinterpret_cast<int*>(bits);
    int& wholeFrame = box[4];
    bits += 8; capWords -= 8;
    const bool wide = (P.tilesX & 3) == 0;                                   // every mask row starts 16-byte aligned: runs of 4 set tiles leave as one store
    const int frameRowWords = (P.tilesX + 31) >> 5, frameWords = frameRowWords * ((P.H + 7) >> 3);
    const bool whole = frameWords <= capWords;                               // workgroup-uniform (every workgroup of the launch, in fact)
    if (whole) {
        for (int w = threadIdx.x; w < frameWords; w += blockDim.x) bits[w] = 0u;
        if (threadIdx.x == 0) wholeFrame = 0;
    }
    if (i < P.maskNumCells) {
        const int4 c = P.maskCells[i];
        const float vs = P.voxelSize;
        const float lo[3] = { P.gridMin[0] + (float)(c.x - 1) * vs, P.gridMin[1] + (float)(c.y - 1) * vs, P.gridMin[2] + (float)(c.z - 1) * vs };
        const float hi[3] = { P.gridMin[0] + (float)(c.x + c.w + 1) * vs, P.gridMin[1] + (float)(c.y + c.w + 1) * vs, P.gridMin[2] + (float)(c.z + c.w + 1) * vs };
        float lox = 3.0e38f, loy = 3.0e38f, hix = -3.0e38f, hiy = -3.0e38f;
        bool front = true;
        const float* V = P.viewRows;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float wx = (k & 1) ? hi[0] : lo[0], wy = (k & 2) ? hi[1] : lo[1], wz = (k & 4) ? hi[2] : lo[2];
            const float vx = V[0] * wx + V[1] * wy + V[2] * wz + V[3];
            const float vy = V[4] * wx + V[5] * wy + V[6] * wz + V[7];
            const float vz = V[8] * wx + V[9] * wy + V[10] * wz + V[11];
            // strictly in front of the eye, with room for the float error of the three dot products
            if (!(vz < -1e-4f * (1.0f + __builtin_fabsf(vx) + __builtin_fabsf(vy) + __builtin_fabsf(vz)))) front = false;
            const float iz = __builtin_amdgcn_rcpf(-vz);          // 1 ulp: the 3-pixel margin below is seven orders of magnitude larger
            const float sx = ((vx * iz) * P.maskInvAspTan * 0.5f + 0.5f) * (float)P.W, sy = (0.5f - (vy * iz) * P.maskInvTanH * 0.5f) * (float)P.H;
            lox = __builtin_fminf(lox, sx); hix = __builtin_fmaxf(hix, sx); loy = __builtin_fminf(loy, sy); hiy = __builtin_fmaxf(hiy, sy);
        }
        if (!front || !(hix - lox < 1.0e7f) || !(hiy - loy < 1.0e7f)) kind = 2;                  // around / behind the eye, or not finite
        else {
            // 3 pixels of margin: 2 as the host's rectangles + 1 for this float evaluation (their errors are ~1e-3 pixel)
            const float fx0 = __builtin_floorf(lox) - 3.0f, fx1 = __builtin_ceilf(hix) + 3.0f, fy0 = __builtin_floorf(loy) - 3.0f, fy1 = __builtin_ceilf(hiy) + 3.0f;
            if (!(fx1 < 0.0f || fy1 < 0.0f || fx0 > (float)(P.W - 1) || fy0 > (float)(P.H - 1))) {       // else: off the screen
                tx0 = (int)__builtin_fmaxf(fx0, 0.0f) >> 3; tx1 = (int)__builtin_fminf(fx1, (float)(P.W - 1)) >> 3;
                ty0 = (int)__builtin_fmaxf(fy0, 0.0f) >> 3; ty1 = (int)__builtin_fminf(fy1, (float)(P.H - 1)) >> 3;
                kind = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > kMaskMaxRectTiles ? 2 : 1;
            }
        }
    }
    int bx0 = 0, by0 = 0, bx1 = P.tilesX - 1, by1 = ((P.H + 7) >> 3) - 1;
    if (!whole) {                                                            // the workgroup's tile bounding box
        if (threadIdx.x == 0) { box[0] = box[1] = 0x7fffffff; box[2] = box[3] = -1; wholeFrame = 0; }
        __syncthreads();
        if (kind == 1) { atomicMin(&box[0], tx0); atomicMin(&box[1], ty0); atomicMax(&box[2], tx1); atomicMax(&box[3], ty1); }
        if (kind == 2) wholeFrame = 1;
        __syncthreads();
        bx0 = wide ? box[0] & ~3 : box[0]; by0 = box[1]; bx1 = box[2]; by1 = box[3];
    }
    const int rowWords = (bx1 - bx0 + 32) >> 5, rows = by1 - by0 + 1;        // a bitmap row: whole words
    const int words = rowWords * rows;
    if (bx1 >= bx0 && words <= capWords) {                                   // workgroup-uniform
        if (!whole) for (int w = threadIdx.x; w < words; w += blockDim.x) bits[w] = 0u;
        __syncthreads();                                                     // the bitmap is clear
        if (whole && kind == 2) wholeFrame = 1;
        if (kind == 1) {
            const int a = tx0 - bx0, b = tx1 - bx0;
            for (int wi = a >> 5; wi <= (b >> 5); wi++) {
                const int l = max(a - (wi << 5), 0), h = min(b - (wi << 5), 31);
                const unsigned m = (h == 31 ? ~0u : (1u << (h + 1)) - 1u) & ~((1u << l) - 1u);
                for (int y = ty0; y <= ty1; y++) atomicOr(&bits[(y - by0) * rowWords + wi], m);
            }
        }
        __syncthreads();
        for (int w = threadIdx.x; w < words; w += blockDim.x) {
            unsigned m = bits[w];
            if (!m) continue;
            const int row = w / rowWords, col = (w - row * rowWords) << 5;
            unsigned* base = mask + (size_t)(by0 + row) * P.tilesX + bx0 + col;
            if (wide) {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (((m >> (4 * q)) & 0xfu) == 0xfu) { mask_put4(base + 4 * q, stamp); m &= ~(0xfu << (4 * q)); }
            }
            while (m) { mask_put(base + __builtin_ctz(m), stamp); m &= m - 1u; }
        }
    } else if (kind == 1) {
        for (int y = ty0; y <= ty1; y++)
            for (int x = tx0; x <= tx1; x++) mask_put(mask + y * P.tilesX + x, stamp);
    }
    if (threadIdx.x == 0 && wholeFrame) mask_put(mask + all, stamp);
    // The hand-off to the frame's waves on other CUs / XCDs (tile_may_hit), in the gfx950 form whose consumer needs no acquire
    // (MI355X_MICROARCH.md, "inter-workgroup visibility", valid forms): EVERY stamp is an sc1 store (mask_put), EVERY storing wave
    // drains its stores before the workgroup's barrier, ONE lane signals for the whole workgroup behind that barrier, and every
    // signal -- the ticket, the "complete" word -- sits behind an explicit s_waitcnt that the compiler cannot drop (its
    // scoreboard pass removes the wait after buffer_wbl2 when a waited atomic has just emptied the counter: the flag could
    // overtake the write-back).
    drain_vector_memory();
    __syncthreads();                                             // every wave's stamps have left the CU and are acknowledged
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // belt and braces (0.15 us per frame for both): with it the producer side is ALSO the
                                                                 // guide's plain-store form -- stores, every wave's drain, barrier, lane-0 release, drain, signal
        drain_vector_memory();
        const unsigned last = (unsigned)P.maskBlocks - 1u;
        if (atomicInc(mask + all + 2, last) == last) {           // the ticket wraps to 0: the next launch / replay starts from zero
            // last ticket: every other mask workgroup's stamps were complete before ITS ticket, which this add has now followed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_vector_memory();
            __hip_atomic_store(mask + all + 1, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The fill duty of launch slot `slot`: chunks slot, slot + launchWaves, ... of the region outside the root rectangle.
// Every pixel there is black after the root's pop (S/RT:254-270, :363): (0,0,0,1), or kShadeMiss in the shade buffer.
template <class F>
__device__ __forceinline__ void for_each_outside_pixel(const RenderParams& P, int lane, int slot, F&& body) {
    const int per = P.fillLeftPer + P.fillRightPer;
    for (int c = slot; c < P.fillChunks; c += P.launchWaves) {            // wave-uniform
        size_t pix;
        bool ok;
        if (c < P.fillTopChunks) {
            const int p = c * kWave + lane;
            ok = p < P.fillTopRows * P.W; pix = (size_t)p;
        } else if (c < P.fillTopChunks + P.fillBotChunks) {
            const int p = (c - P.fillTopChunks) * kWave + lane;
            ok = p < (P.localRows - P.fillBotRow0) * P.W; pix = (size_t)P.fillBotRow0 * P.W + p;
        } else {
            const int m = c - P.fillTopChunks - P.fillBotChunks;
            const int row = m / per, k = m - row * per;
            int x;
            if (k < P.fillLeftPer) { x = k * kWave + lane; ok = x < P.fillLeftW; }
            else { x = P.fillRightX0 + (k - P.fillLeftPer) * kWave + lane; ok = x < P.W; }
            pix = (size_t)(P.fillTopRows + row) * P.W + x;
        }
        if (ok) body(pix);
    }
}
template <int MODE>
__device__ __forceinline__ void fill_outside(const RenderParams& P, float4* __restrict__ out, int lane, int slot) {
    if (!(MODE == kModeColor || MODE == kModeShade || MODE == kModeTimeline) || !P.skipOutside) return;
    for_each_outside_pixel(P, lane, slot, [&](size_t pix) {
        if (MODE == kModeShade) __builtin_nontemporal_store(kShadeMiss, reinterpret_cast<float*>(out) + pix);
        else store_pixel(out + pix, make_float4(0.f, 0.f, 0.f, 1.f));
    });
}

// hipcc does not fold fmax(fmax(a,b),c) into v_max3_f32 when a, b, c are themselves min/max results (it cannot
// prove them canonical), so the 3-input forms are spelled out; the assembler pads them with a few s_nop.
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Pass mask of the 8 children, fast form (no NaN can occur: see `risky`).
// The lo/hi child halves of every axis go through the same operations, so they are computed as 2-vectors:
// hipcc lowers these to v_pk_mul_f32 / v_pk_add_f32, which are IEEE-identical per element (no fusion) and cost a
// lone wave about as much as a scalar-float VALU op -- the critical path of the frame is exactly such lone deep
// waves, so halving the instruction count of this block shortens the frame, not just the wave.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned child_pass_mask_fast(const RenderParams& P, const Ray& r, int cx, int cy, int cz, int half) {
    const float vs = P.voxelSize;
    const float fh = (float)half;
    const float sv = fh * vs;                                  // vec3(node.size) * voxelSize
    const float kEps = __uint_as_float(1u);                   // smallest positive float: tFar > 0  <=>  tFar >= kEps
    const float kBelow1e30 = __uint_as_float(0x7149f2c9u);    // largest float < 1e30f:  tNear < 1e30 <=> tNear <= this
    float tmn[3][2], tmx[3][2];
    const int c[3] = { cx, cy, cz };
    const float o[3] = { r.ox, r.oy, r.oz };
    const float inv[3] = { r.ix, r.iy, r.iz };
    const f32x2 addHalf = { 0.0f, fh };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float fc = (float)c[a];
        const f32x2 fcc = (f32x2){ fc, fc } + addHalf;        // (fc, fc + fh): both exact (integers < 2^24; fc + 0 == fc)
        const f32x2 lo = (f32x2){ P.gridMin[a], P.gridMin[a] } + fcc * vs;     // nodeMin of the low / high children
        const f32x2 hi = lo + sv;                              // their nodeMax
        const f32x2 t1 = (lo - o[a]) * inv[a];
        const f32x2 t2 = (hi - o[a]) * inv[a];
        tmn[a][0] = __builtin_fminf(t1.x, t2.x); tmx[a][0] = __builtin_fmaxf(t1.x, t2.x);
        tmn[a][1] = __builtin_fminf(t1.y, t2.y); tmx[a][1] = __builtin_fmaxf(t1.y, t2.y);
    }
    // fold "tFar > 0" and "tNear < 1e30" into the x terms
    tmn[0][0] = __builtin_fmaxf(tmn[0][0], kEps); tmn[0][1] = __builtin_fmaxf(tmn[0][1], kEps);
    tmx[0][0] = __builtin_fminf(tmx[0][0], kBelow1e30); tmx[0][1] = __builtin_fminf(tmx[0][1], kBelow1e30);
    // verdict of child k = sign bit of (min3 - max3): no NaN can occur here, and a float difference carries the
    // exact sign of the comparison (x - x is +0).  v_alignbit shifts the sign into an accumulator: two 4-deep chains.
#define RTO_D(k) __float_as_uint(min3f(tmx[0][(k) & 1], tmx[1][((k) >> 1) & 1], tmx[2][(k) >> 2]) - \
                                 max3f(tmn[0][(k) & 1], tmn[1][((k) >> 1) & 1], tmn[2][(k) >> 2]))
    unsigned fa = 0, fb = 0;
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(7), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(3), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(6), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(2), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(5), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(1), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(4), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(0), 31);
#undef RTO_D
    return ~((fa << 4) | fb) & 0xffu;
}

// ================================================================ packed kernel, branch-free loop body
// The low-latency algorithm above with the loop body written as straight-line selects: the two lane classes of
// an iteration ("entered a node with work" / "entered a node without work, resume at the deepest level
// that has some") are merged through one packed state word W = pending | internal<<8 | visible<<16 |
// tailAbove<<24 that comes either from the node's descriptor or from the LDS stack entry.  The LDS read is
// unconditional (harmless when unused), the only predicated memory operation is the stack write.
// One tile (= one wave's 64 rays) from launch slot `slot`.
template <int MODE>
__device__ __forceinline__ void trace_tile_packed3(const RenderParams& P, const uint2* __restrict__ desc, float4* __restrict__ out,
                                                   int* __restrict__ stepsOut, Counters* __restrict__ counters, uint2* stk,
                                                   const int lane, const int slot) {
    unsigned long long tl0 = 0;
    int tlIters = 0;
    if (MODE == kModeTimeline) tl0 = wall_clock64();

    // Launch order.  The frame ends when its deepest waves end, so they must start first.  With temporal order the
    // slot -> tile table lists the tiles by their trip count in the previous frame (a scheduling hint only: every
    // tile is rendered exactly once either way); without history, tiles go centre-out from the projected geometry.
    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);

    bool hit = false;
    int steps = 0;
    Ray r;
    bool alive = false;
    // A pixel outside the (conservative, host-computed) screen rectangle of the root box cannot hit it: the root's
    // slab test fails for it, the pixel is black after one pop (S/RT:254-270).  Whole tiles out there skip ray setup.
    const bool outsideRoot = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;
    if (inImage && P.rootVisible) {
        steps = 1;   // the root's own pop
        if (!outsideRoot) {
            r = generate_ray_tab(P, px, py);
            float tNear, tFar, a0, a1, a2, a3, a4, a5;
            alive = slab_exact(P, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
        }
    }
    const bool risky = alive && !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                                  __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));

    const bool anyRisky = __builtin_amdgcn_ballot_w64(risky) != 0ull;   // wave-uniform and fixed for the whole traversal

    unsigned cur = 0;
    int cx = 0, cy = 0, cz = 0;
    int lvl = 0;
    unsigned lvlPending = 0;
    unsigned tailRun = 0;

#if defined(RTO_STAMP)
    unsigned long long stampSum[5] = { 0, 0, 0, 0, 0 };
#define RTO_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); stampSum[i] += t_ - stampLast; stampLast = t_; } while (0)
    unsigned long long stampLast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stampLast) :: "memory");
#else
#define RTO_T(i) do { } while (0)
#endif
    int trips = 0;                                 // wave-uniform (scalar) trip count = this tile's cost
    while (alive) {
        trips++;
        if (MODE == kModeTimeline) tlIters++;
        RTO_T(4);                                   // loop overhead / back edge
        // ---- [A] the node `cur` at (cx,cy,cz), level lvl
        const uint2 d = desc[cur];
        // the resume entry depends only on lvlPending: fetch it now, under the descriptor load and the slab math
        const int L = 31 - __builtin_clz(lvlPending | 1u);
        const uint2 e = stk[L * kWave];
        const int half = 1 << (P.depth - 1 - lvl);
        unsigned passMask;
        if (anyRisky) passMask = child_pass_mask<true>(P, r, cx, cy, cz, half);
        else passMask = child_pass_mask_fast(P, r, cx, cy, cz, half);
        RTO_T(0);                                   // slab math (descriptor + LDS loads in flight)
        const unsigned vm0 = __builtin_amdgcn_ubfe(d.x, 16, 8);
        const unsigned im0 = __builtin_amdgcn_ubfe(d.x, 8, 8);
        const unsigned sm = d.x & vm0;                                  // solid & visible
        unsigned cand = ((im0 & vm0) | sm) & passMask;
#if defined(RTO_STAMP)
        asm volatile("" :: "v"(cand));
#endif
        RTO_T(1);                                   // wait for the descriptor
        const unsigned solidHit = sm & passMask;
        cand &= 0xffffffffu << (31 - __builtin_clz(solidHit | 1u));      // nothing below the first solid hit is reached
        // ---- [B] merge with "resume at the deepest level that still has work"
        const bool noWork = cand == 0;
        const int stepsB = steps + __builtin_popcount(vm0) + (int)tailRun;
        const bool dead = noWork && (lvlPending == 0 || stepsB >= kMaxTraversalSteps);
        steps = noWork ? stepsB : steps;
        const unsigned W = noWork ? e.x : ((d.x & 0x00ffff00u) | cand | (tailRun << 24));
        const unsigned base = noWork ? e.y : d.y;
#if defined(RTO_STAMP)
        asm volatile("" :: "v"(W), "v"(base));
#endif
        RTO_T(2);                                   // [B] up to the merged state word (waits for the LDS entry)
        const int lvl2 = noWork ? L : lvl;
        const int bpos = P.depth - 1 - lvl2;                             // log2 of the child edge at lvl2
        const unsigned childIdx = ((cx >> bpos) & 1) | (((cy >> bpos) & 1) << 1) | (((cz >> bpos) & 1) << 2);
        const unsigned belowPrev = noWork ? ((1u << childIdx) - 1u) : 0xffu;   // children not popped yet at lvl2
        const int keep = (int)(0xffffffffu << (bpos + 1));
        cx &= keep; cy &= keep; cz &= keep;                              // no-op for a freshly entered node
        // ---- [C] pop the next interesting child of level lvl2
        const unsigned pending = W & 0xffu;
        const unsigned im = __builtin_amdgcn_ubfe(W, 8, 8);
        const unsigned vm = __builtin_amdgcn_ubfe(W, 16, 8);
        const unsigned tailAbove = W >> 24;
        const int j = 31 - __builtin_clz(pending | 1u);
        const unsigned bitj = 1u << j;
        const int stepsC = steps + __builtin_popcount(vm & belowPrev & ~((bitj << 1) - 1u));
        const bool capped = stepsC >= kMaxTraversalSteps;                // S/RT:254: the cap ends the loop before this pop
        const bool go = !dead && !capped;
        const bool solid = (im & bitj) == 0;
        steps = dead ? steps : (capped ? kMaxTraversalSteps : stepsC + 1);
        hit = go && solid;                                               // S/RT:278-288
        const bool descend = go && !solid;
        if (descend) stk[lvl2 * kWave] = make_uint2(W ^ bitj, base);
        const unsigned pendingNew = pending ^ bitj;
        const unsigned lb = 1u << lvl2;
        tailRun = pendingNew ? 0u : tailAbove + (unsigned)__builtin_popcount(vm & (bitj - 1u));
        lvlPending = pendingNew ? (lvlPending | lb) : (lvlPending & ~lb);
        cur = base + (unsigned)__builtin_popcount(im & (bitj - 1u));
        const int hl = 1 << bpos;
        cx |= (j & 1) ? hl : 0; cy |= (j & 2) ? hl : 0; cz |= (j & 4) ? hl : 0;   // the popped child (kept on a hit)
        lvl = lvl2 + 1;
        alive = descend;
#if defined(RTO_STAMP)
        asm volatile("" :: "v"(cur), "v"(cx), "v"(lvlPending), "v"(tailRun));
#endif
        RTO_T(3);                                   // [C] pop, stack write, next node
    }
    if (!hit && steps > kMaxTraversalSteps) steps = kMaxTraversalSteps;
    trips = __builtin_amdgcn_readlane(wave_scan_max_nonneg(trips), kWave - 1);      // of the tile's busiest ray (see trace_tile_lean)
    if (P.tileCost && lane == 0 && ty < P.tilesY) P.tileCost[tile] = trips;

    const bool mine = valid && !(P.skipOutside && outsideRoot);     // pixels outside the root rectangle belong to the fill duty
    if (MODE == kModeShade) {
        if (mine) __builtin_nontemporal_store(hit ? shade_term(P, r, cx, cy, cz, P.rootSize >> lvl) : kShadeMiss, reinterpret_cast<float*>(out) + (size_t)ly * P.W + px);
        fill_outside<MODE>(P, out, lane, slot);
    } else if (MODE == kModeColor || MODE == kModeTimeline) {
        if (mine) {
            float4 color = make_float4(0.f, 0.f, 0.f, 1.f);
            if (hit) color = shade_hit(P, r, cx, cy, cz, P.rootSize >> lvl);
            store_pixel(out + (size_t)ly * P.W + px, color);
        }
        fill_outside<MODE>(P, out, lane, slot);
        if (MODE == kModeTimeline) {
            int it = tlIters;
            for (int off = 32; off > 0; off >>= 1) it = max(it, __shfl_down(it, off));
            int act = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tlIters > 0));
            if (lane == 0 && ty < P.tilesY) {
                const unsigned long long tl1 = wall_clock64();
                unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
                unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
                int* rec = stepsOut + (size_t)tile * 8;
                rec[0] = (int)(tl0 & 0xffffffffu); rec[1] = (int)(tl0 >> 32);
                rec[2] = (int)(tl1 & 0xffffffffu); rec[3] = (int)(tl1 >> 32);
                rec[4] = it; rec[5] = (int)hwid; rec[6] = (int)xcc; rec[7] = act;
#if defined(RTO_STAMP)
                rec[0] = (int)stampSum[0]; rec[1] = (int)stampSum[1]; rec[5] = (int)stampSum[2]; rec[6] = (int)stampSum[3]; rec[7] = (int)stampSum[4];
#endif
            }
        }
    } else {
        if (inImage) stepsOut[(size_t)py * P.W + px] = hit ? steps : -steps;
        wave_accumulate(counters, steps, hit, inImage);
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock, RTO_PACKED3_WAVES) void k_trace_packed3(RenderParams P, const uint2* __restrict__ desc,
                                                           float4* __restrict__ out, int* __restrict__ stepsOut,
                                                           Counters* __restrict__ counters) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * P.depth * kWave + lane;
    const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    if (slot >= P.launchWaves) return;
    trace_tile_packed3<MODE>(P, desc, out, stepsOut, counters, stk, lane, slot);
}

// ================================================================ packed kernel, lean loop body (the default)
// k_trace_packed3 spends about as many VALU instructions on bookkeeping as on the 8 slab tests, and most of the
// bookkeeping exists for ONE purpose: knowing `traversalSteps` (S/RT:254, :260) at every pop, so that the 512-pop cap
// can end the loop.  This body does not count pops inside the loop at all.  It relies on two identities of the
// reference's LIFO traversal (every visible child of an entered internal node is pushed, S/RT:313-318, and each of
// them is popped exactly once unless the loop ends first):
//
//   (1) a ray that exhausts its stack has popped   T = 1 + S   nodes, S = sum of popcount(visible children) over the
//       internal nodes it entered.  One v_bcnt_u32_b32 per loop trip accumulates S.
//   (2) when a solid leaf is accepted, the pops so far are   1 + S - R , where R counts the pushed-but-unpopped
//       nodes: for every node on the path from the root to the leaf's parent, its visible children BELOW the child
//       the path took.  R is rebuilt ONCE per ray after the loop from the per-level LDS entries (<= depth reads), and
//       only when it can matter: 1 + S <= 512 already proves the hit lies within the cap.
//
// A hit with 1 + S - R > 512 is a hit the reference never reaches (its loop stopped at 512 pops): the pixel is black
// and the step count 512, exactly what S/RT:254 produces.  A ray is abandoned as soon as S >= 512 + 7*depth, because
// R <= 7*depth makes every later hit fall beyond the cap.  Pixels, per-pixel step counts and frame counters are
// therefore identical to k_trace_packed3's; the loop shrinks from ~180 to ~125 VALU instructions per trip:
// no tail pre-sums, no belowPrev masks, no cap compare per pop, no "first solid hit" cut of the candidate mask
// (children below a solid hit are never popped anyway: the hit ends the ray).
//
// LDS entry of a node whose children have edge 2^b (entry b, b = depth-1-level): .x = pending children (8) | internal
// mask, unmasked (8) << 8 | visible mask (8) << 16, .y = first-internal-child descriptor.  The entry is written on
// every pop -- also when a solid leaf is accepted -- so that after the loop the entries of the leaf's ancestors all
// describe nodes of the hit path (their visible masks feed R).

// 3-input bitwise op, full issue rate on gfx950 (v_bitop3_b32; v_and_or / v_lshl_or / v_bfi / v_cndmask issue at half rate,
// tools/ubench/valu_rate2.hip).  TT = truth table, bit (a<<2 | b<<1 | c).
template <int TT> __device__ __forceinline__ unsigned bop3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
constexpr int kSelC = 0xD8;       // c ? b : a      (c is an all-ones / all-zeros lane mask)
constexpr int kAndOr = 0xEA;      // (a & b) | c
constexpr int kOrAnd = 0xF8;      // a | (b & c)
constexpr int kAndAndNot = 0x40;  // a & b & ~c
constexpr int kReplace = 0x74;    // (a & ~b) | (b & ~c)

// Children that FAIL S/RT:269-274 (bit k set), fast form (no NaN can occur: see `risky`); arithmetic of
// child_pass_mask_fast with one change: min(t1,t2) / max(t1,t2) of an axis are SELECTED by the sign of the ray's
// reciprocal direction instead of computed.  hi >= lo (hi = lo + sv, sv > 0) and float subtraction / multiplication by a
// finite non-zero factor are monotonic, so t1 <= t2 exactly when inv > 0: the select returns the same value as
// v_min / v_max up to the sign of a zero, which no later operation observes -- at full rate (v_bitop3) instead of half
// rate (v_min_f32 / v_max_f32).  sgn[a] = all ones when inv[a] < 0.
// one axis of child_fail_mask_fast: entry / exit parameters of the low (0) and high (1) child halves
// BIASED: c arrives as 0x4B000000 | coordinate -- the bits of the float 2^23 + coordinate --, so that the conversion is one
// full-rate subtraction (exact) instead of a half-rate v_cvt_f32_i32; the same value either way.
constexpr unsigned kCoordBias = 0x4B000000u;
template <bool BIASED>
__device__ __forceinline__ void child_axis_terms(float g, float o, float inv, unsigned sgn, int c, float fh, float vs, float sv,
                                                 float& n0, float& n1, float& f0, float& f1) {
    const float fc = BIASED ? __uint_as_float((unsigned)c) - 8388608.0f : (float)c;
    const f32x2 fcc = (f32x2){ fc, fc } + (f32x2){ 0.0f, fh };       // (fc, fc + fh): both exact (integers < 2^24; fc + 0 == fc)
    const f32x2 lo = (f32x2){ g, g } + fcc * vs;                      // nodeMin of the low / high children
    const f32x2 hi = lo + sv;                                         // their nodeMax
    const f32x2 t1 = (lo - o) * inv;
    const f32x2 t2 = (hi - o) * inv;
    const unsigned a1x = __float_as_uint(t1.x), a2x = __float_as_uint(t2.x), a1y = __float_as_uint(t1.y), a2y = __float_as_uint(t2.y);
    n0 = __uint_as_float(bop3<kSelC>(a1x, a2x, sgn)); f0 = __uint_as_float(bop3<kSelC>(a2x, a1x, sgn));
    n1 = __uint_as_float(bop3<kSelC>(a1y, a2y, sgn)); f1 = __uint_as_float(bop3<kSelC>(a2y, a1y, sgn));
}

// The same on a grid whose planes the host has proven EXACT (rto_api.hip, grid_is_exact): for every k = 0 .. rootSize the product
// k * voxelSize and the sum gridMin + k * voxelSize are computed without rounding, i.e. every plane the reference's arithmetic
// (S/RT:265-266) produces for any node IS the real number gridMin + k * vs.  Then any expression whose real value is that number and
// whose last operation is a single correctly rounded one yields the very same float: the node's three planes per axis are
//   p0 = fma(c, vs, g)            (= g + c * vs: one rounding of an exactly representable value)
//   p1 = p0 + fh * vs             (the low child's nodeMax AND the high child's nodeMin: the same real number, hence the same float)
//   p2 = p1 + fh * vs             (the high child's nodeMax)
// and the entry / exit parameters follow from them by the reference's own two operations, (p - o) * inv: bit for bit the values
// child_axis_terms computes from 4 planes, with 3 planes and without the packed multiply / add pairs.
template <bool BIASED>
__device__ __forceinline__ void child_axis_terms_exact(float g, float o, float inv, unsigned sgn, int c, float fh, float vs, float sv,
                                                       float& n0, float& n1, float& f0, float& f1) {
    (void)fh;
    const float fc = BIASED ? __uint_as_float((unsigned)c) - 8388608.0f : (float)c;
    const float p0 = __builtin_fmaf(fc, vs, g);
    const float p1 = p0 + sv;
    const float p2 = p1 + sv;
    const f32x2 t01 = ((f32x2){ p0, p1 } - o) * inv;
    const float t2 = (p2 - o) * inv;
    const unsigned a0 = __float_as_uint(t01.x), a1 = __float_as_uint(t01.y), a2 = __float_as_uint(t2);
    n0 = __uint_as_float(bop3<kSelC>(a0, a1, sgn)); f0 = __uint_as_float(bop3<kSelC>(a1, a0, sgn));
    n1 = __uint_as_float(bop3<kSelC>(a1, a2, sgn)); f1 = __uint_as_float(bop3<kSelC>(a2, a1, sgn));
}

// clampLo / clampHi (FOLD only): what the entry / exit parameters are clipped with before they are compared.  Defaults: the
// smallest positive float and the largest float below 1e30 (S/RT:235, :273); the octreeRaySkip traversal passes its parent
// interval [enterT, exitT] instead (S/VR:97-100: a child passes when max(tNear, enterT) <= min(tFar, exitT)).
template <bool FOLD = true, bool BIASED = false, bool EXACT = false>
__device__ __forceinline__ unsigned child_fail_mask_fast(float gx, float gy, float gz, float vs, float ox, float oy, float oz,
                                                         float ix, float iy, float iz, unsigned sx, unsigned sy, unsigned sz,
                                                         int cx, int cy, int cz, float fh,
                                                         float clampLo = __uint_as_float(1u), float clampHi = __uint_as_float(0x7149f2c9u)) {
    const float sv = fh * vs;                                  // vec3(node.size) * voxelSize
    const float kEps = clampLo;                               // default: smallest positive float: tFar > 0  <=>  tFar >= kEps
    const float kBelow1e30 = clampHi;                         // default: largest float < 1e30f:  tNear < 1e30 <=> tNear <= this
    float nx0, nx1, fx0, fx1, ny0, ny1, fy0, fy1, nz0, nz1, fz0, fz1;
    if (EXACT) {
        child_axis_terms_exact<BIASED>(gx, ox, ix, sx, cx, fh, vs, sv, nx0, nx1, fx0, fx1);
        child_axis_terms_exact<BIASED>(gy, oy, iy, sy, cy, fh, vs, sv, ny0, ny1, fy0, fy1);
        child_axis_terms_exact<BIASED>(gz, oz, iz, sz, cz, fh, vs, sv, nz0, nz1, fz0, fz1);
    } else {
        child_axis_terms<BIASED>(gx, ox, ix, sx, cx, fh, vs, sv, nx0, nx1, fx0, fx1);
        child_axis_terms<BIASED>(gy, oy, iy, sy, cy, fh, vs, sv, ny0, ny1, fy0, fy1);
        child_axis_terms<BIASED>(gz, oz, iz, sz, cz, fh, vs, sv, nz0, nz1, fz0, fz1);
    }
    // fold "tFar > 0" and "tNear < 1e30" into the x terms (spelled as instructions: behind a bitwise select the compiler
    // would first canonicalise the operand with an extra v_max_f32 x, x)
    // FOLD = false (see `plainWave` in trace_tile_lean): every ray of the wave starts outside the root box, more than a few ulps
    // of its coordinates away, and meets it at 0 < tNear, tFar < 1e29.  A child's box lies inside the root's (up to an ulp), so
    // whenever its own tNear <= tFar holds, both lie in the root's interval: tFar > 0 and tNear < 1e30 follow and the two
    // folds change no verdict.
    if (FOLD) {
        asm("v_max_f32 %0, %1, %2" : "=v"(nx0) : "v"(nx0), "v"(kEps));
        asm("v_max_f32 %0, %1, %2" : "=v"(nx1) : "v"(nx1), "v"(kEps));
        asm("v_min_f32 %0, %1, %2" : "=v"(fx0) : "v"(fx0), "v"(kBelow1e30));
        asm("v_min_f32 %0, %1, %2" : "=v"(fx1) : "v"(fx1), "v"(kBelow1e30));
    }
    // verdict of child k = sign bit of (min3 - max3): no NaN can occur here, and a float difference carries the
    // exact sign of the comparison (x - x is +0).  v_alignbit shifts the sign into an accumulator: two 4-deep chains.
#define RTO_D(kx, ky, kz) __float_as_uint(min3f(fx##kx, fy##ky, fz##kz) - max3f(nx##kx, ny##ky, nz##kz))
    unsigned fa = 0, fb = 0;
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(1, 1, 1), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(1, 1, 0), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(0, 1, 1), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(0, 1, 0), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(1, 0, 1), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(1, 0, 0), 31);
    fa = __builtin_amdgcn_alignbit(fa, RTO_D(0, 0, 1), 31); fb = __builtin_amdgcn_alignbit(fb, RTO_D(0, 0, 0), 31);
#undef RTO_D
    return (fa << 4) | fb;
}

// ---- the 8 verdicts on an EXACT grid, without a single per-child min / max ---------------------------------------------------
// On a grid whose planes are exact (grid_is_exact) a node has three planes per axis, p0 < p1 < p2, and -- subtraction and the
// multiplication by a finite non-zero factor being monotonic -- their parameters come out ordered: with the axis' entry parameter
// e = t(first plane the ray meets), m = t(p1) and exit parameter x, e <= m <= x holds in float.  A child is a choice, per axis, of the
// ENTRY half [e, m] or the EXIT half [m, x]; write S for the axes on which it takes the exit half.  Its verdict in the reference,
// max3(near) <= min3(far) (S/RT:226-236 on the child's box), compares a maximum with a minimum, i.e. asks that EVERY near value be
// <= EVERY far value.  Same-axis pairs hold by the ordering; the others are, with Tn = max(e_x, e_y, e_z) and Tf = min(x_x, x_y, x_z)
// (the node's own interval; the folds tFar > 0, tNear < 1e30 join them as one more lower / upper bound):
//      A_a: m_a <= Tf   for every a in S         B_a: Tn <= m_a   for every a not in S         C_ab: m_a <= m_b   for a in S, b not in S
// -- twelve comparisons of nine numbers per node instead of 8 x (max3, min3, compare), each one float subtraction whose sign IS the
// comparison (no NaN on this path), spread to a lane mask by one arithmetic shift.  (Tn <= Tf itself needs no test: it is the verdict
// that made the walk enter this node, computed from the same floats in its parent's trip -- or by the root test.)  Which CHILD
// INDICES a failed comparison strikes follows from the ray's direction signs alone: ka_a = the children whose a-half is the exit
// half (0xAA / 0xCC / 0xF0, complemented where the ray runs down the axis), x_ab = ka_a ^ ka_b -- six words per ray, made once.
// The result's bits above 7 are not cleared: every caller masks it with a descriptor's 8-bit visibility mask.
struct ExactPat { unsigned kaX, kaY, kaZ, xXY, xXZ, xYZ; };
__device__ __forceinline__ ExactPat exact_patterns(unsigned sgnX, unsigned sgnY, unsigned sgnZ) {
    ExactPat p;
    p.kaX = 0xAAu ^ (sgnX & 0xFFu); p.kaY = 0xCCu ^ (sgnY & 0xFFu); p.kaZ = 0xF0u ^ (sgnZ & 0xFFu);
    p.xXY = p.kaX ^ p.kaY; p.xXZ = p.kaX ^ p.kaZ; p.xYZ = p.kaY ^ p.kaZ;
    return p;
}
constexpr int kOr3 = 0xFE;        // a | b | c
// entry, mid and exit parameter of one axis (child_axis_terms_exact's planes)
template <bool BIASED>
__device__ __forceinline__ void axis_emx_exact(float g, float o, float inv, unsigned sgn, int c, float vs, float sv, float& e, float& m, float& x) {
    const float fc = BIASED ? __uint_as_float((unsigned)c) - 8388608.0f : (float)c;
    const float p0 = __builtin_fmaf(fc, vs, g);
    const float p1 = p0 + sv;
    const float p2 = p1 + sv;
    const f32x2 t02 = ((f32x2){ p0, p2 } - o) * inv;                  // the two outer planes as a pair (one packed subtract, one packed multiply)
    m = (p1 - o) * inv;                                               // the mid plane on its own: it goes into six single subtractions
    const unsigned a0 = __float_as_uint(t02.x), a2 = __float_as_uint(t02.y);
    e = __uint_as_float(bop3<kSelC>(a0, a2, sgn)); x = __uint_as_float(bop3<kSelC>(a2, a0, sgn));
}
// all ones where u <= v FAILS: the sign of v - u (a float difference has the exact sign of the comparison; x - x is +0).
// Two corner cases give "fails" where the comparison holds, and neither can change a child's verdict:
//   * u = +0, v = -0 (v - u = -0): v then is some child's UPPER bound -- a mid parameter m_b with b in the entry half, or Tf -- and
//     equals zero, while that child's lower bound Tn is > 0 (the fold's smallest positive float; in fold-free waves every ray meets the
//     root box at tNear > 0): Tn <= m_b fails for it in any case (B_b), and Tf = -0 < Tn means the node was never entered;
//   * inf - inf = NaN of either sign (parameters that overflowed): the children concerned have m_a = +inf as a lower bound (m_a <= Tf
//     fails, Tf < 1e30) or m_b = -inf as an upper one (Tn <= m_b fails).
__device__ __forceinline__ unsigned fails_le(float u, float v) { return (unsigned)((int)__float_as_uint(v - u) >> 31); }

template <bool FOLD, bool BIASED>
__device__ __forceinline__ unsigned child_fail_mask_exactgrid(float gx, float gy, float gz, float vs, float ox, float oy, float oz,
                                                              float ix, float iy, float iz, unsigned sx, unsigned sy, unsigned sz, const ExactPat& K,
                                                              int cx, int cy, int cz, float fh,
                                                              float clampLo = __uint_as_float(1u), float clampHi = __uint_as_float(0x7149f2c9u)) {
    const float sv = fh * vs;
    float ex, mx, xx, ey, my, xy, ez, mz, xz;
    axis_emx_exact<BIASED>(gx, ox, ix, sx, cx, vs, sv, ex, mx, xx);
    axis_emx_exact<BIASED>(gy, oy, iy, sy, cy, vs, sv, ey, my, xy);
    axis_emx_exact<BIASED>(gz, oz, iz, sz, cz, vs, sv, ez, mz, xz);
    float Tn = max3f(ex, ey, ez), Tf = min3f(xx, xy, xz);
    if (FOLD) {
        asm("v_max_f32 %0, %1, %2" : "=v"(Tn) : "v"(Tn), "v"(clampLo));      // tFar > 0: the smallest positive float is one more lower bound
        asm("v_min_f32 %0, %1, %2" : "=v"(Tf) : "v"(Tf), "v"(clampHi));      // tNear < 1e30: the largest float below it one more upper bound
    }
    const unsigned selX = bop3<kSelC>(fails_le(Tn, mx), fails_le(mx, Tf), K.kaX);    // children with x in S: A_x; the others: B_x
    const unsigned selY = bop3<kSelC>(fails_le(Tn, my), fails_le(my, Tf), K.kaY);
    const unsigned selZ = bop3<kSelC>(fails_le(Tn, mz), fails_le(mz, Tf), K.kaZ);
    unsigned fail = bop3<kOr3>(selX, selY, selZ);
    const unsigned pXY = bop3<kSelC>(fails_le(my, mx), fails_le(mx, my), K.kaX);     // x in S, y not: C_xy; y in S, x not: C_yx
    const unsigned pXZ = bop3<kSelC>(fails_le(mz, mx), fails_le(mx, mz), K.kaX);
    const unsigned pYZ = bop3<kSelC>(fails_le(mz, my), fails_le(my, mz), K.kaY);
    fail = bop3<kOrAnd>(fail, pXY, K.xXY);
    fail = bop3<kOrAnd>(fail, pXZ, K.xXZ);
    fail = bop3<kOrAnd>(fail, pYZ, K.xYZ);
    return fail;
}

// The same for a kernel without registers to spare (the triangle kernel, 96 VGPRs at 5 waves per SIMD): three pattern words per ray
// instead of six (x_ab is formed inside the operation that uses it) and the direction signs taken from the reciprocals in every trip
// (three arithmetic shifts) instead of living in three more registers: 5 more instructions per trip, 6 fewer registers.
// the three patterns in ONE word, ka_x | ka_y << 8 | ka_z << 16: nothing above a pattern's own 8 bits is ever looked at, so a shift
// by a constant (full rate) unpacks one
struct ExactPat3 { unsigned pack; };
__device__ __forceinline__ ExactPat3 exact_patterns3(unsigned sgnX, unsigned sgnY, unsigned sgnZ) {
    ExactPat3 p;
    p.pack = (0xAAu ^ (sgnX & 0xFFu)) | ((0xCCu ^ (sgnY & 0xFFu)) << 8) | ((0xF0u ^ (sgnZ & 0xFFu)) << 16);
    return p;
}
constexpr int kAndXor = 0x60;     // a & (b ^ c)
template <bool FOLD, bool BIASED>
__device__ __forceinline__ unsigned child_fail_mask_exactgrid3(float gx, float gy, float gz, float vs, float ox, float oy, float oz,
                                                               float ix, float iy, float iz, const ExactPat3& K,
                                                               int cx, int cy, int cz, float fh,
                                                               float clampLo = __uint_as_float(1u), float clampHi = __uint_as_float(0x7149f2c9u)) {
    const float sv = fh * vs;
    const unsigned sx = (unsigned)((int)__float_as_uint(ix) >> 31), sy = (unsigned)((int)__float_as_uint(iy) >> 31), sz = (unsigned)((int)__float_as_uint(iz) >> 31);
    float ex, mx, xx, ey, my, xy, ez, mz, xz;
    axis_emx_exact<BIASED>(gx, ox, ix, sx, cx, vs, sv, ex, mx, xx);
    axis_emx_exact<BIASED>(gy, oy, iy, sy, cy, vs, sv, ey, my, xy);
    axis_emx_exact<BIASED>(gz, oz, iz, sz, cz, vs, sv, ez, mz, xz);
    float Tn = max3f(ex, ey, ez), Tf = min3f(xx, xy, xz);
    if (FOLD) {
        asm("v_max_f32 %0, %1, %2" : "=v"(Tn) : "v"(Tn), "v"(clampLo));
        asm("v_min_f32 %0, %1, %2" : "=v"(Tf) : "v"(Tf), "v"(clampHi));
    }
    const unsigned kaX = K.pack, kaY = K.pack >> 8, kaZ = K.pack >> 16;          // bits above 7: don't care
    const unsigned selX = bop3<kSelC>(fails_le(Tn, mx), fails_le(mx, Tf), kaX);
    const unsigned selY = bop3<kSelC>(fails_le(Tn, my), fails_le(my, Tf), kaY);
    const unsigned selZ = bop3<kSelC>(fails_le(Tn, mz), fails_le(mz, Tf), kaZ);
    const unsigned pXY = bop3<kAndXor>(bop3<kSelC>(fails_le(my, mx), fails_le(mx, my), kaX), kaX, kaY);
    const unsigned pXZ = bop3<kAndXor>(bop3<kSelC>(fails_le(mz, mx), fails_le(mx, mz), kaX), kaX, kaZ);
    const unsigned pYZ = bop3<kAndXor>(bop3<kSelC>(fails_le(mz, my), fails_le(my, mz), kaY), kaY, kaZ);
    return bop3<kOr3>(selX, selY, selZ) | bop3<kOr3>(pXY, pXZ, pYZ);
}

// The same 8 verdicts with glm's NaN-faithful (y<x)?y:x min/max, one child at a time (waves that hold a ray with a
// non-finite reciprocal direction -- axis-parallel rays -- are rare: small code and few registers matter here, not
// speed).  Same float expressions as slab_exact on the child's box (S/RT:265-269).
__device__ __forceinline__ unsigned child_fail_mask_exact(float gx, float gy, float gz, float vs, float ox, float oy, float oz,
                                                       float ix, float iy, float iz, int cx, int cy, int cz, int half) {
    const float sv = (float)half * vs;
    unsigned fail = 0;
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        const float mnx = gx + (float)(cx + ((k & 1) ? half : 0)) * vs;
        const float mny = gy + (float)(cy + ((k & 2) ? half : 0)) * vs;
        const float mnz = gz + (float)(cz + ((k & 4) ? half : 0)) * vs;
        const float t1x = (mnx - ox) * ix, t1y = (mny - oy) * iy, t1z = (mnz - oz) * iz;
        const float t2x = ((mnx + sv) - ox) * ix, t2y = ((mny + sv) - oy) * iy, t2z = ((mnz + sv) - oz) * iz;
        const float tn = gmax(gmax(gmin(t1x, t2x), gmin(t1y, t2y)), gmin(t1z, t2z));
        const float tf = gmin(gmin(gmax(t1x, t2x), gmax(t1y, t2y)), gmax(t1z, t2z));
        const bool ok = (tn <= tf) && (tf > 0.0f) && !(tn >= 1e30f);
        fail |= ok ? 0u : (1u << k);
    }
    return fail;
}

// EXACT: 1 / 0 = the launcher knows whether the grid's planes are exact (the host's proof is per scene) and the loop holds only that form's
// child tests; -1 = decided by the frame's flag at run time (the persistent A/B kernel)
template <int MODE, int EXACT = -1>
__device__ __forceinline__ void trace_tile_lean(const RenderParams& P, const uint2* __restrict__ desc, float4* __restrict__ out,
                                                int* __restrict__ stepsOut, Counters* __restrict__ counters, uint2* stk,
                                                const int lane, const int slot) {
    unsigned long long tl0 = 0;
    if (MODE == kModeTimeline) tl0 = wall_clock64();

    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);

    const Geo G = geo_of(P);
    bool hit = false;
    int steps0 = 0;              // the start node's own pop (S/RT:254-270 with nodeIdx 0)
    Ray r;
    bool alive = false;
    const bool outsideRoot = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;
    bool guarded = false;        // this lane's ray needs the tFar > 0 / tNear < 1e30 folds of the child test
    // the start node: the root, unless a frustum update says otherwise (StartState; wave-uniform scalar loads)
    int startVisible = P.rootVisible, startShift = P.depth, startSize = P.rootSize, startX = 0, startY = 0, startZ = 0;
    unsigned startDesc = 0;
    bool startLeaf = false, startSolid = false;
    if (P.start) {
        const StartState st = *P.start;
        startVisible = st.visible; startShift = st.shift; startSize = 1 << st.shift; startX = st.x; startY = st.y; startZ = st.z;
        startDesc = st.desc; startLeaf = st.leaf != 0; startSolid = st.solid != 0;
    }
    const bool tileLive = tile_may_hit(P, tx, ty, slot, lane, tile);    // wave-uniform
    if (inImage && startVisible && tileLive) {
        steps0 = 1;
        if (!outsideRoot) {
            r = generate_ray_tab(P, px, py);
            float tNear, tFar, a0, a1, a2, a3, a4, a5;
            alive = slab_exact(G, r, startX, startY, startZ, startSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
            // The ray starts well outside the root box: its entry is at least 1/1024 of its exit away, leaves it below 1e29,
            // and lies more than 8 ulps of the largest coordinate involved -- as a ray parameter, on the steepest axis --
            // from the origin: a child's planes may differ from the root's by an ulp, which then cannot move a child's
            // parameters across zero.
            const float reach = 0x1p-20f * gmax(gmax(gmax(__builtin_fabsf(a0), __builtin_fabsf(a3)), __builtin_fabsf(r.ox)) * __builtin_fabsf(r.ix),
                                                gmax(gmax(gmax(__builtin_fabsf(a1), __builtin_fabsf(a4)), __builtin_fabsf(r.oy)) * __builtin_fabsf(r.iy),
                                                     gmax(gmax(__builtin_fabsf(a2), __builtin_fabsf(a5)), __builtin_fabsf(r.oz)) * __builtin_fabsf(r.iz)));
            guarded = alive && !(tNear > reach && tNear * 1024.0f >= tFar && tFar < 1e29f);
        }
    }
    const bool risky = alive && !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                                  __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));
    const bool anyRisky = __builtin_amdgcn_ballot_w64(risky) != 0ull;   // wave-uniform, fixed for the whole traversal
    const bool plainWave = __builtin_amdgcn_ballot_w64(guarded) == 0ull;   // wave-uniform: no ray of this wave needs the folds
    const bool exactGrid = EXACT < 0 ? P.exactGrid != 0 : EXACT != 0;
    // the exact form's v_fma_f32 takes the voxel size from a scalar register and so needs the origin in a vector one: kept there across
    // the loop (three registers) instead of being copied in front of every fma
    float gxv = G.gx, gyv = G.gy, gzv = G.gz;
    asm volatile("" : "+v"(gxv), "+v"(gyv), "+v"(gzv));

    // The loop is written for gfx950's VALU issue costs (tools/ubench/valu_rate2.hip, valu_rate3.hip: add / sub / mul f32,
    // add / sub u32, and / or / xor, shifts by an immediate and v_bitop3 issue every 2 cycles per SIMD; min / max, cvt,
    // cmp, cndmask, bfe, variable shifts, bcnt, ffbh and every other 3-operand form every 4): lane classes are merged
    // with bitwise selects on all-ones masks instead of v_cmp + v_cndmask, levels are indexed by the child-edge exponent
    // b = depth-1-level (what the arithmetic needs) and the stack write is unconditional (a lane owns its LDS column;
    // a finished lane's column is never read again except by its own identity-(2) walk, which a hit keeps valid).
    unsigned cur = startDesc;
    int cx = (int)(kCoordBias | (unsigned)startX), cy = (int)(kCoordBias | (unsigned)startY), cz = (int)(kCoordBias | (unsigned)startZ);   // node position, biased (child_axis_terms): the bit operations below never touch the bias
    int bpos = startShift - 1;                                          // log2 of the edge of the current node's children
    unsigned lvlPending = 0;                                            // bit b: the entry of exponent b still has unpopped candidates
    const unsigned sentinel = 1u << P.depth;                            // entry `depth` is a dummy: what "nothing pending" reads
    int S = 0;                                                          // identity (1): children pushed so far
    const int capBound = kMaxTraversalSteps + 7 * P.depth;              // S >= capBound: every later hit lies beyond the cap
    const unsigned sgnX = (unsigned)((int)__float_as_uint(r.ix) >> 31), sgnY = (unsigned)((int)__float_as_uint(r.iy) >> 31),
                   sgnZ = (unsigned)((int)__float_as_uint(r.iz) >> 31);          // all ones: the reciprocal direction is negative
    const ExactPat K = exact_patterns(sgnX, sgnY, sgnZ);                // exact grids: which children a failed comparison strikes (child_fail_mask_exactgrid)
    const char* descBytes = reinterpret_cast<const char*>(desc);
    int trips = 0;                                                      // wave-uniform trip count = this tile's cost
    if (startLeaf) { hit = alive && startSolid; alive = false; }        // a terminal start node (culled-root edge only): one pop, S/RT:277-288
    while (alive) {
        trips++;
        const uint2 d = *reinterpret_cast<const uint2*>(descBytes + (cur << 3));
        // the resume entry depends only on lvlPending: fetch it under the descriptor load and the slab math
        const int Lb = __builtin_ctz(lvlPending | sentinel);
        const uint2 e = stk[Lb * kWave];
        const float fh = __uint_as_float((unsigned)(bpos + 127) << 23);        // (float)(1 << bpos), exact
        unsigned fail8;
        if (anyRisky) fail8 = child_fail_mask_exact(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz,
                                                    r.ix, r.iy, r.iz, cx & 0x7fffff, cy & 0x7fffff, cz & 0x7fffff, 1 << bpos);
        else if (exactGrid) {           // wave-uniform (a kernel argument): the host has proven the grid's planes exact
            if (plainWave) fail8 = child_fail_mask_exactgrid<false, true>(gxv, gyv, gzv, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                                          sgnX, sgnY, sgnZ, K, cx, cy, cz, fh);
            else fail8 = child_fail_mask_exactgrid<true, true>(gxv, gyv, gzv, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                               sgnX, sgnY, sgnZ, K, cx, cy, cz, fh);
        }
        else if (plainWave) fail8 = child_fail_mask_fast<false, true>(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                                      sgnX, sgnY, sgnZ, cx, cy, cz, fh);
        else fail8 = child_fail_mask_fast<true, true>(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                      sgnX, sgnY, sgnZ, cx, cy, cz, fh);
        const unsigned vm0 = (d.x >> 16) & 0xffu;
        S += __builtin_popcount(vm0);
        // children that do more than count a pop: visible internal ones and visible solid leaves that pass the slab test
        const unsigned cand = bop3<kAndAndNot>(d.x | (d.x >> 8), vm0, fail8);
        const unsigned noWork = (unsigned)(((int)cand - 1) >> 31);       // all ones: nothing to pop here
        const bool dead = (cand | lvlPending) == 0 || S >= capBound;
        // merge with "resume at the entry of the smallest exponent (= deepest level) that still has candidates"
        const unsigned W = bop3<kSelC>(bop3<kAndOr>(d.x, 0x00ffff00u, cand), e.x, noWork);
        const unsigned base = bop3<kSelC>(d.y, e.y, noWork);
        const int bpos2 = (int)bop3<kSelC>((unsigned)bpos, (unsigned)Lb, noWork);
        // pop its next candidate: the highest one (children pop 7..0)
        const int j = 31 - __builtin_clz(W & 0xffu);                     // W & 0xff != 0 unless the lane is dead
        const unsigned bitj = 0x80000000u >> (31 - j);                   // = 1u << j as a RIGHT shift: variable right shifts issue at full rate on gfx950, left shifts at half (tools/ubench/valu_rate4.hip)
        const unsigned imv = W >> 8;                                     // bits 0..7: internal mask (unmasked); above: visible mask
        const bool solid = (imv & bitj) == 0;
        hit = !dead && solid;                                            // S/RT:278-288 (cap applied after the loop)
        const unsigned Wn = W ^ bitj;
        stk[bpos2 * kWave] = make_uint2(Wn, base);
        const unsigned hl = 1u << bpos2;
        const unsigned noneLeft = (unsigned)(((int)(Wn & 0xffu) - 1) >> 31);
        lvlPending = bop3<kReplace>(lvlPending, hl, noneLeft);           // set the bit iff candidates are left at this level
        cur = base + (unsigned)__builtin_popcount(imv & (bitj - 1u));
        const unsigned keep = 0u - (hl + hl);
        const unsigned sj = (unsigned)j << bpos2;
        cx = (int)bop3<kOrAnd>((unsigned)cx & keep, sj, hl);             // the popped child (kept on a hit)
        cy = (int)bop3<kOrAnd>((unsigned)cy & keep, sj >> 1, hl);
        cz = (int)bop3<kOrAnd>((unsigned)cz & keep, sj >> 2, hl);
        bpos = bpos2 - 1;
        alive = !dead && !solid;
    }
    // the tile's cost for the launch order: the trips of its BUSIEST ray (a lane keeps the count of its own last trip: lane 0's
    // alone -- what was recorded until round 3 -- underrates every tile whose costly rays are not in its top-left corner)
    if (P.tileCost || MODE == kModeTimeline) {
        trips = __builtin_amdgcn_readlane(wave_scan_max_nonneg(trips), kWave - 1);
        // -1: no trip, but the occupancy mask said geometry may project here (a tile the mask does not cover recorded its cost when it
        // looked the mask up: see tile_may_hit)
        if (P.tileCost && lane == 0 && ty < P.tilesY && tileLive) P.tileCost[tile] = (trips == 0 && P.tileMask) ? -1 : trips;
    }
    const int leafShift = bpos + 1;                                      // on a hit: log2 of the leaf's edge
    cx &= 0x7fffff; cy &= 0x7fffff; cz &= 0x7fffff;                      // plain coordinates for the epilogue

    // identity (2): pops at the accepted leaf.  Needed for every hit when steps are reported, else only when the
    // upper bound 1 + S does not already clear the cap.
    int steps = steps0 + S;
    if (hit && (MODE == kModeSteps || steps > kMaxTraversalSteps)) {
        int R = 0;
        for (int b = startShift - 1; b > bpos; b--) {                    // the entries of the leaf's ancestors (from the start node down)
            const unsigned w = stk[b * kWave].x;
            const unsigned jl = ((cx >> b) & 1) | (((cy >> b) & 1) << 1) | (((cz >> b) & 1) << 2);
            R += __builtin_popcount(__builtin_amdgcn_ubfe(w, 16, 8) & ((1u << jl) - 1u));
        }
        steps -= R;
        if (steps > kMaxTraversalSteps) hit = false;                     // S/RT:254: the loop ended before this pop
    }
    if (steps > kMaxTraversalSteps) steps = kMaxTraversalSteps;

    const bool mine = valid && !(P.skipOutside && outsideRoot);     // pixels outside the root rectangle belong to the fill duty
    if (MODE == kModeShade) {
        if (mine) __builtin_nontemporal_store(hit ? shade_term(P, G, r, cx, cy, cz, 1 << leafShift) : kShadeMiss, reinterpret_cast<float*>(out) + (size_t)ly * P.W + px);
        fill_outside<MODE>(P, out, lane, slot);
    } else if (MODE == kModeColor || MODE == kModeTimeline) {
        if (mine) {
            float4 color = make_float4(0.f, 0.f, 0.f, 1.f);
            if (hit) color = shade_color(shade_term(P, G, r, cx, cy, cz, 1 << leafShift));
            store_pixel(out + (size_t)ly * P.W + px, color);
        }
        fill_outside<MODE>(P, out, lane, slot);
        if (MODE == kModeTimeline) {
            const int act = __builtin_popcountll(__builtin_amdgcn_ballot_w64(steps0 + S > 1));
            if (lane == 0 && ty < P.tilesY) {
                const unsigned long long tl1 = wall_clock64();
                unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
                unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
                int* rec = stepsOut + (size_t)tile * 8;
                rec[0] = (int)(tl0 & 0xffffffffu); rec[1] = (int)(tl0 >> 32);
                rec[2] = (int)(tl1 & 0xffffffffu); rec[3] = (int)(tl1 >> 32);
                rec[4] = trips; rec[5] = (int)hwid; rec[6] = (int)xcc; rec[7] = act | (slot << 8);      // act <= 64; the launch slot above it
            }
        }
    } else {
        if (inImage) stepsOut[(size_t)py * P.W + px] = hit ? steps : -steps;
        wave_accumulate(counters, steps, hit, inImage);
    }
}

// ================================================================ closest-hit kernel on the descriptor tree
// k_trace_closest's traversal (the reference's earlier shader, S/RT:63-138) for canonical trees: same pop order -- an internal
// node's children 7 .. 0, a child's whole subtree before its next sibling -- and the same decisions at every pop (the child's own
// slab test and `tNear >= closestT` with the closestT of THAT moment, S/RT:87-92), so the winner is the reference's in every float
// corner; but 8 bytes per visited internal node instead of 60 per popped node, the stack in LDS ([level][lane]: pending children
// | internal mask, first internal child), and empty leaves -- popped upstream to no effect -- never looked at.  One loop trip =
// one popped child that can matter.  Pops: every entered internal node pushes 8 children and, without an early exit, every one
// is popped: 1 + 8 x entered (frustum culling removes children: such frames take k_trace_closest over the compacted array).
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_closest_lean(RenderParams P, const uint2* __restrict__ desc, float4* __restrict__ out,
                                                          Counters* __restrict__ counters) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;    // entry b = stk[b * 64]: the node whose children have edge 2^b
    const int tile = blockIdx.x * (kBlock / kWave) + wave;
    const int tx = tile % P.tilesX, ty = tile / P.tilesX;
    const int px = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);
    const Geo G = geo_of(P);

    bool hit = false, active = false, enter = false;
    long long entered = 0;
    int steps0 = 0;
    float closestT = 1e30f;                                  // S/RT:66
    int bx = 0, by = 0, bz = 0, bs = 0;                      // the best leaf so far
    Ray r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
    if (inImage && P.rootVisible) {
        steps0 = 1;                                          // the root's own pop
        r = generate_ray_tab(P, px, py);
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        active = enter = slab_exact(G, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= closestT);
    }
    unsigned cur = 0, lvlPending = 0;
    int cx = 0, cy = 0, cz = 0, bpos = P.depth - 1;          // the node to enter: position, log2 of its children's edge
    while (active) {
        if (enter) {                                         // S/RT:125-129: the node's children are pushed -- those that can matter, as a mask
            entered++;
            const uint2 d = desc[cur];
            const unsigned fail8 = child_fail_mask_exact(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, cx, cy, cz, 1 << bpos);
            const unsigned im = (d.x >> 8) & 0xffu;
            const unsigned cand = ((d.x | im) & 0xffu) & ~fail8;     // solid leaves and internal children whose box the ray meets
            stk[bpos * kWave] = make_uint2(cand | (im << 8), d.y);
            lvlPending = cand ? (lvlPending | (1u << bpos)) : (lvlPending & ~(1u << bpos));
            enter = false;
        }
        if (lvlPending == 0) break;                          // S/RT:75: the stack is empty
        const int Lb = __builtin_ctz(lvlPending);            // the deepest node with children still on the stack: LIFO
        const uint2 e = stk[Lb * kWave];
        const int j = 31 - __builtin_clz(e.x & 0xffu);      // children pop 7 .. 0
        const unsigned bit = 1u << j, left = e.x ^ bit;
        stk[Lb * kWave].x = left;
        if ((left & 0xffu) == 0) lvlPending &= ~(1u << Lb);
        const int h = 1 << Lb, keep = ~(2 * h - 1);
        const int chx = (cx & keep) + ((j & 1) ? h : 0), chy = (cy & keep) + ((j & 2) ? h : 0), chz = (cz & keep) + ((j & 4) ? h : 0);
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        const bool pass = slab_exact(G, r, chx, chy, chz, h, tNear, tFar, a0, a1, a2, a3, a4, a5);     // S/RT:87-88, at THIS pop
        if (pass && !(tNear >= closestT)) {                  // S/RT:91-92, with the closestT of this moment
            const unsigned im = (e.x >> 8) & 0xffu;
            if (im & bit) {                                  // an internal child: its subtree before the next sibling
                cur = e.y + (unsigned)__builtin_popcount(im & (bit - 1u));
                cx = chx; cy = chy; cz = chz; bpos = Lb - 1; enter = true;
            } else {                                         // a solid leaf, S/RT:96-104 (== :110-118): strictly nearer replaces, no break
                const float tHit = gmax(0.0f, tNear);
                if (tHit < closestT && tHit <= tFar) { closestT = tHit; hit = true; bx = chx; by = chy; bz = chz; bs = h; }
            }
        }
    }
    const float shade = hit ? shade_term(P, G, r, bx, by, bz, bs) : kShadeMiss;
    if (valid) out[(size_t)ly * P.W + px] = shade_color(shade);
    if (MODE == kModeSteps) {
        unsigned long long pops = inImage ? (unsigned long long)(steps0 + 8 * entered) : 0ull, hits = (inImage && hit) ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { pops += __shfl_down(pops, off); hits += __shfl_down(hits, off); }
        if (lane == 0) { atomicAdd(&counters->pops, pops); atomicAdd(&counters->hits, hits); }
    }
}

// The same winner, near children first.  The exhaustive traversal above ends with the leaf of least tHit among those whose every
// ancestor passes its own slab test (S/RT:87-88) and `tHit <= tFar`, ties to the leaf popped first: a subtree is cut only when
// its tNear >= closestT, a child's tNear is never below its parent's (the planes nest and every float step is monotone; a NaN
// axis the parent's min/max swallow only lowers the parent's tNear), so nothing cut could have replaced the winner.  That set
// and that order do not depend on the order of the walk: this kernel walks children in ray-sign order (child j ^ flip ascending),
// finds a near leaf early, cuts on `tNear > best` (a tie must still be looked at) and on tNear >= 1e30 (closestT's start, S/RT:66),
// and resolves equal tHit by the pop order of the LIFO walk: children pop 7 .. 0, so at the highest level where two leaves' paths
// part the greater child index pops first.  No pop count comes out of it (kModeSteps launches k_closest_lean).
__device__ __forceinline__ bool pops_before(int ax, int ay, int az, int bx, int by, int bz) {
    const unsigned m = (unsigned)((ax ^ bx) | (ay ^ by) | (az ^ bz));
    if (m == 0) return false;
    const int top = 31 - __builtin_clz(m);
    const int ja = ((ax >> top) & 1) | (((ay >> top) & 1) << 1) | (((az >> top) & 1) << 2);
    const int jb = ((bx >> top) & 1) | (((by >> top) & 1) << 1) | (((bz >> top) & 1) << 2);
#ifdef RTO_TEST_WRONG_TIE      // tools/build_variants.sh only: the tie test must fail with this
    return ja < jb;
#else
    return ja > jb;
#endif
}
__device__ __forceinline__ unsigned flip_children(unsigned m, unsigned flip) {      // bit i of the result = bit (i ^ flip) of m
    if (flip & 1u) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (flip & 2u) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (flip & 4u) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    return m;
}
__global__ __launch_bounds__(kBlock) void k_closest_near_first(RenderParams P, const uint2* __restrict__ desc, float4* __restrict__ out) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // launch geometry of the lean kernels: the first workgroups build the occupancy mask (a tile no solid leaf projects into holds no
    // hit under this rule either), waves only for the tiles of the solid geometry's screen rectangle in the stream's launch order,
    // wide stores for the rest
    if ((int)blockIdx.x < P.maskBlocks) { mask_block(P, (int)blockIdx.x, reinterpret_cast<unsigned*>(lds_stack), P.maskLdsBytes >> 2); return; }
    const int slot = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - P.maskBlocks) * (kBlock / kWave) + wave);
    if (slot >= P.launchWaves) return;
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;
    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);
    const bool outside = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;     // the solid leaves' rectangle: no hit outside it
    const bool tileLive = tile_may_hit(P, tx, ty, slot, lane, tile);                             // wave-uniform; a tile outside the mask records its cost there
    const Geo G = geo_of(P);

    bool hit = false, active = false, enter = false;
    float best = 1e30f;
    int bx = 0, by = 0, bz = 0, bs = 0;
    unsigned flip = 0;
    Ray r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
    if (inImage && P.rootVisible && !outside && tileLive) {
        r = generate_ray_tab(P, px, py);
        flip = (r.dx < 0.0f ? 1u : 0u) | (r.dy < 0.0f ? 2u : 0u) | (r.dz < 0.0f ? 4u : 0u);
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        active = enter = slab_exact(G, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
    }
    // as in trace_tile_lean: a wave without a non-finite ray takes the 8 verdicts of a node from the select form (no NaN can occur),
    // and here the cut `tNear > best` rides in its upper clamp (a child passes when max(tNear, eps) <= min(tFar, clamp))
    const bool risky = active && !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                                   __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));
    const bool anyRisky = __builtin_amdgcn_ballot_w64(risky) != 0ull;
    const unsigned sgnX = (unsigned)((int)__float_as_uint(r.ix) >> 31), sgnY = (unsigned)((int)__float_as_uint(r.iy) >> 31),
                   sgnZ = (unsigned)((int)__float_as_uint(r.iz) >> 31);
    const float kBelow1e30 = __uint_as_float(0x7149f2c9u);
    unsigned cur = 0, lvlPending = 0;
    int cx = 0, cy = 0, cz = 0, bpos = P.depth - 1;
    int entered = 0;                                        // internal nodes this lane entered: the tile's cost is its busiest lane's
    while (active) {
        if (enter) {
            entered++;
            const uint2 d = desc[cur];
            unsigned fail8;
            if (anyRisky) fail8 = child_fail_mask_exact(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, cx, cy, cz, 1 << bpos);
            else fail8 = child_fail_mask_fast<true, false>(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, sgnX, sgnY, sgnZ, cx, cy, cz,
                                                           (float)(1 << bpos), __uint_as_float(1u), gmax(__uint_as_float(1u), gmin(best, kBelow1e30)));   // never below eps: best may be 0 and a tie at 0 still counts; the pop re-tests exactly
            const unsigned im = (d.x >> 8) & 0xffu;
            const unsigned cand = flip_children(((d.x | im) & 0xffu) & ~fail8, flip);      // bit i: child i ^ flip
            stk[bpos * kWave] = make_uint2(cand | (im << 8), d.y);
            lvlPending = cand ? (lvlPending | (1u << bpos)) : (lvlPending & ~(1u << bpos));
            enter = false;
        }
        if (lvlPending == 0) break;
        const int Lb = __builtin_ctz(lvlPending);
        const uint2 e = stk[Lb * kWave];
        const int i = __builtin_ctz(e.x & 0xffu);           // the nearest child left: the ray's own octant first
        const unsigned left = e.x ^ (1u << i);
        stk[Lb * kWave].x = left;
        if ((left & 0xffu) == 0) lvlPending &= ~(1u << Lb);
        const int j = i ^ (int)flip;
        const unsigned bit = 1u << j;
        const int h = 1 << Lb, keep = ~(2 * h - 1);
        const int chx = (cx & keep) + ((j & 1) ? h : 0), chy = (cy & keep) + ((j & 2) ? h : 0), chz = (cz & keep) + ((j & 4) ? h : 0);
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        const bool pass = slab_exact(G, r, chx, chy, chz, h, tNear, tFar, a0, a1, a2, a3, a4, a5);
        if (pass && !(tNear >= 1e30f) && !(tNear > best)) {
            const unsigned im = (e.x >> 8) & 0xffu;
            if (im & bit) {
                cur = e.y + (unsigned)__builtin_popcount(im & (bit - 1u));
                cx = chx; cy = chy; cz = chz; bpos = Lb - 1; enter = true;
            } else {
                const float tHit = gmax(0.0f, tNear);
                // three plain booleans, not `tHit < best || (hit && tHit == best && pops_before(..))`: ROCm 7.2's structurizer turned the
                // short-circuit form, inside this divergent loop, into exec-mask bookkeeping that lost the update for a lane whose first
                // candidate was a degenerate touch (tNear == tFar) -- one pixel of the Calgary golden frame, found when the launch geometry
                // changed which rays share a wave (docs/LAB_NOTES.md)
                const bool nearer = tHit < best;
                const bool tie = tHit == best;             // only after a hit: best starts at 1e30 and tHit < 1e30 (the cut above); no lane-mask boolean in the update
                const bool first = pops_before(chx, chy, chz, bx, by, bz);
                if (tHit <= tFar && (nearer || (tie && first))) {
                    best = tHit; hit = true; bx = chx; by = chy; bz = chz; bs = h;
                }
            }
        }
    }
    if (P.tileCost && tileLive) {
        const int cost = __builtin_amdgcn_readlane(wave_scan_max_nonneg(entered), kWave - 1);
        if (lane == 0 && ty < P.tilesY) P.tileCost[tile] = (cost == 0 && P.tileMask) ? -1 : (cost + 3) >> 2;
    }
    const float shade = hit ? shade_term(P, G, r, bx, by, bz, bs) : kShadeMiss;
    if (valid && !(P.skipOutside && outside)) store_pixel(out + (size_t)ly * P.W + px, shade_color(shade));
    fill_outside<kModeColor>(P, out, lane, slot);
}

#ifndef RTO_LEAN_WAVES
#define RTO_LEAN_WAVES 6
#endif
template <int MODE, int EXACT>
__global__ __launch_bounds__(kBlock, RTO_LEAN_WAVES) void k_trace_lean(RenderParams P, const uint2* __restrict__ desc,
                                                           float4* __restrict__ out, int* __restrict__ stepsOut,
                                                           Counters* __restrict__ counters) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;    // entry b = stk[b * 64], b in [0, depth]
    if ((int)blockIdx.x < P.maskBlocks) { mask_block(P, (int)blockIdx.x, reinterpret_cast<unsigned*>(lds_stack), P.maskLdsBytes >> 2); return; }        // the first workgroups build the occupancy mask
    const int slot = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - P.maskBlocks) * (int)(blockDim.x >> 6) + wave);
    if (slot >= P.launchWaves) return;
    trace_tile_lean<MODE, EXACT>(P, desc, out, stepsOut, counters, stk, lane, slot);
}

// Several frames in ONE launch (rto_render_batch_device, and every rank of the multi-GPU split: its parts of a batch of
// frames).  A frame's kernel cannot finish before its deepest tile has walked its ~140 dependent node visits (~36 us at
// config 2, whatever share of the frame the launch covers), and while that tile walks most wave slots are idle; waves of
// other frames fill them.  Launch slots are dealt round-robin over the frames (wave g -> frame g % n, slot g / n), so the
// costliest tiles of every frame start first.  Each frame brings its own RenderParams (camera, rectangle, launch order).
constexpr int kMaxBatch = 8;
struct RenderBatch {
    RenderParams P[kMaxBatch];
    float4* out[kMaxBatch];
    int n;
};
template <int MODE, int EXACT>
__global__ __launch_bounds__(kBlock, RTO_LEAN_WAVES) void k_trace_lean_batch(RenderBatch B, const uint2* __restrict__ desc) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int mb = B.P[0].maskBlocks;                          // mask workgroups per frame (every frame of a batch: the same octree)
    if ((int)blockIdx.x < mb * B.n) { const int fm = (int)blockIdx.x / mb; mask_block(B.P[fm], (int)blockIdx.x - fm * mb, reinterpret_cast<unsigned*>(lds_stack), B.P[fm].maskLdsBytes >> 2); return; }
    const int g = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - mb * B.n) * (int)(blockDim.x >> 6) + wave);
    const int slot = g / B.n, f = g - slot * B.n;
    const RenderParams& P = B.P[f];
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;
    if (slot >= P.launchWaves) return;
    trace_tile_lean<MODE, EXACT>(P, desc, B.out[f], nullptr, nullptr, stk, lane, slot);
}

// Persistent-threads form of the default kernel (RTO_KERNEL_PACKED_PERSISTENT): the grid only fills the machine, every
// wave renders its first tile and then keeps taking launch slots from a global counter until none is left.  The host
// zeroes the counter with a memset node in front of every launch, so each launch is self-contained (safe to capture
// into a HIP graph and to replay any number of times).
// 5 waves per SIMD, not the 6 of k_trace_lean: the tile function inside a loop keeps the slot bookkeeping live across it
// and at 6 (80 VGPRs) two of them were spilled to scratch in the hot loop (12 bytes); at 5 (<= 96) nothing is.
#ifndef RTO_PERSIST_WAVES
#define RTO_PERSIST_WAVES 5
#endif
template <int MODE>
__global__ __launch_bounds__(kBlock, RTO_PERSIST_WAVES) void k_trace_lean_persistent(RenderParams P, const uint2* __restrict__ desc,
                                                           float4* __restrict__ out, int* __restrict__ stepsOut,
                                                           Counters* __restrict__ counters, int* __restrict__ queue) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;
    const int tiles = P.launchWaves;
    const int firstFree = gridDim.x * (kBlock / kWave);
    int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    int left = 1;                                              // slots of the current chunk still to render
    while (slot < tiles) {                                     // wave-uniform; ends once the counter has run past the last slot
        trace_tile_lean<MODE>(P, desc, out, stepsOut, counters, stk, lane, slot);
        slot++;
        if (--left == 0) {
            // same-address atomics serialise (about 10 ns each): one per tile would cost 0.3 ms per 1080p frame,
            // so a wave takes kPersistChunk consecutive slots at a time
            int next = 0;
            if (lane == 0) next = atomicAdd(queue, kPersistChunk);
            slot = firstFree + __builtin_amdgcn_readfirstlane(next);
            left = kPersistChunk;
        }
    }
}

// ================================================================ frustum culling (N3)
// GPU form of the CPU loop + compaction of renderSceneComputeWithCulling
// (S/RT:743-802) with Frustum::testAABB (453-skeleton/Frustum.cpp:52-93).
struct CullParams {
    float planes[24];      // LEFT, RIGHT, TOP, BOTTOM, NEAR, FAR; normalised (Frustum.cpp:5-48), from the host
    float gridMin[3];
    float voxelSize;
    float margin;          // 150.0f at S/RT:755
};

// testAABB(...) != -1.  Only the positive-vertex test can return -1 (Frustum.cpp:68-78).
__device__ __forceinline__ bool node_visible(const CullParams& C, int x, int y, int z, int size) {
    float mn[3] = { C.gridMin[0] + (float)x * C.voxelSize, C.gridMin[1] + (float)y * C.voxelSize,
                    C.gridMin[2] + (float)z * C.voxelSize };                     // S/RT:747-751
    float ext = (float)size * C.voxelSize;                                      // S/RT:752
    float emn[3], emx[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { emn[a] = mn[a] - C.margin; emx[a] = (mn[a] + ext) + C.margin; }
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float* pl = C.planes + i * 4;
        float px = pl[0] > 0 ? emx[0] : emn[0];
        float py = pl[1] > 0 ? emx[1] : emn[1];
        float pz = pl[2] > 0 ? emx[2] : emn[2];
        float tx = pl[0] * px, ty = pl[1] * py, tz = pl[2] * pz;
        if ((tx + ty + tz) + pl[3] < 0) inside = false;
    }
    return inside;
}

// one thread per node: visibility flag + per-block visible count
__global__ __launch_bounds__(kBlock) void k_cull_flags(CullParams C, const rto_node* __restrict__ nodes, int64_t n,
                                                        uint8_t* __restrict__ vis, int* __restrict__ blockCount, int64_t* __restrict__ rootVisible) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool v = false;
    if (i < n) {
        const rto_node* nd = nodes + i;
        v = node_visible(C, nd->x, nd->y, nd->z, nd->size);
        vis[i] = v ? 1 : 0;
        if (i == 0) *rootVisible = v ? 1 : 0;          // read back together with the count of visible nodes
    }
    int cnt = __syncthreads_count(v ? 1 : 0);
    if (threadIdx.x == 0) blockCount[blockIdx.x] = cnt;
}

// single block: exclusive scan of the per-block counts; total -> *visibleCount
__global__ __launch_bounds__(1024) void k_scan_block_counts(const int* __restrict__ blockCount, int nb,
                                                             int* __restrict__ blockBase, int64_t* __restrict__ visibleCount) {
    __shared__ int partial[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, nb);
    int s = 0;
    for (int i = lo; i < hi; i++) s += blockCount[i];
    partial[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        int v = (t >= off) ? partial[t - off] : 0;
        __syncthreads();
        partial[t] += v;
        __syncthreads();
    }
    int run = partial[t] - s;   // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; i++) { blockBase[i] = run; run += blockCount[i]; }
    if (t == 1023) *visibleCount = partial[1023];
}

// old index -> new index (-1 if culled): S/RT:765-772
__global__ __launch_bounds__(kBlock) void k_cull_remap(const uint8_t* __restrict__ vis, int64_t n,
                                                        const int* __restrict__ blockBase, int* __restrict__ remap) {
    __shared__ int waveTotal[kBlock / kWave];
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool v = (i < n) && vis[i];
    const unsigned long long b = __builtin_amdgcn_ballot_w64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __builtin_popcountll(b & ((1ull << lane) - 1ull));
    if (lane == 0) waveTotal[wave] = __builtin_popcountll(b);
    __syncthreads();
    int base = blockBase[blockIdx.x];
    for (int w = 0; w < wave; w++) base += waveTotal[w];
    if (i < n) remap[i] = v ? base + before : -1;
}

// S/RT:778-802: copy visible nodes, remap children of non-leaf nodes (culled child -> -1)
__global__ __launch_bounds__(kBlock) void k_cull_compact(const rto_node* __restrict__ nodes, int64_t n,
                                                          const int* __restrict__ remap, rto_node* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int dst = remap[i];
    if (dst < 0) return;
    rto_node nd = nodes[i];
    if (!nd.isLeaf) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            int oc = nd.child[c];
            nd.child[c] = (oc >= 0 && oc < n) ? remap[oc] : -1;
        }
    }
    out[dst] = nd;
}

// packed tree: visibility of the 8 children of every internal node -> descriptor bits 16..23
__global__ __launch_bounds__(kBlock) void k_desc_vismask(const uint8_t* __restrict__ vis, const int* __restrict__ descFirstChild,
                                                          int64_t nInternal, uint2* __restrict__ desc) {
    int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= nInternal) return;
    const int c0 = descFirstChild[d];
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) m |= vis[c0 + k] ? (1u << k) : 0u;
    desc[d].x = (desc[d].x & 0xff00ffffu) | (m << 16);
}

__global__ __launch_bounds__(kBlock) void k_desc_visall(int64_t nInternal, uint2* __restrict__ desc) {
    int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d < nInternal) desc[d].x |= 0xff0000u;
}

// position and size of every internal node of a canonical tree (descriptor order): child 0 shares its parent's origin
__global__ __launch_bounds__(kBlock) void k_desc_pos(const rto_node* __restrict__ nodes, const int* __restrict__ descFirstChild, int64_t nInternal,
                                                      int4* __restrict__ descPos) {
    int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= nInternal) return;
    const rto_node* c0 = nodes + descFirstChild[d];
    descPos[d] = make_int4(c0->x, c0->y, c0->z, c0->size * 2);
}

// what k_cull_desc leaves when every node is visible (rto_update_frustum's host-side proof): traversals start at the root
__global__ void k_start_at_root(StartState* __restrict__ start, int depth, long long visibleCount) {
    StartState st;
    st.visible = 1; st.desc = 0; st.x = st.y = st.z = 0; st.shift = depth; st.leaf = 0; st.solid = 0;
    st.rootVisible = 1; st.firstVisible = 0;
    st.visibleCount = visibleCount;
    st.ticket = 0;
    *start = st;
}

// The whole frustum update of a canonical tree in ONE launch, nothing read back (rto_update_frustum): thread d tests the 8
// children of internal node d (every node but the root is the child of exactly one internal node; thread 0 adds the root) with
// the same float operations as k_cull_flags on the same integer coordinates -- positions come from descPos (16 B per internal
// node, 0.75 MB at config 2) instead of the 60-byte records (22.5 MB) --, writes their flags (`vis`: the visibility map
// octreeRaySkip and the on-demand compaction read) and the visibility byte of its descriptor.  The block that finishes last
// (ticket) reduces the blocks' counts / first visible indices and records where traversals start (StartState): the root, or
// -- root culled, descendants visible: S/RT:765-812 -- the first visible node in BFS order, whose descriptor it finds by
// walking down from the root along the node's coordinates.  Partials cross XCDs: agent-scope stores / loads around the ticket.
__global__ __launch_bounds__(kBlock) void k_cull_desc(CullParams C, const int4* __restrict__ descPos, const int* __restrict__ descFirstChild,
                                                       const rto_node* __restrict__ nodes, int64_t nInternal, int depth,
                                                       uint2* __restrict__ desc, uint8_t* __restrict__ vis,
                                                       int* __restrict__ blockCount, int* __restrict__ blockFirst, StartState* __restrict__ start) {
    __shared__ int redCount[kBlock / kWave], redFirst[kBlock / kWave];
    __shared__ int isLast, rootSeen;
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int cnt = 0, first = 0x7fffffff;
    // Every block looks at the root itself (one box against six planes): with the root visible -- every update but the
    // float-rounding edge below and the "everything culled" case -- traversals start at the root, nobody needs the first
    // visible node, and the survivors are only counted when somebody asks (sync_cull_state): no partials, no ticket, no
    // fences; the kernel is then a plain streaming pass (8.4 -> ~3 us at config 2: the two fences and the same-address atomics
    // were most of it).
    if (threadIdx.x == 0) { const int4 r = descPos[0]; rootSeen = node_visible(C, r.x, r.y, r.z, r.w) ? 1 : 0; }
    if (d < nInternal) {
        const int4 p = descPos[d];
        const int half = p.w >> 1;
        const int c0 = descFirstChild[d];
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool v = node_visible(C, p.x + ((k & 1) ? half : 0), p.y + ((k & 2) ? half : 0), p.z + ((k & 4) ? half : 0), half);
            m |= v ? (1u << k) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) vis[c0 + k] = (uint8_t)((m >> k) & 1u);      // c0 = 1 (mod 8): byte stores, contiguous across the threads of a wave
        desc[d].x = (desc[d].x & 0xff00ffffu) | (m << 16);
        cnt = __builtin_popcount(m);
        if (m) first = c0 + __builtin_ctz(m);
        if (d == 0) {                                        // the root itself
            const bool v = node_visible(C, p.x, p.y, p.z, p.w);
            vis[0] = v ? 1 : 0;
            cnt += v ? 1 : 0;
            if (v) first = 0;
        }
    }
    __syncthreads();
    if (rootSeen) {                                              // workgroup-uniform
        if (d == 0) {
            StartState st;
            st.visible = 1; st.desc = 0; st.x = st.y = st.z = 0; st.shift = depth; st.leaf = 0; st.solid = 0;
            st.rootVisible = 1; st.firstVisible = 0;
            st.visibleCount = -1;                                // not counted: the flags in `vis` are, when somebody asks
            st.ticket = 0;
            *start = st;
        }
        return;
    }
    for (int off = 32; off > 0; off >>= 1) { cnt += __shfl_down(cnt, off); first = min(first, __shfl_down(first, off)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { redCount[wave] = cnt; redFirst[wave] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0, f = 0x7fffffff;
        for (int w = 0; w < kBlock / kWave; w++) { c += redCount[w]; f = min(f, redFirst[w]); }
        __hip_atomic_store(&blockCount[blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&blockFirst[blockIdx.x], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                     // release: the partials before the ticket
        drain_vector_memory();                               // ... and acknowledged before it (the wait the compiler may drop)
        const unsigned t = atomicAdd(&start->ticket, 1u);
        isLast = t == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!isLast) return;
    __threadfence();                                         // acquire: every block's partials (sc1 stores by the signalling lane, sc1 loads below)
    drain_vector_memory();                                   // the invalidate has completed before the first load
    long long total = 0;
    int f = 0x7fffffff;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kBlock) {
        total += __hip_atomic_load(&blockCount[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = min(f, __hip_atomic_load(&blockFirst[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    for (int off = 32; off > 0; off >>= 1) { total += __shfl_down(total, off); f = min(f, __shfl_down(f, off)); }
    __shared__ long long redTotal[kBlock / kWave];
    if (lane == 0) { redTotal[wave] = total; redFirst[wave] = f; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    total = 0; f = 0x7fffffff;
    for (int w = 0; w < kBlock / kWave; w++) { total += redTotal[w]; f = min(f, redFirst[w]); }
    StartState st;
    st.visible = total > 0 ? 1 : 0;
    st.desc = 0; st.x = st.y = st.z = 0; st.shift = depth; st.leaf = 0; st.solid = 0;
    st.rootVisible = f == 0 ? 1 : 0;
    st.firstVisible = f;
    st.visibleCount = total;
    st.ticket = 0;                                           // the next launch (or graph replay) starts from zero again
    if (total > 0 && f != 0) {
        // root culled, descendants visible: the traversal starts at the first visible node in BFS order
        const rto_node nd = nodes[f];
        st.x = nd.x; st.y = nd.y; st.z = nd.z;
        int sh = 0;
        while ((1 << sh) < nd.size) sh++;
        st.shift = sh;
        st.leaf = (nd.isUniform == 1 || nd.isLeaf == 1) ? 1 : 0;
        st.solid = nd.isSolid == 1 ? 1 : 0;
        unsigned cur = 0;
        for (int b = depth - 1; b >= sh; b--) {              // child-edge exponent b: the step from edge 2^(b+1) to 2^b
            const unsigned j = ((nd.x >> b) & 1) | (((nd.y >> b) & 1) << 1) | (((nd.z >> b) & 1) << 2);
            const uint2 e = desc[cur];
            cur = e.y + (unsigned)__builtin_popcount((e.x >> 8) & 0xffu & ((1u << j) - 1u));
        }
        st.desc = cur;                                       // meaningful only for an internal start node
    }
    *start = st;
}

// per-block count of visible nodes from the flags (the compaction is made on demand: ensure_compact)
__global__ __launch_bounds__(kBlock) void k_vis_block_counts(const uint8_t* __restrict__ vis, int64_t n, int* __restrict__ blockCount) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int cnt = __syncthreads_count((i < n && vis[i]) ? 1 : 0);
    if (threadIdx.x == 0) blockCount[blockIdx.x] = cnt;
}

// ================================================================ N2: leaf triangles + shadow ray (config 5)
// No upstream counterpart (the reference has no ray/triangle code): semantics are this project's, fixed in
// include/rto_hip.h and DESIGN.md ("parity unpinned").  Traversal = S/RT:239-327 unchanged (LIFO order, 512-pop cap), but a
// popped leaf that passes the slab test is tested against its Marching-Cubes triangles (Moeller-Trumbore, nearest
// t > 0 within the leaf); the first leaf in pop order with a triangle hit ends the traversal.  Shading: face
// normal turned towards the ray, the reference's Lambert term, one shadow ray towards the light.
struct TriScene {
    const rto_node* nodes;
    const float* tris;          // 12 floats per triangle: v0, v1, v2, face normal
    const int* triOffset;       // numNodes + 1
};

__device__ __forceinline__ bool ray_triangle(float ox, float oy, float oz, float dx, float dy, float dz,
                                             const float* __restrict__ T, float& tOut) {
    const float v0x = T[0], v0y = T[1], v0z = T[2];
    const float e1x = T[3] - v0x, e1y = T[4] - v0y, e1z = T[5] - v0z;
    const float e2x = T[6] - v0x, e2y = T[7] - v0y, e2z = T[8] - v0z;
    // p = cross(rd, e2)  (glm cross: x.y*y.z - y.y*x.z, ...)
    const float px = dy * e2z - e2y * dz, py = dz * e2x - e2z * dx, pz = dx * e2y - e2x * dy;
    const float det = e1x * px + e1y * py + e1z * pz;
    if (__builtin_fabsf(det) < 1e-12f) return false;
    const float invDet = 1.0f / det;
    const float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    const float u = (tx * px + ty * py + tz * pz) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
    const float v = (dx * qx + dy * qy + dz * qz) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = (e2x * qx + e2y * qy + e2z * qz) * invDet;
    if (!(t > 0.0f)) return false;
    tOut = t;
    return true;
}

struct TriHit { bool hit; int steps; float t; float nx, ny, nz; };

__device__ __forceinline__ TriHit trace_triangles(const RenderParams& P, const TriScene& S, const Ray& r) {
    TriHit h; h.hit = false; h.steps = 0; h.t = 1e30f; h.nx = h.ny = h.nz = 0.f;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    const float closestT = 1e30f;
    while (sp > 0 && h.steps < kMaxTraversalSteps) {
        sp--;
        const int nodeIdx = stack[sp];
        if (nodeIdx < 0) continue;
        h.steps++;
        const rto_node nd = S.nodes[nodeIdx];
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        if (!slab_exact(P, r, nd.x, nd.y, nd.z, nd.size, tNear, tFar, a0, a1, a2, a3, a4, a5)) continue;
        if (tNear >= closestT) continue;
        if (nd.isUniform == 1 || nd.isLeaf == 1) {
            float bestT = closestT;
            int best = -1;
            const int k1 = S.triOffset[nodeIdx + 1];
            for (int k = S.triOffset[nodeIdx]; k < k1; k++) {
                float t;
                if (ray_triangle(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, S.tris + (size_t)k * 12, t) && t < bestT) { bestT = t; best = k; }
            }
            if (best >= 0) {
                h.hit = true; h.t = bestT;
                h.nx = S.tris[(size_t)best * 12 + 9]; h.ny = S.tris[(size_t)best * 12 + 10]; h.nz = S.tris[(size_t)best * 12 + 11];
                break;
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = nd.child[i];
            if (c >= 0) stack[sp++] = c;
        }
    }
    return h;
}

template <int MODE, bool SHADE>
__global__ __launch_bounds__(kBlock) void k_trace_triangles(RenderParams P, TriScene S, int shadow, float4* __restrict__ out,
                                                             Counters* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int tx = tile % P.tilesX, ty = tile / P.tilesX;
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);
    float shade = kShadeMiss;
    int steps = 0;
    bool hit = false;
    if (inImage) {
        const Ray r = generate_ray_tab(P, px, py);
        const TriHit h = trace_triangles(P, S, r);
        steps = h.steps;
        hit = h.hit;
        if (h.hit) {
            float nx = h.nx, ny = h.ny, nz = h.nz;
            if (nx * r.dx + ny * r.dy + nz * r.dz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
            float ndotl = gmax(0.0f, nx * P.lightNeg[0] + ny * P.lightNeg[1] + nz * P.lightNeg[2]);
            if (shadow) {
                const float bias = P.voxelSize * 1e-3f;
                const float hx = r.ox + r.dx * h.t, hy = r.oy + r.dy * h.t, hz = r.oz + r.dz * h.t;
                Ray s;
                s.ox = hx + nx * bias; s.oy = hy + ny * bias; s.oz = hz + nz * bias;
                s.dx = P.lightNeg[0]; s.dy = P.lightNeg[1]; s.dz = P.lightNeg[2];
                s.ix = 1.0f / s.dx; s.iy = 1.0f / s.dy; s.iz = 1.0f / s.dz;
                const TriHit sh = trace_triangles(P, S, s);
                steps += sh.steps;
                if (sh.hit) ndotl = 0.0f;
            }
            shade = ndotl;
        }
    }
    if (valid) {
        if (SHADE) __builtin_nontemporal_store(shade, reinterpret_cast<float*>(out) + (size_t)ly * P.W + px);
        else store_pixel(out + (size_t)ly * P.W + px, shade_color(shade));
    }
    if (MODE == kModeSteps) {
        // counters: pops = steps of the primary + shadow traversals; capped = primary misses that ran into the cap
        unsigned long long pops = inImage ? (unsigned long long)steps : 0ull, hits = (inImage && hit) ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { pops += __shfl_down(pops, off); hits += __shfl_down(hits, off); }
        if (lane == 0) { atomicAdd(&counters->pops, pops); atomicAdd(&counters->hits, hits); }
    }
}

// ---------------------------------------------------------------- N2 on the packed descriptors
// Same semantics as k_trace_triangles, on the child descriptors: bits 24..31 of a descriptor's .x mark the leaf
// children that own triangles (k_desc_trimask).  Solid leaves are no longer hits by themselves; the interesting
// children of a node are its internal children and its triangle-owning leaves that pass the slab test.  Popping a
// triangle leaf runs Moeller-Trumbore over its range (triOffset is indexed by node: first child + slot) and, on a
// miss, continues with the next child of the SAME node, so the per-level state (W, base, first child) is carried
// in registers and the LDS entry is 16 bytes.
struct PackedTriScene {
    const uint2* desc;
    const int* descFirstChild;
    const float* tris;
    const int* triOffset;
};

__global__ __launch_bounds__(kBlock) void k_desc_trimask(const int* __restrict__ descFirstChild, const int* __restrict__ triOffset,
                                                          int64_t nInternal, uint2* __restrict__ desc) {
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= nInternal) return;
    const int c0 = descFirstChild[d];
    const unsigned x = desc[d].x;
    const unsigned im = (x >> 8) & 0xffu;
    unsigned tm = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (!((im >> k) & 1u) && triOffset[c0 + k + 1] > triOffset[c0 + k]) tm |= 1u << k;
    desc[d].x = (x & 0x00ffffffu) | (tm << 24);
}

struct PTriHit { bool hit; int steps; float t; int tri; };

__device__ __forceinline__ PTriHit trace_packed_triangles(const RenderParams& P, const PackedTriScene& S, const Ray& r, uint4* stk) {
    PTriHit h; h.hit = false; h.steps = 1; h.t = 1e30f; h.tri = -1;
    {
        float tNear, tFar, a0, a1, a2, a3, a4, a5;
        if (!(slab_exact(P, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f))) return h;
    }
    const bool risky = !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                         __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));
    unsigned cur = 0;
    int cx = 0, cy = 0, cz = 0;
    int lvl = 0;                   // level of the node whose children are being popped (after the visit)
    unsigned lvlPending = 0, tailRun = 0;
    unsigned W = 0, base = 0;      // state of level `lvl`: pending | internal<<8 | visible<<16 | tailAbove<<24
    int c0 = 0;                    // node index of its first child
    unsigned belowPrev = 0xffu;    // children of it not popped yet
    bool needVisit = true;
    int steps = 1;
    bool alive = true;
    bool haveLeaf = false;         // a popped triangle leaf waits for its test
    int leafNode = 0;
    // "while-while" form: lanes first walk nodes until each holds a triangle leaf (or is done), then all of them run
    // their triangle loops together -- instead of paying a node visit AND the longest triangle loop in every iteration.
    while (alive) {
        while (alive && !haveLeaf) {
            if (needVisit) {
                const uint2 d = S.desc[cur];
                c0 = S.descFirstChild[cur];
                const int half = 1 << (P.depth - 1 - lvl);
                unsigned passMask;
                if (__builtin_amdgcn_ballot_w64(risky) != 0ull) passMask = child_pass_mask<true>(P, r, cx, cy, cz, half);
                else passMask = child_pass_mask_fast(P, r, cx, cy, cz, half);
                const unsigned vm0 = __builtin_amdgcn_ubfe(d.x, 16, 8);
                const unsigned im0 = __builtin_amdgcn_ubfe(d.x, 8, 8);
                const unsigned tm0 = d.x >> 24;
                W = (d.x & 0x00ffff00u) | (((im0 | tm0) & vm0) & passMask) | (tailRun << 24);
                base = d.y;
                belowPrev = 0xffu;
                needVisit = false;
            }
            if ((W & 0xffu) == 0) {
                // this level is exhausted: its remaining children only count steps; resume at the deepest level with work
                steps += __builtin_popcount(((W >> 16) & 0xffu) & belowPrev) + (int)(W >> 24);
                if (lvlPending == 0 || steps >= kMaxTraversalSteps) { alive = false; break; }
                const int L = 31 - __builtin_clz(lvlPending);
                const uint4 e = stk[L * kWave];
                W = e.x; base = e.y; c0 = (int)e.z;
                lvlPending &= ~(1u << L);          // its state lives in registers again; a later descent re-files it
                const int bpos = P.depth - 1 - L;
                const unsigned childIdx = ((cx >> bpos) & 1) | (((cy >> bpos) & 1) << 1) | (((cz >> bpos) & 1) << 2);
                belowPrev = (1u << childIdx) - 1u;
                const int keep = (int)(0xffffffffu << (bpos + 1));
                cx &= keep; cy &= keep; cz &= keep;
                lvl = L;
            }
            // pop the next interesting child of level lvl
            const unsigned pending = W & 0xffu, im = (W >> 8) & 0xffu, vm = (W >> 16) & 0xffu, tailAbove = W >> 24;
            const int j = 31 - __builtin_clz(pending);
            const unsigned bitj = 1u << j;
            const int stepsC = steps + __builtin_popcount(vm & belowPrev & ~((bitj << 1) - 1u));
            if (stepsC >= kMaxTraversalSteps) { steps = kMaxTraversalSteps; alive = false; break; }
            steps = stepsC + 1;
            W ^= bitj;
            belowPrev = bitj - 1u;
            if (im & bitj) {
                stk[lvl * kWave] = make_uint4(W, base, (unsigned)c0, 0u);
                if (W & 0xffu) { lvlPending |= (1u << lvl); tailRun = 0; }
                else { lvlPending &= ~(1u << lvl); tailRun = tailAbove + (unsigned)__builtin_popcount(vm & (bitj - 1u)); }
                cur = base + (unsigned)__builtin_popcount(im & (bitj - 1u));
                const int hl = 1 << (P.depth - 1 - lvl);
                cx |= (j & 1) ? hl : 0; cy |= (j & 2) ? hl : 0; cz |= (j & 4) ? hl : 0;
                lvl++;
                needVisit = true;
            } else {
                haveLeaf = true;               // a leaf that owns triangles and passed the slab test
                leafNode = c0 + j;
            }
        }
        if (haveLeaf) {
            // nearest t > 0 within the leaf
            float bestT = 1e30f;
            int best = -1;
            const int k1 = S.triOffset[leafNode + 1];
            for (int k = S.triOffset[leafNode]; k < k1; k++) {
                float t;
                if (ray_triangle(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, S.tris + (size_t)k * 12, t) && t < bestT) { bestT = t; best = k; }
            }
            if (best >= 0) { h.hit = true; h.t = bestT; h.tri = best; alive = false; }
            haveLeaf = false;
        }
    }
    if (!h.hit && steps > kMaxTraversalSteps) steps = kMaxTraversalSteps;
    h.steps = steps;
    return h;
}

template <int MODE, bool SHADE>
__global__ __launch_bounds__(kBlock) void k_trace_packed_triangles(RenderParams P, PackedTriScene S, int shadow, float4* __restrict__ out,
                                                                    Counters* __restrict__ counters) {
    extern __shared__ uint4 lds_stack4[];   // [wave][level][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint4* stk = lds_stack4 + (size_t)wave * P.depth * kWave + lane;
    const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    if (slot >= P.launchWaves) return;
    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);
    // outside this rectangle no ray can meet the geometry (instrumented frames: the root box, whose miss costs exactly the
    // root's pop; colour / shade frames: the solid leaves' box widened by a voxel, which holds every triangle)
    const bool outside = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;
    float shade = kShadeMiss;
    int steps = 0;
    bool hit = false;
    if (inImage && outside) steps = 1;
    if (inImage && !outside) {
        const Ray r = generate_ray_tab(P, px, py);
        const PTriHit h = trace_packed_triangles(P, S, r, stk);
        steps = h.steps;
        hit = h.hit;
        if (h.hit) {
            float nx = S.tris[(size_t)h.tri * 12 + 9], ny = S.tris[(size_t)h.tri * 12 + 10], nz = S.tris[(size_t)h.tri * 12 + 11];
            if (nx * r.dx + ny * r.dy + nz * r.dz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
            float ndotl = gmax(0.0f, nx * P.lightNeg[0] + ny * P.lightNeg[1] + nz * P.lightNeg[2]);
            if (shadow) {
                const float bias = P.voxelSize * 1e-3f;
                const float hx = r.ox + r.dx * h.t, hy = r.oy + r.dy * h.t, hz = r.oz + r.dz * h.t;
                Ray s;
                s.ox = hx + nx * bias; s.oy = hy + ny * bias; s.oz = hz + nz * bias;
                s.dx = P.lightNeg[0]; s.dy = P.lightNeg[1]; s.dz = P.lightNeg[2];
                s.ix = 1.0f / s.dx; s.iy = 1.0f / s.dy; s.iz = 1.0f / s.dz;
                const PTriHit sh = trace_packed_triangles(P, S, s, stk);
                steps += sh.steps;
                if (sh.hit) ndotl = 0.0f;
            }
            shade = ndotl;
        }
    }
    if (P.tileCost) {
        // this tile's cost for the launch order: the pops of its busiest ray (primary + shadow), 16 per bucket
        int cst = steps;
        for (int off = 32; off > 0; off >>= 1) cst = max(cst, __shfl_xor(cst, off));
        if (lane == 0 && ty < P.tilesY) P.tileCost[tile] = cst >> 4;
    }
    if (valid && !(P.skipOutside && outside)) {
        if (SHADE) __builtin_nontemporal_store(shade, reinterpret_cast<float*>(out) + (size_t)ly * P.W + px);
        else store_pixel(out + (size_t)ly * P.W + px, shade_color(shade));
    }
    if (MODE == kModeColor) fill_outside<SHADE ? kModeShade : kModeColor>(P, out, lane, slot);
    if (MODE == kModeSteps) {
        unsigned long long pops = inImage ? (unsigned long long)steps : 0ull, hits = (inImage && hit) ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { pops += __shfl_down(pops, off); hits += __shfl_down(hits, off); }
        if (lane == 0) { atomicAdd(&counters->pops, pops); atomicAdd(&counters->hits, hits); }
    }
}

// ---------------------------------------------------------------- N2 in the lean form (default for canonical trees)
// k_trace_packed_triangles with the loop of k_trace_lean.  What makes that possible is ONE index space for everything a
// ray can pop and act on: the "interesting" children of a node -- its internal children and its triangle-owning leaves --
// get consecutive records in `rec` (k_unified_*), so `first record + popcount(interesting children below j)` addresses
// child j whether it is a node (record = its descriptor: masks | first record of ITS interesting children) or a leaf
// (record = first triangle, triangle count).  A stack entry is then the same 8 bytes as in k_trace_lean (the packed form
// needs 16: it carries the node index of the first child to reach triOffset), which doubles the waves an LDS-limited CU
// holds, and the pop counters follow from identities (1) and (2) of trace_tile_lean instead of being carried.
//   * a lane that pops a triangle leaf waits; when 16 lanes wait (or nobody walks any more) the triangles of the waiting
//     lanes are dealt out to all 64 lanes, pair by pair; a miss pops the next pending candidate there and then (no visit:
//     the entry is on the stack);
//   * a lane whose primary ray has found its triangle starts its shadow ray AT ONCE, inside the same loop, instead of
//     idling until the whole wave has finished its primary rays.
struct LeanTriScene {
    const uint2* rec;           // [0] the root's descriptor; see above
    const float* tris;          // 12 floats per triangle: v0, v1, v2, face normal
#if defined(RTO_TRI_TIMELINE)
    int* timeline;              // A/B build: 8 ints per tile {start lo, hi, end lo, hi (100 MHz), trips, rounds, HW_ID, XCC_ID | slot << 4}
#endif
#if defined(RTO_TRI_STAMP)
    unsigned* stamp;            // A/B build (tools/tri_stamp.py): 24 words per tile = 7 phases x {s_memtime ticks, times entered, lanes with work}
#endif
};

// interesting children per descriptor (after k_desc_trimask)
__global__ __launch_bounds__(kBlock) void k_unified_count(const uint2* __restrict__ desc, int64_t nInternal, int* __restrict__ count) {
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= nInternal) return;
    const unsigned x = desc[d].x;
    count[d] = __builtin_popcount(((x >> 8) | (x >> 24)) & 0xffu);
}

// first[d] = exclusive scan of count: the records of d's interesting children start at 1 + first[d] (record 0 is the root)
__global__ __launch_bounds__(kBlock) void k_unified_fill(const uint2* __restrict__ desc, const int* __restrict__ descFirstChild,
                                                          const int* __restrict__ triOffset, const int* __restrict__ first, int64_t nInternal,
                                                          uint2* __restrict__ rec) {
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= nInternal) return;
    const uint2 dd = desc[d];
    const unsigned im = (dd.x >> 8) & 0xffu, tm = dd.x >> 24, it = im | tm;
    const int c0 = descFirstChild[d];
    const unsigned r0 = 1u + (unsigned)first[d];
    if (d == 0) rec[0] = make_uint2(dd.x, r0);
    for (int k = 0; k < 8; k++) {
        if (!((it >> k) & 1u)) continue;
        const unsigned idx = r0 + (unsigned)__builtin_popcount(it & ((1u << k) - 1u));
        if ((im >> k) & 1u) {
            const unsigned cd = dd.y + (unsigned)__builtin_popcount(im & ((1u << k) - 1u));
            rec[idx] = make_uint2(desc[cd].x, 1u + (unsigned)first[cd]);
        } else {
            const int t0 = triOffset[c0 + k];
            rec[idx] = make_uint2((unsigned)t0, (unsigned)(triOffset[c0 + k + 1] - t0));
        }
    }
}

#ifndef RTO_TRI_WAVES
#define RTO_TRI_WAVES 6         // round 5: without the SLP vectoriser the kernel fits 80 VGPRs: 6 waves per SIMD (LDS: 24 x 5.6 KB of 160)
#endif
#ifndef RTO_TRI_BATCH
#define RTO_TRI_BATCH 16
#endif
template <int MODE, bool SHADE>
__device__ __forceinline__ void trace_tile_lean_triangles(const RenderParams& P, const LeanTriScene& Sc, int shadow, float4* __restrict__ out,
                                                          Counters* __restrict__ counters, uint2* stk, unsigned long long* keys,
                                                          const int lane, const int slot) {
#if defined(RTO_TRI_TIMELINE)
    const unsigned long long profT0 = wall_clock64();
#endif
#if defined(RTO_TRI_STAMP)
    // phases: 0 prologue, 1 node loop, 2 round set-up (leaf records, prefix sum, keys), 3 pair chunks (dealing + Moeller-Trumbore),
    // 4 key read-back + pop-next, 5 a ray's end (accounting, shadow ray start), 6 epilogue.  The stamp's s_waitcnt also drains
    // the wave's LDS queue: the build measures where the time goes, its frame is slower than the product's.
    unsigned stTicks[7] = { 0, 0, 0, 0, 0, 0, 0 }, stEnter[7] = { 0, 0, 0, 0, 0, 0, 0 }, stLanes[7] = { 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long stLast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stLast) :: "memory");
#define RTO_TS(i, lanes) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0); stTicks[i] += (unsigned)(t_ - stLast); stEnter[i] += 1u; stLanes[i] += (unsigned)(lanes); stLast = t_; } while (0)
#define RTO_TS_COUNT(mask) __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64(mask)))
#else
#define RTO_TS(i, lanes) do { } while (0)
#endif
    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && (py < P.H);
    // outside this rectangle no ray can meet the geometry (instrumented frames: the root box, whose miss costs exactly the
    // root's pop; colour / shade frames: the solid leaves' box widened by a voxel, which holds every triangle)
    const bool outside = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;
    const Geo G = geo_of(P);
    float lnx = P.lightNeg[0], lny = P.lightNeg[1], lnz = P.lightNeg[2];
    asm volatile("" : "+s"(lnx), "+s"(lny), "+s"(lnz));                    // see geo_of
    // every shadow ray has this direction: its reciprocal and signs are computed once per wave, not once per hit
    const float lix = P.lightInv[0], liy = P.lightInv[1], liz = P.lightInv[2];     // from the host: three divisions per wave less
    const unsigned lsx = (unsigned)((int)__float_as_uint(lix) >> 31), lsy = (unsigned)((int)__float_as_uint(liy) >> 31),
                   lsz = (unsigned)((int)__float_as_uint(liz) >> 31);

    float shade = kShadeMiss;
    int stepsTotal = 0;            // pops of the finished rays of this pixel (primary, then shadow)
    bool hitPrimary = false;
    Ray r;
    bool alive = false;            // walking the tree
    const bool tileLive = tile_may_hit(P, tx, ty, slot, lane, tile);     // the occupancy mask (colour / shade frames): wave-uniform; a tile outside it records its cost there
    if (inImage && tileLive) {
        stepsTotal = 1;            // the root's own pop of the primary ray
        if (!outside) {
            r = generate_ray_tab(P, px, py);
            float tNear, tFar, a0, a1, a2, a3, a4, a5;
            alive = slab_exact(G, r, 0, 0, 0, P.rootSize, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
        }
    }
    const bool risky = alive && !(__builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                                  __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) && __builtin_isfinite(r.oz));
    const bool anyRisky = __builtin_amdgcn_ballot_w64(risky) != 0ull;   // wave-uniform; shadow rays are never risky themselves
    const bool exactGrid = P.exactGrid != 0;
                                                                         // (finite light direction), the exact form serves them too
    unsigned cur = 0;
    int cx = (int)kCoordBias, cy = (int)kCoordBias, cz = (int)kCoordBias;      // node position, biased (child_axis_terms): the bit operations below never touch the bias
    int bpos = P.depth - 1;
    unsigned lvlPending = 0;
    const unsigned sentinel = 1u << P.depth;
    int S = 0;                                                          // identity (1), for the ray in flight
    const int capBound = kMaxTraversalSteps + 7 * P.depth;
    // exact grids (child_fail_mask_exactgrid3): which children a failed comparison strikes; remade when the shadow ray starts.  The
    // direction signs themselves are taken from the reciprocals in every trip (three registers fewer across the loop)
    ExactPat3 K = exact_patterns3((unsigned)((int)__float_as_uint(r.ix) >> 31), (unsigned)((int)__float_as_uint(r.iy) >> 31), (unsigned)((int)__float_as_uint(r.iz) >> 31));
    bool haveLeaf = false;         // a popped triangle leaf (record `cur`) waits for its test
    bool ended = false;            // the ray in flight ran out of nodes (or into the cap) in the node loop
    bool shadowRay = false;        // the ray in flight is the pixel's shadow ray
    float ndotl = 0.0f;
    const char* recBytes = reinterpret_cast<const char*>(Sc.rec);
    // scratch of the triangle rounds: this wave's row of the dummy stack entry `depth` (nothing valid is ever read from it,
    // and no node trip runs while a round uses it)
    unsigned* marks = reinterpret_cast<unsigned*>(stk - lane + P.depth * kWave);
    int trips = 0, rounds = 0;
#if defined(RTO_TRI_PROFILE)       // A/B build (tools/tri_profile.py): where a frame's instructions go
    unsigned long long profLaneTrips = 0, profChunks = 0, profPairs = 0, profWaveTrips = 0;
#endif
    int chunks = 0;                // triangle chunks of this wave (wave-uniform): part of the tile's cost
#if defined(RTO_TRI_STAMP)
    unsigned stWaveTrips = 0;      // loop bodies of the node phase this WAVE issued (>= the trips of its busiest lane: every round pays its own longest lane)
    RTO_TS(0, RTO_TS_COUNT(alive));
#endif

    for (;;) {
        // node loop: until nobody walks, or RTO_TRI_BATCH lanes wait with a leaf (the tests cost by the pair, so a round
        // for 16 leaves -- ~70 pairs -- fills the wave; measured at config 5: 2 / 4 / 8 / 12 / 16 / 20 / 32 / 48 / 64 lanes ->
        // 632 / 599 / 554 / 535 / 533 / 534 / 560 / 614 / 641 us)
        // Shape of the loop: a plain divergent `while (alive)` (lanes drop out as they pop a leaf or end) with a wave-uniform
        // break -- `waiting` lives in a scalar register: the lanes waiting at the start of the round plus, per trip, the lanes
        // that have just popped a leaf (the ballot sees exactly the lanes walking in this trip).  The earlier form, a loop on
        // the two ballots around `if (alive)`, cost ~20 VALU instructions per trip in register copies at the loop's edges.
        int waiting = __builtin_popcountll(__builtin_amdgcn_ballot_w64(haveLeaf));
#if defined(RTO_TRI_STAMP)
        int stT = 0;               // this lane's trips in this round: the wave issued max(stT) loop bodies for sum(stT) lane-trips
#endif
        if (waiting < RTO_TRI_BATCH)
        while (alive) {
            trips++;
#if defined(RTO_TRI_PROFILE)
            { const unsigned long long act = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));     // lanes walking in this trip
              profLaneTrips += __builtin_amdgcn_readfirstlane((int)act); profWaveTrips += 1; }
#endif
#if defined(RTO_TRI_STAMP)
            stT++;
#endif
            const uint2 d = *reinterpret_cast<const uint2*>(recBytes + (cur << 3));
            const int Lb = __builtin_ctz(lvlPending | sentinel);
            const uint2 e = stk[Lb * kWave];
            const float fh = __uint_as_float((unsigned)(bpos + 127) << 23);        // (float)(1 << bpos), exact (bpos = -1 after a finest leaf: unused)
            unsigned fail8;
            if (anyRisky) fail8 = child_fail_mask_exact(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz,
                                                        r.ix, r.iy, r.iz, cx & 0x7fffff, cy & 0x7fffff, cz & 0x7fffff, 1 << (bpos & 31));
            else if (exactGrid) fail8 = child_fail_mask_exactgrid3<true, true>(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                                               K, cx, cy, cz, fh);                    // grid planes proven exact: 9 plane parameters, 12 comparisons
            else fail8 = child_fail_mask_fast<true, true>(G.gx, G.gy, G.gz, G.vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                          (unsigned)((int)__float_as_uint(r.ix) >> 31), (unsigned)((int)__float_as_uint(r.iy) >> 31),
                                                          (unsigned)((int)__float_as_uint(r.iz) >> 31), cx, cy, cz, fh);
            const unsigned vm0 = (d.x >> 16) & 0xffu;
            S += __builtin_popcount(vm0);
            // children that do more than count a pop: visible internal ones and visible triangle leaves that pass the slab test
            const unsigned cand = bop3<kAndAndNot>((d.x >> 8) | (d.x >> 24), vm0, fail8);
            const unsigned noWork = (unsigned)(((int)cand - 1) >> 31);
            const bool dead = (cand | lvlPending) == 0 || S >= capBound;
            const unsigned W = bop3<kSelC>(bop3<kAndOr>(d.x, 0xffffff00u, cand), e.x, noWork);
            const unsigned base = bop3<kSelC>(d.y, e.y, noWork);
            const int bpos2 = (int)bop3<kSelC>((unsigned)bpos, (unsigned)Lb, noWork);
            const int j = 31 - __builtin_clz(W & 0xffu);
            const unsigned bitj = 0x80000000u >> (31 - j);                   // = 1u << j as a RIGHT shift: variable right shifts issue at full rate on gfx950, left shifts at half (tools/ubench/valu_rate4.hip)
            const bool leaf = ((W >> 8) & bitj) == 0;
            const unsigned Wn = W ^ bitj;
            stk[bpos2 * kWave] = make_uint2(Wn, base);
            const unsigned hl = 1u << bpos2;
            const unsigned noneLeft = (unsigned)(((int)(Wn & 0xffu) - 1) >> 31);
            lvlPending = bop3<kReplace>(lvlPending, hl, noneLeft);
            cur = base + (unsigned)__builtin_popcount(((W >> 8) | (W >> 24)) & (bitj - 1u));    // record of the popped child
            const unsigned keep = 0u - (hl + hl);
            const unsigned sj = (unsigned)j << bpos2;
            cx = (int)bop3<kOrAnd>((unsigned)cx & keep, sj, hl);
            cy = (int)bop3<kOrAnd>((unsigned)cy & keep, sj >> 1, hl);
            cz = (int)bop3<kOrAnd>((unsigned)cz & keep, sj >> 2, hl);
            bpos = bpos2 - 1;
            haveLeaf = !dead && leaf;
            ended = dead;
            alive = !dead && !leaf;
            waiting += __builtin_popcountll(__builtin_amdgcn_ballot_w64(haveLeaf));
            if (waiting >= RTO_TRI_BATCH) break;
        }
        rounds++;
#if defined(RTO_TRI_STAMP)
        stWaveTrips += (unsigned)__builtin_amdgcn_readlane(wave_scan_max_nonneg(stT), kWave - 1);
        RTO_TS(1, __builtin_amdgcn_readlane(wave_scan_add(stT), kWave - 1));
#endif
        // ---- the waiting lanes' triangles, tested by ALL 64 lanes (S/RT semantics of the pop: it happens only below the
        //      cap).  Leaves own 1 to several dozen triangles (big uniform leaves), so "each lane loops over its own leaf"
        //      runs at 21 % lane utilisation behind the longest leaf.  Instead the (leaf, triangle) pairs of the whole wave
        //      are numbered by a prefix sum and dealt out 64 at a time: pair w belongs to the first lane whose inclusive
        //      prefix exceeds w (binary search with ds_bpermute), the worker fetches that lane's ray by ds_bpermute, and a
        //      hit is folded into the owner's 64-bit key (t bits, triangle index) with ds_min_u64 -- the same winner as the
        //      sequential loop: smallest t, then smallest index.
        bool endedHit = false;
        int Rrest = 0;             // identity (2) at the leaf that was hit, when it was needed
        bool haveR = false;
        float bestT = 1e30f;
        int best = -1;
        if (__builtin_amdgcn_ballot_w64(haveLeaf) != 0ull) {
            if (haveLeaf && (1 + S > kMaxTraversalSteps || MODE == kModeSteps)) {
                const uint2* stkR = stk;
                asm volatile("" : "+v"(stkR));       // this rare walk takes its base from `stk` HERE: a pre-offset copy hoisted to the kernel's head was a register held (and at 6 waves per SIMD spilled) for the whole walk
                for (int b = P.depth - 1; b > bpos; b--) {
                    const unsigned w = stkR[b * kWave].x;
                    const unsigned jl = ((cx >> b) & 1) | (((cy >> b) & 1) << 1) | (((cz >> b) & 1) << 2);
                    Rrest += __builtin_popcount(__builtin_amdgcn_ubfe(w, 16, 8) & ((1u << jl) - 1u));
                }
                haveR = true;
                if (1 + S - Rrest > kMaxTraversalSteps) { ended = true; S = capBound; haveLeaf = false; }   // S/RT:254: the loop ended before this pop
            }
            uint2 tr = make_uint2(0u, 0u);                          // first triangle, triangles of this lane's leaf
            if (haveLeaf) tr = *reinterpret_cast<const uint2*>(recBytes + (cur << 3));
            const int cnt = (int)tr.y;
            const int incl = wave_scan_add(cnt);
            const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
            const int first = (int)tr.x - (incl - cnt);             // pair w of this lane's leaf is triangle first + w
            keys[lane] = ~0ull;
            __builtin_amdgcn_wave_barrier();
            RTO_TS(2, RTO_TS_COUNT(haveLeaf));
            for (int w0 = 0; w0 < total; w0 += kWave) {
#if defined(RTO_TRI_PROFILE)
                profChunks++; profPairs += (unsigned long long)min(kWave, total - w0);
#endif
                chunks++;
                const int w = w0 + lane;
                // owner of pair w = the lane whose range [incl - cnt, incl) holds it.  Every lane whose range meets this chunk
                // writes its number at the chunk position where the range begins (ranges are disjoint and in lane order, so are
                // the marks); the owner of a position is then the last mark at or before it: a running maximum in registers.
                // (A wave's LDS operations execute in order: the zeroes land before the marks, the marks before the read.)
                marks[lane] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const int p = incl - cnt - w0;
                if (cnt > 0 && p < kWave && incl > w0) marks[max(p, 0)] = (unsigned)lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const int owner = wave_scan_max_nonneg((int)marks[lane]);
                const int oa = owner << 2;                          // ds_bpermute address of the owner's lane
                const int k = __builtin_amdgcn_ds_bpermute(oa, first) + w;
#define RTO_FROM(x) __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(x)))
                const float oox = RTO_FROM(r.ox), ooy = RTO_FROM(r.oy), ooz = RTO_FROM(r.oz);
                const float odx = RTO_FROM(r.dx), ody = RTO_FROM(r.dy), odz = RTO_FROM(r.dz);
#undef RTO_FROM
                if (w < total) {
                    float t;
                    if (ray_triangle(oox, ooy, ooz, odx, ody, odz, Sc.tris + (size_t)k * 12, t) && t < 1e30f)
                        atomicMin(&keys[owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)k);   // t > 0: its bits order like the value
                }
            }
            __builtin_amdgcn_wave_barrier();
            RTO_TS(3, total);
#if defined(RTO_TRI_STAMP)
            const int stWaiting = RTO_TS_COUNT(haveLeaf);
#endif
            if (haveLeaf) {
                const unsigned long long key = keys[lane];
                if (key != ~0ull) { best = (int)(unsigned)key; bestT = __uint_as_float((unsigned)(key >> 32)); }
                if (best >= 0) { ended = true; endedHit = true; haveLeaf = false; }
                else if (lvlPending == 0) { ended = true; haveLeaf = false; }              // nothing left: the ray misses
                else {
                    // pop the next pending candidate right here -- no visit, the entry is on the stack; if that is another
                    // triangle leaf the lane waits for the next round (batched tests beat chained ones)
                    const int Lb = __builtin_ctz(lvlPending);
                    const uint2 e = stk[Lb * kWave];
                    const int j = 31 - __builtin_clz(e.x & 0xffu);
                    const unsigned bitj = 1u << j;
                    const unsigned Wn = e.x ^ bitj;
                    stk[Lb * kWave] = make_uint2(Wn, e.y);
                    const unsigned hl = 1u << Lb;
                    if ((Wn & 0xffu) == 0) lvlPending &= ~hl;
                    cur = e.y + (unsigned)__builtin_popcount(((e.x >> 8) | (e.x >> 24)) & (bitj - 1u));
                    const unsigned keep = 0u - (hl + hl);
                    cx = (int)(((unsigned)cx & keep) | ((j & 1) ? hl : 0u));
                    cy = (int)(((unsigned)cy & keep) | ((j & 2) ? hl : 0u));
                    cz = (int)(((unsigned)cz & keep) | ((j & 4) ? hl : 0u));
                    bpos = Lb - 1;
                    if ((e.x >> 8) & bitj) { haveLeaf = false; alive = true; }             // an internal child: back to the node loop
                }
            }
            RTO_TS(4, stWaiting);
        }
        // ---- a ray has ended: account for it; a primary hit starts the shadow ray
#if defined(RTO_TRI_STAMP)
        const int stEnded = RTO_TS_COUNT(ended);
#endif
        if (ended) {
            ended = false;
            int steps = 1 + S;
            if (endedHit && haveR) steps -= Rrest;
            if (steps > kMaxTraversalSteps) steps = kMaxTraversalSteps;
            stepsTotal += steps - (shadowRay ? 0 : 1);                 // the primary ray's root pop is already in
            if (!shadowRay) {
                if (endedHit) {
                    hitPrimary = true;
                    float nx = Sc.tris[(size_t)best * 12 + 9], ny = Sc.tris[(size_t)best * 12 + 10], nz = Sc.tris[(size_t)best * 12 + 11];
                    if (nx * r.dx + ny * r.dy + nz * r.dz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
                    ndotl = gmax(0.0f, nx * lnx + ny * lny + nz * lnz);
                    shade = ndotl;
                    // a hit that faces away from the light is black whatever its shadow ray finds: only the instrumented frames (which
                    // count that ray's pops) still trace it
                    if (shadow && (MODE == kModeSteps || ndotl > 0.0f)) {
                        const float bias = G.vs * 1e-3f;
                        const float hx = r.ox + r.dx * bestT, hy = r.oy + r.dy * bestT, hz = r.oz + r.dz * bestT;
                        r.ox = hx + nx * bias; r.oy = hy + ny * bias; r.oz = hz + nz * bias;
                        r.dx = lnx; r.dy = lny; r.dz = lnz;
                        r.ix = lix; r.iy = liy; r.iz = liz;
                        K = exact_patterns3(lsx, lsy, lsz);
                        shadowRay = true;
                        float tNear, tFar, a0, a1, a2, a3, a4, a5;
                        int rootEdge = P.rootSize;
                        asm volatile("" : "+s"(rootEdge));          // the root box is worked out HERE, once per pixel: hoisted out of the walk its three max planes are
                                                                     // wave-uniform values in vector registers for the whole kernel (the first to be spilled at 6 waves per SIMD)
                        int rootAt = 0;
                        asm volatile("" : "+s"(rootAt));            // ... and so are its three min planes
                        alive = slab_exact(G, r, rootAt, rootAt, rootAt, rootEdge, tNear, tFar, a0, a1, a2, a3, a4, a5) && !(tNear >= 1e30f);
                        if (!alive) stepsTotal += 1;                  // the shadow ray pops the root and misses it
                        cur = 0; cx = cy = cz = (int)kCoordBias; bpos = P.depth - 1; lvlPending = 0; S = 0;
                    }
                }
            } else if (endedHit) shade = 0.0f;                         // something between the hit and the light
        }
        RTO_TS(5, stEnded);
        if (__builtin_amdgcn_ballot_w64(alive || haveLeaf) == 0ull) break;
    }
#if defined(RTO_TRI_TIMELINE)
    if (Sc.timeline && MODE == kModeColor) {
        int tmax = trips;
        for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
        if (lane == 0 && ty < P.tilesY) {
            const unsigned long long tl1 = wall_clock64();
            const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
            int* rec = Sc.timeline + (size_t)tile * 8;
            rec[0] = (int)(profT0 & 0xffffffffu); rec[1] = (int)(profT0 >> 32); rec[2] = (int)(tl1 & 0xffffffffu); rec[3] = (int)(tl1 >> 32);
            rec[4] = tmax; rec[5] = rounds | (chunks << 12); rec[6] = (int)hwid; rec[7] = (int)(xcc & 15u) | (slot << 4);
        }
    }
#endif
#if defined(RTO_TRI_PROFILE)
    if (counters && MODE == kModeColor) {
        // profLaneTrips / profWaveTrips live in the lanes that walked; take the wave maximum (every walking lane saw the same counts while it walked)
        unsigned long long lt = profLaneTrips, wt = profWaveTrips;
        for (int off = 32; off > 0; off >>= 1) { lt = max(lt, (unsigned long long)__shfl_xor((long long)lt, off)); wt = max(wt, (unsigned long long)__shfl_xor((long long)wt, off)); }
        if (lane == 0) {
            atomicAdd(&counters->pops, (wt << 32) | (unsigned long long)rounds);                           // (longest lane's) trips | rounds
            atomicAdd(&counters->hits, lt);                                                                // lanes walking beside the longest lane, summed over its trips
            atomicAdd(&counters->capped, (profChunks << 32) | profPairs);                                  // triangle chunks | pairs
        }
    }
#endif
    if (P.tileCost) {
        // the wave's time on a full machine, fitted over a frame's timeline (tools/tri_timeline.py): 0.67 us per trip of its busiest
        // lane + 1.88 per round + 0.62 per chunk.  Until round 3 the chunks were left out: the tiles of big leaves (200 chunks in 9
        // rounds) were rated cheap, started at 300 of 440 us and ended the frame alone
        int cost = trips;
        for (int off = 32; off > 0; off >>= 1) cost = max(cost, __shfl_xor(cost, off));
        cost += 3 * rounds + chunks;
        if (lane == 0 && ty < P.tilesY && tileLive) P.tileCost[tile] = ((cost >> 3) == 0 && P.tileMask) ? -1 : cost >> 3;   // inside the mask, no work: the rim (as trace_tile_lean)
    }
    if (valid && !(P.skipOutside && outside)) {
        // the pixel's place is worked out again from the lane number (behind a barrier the compiler cannot see through): two registers
        // that would otherwise live across the whole walk -- the kernel sits at its budget of 96 (5 waves per SIMD)
        int lane2 = lane;
        asm volatile("" : "+v"(lane2));
        const size_t pix = (size_t)(ty * 8 + (lane2 >> 3)) * P.W + (size_t)(tx * 8 + (lane2 & 7));
        if (SHADE) __builtin_nontemporal_store(shade, reinterpret_cast<float*>(out) + pix);
        else store_pixel(out + pix, shade_color(shade));
    }
    if (MODE == kModeColor) fill_outside<SHADE ? kModeShade : kModeColor>(P, out, lane, slot);
    if (MODE == kModeSteps) {
        unsigned long long pops = inImage ? (unsigned long long)stepsTotal : 0ull, hits = (inImage && hitPrimary) ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { pops += __shfl_down(pops, off); hits += __shfl_down(hits, off); }
        if (lane == 0) { atomicAdd(&counters->pops, pops); atomicAdd(&counters->hits, hits); }
    }
#if defined(RTO_TRI_STAMP)
    RTO_TS(6, RTO_TS_COUNT(valid));
    int stTrips = trips;
    for (int off = 32; off > 0; off >>= 1) stTrips = max(stTrips, __shfl_xor(stTrips, off));
    if (Sc.stamp && MODE == kModeColor && lane == 0 && ty < P.tilesY) {
        unsigned* rec = Sc.stamp + (size_t)tile * 24;
#pragma unroll
        for (int i = 0; i < 7; i++) { rec[i * 3] = stTicks[i]; rec[i * 3 + 1] = stEnter[i]; rec[i * 3 + 2] = stLanes[i]; }
        rec[21] = (unsigned)stTrips | (stWaveTrips << 16); rec[22] = (unsigned)chunks; rec[23] = (unsigned)rounds | 0x80000000u;
    }
#endif
#undef RTO_TS
}

template <int MODE, bool SHADE>
__global__ __launch_bounds__(kBlock, RTO_TRI_WAVES) void k_trace_lean_triangles(RenderParams P, LeanTriScene Sc, int shadow, float4* __restrict__ out,
                                                                                 Counters* __restrict__ counters) {
    extern __shared__ uint2 lds_stack[];   // [wave][level][lane], then the keys of the triangle rounds [wave][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;    // entry b = stk[b * 64], b in [0, depth]
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds_stack + (size_t)(blockDim.x >> 6) * (P.depth + 1) * kWave) + wave * kWave;
    if ((int)blockIdx.x < P.maskBlocks) { mask_block(P, (int)blockIdx.x, reinterpret_cast<unsigned*>(lds_stack), P.maskLdsBytes >> 2); return; }
    const int slot = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - P.maskBlocks) * (int)(blockDim.x >> 6) + wave);
    if (slot >= P.launchWaves) return;
    trace_tile_lean_triangles<MODE, SHADE>(P, Sc, shadow, out, counters, stk, keys, lane, slot);
}

// Several frames (or the parts of several frames: a rank of the multi-GPU split) in one launch, as k_trace_lean_batch: a
// part's kernel lasts as long as its deepest tile (~0.24 ms at config 5 whatever the part's share of the pixels).
template <bool SHADE>
__global__ __launch_bounds__(kBlock, RTO_TRI_WAVES) void k_trace_lean_triangles_batch(RenderBatch B, LeanTriScene Sc, int shadow) {
    extern __shared__ uint2 lds_stack[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int mb = B.P[0].maskBlocks;
    if ((int)blockIdx.x < mb * B.n) { const int fm = (int)blockIdx.x / mb; mask_block(B.P[fm], (int)blockIdx.x - fm * mb, reinterpret_cast<unsigned*>(lds_stack), B.P[fm].maskLdsBytes >> 2); return; }
    const int g = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - mb * B.n) * (int)(blockDim.x >> 6) + wave);
    const int slot = g / B.n, f = g - slot * B.n;
    const RenderParams& P = B.P[f];
    uint2* stk = lds_stack + (size_t)wave * (P.depth + 1) * kWave + lane;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds_stack + (size_t)(blockDim.x >> 6) * (P.depth + 1) * kWave) + wave * kWave;
    if (slot >= P.launchWaves) return;
    trace_tile_lean_triangles<kModeColor, SHADE>(P, Sc, shadow, B.out[f], nullptr, stk, keys, lane, slot);
}

// ================================================================ N1: octreeRaySkip
// Iterative form of the reference's recursive octreeRaySkip (453-skeleton/VolumeRaycastRenderer.cpp:50-155, "S/VR"):
// children are tried in order of increasing Hamming distance from the octant of the ray's positive
// direction bits (ties by octant index, S/VR:122-152); the first child whose subtree returns a finite
// distance ends the search.  One thread per ray over the 60-byte node array; `vis` (optional) plays the
// role of the reference's visibility map (S/VR:64-67): a node flagged 0 returns 1e30.
struct SkipFrame { int idx; float enterT, exitT; int next; };

__global__ __launch_bounds__(kBlock) void k_octree_ray_skip(const rto_node* __restrict__ nodes, const uint8_t* __restrict__ vis,
                                                             float gx, float gy, float gz, float vx,
                                                             float rox, float roy, float roz,
                                                             const float* __restrict__ rd, int64_t n, float tMin0, float tMax0,
                                                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float dx = rd[3 * i], dy = rd[3 * i + 1], dz = rd[3 * i + 2];
    // S/VR:81-87
    float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    const float smallValue = 1e-10f;
    if (__builtin_fabsf(dx) < smallValue) ix = dx >= 0 ? 1e10f : -1e10f;
    if (__builtin_fabsf(dy) < smallValue) iy = dy >= 0 ? 1e10f : -1e10f;
    if (__builtin_fabsf(dz) < smallValue) iz = dz >= 0 ? 1e10f : -1e10f;
    const int dirMask = ((dx > 0) ? 1 : 0) | ((dy > 0) ? 2 : 0) | ((dz > 0) ? 4 : 0);   // S/VR:114-116

    SkipFrame st[kMaxDepth + 2];
    int sp = 0;
    st[0].idx = 0; st[0].enterT = tMin0; st[0].exitT = tMax0; st[0].next = -1;   // next == -1: node not examined yet
    float result = 1e30f;
    while (sp >= 0) {
        SkipFrame& f = st[sp];
        if (f.next < 0) {
            // first visit of this node: S/VR:60-110
            const rto_node nd = nodes[f.idx];
            bool reject = (vis != nullptr) && (vis[f.idx] == 0);
            const float wx0 = gx + (float)nd.x * vx, wy0 = gy + (float)nd.y * vx, wz0 = gz + (float)nd.z * vx;
            const float wSize = (float)nd.size * vx;
            const float t1x = (wx0 - rox) * ix, t1y = (wy0 - roy) * iy, t1z = (wz0 - roz) * iz;
            const float t2x = ((wx0 + wSize) - rox) * ix, t2y = ((wy0 + wSize) - roy) * iy, t2z = ((wz0 + wSize) - roz) * iz;
            const float tNx = gmin(t1x, t2x), tNy = gmin(t1y, t2y), tNz = gmin(t1z, t2z);
            const float tFx = gmax(t1x, t2x), tFy = gmax(t1y, t2y), tFz = gmax(t1z, t2z);
            const float enterT = gmax(gmax(tNx, tNy), gmax(tNz, f.enterT));
            const float exitT = gmin(gmin(tFx, tFy), gmin(tFz, f.exitT));
            if (!reject && enterT > exitT) reject = true;
            if (reject) { sp--; continue; }                          // returns 1e30 to the parent
            if (nd.isLeaf) {
                if (nd.isSolid) { result = enterT; break; }          // finite: every ancestor returns it at once (S/VR:146-149)
                sp--; continue;
            }
            f.enterT = enterT; f.exitT = exitT; f.next = 0;
        }
        // next child in (Hamming distance, octant) order
        int child = -1;
        while (f.next < 8) {
            const int k = f.next++;
            // k-th octant in the order: distance 0 (1), 1 (3), 2 (3), 3 (1); within a distance by octant index
            int octant = -1, seen = 0;
            for (int dist = 0; dist <= 3 && octant < 0; dist++)
                for (int o = 0; o < 8; o++)
                    if (__builtin_popcount(o ^ dirMask) == dist) { if (seen == k) { octant = o; break; } seen++; }
            const rto_node* nd = nodes + f.idx;
            child = nd->child[octant];
            if (child >= 0) break;
            child = -1;
        }
        if (child < 0) { sp--; continue; }                           // all children returned 1e30
        const float eT = f.enterT, xT = f.exitT;
        sp++;
        st[sp].idx = child; st[sp].enterT = eT; st[sp].exitT = xT; st[sp].next = -1;
    }
    out[i] = result;
}

// The same search on the child descriptors (canonical octrees): one 8-byte descriptor per visited internal node
// instead of one dependent 60-byte gather per visited node, and the 8 child intervals come from registers.  The
// reference calls this for 49 probe rays per frame (S/VR:1602-1647), so what matters is the latency of one ray.
// Identical float operations per child as k_octree_ray_skip (box from gridMin + float(c)*voxel, interval clipped by
// the parent's); `useVis` applies the descriptors' visibility masks (what rto_update_frustum last computed).

__device__ __forceinline__ bool skip_interval(float g, float vs, float o, float inv, int c, int size, float& tN, float& tF) {
    const float w0 = g + (float)c * vs;
    const float wSize = (float)size * vs;
    const float t1 = (w0 - o) * inv, t2 = ((w0 + wSize) - o) * inv;
    tN = gmin(t1, t2); tF = gmax(t1, t2);
    return true;
}

// ---------------------------------------------------------------- N1 as a render mode, and N1's consumer
// skip_traverse: octreeRaySkip's search on the descriptor tree with the per-ray frames in LDS ([level][lane], 16 bytes each) instead
// of a private array, and an O(1) return to the deepest level that still has untried children (`pend`), so that a whole
// frame of rays (rto_render_skip_device: "nearest hit" as SURVEY.md section 8f asks for it) runs at a useful rate.  Same float
// operations per child, same child order (Hamming distance from the octant of the positive direction bits, ties by octant
// index, S/VR:122-152), same visibility rule; pinned per pixel by the reference's compiled function (ref_ray_skip.npz).
// Entry of the node at depth L (children of edge rootSize >> (L + 1)):
//   .x = untried candidates in TRAVERSAL order (8) | solid mask << 8 | internal mask << 16    .y = first internal child
//   .z / .w = the node's clipped interval [enterT, exitT]
struct SkipRay { float ox, oy, oz, ix, iy, iz; unsigned order; unsigned sx, sy, sz; bool finite; int dirMask; };

// Child order of octreeRaySkip as a table: for the octant of the ray's positive direction bits (dirMask) and a mask of
// candidate children, the same candidates with bit p = the p-th child in the reference's order (Hamming distance from dirMask,
// ties by octant index: S/VR:122-131).  8 x 256 bytes, copied into LDS by the kernels that walk (one ds_read_u8 per visited
// node instead of 27 VALU instructions of shifts and masks).
struct SkipPermTable { unsigned char v[8 * 256]; };
constexpr SkipPermTable make_skip_perm_table() {
    SkipPermTable t{};
    for (int m = 0; m < 8; m++) {
        int order[8] = {};
        int p = 0;
        for (int dist = 0; dist <= 3; dist++)
            for (int o = 0; o < 8; o++) {
                int diff = o ^ m, bits = 0;
                while (diff) { bits += diff & 1; diff >>= 1; }
                if (bits == dist) order[p++] = o;
            }
        for (int cand = 0; cand < 256; cand++) {
            unsigned om = 0;
            for (int q = 0; q < 8; q++) om |= (unsigned)((cand >> order[q]) & 1) << q;
            t.v[m * 256 + cand] = (unsigned char)om;
        }
    }
    return t;
}
__device__ const SkipPermTable kSkipPerm = make_skip_perm_table();

// copies the table into `lut` (2 KB of LDS); every thread of the workgroup calls it, then the workgroup synchronises
__device__ __forceinline__ void load_skip_perm(unsigned char* lut) {
    const uint2* src = reinterpret_cast<const uint2*>(kSkipPerm.v);
    uint2* dst = reinterpret_cast<uint2*>(lut);
    for (int i = (int)threadIdx.x; i < 8 * 256 / 8; i += (int)blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ SkipRay skip_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
    SkipRay r;
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;                  // S/VR:81-87
    const float smallValue = 1e-10f;
    if (__builtin_fabsf(dx) < smallValue) r.ix = dx >= 0 ? 1e10f : -1e10f;
    if (__builtin_fabsf(dy) < smallValue) r.iy = dy >= 0 ? 1e10f : -1e10f;
    if (__builtin_fabsf(dz) < smallValue) r.iz = dz >= 0 ? 1e10f : -1e10f;
    const int dirMask = ((dx > 0) ? 1 : 0) | ((dy > 0) ? 2 : 0) | ((dz > 0) ? 4 : 0);   // S/VR:114-116
    unsigned order = 0;                                                    // the 8 octants by (Hamming distance from dirMask, octant index), 3 bits each
    int p = 0;
#pragma unroll
    for (int dist = 0; dist <= 3; dist++)
#pragma unroll
        for (int o = 0; o < 8; o++)
            if (__builtin_popcount(o ^ dirMask) == dist) { order |= (unsigned)o << (3 * p); p++; }
    r.order = order;
    r.dirMask = dirMask;
    r.sx = (unsigned)((int)__float_as_uint(r.ix) >> 31); r.sy = (unsigned)((int)__float_as_uint(r.iy) >> 31); r.sz = (unsigned)((int)__float_as_uint(r.iz) >> 31);
    r.finite = __builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) && __builtin_isfinite(ox) && __builtin_isfinite(oy) && __builtin_isfinite(oz);
    return r;
}

// returns the distance (1e30: nothing); on a hit lx, ly, lz, ls = the solid leaf's position and edge
__device__ __forceinline__ float skip_traverse(const uint2* __restrict__ desc, const uint8_t* __restrict__ vis, bool useVis, int rootSize,
                                               float gx, float gy, float gz, float vs, const SkipRay& r, float tMin0, float tMax0,
                                               uint4* stk /* this lane's column: entry(L) = stk[L * 64] */, int& lx, int& ly, int& lz, int& ls,
                                               int* visits = nullptr /* internal nodes entered (the render mode's tile cost) */,
                                               const unsigned char* perm = nullptr /* kSkipPerm in LDS, or NULL */) {
    // interval of the box (bx, by, bz) of edge `size`, clipped by [pe, px]: the operations of S/VR:70-100
    auto interval = [&](int bx, int by, int bz, int size, float pe, float px, float& enterT, float& exitT) {
        float tNx, tFx, tNy, tFy, tNz, tFz;
        skip_interval(gx, vs, r.ox, r.ix, bx, size, tNx, tFx);
        skip_interval(gy, vs, r.oy, r.iy, by, size, tNy, tFy);
        skip_interval(gz, vs, r.oz, r.iz, bz, size, tNz, tFz);
        enterT = gmax(gmax(tNx, tNy), gmax(tNz, pe));
        exitT = gmin(gmin(tFx, tFy), gmin(tFz, px));
    };
    // The same interval for finite rays in a third of the instructions: per axis the smaller / larger of (t1, t2) SELECTED by the
    // sign of the reciprocal (hi >= lo and the operations are monotonic: the same float, see child_axis_terms), the 4-way
    // max / min as v_max3 + v_max.  The hardware's max / min differ from the reference's (a < b) ? b : a only in the sign of
    // a zero result, and only enterT is ever handed out: a zero enterT is recomputed with the reference's form.
    auto interval_fast = [&](int bx, int by, int bz, int size, float pe, float px, float& enterT, float& exitT) {
        const float wSize = (float)size * vs;
        float n[3], f[3];
        const float g3[3] = { gx, gy, gz }, o3[3] = { r.ox, r.oy, r.oz }, i3[3] = { r.ix, r.iy, r.iz };
        const unsigned s3[3] = { r.sx, r.sy, r.sz };
        const int b3[3] = { bx, by, bz };
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float w0 = g3[a] + (float)b3[a] * vs;
            const unsigned t1 = __float_as_uint((w0 - o3[a]) * i3[a]), t2 = __float_as_uint(((w0 + wSize) - o3[a]) * i3[a]);
            n[a] = __uint_as_float(bop3<kSelC>(t1, t2, s3[a])); f[a] = __uint_as_float(bop3<kSelC>(t2, t1, s3[a]));
        }
        float en, ex;
        asm("v_max_f32 %0, %1, %2" : "=v"(en) : "v"(max3f(n[0], n[1], n[2])), "v"(pe));
        asm("v_min_f32 %0, %1, %2" : "=v"(ex) : "v"(min3f(f[0], f[1], f[2])), "v"(px));
        enterT = en; exitT = ex;
        if (en == 0.0f) interval(bx, by, bz, size, pe, px, enterT, exitT);      // rare: the sign of the zero is the reference's
    };
    float e, x;
    interval(0, 0, 0, rootSize, tMin0, tMax0, e, x);
    // wave-uniform: every ray of the wave is finite (then no NaN / inf - inf can arise in the children's verdicts)
    const bool allFinite = __builtin_amdgcn_ballot_w64(!(r.finite && __builtin_isfinite(e) && __builtin_isfinite(x) && __builtin_fabsf(e) < 1e30f)) == 0ull;
    if ((useVis && vis[0] == 0) || e > x) return 1e30f;
    unsigned pend = 0;                       // bit L: the entry of depth L still has untried candidates
    int level = 0, cx = 0, cy = 0, cz = 0;   // the node being entered
    unsigned cur = 0;
    for (;;) {
        {   // enter the internal node `cur` at (cx, cy, cz), depth `level`, interval [e, x]
            if (visits) ++*visits;
            const uint2 d = desc[cur];
            const int half = rootSize >> (level + 1);
            const unsigned sm = d.x & 0xffu, im = (d.x >> 8) & 0xffu, vm = useVis ? ((d.x >> 16) & 0xffu) : 0xffu;
            unsigned pass = 0;
            if (allFinite) {
                // which children have a non-empty interval: the arithmetic of the lean kernel's child test (same box expressions as
                // S/VR:70-97; min / max per axis selected by the sign of the reciprocal; v_min3 / v_max3), the parent's interval
                // folded in.  Only the VERDICTS are taken from here -- they do not depend on the sign of a zero; the distance that
                // is returned is recomputed for the one child that is entered with the reference's own min / max below.
                pass = ~child_fail_mask_fast<true, false>(gx, gy, gz, vs, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.sx, r.sy, r.sz, cx, cy, cz, (float)half, e, x) & 0xffu;
            } else {
                float tN[3][2], tF[3][2];
                skip_interval(gx, vs, r.ox, r.ix, cx, half, tN[0][0], tF[0][0]); skip_interval(gx, vs, r.ox, r.ix, cx + half, half, tN[0][1], tF[0][1]);
                skip_interval(gy, vs, r.oy, r.iy, cy, half, tN[1][0], tF[1][0]); skip_interval(gy, vs, r.oy, r.iy, cy + half, half, tN[1][1], tF[1][1]);
                skip_interval(gz, vs, r.oz, r.iz, cz, half, tN[2][0], tF[2][0]); skip_interval(gz, vs, r.oz, r.iz, cz + half, half, tN[2][1], tF[2][1]);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float ce = gmax(gmax(tN[0][k & 1], tN[1][(k >> 1) & 1]), gmax(tN[2][k >> 2], e));
                    const float cxit = gmin(gmin(tF[0][k & 1], tF[1][(k >> 1) & 1]), gmin(tF[2][k >> 2], x));
                    pass |= (ce > cxit) ? 0u : (1u << k);
                }
            }
            const unsigned cand = pass & vm & (sm | im);
            unsigned om = 0;
            if (perm) om = perm[r.dirMask * 256 + (int)cand];
            else {
#pragma unroll
                for (int p = 0; p < 8; p++) om |= ((cand >> ((r.order >> (3 * p)) & 7u)) & 1u) << p;
            }
            stk[level * kWave] = make_uint4(om | (sm << 8) | (im << 16), d.y, __float_as_uint(e), __float_as_uint(x));
            pend = (pend & ~(1u << level)) | (om ? (1u << level) : 0u);
        }
        if (pend == 0) return 1e30f;                                     // every branch returned 1e30
        const int L = 31 - __builtin_clz(pend);                          // the deepest node with an untried child: where the recursion continues
        uint4 en = stk[L * kWave];
        const int p = __builtin_ctz(en.x & 0xffu);
        const int k = (int)((r.order >> (3 * p)) & 7u);
        en.x &= ~(1u << p);
        stk[L * kWave].x = en.x;
        if ((en.x & 0xffu) == 0) pend &= ~(1u << L);
        const int edge = rootSize >> L, half = edge >> 1;                // the node at depth L and its children
        const int nx = cx & ~(edge - 1), ny = cy & ~(edge - 1), nz = cz & ~(edge - 1);
        const int bx = nx + ((k & 1) ? half : 0), by = ny + ((k & 2) ? half : 0), bz = nz + ((k & 4) ? half : 0);
        if (allFinite) interval_fast(bx, by, bz, half, __uint_as_float(en.z), __uint_as_float(en.w), e, x);
        else interval(bx, by, bz, half, __uint_as_float(en.z), __uint_as_float(en.w), e, x);
        if ((en.x >> (8 + k)) & 1u) { lx = bx; ly = by; lz = bz; ls = half; return e; }      // solid leaf: finite, every ancestor returns it (S/VR:146-149)
        cur = en.y + (unsigned)__builtin_popcount((en.x >> 16) & 0xffu & ((1u << k) - 1u));
        cx = bx; cy = by; cz = bz; level = L + 1;
    }
}

// rto_octree_ray_skip on a canonical tree: one thread per ray (shared origin), the per-ray frames in LDS like every other user of
// skip_traverse -- until round 4 this kernel kept them in a private array: 688 bytes of scratch per thread.
__global__ __launch_bounds__(kBlock) void k_octree_ray_skip_packed(const uint2* __restrict__ desc, const uint8_t* __restrict__ vis, int useVis,
                                                                    int rootSize, int depth,
                                                                    float gx, float gy, float gz, float vx,
                                                                    float rox, float roy, float roz,
                                                                    const float* __restrict__ rd, int64_t n, float tMin0, float tMax0,
                                                                    float* __restrict__ out) {
    extern __shared__ uint4 lds_skip[];    // [wave][level][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* stk = lds_skip + (size_t)wave * depth * kWave + lane;
    __shared__ unsigned char permLut[8 * 256];
    load_skip_perm(permLut);
    __syncthreads();                                                     // before any thread leaves
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const SkipRay r = skip_ray(rox, roy, roz, rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    int lx, ly, lz, ls;
    out[i] = skip_traverse(desc, vis, useVis != 0, rootSize, gx, gy, gz, vx, r, tMin0, tMax0, stk, lx, ly, lz, ls, nullptr, permLut);
}

// One thread per pixel (a wave = an 8x8 tile): distance of the ray of generateRay through octreeRaySkip(root, ro, rd, 0, 1e30)
// and, optionally, the reference's shade (S/RT:283-285, 331-336) of the leaf that distance belongs to, at tHit = that distance.
__global__ __launch_bounds__(kBlock) void k_skip_render(RenderParams P, const uint2* __restrict__ desc, const uint8_t* __restrict__ vis, int useVis,
                                                         float4* __restrict__ outRGBA, float* __restrict__ outT) {
    extern __shared__ uint4 lds_skip[];    // [wave][level][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* stk = lds_skip + (size_t)wave * P.depth * kWave + lane;
    __shared__ unsigned char permLut[8 * 256];
    load_skip_perm(permLut);
    __syncthreads();                                                     // before any wave leaves
    // launch geometry of the lean kernels: the first workgroups build the occupancy mask, waves only for the tiles of the solid
    // geometry's screen rectangle in the stream's launch order (costliest tiles of earlier frames first), wide stores for the rest
    if ((int)blockIdx.x < P.maskBlocks) { mask_block(P, (int)blockIdx.x, reinterpret_cast<unsigned*>(lds_skip), P.maskLdsBytes >> 2); return; }
    const int slot = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - P.maskBlocks) * (kBlock / kWave) + wave);
    if (slot >= P.launchWaves) return;
    int tile, tx, ty;
    resolve_slot(P, slot, tx, ty, tile);
    const int px = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
    const bool valid = (ty < P.tilesY) && (px < P.W) && (ly < P.localRows);
    const int py = global_row(P, ly);
    const bool inImage = valid && py < P.H;
    // outside the solid leaves' screen rectangle (widened by a voxel) no ray meets a solid leaf: the search returns 1e30
    const bool outside = px < P.rootX0 || px > P.rootX1 || py < P.rootY0 || py > P.rootY1;
    const Geo G = geo_of(P);
    float t = 1e30f;
    int lx = 0, lyy = 0, lz = 0, ls = 0, visits = 0;
    Ray g;
    g.ox = g.oy = g.oz = g.dx = g.dy = g.dz = 0.0f;
    const bool tileLive = tile_may_hit(P, tx, ty, slot, lane, tile);     // wave-uniform; a tile outside the mask records its cost there (0, or -1 on the rim)
    if (inImage && !outside && tileLive) {
        g = generate_ray_tab(P, px, py);
        const SkipRay r = skip_ray(g.ox, g.oy, g.oz, g.dx, g.dy, g.dz);
        t = skip_traverse(desc, vis, useVis != 0, P.rootSize, G.gx, G.gy, G.gz, G.vs, r, 0.0f, 1e30f, stk, lx, lyy, lz, ls, &visits, permLut);
    }
    if (P.tileCost) {                                                    // this tile's cost for the launch order: the visits of its busiest ray
        int cst = visits;
        for (int off = 32; off > 0; off >>= 1) cst = max(cst, __shfl_xor(cst, off));
        if (lane == 0 && ty < P.tilesY && tileLive) P.tileCost[tile] = (cst == 0 && P.tileMask) ? -1 : cst;
    }
    if (inImage && !(P.skipOutside && outside)) {
        const size_t pix = (size_t)ly * P.W + px;
        if (outT) __builtin_nontemporal_store(t, outT + pix);
        if (outRGBA) {
            float4 color = make_float4(0.f, 0.f, 0.f, 1.f);
            if (t < 1e30f) {
                const float mnx = G.gx + (float)lx * G.vs, mny = G.gy + (float)lyy * G.vs, mnz = G.gz + (float)lz * G.vs;
                const float ext = (float)ls * G.vs;
                const float cx = 0.5f * (mnx + (mnx + ext)), cy = 0.5f * (mny + (mny + ext)), cz = 0.5f * (mnz + (mnz + ext));
                const float qx = (g.ox + g.dx * t) - cx, qy = (g.oy + g.dy * t) - cy, qz = (g.oz + g.dz * t) - cz;
                const float inv = inversesqrt(qx * qx + qy * qy + qz * qz);
                color = shade_color(gmax(0.0f, (qx * inv) * P.lightNeg[0] + (qy * inv) * P.lightNeg[1] + (qz * inv) * P.lightNeg[2]));
            }
            store_pixel(outRGBA + pix, color);
        }
    }
    if (P.skipOutside)
        for_each_outside_pixel(P, lane, slot, [&](size_t pix) {
            if (outT) __builtin_nontemporal_store(1e30f, outT + pix);
            if (outRGBA) store_pixel(outRGBA + pix, make_float4(0.f, 0.f, 0.f, 1.f));
        });
}

// octreeRaySkip's consumer in drawRaycast (S/VR:1602-1663) in ONE launch of one wave, nothing copied: lane i < 49 makes the
// i-th probe direction (the reference's glm operations on inverse(P), inverse(V): both pixel independent, from the host),
// walks the tree, the wave picks the value std::sort would leave at index int(n * 0.15f) among the valid distances by
// ranking (no sort needed for one order statistic), x 0.75, blended 0.4 old + 0.6 new into *skip (device memory).
struct ProbeParams { float invP[16], invV[16]; float eye[3]; float gx, gy, gz, vs; int rootSize, depth; };

__global__ __launch_bounds__(kWave) void k_probe_skip(ProbeParams Q, const uint2* __restrict__ desc, const uint8_t* __restrict__ vis, int useVis,
                                                       float* __restrict__ skip, float* __restrict__ probeT /* optional: the 49 distances */) {
    extern __shared__ uint4 lds_skip[];
#if defined(RTO_PROBE_STAMP)       // A/B build (tools/probe_stamp.py): skip[1..3] = us spent in set-up / traversal / rank + blend
    const unsigned long long st0 = wall_clock64();
#endif
    const int lane = threadIdx.x;
    uint4* stk = lds_skip + lane;
    __shared__ unsigned char permLut[8 * 256];
    // the table's 2 KB: 32 bytes per lane, fetched first and stored after the ray set-up (a launch's first loads are its slowest)
    const uint4 lutA = reinterpret_cast<const uint4*>(kSkipPerm.v)[lane * 2], lutB = reinterpret_cast<const uint4*>(kSkipPerm.v)[lane * 2 + 1];
    const uint2 rootDesc = desc[0];                                       // warms the root's line while the rays are set up
    const float previous = *skip;                                         // the value the blend needs at the very end: asked for now
    float t = 1e30f;
    // the probe's direction as three plain floats: a SkipRay that lives across the barrier below was kept in SCRATCH (24 bytes: two
    // scratch stores at the kernel's head, two before the rank step, the loads in front of the traversal -- in a one-wave latency chain)
    float pdx = 1.f, pdy = 1.f, pdz = 1.f;
    if (lane < 49) {
        const int gridSize = 7;
        const float sampleOffset = 0.2f;
        const int x = lane % gridSize, y = lane / gridSize;
        const float ndcX = ((float)x / (float)(gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
        const float ndcY = ((float)y / (float)(gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
        const float c[4] = { ndcX, ndcY, 1.f, 1.f };
        float vp[4], wp[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) vp[rr] = (Q.invP[rr] * c[0] + Q.invP[4 + rr] * c[1]) + (Q.invP[8 + rr] * c[2] + Q.invP[12 + rr] * c[3]);
        const float w = vp[3];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) vp[rr] = vp[rr] / w;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) wp[rr] = (Q.invV[rr] * vp[0] + Q.invV[4 + rr] * vp[1]) + (Q.invV[8 + rr] * vp[2] + Q.invV[12 + rr] * vp[3]);
        const float qx = wp[0] - Q.eye[0], qy = wp[1] - Q.eye[1], qz = wp[2] - Q.eye[2];
        const float inv = inversesqrt(qx * qx + qy * qy + qz * qz);
        pdx = qx * inv; pdy = qy * inv; pdz = qz * inv;
    }
    reinterpret_cast<uint4*>(permLut)[lane * 2] = lutA; reinterpret_cast<uint4*>(permLut)[lane * 2 + 1] = lutB;
    asm volatile("" :: "v"(rootDesc.x), "v"(rootDesc.y));
    __syncthreads();
#if defined(RTO_PROBE_STAMP)
    const unsigned long long st1 = wall_clock64();
#endif
    if (lane < 49) {
        const SkipRay r = skip_ray(Q.eye[0], Q.eye[1], Q.eye[2], pdx, pdy, pdz);
        int a, b, cc, dd;
        t = skip_traverse(desc, vis, useVis != 0, Q.rootSize, Q.gx, Q.gy, Q.gz, Q.vs, r, 0.0f, 1e30f, stk, a, b, cc, dd, nullptr, permLut);
        if (probeT) probeT[lane] = t;
    }
#if defined(RTO_PROBE_STAMP)
    const unsigned long long st2 = wall_clock64();
#endif
    const bool ok = lane < 49 && t < 1e30f && t > 0.0f;                  // S/VR:1640-1642
    const unsigned long long okMask = __builtin_amdgcn_ballot_w64(ok);
    const int nv = __builtin_popcountll(okMask);
    // rank of every valid distance: probe j's value comes down through v_readlane (a scalar, 4 cycles) -- as 49 ds_bpermute round
    // trips this step was 4.3 us of the call's 16.9 (tools/probe_stamp.py)
    // A valid distance is positive: its bits order like its value, and (bits, probe number) as ONE 64-bit key orders the probes the
    // way the comparisons of std::sort's result would -- one v_cmp_lt_u64 per probe, no branch; an invalid probe's key is all ones.
    int rank = 0;
    const int tBits = __float_as_int(t);
    const unsigned long long myKey = ((unsigned long long)(unsigned)tBits << 32) | (unsigned)lane;
#pragma unroll
    for (int j = 0; j < 49; j++) {
        const unsigned hi = ((okMask >> j) & 1ull) ? (unsigned)__builtin_amdgcn_readlane(tBits, j) : 0xffffffffu;   // scalar select
        rank += ((((unsigned long long)hi << 32) | (unsigned)j) < myKey) ? 1 : 0;
    }
    int safeIndex = (int)((float)nv * 0.15f);                            // S/VR:1650
    if (safeIndex < 0) safeIndex = 0;
    const unsigned long long pick = __builtin_amdgcn_ballot_w64(ok && rank == safeIndex);
    float skipDistance = 0.0f;
    if (nv > 0) skipDistance = __int_as_float(__builtin_amdgcn_readlane(tBits, (int)__builtin_ctzll(pick))) * 0.75f;   // :1651-1654
    const float blendFactor = 0.4f;                                      // :1659-1661
    if (lane == 0) *skip = previous * blendFactor + skipDistance * (1.0f - blendFactor);
#if defined(RTO_PROBE_STAMP)
    if (lane == 0) { const unsigned long long st3 = wall_clock64(); skip[1] = (float)(st1 - st0) * 0.01f; skip[2] = (float)(st2 - st1) * 0.01f; skip[3] = (float)(st3 - st2) * 0.01f; }
#endif
}

// ================================================================ N4: octree construction on the GPU
// createOctreeFromVoxelGrid + setOctree (453-skeleton/OctreeVoxel.cpp:704-778, RayTracerBVH.cpp:443-490) without
// a pointer tree: (1) bottom-up occupancy pyramid over the voxel grid (state 0 = all EMPTY incl. the
// out-of-grid part, 1 = all FILLED, 2 = mixed); (2) level-order emission -- the BFS numbering of setOctree IS
// level order with the children of a level's internal nodes appended in node order -- using one flag scan
// per level; (3) node records and child descriptors written straight into HBM.
struct PyramidView {
    const uint8_t* level[kMaxDepth + 1];   // level[0] = the voxels
    int nx[kMaxDepth + 1], ny[kMaxDepth + 1], nz[kMaxDepth + 1];
};

__device__ __forceinline__ int pyr_state(const PyramidView& V, int lv, int x, int y, int z) {
    const int i = x >> lv, j = y >> lv, k = z >> lv;
    if (i >= V.nx[lv] || j >= V.ny[lv] || k >= V.nz[lv]) return 0;   // wholly outside the grid: EMPTY (getVoxelSafe)
    return V.level[lv][(size_t)i + (size_t)j * V.nx[lv] + (size_t)k * V.nx[lv] * V.ny[lv]];
}

// one thread per cell of level lv >= 1; children at level lv-1 (voxels for lv == 1)
// blockMixed[block] = number of mixed cells the block produced.  A mixed cell of pyramid level lv IS an internal node
// of the tree (its ancestors are mixed too), so these counts give every tree level's size before the tree is walked.
__global__ __launch_bounds__(kBlock) void k_pyramid_level(const uint8_t* __restrict__ child, int cnx, int cny, int cnz,
                                                           uint8_t* __restrict__ out, int nx, int ny, int nz,
                                                           int ext, int dimX, int dimY, int dimZ, int* __restrict__ blockMixed) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t total = (size_t)nx * ny * nz;
    bool isMixed = false;
    if (t < total) {
        const int i = (int)(t % nx), j = (int)((t / nx) % ny), k = (int)(t / ((size_t)nx * ny));
        // the cell sticks out of the grid (its out-of-grid voxels are EMPTY) iff its extent passes a grid dimension
        bool any0 = (i + 1) * (long)ext > dimX || (j + 1) * (long)ext > dimY || (k + 1) * (long)ext > dimZ;
        bool any1 = false, mixed = false;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int ci = 2 * i + (c & 1), cj = 2 * j + ((c >> 1) & 1), ck = 2 * k + (c >> 2);
            if (ci >= cnx || cj >= cny || ck >= cnz) continue;
            const uint8_t s = child[(size_t)ci + (size_t)cj * cnx + (size_t)ck * cnx * cny];
            mixed |= (s == 2); any1 |= (s == 1); any0 |= (s == 0);
        }
        isMixed = mixed || (any0 && any1);
        out[t] = isMixed ? 2 : (any1 ? 1 : 0);
    }
    const int cnt = __syncthreads_count(isMixed ? 1 : 0);
    if (threadIdx.x == 0) blockMixed[blockIdx.x] = cnt;
}

// Level 1 straight from the voxels, 16 B per load: one thread = 8 consecutive cells along x = 16 voxels of each of the
// four (y, z) rows under them.  Same result as k_pyramid_level; needs dimX % 16 == 0 (rows stay 16-byte aligned).
__global__ __launch_bounds__(kBlock) void k_pyramid_level1_wide(const uint8_t* __restrict__ vox, int dimX, int dimY, int dimZ,
                                                                 uint8_t* __restrict__ out, int nx, int ny, int nz,
                                                                 int* __restrict__ blockMixed) {
    __shared__ int waveTotal[kBlock / kWave];
    const int groupsX = nx / 8;
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t total = (size_t)groupsX * ny * nz;
    int mixedCount = 0;
    if (t < total) {
        const int gx = (int)(t % groupsX), j = (int)((t / groupsX) % ny), k = (int)(t / ((size_t)groupsX * ny));
        const bool overhang = (j + 1) * 2 > dimY || (k + 1) * 2 > dimZ;     // out-of-grid voxels are EMPTY
        uint4 row[4];
        bool have[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int y = 2 * j + (r & 1), z = 2 * k + (r >> 1);
            have[r] = y < dimY && z < dimZ;
            row[r] = make_uint4(0, 0, 0, 0);
            if (have[r]) row[r] = *reinterpret_cast<const uint4*>(vox + ((size_t)z * dimY + y) * dimX + (size_t)gx * 16);
        }
        unsigned long long packed = 0;
#pragma unroll
        for (int o = 0; o < 8; o++) {
            bool any0 = overhang, any1 = false, mixed = false;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!have[r]) continue;
                const unsigned w = (o >> 1) == 0 ? row[r].x : (o >> 1) == 1 ? row[r].y : (o >> 1) == 2 ? row[r].z : row[r].w;
                const unsigned a = (w >> ((o & 1) * 16)) & 0xffu, b = (w >> ((o & 1) * 16 + 8)) & 0xffu;
                mixed |= (a == 2) | (b == 2); any1 |= (a == 1) | (b == 1); any0 |= (a == 0) | (b == 0);
            }
            const bool isMixed = mixed || (any0 && any1);
            mixedCount += isMixed ? 1 : 0;
            packed |= (unsigned long long)(isMixed ? 2 : (any1 ? 1 : 0)) << (8 * o);
        }
        *reinterpret_cast<unsigned long long*>(out + ((size_t)k * ny + j) * nx + (size_t)gx * 8) = packed;
    }
    for (int off = 32; off > 0; off >>= 1) mixedCount += __shfl_down(mixedCount, off);
    if ((threadIdx.x & 63) == 0) waveTotal[threadIdx.x >> 6] = mixedCount;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < kBlock / kWave; w++) s += waveTotal[w]; blockMixed[blockIdx.x] = s; }
}

// one block per pyramid level: levelMixed[l] = sum of that level's per-block counts
struct LevelCountView { const int* blockMixed[kMaxDepth + 1]; int numBlocks[kMaxDepth + 1]; };
__global__ __launch_bounds__(1024) void k_sum_level_counts(LevelCountView V, long long* __restrict__ levelMixed) {
    __shared__ long long partial[1024 / kWave];
    const int l = blockIdx.x + 1;
    long long sum = 0;
    for (int i = threadIdx.x; i < V.numBlocks[l]; i += 1024) sum += V.blockMixed[l][i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < 1024 / kWave; w++) t += partial[w]; levelMixed[l] = t; }
}

// classify the nodes of one tree level: internal flag + per-block internal count
__global__ __launch_bounds__(kBlock) void k_build_classify(PyramidView V, const int4* __restrict__ coords, int64_t m, int lv,
                                                            uint8_t* __restrict__ state, uint8_t* __restrict__ flag,
                                                            int* __restrict__ blockCount) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool internal = false;
    if (i < m) {
        const int4 c = coords[i];
        const int s = pyr_state(V, lv, c.x, c.y, c.z);
        internal = (lv > 0) && (s == 2);
        state[i] = (uint8_t)s;
        flag[i] = internal ? 1 : 0;
    }
    const int cnt = __syncthreads_count(internal ? 1 : 0);
    if (threadIdx.x == 0) blockCount[blockIdx.x] = cnt;
}

// rank of the internal nodes of a level (exclusive count of the flags before them, -1 for leaves) and, in the same
// pass, the coordinates of their children = the next level's node list
__global__ __launch_bounds__(kBlock) void k_build_rank_children(const uint8_t* __restrict__ flag, const int4* __restrict__ coords, int64_t m,
                                                                 const int* __restrict__ blockBase, int half, int* __restrict__ rank,
                                                                 int4* __restrict__ next) {
    __shared__ int waveTotal[kBlock / kWave];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool v = (i < m) && flag[i];
    const unsigned long long b = __builtin_amdgcn_ballot_w64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __builtin_popcountll(b & ((1ull << lane) - 1ull));
    if (lane == 0) waveTotal[wave] = __builtin_popcountll(b);
    __syncthreads();
    int base = blockBase[blockIdx.x];
    for (int w = 0; w < wave; w++) base += waveTotal[w];
    if (i >= m) return;
    const int r = v ? base + before : -1;
    rank[i] = r;
    if (r < 0 || !next) return;
    const int4 c = coords[i];
#pragma unroll
    for (int k = 0; k < 8; k++)   // S/OctreeVoxel.cpp:751-754: bit0 -> +x, bit1 -> +y, bit2 -> +z
        next[(size_t)r * 8 + k] = make_int4(c.x + ((k & 1) ? half : 0), c.y + ((k & 2) ? half : 0), c.z + ((k & 4) ? half : 0), 0);
}

// node records (GPUNodes) and child descriptors of one level
__global__ __launch_bounds__(kBlock) void k_build_emit(const int4* __restrict__ coords, const uint8_t* __restrict__ state,
                                                        const int* __restrict__ rank, int64_t m, int size,
                                                        int64_t levelBase,          // flat index of this level's first node
                                                        int64_t internalBase,       // descriptor index of this level's first internal node
                                                        const uint8_t* __restrict__ childState,   // next level (8 per internal node), may be null
                                                        const int* __restrict__ childRank,        // next level's internal ranks (-1 = leaf)
                                                        int64_t nextInternalBase,
                                                        rto_node* __restrict__ nodes, uint2* __restrict__ desc, int* __restrict__ descFirstChild) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int4 c = coords[i];
    const int r = rank[i];
    rto_node nd;
    nd.x = c.x; nd.y = c.y; nd.z = c.z; nd.size = size;
    const bool internal = r >= 0;
    nd.isLeaf = internal ? 0 : 1;
    nd.isUniform = nd.isLeaf;                                  // S/OctreeVoxel.cpp:716-745: leaf <=> uniform
    nd.isSolid = (!internal && state[i] == 1) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 8; k++) nd.child[k] = -1;
    if (internal) {
        const int64_t c0 = levelBase + m + (int64_t)r * 8;     // the next level starts right after this one
        unsigned imask = 0, smask = 0;
        int firstInternal = -1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            nd.child[k] = (int)(c0 + k);
            const int cr = childRank[(size_t)r * 8 + k];
            if (cr >= 0) { imask |= 1u << k; if (firstInternal < 0) firstInternal = cr; }
            else if (childState[(size_t)r * 8 + k] == 1) smask |= 1u << k;
        }
        const int64_t d = internalBase + r;
        desc[d] = make_uint2(smask | (imask << 8) | 0xff0000u, firstInternal >= 0 ? (unsigned)(nextInternalBase + firstInternal) : 0u);
        descFirstChild[d] = (int)c0;
    }
    nodes[levelBase + i] = nd;
}

// ---------------------------------------------------------------- N4, second form: the same tree in FOUR launches
// The level-by-level form above needs ~5 dependent launches per tree level (~50 at 512^3: launch-bound, 0.48 ms).  The BFS
// numbering of setOctree has a closed form that removes the dependency between levels: children are appended in child
// order k = x | y<<1 | z<<2 below parents that are themselves in that order, so WITHIN a tree level the nodes are sorted by
// the Morton code of their cell (x bit lowest).  Hence, with every pyramid level stored in Morton order,
//     rank of a mixed cell among the mixed cells of its level  =  an exclusive prefix sum over that level's array,
//     node index of child k of mixed cell P (level l+1)         =  levelBase + 8 * rank(P) + k,
// and every level can be ranked and emitted at once:
//   k_mb_bricks      voxels -> pyramid levels 1..5 of one 32^3 brick (LDS), written in Morton order, with the number of
//                    mixed children of every cell (cnt); one block per brick that touches the grid
//                    mixed children of every cell (cnt); one block per run of bricks that touches the grid; the sums of cnt
//                    per 1024-cell chunk go to the scan's chunk sums as it goes (integer atomics)
//   k_mb_top_scan    one block: levels 6..R from level 5, then the exclusive scan of the chunk sums per level and the
//                    per-level totals -> node / descriptor bases -> ONE read-back sizes the outputs
//   k_mb_group_ranks exclusive scan of cnt inside every chunk: G[level][p] = mixed cells of the level below that precede
//                    p's first child; cellOf[descriptor] = Morton index of the cell that becomes that internal node
//   k_mb_emit        one thread per CHILD of an internal node (8 per cell): the records of a block are contiguous in the
//                    array (node 1 + 8 d + k), so they are staged in LDS and leave as 16-byte stores
// Bytes: voxels read once (1 B/voxel), pyramid + counts written and read once (2 x 2/7 B/voxel), 60 B per node written.
constexpr int kMbBrickLevels = 5;                 // a brick is 32^3 voxels
constexpr int kMbMaxDepth = 10;                   // Morton arrays of level 1 hold 8^(R-1) bytes: 134 MB at R = 10
constexpr int kMbChunk = 1024;

struct MbLevels {
    uint8_t* state[kMbMaxDepth + 1];              // [l] Morton-ordered states of pyramid level l (1..R): 0 EMPTY, 1 FILLED, 2 mixed
    uint8_t* cnt[kMbMaxDepth + 1];                // [l] mixed children (level l-1) of every cell of level l (2..R)
    int* group[kMbMaxDepth + 1];                  // [l] exclusive scan of cnt[l]: rank of the first child among the mixed cells of level l-1
    long long cntOffset[kMbMaxDepth + 2];         // [l] offset of cnt[l] in the concatenated scan domain (levels padded to kMbChunk)
    long long emitOffset[kMbMaxDepth + 2];        // [l] offset of level l (1..R) in k_mb_emit's thread domain; [R+1] = its size
    int R;
    int dimX, dimY, dimZ;
};

struct MbTables {                                 // written by k_mb_scan_chunks, read by the host (sizes) and by k_mb_emit
    long long levelBase[kMbMaxDepth + 2];         // [L] flat index of tree level L's first node
    long long internalBase[kMbMaxDepth + 2];      // [L] descriptor index of tree level L's first internal node
    long long total, internal;
    int solidLo[3], solidHi[3];                   // voxel box of the level-1 cells that hold FILLED voxels (lo > hi: none)
};

__device__ __forceinline__ unsigned mb_spread3(unsigned v) {   // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ unsigned mb_compact3(unsigned v) {  // every third bit -> 10 bits
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030c30c3u;
    v = (v | (v >> 4)) & 0x0300f00fu;
    v = (v | (v >> 8)) & 0x030000ffu;
    v = (v | (v >> 16)) & 0x3ffu;
    return v;
}
__device__ __forceinline__ unsigned mb_morton(unsigned x, unsigned y, unsigned z) { return mb_spread3(x) | (mb_spread3(y) << 1) | (mb_spread3(z) << 2); }

__device__ __forceinline__ int mb_voxel(const uint8_t* __restrict__ vox, int dimX, int dimY, int dimZ, int x, int y, int z) {
    if (x >= dimX || y >= dimY || z >= dimZ) return 0;                     // getVoxelSafe: outside reads EMPTY
    return vox[(size_t)x + (size_t)y * dimX + (size_t)z * dimX * dimY] == 1 ? 1 : 0;
}

// state of a cell from its 8 children states; *mixedChildren = how many of them are mixed
__device__ __forceinline__ int mb_combine(const int s[8], int* mixedChildren) {
    bool any0 = false, any1 = false;
    int mixed = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { mixed += s[k] == 2; any1 |= s[k] == 1; any0 |= s[k] == 0; }
    *mixedChildren = mixed;
    return (mixed || (any0 && any1)) ? 2 : (any1 ? 1 : 0);
}

// One block per run of kMbRun 32^3-voxel bricks along x (a wave instruction then reads 32 * kMbRun contiguous bytes of a voxel
// row instead of 32: the voxels are x-fastest and a single brick's rows are 32 bytes).  B = min(5, R) levels; a grid
// smaller than a brick has one brick that reaches above the root: cells beyond the root's domain are not written.  LDS
// holds each brick's levels in LOCAL MORTON order, so the 8 children of a cell are 8 consecutive bytes and the write-out is
// a straight 16-byte copy.  blockBox: per block, the box (level-1 cell precision, voxel units) of its cells that hold
// FILLED voxels.
#ifndef RTO_MB_RUN
#define RTO_MB_RUN 2
#endif
constexpr int kMbRun = RTO_MB_RUN;
constexpr int kMbBrickLds = 4096 + 512 + 64 + 16 + 16;       // levels 1..5 of one brick, 16-byte aligned each

// 0x80 in every byte of w that equals 1 (exact: no carries between bytes)
__device__ __forceinline__ unsigned mb_bytes_eq1(unsigned w) {
    const unsigned t = w ^ 0x01010101u;
    return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}

__global__ __launch_bounds__(kBlock) void k_mb_bricks(const uint8_t* __restrict__ vox, MbLevels Lv, int bricksX, int runsX, int bricksY,
                                                       int* __restrict__ blockBox /* [block][6] */, int* __restrict__ chunkSum /* zeroed */) {
    __shared__ __attribute__((aligned(16))) uint8_t st[kMbRun][kMbBrickLds];
    __shared__ __attribute__((aligned(16))) uint8_t ct[kMbRun][kMbBrickLds];          // mixed-children counts (level 1: always 0)
    __shared__ int box[6];
    __shared__ int mixedSum[kMbRun][kMbBrickLevels + 1];                              // [brick][l]: sum of cnt over the brick's cells of level l
    const int R = Lv.R;
    const int B = R < kMbBrickLevels ? R : kMbBrickLevels;
    const int rx = blockIdx.x % runsX, by = (blockIdx.x / runsX) % bricksY, bz = blockIdx.x / (runsX * bricksY);
    const int bx0 = rx * kMbRun;                                                  // first brick of the run
    const int nb = min(kMbRun, bricksX - bx0);                                    // bricks of this run that touch the grid
    const int x0 = bx0 << kMbBrickLevels, y0 = by << kMbBrickLevels, z0 = bz << kMbBrickLevels;
    if (threadIdx.x < 6) box[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : -0x7fffffff;
    if (threadIdx.x < kMbRun * (kMbBrickLevels + 1)) (&mixedSum[0][0])[threadIdx.x] = 0;
    __syncthreads();
    // ---- level 1 from the voxels: one thread = 8 cells along x (16 voxels of four rows); 2 * nb such strips per cell row
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
    const int stripsX = 2 * nb;
    for (int t = threadIdx.x; t < stripsX * 256; t += kBlock) {
        const int s16 = t % stripsX, j = (t / stripsX) & 15, k = t / (stripsX * 16);   // 16-voxel strip along x, cell row, cell layer
        const int sb = s16 >> 1, sx = (s16 & 1) * 8;                                   // brick of the run, strip origin in its level-1 cells
        const int vx = x0 + s16 * 16, vy = y0 + j * 2, vz = z0 + k * 2;
        uint4 row[4];
        const bool vec = (Lv.dimX % 16 == 0) && vx + 16 <= Lv.dimX;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int y = vy + (r & 1), z = vz + (r >> 1);
            row[r] = make_uint4(0, 0, 0, 0);
            if (y < Lv.dimY && z < Lv.dimZ) {
                if (vec) row[r] = *reinterpret_cast<const uint4*>(vox + ((size_t)z * Lv.dimY + y) * Lv.dimX + vx);
                else {
                    unsigned w[4] = { 0, 0, 0, 0 };
                    for (int q = 0; q < 16; q++)
                        if (vx + q < Lv.dimX) w[q >> 2] |= (unsigned)vox[((size_t)z * Lv.dimY + y) * Lv.dimX + vx + q] << ((q & 3) * 8);
                    row[r] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
        // 16 bytes of the four rows at once: a byte is FILLED iff it equals 1, anything else (incl. outside the grid) is EMPTY
        const unsigned mjk = (mb_spread3((unsigned)j) << 1) | (mb_spread3((unsigned)k) << 2) | mb_spread3((unsigned)sx);   // sx is 0 or 8: no carry into o
        unsigned anyMask = 0;                                                   // bit o: cell o holds a FILLED voxel
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned e0 = mb_bytes_eq1(q == 0 ? row[0].x : q == 1 ? row[0].y : q == 2 ? row[0].z : row[0].w);
            const unsigned e1 = mb_bytes_eq1(q == 0 ? row[1].x : q == 1 ? row[1].y : q == 2 ? row[1].z : row[1].w);
            const unsigned e2 = mb_bytes_eq1(q == 0 ? row[2].x : q == 1 ? row[2].y : q == 2 ? row[2].z : row[2].w);
            const unsigned e3 = mb_bytes_eq1(q == 0 ? row[3].x : q == 1 ? row[3].y : q == 2 ? row[3].z : row[3].w);
            const unsigned anyW = e0 | e1 | e2 | e3, allW = e0 & e1 & e2 & e3;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int o = 2 * q + h;
                const bool any1 = ((anyW >> (16 * h)) & 0x8080u) != 0, all1 = ((allW >> (16 * h)) & 0x8080u) == 0x8080u;
                constexpr unsigned kSpreadO[8] = { 0x0u, 0x1u, 0x8u, 0x9u, 0x40u, 0x41u, 0x48u, 0x49u };     // mb_spread3(o)
                st[sb][kSpreadO[o] | mjk] = any1 ? (all1 ? 1 : 2) : 0;
                anyMask |= any1 ? 1u << o : 0u;
            }
        }
        if (anyMask) {
            lo[0] = min(lo[0], vx + 2 * (int)__builtin_ctz(anyMask)); hi[0] = max(hi[0], vx + 2 * (32 - (int)__builtin_clz(anyMask)));
            lo[1] = min(lo[1], vy); hi[1] = max(hi[1], vy + 2); lo[2] = min(lo[2], vz); hi[2] = max(hi[2], vz + 2);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_down(lo[a], o)); hi[a] = max(hi[a], __shfl_down(hi[a], o)); }
        if ((threadIdx.x & 63) == 0) { if (lo[a] != 0x7fffffff) atomicMin(&box[a], lo[a]); if (hi[a] != -0x7fffffff) atomicMax(&box[3 + a], hi[a]); }
    }
    for (int t = threadIdx.x; t < nb * (4096 / 4); t += kBlock) reinterpret_cast<unsigned*>(ct[t >> 10])[t & 1023] = 0;      // level 1 has no mixed children
    __syncthreads();
    if (threadIdx.x < 6) blockBox[(size_t)blockIdx.x * 6 + threadIdx.x] = box[threadIdx.x];    // the host clips the union to the grid
    // ---- levels 2..B inside LDS: the children of cell m are bytes 8m .. 8m+7 of the level below
    int off = 0, cells = 4096;
    for (int l = 2; l <= B; l++) {
        const int offN = off + (cells < 16 ? 16 : cells), cellsN = cells >> 3;
        for (int t = threadIdx.x; t < cellsN * nb; t += kBlock) {
            const int sb = t / cellsN, m = t - sb * cellsN;
            const uint2 ch = *reinterpret_cast<const uint2*>(st[sb] + off + 8 * m);
            int s8[8];
#pragma unroll
            for (int c = 0; c < 8; c++) s8[c] = (int)(((c < 4 ? ch.x : ch.y) >> ((c & 3) * 8)) & 0xffu);
            int mixedChildren;
            st[sb][offN + m] = (uint8_t)mb_combine(s8, &mixedChildren);
            ct[sb][offN + m] = (uint8_t)mixedChildren;
            if (mixedChildren) atomicAdd(&mixedSum[sb][l], mixedChildren);        // mixed cells are a few per cent of a surface scene
        }
        __syncthreads();
        off = offN; cells = cellsN;
    }
    // ---- this block's share of the scan's chunk sums (k_mb_top_scan adds the levels above the bricks)
    if ((int)threadIdx.x < nb * (B - 1)) {
        const int sb = threadIdx.x / (B - 1), l = 2 + threadIdx.x % (B - 1);
        const int v = mixedSum[sb][l];
        if (v) {
            const unsigned mb = mb_morton((unsigned)(bx0 + sb), (unsigned)by, (unsigned)bz);
            const long long cell0 = R >= kMbBrickLevels ? (long long)mb << (3 * (kMbBrickLevels - l)) : 0;
            atomicAdd(&chunkSum[(Lv.cntOffset[l] + cell0) / kMbChunk], v);
        }
    }
    // ---- write-out (16 bytes per thread where a level has them): level l holds 8^(5-l) cells of a brick at morton(brick) << 3*(5-l)
    const bool whole = R >= kMbBrickLevels;                       // the bricks lie inside the root cube: every local cell exists
    for (int sb = 0; sb < nb; sb++) {
        const unsigned mb = mb_morton((unsigned)(bx0 + sb), (unsigned)by, (unsigned)bz);
        off = 0; cells = 4096;
        for (int l = 1; l <= B; l++) {
            const unsigned domain = 1u << (R - l);                // cells per edge of level l's whole domain
            const size_t base = (size_t)mb << (3 * (kMbBrickLevels - l));
            if (whole && cells >= 16) {
                for (int t = threadIdx.x; t < cells / 16; t += kBlock) {
                    reinterpret_cast<uint4*>(Lv.state[l] + base)[t] = reinterpret_cast<const uint4*>(st[sb] + off)[t];
                    if (l >= 2) reinterpret_cast<uint4*>(Lv.cnt[l] + base)[t] = reinterpret_cast<const uint4*>(ct[sb] + off)[t];
                }
            } else {
                for (int t = threadIdx.x; t < cells; t += kBlock) {
                    const unsigned i = mb_compact3((unsigned)t), j = mb_compact3((unsigned)t >> 1), k = mb_compact3((unsigned)t >> 2);
                    if ((unsigned)(((bx0 + sb) << kMbBrickLevels) >> l) + i < domain && (unsigned)(y0 >> l) + j < domain && (unsigned)(z0 >> l) + k < domain) {
                        Lv.state[l][base + t] = st[sb][off + t];
                        if (l >= 2) Lv.cnt[l][base + t] = ct[sb][off + t];
                    }
                }
            }
            off += cells < 16 ? 16 : cells; cells >>= 3;
        }
    }
}

// scan domain: cnt[2], cnt[3], ..., cnt[R] back to back, each level padded to a multiple of kMbChunk
__device__ __forceinline__ int mb_level_of(const MbLevels& Lv, long long idx) {
    int l = 2;
    while (l < Lv.R && idx >= Lv.cntOffset[l + 1]) l++;
    return l;
}

// One block, after the bricks: (1) levels B+1..R from level B (at most 8^4 cells on level 6 of a 1024^3 grid) and their share
// of the chunk sums; (2) the union of the brick boxes; (3) per level, the exclusive scan of the chunk sums (in place) and the
// level totals -> bases of every tree level.
__global__ __launch_bounds__(1024) void k_mb_top_scan(MbLevels Lv, const int* __restrict__ brickBox, int numBricks, int* __restrict__ chunkSum,
                                                      MbTables* __restrict__ T) {
    __shared__ int part[1024 / kWave][6];
    __shared__ int waveTotal[1024 / kWave];
    __shared__ long long mixedOf[kMbMaxDepth + 2];              // [l] mixed cells of pyramid level l
    const int R = Lv.R, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int B = R < kMbBrickLevels ? R : kMbBrickLevels;
    for (int l = B + 1; l <= R; l++) {
        const long long cells = 1ll << (3 * (R - l));
        for (long long p = t; p < cells; p += 1024) {
            const uint2 ch = *reinterpret_cast<const uint2*>(Lv.state[l - 1] + 8 * p);
            int s8[8];
#pragma unroll
            for (int c = 0; c < 8; c++) s8[c] = (int)(((c < 4 ? ch.x : ch.y) >> ((c & 3) * 8)) & 0xffu);
            int mixedChildren;
            Lv.state[l][p] = (uint8_t)mb_combine(s8, &mixedChildren);
            Lv.cnt[l][p] = (uint8_t)mixedChildren;
            if (mixedChildren) atomicAdd(&chunkSum[(Lv.cntOffset[l] + p) / kMbChunk], mixedChildren);
        }
        __threadfence_block();
        __syncthreads();
    }
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
    for (int b = t; b < numBricks; b += 1024)
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], brickBox[(size_t)b * 6 + a]); hi[a] = max(hi[a], brickBox[(size_t)b * 6 + 3 + a]); }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_down(lo[a], o)); hi[a] = max(hi[a], __shfl_down(hi[a], o)); }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; a++) { part[wave][a] = lo[a]; part[wave][3 + a] = hi[a]; }
    __threadfence();
    __syncthreads();
    if (t < 6) {
        int v = part[0][t];
        for (int w = 1; w < 1024 / kWave; w++) v = t < 3 ? min(v, part[w][t]) : max(v, part[w][t]);
        if (t < 3) T->solidLo[t] = v; else T->solidHi[t - 3] = v;
    }
    // ---- the scan.  The sums were made by atomics (the bricks' and this block's own): read them at the L2
    if (t == 0) mixedOf[R] = Lv.state[R][0] == 2 ? 1 : 0;
    for (int l = 2; l <= R; l++) {
        const long long c0 = Lv.cntOffset[l] / kMbChunk, c1 = Lv.cntOffset[l + 1] / kMbChunk;    // this level's chunks
        const int nb = (int)(c1 - c0);
        const int per = (nb + 1023) / 1024;
        const int first = min(t * per, nb), last = min(first + per, nb);
        int s = 0;
        for (int i = first; i < last; i++) s += __hip_atomic_load(&chunkSum[c0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int incl = s;
        for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == kWave - 1) waveTotal[wave] = incl;
        __syncthreads();
        int run = incl - s, all = 0;
        for (int w = 0; w < 1024 / kWave; w++) { const int v = waveTotal[w]; if (w < wave) run += v; all += v; }
        for (int i = first; i < last; i++) {
            const int v = __hip_atomic_load(&chunkSum[c0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chunkSum[c0 + i] = run; run += v;
        }
        if (t == 0) mixedOf[l - 1] = all;                        // cnt[l] counts the mixed cells of level l-1
        __syncthreads();
    }
    if (t == 0) {
        // tree level L = R - l: 1 node at L = 0, else 8 per mixed cell of pyramid level l+1; internal nodes = mixed cells (l >= 1)
        long long nodes = 0, internal = 0;
        for (int L = 0; L <= R; L++) {
            const int l = R - L;
            const long long m = L == 0 ? 1 : 8 * mixedOf[l + 1];
            T->levelBase[L] = nodes; T->internalBase[L] = internal;
            nodes += m; internal += l >= 1 ? mixedOf[l] : 0;
            if (m == 0) { for (int q = L + 1; q <= R + 1; q++) { T->levelBase[q] = nodes; T->internalBase[q] = internal; } break; }
            if (L == R) { T->levelBase[R + 1] = nodes; T->internalBase[R + 1] = internal; }
        }
        T->total = nodes; T->internal = internal;
    }
}

// group[l][p] = chunk base + exclusive prefix of cnt[l] inside the chunk; and, for every mixed child of p (a cell of level
// l-1 that becomes an internal node), cellOf[descriptor index] = its Morton index: the emission then runs one thread per
// internal node with every lane busy (mixed cells are ~1 % of the level-1 cells of a surface scene).
__global__ __launch_bounds__(kBlock) void k_mb_group_ranks(MbLevels Lv, const int* __restrict__ chunkBase, const MbTables* __restrict__ T,
                                                            unsigned* __restrict__ cellOf) {
    __shared__ int waveTotal[kBlock / kWave];
    const long long c0 = (long long)blockIdx.x * kMbChunk;
    const int l = mb_level_of(Lv, c0);
    const long long cells = 1ll << (3 * (Lv.R - l)), local0 = c0 - Lv.cntOffset[l];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each thread owns 4 consecutive cells
    const long long p0 = local0 + (long long)threadIdx.x * 4;
    int v[4], sum = 0;
    if (p0 + 3 < cells) {
        const unsigned w = *reinterpret_cast<const unsigned*>(Lv.cnt[l] + p0);
#pragma unroll
        for (int q = 0; q < 4; q++) { v[q] = (int)((w >> (8 * q)) & 0xffu); sum += v[q]; }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) { v[q] = p0 + q < cells ? Lv.cnt[l][p0 + q] : 0; sum += v[q]; }
    }
    int incl = sum;
    for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == kWave - 1) waveTotal[wave] = incl;
    __syncthreads();
    int run = chunkBase[blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; w++) run += waveTotal[w];
    const long long dBase = T->internalBase[Lv.R - (l - 1)];                // descriptors of tree level R-(l-1) = the mixed cells of level l-1
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (p0 + q >= cells) break;
        Lv.group[l][p0 + q] = run;
        if (v[q]) {
            const uint2 ch = *reinterpret_cast<const uint2*>(Lv.state[l - 1] + 8 * (p0 + q));
            int r = run;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if ((((j < 4 ? ch.x : ch.y) >> ((j & 3) * 8)) & 0xffu) == 2u) cellOf[dBase + r++] = (unsigned)(8 * (p0 + q) + j);
        }
        run += v[q];
    }
}

// A mixed cell (Morton index p of pyramid level l) is an internal node, descriptor index d: it owns its descriptor and the
// records of its 8 children, nodes 1 + 8 d .. 1 + 8 d + 7 (tree level L's nodes follow level L-1's, 8 per internal node of
// the level above, in descriptor order: levelBase[L + 1] + 8 * rank = 1 + 8 d).  One thread per child; a block's 256 records
// (15,360 bytes) are contiguous, so they are built in LDS -- at the same 16-byte phase as their place in the array -- and
// leave as full 16-byte stores.  Block 0 also writes the root's own record.
constexpr int kMbEmitCells = kBlock / 8;
constexpr int kMbNodeWords = (int)(sizeof(rto_node) / 4);
__global__ __launch_bounds__(kBlock) void k_mb_emit(const uint8_t* __restrict__ vox, MbLevels Lv, const MbTables* __restrict__ T,
                                                     const unsigned* __restrict__ cellOf, rto_node* __restrict__ nodes, uint2* __restrict__ desc,
                                                     int* __restrict__ descFirstChild) {
    __shared__ __attribute__((aligned(16))) int stage[4 + kBlock * kMbNodeWords];
    const int R = Lv.R, t = threadIdx.x, k = t & 7;
    const long long internal = T->internal;
    const long long d0 = (long long)blockIdx.x * kMbEmitCells, d = d0 + (t >> 3);
    if (blockIdx.x == 0 && t == 0) {                               // the root's own record
        const int state = Lv.state[R][0];
        rto_node nd;
        nd.x = nd.y = nd.z = 0; nd.size = 1 << R;
        nd.isLeaf = state == 2 ? 0 : 1; nd.isUniform = nd.isLeaf; nd.isSolid = state == 1 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) nd.child[j] = state == 2 ? 1 + j : -1;
        nodes[0] = nd;
    }
    if (d < internal) {
        int L = 0;
        while (L < R && d >= T->internalBase[L + 1]) L++;
        const int l = R - L;
        const long long p = L == 0 ? 0 : (long long)cellOf[d];
        const unsigned cx = mb_compact3((unsigned)p), cy = mb_compact3((unsigned)(p >> 1)), cz = mb_compact3((unsigned)(p >> 2));
        const int half = 1 << (l - 1);
        const int x = (int)(2 * cx + (k & 1)), y = (int)(2 * cy + ((k >> 1) & 1)), z = (int)(2 * cz + (k >> 2));     // in cells of level l-1
        unsigned imask = 0, smask = 0;                             // children that are internal / solid leaves
        if (l >= 2) {
            const uint2 ch = *reinterpret_cast<const uint2*>(Lv.state[l - 1] + 8 * p);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned sj = ((j < 4 ? ch.x : ch.y) >> ((j & 3) * 8)) & 0xffu;
                imask |= sj == 2u ? 1u << j : 0u; smask |= sj == 1u ? 1u << j : 0u;
            }
        } else {
            const bool filled = mb_voxel(vox, Lv.dimX, Lv.dimY, Lv.dimZ, x, y, z) == 1;
            smask = (unsigned)((__ballot(filled) >> (threadIdx.x & 56)) & 0xffull);      // the 8 lanes of this cell
        }
        const bool isInternal = (imask >> k) & 1u;
        const long long childRank0 = l >= 2 ? (long long)Lv.group[l][p] : 0;   // rank of the first child among the mixed cells of level l-1
        const int before = __builtin_popcount(imask & ((1u << k) - 1u));       // mixed children before child k
        const int g0 = isInternal ? (int)(T->levelBase[L + 2] + 8 * (childRank0 + before)) : 0;
        int* w = stage + 3 + t * kMbNodeWords;
        w[0] = x * half; w[1] = y * half; w[2] = z * half; w[3] = half;
        w[4] = isInternal ? 0 : 1;                                 // isLeaf
        w[5] = (smask >> k) & 1u;                                  // isSolid
        w[6] = w[4];                                               // isUniform (S/OctreeVoxel.cpp:716-745: leaf <=> uniform)
#pragma unroll
        for (int j = 0; j < 8; j++) w[7 + j] = isInternal ? g0 + j : -1;
        if (k == 0) {
            desc[d] = make_uint2(smask | (imask << 8) | 0xff0000u, imask ? (unsigned)(T->internalBase[L + 1] + childRank0) : 0u);
            descFirstChild[d] = (int)(1 + 8 * d);
        }
    }
    __syncthreads();
    const long long left = internal - d0;
    const int words = (int)(left < kMbEmitCells ? (left > 0 ? left : 0) : kMbEmitCells) * 8 * kMbNodeWords;       // this block's records
    // global word (1 + 8 d0) * 15 + i  <->  stage[3 + i]: (15 + 120 d0) % 4 == 3, so 16-byte groups coincide
    int* g = reinterpret_cast<int*>(nodes) + (1 + 8 * d0) * kMbNodeWords - 3;
    for (int c = t; 4 * c < 3 + words; c += kBlock) {
        if (c > 0 && 4 * c + 4 <= 3 + words) *reinterpret_cast<int4*>(g + 4 * c) = *reinterpret_cast<const int4*>(stage + 4 * c);
        else
            for (int q = 0; q < 4; q++)
                if (4 * c + q >= 3 && 4 * c + q < 3 + words) g[4 * c + q] = stage[4 * c + q];
    }
}

// bounding box of the solid leaves (voxel units) for the launch-order heuristic.  Grid-stride over the nodes with a
// small grid, wave + block reduction, then at most 6 atomics per BLOCK: same-address atomics serialise at ~10 ns
// each (one set per 64 nodes cost 1 ms at 1.5 M nodes, one set per wave of a 1024-block grid still 0.29 ms).
__global__ __launch_bounds__(kBlock) void k_solid_bbox(const rto_node* __restrict__ nodes, int64_t n, int* __restrict__ bbox /* lo[3], hi[3] */) {
    __shared__ int part[kBlock / kWave][6];
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const rto_node* nd = nodes + i;
        if ((nd->isLeaf == 1 || nd->isUniform == 1) && nd->isSolid == 1) {
            const int x = nd->x, y = nd->y, z = nd->z, sz = nd->size;
            lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
            hi[0] = max(hi[0], x + sz); hi[1] = max(hi[1], y + sz); hi[2] = max(hi[2], z + sz);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], __shfl_down(lo[a], off));
            hi[a] = max(hi[a], __shfl_down(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { part[threadIdx.x >> 6][a] = lo[a]; part[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        int v = part[0][a];
        for (int w = 1; w < kBlock / kWave; w++) v = a < 3 ? min(v, part[w][a]) : max(v, part[w][a]);
        if (a < 3) { if (v != 0x7fffffff) atomicMin(&bbox[a], v); }
        else if (v != -0x7fffffff) atomicMax(&bbox[a], v);
    }
}

// ================================================================ N2: leaf-triangle buffer on the GPU
// What MarchingCubesRenderer::render emits per leaf (S/Renderer.cpp:14-36 -> localMC, S/OctreeVoxel.cpp:780-879), as
// host/LocalMC.cpp states it, built in HBM: node i's triangles at triOffset[i]..triOffset[i+1], 12 floats each, in
// the host builder's order (cells of a leaf in z, y, x order, a cell's triangles in case-table order) and with its
// float operations (pos = min + float(c)*voxel; vertex = p1 + 0.5*(p2 - p1); normal = normalize(cross(..))).
// Only cells on a leaf's three max faces can straddle occupancy (a leaf is uniform, a cell reads corners x..x+1),
// so a leaf of edge s has at most 3s^2-3s+1 candidate cells, enumerated in the same z, y, x order.
//   count:  one thread per leaf up to kLeafSerialCandidates candidates, one wave per larger leaf
//   scan :  block sums -> k_scan_block_counts -> per-block exclusive scan = triOffset
//   emit :  same walk, writing at triOffset[node] + running count
struct LeafTriParams {
    const rto_node* nodes;
    int64_t n;
    const uint8_t* vox;          // x fastest, 0 EMPTY / 1 FILLED
    int dimX, dimY, dimZ;
    float minX, minY, minZ, vs;
    const unsigned long long* cases;   // 256 x (triangle count << 60 | 15 edge nibbles, first edge in the low nibble)
};

constexpr int kLeafSerialCandidates = 64;

struct LeafCells { int x0, y0, z0, s, ex, ey, ez, perLayer, nonTop, rowsNotMax, total; };

__device__ __forceinline__ LeafCells leaf_cells(const LeafTriParams& P, const rto_node& nd) {
    LeafCells L;
    L.x0 = nd.x; L.y0 = nd.y; L.z0 = nd.z; L.s = nd.size;
    // localMC's loop bounds: c < c0 + size && c < dim - 1
    L.ex = max(0, min(nd.x + nd.size, P.dimX - 1) - nd.x);
    L.ey = max(0, min(nd.y + nd.size, P.dimY - 1) - nd.y);
    L.ez = max(0, min(nd.z + nd.size, P.dimZ - 1) - nd.z);
    if (L.ex == 0 || L.ey == 0 || L.ez == 0) { L.perLayer = L.nonTop = L.rowsNotMax = L.total = 0; return L; }
    const bool fullX = L.ex == L.s, fullY = L.ey == L.s, fullZ = L.ez == L.s;
    L.rowsNotMax = L.ey - (fullY ? 1 : 0);                    // rows of a non-top layer that only contribute x = s-1
    L.perLayer = (fullX ? L.rowsNotMax : 0) + (fullY ? L.ex : 0);
    L.nonTop = L.ez - (fullZ ? 1 : 0);
    L.total = L.nonTop * L.perLayer + (fullZ ? L.ex * L.ey : 0);
    return L;
}

// candidate c (0 <= c < L.total) in z, y, x order -> local cell (i, j, k)
__device__ __forceinline__ void leaf_candidate(const LeafCells& L, int c, int& i, int& j, int& k) {
    const int below = L.nonTop * L.perLayer;
    if (c >= below) { const int r = c - below; k = L.s - 1; j = r / L.ex; i = r - j * L.ex; return; }
    k = c / L.perLayer;
    int r = c - k * L.perLayer;
    if (L.ex == L.s) {
        if (r < L.rowsNotMax) { j = r; i = L.s - 1; return; }
        r -= L.rowsNotMax;
    }
    j = L.s - 1; i = r;                                        // the max row (exists: perLayer counted it)
}

__device__ __forceinline__ int cell_case(const LeafTriParams& P, int x, int y, int z) {
    // corner numbering of S/OctreeVoxel.cpp:802-817: 0 (0,0,0) 1 (1,0,0) 2 (1,1,0) 3 (0,1,0) 4..7 the same at z+1
    const size_t sy = (size_t)P.dimX, sz = (size_t)P.dimX * P.dimY;
    const uint8_t* v = P.vox + (size_t)z * sz + (size_t)y * sy + x;
    int idx = 0;
    idx |= (v[0] == 1) ? 1 : 0;           idx |= (v[1] == 1) ? 2 : 0;
    idx |= (v[sy + 1] == 1) ? 4 : 0;      idx |= (v[sy] == 1) ? 8 : 0;
    idx |= (v[sz] == 1) ? 16 : 0;         idx |= (v[sz + 1] == 1) ? 32 : 0;
    idx |= (v[sz + sy + 1] == 1) ? 64 : 0; idx |= (v[sz + sy] == 1) ? 128 : 0;
    return idx;
}

__device__ __forceinline__ void emit_cell_triangles(const LeafTriParams& P, int x, int y, int z, unsigned long long cs, float* __restrict__ out) {
    const int cornerOff[8][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 1, 1, 1 }, { 0, 1, 1 } };
    const int edgeCorner[12][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 0 }, { 4, 5 }, { 5, 6 }, { 6, 7 }, { 7, 4 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
    const int ntri = (int)(cs >> 60);
    for (int t = 0; t < ntri; t++) {
        float v[3][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int e = (int)((cs >> (4 * (3 * t + q))) & 15ull);
            const int a = edgeCorner[e][0], b = edgeCorner[e][1];
            const float p1x = P.minX + (float)(x + cornerOff[a][0]) * P.vs, p1y = P.minY + (float)(y + cornerOff[a][1]) * P.vs,
                        p1z = P.minZ + (float)(z + cornerOff[a][2]) * P.vs;
            const float p2x = P.minX + (float)(x + cornerOff[b][0]) * P.vs, p2y = P.minY + (float)(y + cornerOff[b][1]) * P.vs,
                        p2z = P.minZ + (float)(z + cornerOff[b][2]) * P.vs;
            // vertexInterp with values -1 / +1 and iso 0: mu = (0 - v1) / (v2 - v1) = 0.5 exactly
            v[q][0] = p1x + 0.5f * (p2x - p1x); v[q][1] = p1y + 0.5f * (p2y - p1y); v[q][2] = p1z + 0.5f * (p2z - p1z);
        }
        const float ax = v[1][0] - v[0][0], ay = v[1][1] - v[0][1], az = v[1][2] - v[0][2];
        const float bx = v[2][0] - v[0][0], by = v[2][1] - v[0][1], bz = v[2][2] - v[0][2];
        const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;      // glm cross
        const float tx = cx * cx, ty = cy * cy, tz = cz * cz;
        const float inv = inversesqrt(tx + ty + tz);
        float* o = out + (size_t)t * 12;
#pragma unroll
        for (int q = 0; q < 3; q++) { o[3 * q] = v[q][0]; o[3 * q + 1] = v[q][1]; o[3 * q + 2] = v[q][2]; }
        o[9] = cx * inv; o[10] = cy * inv; o[11] = cz * inv;
    }
}

__device__ __forceinline__ bool is_leaf_node(const rto_node& nd) { return nd.isLeaf == 1; }

constexpr int kLeafChunk = 2048;     // candidates one wave walks for a big leaf (32 steps); big leaves are cut into such chunks

// pass 1 (EMIT = false): triCount[node]; leaves with more candidates than a thread should walk go to bigList (and
//                        their chunk count to *chunkTotal).
// pass 2 (EMIT = true) : the same walk writing the triangles.
template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_leaftri_small(LeafTriParams P, int* __restrict__ triCount, const int* __restrict__ triOffset,
                                                           int* __restrict__ bigList, int* __restrict__ bigCount /* [0] leaves, [1] chunks */,
                                                           float* __restrict__ tris) {
    const int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (node >= P.n) return;
    const rto_node nd = P.nodes[node];
    if (!is_leaf_node(nd)) { if (!EMIT) triCount[node] = 0; return; }
    const LeafCells L = leaf_cells(P, nd);
    if (L.total > kLeafSerialCandidates) {
        if (!EMIT) {
            triCount[node] = 0;
            bigList[atomicAdd(&bigCount[0], 1)] = (int)node;
            atomicAdd(&bigCount[1], (L.total + kLeafChunk - 1) / kLeafChunk);
        }
        return;
    }
    int run = EMIT ? triOffset[node] : 0;
    for (int c = 0; c < L.total; c++) {
        int i, j, k;
        leaf_candidate(L, c, i, j, k);
        const unsigned long long cs = P.cases[cell_case(P, L.x0 + i, L.y0 + j, L.z0 + k)];
        if (EMIT) emit_cell_triangles(P, L.x0 + i, L.y0 + j, L.z0 + k, cs, tris + (size_t)run * 12);
        run += (int)(cs >> 60);
    }
    if (!EMIT) triCount[node] = run;
}

// one thread per big leaf: claim its chunk range and list (leaf slot, chunk number) per chunk
__global__ __launch_bounds__(kBlock) void k_leaftri_plan(LeafTriParams P, const int* __restrict__ bigList, int bigCount, int* __restrict__ cursor,
                                                          int* __restrict__ bigFirstChunk, int2* __restrict__ chunks) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= bigCount) return;
    const LeafCells L = leaf_cells(P, P.nodes[bigList[b]]);
    const int nch = (L.total + kLeafChunk - 1) / kLeafChunk;
    const int first = atomicAdd(cursor, nch);
    bigFirstChunk[b] = first;
    for (int q = 0; q < nch; q++) chunks[first + q] = make_int2(b, q);
}

// one wave per chunk.  EMIT = false: chunkCount[chunk]; EMIT = true: triangles at triOffset[node] + chunkOff[chunk] + running
template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_leaftri_big(LeafTriParams P, const int* __restrict__ bigList, const int2* __restrict__ chunks, int numChunks,
                                                         int* __restrict__ chunkCount, const int* __restrict__ chunkOff,
                                                         const int* __restrict__ triOffset, float* __restrict__ tris) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (ch >= numChunks) return;                               // wave-uniform
    const int2 cq = chunks[ch];
    const int node = bigList[cq.x];
    const rto_node nd = P.nodes[node];
    const LeafCells L = leaf_cells(P, nd);
    const int cBegin = cq.y * kLeafChunk, cEnd = min(L.total, cBegin + kLeafChunk);
    int run = EMIT ? triOffset[node] + chunkOff[ch] : 0;
    for (int c0 = cBegin; c0 < cEnd; c0 += kWave) {
        const int c = c0 + lane;
        unsigned long long cs = 0;
        int i = 0, j = 0, k = 0;
        if (c < cEnd) {
            leaf_candidate(L, c, i, j, k);
            cs = P.cases[cell_case(P, L.x0 + i, L.y0 + j, L.z0 + k)];
        }
        const int nt = (int)(cs >> 60);
        int incl = nt;                                         // inclusive wave prefix of the triangle counts
        for (int off = 1; off < kWave; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (EMIT && nt) emit_cell_triangles(P, L.x0 + i, L.y0 + j, L.z0 + k, cs, tris + (size_t)(run + incl - nt) * 12);
        run += __shfl(incl, kWave - 1);
    }
    if (!EMIT && lane == 0) chunkCount[ch] = run;
}

// one thread per big leaf: exclusive prefix of its chunks' counts (chunk order = candidate order), total -> triCount
__global__ __launch_bounds__(kBlock) void k_leaftri_bigsum(LeafTriParams P, const int* __restrict__ bigList, int bigCount,
                                                            const int* __restrict__ bigFirstChunk, const int* __restrict__ chunkCount,
                                                            int* __restrict__ chunkOff, int* __restrict__ triCount) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= bigCount) return;
    const int node = bigList[b];
    const LeafCells L = leaf_cells(P, P.nodes[node]);
    const int nch = (L.total + kLeafChunk - 1) / kLeafChunk;
    const int first = bigFirstChunk[b];
    int run = 0;
    for (int q = 0; q < nch; q++) { chunkOff[first + q] = run; run += chunkCount[first + q]; }
    triCount[node] = run;
}

// block sums of an int array (for the exclusive scan that turns triCount into triOffset)
__global__ __launch_bounds__(kBlock) void k_block_sums(const int* __restrict__ v, int64_t n, int* __restrict__ blockSum) {
    __shared__ int waveTotal[kBlock / kWave];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int x = i < n ? v[i] : 0;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if ((threadIdx.x & 63) == 0) waveTotal[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kBlock / kWave; w++) t += waveTotal[w]; blockSum[blockIdx.x] = t; }
}

// out[i] = blockBase[block] + exclusive prefix within the block; out[n] = total
__global__ __launch_bounds__(kBlock) void k_block_exclusive_scan(const int* __restrict__ v, int64_t n, const int* __restrict__ blockBase,
                                                                  const int64_t* __restrict__ total, int* __restrict__ out) {
    __shared__ int waveTotal[kBlock / kWave];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = i < n ? v[i] : 0;
    int incl = x;
    for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) waveTotal[wave] = incl;
    __syncthreads();
    int base = blockBase[blockIdx.x];
    for (int w = 0; w < wave; w++) base += waveTotal[w];
    if (i < n) out[i] = base + incl - x;
    if (i == 0) out[n] = (int)*total;
}

// ================================================================ occupancy mask: the coarse cells (built once per octree)
// Which level to take the cells from: per depth, the number of internal nodes and of solid leaves (one pass over the descriptors;
// the children of the node of descriptor d lie one level below it).  counts: [0..kMaxDepth] internal, [kMaxDepth+1 ..] solid leaves.
__global__ __launch_bounds__(kBlock) void k_cells_count(const uint2* __restrict__ desc, const int4* __restrict__ descPos, int64_t nInternal, int depth,
                                                         int* __restrict__ counts) {
    __shared__ int h[2 * (kMaxDepth + 1)];
    for (int i = threadIdx.x; i < 2 * (kMaxDepth + 1); i += kBlock) h[i] = 0;
    __syncthreads();
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d < nInternal) {
        const unsigned w = desc[d].x;
        const int l = depth - (31 - __builtin_clz((unsigned)descPos[d].w)) + 1;       // depth of the CHILDREN of this node
        if (l >= 0 && l <= kMaxDepth) {
            atomicAdd(&h[l], __builtin_popcount((w >> 8) & 0xffu));
            atomicAdd(&h[kMaxDepth + 1 + l], __builtin_popcount(w & 0xffu));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (kMaxDepth + 1); i += kBlock) if (h[i]) atomicAdd(&counts[i], h[i]);
}

// The cells of level L: internal children at depth L, solid-leaf children at depth <= L (position, edge), in any order.
__global__ __launch_bounds__(kBlock) void k_cells_fill(const uint2* __restrict__ desc, const int4* __restrict__ descPos, int64_t nInternal, int depth, int L,
                                                        int4* __restrict__ cells, int capacity, int* __restrict__ cursor) {
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned sel = 0;
    int4 p = make_int4(0, 0, 0, 0);
    if (d < nInternal) {
        const unsigned w = desc[d].x;
        p = descPos[d];
        const int l = depth - (31 - __builtin_clz((unsigned)p.w)) + 1;
        if (l <= L) sel = (w & 0xffu) | (l == L ? ((w >> 8) & 0xffu) : 0u);
    }
    const int n = __builtin_popcount(sel);
    int incl = n;                                              // wave-level prefix sum: one atomic per wave
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    int base = 0;
    if (lane == kWave - 1 && incl > 0) base = atomicAdd(cursor, incl);
    base = __shfl(base, kWave - 1) + incl - n;
    const int half = p.w >> 1;
    while (sel) {
        const int k = __builtin_ctz(sel);
        sel &= sel - 1;
        if (base < capacity) cells[base] = make_int4(p.x + ((k & 1) ? half : 0), p.y + ((k & 2) ? half : 0), p.z + ((k & 4) ? half : 0), half);
        base++;
    }
}

// ================================================================ multi-GPU reassembly
// d_gathered: numParts compact buffers, each padded to partRows rows of W pixels; part p starts partStride pixels after
// part p-1 (partRows * W when one frame was gathered; batch * partRows * W when the ranks shipped `batch` frames in
// one collective, the caller then passes the pointer to its frame inside part 0).
// One block row per image row (the band arithmetic is per row, not per pixel), blockIdx.z = frame of the batch; a frame of
// the batch starts srcFrameStride elements after the previous one inside every part, and dstFrameStride bytes after it in
// the output.  HBM-bound: 16 B (4 B) read and 16 B written per pixel.
// Row y of a frame cut into bands of bandRows rows, band b owned by part b % numParts: which part holds it and at which row of
// that part's compact buffer (a part's bands lie back to back).  ONE definition for the assembly kernels and for the exported
// planner (rto_split_row_source), which the multi-process tests drive.
__host__ __device__ inline void band_row_source(int y, int numParts, int bandRows, int& part, int& localRow) {
    const int gband = y / bandRows, r = y - gband * bandRows;
    part = gband % numParts;
    localRow = (gband / numParts) * bandRows + r;
}

__global__ void k_assemble(const float4* __restrict__ gathered, char* __restrict__ frames, size_t srcFrameStride, size_t dstFrameStride,
                           int W, int H, int numParts, int bandRows, size_t partStride) {
    const int y = blockIdx.y;
    int part, lrow;
    band_row_source(y, numParts, bandRows, part, lrow);
    const float4* src = gathered + (size_t)blockIdx.z * srcFrameStride + (size_t)part * partStride + (size_t)lrow * W;
    float4* dst = reinterpret_cast<float4*>(frames + (size_t)blockIdx.z * dstFrameStride) + (size_t)y * W;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) store_pixel(dst + x, src[x]);
}

// Same re-interleave for the 4-byte shade payload, finishing the colour expression (shade_color) on the way.
__global__ void k_assemble_shade(const float* __restrict__ gathered, char* __restrict__ frames, size_t srcFrameStride, size_t dstFrameStride,
                                 int W, int H, int numParts, int bandRows, size_t partStride) {
    const int y = blockIdx.y;
    int part, lrow;
    band_row_source(y, numParts, bandRows, part, lrow);
    const float* src = gathered + (size_t)blockIdx.z * srcFrameStride + (size_t)part * partStride + (size_t)lrow * W;
    float4* dst = reinterpret_cast<float4*>(frames + (size_t)blockIdx.z * dstFrameStride) + (size_t)y * W;
    // consecutive lanes take consecutive pixels: 256 B read, 1 KB written per wave instruction, whole lines either way
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) store_pixel(dst + x, shade_color(src[x]));
}

// The multi-GPU payload without its background: outside the columns of the geometry's screen rectangle every pixel of a
// frame is a miss by construction (the same rectangle the traversal kernels use), so a rank ships only the window
// [x0, x0 + w) of every row of its part and rank 0 paints the rest itself -- about a third of the bytes on the links at
// config 2.  CropInfo: the windows of the frames of one batch and where each frame starts inside a rank's packed part.
constexpr int kMaxCropFrames = 32;
struct CropInfo {
    int n;
    int x0[kMaxCropFrames], w[kMaxCropFrames];
    long long off[kMaxCropFrames];
};

// [frame][rows][W] -> [frame: rows x w_i], one block row per part row, blockIdx.z = frame
__global__ void k_pack_columns(const float* __restrict__ src, float* __restrict__ dst, CropInfo C, size_t srcFrameStride, int W) {
    const int i = blockIdx.z, y = blockIdx.y;
    const int w = C.w[i];
    const float* s = src + (size_t)i * srcFrameStride + (size_t)y * W + C.x0[i];
    float* d = dst + C.off[i] + (size_t)y * w;
    for (int x = threadIdx.x; x < w; x += blockDim.x) d[x] = s[x];
}

// k_assemble_shade over packed parts: inside the window the colour of the shipped Lambert term, outside it the background
__global__ void k_assemble_shade_crop(const float* __restrict__ gathered, char* __restrict__ frames, CropInfo C, size_t dstFrameStride,
                                      int W, int H, int numParts, int bandRows, size_t partStride) {
    const int i = blockIdx.z, y = blockIdx.y;
    int part, lrow;
    band_row_source(y, numParts, bandRows, part, lrow);
    const int x0 = C.x0[i], w = C.w[i];
    const float* src = gathered + (size_t)part * partStride + C.off[i] + (size_t)lrow * w;
    float4* dst = reinterpret_cast<float4*>(frames + (size_t)i * dstFrameStride) + (size_t)y * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const unsigned rel = (unsigned)(x - x0);
        store_pixel(dst + x, shade_color(rel < (unsigned)w ? src[rel] : kShadeMiss));
    }
}

}  // namespace rto
