/*
 * rto_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's ray->octree hot path
 * (abodthedude25/Ray_Tracing_Octrees, 453-skeleton/, abbreviated S/ below).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (ray_tracing_octrees_amd/) never does.
 *
 * Parity status: octree build, BFS flatten, camera, glm matrix helpers,
 * frustum test and the sceneCache.bin reader are PINNED against the
 * reference's own compiled sources (oracle/_ref, see oracle/Makefile and
 * tests/golden/).  The traversal/shading kernel (S/RayTracerBVH.cpp:182-369)
 * is GLSL that cannot execute here or on the GPU box and the reference holds
 * no rendered image, so for that part parity vs the real GLSL execution is
 * UNPINNED: this file is the definition of the expected pixels.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (no -ffast-math).  All float
 * expressions are written in the operation order of the GLSL / glm 0.9.9.7
 * source they restate; the HIP kernels reproduce the same order.
 */
#ifndef RTO_ORACLE_H
#define RTO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* S/RayTracerBVH.h:21-26 `struct GPUNodes` == GLSL OctreeNodeGPUStruct
 * (S/RayTracerBVH.cpp:195-204): 15 x int32 = 60 bytes, child = -1 if none. */
typedef struct orc_node {
    int32_t x, y, z, size;
    int32_t isLeaf, isSolid, isUniform;
    int32_t child[8];
} orc_node;

/* S/OctreeVoxel.h:28-42 `VoxelGrid` as a POD view (data: 0 EMPTY, 1 FILLED,
 * x fastest: index = x + y*dimX + z*dimX*dimY). */
typedef struct orc_grid {
    int32_t dimX, dimY, dimZ;
    float minX, minY, minZ;
    float voxelSize;
    uint8_t* data; /* dimX*dimY*dimZ bytes, owned by caller unless stated */
} orc_grid;

typedef struct orc_stats {
    uint64_t rays;         /* pixels traced                              */
    uint64_t pops;         /* node pops (traversalSteps summed)          */
    uint64_t hits;         /* rays with an accepted hit                  */
    uint64_t capped;       /* rays that ended because steps reached 512  */
    uint64_t internal;     /* internal nodes whose children were pushed  */
    uint32_t max_stack;    /* max stack pointer observed                 */
    uint32_t pad;
} orc_stats;

/* ---- scene (S/main.cpp:337-372, 1052-1070, 376-422) ------------------- */
void orc_generate_test_sphere(int dimX, int dimY, int dimZ, uint8_t* out);
int  orc_recenter_filled_voxels(orc_grid* g);           /* 0 = no filled voxel */
void orc_make_test_sphere_grid(int dim, orc_grid* g);   /* mallocs g->data     */

/* ---- sceneCache.bin (S/CacheUtils.cpp:33-59) --------------------------- */
int  orc_load_voxel_grid(const char* path, orc_grid* g); /* mallocs g->data; 1 ok */
int  orc_save_voxel_grid(const char* path, const orc_grid* g);

/* ---- octree build + flatten (S/OctreeVoxel.cpp:692-778, S/RayTracerBVH.cpp:430-490) */
/* Returns node count; *out is malloc'd (orc_free). 0 if the grid is empty-dimensioned. */
int64_t orc_build_flat_octree(const orc_grid* g, orc_node** out);
void    orc_free(void* p);

/* ---- glm 0.9.9.7 helpers, column-major float[16] ----------------------- */
void  orc_mat4_inverse(const float m[16], float out[16]);
void  orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void  orc_perspective(float fovyRad, float aspect, float zNear, float zFar, float out[16]);
void  orc_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
float orc_radians(float deg);

/* ---- Camera (S/Camera.cpp:8-29, 48-52, 76-82) --------------------------- */
typedef struct orc_camera { float theta, phi, radius; float target[3]; } orc_camera;
void orc_camera_init(orc_camera* c, float theta, float phi, float radius);
void orc_camera_pos(const orc_camera* c, float out[3]);
void orc_camera_view(const orc_camera* c, float out[16]);
void orc_camera_pan(orc_camera* c, float dx, float dy);

/* ---- Frustum (S/Frustum.cpp:5-93) --------------------------------------- */
void orc_frustum_planes(const float viewProj[16], float planes[24]);
int  orc_frustum_test_aabb(const float planes[24], const float bmin[3], const float bmax[3], float margin);

/* ---- culling compaction (S/RayTracerBVH.cpp:731-802) -------------------- */
/* out must hold n nodes; visible[] (n bytes, optional) gets the per-node flag.
 * Returns the visible count. */
int64_t orc_cull_compact_planes(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                                const float planes[24], float margin, orc_node* out, uint8_t* visible);
int64_t orc_cull_compact(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                         const float view[16], float fovDeg, float aspect,
                         orc_node* out, uint8_t* visible);

/* ---- the kernel (S/RayTracerBVH.cpp:226-368) ---------------------------- */
/* Renders rows [y0,y1) of a W x H image into out (full-frame pointer,
 * RGBA32F row-major, row 0 = top).  nthreads<=1: scalar; >1: OpenMP rows. */
void orc_render(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                const float view[16], const float camPos[3], float aspect, float fovDeg,
                int W, int H, int y0, int y1, float* out, orc_stats* stats, int nthreads);

/* Per-pixel traversal step counts (for tests of the 512 cap). steps: W*H int32. */
/* closest-hit traversal of the reference's earlier (block-commented) shader, S/RayTracerBVH.cpp:63-138 */
void orc_render_closest(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                        const float view[16], const float camPos[3], float aspect, float fovDeg,
                        int W, int H, float* out, orc_stats* stats, int nthreads);
void orc_render_steps(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                      const float view[16], const float camPos[3], float aspect, float fovDeg,
                      int W, int H, int32_t* steps);

void orc_render_visits(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                       const float view[16], const float camPos[3], float aspect, float fovDeg,
                       int W, int H, int32_t* visits);

/* ---- N1: front-to-back nearest hit (S/VolumeRaycastRenderer.cpp:50-155) - */
float orc_octree_ray_skip(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                          const float ro[3], const float rd[3], float tMin, float tMax);
float orc_octree_ray_skip_vis(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                              const float ro[3], const float rd[3], float tMin, float tMax, const uint8_t* vis);

/* generateRay for every pixel (W*H x 3), the nearest-hit render mode built on octreeRaySkip, and octreeRaySkip's consumer in
 * drawRaycast (S/VolumeRaycastRenderer.cpp:1602-1663): see rto_oracle.c. */
void orc_generate_rays(const float view[16], const float camPos[3], float aspect, float fovDeg, int W, int H, float* rd);
void orc_render_skip(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                     const float view[16], const float camPos[3], float aspect, float fovDeg, int W, int H,
                     const uint8_t* vis, float* outRGBA, float* outT, int nthreads);
void orc_probe_rays(const float view[16], const float eye[3], float aspect, float* rd);
float orc_probe_skip_distance(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                              const float view[16], const float eye[3], float aspect, const uint8_t* vis, float lastSkipDistance);

/* ---- N2: leaf triangles + shadow ray (BASELINE config 5) -------------------------------------------
 * localMC is a restatement of S/OctreeVoxel.cpp:780-879 and IS pinned by the reference's own triangles
 * (tests/golden/ref_localmc_sphere16.npz, ref_mc_cases.npz).  The renderer below has NO reference counterpart
 * (no ray/triangle code exists upstream): it is this project's definition of config 5 --
 * "parity unpinned".  Semantics:
 *   - triangle buffer: for every node i in flat (BFS) order that is a leaf, the triangles
 *     localMC(grid, x, y, z, size) (what MarchingCubesRenderer::render emits per leaf, S/Renderer.cpp:14-36);
 *     12 floats per triangle (v0, v1, v2, face normal); triOffset[i]..triOffset[i+1] is node i's range.
 *   - traversal: exactly intersectOctreeIterative (same LIFO order, 512-pop cap), except that a popped leaf
 *     (solid or not) that passes the slab test is tested against its triangles (Moeller-Trumbore, t > 0, nearest
 *     within the leaf); the first leaf in pop order with a triangle hit ends the traversal.
 *   - shading: n = face normal, flipped to face the ray; c = (1,.8,.6)*max(0, dot(n, -L)) + 0.1 with the
 *     reference's light L = normalize(-1,-1,-1); if `shadow`, one ray from p + n*1e-3*voxelSize towards -L
 *     through the same traversal: any triangle hit drops the diffuse term (c = 0.1). Miss: (0,0,0,1). */
int64_t orc_local_mc(const orc_grid* g, int x0, int y0, int z0, int size, float** out18);   /* 18 floats/tri as the reference's MCTriangle */
/* Returns the triangle count; *tris (12 floats each) and *triOffset (n+1 ints) are malloc'd. */
int64_t orc_build_leaf_triangles(const orc_grid* g, const orc_node* nodes, int64_t n, float** tris, int32_t** triOffset);
void orc_render_triangles(const orc_node* nodes, int64_t n, const float* tris, const int32_t* triOffset,
                          const float gridMin[3], float voxelSize, const float view[16], const float camPos[3],
                          float aspect, float fovDeg, int W, int H, int shadow, float* out, orc_stats* stats, int nthreads);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
