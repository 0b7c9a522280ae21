// ref_shim.cpp -- ORACLE-SIDE test infrastructure (never shipped, never linked by the product).
//
// A thin extern "C" door onto the REAL reference code, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile target
// `ref`; outputs only into oracle/_ref/).  It lets tests/golden/make_golden.py
// and the oracle-pinning tests call the reference's own
//   createOctreeFromVoxelGrid / getVoxelSafe / localMC   (453-skeleton/OctreeVoxel.cpp)
//   Camera::getView / getPos / pan                        (453-skeleton/Camera.cpp)
//   Frustum::Frustum / testAABB                           (453-skeleton/Frustum.cpp)
//   loadVoxelGrid                                         (453-skeleton/CacheUtils.cpp)
//   glm::inverse / perspective / operator*                (thirdparty/glm-0.9.9.7, header-only)
// No reference source text is copied here: this file only includes the
// reference headers and calls their functions.
//
// What can NOT be reached this way: RayTracerBVH.cpp (GLSL compute + GL calls)
// -- there is no GL context in this image, and GL is not stubbed.  The BFS
// flatten below therefore walks the reference-built OctreeNode* tree with this
// repo's own restatement of setOctree's numbering (453-skeleton/RayTracerBVH.cpp:443-490).
#include "OctreeVoxel.h"
#include "Camera.h"
#include "Frustum.h"
#include "CacheUtils.h"

#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

namespace {
struct FlatNode {           // == GPUNodes, 453-skeleton/RayTracerBVH.h:21-26
    int32_t x, y, z, size, isLeaf, isSolid, isUniform;
    int32_t child[8];
};
struct GridPOD {            // == orc_grid
    int32_t dimX, dimY, dimZ;
    float minX, minY, minZ, voxelSize;
    uint8_t* data;
};

VoxelGrid toGrid(const GridPOD* g) {
    VoxelGrid grid;
    grid.dimX = g->dimX; grid.dimY = g->dimY; grid.dimZ = g->dimZ;
    grid.minX = g->minX; grid.minY = g->minY; grid.minZ = g->minZ;
    grid.voxelSize = g->voxelSize;
    size_t n = (size_t)g->dimX * g->dimY * g->dimZ;
    grid.data.resize(n);
    for (size_t i = 0; i < n; i++) grid.data[i] = g->data[i] ? VoxelState::FILLED : VoxelState::EMPTY;
    return grid;
}

// silence the reference's chatty std::cout while we call it
struct CoutMute {
    std::streambuf* old;
    std::ostringstream sink;
    CoutMute() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~CoutMute() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

// reference createOctreeFromVoxelGrid + BFS numbering of setOctree
int64_t ref_build_flat_octree(const GridPOD* g, FlatNode** out) {
    *out = nullptr;
    VoxelGrid grid = toGrid(g);
    OctreeNode* root = createOctreeFromVoxelGrid(grid);
    if (!root) return 0;
    std::vector<OctreeNode*> order;
    order.push_back(root);
    std::vector<FlatNode> flat(1);
    for (size_t head = 0; head < order.size(); head++) {
        OctreeNode* nd = order[head];
        FlatNode f;
        f.x = nd->x; f.y = nd->y; f.z = nd->z; f.size = nd->size;
        f.isLeaf = nd->isLeaf ? 1 : 0; f.isSolid = nd->isSolid ? 1 : 0; f.isUniform = nd->isUniform ? 1 : 0;
        for (int i = 0; i < 8; i++) f.child[i] = -1;
        if (!nd->isLeaf) {
            for (int i = 0; i < 8; i++) {
                if (nd->children[i]) {
                    f.child[i] = (int32_t)order.size();
                    order.push_back(nd->children[i]);
                    flat.emplace_back();
                }
            }
        }
        flat[head] = f;
    }
    freeOctree(root);
    FlatNode* o = (FlatNode*)std::malloc(flat.size() * sizeof(FlatNode));
    std::memcpy(o, flat.data(), flat.size() * sizeof(FlatNode));
    *out = o;
    return (int64_t)flat.size();
}

void ref_free(void* p) { std::free(p); }

int ref_get_voxel_safe(const GridPOD* g, int x, int y, int z) {
    VoxelGrid grid = toGrid(g);
    return (int)getVoxelSafe(grid, x, y, z);
}

// Camera(theta, phi, r); optional pan(dx, dy) first (as 453-skeleton/main.cpp:509,521 does)
void ref_camera(float theta, float phi, float radius, int doPan, float panDx, float panDy,
                float view[16], float pos[3], float target[3]) {
    Camera cam(theta, phi, radius);
    if (doPan) cam.pan(panDx, panDy);
    glm::mat4 v = cam.getView();
    std::memcpy(view, &v[0][0], 16 * sizeof(float));
    glm::vec3 p = cam.getPos();
    pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
    const glm::vec3& t = cam.getTarget();
    target[0] = t.x; target[1] = t.y; target[2] = t.z;
}

void ref_glm_inverse(const float m[16], float out[16]) {
    glm::mat4 a; std::memcpy(&a[0][0], m, 64);
    glm::mat4 r = glm::inverse(a);
    std::memcpy(out, &r[0][0], 64);
}

void ref_glm_mul(const float a_[16], const float b_[16], float out[16]) {
    glm::mat4 a, b; std::memcpy(&a[0][0], a_, 64); std::memcpy(&b[0][0], b_, 64);
    glm::mat4 r = a * b;
    std::memcpy(out, &r[0][0], 64);
}

void ref_glm_perspective(float fovyRad, float aspect, float zn, float zf, float out[16]) {
    glm::mat4 r = glm::perspective(fovyRad, aspect, zn, zf);
    std::memcpy(out, &r[0][0], 64);
}

float ref_glm_radians(float deg) { return glm::radians(deg); }

void ref_glm_normalize3(const float v[3], float out[3]) {
    glm::vec3 r = glm::normalize(glm::vec3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void ref_glm_normalize4(const float v[4], float out[4]) {
    glm::vec4 r = glm::normalize(glm::vec4(v[0], v[1], v[2], v[3]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void ref_glm_mat_vec(const float m_[16], const float v[4], float out[4]) {
    glm::mat4 m; std::memcpy(&m[0][0], m_, 64);
    glm::vec4 r = m * glm::vec4(v[0], v[1], v[2], v[3]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// Frustum(viewProj).testAABB(min, max, margin) for n boxes
void ref_frustum_test(const float vp_[16], const float* mins, const float* maxs, int64_t n, float margin, int32_t* out) {
    glm::mat4 vp; std::memcpy(&vp[0][0], vp_, 64);
    Frustum fr(vp);
    for (int64_t i = 0; i < n; i++)
        out[i] = fr.testAABB(glm::vec3(mins[3 * i], mins[3 * i + 1], mins[3 * i + 2]),
                             glm::vec3(maxs[3 * i], maxs[3 * i + 1], maxs[3 * i + 2]), margin);
}

// loadVoxelGrid; data is malloc'd (ref_free)
int ref_load_voxel_grid(const char* path, GridPOD* g) {
    CoutMute mute;
    VoxelGrid grid;
    if (!loadVoxelGrid(path, grid)) return 0;
    g->dimX = grid.dimX; g->dimY = grid.dimY; g->dimZ = grid.dimZ;
    g->minX = grid.minX; g->minY = grid.minY; g->minZ = grid.minZ; g->voxelSize = grid.voxelSize;
    g->data = (uint8_t*)std::malloc(grid.data.size() ? grid.data.size() : 1);
    for (size_t i = 0; i < grid.data.size(); i++) g->data[i] = (uint8_t)grid.data[i];
    return 1;
}

// localMC(grid, x0,y0,z0,size): returns triangle count; *out = 18 floats per triangle (v0,v1,v2, then the 3 normals)
int64_t ref_local_mc(const GridPOD* g, int x0, int y0, int z0, int size, float** out) {
    VoxelGrid grid = toGrid(g);
    std::vector<MCTriangle> tris = localMC(grid, x0, y0, z0, size);
    float* o = (float*)std::malloc((tris.size() ? tris.size() : 1) * 18 * sizeof(float));
    for (size_t i = 0; i < tris.size(); i++)
        for (int v = 0; v < 3; v++) {
            o[i * 18 + v * 3 + 0] = tris[i].v[v].x; o[i * 18 + v * 3 + 1] = tris[i].v[v].y; o[i * 18 + v * 3 + 2] = tris[i].v[v].z;
            o[i * 18 + 9 + v * 3 + 0] = tris[i].normal[v].x; o[i * 18 + 9 + v * 3 + 1] = tris[i].normal[v].y; o[i * 18 + 9 + v * 3 + 2] = tris[i].normal[v].z;
        }
    *out = o;
    return (int64_t)tris.size();
}

}  // extern "C"
