// host_capi.cpp -- extern "C" doors onto the C++ host layer so Python (tests, bench.py) can drive the
// same classes a C++ application would: VoxelGrid / createOctreeFromVoxelGrid / Camera / Frustum /
// CacheUtils / RayTracerBVH.  Pure plumbing; no algorithm lives here.
#include <cstdint>
#include <cstring>
#include <new>
#include <string>

#include "CacheUtils.h"
#include "Camera.h"
#include "Frustum.h"
#include "OctreeVoxel.h"
#include "RayTracerBVH.h"
#include "Renderer.h"

extern "C" {

// ---------------------------------------------------------------- VoxelGrid
VoxelGrid* rtoh_grid_new(int dx, int dy, int dz, float minx, float miny, float minz, float voxelSize, const uint8_t* data) {
    VoxelGrid* g = new VoxelGrid();
    g->dimX = dx; g->dimY = dy; g->dimZ = dz;
    g->minX = minx; g->minY = miny; g->minZ = minz;
    g->voxelSize = voxelSize;
    const size_t n = (size_t)dx * dy * dz;
    g->data.resize(n);
    if (data) for (size_t i = 0; i < n; i++) g->data[i] = data[i] ? VoxelState::FILLED : VoxelState::EMPTY;
    return g;
}
VoxelGrid* rtoh_grid_test_sphere(int dim) { return new VoxelGrid(makeTestSphereGrid(dim)); }
VoxelGrid* rtoh_grid_load(const char* path) {
    VoxelGrid* g = new VoxelGrid();
    if (!loadVoxelGrid(path, *g)) { delete g; return nullptr; }
    return g;
}
VoxelGrid* rtoh_grid_load_partial(const char* path, int startLayer, int numLayers) {
    VoxelGrid* g = new VoxelGrid();
    if (!loadVoxelGridPartial(path, *g, startLayer, numLayers)) { delete g; return nullptr; }
    return g;
}
int rtoh_grid_save(const VoxelGrid* g, const char* path) { return saveVoxelGrid(path, *g) ? 1 : 0; }
void rtoh_grid_free(VoxelGrid* g) { delete g; }
void rtoh_grid_info(const VoxelGrid* g, int dims[3], float mn[3], float* voxelSize) {
    dims[0] = g->dimX; dims[1] = g->dimY; dims[2] = g->dimZ;
    mn[0] = g->minX; mn[1] = g->minY; mn[2] = g->minZ;
    *voxelSize = g->voxelSize;
}
void rtoh_grid_data(const VoxelGrid* g, uint8_t* out) {
    for (size_t i = 0; i < g->data.size(); i++) out[i] = (uint8_t)g->data[i];
}
int64_t rtoh_grid_count(const VoxelGrid* g) { return (int64_t)g->data.size(); }
int rtoh_grid_recenter(VoxelGrid* g) { return recenterFilledVoxels(*g) ? 1 : 0; }
int rtoh_get_voxel_safe(const VoxelGrid* g, int x, int y, int z) { return (int)getVoxelSafe(*g, x, y, z); }

// ---------------------------------------------------------------- octree
OctreeNode* rtoh_octree_build(const VoxelGrid* g) { return createOctreeFromVoxelGrid(*g); }
void rtoh_octree_free(OctreeNode* root) { freeOctree(root); }
int64_t rtoh_octree_map_size() { return (int64_t)g_octreeMap.size(); }
// flatten into caller memory; returns the node count (call with out == NULL to size the buffer)
int64_t rtoh_octree_flatten(const OctreeNode* root, GPUNodes* out, int64_t capacity) {
    const std::vector<GPUNodes> flat = RayTracerBVH::flatten(root);
    if (out && capacity >= (int64_t)flat.size()) std::memcpy(out, flat.data(), flat.size() * sizeof(GPUNodes));
    return (int64_t)flat.size();
}
int rtoh_octree_neighbors(const VoxelGrid* g, int x, int y, int z, int size, int* outXYZS /* 6*4 */) {
    (void)g; (void)size;
    auto it = g_octreeMap.find(buildKey(x, y, z));
    if (it == g_octreeMap.end()) return -1;
    const std::vector<OctreeNode*> nb = getNeighbors(it->second, g_octreeMap);
    for (size_t i = 0; i < nb.size() && i < 6; i++) {
        outXYZS[i * 4 + 0] = nb[i]->x; outXYZS[i * 4 + 1] = nb[i]->y; outXYZS[i * 4 + 2] = nb[i]->z; outXYZS[i * 4 + 3] = nb[i]->size;
    }
    return (int)nb.size();
}

// localMC / MarchingCubesRenderer: 18 floats per triangle (v0,v1,v2, n0,n1,n2); call with out == NULL to size
static int64_t copy_tris(const std::vector<MCTriangle>& tris, float* out, int64_t capacityTris) {
    if (out && capacityTris >= (int64_t)tris.size())
        for (size_t i = 0; i < tris.size(); i++)
            for (int v = 0; v < 3; v++)
                for (int a = 0; a < 3; a++) {
                    out[i * 18 + v * 3 + a] = tris[i].v[v][a];
                    out[i * 18 + 9 + v * 3 + a] = tris[i].normal[v][a];
                }
    return (int64_t)tris.size();
}
int64_t rtoh_local_mc(const VoxelGrid* g, int x0, int y0, int z0, int size, float* out, int64_t capacityTris) {
    return copy_tris(localMC(*g, x0, y0, z0, size), out, capacityTris);
}
int64_t rtoh_mc_renderer(const OctreeNode* root, const VoxelGrid* g, float* out, int64_t capacityTris) {
    MarchingCubesRenderer r;
    return copy_tris(r.render(root, *g, root ? root->x : 0, root ? root->y : 0, root ? root->z : 0, root ? root->size : 0), out, capacityTris);
}

// leaf-triangle buffer for the triangle ray path: returns the triangle count; call with tris == NULL to size
int64_t rtoh_build_leaf_triangles(const VoxelGrid* g, const GPUNodes* nodes, int64_t n, float* tris, int64_t capacityTris, int32_t* triOffset) {
    std::vector<float> t;
    std::vector<int32_t> off;
    buildLeafTriangles(*g, GPUNodesView{ reinterpret_cast<const int32_t*>(nodes), n }, t, off);
    const int64_t count = (int64_t)(t.size() / 12);
    if (tris && capacityTris >= count && triOffset) {
        std::memcpy(tris, t.data(), t.size() * sizeof(float));
        std::memcpy(triOffset, off.data(), off.size() * sizeof(int32_t));
    }
    return count;
}

// ---------------------------------------------------------------- Camera
Camera* rtoh_camera_new(float theta, float phi, float radius) { return new Camera(theta, phi, radius); }
void rtoh_camera_free(Camera* c) { delete c; }
void rtoh_camera_pan(Camera* c, float dx, float dy) { c->pan(dx, dy); }
void rtoh_camera_increment(Camera* c, float dTheta, float dPhi, float dR) {
    c->incrementTheta(dTheta); c->incrementPhi(dPhi); c->incrementR(dR);
}
void rtoh_camera_set_target(Camera* c, const float t[3]) { c->setTarget(rtmath::vec3(t[0], t[1], t[2])); }
void rtoh_camera_get(const Camera* c, float view[16], float pos[3], float target[3], float tpr[3]) {
    const rtmath::mat4 v = c->getView();
    std::memcpy(view, v.data(), 64);
    const rtmath::vec3 p = c->getPos();
    pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
    target[0] = c->target.x; target[1] = c->target.y; target[2] = c->target.z;
    tpr[0] = c->theta; tpr[1] = c->phi; tpr[2] = c->radius;
}

// ---------------------------------------------------------------- math / frustum
void rtoh_mat4_inverse(const float m[16], float out[16]) { std::memcpy(out, rtmath::inverse(rtmath::mat4::from(m)).data(), 64); }
void rtoh_mat4_mul(const float a[16], const float b[16], float out[16]) {
    std::memcpy(out, (rtmath::mat4::from(a) * rtmath::mat4::from(b)).data(), 64);
}
void rtoh_perspective(float fovyRad, float aspect, float zn, float zf, float out[16]) {
    std::memcpy(out, rtmath::perspective(fovyRad, aspect, zn, zf).data(), 64);
}
float rtoh_radians(float deg) { return rtmath::radians(deg); }
void rtoh_frustum_test(const float vp[16], const float* mins, const float* maxs, int64_t n, float margin, int32_t* out) {
    const Frustum fr(rtmath::mat4::from(vp));
    for (int64_t i = 0; i < n; i++)
        out[i] = fr.testAABB(rtmath::vec3(mins[3 * i], mins[3 * i + 1], mins[3 * i + 2]),
                             rtmath::vec3(maxs[3 * i], maxs[3 * i + 1], maxs[3 * i + 2]), margin);
}

// ---------------------------------------------------------------- RayTracerBVH
RayTracerBVH* rtoh_rt_new(int device) {
    RayTracerBVH* rt = new RayTracerBVH();
    rt->setDevice(device);
    return rt;
}
void rtoh_rt_free(RayTracerBVH* rt) { delete rt; }
void rtoh_rt_set_devices(RayTracerBVH* rt, int n, int bandRows) { rt->setDevices(n, bandRows); }
void rtoh_rt_ensure_compute_initialized(RayTracerBVH* rt) { rt->ensureComputeInitialized(); }
void rtoh_rt_set_octree(RayTracerBVH* rt, OctreeNode* root, const VoxelGrid* g) { rt->setOctree(root, *g); }
void rtoh_rt_set_octree_from_grid(RayTracerBVH* rt, const VoxelGrid* g) { rt->setOctreeFromGrid(*g); }
void rtoh_rt_set_frustum_culling_enabled(RayTracerBVH* rt, int enabled) { rt->setFrustumCullingEnabled(enabled != 0); }
void rtoh_rt_render_scene_compute(RayTracerBVH* rt, const Camera* cam, int w, int h, float aspect, float fovDeg) {
    rt->renderSceneCompute(*cam, w, h, aspect, fovDeg);
}
void rtoh_rt_render_scene_compute_with_culling(RayTracerBVH* rt, const Camera* cam, int w, int h, float aspect,
                                               float fovDeg, int updateFrustum) {
    rt->renderSceneComputeWithCulling(*cam, w, h, aspect, fovDeg, updateFrustum != 0);
}
void rtoh_rt_build_leaf_triangles(RayTracerBVH* rt) { rt->buildLeafTriangles(); }
void rtoh_rt_build_leaf_triangles_on_host(RayTracerBVH* rt) { rt->buildLeafTrianglesOnHost(); }
void rtoh_rt_render_scene_triangles(RayTracerBVH* rt, const Camera* cam, int w, int h, float aspect, float fovDeg, int shadow) {
    rt->renderSceneTriangles(*cam, w, h, aspect, fovDeg, shadow != 0);
}
int64_t rtoh_rt_num_nodes(const RayTracerBVH* rt) { return (int64_t)rt->numNodes(); }
int rtoh_rt_framebuffer(const RayTracerBVH* rt, float* out, int64_t capacityFloats, int* w, int* h) {
    *w = rt->frameWidth(); *h = rt->frameHeight();
    const std::vector<float>& fb = rt->framebuffer();
    if (fb.empty()) return 0;
    if (out && capacityFloats >= (int64_t)fb.size()) std::memcpy(out, fb.data(), fb.size() * sizeof(float));
    return 1;
}
void rtoh_rt_finish(const RayTracerBVH* rt) { rt->finish(); }
void* rtoh_rt_context(const RayTracerBVH* rt) { return rt->context(); }
const char* rtoh_rt_last_error(const RayTracerBVH* rt) { return rt->lastError().c_str(); }

}  // extern "C"
