// RayTracerBVH.cpp -- host side of the drop-in class; see RayTracerBVH.h.
// Call structure follows 453-skeleton/RayTracerBVH.cpp:393-892; every GL call there maps to one
// rto_* call here (table in INTEGRATION.md).
#include "RayTracerBVH.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace {

// The C ABI, resolved at run time so this translation unit builds with a plain C++ compiler.
struct HipApi {
    void* handle = nullptr;
    decltype(&rto_create) create = nullptr;
    decltype(&rto_destroy) destroy = nullptr;
    decltype(&rto_last_error) last_error = nullptr;
    decltype(&rto_upload_octree) upload_octree = nullptr;
    decltype(&rto_build_octree) build_octree = nullptr;
    decltype(&rto_octree_info_get) octree_info = nullptr;
    decltype(&rto_update_frustum) update_frustum = nullptr;
    decltype(&rto_render_host) render_host = nullptr;
    decltype(&rto_upload_leaf_triangles) upload_leaf_triangles = nullptr;
    decltype(&rto_build_leaf_triangles) build_leaf_triangles = nullptr;
    decltype(&rto_render_resident) render_resident = nullptr;
    decltype(&rto_download_resident) download_resident = nullptr;
    decltype(&rto_synchronize) synchronize = nullptr;
    decltype(&rto_timing_begin) timing_begin = nullptr;
    decltype(&rto_render_triangles_host) render_triangles_host = nullptr;
    decltype(&rto_comm_create_all) comm_create_all = nullptr;
    decltype(&rto_comm_destroy) comm_destroy = nullptr;
    decltype(&rto_comm_last_error) comm_last_error = nullptr;
    decltype(&rto_comm_render_resident_all) comm_render_resident_all = nullptr;
    std::string error;

    bool load() {
        if (handle) return true;
        std::vector<std::string> candidates;
        if (const char* env = std::getenv("RTO_HIP_LIB")) { if (*env) candidates.emplace_back(env); }      // set but empty: as if unset
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&anchor), &info) && info.dli_fname) {   // next to this library
            std::string dir(info.dli_fname);
            const size_t slash = dir.find_last_of('/');
            dir = slash == std::string::npos ? "." : dir.substr(0, slash);
            candidates.push_back(dir + "/librto_hip.so");
        }
        candidates.emplace_back("librto_hip.so");
        for (const std::string& path : candidates) {
            handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
            error = dlerror();
        }
        if (!handle) return false;
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(handle, name);
            if (!p) { ok = false; error = std::string("missing symbol ") + name; }
            return p;
        };
        create = reinterpret_cast<decltype(create)>(sym("rto_create"));
        destroy = reinterpret_cast<decltype(destroy)>(sym("rto_destroy"));
        last_error = reinterpret_cast<decltype(last_error)>(sym("rto_last_error"));
        upload_octree = reinterpret_cast<decltype(upload_octree)>(sym("rto_upload_octree"));
        build_octree = reinterpret_cast<decltype(build_octree)>(sym("rto_build_octree"));
        octree_info = reinterpret_cast<decltype(octree_info)>(sym("rto_octree_info_get"));
        update_frustum = reinterpret_cast<decltype(update_frustum)>(sym("rto_update_frustum"));
        render_host = reinterpret_cast<decltype(render_host)>(sym("rto_render_host"));
        upload_leaf_triangles = reinterpret_cast<decltype(upload_leaf_triangles)>(sym("rto_upload_leaf_triangles"));
        build_leaf_triangles = reinterpret_cast<decltype(build_leaf_triangles)>(sym("rto_build_leaf_triangles"));
        render_resident = reinterpret_cast<decltype(render_resident)>(sym("rto_render_resident"));
        download_resident = reinterpret_cast<decltype(download_resident)>(sym("rto_download_resident"));
        synchronize = reinterpret_cast<decltype(synchronize)>(sym("rto_synchronize"));
        timing_begin = reinterpret_cast<decltype(timing_begin)>(sym("rto_timing_begin"));
        render_triangles_host = reinterpret_cast<decltype(render_triangles_host)>(sym("rto_render_triangles_host"));
        comm_create_all = reinterpret_cast<decltype(comm_create_all)>(sym("rto_comm_create_all"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(sym("rto_comm_destroy"));
        comm_last_error = reinterpret_cast<decltype(comm_last_error)>(sym("rto_comm_last_error"));
        comm_render_resident_all = reinterpret_cast<decltype(comm_render_resident_all)>(sym("rto_comm_render_resident_all"));
        if (!ok) { dlclose(handle); handle = nullptr; }
        return ok;
    }
    static void anchor() {}
};

HipApi& api() {
    static HipApi a;
    return a;
}

}  // namespace

RayTracerBVH::RayTracerBVH()
    : m_octreeRoot(nullptr), m_numNodes(0), m_computeInited(false), m_computeOk(false),
      m_frustumCullingEnabled(true), m_device(0), m_ctx(nullptr), m_frameW(0), m_frameH(0) {}

RayTracerBVH::~RayTracerBVH() {
    for (rto_comm* m : m_comms) if (m) api().comm_destroy(m);
    for (rto_context* c : m_ctxs) if (c) api().destroy(c);
}

// Runs `call(ctx)` (an rto_* call returning RTO_OK or an error code) on every GPU's context.
template <class F> bool RayTracerBVH::forEachContext(F&& call, const char* what) {
    for (rto_context* c : m_ctxs) {
        if (call(c) != RTO_OK) {
            m_lastError = api().last_error(c);
            std::cerr << "[RayTracerBVH] " << what << " failed: " << m_lastError << std::endl;
            return false;
        }
    }
    return true;
}

// One frame into the (first GPU's) resident framebuffer: a plain render, or the split over all GPUs + one gather.
bool RayTracerBVH::renderFrame(const rto_frame& f, int mode) {
    if (m_comms.empty()) {
        if (api().render_resident(m_ctx, &f, mode) == RTO_OK) return true;
        m_lastError = api().last_error(m_ctx);
        return false;
    }
    if (api().comm_render_resident_all(m_comms.data(), (int)m_comms.size(), &f, mode) == RTO_OK) return true;
    m_lastError = api().comm_last_error(m_comms[0]);
    return false;
}

std::vector<GPUNodes> RayTracerBVH::flatten(const OctreeNode* root) {
    // Breadth-first; a child receives the next free index at the moment its parent is dequeued, so the 8
    // children of an internal node are consecutive and the root is 0 (RayTracerBVH.cpp:443-490).
    std::vector<GPUNodes> flat;
    if (!root) return flat;
    std::vector<const OctreeNode*> order;
    order.push_back(root);
    for (size_t head = 0; head < order.size(); head++) {
        const OctreeNode* nd = order[head];
        GPUNodes g;
        g.x = nd->x; g.y = nd->y; g.z = nd->z; g.size = nd->size;
        g.isLeaf = nd->isLeaf ? 1 : 0;
        g.isSolid = nd->isSolid ? 1 : 0;
        g.isUniform = nd->isUniform ? 1 : 0;
        for (int& c : g.child) c = -1;
        if (!nd->isLeaf)
            for (int i = 0; i < 8; i++)
                if (const OctreeNode* c = nd->children[i]) {
                    g.child[i] = static_cast<int>(order.size());
                    order.push_back(c);
                }
        flat.push_back(g);
    }
    return flat;
}

void RayTracerBVH::setOctree(OctreeNode* root, const VoxelGrid& grid) {
    m_octreeRoot = root;
    m_grid = grid;                      // the reference copies the grid too (RayTracerBVH.cpp:433)
    m_flatNodes.clear();
    m_numNodes = 0;
    if (!root) return;                  // :439
    m_flatNodes = flatten(root);
    m_numNodes = static_cast<int>(m_flatNodes.size());
    if (!m_ctx) return;                 // uploaded by ensureComputeInitialized() once the device exists
    const float gridMin[3] = { m_grid.minX, m_grid.minY, m_grid.minZ };
    if (!forEachContext([&](rto_context* c) {
            return api().upload_octree(c, reinterpret_cast<const rto_node*>(m_flatNodes.data()), m_numNodes, gridMin, m_grid.voxelSize);
        }, "octree upload"))
        m_computeOk = false;
}

void RayTracerBVH::setOctreeFromGrid(const VoxelGrid& grid) {
    m_octreeRoot = nullptr;
    m_grid = grid;
    m_flatNodes.clear();
    m_numNodes = 0;
    if (!m_computeInited) ensureComputeInitialized();
    if (!m_computeOk || grid.dimX <= 0 || grid.dimY <= 0 || grid.dimZ <= 0) return;
    const float gridMin[3] = { grid.minX, grid.minY, grid.minZ };
    static_assert(sizeof(VoxelState) == 1, "VoxelGrid.data is one byte per voxel");
    if (!forEachContext([&](rto_context* c) {
            return api().build_octree(c, reinterpret_cast<const uint8_t*>(grid.data.data()), grid.dimX, grid.dimY, grid.dimZ, gridMin, grid.voxelSize);
        }, "GPU octree build"))
        return;
    rto_octree_info info;
    if (api().octree_info(m_ctx, &info) == RTO_OK) m_numNodes = static_cast<int>(info.num_nodes);
}

void RayTracerBVH::ensureComputeInitialized() {
    if (m_computeInited) return;
    m_computeInited = true;
    if (!api().load()) {
        m_lastError = "cannot load librto_hip.so: " + api().error;
        std::cerr << "[RayTracerBVH] " << m_lastError << std::endl;
        return;
    }
    for (int i = 0; i < m_numDevices; i++) {
        rto_context* c = nullptr;
        if (api().create(m_device + i, &c) != RTO_OK) {
            m_lastError = api().last_error(nullptr);
            std::cerr << "[RayTracerBVH] " << m_lastError << std::endl;
            for (rto_context* d : m_ctxs) api().destroy(d);
            m_ctxs.clear();
            m_ctx = nullptr;
            return;
        }
        m_ctxs.push_back(c);
        api().timing_begin(c, -1);                     // nobody reads kernel timings through this class: no event pair per launch
    }
    m_ctx = m_ctxs[0];
    if (m_numDevices > 1) {
        m_comms.assign((size_t)m_numDevices, nullptr);
        if (api().comm_create_all(m_ctxs.data(), m_numDevices, m_bandRows, m_comms.data()) != RTO_OK) {
            m_lastError = api().last_error(m_ctx);
            std::cerr << "[RayTracerBVH] " << m_lastError << std::endl;
            m_comms.clear();
            return;                                    // m_computeOk stays false: no silent single-GPU fallback
        }
    }
    m_computeOk = true;
    if (m_numNodes > 0 && !m_flatNodes.empty()) {   // setOctree() came first
        const float gridMin[3] = { m_grid.minX, m_grid.minY, m_grid.minZ };
        if (!forEachContext([&](rto_context* c) {
                return api().upload_octree(c, reinterpret_cast<const rto_node*>(m_flatNodes.data()), m_numNodes, gridMin, m_grid.voxelSize);
            }, "octree upload"))
            m_computeOk = false;
    }
}

bool RayTracerBVH::render(const Camera& camera, int width, int height, float aspect, float fovDeg) {
    rto_frame f;
    const auto view = camera.getView();            // rtmath::mat4 or glm::mat4: both column-major, m[col][row]
    std::memcpy(f.view, &view[0][0], sizeof f.view);
    const auto pos = camera.getPos();
    f.cam_pos[0] = pos.x; f.cam_pos[1] = pos.y; f.cam_pos[2] = pos.z;
    f.aspect = aspect;
    f.fov_deg = fovDeg;
    f.width = width;
    f.height = height;
    if (width <= 0 || height <= 0) return false;
    // like the reference's texture, the frame stays on the GPU (asynchronous); framebuffer() fetches it on demand
    m_frameW = m_frameH = 0; m_frameStale = false;
    if (!renderFrame(f, RTO_RESIDENT_OCTREE)) {
        std::cerr << "[RayTracerBVH] render failed: " << m_lastError << std::endl;
        m_frame.clear();
        return false;
    }
    m_frameW = width; m_frameH = height; m_frameStale = true;
    return true;
}

void RayTracerBVH::buildLeafTriangles() {
    if (!m_computeInited || !m_computeOk) {
        std::cerr << "[RayTracerBVH] Compute pipeline not initialized or failed.\n";
        return;
    }
    if (m_numNodes <= 0) return;
    // the triangles MarchingCubesRenderer::render would emit per leaf, built in HBM from the grid given to setOctree()
    // (after setOctreeFromGrid the voxels are already resident: NULL)
    static_assert(sizeof(VoxelState) == 1, "VoxelState is a byte upstream (S/OctreeVoxel.h:10-13)");
    const uint8_t* vox = m_flatNodes.empty() ? nullptr : reinterpret_cast<const uint8_t*>(m_grid.data.data());
    forEachContext([&](rto_context* c) { return api().build_leaf_triangles(c, vox, m_grid.dimX, m_grid.dimY, m_grid.dimZ); }, "leaf-triangle build");
}

#ifndef RTO_REFERENCE_HEADERS
// The same buffer made by this repo's host builder (localMC per leaf) and uploaded: the cross-check of the GPU build.
void RayTracerBVH::buildLeafTrianglesOnHost() {
    if (!m_computeInited || !m_computeOk) {
        std::cerr << "[RayTracerBVH] Compute pipeline not initialized or failed.\n";
        return;
    }
    if (m_flatNodes.empty()) return;
    std::vector<float> tris;
    std::vector<int32_t> off;
    ::buildLeafTriangles(m_grid, GPUNodesView{ reinterpret_cast<const int32_t*>(m_flatNodes.data()), (int64_t)m_flatNodes.size() }, tris, off);
    forEachContext([&](rto_context* c) { return api().upload_leaf_triangles(c, tris.data(), (int64_t)(tris.size() / 12), off.data()); }, "triangle upload");
}
#endif

void RayTracerBVH::renderSceneTriangles(const Camera& camera, int width, int height, float aspect, float fovDeg, bool shadow) {
    if (!m_computeInited || !m_computeOk) {
        std::cerr << "[RayTracerBVH] Compute pipeline not initialized or failed.\n";
        return;
    }
    if (m_numNodes <= 0 || width <= 0 || height <= 0) return;
    rto_frame f;
    const auto view = camera.getView();
    std::memcpy(f.view, &view[0][0], sizeof f.view);
    const auto pos = camera.getPos();
    f.cam_pos[0] = pos.x; f.cam_pos[1] = pos.y; f.cam_pos[2] = pos.z;
    f.aspect = aspect; f.fov_deg = fovDeg; f.width = width; f.height = height;
    m_frameW = m_frameH = 0; m_frameStale = false;
    if (!renderFrame(f, shadow ? RTO_RESIDENT_TRIANGLES_SHADOW : RTO_RESIDENT_TRIANGLES)) {
        std::cerr << "[RayTracerBVH] render failed: " << m_lastError << std::endl;
        m_frame.clear();
        return;
    }
    m_frameW = width; m_frameH = height; m_frameStale = true;
}

const std::vector<float>& RayTracerBVH::framebuffer() const {
    if (m_frameStale && m_ctx) {
        m_frame.resize(static_cast<size_t>(m_frameW) * m_frameH * 4);
        if (api().download_resident(m_ctx, m_frame.data()) != RTO_OK) {
            m_lastError = api().last_error(m_ctx);
            std::cerr << "[RayTracerBVH] frame read-back failed: " << m_lastError << std::endl;
            m_frame.clear();
        }
        m_frameStale = false;
    }
    return m_frame;
}

void RayTracerBVH::finish() const {
    for (rto_context* c : m_ctxs) (void)api().synchronize(c);
}

void RayTracerBVH::renderSceneCompute(const Camera& camera, int width, int height, float aspect, float fovDeg) {
    if (!m_computeInited || !m_computeOk) {
        std::cerr << "[RayTracerBVH] Compute pipeline not initialized or failed.\n";
        return;
    }
    if (m_numNodes <= 0) return;
    render(camera, width, height, aspect, fovDeg);
}

void RayTracerBVH::renderSceneComputeWithCulling(const Camera& camera, int width, int height, float aspect,
                                                 float fovDeg, bool updateFrustum) {
    if (!m_computeInited || !m_computeOk) {
        std::cerr << "[RayTracerBVH] Compute pipeline not initialized or failed.\n";
        return;
    }
    if (m_numNodes <= 0) {
        std::printf("No nodes to render.\n");
        return;
    }
    if (updateFrustum) {
        // the reference recomputes visibility on the CPU and re-uploads the compacted array
        // (RayTracerBVH.cpp:725-813); here the same test and compaction run on the GPU.
        const auto view = camera.getView();
        if (!forEachContext([&](rto_context* c) { return api().update_frustum(c, &view[0][0], fovDeg, aspect, 1); }, "frustum update")) return;
    }
    render(camera, width, height, aspect, fovDeg);
}
