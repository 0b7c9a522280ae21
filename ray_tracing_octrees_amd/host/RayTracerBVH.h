// RayTracerBVH.h -- drop-in for the reference class of the same name
// (453-skeleton/RayTracerBVH.h:28-80; used at 453-skeleton/main.cpp:1127-1131 and :1357-1363).
// Same public methods, same argument meaning, same error behaviour (void + a line on std::cerr);
// the OpenGL compute dispatch is replaced by hand-written gfx950 kernels behind the C ABI of
// include/rto_hip.h, loaded from librto_hip.so.  If that library or a gfx950 device is missing,
// ensureComputeInitialized() reports it on std::cerr and every render call keeps printing the
// reference's "[RayTracerBVH] Compute pipeline not initialized or failed." -- there is no CPU path.
//
// One addition: the reference leaves the image in a GL texture and never reads it back; here the
// last frame is available through framebuffer() (RGBA32F, row-major, row 0 = top).
#pragma once

#include <string>
#include <vector>

// "OctreeVoxel.h" / "Camera.h" resolve to this directory's headers by default.  Compiled with
// -DRTO_REFERENCE_HEADERS and the reference's 453-skeleton/ + glm first on the include path, the very same
// class builds against the reference's own OctreeNode / VoxelGrid / Camera (glm types): that is the
// drop-in build INTEGRATION.md describes; the repo's test tree builds and runs exactly that configuration.
#include "rto_hip.h"
#ifdef RTO_REFERENCE_HEADERS
#include <Camera.h>        // angle brackets: taken from the include path (the reference's 453-skeleton/), not from this directory
#include <OctreeVoxel.h>
#include <glm/glm.hpp>
namespace rto_host { using vec3 = glm::vec3; }
#else
#include "Camera.h"
#include "Frustum.h"
#include "OctreeVoxel.h"
#include "rtmath.h"
namespace rto_host { using vec3 = rtmath::vec3; }
#endif

// GPU-side node record, same name and layout as the reference's (RayTracerBVH.h:21-26)
struct GPUNodes {
    int x, y, z, size;
    int isLeaf, isSolid;
    int isUniform;
    int child[8];
};
static_assert(sizeof(GPUNodes) == sizeof(rto_node), "GPUNodes must stay 60 bytes");

struct Ray {
    rto_host::vec3 origin;
    rto_host::vec3 direction;
};

class RayTracerBVH {
public:
    RayTracerBVH();
    ~RayTracerBVH();
    RayTracerBVH(const RayTracerBVH&) = delete;
    RayTracerBVH& operator=(const RayTracerBVH&) = delete;

    void setOctree(OctreeNode* root, const VoxelGrid& grid);
    void ensureComputeInitialized();
    void renderSceneCompute(const Camera& camera, int width, int height, float aspect, float fovDeg);
    void setFrustumCullingEnabled(bool enabled) { m_frustumCullingEnabled = enabled; }
    void renderSceneComputeWithCulling(const Camera& camera, int width, int height, float aspect, float fovDeg,
                                       bool updateFrustum);

    // ---- additions (not in the reference) ----
    // createOctreeFromVoxelGrid + setOctree in one step ON THE GPU (rto_build_octree): no pointer tree, no host
    // flatten.  The resident array equals what setOctree(createOctreeFromVoxelGrid(grid), grid) uploads.
    void setOctreeFromGrid(const VoxelGrid& grid);
    // Triangle ray path of BASELINE config 5 (no upstream counterpart): builds, in HBM, the per-leaf Marching-Cubes
    // triangle buffer (what MarchingCubesRenderer::render emits per leaf) from the grid given to setOctree() /
    // setOctreeFromGrid(); then renders with Moeller-Trumbore hits on those triangles and, if `shadow`, one shadow ray
    // per hit.  Same framebuffer() as the other calls.
    void buildLeafTriangles();
    void renderSceneTriangles(const Camera& camera, int width, int height, float aspect, float fovDeg, bool shadow);
#ifndef RTO_REFERENCE_HEADERS   // needs this repo's localMC: the same buffer made on the host and uploaded (cross-check)
    void buildLeafTrianglesOnHost();
#endif
    // BFS numbering of setOctree (RayTracerBVH.cpp:443-490) without touching the GPU.
    static std::vector<GPUNodes> flatten(const OctreeNode* root);
    const std::vector<GPUNodes>& flatNodes() const { return m_flatNodes; }   // empty after setOctreeFromGrid
    int numNodes() const { return m_numNodes; }
    // The renders are asynchronous and leave the frame on the GPU, as the reference's texture is (its image is never
    // read back); framebuffer() copies it to the host on first use after a render (width*height*4 floats, row 0 = top)
    // and finish() just waits for the GPU -- the counterpart of glFinish for timing loops.
    const std::vector<float>& framebuffer() const;
    void finish() const;
    int frameWidth() const { return m_frameW; }
    int frameHeight() const { return m_frameH; }
    void setDevice(int ordinal) { m_device = ordinal; }                  // before ensureComputeInitialized()
    // Multi-GPU (no upstream counterpart; before ensureComputeInitialized()): the GPUs `m_device .. m_device + n - 1` of this
    // node each hold the octree and render the bands `b % n` of every frame (bands of bandRows rows; from 4 GPUs on the first
    // one only gathers and assembles and the others render the bands `b % (n - 1)`); ONE grouped RCCL
    // send/recv per frame lands the parts on the first GPU, which assembles the image (rto_comm_* in rto_hip.h).  The
    // reference's call sequence stays as it is; framebuffer() reads the assembled frame.
    void setDevices(int n, int bandRows = 16) { m_numDevices = n < 1 ? 1 : n; m_bandRows = bandRows; }
    int numDevices() const { return m_numDevices; }
    rto_context* context() const { return m_ctx; }
    const std::string& lastError() const { return m_lastError; }

private:
    bool render(const Camera& camera, int width, int height, float aspect, float fovDeg);

    OctreeNode* m_octreeRoot;
    VoxelGrid m_grid;
    std::vector<GPUNodes> m_flatNodes;
    int m_numNodes;

    bool m_computeInited;
    bool m_computeOk;
    bool m_frustumCullingEnabled;
    int m_device;
    int m_numDevices = 1, m_bandRows = 16;
    rto_context* m_ctx;                       // the first GPU's context (== m_ctxs[0])
    std::vector<rto_context*> m_ctxs;         // one per GPU
    std::vector<rto_comm*> m_comms;           // setDevices(n > 1): the single-process communicator group
    template <class F> bool forEachContext(F&& call, const char* what);
    bool renderFrame(const rto_frame& f, int mode);
    mutable std::string m_lastError;

    mutable std::vector<float> m_frame;
    mutable bool m_frameStale = false;   // the GPU holds a newer frame than m_frame
    int m_frameW, m_frameH;
};
