// ref_shim_vr.cpp -- ORACLE-SIDE test infrastructure (never shipped, never linked by the product).
//
// A door onto the reference's only CPU ray/octree code: the file-static function
//   octreeRaySkip(node, ro, rd, tMin, tMax, grid, visibility, radiationTex)   453-skeleton/VolumeRaycastRenderer.cpp:50-155
// A static function can only be reached from inside its translation unit, so this TU #includes the reference's
// .cpp where it lies (no text of it is copied).  The reference file also holds the GL renderer class; it compiles
// against the reference's own vendored headers (thirdparty/glad-opengl-4.6-core, glfw-3.4, glm-0.9.9.7) and is
// linked with the reference's own vendored glad.c (the table of GL function pointers, all null here: no GL call is ever
// made through this door) and the reference objects Camera.o / Frustum.o / OctreeVoxel.o / Renderer.o.  Nothing is
// stubbed or stood in for.  See oracle/Makefile target `refvr`; output only into oracle/_ref/.
#include "/root/reference/453-skeleton/VolumeRaycastRenderer.cpp"

#include <cstdint>
#include <cstring>

namespace {
struct VrGridPOD {          // == orc_grid / ref_shim.cpp's GridPOD
    int32_t dimX, dimY, dimZ;
    float minX, minY, minZ, voxelSize;
    uint8_t* data;
};
}  // namespace

extern "C" {

// For each of the n directions rd[i] (origin ro): out[i] = octreeRaySkip(root, ro, rd[i], tMin, tMax, grid, vis).
// The tree is the reference's own createOctreeFromVoxelGrid(grid).  nodeVisible (optional): one flag per node in the BFS
// numbering of setOctree (453-skeleton/RayTracerBVH.cpp:443-490, the order ref_build_flat_octree emits); it becomes the
// unordered_map<const OctreeNode*, bool> the reference consults at :64-67.  Returns the node count (0 on failure).
int64_t refvr_octree_ray_skip(const VrGridPOD* g, const float ro[3], const float* rd, int64_t n, float tMin, float tMax,
                              const uint8_t* nodeVisible, int64_t numFlags, float* out) {
    VoxelGrid grid;
    grid.dimX = g->dimX; grid.dimY = g->dimY; grid.dimZ = g->dimZ;
    grid.minX = g->minX; grid.minY = g->minY; grid.minZ = g->minZ;
    grid.voxelSize = g->voxelSize;
    const size_t cells = (size_t)g->dimX * g->dimY * g->dimZ;
    grid.data.resize(cells);
    for (size_t i = 0; i < cells; i++) grid.data[i] = g->data[i] ? VoxelState::FILLED : VoxelState::EMPTY;
    OctreeNode* root = createOctreeFromVoxelGrid(grid);
    if (!root) return 0;
    std::vector<const OctreeNode*> order;          // BFS numbering
    order.push_back(root);
    for (size_t head = 0; head < order.size(); head++) {
        const OctreeNode* nd = order[head];
        if (nd->isLeaf) continue;
        for (int i = 0; i < 8; i++)
            if (nd->children[i]) order.push_back(nd->children[i]);
    }
    std::unordered_map<const OctreeNode*, bool> vis;
    if (nodeVisible) {
        if (numFlags != (int64_t)order.size()) { freeOctree(root); return 0; }
        for (size_t i = 0; i < order.size(); i++) vis[order[i]] = nodeVisible[i] != 0;
    }
    const glm::vec3 o(ro[0], ro[1], ro[2]);
    for (int64_t i = 0; i < n; i++) {
        const glm::vec3 d(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        out[i] = octreeRaySkip(root, o, d, tMin, tMax, grid, nodeVisible ? &vis : nullptr, 0);
    }
    const int64_t count = (int64_t)order.size();
    freeOctree(root);
    return count;
}

// The 7x7 probe directions of drawRaycast (453-skeleton/VolumeRaycastRenderer.cpp:1602-1630): same glm calls, same
// order, for the view matrix / eye the caller got from the reference Camera.  rd: 49 x 3.
void refvr_probe_rays(const float view[16], const float eye[3], float aspect, float* rd) {
    const int gridSize = 7;
    const float sampleOffset = 0.2f;
    glm::mat4 V; std::memcpy(&V[0][0], view, 64);
    glm::mat4 P = glm::perspective(glm::radians(45.0f), aspect, 0.1f, 5000.0f);
    glm::mat4 invV = glm::inverse(V);
    glm::mat4 invP = glm::inverse(P);
    glm::vec3 ro(eye[0], eye[1], eye[2]);
    for (int y = 0; y < gridSize; y++)
        for (int x = 0; x < gridSize; x++) {
            float ndcX = ((float)x / (gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
            float ndcY = ((float)y / (gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
            glm::vec4 clipPos(ndcX, ndcY, 1.f, 1.f);
            glm::vec4 viewPos = invP * clipPos;
            viewPos /= viewPos.w;
            glm::vec4 worldPos4 = invV * viewPos;
            glm::vec3 d = glm::normalize(glm::vec3(worldPos4) - ro);
            rd[3 * (y * gridSize + x) + 0] = d.x; rd[3 * (y * gridSize + x) + 1] = d.y; rd[3 * (y * gridSize + x) + 2] = d.z;
        }
}

}  // extern "C"
