// oracle/glsl_driver.cpp -- TEST INFRASTRUCTURE.  Runs the reference's GLSL compute shader TEXT as C++ under glm semantics.
//
// The shader of RayTracerBVH (453-skeleton/RayTracerBVH.cpp:207-368, "S/RT") needs an OpenGL 4.3 compute context, which this image
// does not have.  But its functions (struct Ray, intersectAABB, intersectOctreeIterative, shade, generateRay: S/RT:221-355; and
// the earlier, block-commented shader at S/RT:46-166, the reference's closest-hit traversal) are plain GLSL that glm -- the
// reference's own vendored glm-0.9.9.7 -- compiles as C++ once `out T x` reads `T& x`.  `make -C oracle glsl` cuts those lines out
// of the reference file where it lies (sed, into oracle/_ref/*.inc: git-ignored, no reference text enters the repo) and compiles
// this driver around them with  g++ -O2 -ffp-contract=off -fsingle-precision-constant  (GLSL literals are float).  The uniforms
// and the SSBO are the globals below; glsl_render is main() (S/RT:357-367) over every pixel.
//
// What it pins: that oracle/rto_oracle.c -- the hand restatement everything else is compared with -- computes the same bits as
// the shader text does under glm's arithmetic (tests/test_oracle_golden.py, tests/golden/glsl_images_small.npz).  glm stands in
// for a GLSL compiler, so this is corroboration of A3-A6, not an execution of the reference.
#include <glm/glm.hpp>
#include <cstdint>
#include <cstring>
using namespace glm;

struct OctreeNodeGPUStruct { int x, y, z, size, isLeaf, isSolid, isUniform; int child[8]; };    // S/RT:195-204 (std430: 15 ints)
static_assert(sizeof(OctreeNodeGPUStruct) == 60, "GPUNodes layout");

#define MAX_TRAVERSAL_STEPS 512                                                                   // S/RT:192
static const OctreeNodeGPUStruct* nodes;                                                          // S/RT:206-208
static int numNodes;                                                                              // S/RT:210-219
static vec3 gridMin;
static float voxelSize;
static mat4 invVP, viewMat;
static vec3 cameraPos;
static float aspect, fov;
static int imageWidth, imageHeight;

#include GLSL_INC          // the shader's functions, cut from the reference by the Makefile

extern "C" void glsl_render(const void* nodes_, int n, const float* grid_min, float voxel_size, const float* view16, const float* cam_pos,
                            float aspect_, float fov_deg, int W, int H, float* rgba) {
    nodes = static_cast<const OctreeNodeGPUStruct*>(nodes_);
    numNodes = n;
    gridMin = vec3(grid_min[0], grid_min[1], grid_min[2]);
    voxelSize = voxel_size;
    std::memcpy(&viewMat, view16, sizeof(mat4));
    invVP = mat4(1.0f);
    cameraPos = vec3(cam_pos[0], cam_pos[1], cam_pos[2]);
    aspect = aspect_; fov = fov_deg; imageWidth = W; imageHeight = H;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {                                                             // main(), S/RT:357-367
            Ray ray = generateRay(x, y, imageWidth, imageHeight, cameraPos, viewMat, fov, aspect);
            vec3 hitPoint, hitNormal;
            bool hit = intersectOctreeIterative(ray.origin, ray.direction, hitPoint, hitNormal);
            vec3 color = hit ? shade(hitPoint, hitNormal) : vec3(0.0f);
            float* p = rgba + ((size_t)y * W + x) * 4;
            p[0] = color.x; p[1] = color.y; p[2] = color.z; p[3] = 1.0f;
        }
}
