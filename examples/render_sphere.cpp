// render_sphere.cpp -- the reference's call sequence (453-skeleton/main.cpp:1052-1070, 1127-1131, 1357-1363) in plain
// C++ against this repo's host layer: build the test-sphere grid, the octree, hand it to RayTracerBVH and render.
// No Python, no torch: g++ only; the class loads librto_hip.so (RTO_HIP_LIB or next to the binary) at run time.
//
//   make -C examples && RTO_HIP_LIB=ray_tracing_octrees_amd/librto_hip.so examples/render_sphere 256 1920 1080 out.ppm
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Camera.h"
#include "OctreeVoxel.h"
#include "RayTracerBVH.h"

int main(int argc, char** argv) {
    const int dim = argc > 1 ? std::atoi(argv[1]) : 256;
    const int W = argc > 2 ? std::atoi(argv[2]) : 1920;
    const int H = argc > 3 ? std::atoi(argv[3]) : 1080;
    const char* out = argc > 4 ? argv[4] : nullptr;
    if (dim <= 0 || W <= 0 || H <= 0) { std::fprintf(stderr, "usage: render_sphere [dim [width height [out.ppm]]]\n"); return 2; }

    VoxelGrid grid = makeTestSphereGrid(dim);                  // main.cpp:337-372, 1052-1070, 376-422
    OctreeNode* root = createOctreeFromVoxelGrid(grid);
    Camera camera(0.5f, 0.7f, 1.8f);

    RayTracerBVH bvhRayTracer;
    bvhRayTracer.ensureComputeInitialized();
    bvhRayTracer.setOctree(root, grid);
    if (bvhRayTracer.numNodes() <= 0 || !bvhRayTracer.lastError().empty()) {
        std::fprintf(stderr, "no MI355X path: %s\n", bvhRayTracer.lastError().c_str());
        return 1;                                               // there is no CPU fallback
    }
    const float aspect = float(W) / float(H);
    bvhRayTracer.renderSceneCompute(camera, W, H, aspect, 45.0f);          // warm-up (first launch, tables)
    const int frames = 100;
    bvhRayTracer.finish();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; i++) bvhRayTracer.renderSceneCompute(camera, W, H, aspect, 45.0f);
    bvhRayTracer.finish();                                      // the frames stay on the GPU, like the reference's texture
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / frames;
    // what 453-skeleton/main.cpp:1357-1363 does while the camera moves: frustum update + frame, every frame
    bvhRayTracer.setFrustumCullingEnabled(true);
    bvhRayTracer.renderSceneComputeWithCulling(camera, W, H, aspect, 45.0f, true);
    bvhRayTracer.finish();
    const auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; i++) bvhRayTracer.renderSceneComputeWithCulling(camera, W, H, aspect, 45.0f, true);
    bvhRayTracer.finish();
    const double sc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / frames;
    const std::vector<float>& fb = bvhRayTracer.framebuffer();
    size_t lit = 0;
    for (size_t p = 0; p < fb.size(); p += 4) lit += fb[p] != 0.0f;
    std::printf("%d^3 sphere, %d nodes, %dx%d: %zu lit pixels, %.4f ms per renderSceneCompute (frame left on the GPU)\n",
                dim, bvhRayTracer.numNodes(), W, H, lit, s * 1e3);
    std::printf("with a frustum update before every frame (renderSceneComputeWithCulling, updateFrustum = true): %.4f ms\n", sc * 1e3);
    if (out) {
        if (FILE* f = std::fopen(out, "wb")) {
            std::fprintf(f, "P6\n%d %d\n255\n", W, H);
            for (size_t p = 0; p < fb.size(); p += 4)
                for (int c = 0; c < 3; c++) std::fputc((int)std::lround(std::min(1.0f, std::max(0.0f, fb[p + c])) * 255.0f), f);
            std::fclose(f);
        }
    }
    freeOctree(root);
    return 0;
}
