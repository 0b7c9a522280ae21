// Renderer.cpp -- MarchingCubesRenderer::render, behaviour of 453-skeleton/Renderer.cpp:14-36.
#include "Renderer.h"

std::vector<MCTriangle> MarchingCubesRenderer::render(const OctreeNode* node, const VoxelGrid& grid, int x0, int y0, int z0, int size) {
    std::vector<MCTriangle> out;
    if (!node) return out;
    if (node->isLeaf) return localMC(grid, x0, y0, z0, size);
    const int half = size / 2;
    for (int i = 0; i < 8; i++) {
        const std::vector<MCTriangle> sub = render(node->children[i], grid, x0 + ((i & 1) ? half : 0), y0 + ((i & 2) ? half : 0),
                                                   z0 + ((i & 4) ? half : 0), half);
        out.insert(out.end(), sub.begin(), sub.end());
    }
    return out;
}
