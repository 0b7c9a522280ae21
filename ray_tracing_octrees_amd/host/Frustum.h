// Frustum.h -- six-plane view frustum with the reference's interface (453-skeleton/Frustum.h:6-24).
#pragma once

#include <array>

#include "rtmath.h"

class Frustum {
public:
    enum Planes { LEFT = 0, RIGHT, TOP, BOTTOM, NEAR, FAR, COUNT };

    explicit Frustum(const rtmath::mat4& viewProj);
    // 1 = inside, 0 = straddles a plane, -1 = outside; the box is first grown by extraMargin on every side
    int testAABB(const rtmath::vec3& min, const rtmath::vec3& max, float extraMargin) const;

    const float* planes() const { return &m_planes[0][0]; }   // 6 x (a,b,c,d), what the GPU cull kernel takes

private:
    std::array<std::array<float, 4>, COUNT> m_planes;
};
