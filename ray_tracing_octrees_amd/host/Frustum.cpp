// Frustum.cpp -- behaviour of 453-skeleton/Frustum.cpp:5-93 (Gribb-Hartmann extraction, p/n-vertex test).
#include "Frustum.h"

using rtmath::vec3;

Frustum::Frustum(const rtmath::mat4& viewProj) {
    float flat[24];
    rtmath::frustum_planes(viewProj, flat);
    for (int i = 0; i < COUNT; i++)
        for (int k = 0; k < 4; k++) m_planes[i][k] = flat[i * 4 + k];
}

int Frustum::testAABB(const vec3& min, const vec3& max, float extraMargin) const {
    const vec3 lo = min - vec3(extraMargin), hi = max + vec3(extraMargin);
    int verdict = 1;
    for (int i = 0; i < COUNT; i++) {
        const std::array<float, 4>& pl = m_planes[i];
        const vec3 n(pl[0], pl[1], pl[2]);
        const vec3 farCorner(pl[0] > 0 ? hi.x : lo.x, pl[1] > 0 ? hi.y : lo.y, pl[2] > 0 ? hi.z : lo.z);
        if (rtmath::dot(n, farCorner) + pl[3] < 0) return -1;
        const vec3 nearCorner(pl[0] < 0 ? hi.x : lo.x, pl[1] < 0 ? hi.y : lo.y, pl[2] < 0 ? hi.z : lo.z);
        if (rtmath::dot(n, nearCorner) + pl[3] < 0) verdict = 0;
    }
    return verdict;
}
