// rtmath.h -- the handful of glm 0.9.9.7 operations the ray path needs, in glm's exact
// operation order (column-major mat4, RH, [-1,1] clip), so that host-computed uniforms
// (inverse(view), tan(fov/2), frustum planes) are bit-identical to what the reference's
// host code would feed its shader.  Header-only, no dependencies.
//
// Follows (thirdparty/glm-0.9.9.7/glm/...): detail/func_matrix.inl:294-352 (inverse),
// detail/type_mat4x4.inl:630-648 (mat*mat), ext/matrix_clip_space.inl:249-262 (perspectiveRH_NO),
// ext/matrix_transform.inl:99-119 (lookAtRH), detail/func_geometric.inl (dot/cross/normalize),
// detail/func_trigonometric.inl:9-14 (radians).
#pragma once

#include <cmath>
#include <cstring>

namespace rtmath {

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(const vec3& a, const vec3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator-(const vec3& a) { return { -a.x, -a.y, -a.z }; }
inline vec3 operator*(const vec3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, const vec3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, const vec3& b) { a = a - b; return a; }

inline float dot(const vec3& a, const vec3& b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline vec3 cross(const vec3& x, const vec3& y) {
    return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
inline vec3 normalize(const vec3& v) { return v * inversesqrt(dot(v, v)); }
inline float length(const vec3& v) { return std::sqrt(dot(v, v)); }
inline float radians(float deg) { return deg * static_cast<float>(0.01745329251994329576923690768489); }

// column-major 4x4: m.c[col][row], &m.c[0][0] is the float[16] the C ABI takes
struct mat4 {
    float c[4][4];
    mat4() { std::memset(c, 0, sizeof c); }
    explicit mat4(float d) { std::memset(c, 0, sizeof c); c[0][0] = c[1][1] = c[2][2] = c[3][3] = d; }
    float* operator[](int col) { return c[col]; }
    const float* operator[](int col) const { return c[col]; }
    const float* data() const { return &c[0][0]; }
    float* data() { return &c[0][0]; }
    static mat4 from(const float* p) { mat4 m; std::memcpy(m.c, p, 64); return m; }
};

inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r;
    for (int col = 0; col < 4; col++)
        for (int k = 0; k < 4; k++)
            r[col][k] = a[0][k] * b[col][0] + a[1][k] * b[col][1] + a[2][k] * b[col][2] + a[3][k] * b[col][3];
    return r;
}

inline mat4 inverse(const mat4& m) {
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    const float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    const float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    const float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    const float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    const float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    const float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    const float Vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] };
    const float Vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    const float Vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] };
    const float Vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    const float SignA[4] = { +1.f, -1.f, +1.f, -1.f };
    const float SignB[4] = { -1.f, +1.f, -1.f, +1.f };

    mat4 Inv;
    for (int i = 0; i < 4; i++) {
        Inv[0][i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * SignA[i];
        Inv[1][i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * SignB[i];
        Inv[2][i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * SignA[i];
        Inv[3][i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * SignB[i];
    }
    float d0 = m[0][0] * Inv[0][0], d1 = m[0][1] * Inv[1][0], d2 = m[0][2] * Inv[2][0], d3 = m[0][3] * Inv[3][0];
    float OneOverDeterminant = 1.0f / ((d0 + d1) + (d2 + d3));
    mat4 r;
    for (int col = 0; col < 4; col++)
        for (int k = 0; k < 4; k++) r[col][k] = Inv[col][k] * OneOverDeterminant;
    return r;
}

inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
    float tanHalfFovy = std::tan(fovy / 2.0f);
    mat4 r;
    r[0][0] = 1.0f / (aspect * tanHalfFovy);
    r[1][1] = 1.0f / (tanHalfFovy);
    r[2][2] = -(zFar + zNear) / (zFar - zNear);
    r[2][3] = -1.0f;
    r[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}

inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r(1.0f);
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

// Gribb-Hartmann planes in the reference's order LEFT, RIGHT, TOP, BOTTOM, NEAR, FAR, normalised
// by the xyz length (453-skeleton/Frustum.cpp:5-48).  out: 6 x (a,b,c,d).
inline void frustum_planes(const mat4& vp, float out[24]) {
    enum { LEFT = 0, RIGHT, TOP, BOTTOM, NEAR_, FAR_ };
    for (int k = 0; k < 4; k++) {
        out[LEFT * 4 + k] = vp[k][3] + vp[k][0];
        out[RIGHT * 4 + k] = vp[k][3] - vp[k][0];
        out[BOTTOM * 4 + k] = vp[k][3] + vp[k][1];
        out[TOP * 4 + k] = vp[k][3] - vp[k][1];
        out[NEAR_ * 4 + k] = vp[k][3] + vp[k][2];
        out[FAR_ * 4 + k] = vp[k][3] - vp[k][2];
    }
    for (int i = 0; i < 6; i++) {
        float* p = out + i * 4;
        float len = length(vec3(p[0], p[1], p[2]));
        p[0] = p[0] / len; p[1] = p[1] / len; p[2] = p[2] / len; p[3] = p[3] / len;
    }
}

}  // namespace rtmath
