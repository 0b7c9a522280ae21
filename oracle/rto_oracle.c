/*
 * rto_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rto_oracle.h for the parity status (host pieces pinned against the
 * compiled reference; traversal/shading = restatement of GLSL, unpinned).
 *
 * S/ = /root/reference/453-skeleton/.  glm = thirdparty/glm-0.9.9.7/glm.
 * Compile with -O2 -ffp-contract=off (no fast-math): every float expression
 * below is a single IEEE binary32 operation per source operator.
 */
#include "rto_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* glm-flavoured scalar helpers                                         */
/* ------------------------------------------------------------------ */

/* glm/detail/func_common.inl: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x.
 * GLSL leaves min/max with NaN undefined; this is the convention the whole
 * project mirrors. */
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gmax(float x, float y) { return (x < y) ? y : x; }

/* glm/detail/func_exponential.inl:136-139 */
static inline float inversesqrt_(float x) { return 1.0f / sqrtf(x); }

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_muls(v3 a, float s) { return v3_(a.x * s, a.y * s, a.z * s); }
/* glm/detail/func_geometric.inl:48-55: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float v3_dot(v3 a, v3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
/* func_geometric.inl:68-79 */
static inline v3 v3_cross(v3 x, v3 y) {
    return v3_(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:82-90: v * inversesqrt(dot(v,v)) */
static inline v3 v3_normalize(v3 v) { return v3_muls(v, inversesqrt_(v3_dot(v, v))); }

float orc_radians(float deg) {
    /* glm/detail/func_trigonometric.inl:9-14 */
    return deg * (float)0.01745329251994329576923690768489;
}

/* column-major: m[c*4 + r] == glm m[c][r] */
#define M(m, c, r) ((m)[(c) * 4 + (r)])

void orc_mat4_inverse(const float m[16], float out[16]) {
    /* glm/detail/func_matrix.inl:294-352 (compute_inverse<4,4>) */
    float Coef00 = M(m,2,2) * M(m,3,3) - M(m,3,2) * M(m,2,3);
    float Coef02 = M(m,1,2) * M(m,3,3) - M(m,3,2) * M(m,1,3);
    float Coef03 = M(m,1,2) * M(m,2,3) - M(m,2,2) * M(m,1,3);

    float Coef04 = M(m,2,1) * M(m,3,3) - M(m,3,1) * M(m,2,3);
    float Coef06 = M(m,1,1) * M(m,3,3) - M(m,3,1) * M(m,1,3);
    float Coef07 = M(m,1,1) * M(m,2,3) - M(m,2,1) * M(m,1,3);

    float Coef08 = M(m,2,1) * M(m,3,2) - M(m,3,1) * M(m,2,2);
    float Coef10 = M(m,1,1) * M(m,3,2) - M(m,3,1) * M(m,1,2);
    float Coef11 = M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2);

    float Coef12 = M(m,2,0) * M(m,3,3) - M(m,3,0) * M(m,2,3);
    float Coef14 = M(m,1,0) * M(m,3,3) - M(m,3,0) * M(m,1,3);
    float Coef15 = M(m,1,0) * M(m,2,3) - M(m,2,0) * M(m,1,3);

    float Coef16 = M(m,2,0) * M(m,3,2) - M(m,3,0) * M(m,2,2);
    float Coef18 = M(m,1,0) * M(m,3,2) - M(m,3,0) * M(m,1,2);
    float Coef19 = M(m,1,0) * M(m,2,2) - M(m,2,0) * M(m,1,2);

    float Coef20 = M(m,2,0) * M(m,3,1) - M(m,3,0) * M(m,2,1);
    float Coef22 = M(m,1,0) * M(m,3,1) - M(m,3,0) * M(m,1,1);
    float Coef23 = M(m,1,0) * M(m,2,1) - M(m,2,0) * M(m,1,1);

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };

    float Vec0[4] = { M(m,1,0), M(m,0,0), M(m,0,0), M(m,0,0) };
    float Vec1[4] = { M(m,1,1), M(m,0,1), M(m,0,1), M(m,0,1) };
    float Vec2[4] = { M(m,1,2), M(m,0,2), M(m,0,2), M(m,0,2) };
    float Vec3[4] = { M(m,1,3), M(m,0,3), M(m,0,3), M(m,0,3) };

    static const float SignA[4] = { +1.f, -1.f, +1.f, -1.f };
    static const float SignB[4] = { -1.f, +1.f, -1.f, +1.f };
    float Inv[16];
    for (int i = 0; i < 4; i++) {
        float Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        M(Inv,0,i) = Inv0 * SignA[i];
        M(Inv,1,i) = Inv1 * SignB[i];
        M(Inv,2,i) = Inv2 * SignA[i];
        M(Inv,3,i) = Inv3 * SignB[i];
    }
    /* Row0 = (Inv[0][0], Inv[1][0], Inv[2][0], Inv[3][0]); Dot0 = m[0] * Row0 */
    float d0 = M(m,0,0) * M(Inv,0,0);
    float d1 = M(m,0,1) * M(Inv,1,0);
    float d2 = M(m,0,2) * M(Inv,2,0);
    float d3 = M(m,0,3) * M(Inv,3,0);
    float Dot1 = (d0 + d1) + (d2 + d3);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; i++) out[i] = Inv[i] * OneOverDeterminant;
}

void orc_mat4_mul(const float a[16], const float b[16], float out[16]) {
    /* glm/detail/type_mat4x4.inl:630-648: left-to-right sums per column */
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 4; k++)
            r[c * 4 + k] = M(a,0,k) * M(b,c,0) + M(a,1,k) * M(b,c,1) + M(a,2,k) * M(b,c,2) + M(a,3,k) * M(b,c,3);
    memcpy(out, r, sizeof r);
}

void orc_perspective(float fovy, float aspect, float zNear, float zFar, float out[16]) {
    /* glm/ext/matrix_clip_space.inl:249-262 perspectiveRH_NO (default clip control) */
    float tanHalfFovy = tanf(fovy / 2.0f);
    memset(out, 0, 16 * sizeof(float));
    M(out,0,0) = 1.0f / (aspect * tanHalfFovy);
    M(out,1,1) = 1.0f / (tanHalfFovy);
    M(out,2,2) = -(zFar + zNear) / (zFar - zNear);
    M(out,2,3) = -1.0f;
    M(out,3,2) = -(2.0f * zFar * zNear) / (zFar - zNear);
}

void orc_look_at(const float eye_[3], const float center_[3], const float up_[3], float out[16]) {
    /* glm/ext/matrix_transform.inl:99-119 lookAtRH */
    v3 eye = v3_(eye_[0], eye_[1], eye_[2]);
    v3 center = v3_(center_[0], center_[1], center_[2]);
    v3 up = v3_(up_[0], up_[1], up_[2]);
    v3 f = v3_normalize(v3_sub(center, eye));
    v3 s = v3_normalize(v3_cross(f, up));
    v3 u = v3_cross(s, f);
    memset(out, 0, 16 * sizeof(float));
    M(out,0,0) = 1.f; M(out,1,1) = 1.f; M(out,2,2) = 1.f; M(out,3,3) = 1.f;
    M(out,0,0) = s.x; M(out,1,0) = s.y; M(out,2,0) = s.z;
    M(out,0,1) = u.x; M(out,1,1) = u.y; M(out,2,1) = u.z;
    M(out,0,2) = -f.x; M(out,1,2) = -f.y; M(out,2,2) = -f.z;
    M(out,3,0) = -v3_dot(s, eye);
    M(out,3,1) = -v3_dot(u, eye);
    M(out,3,2) = v3_dot(f, eye);
}

/* ------------------------------------------------------------------ */
/* Camera (S/Camera.cpp)                                                */
/* ------------------------------------------------------------------ */
void orc_camera_init(orc_camera* c, float theta, float phi, float radius) {
    /* S/Camera.cpp:8-9 */
    c->theta = theta; c->phi = phi; c->radius = radius;
    c->target[0] = c->target[1] = c->target[2] = 0.0f;
}

static v3 camera_eye(const orc_camera* c) {
    /* S/Camera.cpp:21-29: radius * vec3(cos(t)*sin(p), sin(t), cos(t)*cos(p)) + target */
    v3 d = v3_(cosf(c->theta) * sinf(c->phi), sinf(c->theta), cosf(c->theta) * cosf(c->phi));
    v3 e = v3_(c->radius * d.x, c->radius * d.y, c->radius * d.z);
    return v3_add(e, v3_(c->target[0], c->target[1], c->target[2]));
}

void orc_camera_pos(const orc_camera* c, float out[3]) {
    v3 e = camera_eye(c);
    out[0] = e.x; out[1] = e.y; out[2] = e.z;
}

void orc_camera_view(const orc_camera* c, float out[16]) {
    /* S/Camera.cpp:11-19 */
    v3 e = camera_eye(c);
    float eye[3] = { e.x, e.y, e.z };
    float up[3] = { 0.0f, 1.0f, 0.0f };
    orc_look_at(eye, c->target, up, out);
}

void orc_camera_pan(orc_camera* c, float dx, float dy) {
    /* S/Camera.cpp:48-52 (getLookDir) and :76-82 (pan) */
    v3 tgt = v3_(c->target[0], c->target[1], c->target[2]);
    v3 look = v3_normalize(v3_sub(tgt, camera_eye(c)));
    v3 right = v3_normalize(v3_cross(look, v3_(0.f, 1.f, 0.f)));
    v3 up = v3_normalize(v3_cross(right, look));
    /* (-dx * right + dy * up) * (radius * 0.001f) */
    v3 a = v3_((-dx) * right.x, (-dx) * right.y, (-dx) * right.z);
    v3 b = v3_(dy * up.x, dy * up.y, dy * up.z);
    v3 s = v3_add(a, b);
    float k = c->radius * 0.001f;
    c->target[0] += s.x * k; c->target[1] += s.y * k; c->target[2] += s.z * k;
}

/* ------------------------------------------------------------------ */
/* Frustum (S/Frustum.cpp)                                              */
/* ------------------------------------------------------------------ */
void orc_frustum_planes(const float vp[16], float planes[24]) {
    /* S/Frustum.cpp:5-48. Plane order: LEFT, RIGHT, TOP, BOTTOM, NEAR, FAR (S/Frustum.h:8-16) */
    enum { LEFT = 0, RIGHT, TOP, BOTTOM, NEAR_, FAR_ };
    for (int k = 0; k < 4; k++) {
        planes[LEFT * 4 + k]   = M(vp,k,3) + M(vp,k,0);
        planes[RIGHT * 4 + k]  = M(vp,k,3) - M(vp,k,0);
        planes[BOTTOM * 4 + k] = M(vp,k,3) + M(vp,k,1);
        planes[TOP * 4 + k]    = M(vp,k,3) - M(vp,k,1);
        planes[NEAR_ * 4 + k]  = M(vp,k,3) + M(vp,k,2);
        planes[FAR_ * 4 + k]   = M(vp,k,3) - M(vp,k,2);
    }
    for (int i = 0; i < 6; i++) {
        float* p = planes + i * 4;
        float len = sqrtf(v3_dot(v3_(p[0], p[1], p[2]), v3_(p[0], p[1], p[2])));
        p[0] = p[0] / len; p[1] = p[1] / len; p[2] = p[2] / len; p[3] = p[3] / len;
    }
}

int orc_frustum_test_aabb(const float planes[24], const float bmin[3], const float bmax[3], float margin) {
    /* S/Frustum.cpp:52-93 */
    v3 emin = v3_(bmin[0] - margin, bmin[1] - margin, bmin[2] - margin);
    v3 emax = v3_(bmax[0] + margin, bmax[1] + margin, bmax[2] + margin);
    int result = 1;
    for (int i = 0; i < 6; i++) {
        const float* pl = planes + i * 4;
        v3 nrm = v3_(pl[0], pl[1], pl[2]);
        v3 p = v3_(pl[0] > 0 ? emax.x : emin.x, pl[1] > 0 ? emax.y : emin.y, pl[2] > 0 ? emax.z : emin.z);
        if (v3_dot(nrm, p) + pl[3] < 0) return -1;
        v3 n = v3_(pl[0] < 0 ? emax.x : emin.x, pl[1] < 0 ? emax.y : emin.y, pl[2] < 0 ? emax.z : emin.z);
        if (v3_dot(nrm, n) + pl[3] < 0) result = 0;
    }
    return result;
}

/* ------------------------------------------------------------------ */
/* Scene (S/main.cpp)                                                   */
/* ------------------------------------------------------------------ */
static inline float fmin3f(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }

void orc_generate_test_sphere(int dimX, int dimY, int dimZ, uint8_t* out) {
    /* S/main.cpp:337-372 (density +1/-1) folded with :1060-1067 (FILLED iff density > 0) */
    float cx = 0.5f * (dimX - 1);
    float cy = 0.5f * (dimY - 1);
    float cz = 0.5f * (dimZ - 1);
    float rOuter = 0.4f * fmin3f((float)dimX, (float)dimY, (float)dimZ);
    float rInner = 0.2f * fmin3f((float)dimX, (float)dimY, (float)dimZ);
    for (int z = 0; z < dimZ; z++)
        for (int y = 0; y < dimY; y++)
            for (int x = 0; x < dimX; x++) {
                float dx = x - cx, dy = y - cy, dz = z - cz;
                float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                size_t idx = (size_t)x + (size_t)y * dimX + (size_t)z * ((size_t)dimX * dimY);
                out[idx] = (dist < rInner || dist > rOuter) ? 0 : 1;
            }
}

int orc_recenter_filled_voxels(orc_grid* g) {
    /* S/main.cpp:376-422 */
    float mnx = 3.402823466e+38f, mny = mnx, mnz = mnx;
    float mxx = -3.402823466e+38f, mxy = mxx, mxz = mxx;
    for (int z = 0; z < g->dimZ; ++z)
        for (int y = 0; y < g->dimY; ++y)
            for (int x = 0; x < g->dimX; ++x) {
                size_t idx = (size_t)x + (size_t)y * g->dimX + (size_t)z * ((size_t)g->dimX * g->dimY);
                if (g->data[idx] == 1) {
                    float cx = g->minX + (x + 0.5f) * g->voxelSize;
                    float cy = g->minY + (y + 0.5f) * g->voxelSize;
                    float cz = g->minZ + (z + 0.5f) * g->voxelSize;
                    if (cx < mnx) mnx = cx;
                    if (cy < mny) mny = cy;
                    if (cz < mnz) mnz = cz;
                    if (cx > mxx) mxx = cx;
                    if (cy > mxy) mxy = cy;
                    if (cz > mxz) mxz = cz;
                }
            }
    if (mnx > mxx) return 0;
    float centerX = 0.5f * (mnx + mxx);
    float centerY = 0.5f * (mny + mxy);
    float centerZ = 0.5f * (mnz + mxz);
    g->minX -= centerX; g->minY -= centerY; g->minZ -= centerZ;
    return 1;
}

void orc_make_test_sphere_grid(int dim, orc_grid* g) {
    /* S/main.cpp:1052-1070 then :1074 */
    g->dimX = g->dimY = g->dimZ = dim;
    g->minX = g->minY = g->minZ = -0.5f;
    g->voxelSize = 1.f / dim;
    g->data = (uint8_t*)malloc((size_t)dim * dim * dim);
    orc_generate_test_sphere(dim, dim, dim, g->data);
    orc_recenter_filled_voxels(g);
}

/* ------------------------------------------------------------------ */
/* sceneCache.bin (S/CacheUtils.cpp:5-59)                               */
/* ------------------------------------------------------------------ */
int orc_load_voxel_grid(const char* path, orc_grid* g) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    uint64_t n = 0;
    int ok = fread(&g->dimX, 4, 1, f) == 1 && fread(&g->dimY, 4, 1, f) == 1 && fread(&g->dimZ, 4, 1, f) == 1 &&
             fread(&g->minX, 4, 1, f) == 1 && fread(&g->minY, 4, 1, f) == 1 && fread(&g->minZ, 4, 1, f) == 1 &&
             fread(&g->voxelSize, 4, 1, f) == 1 && fread(&n, 8, 1, f) == 1;
    if (!ok) { fclose(f); return 0; }
    g->data = (uint8_t*)malloc(n ? n : 1);
    ok = fread(g->data, 1, n, f) == n;
    fclose(f);
    if (!ok) { free(g->data); g->data = NULL; return 0; }
    return 1;
}

int orc_save_voxel_grid(const char* path, const orc_grid* g) {
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    uint64_t n = (uint64_t)g->dimX * g->dimY * g->dimZ;
    fwrite(&g->dimX, 4, 1, f); fwrite(&g->dimY, 4, 1, f); fwrite(&g->dimZ, 4, 1, f);
    fwrite(&g->minX, 4, 1, f); fwrite(&g->minY, 4, 1, f); fwrite(&g->minZ, 4, 1, f);
    fwrite(&g->voxelSize, 4, 1, f); fwrite(&n, 8, 1, f);
    fwrite(g->data, 1, n, f);
    fclose(f);
    return 1;
}

/* ------------------------------------------------------------------ */
/* Octree build (S/OctreeVoxel.cpp:692-778) + flatten (S/RayTracerBVH.cpp:430-490) */
/* ------------------------------------------------------------------ */
typedef struct tnode {
    int32_t x, y, z, size;
    uint8_t isLeaf, isSolid, isUniform;
    struct tnode* children[8];
} tnode;

typedef struct { tnode** blocks; size_t nblocks, cap, used; size_t total; } arena;
#define ARENA_BLOCK 65536

static tnode* arena_new(arena* a) {
    if (a->nblocks == 0 || a->used == ARENA_BLOCK) {
        if (a->nblocks == a->cap) {
            a->cap = a->cap ? a->cap * 2 : 16;
            a->blocks = (tnode**)realloc(a->blocks, a->cap * sizeof(tnode*));
        }
        a->blocks[a->nblocks++] = (tnode*)malloc(ARENA_BLOCK * sizeof(tnode));
        a->used = 0;
    }
    a->total++;
    return &a->blocks[a->nblocks - 1][a->used++];
}

static void arena_free(arena* a) {
    for (size_t i = 0; i < a->nblocks; i++) free(a->blocks[i]);
    free(a->blocks);
}

static inline uint8_t voxel_safe(const orc_grid* g, int x, int y, int z) {
    /* S/OctreeVoxel.cpp:692-701 getVoxelSafe: out of range == EMPTY */
    if (x < 0 || y < 0 || z < 0 || x >= g->dimX || y >= g->dimY || z >= g->dimZ) return 0;
    return g->data[(size_t)x + (size_t)y * g->dimX + (size_t)z * ((size_t)g->dimX * g->dimY)];
}

static tnode* build_rec(const orc_grid* g, int x0, int y0, int z0, int size, arena* a) {
    /* S/OctreeVoxel.cpp:704-762 buildOctreeRec (the g_octreeMap side table is not on the ray path) */
    tnode* node = arena_new(a);
    node->x = x0; node->y = y0; node->z = z0; node->size = size;
    node->isLeaf = node->isSolid = node->isUniform = 0;
    for (int i = 0; i < 8; i++) node->children[i] = NULL;

    if (size == 1) {
        node->isLeaf = 1;
        node->isSolid = (voxel_safe(g, x0, y0, z0) == 1);
        node->isUniform = 1;
        return node;
    }
    int allSame = 1;
    uint8_t firstVal = voxel_safe(g, x0, y0, z0);
    for (int zz = z0; zz < z0 + size && allSame; zz++)
        for (int yy = y0; yy < y0 + size && allSame; yy++)
            for (int xx = x0; xx < x0 + size; xx++)
                if (voxel_safe(g, xx, yy, zz) != firstVal) { allSame = 0; break; }
    if (allSame) {
        node->isLeaf = 1; node->isUniform = 1; node->isSolid = (firstVal == 1);
        return node;
    }
    int half = size / 2;
    for (int i = 0; i < 8; i++) {
        int ox = x0 + ((i & 1) ? half : 0);
        int oy = y0 + ((i & 2) ? half : 0);
        int oz = z0 + ((i & 4) ? half : 0);
        node->children[i] = build_rec(g, ox, oy, oz, half, a);
    }
    return node;
}

int64_t orc_build_flat_octree(const orc_grid* g, orc_node** out) {
    *out = NULL;
    /* S/OctreeVoxel.cpp:765-778 createOctreeFromVoxelGrid */
    if (g->dimX == 0 || g->dimY == 0 || g->dimZ == 0) return 0;
    int maxDim = g->dimX > g->dimY ? g->dimX : g->dimY;
    if (g->dimZ > maxDim) maxDim = g->dimZ;
    int sizePow2 = 1;
    while (sizePow2 < maxDim) sizePow2 <<= 1;
    arena a; memset(&a, 0, sizeof a);
    tnode* root = build_rec(g, 0, 0, 0, sizePow2, &a);

    /* S/RayTracerBVH.cpp:443-490 setOctree: BFS; a child gets the next free index when its parent is dequeued. */
    size_t n = a.total;
    orc_node* flat = (orc_node*)malloc(n * sizeof(orc_node));
    tnode** queue = (tnode**)malloc(n * sizeof(tnode*));
    size_t qh = 0, qt = 0, count = 1;
    queue[qt++] = root;          /* node at queue position i has flat index i */
    while (qh < qt) {
        size_t idx = qh;
        tnode* nd = queue[qh++];
        orc_node* o = &flat[idx];
        o->x = nd->x; o->y = nd->y; o->z = nd->z; o->size = nd->size;
        o->isLeaf = nd->isLeaf ? 1 : 0; o->isSolid = nd->isSolid ? 1 : 0; o->isUniform = nd->isUniform ? 1 : 0;
        for (int i = 0; i < 8; i++) o->child[i] = -1;
        if (!nd->isLeaf) {
            for (int i = 0; i < 8; i++) {
                tnode* c = nd->children[i];
                if (c) { o->child[i] = (int32_t)count; queue[qt++] = c; count++; }
            }
        }
    }
    free(queue);
    arena_free(&a);
    *out = flat;
    return (int64_t)n;
}

void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------ */
/* Frustum-culling compaction (S/RayTracerBVH.cpp:731-802)              */
/* ------------------------------------------------------------------ */
/* the loop + compaction for given planes and margin (the reference derives them at :731-734, :755) */
int64_t orc_cull_compact_planes(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                                const float planes[24], float margin, orc_node* out, uint8_t* visible) {
    uint8_t* vis = visible ? visible : (uint8_t*)malloc((size_t)n);
    int32_t* remap = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++) {
        const orc_node* nd = &nodes[i];
        /* :747-752 */
        float mn[3] = { gridMin[0] + nd->x * voxelSize, gridMin[1] + nd->y * voxelSize, gridMin[2] + nd->z * voxelSize };
        float ext = nd->size * voxelSize;
        float mx[3] = { mn[0] + ext, mn[1] + ext, mn[2] + ext };
        vis[i] = orc_frustum_test_aabb(planes, mn, mx, margin) != -1;     /* :755-761 */
    }
    for (int64_t i = 0; i < n; i++) remap[i] = vis[i] ? (int32_t)count++ : -1;   /* :765-772 */
    for (int64_t i = 0; i < n; i++) {                                              /* :778-802 */
        if (!vis[i]) continue;
        orc_node* o = &out[remap[i]];
        *o = nodes[i];
        if (!nodes[i].isLeaf) {
            for (int c = 0; c < 8; c++) {
                int32_t oc = nodes[i].child[c];
                o->child[c] = (oc >= 0 && oc < n && vis[oc]) ? remap[oc] : -1;
            }
        }
    }
    free(remap);
    if (!visible) free(vis);
    return count;
}

int64_t orc_cull_compact(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                         const float view[16], float fovDeg, float aspect,
                         orc_node* out, uint8_t* visible) {
    float proj[16], vp[16], planes[24];
    orc_perspective(orc_radians(fovDeg), aspect, 0.01f, 5000.f, proj);   /* :733 */
    orc_mat4_mul(proj, view, vp);                                          /* :734 */
    orc_frustum_planes(vp, planes);
    return orc_cull_compact_planes(nodes, n, gridMin, voxelSize, planes, 150.0f, out, visible);   /* margin: :755 */
}

/* ------------------------------------------------------------------ */
/* The kernel (S/RayTracerBVH.cpp:182-369)                              */
/* ------------------------------------------------------------------ */
#define MAX_TRAVERSAL_STEPS 512   /* :192 */

typedef struct {
    float gridMin[3], voxelSize;
    float invView[16];
    float camPos[3];
    float aspect, tanHalfFov;
    int W, H;
} frame_consts;

static void frame_setup(frame_consts* fc, const float gridMin[3], float voxelSize, const float view[16],
                        const float camPos[3], float aspect, float fovDeg, int W, int H) {
    memcpy(fc->gridMin, gridMin, 12); fc->voxelSize = voxelSize;
    orc_mat4_inverse(view, fc->invView);             /* :348, pixel-independent */
    memcpy(fc->camPos, camPos, 12);
    fc->aspect = aspect;
    float fovRad = orc_radians(fovDeg);              /* :340 */
    fc->tanHalfFov = tanf(fovRad * 0.5f);            /* :344, pixel-independent */
    fc->W = W; fc->H = H;
}

static inline void generate_ray(const frame_consts* fc, int px, int py, v3* dir) {
    /* :338-355 */
    float nx = ((float)px + 0.5f) / (float)fc->W * 2.0f - 1.0f;
    float ny = 1.0f - ((float)py + 0.5f) / (float)fc->H * 2.0f;
    nx *= fc->aspect;
    nx *= fc->tanHalfFov;
    ny *= fc->tanHalfFov;
    /* normalize(vec4(nx, ny, -1, 0)); vec4 dot = (x*x + y*y) + (z*z + w*w) (func_geometric.inl:58-65) */
    float d4 = (nx * nx + ny * ny) + ((-1.0f) * (-1.0f) + 0.0f * 0.0f);
    float inv4 = inversesqrt_(d4);
    float vx = nx * inv4, vy = ny * inv4, vz = (-1.0f) * inv4, vw = 0.0f * inv4;
    /* invView * rayDirView (type_mat4x4.inl:561-571): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
    const float* m = fc->invView;
    v3 w;
    w.x = (M(m,0,0) * vx + M(m,1,0) * vy) + (M(m,2,0) * vz + M(m,3,0) * vw);
    w.y = (M(m,0,1) * vx + M(m,1,1) * vy) + (M(m,2,1) * vz + M(m,3,1) * vw);
    w.z = (M(m,0,2) * vx + M(m,1,2) * vy) + (M(m,2,2) * vz + M(m,3,2) * vw);
    *dir = v3_normalize(w);
}

static inline int intersect_aabb(v3 ro, v3 rd, v3 bmin, v3 bmax, float* tNear, float* tFar) {
    /* :226-236 */
    v3 invDir = v3_(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    v3 t1 = v3_((bmin.x - ro.x) * invDir.x, (bmin.y - ro.y) * invDir.y, (bmin.z - ro.z) * invDir.z);
    v3 t2 = v3_((bmax.x - ro.x) * invDir.x, (bmax.y - ro.y) * invDir.y, (bmax.z - ro.z) * invDir.z);
    v3 tMin = v3_(gmin(t1.x, t2.x), gmin(t1.y, t2.y), gmin(t1.z, t2.z));
    v3 tMax = v3_(gmax(t1.x, t2.x), gmax(t1.y, t2.y), gmax(t1.z, t2.z));
    *tNear = gmax(gmax(tMin.x, tMin.y), tMin.z);
    *tFar = gmin(gmin(tMax.x, tMax.y), tMax.z);
    return (*tNear <= *tFar && *tFar > 0.0f);
}

typedef struct { int hit; int steps; int max_sp; int internal; v3 normal; } trace_result;

static inline trace_result trace(const orc_node* nodes, const frame_consts* fc, v3 ro, v3 rd) {
    /* :239-327 intersectOctreeIterative */
    trace_result r; r.hit = 0; r.steps = 0; r.max_sp = 1; r.internal = 0; r.normal = v3_(0, 0, 0);
    float closestT = 1e30f;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    int traversalSteps = 0;
    v3 gmn = v3_(fc->gridMin[0], fc->gridMin[1], fc->gridMin[2]);
    float vs = fc->voxelSize;

    while (sp > 0 && traversalSteps < MAX_TRAVERSAL_STEPS) {
        sp--;
        int nodeIdx = stack[sp];
        if (nodeIdx < 0) continue;
        traversalSteps++;
        const orc_node* node = &nodes[nodeIdx];
        /* :265-266 */
        v3 nodeMin = v3_(gmn.x + (float)node->x * vs, gmn.y + (float)node->y * vs, gmn.z + (float)node->z * vs);
        float ext = (float)node->size * vs;
        v3 nodeMax = v3_(nodeMin.x + ext, nodeMin.y + ext, nodeMin.z + ext);
        float tNear, tFar;
        if (!intersect_aabb(ro, rd, nodeMin, nodeMax, &tNear, &tFar)) continue;
        if (tNear >= closestT) continue;
        if (node->isUniform == 1 || node->isLeaf == 1) {
            /* :277-311 -- the isUniform and isLeaf branches have identical bodies */
            if (node->isSolid == 1) {
                float tHit = gmax(0.0f, tNear);
                if (tHit < closestT && tHit <= tFar) {
                    closestT = tHit;
                    r.hit = 1;
                    v3 center = v3_(0.5f * (nodeMin.x + nodeMax.x), 0.5f * (nodeMin.y + nodeMax.y), 0.5f * (nodeMin.z + nodeMax.z));
                    v3 p = v3_(ro.x + rd.x * tHit, ro.y + rd.y * tHit, ro.z + rd.z * tHit);
                    r.normal = v3_normalize(v3_sub(p, center));
                    break;
                }
            }
            continue;
        }
        r.internal++;
        for (int i = 0; i < 8; i++) {
            int childIdx = node->child[i];
            if (childIdx >= 0) stack[sp++] = childIdx;
        }
        if (sp > r.max_sp) r.max_sp = sp;
    }
    r.steps = traversalSteps;
    return r;
}

/* The reference's CLOSEST-hit traversal: the earlier shader kept block-commented in the same file (S/RayTracerBVH.cpp:63-138).
 * Same LIFO order and slab test as trace(); no early break on the first accepted leaf and no step cap: every node whose tNear lies
 * below the best hit so far is visited, the leaf with the smallest tHit wins (a later leaf replaces it only when strictly closer:
 * ties go to the leaf popped first).  Dead code upstream -- the only traversal rule of the reference this repo could not render
 * until round 4 (rto_render_closest_*).  steps: nodes popped (uncapped). */
static inline trace_result trace_closest(const orc_node* nodes, const frame_consts* fc, v3 ro, v3 rd) {
    trace_result r; r.hit = 0; r.steps = 0; r.max_sp = 1; r.internal = 0; r.normal = v3_(0, 0, 0);
    float closestT = 1e30f;                                   /* :66 */
    int stack[128];                                           /* :71 */
    int sp = 0;
    stack[sp++] = 0;
    v3 gmn = v3_(fc->gridMin[0], fc->gridMin[1], fc->gridMin[2]);
    float vs = fc->voxelSize;
    while (sp > 0) {                                          /* :75 */
        sp--;
        int nodeIdx = stack[sp];
        if (nodeIdx < 0) continue;
        r.steps++;
        const orc_node* node = &nodes[nodeIdx];
        v3 nodeMin = v3_(gmn.x + (float)node->x * vs, gmn.y + (float)node->y * vs, gmn.z + (float)node->z * vs);   /* :83-84 */
        float ext = (float)node->size * vs;
        v3 nodeMax = v3_(nodeMin.x + ext, nodeMin.y + ext, nodeMin.z + ext);
        float tNear, tFar;
        if (!intersect_aabb(ro, rd, nodeMin, nodeMax, &tNear, &tFar)) continue;   /* :87-88 */
        if (tNear >= closestT) continue;                      /* :91-92 */
        if (node->isUniform == 1 || node->isLeaf == 1) {      /* :94-122: identical bodies, no break */
            if (node->isSolid == 1) {
                float tHit = gmax(0.0f, tNear);
                if (tHit < closestT && tHit <= tFar) {
                    closestT = tHit;
                    r.hit = 1;
                    v3 center = v3_(0.5f * (nodeMin.x + nodeMax.x), 0.5f * (nodeMin.y + nodeMax.y), 0.5f * (nodeMin.z + nodeMax.z));
                    v3 p = v3_(ro.x + rd.x * tHit, ro.y + rd.y * tHit, ro.z + rd.z * tHit);
                    r.normal = v3_normalize(v3_sub(p, center));
                }
            }
            continue;
        }
        r.internal++;
        for (int i = 0; i < 8; i++) {                         /* :125-129 */
            int childIdx = node->child[i];
            if (childIdx >= 0) stack[sp++] = childIdx;
        }
        if (sp > r.max_sp) r.max_sp = sp;
    }
    return r;
}

static inline void shade_store(const trace_result* tr, float* px) {
    /* :331-336 shade, :366-367 store */
    if (tr->hit) {
        v3 l = v3_normalize(v3_(-1.0f, -1.0f, -1.0f));
        v3 nl = v3_(-l.x, -l.y, -l.z);
        float ndotl = gmax(0.0f, v3_dot(tr->normal, nl));
        px[0] = 1.0f * ndotl + 0.1f;
        px[1] = 0.8f * ndotl + 0.1f;
        px[2] = 0.6f * ndotl + 0.1f;
    } else {
        px[0] = px[1] = px[2] = 0.0f;
    }
    px[3] = 1.0f;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_render(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                const float view[16], const float camPos[3], float aspect, float fovDeg,
                int W, int H, int y0, int y1, float* out, orc_stats* stats, int nthreads) {
    (void)n;
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
    uint64_t pops = 0, hits = 0, capped = 0, internal = 0;
    unsigned max_stack = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+:pops,hits,capped,internal) reduction(max:max_stack)
#endif
    for (int py = y0; py < y1; py++) {
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            trace_result tr = trace(nodes, &fc, ro, rd);
            shade_store(&tr, out + ((size_t)py * W + px) * 4);
            pops += (uint64_t)tr.steps; hits += (uint64_t)tr.hit; internal += (uint64_t)tr.internal;
            capped += (uint64_t)(!tr.hit && tr.steps >= MAX_TRAVERSAL_STEPS);
            if ((unsigned)tr.max_sp > max_stack) max_stack = (unsigned)tr.max_sp;
        }
    }
    if (stats) {
        stats->rays = (uint64_t)W * (uint64_t)(y1 - y0);
        stats->pops = pops; stats->hits = hits; stats->capped = capped; stats->internal = internal;
        stats->max_stack = max_stack; stats->pad = 0;
    }
}

/* The closest-hit frame (trace_closest): same rays, same shade.  stats->pops counts the popped nodes (no cap: `capped` stays 0). */
void orc_render_closest(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                        const float view[16], const float camPos[3], float aspect, float fovDeg,
                        int W, int H, float* out, orc_stats* stats, int nthreads) {
    (void)n;
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
    uint64_t pops = 0, hits = 0, internal = 0;
    unsigned max_stack = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+:pops,hits,internal) reduction(max:max_stack)
#endif
    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            trace_result tr = trace_closest(nodes, &fc, ro, rd);
            shade_store(&tr, out + ((size_t)py * W + px) * 4);
            pops += (uint64_t)tr.steps; hits += (uint64_t)tr.hit; internal += (uint64_t)tr.internal;
            if ((unsigned)tr.max_sp > max_stack) max_stack = (unsigned)tr.max_sp;
        }
    }
    if (stats) {
        stats->rays = (uint64_t)W * (uint64_t)H;
        stats->pops = pops; stats->hits = hits; stats->capped = 0; stats->internal = internal;
        stats->max_stack = max_stack; stats->pad = 0;
    }
}

void orc_render_steps(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                      const float view[16], const float camPos[3], float aspect, float fovDeg,
                      int W, int H, int32_t* steps) {
    (void)n;
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            trace_result tr = trace(nodes, &fc, ro, rd);
            steps[(size_t)py * W + px] = tr.hit ? tr.steps : -tr.steps;
        }
}

/* per-pixel count of internal nodes whose children were pushed (analysis aid for the GPU work distribution) */
void orc_render_visits(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                       const float view[16], const float camPos[3], float aspect, float fovDeg,
                       int W, int H, int32_t* visits) {
    (void)n;
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            trace_result tr = trace(nodes, &fc, ro, rd);
            visits[(size_t)py * W + px] = tr.internal;
        }
}

/* ------------------------------------------------------------------ */
/* N1: octreeRaySkip (S/VolumeRaycastRenderer.cpp:50-155), on the flat array */
/* ------------------------------------------------------------------ */
static float ray_skip_rec(const orc_node* nodes, int32_t idx, const float gridMin[3], float vx,
                          v3 ro, v3 rd, float tMin, float tMax, const uint8_t* vis, int32_t* hitLeaf) {
    if (idx < 0) return 1e30f;                                   /* :60-62 */
    if (vis && !vis[idx]) return 1e30f;                          /* :64-67: a node the visibility map holds as false */
    const orc_node* node = &nodes[idx];
    /* :70-77 */
    float wx0 = gridMin[0] + node->x * vx;
    float wy0 = gridMin[1] + node->y * vx;
    float wz0 = gridMin[2] + node->z * vx;
    float wSize = node->size * vx;
    v3 bmin = v3_(wx0, wy0, wz0);
    v3 bmax = v3_(wx0 + wSize, wy0 + wSize, wz0 + wSize);
    /* :81-87 */
    v3 invRd = v3_(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    const float smallValue = 1e-10f;
    if (fabsf(rd.x) < smallValue) invRd.x = rd.x >= 0 ? 1e10f : -1e10f;
    if (fabsf(rd.y) < smallValue) invRd.y = rd.y >= 0 ? 1e10f : -1e10f;
    if (fabsf(rd.z) < smallValue) invRd.z = rd.z >= 0 ? 1e10f : -1e10f;
    /* :90-97 */
    v3 t1 = v3_((bmin.x - ro.x) * invRd.x, (bmin.y - ro.y) * invRd.y, (bmin.z - ro.z) * invRd.z);
    v3 t2 = v3_((bmax.x - ro.x) * invRd.x, (bmax.y - ro.y) * invRd.y, (bmax.z - ro.z) * invRd.z);
    v3 tN = v3_(gmin(t1.x, t2.x), gmin(t1.y, t2.y), gmin(t1.z, t2.z));
    v3 tF = v3_(gmax(t1.x, t2.x), gmax(t1.y, t2.y), gmax(t1.z, t2.z));
    /* std::max/std::min: max(a,b) = (a<b)?b:a ; min(a,b) = (b<a)?b:a -- same as gmax/gmin */
    float enterT = gmax(gmax(tN.x, tN.y), gmax(tN.z, tMin));
    float exitT = gmin(gmin(tF.x, tF.y), gmin(tF.z, tMax));
    if (enterT > exitT) return 1e30f;                             /* :100-102 */
    if (node->isLeaf) {                                           /* :105-110 */
        if (!node->isSolid) return 1e30f;
        if (hitLeaf) *hitLeaf = idx;                                 /* (not in the reference: which leaf the distance belongs to) */
        return enterT;
    }
    int dirMask = ((rd.x > 0) ? 1 : 0) | ((rd.y > 0) ? 2 : 0) | ((rd.z > 0) ? 4 : 0);   /* :114-116 */
    float bestT = 1e30f;
    for (int dist = 0; dist <= 3; dist++) {                        /* :122-152 */
        for (int octant = 0; octant < 8; octant++) {
            int diff = octant ^ dirMask, bitDiff = 0;
            while (diff) { bitDiff += diff & 1; diff >>= 1; }
            if (bitDiff != dist) continue;
            int32_t child = node->child[octant];
            if (child < 0) continue;
            float childT = ray_skip_rec(nodes, child, gridMin, vx, ro, rd, enterT, exitT, vis, hitLeaf);
            if (childT < bestT) {
                bestT = childT;
                if (childT < 1e30f) return childT;
            }
        }
    }
    return bestT;
}

float orc_octree_ray_skip(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                          const float ro[3], const float rd[3], float tMin, float tMax) {
    if (n <= 0) return 1e30f;
    return ray_skip_rec(nodes, 0, gridMin, voxelSize, v3_(ro[0], ro[1], ro[2]), v3_(rd[0], rd[1], rd[2]), tMin, tMax, NULL, NULL);
}

/* the same with the reference's visibility map (S/VolumeRaycastRenderer.cpp:64-67) as one flag per node of the flat array */
float orc_octree_ray_skip_vis(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                              const float ro[3], const float rd[3], float tMin, float tMax, const uint8_t* vis) {
    if (n <= 0) return 1e30f;
    return ray_skip_rec(nodes, 0, gridMin, voxelSize, v3_(ro[0], ro[1], ro[2]), v3_(rd[0], rd[1], rd[2]), tMin, tMax, vis, NULL);
}

/* The per-pixel ray directions of S/RayTracerBVH.cpp:338-355 (generateRay): W*H x 3 floats, row 0 = top. */
void orc_generate_rays(const float view[16], const float camPos[3], float aspect, float fovDeg, int W, int H, float* rd) {
    frame_consts fc;
    const float zero[3] = { 0, 0, 0 };
    frame_setup(&fc, zero, 1.0f, view, camPos, aspect, fovDeg, W, H);
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            v3 d;
            generate_ray(&fc, px, py, &d);
            float* o = rd + ((size_t)py * W + px) * 3;
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
}

/* N1 as a render mode ("octreeRaySkip as a second kernel mode = nearest-hit"): per pixel, the ray of
 * generateRay (S/RayTracerBVH.cpp:338-355) goes through octreeRaySkip(root, ro, rd, 0, 1e30, grid, visibility)
 * (S/VolumeRaycastRenderer.cpp:50-155).  outT: the distance it returns (1e30: nothing).  outRGBA (may be NULL): the
 * reference's shade (S/RayTracerBVH.cpp:283-285, 331-336) of the leaf that distance belongs to, with tHit = that distance;
 * (0,0,0,1) without a hit.  The distances are pinned by the reference's compiled function (tests/golden/ref_ray_skip.npz,
 * "pixels_*"); the colour is this project's combination of the two reference pieces. */
void orc_render_skip(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                     const float view[16], const float camPos[3], float aspect, float fovDeg, int W, int H,
                     const uint8_t* vis, float* outRGBA, float* outT, int nthreads) {
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            int32_t leaf = -1;
            float t = n > 0 ? ray_skip_rec(nodes, 0, gridMin, voxelSize, ro, rd, 0.0f, 1e30f, vis, &leaf) : 1e30f;
            const size_t pix = (size_t)py * W + px;
            if (outT) outT[pix] = t;
            if (!outRGBA) continue;
            trace_result tr; tr.hit = 0; tr.steps = 0; tr.max_sp = 0; tr.internal = 0; tr.normal = v3_(0, 0, 0);
            if (t < 1e30f && leaf >= 0) {
                const orc_node* nd = &nodes[leaf];
                v3 nodeMin = v3_(gridMin[0] + (float)nd->x * voxelSize, gridMin[1] + (float)nd->y * voxelSize, gridMin[2] + (float)nd->z * voxelSize);
                float ext = (float)nd->size * voxelSize;
                v3 nodeMax = v3_(nodeMin.x + ext, nodeMin.y + ext, nodeMin.z + ext);
                v3 center = v3_(0.5f * (nodeMin.x + nodeMax.x), 0.5f * (nodeMin.y + nodeMax.y), 0.5f * (nodeMin.z + nodeMax.z));
                v3 p = v3_(ro.x + rd.x * t, ro.y + rd.y * t, ro.z + rd.z * t);
                tr.hit = 1;
                tr.normal = v3_normalize(v3_sub(p, center));
            }
            shade_store(&tr, outRGBA + pix * 4);
        }
}

/* The consumer of octreeRaySkip in drawRaycast (S/VolumeRaycastRenderer.cpp:1602-1663): 7x7 probe directions through
 * inverse(perspective(45 deg, aspect, 0.1, 5000)) and inverse(view), the traversal, the 15th percentile of the valid
 * distances x 0.75, and the temporal blend with the previous value (0.4 old + 0.6 new).  rd (may be NULL): the 49 directions. */
void orc_probe_rays(const float view[16], const float eye[3], float aspect, float* rd) {
    float P[16], invP[16], invV[16];
    orc_perspective(orc_radians(45.0f), aspect, 0.1f, 5000.0f, P);
    orc_mat4_inverse(P, invP);
    orc_mat4_inverse(view, invV);
    const int gridSize = 7;
    const float sampleOffset = 0.2f;
    for (int y = 0; y < gridSize; y++)
        for (int x = 0; x < gridSize; x++) {
            float ndcX = ((float)x / (gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
            float ndcY = ((float)y / (gridSize - 1) - 0.5f) * 2.0f * sampleOffset;
            const float c[4] = { ndcX, ndcY, 1.f, 1.f };
            float vp[4], wp[4];
            /* glm mat4 * vec4 (type_mat4x4.inl:561-571): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
            for (int r = 0; r < 4; r++) vp[r] = (M(invP, 0, r) * c[0] + M(invP, 1, r) * c[1]) + (M(invP, 2, r) * c[2] + M(invP, 3, r) * c[3]);
            const float w = vp[3];
            for (int r = 0; r < 4; r++) vp[r] = vp[r] / w;                     /* viewPos /= viewPos.w */
            for (int r = 0; r < 4; r++) wp[r] = (M(invV, 0, r) * vp[0] + M(invV, 1, r) * vp[1]) + (M(invV, 2, r) * vp[2] + M(invV, 3, r) * vp[3]);
            v3 d = v3_normalize(v3_(wp[0] - eye[0], wp[1] - eye[1], wp[2] - eye[2]));
            float* o = rd + 3 * (y * gridSize + x);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
}

static int cmp_float(const void* a, const void* b) { float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }

float orc_probe_skip_distance(const orc_node* nodes, int64_t n, const float gridMin[3], float voxelSize,
                              const float view[16], const float eye[3], float aspect, const uint8_t* vis, float lastSkipDistance) {
    float rd[49 * 3], valid[49];
    int nv = 0;
    orc_probe_rays(view, eye, aspect, rd);
    for (int i = 0; i < 49; i++) {
        float t = n > 0 ? ray_skip_rec(nodes, 0, gridMin, voxelSize, v3_(eye[0], eye[1], eye[2]), v3_(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]), 0.0f, 1e30f, vis, NULL) : 1e30f;
        if (t < 1e30f && t > 0.0f) valid[nv++] = t;                            /* :1640-1642 */
    }
    float skipDistance = 0.0f;
    if (nv > 0) {
        qsort(valid, (size_t)nv, sizeof(float), cmp_float);                    /* :1649 */
        int safeIndex = (int)((float)nv * 0.15f);                              /* :1650 */
        if (safeIndex < 0) safeIndex = 0;
        skipDistance = valid[safeIndex];
        skipDistance *= 0.75f;                                                 /* :1654 */
    }
    const float blendFactor = 0.4f;                                            /* :1659-1661 */
    return lastSkipDistance * blendFactor + skipDistance * (1.0f - blendFactor);
}

/* ------------------------------------------------------------------ */
/* N2: localMC (S/OctreeVoxel.cpp:633-640, 780-879), leaf triangles, triangle renderer */
/* ------------------------------------------------------------------ */
#include "mc_cases.inc"   /* derived by probing the compiled reference: tools/gen_mc_tables.py */

static const int kMcCorner[8][3] = { {0,0,0},{1,0,0},{1,1,0},{0,1,0},{0,0,1},{1,0,1},{1,1,1},{0,1,1} };
static const int kMcEdge[12][2] = { {0,1},{1,2},{2,3},{3,0},{4,5},{5,6},{6,7},{7,4},{0,4},{1,5},{2,6},{3,7} };   /* S/OctreeVoxel.h:16-20 */

static inline v3 vertex_interp(float iso, v3 p1, v3 p2, float v1, float v2) {
    /* S/OctreeVoxel.cpp:633-640 */
    if (fabsf(iso - v1) < 0.00001f) return p1;
    if (fabsf(iso - v2) < 0.00001f) return p2;
    if (fabsf(v1 - v2) < 0.00001f) return p1;
    float mu = (iso - v1) / (v2 - v1);
    v3 d = v3_sub(p2, p1);
    return v3_add(p1, v3_(mu * d.x, mu * d.y, mu * d.z));
}

static inline int hexval(char c) { return c <= '9' ? c - '0' : c - 'a' + 10; }

typedef struct { float* p; size_t n, cap; } fvec;
static void fvec_push(fvec* v, const float* src, size_t k) {
    if (v->n + k > v->cap) { v->cap = (v->cap ? v->cap * 2 : 4096) + k; v->p = (float*)realloc(v->p, v->cap * sizeof(float)); }
    memcpy(v->p + v->n, src, k * sizeof(float)); v->n += k;
}

/* appends triangles of the cells [x0,x0+size)^3; stride 18 (reference MCTriangle) or 12 (v0,v1,v2,n) */
static void local_mc_append(const orc_grid* g, int x0, int y0, int z0, int size, fvec* out, int stride) {
    const float vx = g->voxelSize;
    for (int z = z0; z < z0 + size && z < g->dimZ - 1; z++)
        for (int y = y0; y < y0 + size && y < g->dimY - 1; y++)
            for (int x = x0; x < x0 + size && x < g->dimX - 1; x++) {
                v3 pos[8]; float val[8]; int cube = 0;
                for (int i = 0; i < 8; i++) {
                    int cx = x + kMcCorner[i][0], cy = y + kMcCorner[i][1], cz = z + kMcCorner[i][2];
                    pos[i] = v3_(g->minX + cx * vx, g->minY + cy * vx, g->minZ + cz * vx);
                    int oob = cx < 0 || cy < 0 || cz < 0 || cx >= g->dimX || cy >= g->dimY || cz >= g->dimZ;
                    val[i] = (!oob && g->data[(size_t)cx + (size_t)cy * g->dimX + (size_t)cz * ((size_t)g->dimX * g->dimY)] == 1) ? -1.0f : 1.0f;
                    if (val[i] < 0) cube |= 1 << i;
                }
                const char* edges = kMcCaseEdges[cube];
                if (edges[0] == 'f') continue;
                v3 vert[12];
                for (const char* e = edges; *e != 'f'; e++) {
                    int id = hexval(*e), a = kMcEdge[id][0], b = kMcEdge[id][1];
                    vert[id] = vertex_interp(0.0f, pos[a], pos[b], val[a], val[b]);
                }
                for (const char* e = edges; *e != 'f'; e += 3) {
                    v3 a = vert[hexval(e[0])], b = vert[hexval(e[1])], c = vert[hexval(e[2])];
                    v3 nrm = v3_normalize(v3_cross(v3_sub(b, a), v3_sub(c, a)));
                    float t[18] = { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, nrm.x, nrm.y, nrm.z, nrm.x, nrm.y, nrm.z, nrm.x, nrm.y, nrm.z };
                    fvec_push(out, t, (size_t)stride);
                }
            }
}

int64_t orc_local_mc(const orc_grid* g, int x0, int y0, int z0, int size, float** out18) {
    fvec v = { NULL, 0, 0 };
    local_mc_append(g, x0, y0, z0, size, &v, 18);
    if (!v.p) v.p = (float*)malloc(4);
    *out18 = v.p;
    return (int64_t)(v.n / 18);
}

int64_t orc_build_leaf_triangles(const orc_grid* g, const orc_node* nodes, int64_t n, float** tris, int32_t** triOffset) {
    fvec v = { NULL, 0, 0 };
    int32_t* off = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) {
        off[i] = (int32_t)(v.n / 12);
        if (nodes[i].isLeaf == 1) local_mc_append(g, nodes[i].x, nodes[i].y, nodes[i].z, nodes[i].size, &v, 12);
    }
    off[n] = (int32_t)(v.n / 12);
    if (!v.p) v.p = (float*)malloc(4);
    *tris = v.p; *triOffset = off;
    return (int64_t)(v.n / 12);
}

/* Moeller-Trumbore, fixed operation order (the HIP kernel mirrors it) */
static inline int ray_triangle(v3 ro, v3 rd, const float* T, float* tOut) {
    v3 v0 = v3_(T[0], T[1], T[2]), v1 = v3_(T[3], T[4], T[5]), v2 = v3_(T[6], T[7], T[8]);
    v3 e1 = v3_sub(v1, v0), e2 = v3_sub(v2, v0);
    v3 p = v3_cross(rd, e2);
    float det = v3_dot(e1, p);
    if (fabsf(det) < 1e-12f) return 0;
    float invDet = 1.0f / det;
    v3 tv = v3_sub(ro, v0);
    float u = v3_dot(tv, p) * invDet;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3_cross(tv, e1);
    float v = v3_dot(rd, q) * invDet;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = v3_dot(e2, q) * invDet;
    if (!(t > 0.0f)) return 0;
    *tOut = t;
    return 1;
}

typedef struct { int hit; int steps; float t; v3 normal; } tri_result;

static inline tri_result trace_triangles(const orc_node* nodes, const float* tris, const int32_t* triOffset,
                                         const frame_consts* fc, v3 ro, v3 rd) {
    tri_result r; r.hit = 0; r.steps = 0; r.t = 1e30f; r.normal = v3_(0, 0, 0);
    float closestT = 1e30f;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    int traversalSteps = 0;
    v3 gmn = v3_(fc->gridMin[0], fc->gridMin[1], fc->gridMin[2]);
    float vs = fc->voxelSize;
    while (sp > 0 && traversalSteps < MAX_TRAVERSAL_STEPS) {
        sp--;
        int nodeIdx = stack[sp];
        if (nodeIdx < 0) continue;
        traversalSteps++;
        const orc_node* node = &nodes[nodeIdx];
        v3 nodeMin = v3_(gmn.x + (float)node->x * vs, gmn.y + (float)node->y * vs, gmn.z + (float)node->z * vs);
        float ext = (float)node->size * vs;
        v3 nodeMax = v3_(nodeMin.x + ext, nodeMin.y + ext, nodeMin.z + ext);
        float tNear, tFar;
        if (!intersect_aabb(ro, rd, nodeMin, nodeMax, &tNear, &tFar)) continue;
        if (tNear >= closestT) continue;
        if (node->isUniform == 1 || node->isLeaf == 1) {
            float bestT = closestT; int best = -1;
            for (int k = triOffset[nodeIdx]; k < triOffset[nodeIdx + 1]; k++) {
                float t;
                if (ray_triangle(ro, rd, tris + (size_t)k * 12, &t) && t < bestT) { bestT = t; best = k; }
            }
            if (best >= 0) {
                r.hit = 1; r.t = bestT;
                r.normal = v3_(tris[(size_t)best * 12 + 9], tris[(size_t)best * 12 + 10], tris[(size_t)best * 12 + 11]);
                break;
            }
            continue;
        }
        for (int i = 0; i < 8; i++) {
            int childIdx = node->child[i];
            if (childIdx >= 0) stack[sp++] = childIdx;
        }
    }
    r.steps = traversalSteps;
    return r;
}

void orc_render_triangles(const orc_node* nodes, int64_t n, const float* tris, const int32_t* triOffset,
                          const float gridMin[3], float voxelSize, const float view[16], const float camPos[3],
                          float aspect, float fovDeg, int W, int H, int shadow, float* out, orc_stats* stats, int nthreads) {
    (void)n;
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
    v3 l = v3_normalize(v3_(-1.0f, -1.0f, -1.0f));
    v3 nl = v3_(-l.x, -l.y, -l.z);
    float bias = voxelSize * 1e-3f;
    uint64_t pops = 0, hits = 0, capped = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+:pops,hits,capped)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            tri_result tr = trace_triangles(nodes, tris, triOffset, &fc, ro, rd);
            float* o = out + ((size_t)py * W + px) * 4;
            pops += (uint64_t)tr.steps;
            if (!tr.hit) { o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f; capped += (uint64_t)(tr.steps >= MAX_TRAVERSAL_STEPS); continue; }
            hits++;
            v3 nrm = tr.normal;
            if (v3_dot(nrm, rd) > 0.0f) nrm = v3_(-nrm.x, -nrm.y, -nrm.z);
            float ndotl = gmax(0.0f, v3_dot(nrm, nl));
            if (shadow) {
                v3 p = v3_(ro.x + rd.x * tr.t, ro.y + rd.y * tr.t, ro.z + rd.z * tr.t);
                v3 so = v3_(p.x + nrm.x * bias, p.y + nrm.y * bias, p.z + nrm.z * bias);
                tri_result sh = trace_triangles(nodes, tris, triOffset, &fc, so, nl);
                pops += (uint64_t)sh.steps;
                if (sh.hit) ndotl = 0.0f;
            }
            o[0] = 1.0f * ndotl + 0.1f; o[1] = 0.8f * ndotl + 0.1f; o[2] = 0.6f * ndotl + 0.1f; o[3] = 1.0f;
        }
    if (stats) { stats->rays = (uint64_t)W * H; stats->pops = pops; stats->hits = hits; stats->capped = capped; stats->internal = 0; stats->max_stack = 0; stats->pad = 0; }
}
