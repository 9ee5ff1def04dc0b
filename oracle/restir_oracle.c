/*
 * restir_oracle.c -- CPU restatement (plain C99, FP32, no FMA contraction) of the ReSTIR-DI
 * per-pixel pipeline and MTBVH traversal of HummaWhite/ReSTIR.
 *
 * TEST INFRASTRUCTURE ONLY -- see restir_oracle.h for the rules and the pinning status.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include "restir_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * GLM 0.9.6.3 scalar/vector semantics (external/include/glm/detail/func_common.inl,
 * func_geometric.inl, func_exponential.inl)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

#define PI_F      3.1415926535897932384626422832795028841971f  /* mathUtil.h:11 */
#define PI_TWO_F  6.2831853071795864769252867665590057683943f  /* mathUtil.h:12 */
#define GLM_PI_F  ((float)3.14159265358979323846264338327950288) /* glm::pi<float>() */
#define NULL_PRIM (-1)
#define NULL_TEXTURE (-1)     /* material.h:11 */
#define PROCEDURAL_TEX (-2)   /* material.h:13 */
#define INVALID_PDF (-1.f)

static inline float g_abs(float x) { return x >= 0.f ? x : -x; }           /* func_common.inl:56 */
static inline float g_min(float x, float y) { return x < y ? x : y; }      /* :413 */
static inline float g_max(float x, float y) { return x > y ? x : y; }      /* :434 */
static inline int   i_min(int x, int y) { return x < y ? x : y; }
static inline int   i_max(int x, int y) { return x > y ? x : y; }
static inline int   i_clamp(int x, int lo, int hi) { return i_min(i_max(x, lo), hi); }
static inline float g_radians(float d) { return d * (float)0.01745329251994329576923690768489; }
static inline float g_inversesqrt(float x) { return 1.f / sqrtf(x); }      /* func_exponential.inl:150 */

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 dvv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 scl(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 dvs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 adds(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* func_geometric.inl:69-70 */
static inline v3 cross(v3 x, v3 y) {                                              /* :138-141 */
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float length3(v3 v) { return sqrtf(dot(v, v)); }                    /* :99 */
static inline v3 normalize3(v3 v) { return scl(v, g_inversesqrt(dot(v, v))); }    /* :158 */
static inline v3 vmin(v3 a, v3 b) { return V3(g_min(a.x, b.x), g_min(a.y, b.y), g_min(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return V3(g_max(a.x, b.x), g_max(a.y, b.y), g_max(a.z, b.z)); }
static inline v3 mix3s(v3 x, v3 y, float a) { return add(x, scl(sub(y, x), a)); } /* x + a*(y-x) */
static inline v3 mix3v(v3 x, v3 y, v3 a) { return add(x, mul(a, sub(y, x))); }
static inline float mixf(float x, float y, float a) { return x + a * (y - x); }
static inline float comp(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* column-major mat3 * vec3 (type_mat3x3.inl:487-493) */
static inline v3 m3mul(v3 c0, v3 c1, v3 c2, v3 v) {
    return V3(c0.x * v.x + c1.x * v.y + c2.x * v.z,
              c0.y * v.x + c1.y * v.y + c2.y * v.z,
              c0.z * v.x + c1.z * v.y + c2.z * v.z);
}

/* float -> int conversion with the device semantics the reference kernels run under
 * (cvt.rzi.s32.f32 / v_cvt_i32_f32: truncate, saturate, NaN -> 0).  C leaves the
 * out-of-range cases undefined, so they are spelled out. */
static inline int f2i(float f) {
    if (isnan(f)) return 0;
    if (f >= 2147483648.f) return INT_MAX;
    if (f <= -2147483648.f) return INT_MIN;
    return (int)f;
}
/* host-side (x86 cvttss2si) semantics, used only by the host BVH builder (bvh.cpp:83,118) */
static inline int f2i_host(float f) {
    if (isnan(f) || f >= 2147483648.f || f < -2147483648.f) return INT_MIN;
    return (int)f;
}

static inline int is_nan_or_inf(float x) { return isnan(x) || isinf(x); }   /* mathUtil.h:56-58 */
static inline int has_nan_or_inf(v3 v) {                                   /* mathUtil.h:60-62 */
    return isnan(v.x) || isnan(v.y) || isnan(v.z) || isinf(v.x) || isinf(v.y) || isinf(v.z);
}
static inline float sat_dot(v3 a, v3 b) { return g_max(dot(a, b), 0.f); }   /* :64-66 */
static inline float abs_dot(v3 a, v3 b) { return g_abs(dot(a, b)); }        /* :68-70 */
static inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; } /* :72-75 */
static inline float luminance(v3 c) { return dot(c, V3(.2126f, .7152f, .0722f)); } /* :119-123 */
static inline float triangle_area(v3 v0, v3 v1, v3 v2) {                    /* :86-88 */
    return length3(cross(sub(v1, v0), sub(v2, v0))) * .5f;
}
static inline v3 triangle_normal(v3 v0, v3 v1, v3 v2) {                     /* :90-92 */
    return normalize3(cross(sub(v1, v0), sub(v2, v0)));
}
static inline v3 sample_triangle_uniform(v3 v0, v3 v1, v3 v2, float ru, float rv) { /* :94-100 */
    float r = sqrtf(rv);
    float u = 1.f - r;
    float v = ru * r;
    return add(add(scl(v1, u), scl(v2, v)), scl(v0, 1.f - u - v));
}
static inline v2 to_concentric_disk(float x, float y) {                     /* :128-132 (polar map) */
    float r = sqrtf(x);
    float theta = y * PI_F * 2.0f;
    v2 o = { cosf(theta) * r, sinf(theta) * r };
    return o;
}
static inline float pdf_area_to_solid_angle(float pdf, v3 x, v3 y, v3 ny) { /* :182-185 */
    v3 yx = sub(x, y);
    return pdf * dot(yx, yx) / abs_dot(ny, normalize3(yx));
}
static inline uint32_t utilhash(uint32_t a) {                               /* :190-198 */
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}

/* ------------------------------------------------------------------------------------------
 * RNG: thrust::default_random_engine == minstd_rand == LCG<uint32, 48271, 0, 2^31-1>
 * (thrust/random/linear_congruential_engine.h, detail/linear_congruential_engine.inl seed/step,
 * detail/mod.h Schrage form, detail/uniform_real_distribution.inl:67-80).  The reference pins no
 * Thrust version (CMakeLists.txt:26); validated here against rocThrust 2.8.5 (oracle/_ref).
 * ---------------------------------------------------------------------------------------- */
/* One sampler type for both branches of src/sampler.h: data == NULL is the default thrust engine (sampler.h:38-49), data != NULL the
 * Sobol branch (sampler.h:9-36, SAMPLER_USE_SOBOL): Sampler{ptr, scramble, data}. */
typedef struct { uint32_t x; const uint32_t* data; uint32_t scramble; int ptr; } rng_t;
#define SOBOL_SAMPLE_DIM 200      /* sampler.h:11 SobolSampleDim */

#define LCG_A 48271u
#define LCG_M 2147483647u

static inline rng_t rng_seed_raw(uint32_t s) {
    rng_t r;
    uint32_t v = s % LCG_M;
    r.x = (v == 0) ? 1u : v;      /* c == 0 and s % m == 0 -> state 1 */
    r.data = 0; r.scramble = 0u; r.ptr = 0;
    return r;
}
/* sampler.h:41-44 (default engine); sampler.h:30-32 (Sobol: Sampler(iter * SobolSampleDim + dim, utilhash(index), data)).
 * `data` is scene->sampleSequence (restir.cu:127, pathtrace.cu:170,288,339), NULL for the default engine. */
static inline rng_t make_seeded_random_engine(int iter, int index, int dim, const uint32_t* data) {
    if (data) {
        rng_t r;
        r.x = 0u; r.data = data; r.scramble = utilhash((uint32_t)index); r.ptr = iter * SOBOL_SAMPLE_DIM + dim;
        return r;
    }
    uint32_t h = utilhash((1u << 31) | ((uint32_t)dim << 22) | (uint32_t)iter) ^ utilhash((uint32_t)index);
    return rng_seed_raw(h);
}
static inline uint32_t rng_next(rng_t* r) {
    const uint32_t q = LCG_M / LCG_A, rr = LCG_M % LCG_A;   /* Schrage (mod.h:38-52) */
    uint32_t x = r->x;
    uint32_t t1 = LCG_A * (x % q);
    uint32_t t2 = rr * (x / q);
    x = (t1 >= t2) ? (t1 - t2) : (LCG_M - t2 + t1);
    r->x = x;
    return x;
}
/* sampler.h:46-48: uniform_real_distribution<float>(0,1): float(x - min) / (1.f + float(max - min)) */
static inline float sample1D(rng_t* r) {
    if (r->data) {                                   /* Sampler::sample, sampler.h:19-23 */
        uint32_t v = r->data[r->ptr++] ^ r->scramble;
        r->scramble = utilhash(r->scramble);
        return (float)v * 0x1p-32f;                  /* `r * 0x1p-32f`: uint32 -> float (round to nearest), then an exact scaling */
    }
    float result = (float)(uint32_t)(rng_next(r) - 1u);
    result /= (1.f + (float)(uint32_t)(2147483646u - 1u));
    return (result * (1.f - 0.f)) + 0.f;
}
typedef struct { float x, y, z, w; } v4;
/* sampler.h:51-61: components drawn left to right */
static inline v2 sample2D(rng_t* r) { v2 o; o.x = sample1D(r); o.y = sample1D(r); return o; }
static inline v4 sample4D(rng_t* r) {
    v4 o; o.x = sample1D(r); o.y = sample1D(r); o.z = sample1D(r); o.w = sample1D(r); return o;
}

/* ------------------------------------------------------------------------------------------
 * Ray / triangle / AABB
 * ---------------------------------------------------------------------------------------- */
typedef struct { v3 origin, direction; } ray_t;

/* intersections.h:13-15 */
static inline ray_t make_offseted_ray(v3 ori, v3 dir) {
    ray_t r; r.origin = add(ori, scl(dir, 1e-5f)); r.direction = dir; return r;
}

/* intersections.h:17-54 */
static int intersect_triangle(ray_t ray, v3 v0, v3 v1, v3 v2_, v2* bary, float* dist) {
    v3 e01 = sub(v1, v0);
    v3 e02 = sub(v2_, v0);
    v3 ori = ray.origin;
    v3 dir = ray.direction;
    v3 p = cross(dir, e02);

    float det = dot(p, e01);
    if (g_abs(det) < FLT_EPSILON) {
        return 0;
    }
    v3 v0ToOri = sub(ori, v0);
    if (det < 0.f) {
        det = -det;
        v0ToOri = neg(v0ToOri);
    }
    bary->x = dot(v0ToOri, p);
    if (bary->x < 0.f || bary->x > det) {
        return 0;
    }
    v3 perp = cross(v0ToOri, e01);
    bary->y = dot(dir, perp);
    if (bary->y < 0.f || bary->x + bary->y > det) {
        return 0;
    }
    float detInv = 1.f / det;
    *dist = dot(e02, perp) * detInv;
    bary->x *= detInv;
    bary->y *= detInv;
    return *dist > 0.f;
}

static inline int between(float x, float lo, float hi) { return x >= lo && x <= hi; } /* mathUtil.h:32-34 */

/* bvh.h:69-79 */
static inline int dist_min_max(float tMin1, float tMin2, float tMax1, float tMax2, float* tMin) {
    *tMin = fminf(tMin1, tMin2);
    float tMax = fmaxf(tMax1, tMax2);
    return (tMax >= 0.f && tMax >= *tMin);
}
static inline int dist_max_min(float tMin1, float tMin2, float tMax1, float tMax2, float* tMin) {
    *tMin = fmaxf(tMin1, tMin2);
    float tMax = fminf(tMax1, tMax2);
    return (tMax >= 0.f && tMax >= *tMin);
}

/* bvh.h:85-157 */
static int aabb_intersect(v3 pMin, v3 pMax, ray_t ray, float* tMin) {
    const float Eps = 1e-6f;
    v3 ori = ray.origin;
    v3 dir = ray.direction;

    if (g_abs(dir.x) > 1.f - Eps) {
        if (between(ori.y, pMin.y, pMax.y) && between(ori.z, pMin.z, pMax.z)) {
            float dirInvX = 1.f / dir.x;
            float t1 = (pMin.x - ori.x) * dirInvX;
            float t2 = (pMax.x - ori.x) * dirInvX;
            return dist_min_max(t1, t2, t1, t2, tMin);
        }
        return 0;
    }
    else if (g_abs(dir.y) > 1.f - Eps) {
        if (between(ori.z, pMin.z, pMax.z) && between(ori.x, pMin.x, pMax.x)) {
            float dirInvY = 1.f / dir.y;
            float t1 = (pMin.y - ori.y) * dirInvY;
            float t2 = (pMax.y - ori.y) * dirInvY;
            return dist_min_max(t1, t2, t1, t2, tMin);
        }
        return 0;
    }
    else if (g_abs(dir.z) > 1.f - Eps) {
        if (between(ori.x, pMin.x, pMax.x) && between(ori.y, pMin.y, pMax.y)) {
            float dirInvZ = 1.f / dir.z;
            float t1 = (pMin.z - ori.z) * dirInvZ;
            float t2 = (pMax.z - ori.z) * dirInvZ;
            return dist_min_max(t1, t2, t1, t2, tMin);
        }
        return 0;
    }
    v3 dirInv = V3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
    v3 t1 = mul(sub(pMin, ori), dirInv);
    v3 t2 = mul(sub(pMax, ori), dirInv);

    v3 tNear = vmin(t1, t2);
    v3 tFar = vmax(t1, t2);
    v3 tDist = sub(tFar, tNear);

    float yz = tFar.z - tNear.y;
    float zx = tFar.x - tNear.z;
    float xy = tFar.y - tNear.x;

    if (g_abs(dir.x) < Eps && tDist.y + tDist.z > yz) {
        return dist_max_min(tNear.y, tNear.z, tFar.y, tFar.z, tMin);
    }
    if (g_abs(dir.y) < Eps && tDist.z + tDist.x > zx) {
        return dist_max_min(tNear.z, tNear.x, tFar.z, tFar.x, tMin);
    }
    if (g_abs(dir.z) < Eps && tDist.x + tDist.y > xy) {
        return dist_max_min(tNear.x, tNear.y, tFar.x, tFar.y, tMin);
    }
    if (tDist.y + tDist.z > yz && tDist.z + tDist.x > zx && tDist.x + tDist.y > xy) {
        return dist_max_min(fmaxf(tNear.x, tNear.y), tNear.z, fminf(tFar.x, tFar.y), tFar.z, tMin);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Camera (sceneStructs.h:22-126)
 * ---------------------------------------------------------------------------------------- */
static inline v3 cam_m3inv_mul(const orc_camera* c, v3 v) {
    const float* m = c->rotationMatInv;
    return m3mul(V3(m[0], m[1], m[2]), V3(m[3], m[4], m[5]), V3(m[6], m[7], m[8]), v);
}

/* sceneStructs.h:69-86 */
static ray_t camera_sample(const orc_camera* c, int x, int y, v4 r) {
    ray_t ray;
    float aspect = (float)c->resolution[0] / (float)c->resolution[1];
    float tanFovY = tanf(g_radians(c->fov[1]));
    v2 pixelSize = { 1.f / (float)c->resolution[0], 1.f / (float)c->resolution[1] };
    v2 scr = { (float)x * pixelSize.x, (float)y * pixelSize.y };
    v2 ruv = { scr.x + pixelSize.x * r.x, scr.y + pixelSize.y * r.y };
    ruv.x = 1.f - ruv.x * 2.f;
    ruv.y = 1.f - ruv.y * 2.f;

    v3 pLens = V3(0.f * c->lensRadius, 0.f * c->lensRadius, 0.f);
    v3 pFocusPlane = scl(V3(ruv.x * aspect * tanFovY, ruv.y * 1.f * tanFovY, 1.f), c->focalDist);
    v3 dir = sub(pFocusPlane, pLens);

    ray.direction = normalize3(m3mul(ld3(c->right), ld3(c->up), ld3(c->view), dir));
    ray.origin = add(add(ld3(c->position), scl(ld3(c->right), pLens.x)), scl(ld3(c->up), pLens.y));
    return ray;
}

/* sceneStructs.h:23-41 */
static v2 camera_raster_uv(const orc_camera* c, v3 pos) {
    v3 dir = normalize3(sub(pos, ld3(c->position)));
    float d = 1.f / dot(dir, ld3(c->view));
    v3 p = cam_m3inv_mul(c, scl(dir, d));
    float aspect = (float)c->resolution[0] / (float)c->resolution[1];
    float tanFovY = tanf(g_radians(c->fov[1]));
    p = dvv(p, V3(aspect * tanFovY, 1.f * tanFovY, 1.f));
    v2 ndc = { -p.x, -p.y };
    v2 o = { ndc.x * .5f + .5f, ndc.y * .5f + .5f };
    return o;
}
/* sceneStructs.h:43-46 */
static void camera_raster_coord(const orc_camera* c, v3 pos, int* ox, int* oy) {
    v2 ndc = camera_raster_uv(c, pos);
    *ox = f2i((float)c->resolution[0] * ndc.x);
    *oy = f2i((float)c->resolution[1] * ndc.y);
}
/* sceneStructs.h:48-64 */
static v3 camera_get_position(const orc_camera* c, int x, int y, float dist) {
    float aspect = (float)c->resolution[0] / (float)c->resolution[1];
    float tanFovY = tanf(g_radians(c->fov[1]));
    v2 pixelSize = { 1.f / (float)c->resolution[0], 1.f / (float)c->resolution[1] };
    v2 scr = { (float)x * pixelSize.x, (float)y * pixelSize.y };
    v2 ruv = { scr.x + pixelSize.x * .5f, scr.y + pixelSize.y * .5f };
    ruv.x = 1.f - ruv.x * 2.f;
    ruv.y = 1.f - ruv.y * 2.f;

    v3 pLens = V3(0.f * c->lensRadius, 0.f * c->lensRadius, 0.f);
    v3 pFocusPlane = scl(V3(ruv.x * aspect * tanFovY, ruv.y * 1.f * tanFovY, 1.f), c->focalDist);
    v3 dir = sub(pFocusPlane, pLens);
    dir = normalize3(m3mul(ld3(c->right), ld3(c->up), ld3(c->view), dir));
    v3 ori = add(add(ld3(c->position), scl(ld3(c->right), pLens.x)), scl(ld3(c->up), pLens.y));
    return add(ori, scl(dir, dist));
}

/* sceneStructs.h:88-102 (host); glm::inverse(mat3): type_mat3x3.inl:37-56 */
void orc_camera_update(orc_camera* c) {
    float yaw = g_radians(c->rotation[0]);
    float pitch = g_radians(c->rotation[1]);
    v3 view;
    view.x = cosf(yaw) * cosf(pitch);
    view.z = sinf(yaw) * cosf(pitch);
    view.y = sinf(pitch);
    view = normalize3(view);
    v3 right = normalize3(cross(view, V3(0.f, 1.f, 0.f)));
    v3 up = normalize3(cross(right, view));
    st3(c->view, view); st3(c->right, right); st3(c->up, up);

    float m[3][3] = { { right.x, right.y, right.z }, { up.x, up.y, up.z }, { view.x, view.y, view.z } };
    float ood = 1.f / (
        + m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
        - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
        + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    float inv[3][3];
    inv[0][0] = + (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
    inv[1][0] = - (m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
    inv[2][0] = + (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
    inv[0][1] = - (m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
    inv[1][1] = + (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
    inv[2][1] = - (m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
    inv[0][2] = + (m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
    inv[1][2] = - (m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
    inv[2][2] = + (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++)
            c->rotationMatInv[col * 3 + row] = inv[col][row];
}

/* ------------------------------------------------------------------------------------------
 * Material::BSDF (material.h:34-124,171-186,218-228)
 * ---------------------------------------------------------------------------------------- */
enum { MAT_LAMBERTIAN = 0, MAT_METALLIC = 1, MAT_DIELECTRIC = 2, MAT_DISNEY = 3, MAT_LIGHT = 4 };

static inline float schlick_g(float cosTheta, float alpha) {       /* material.h:63-66 */
    float a = alpha * .5f;
    return cosTheta / (cosTheta * (1.f - a) + a);
}
static inline float smith_g(float cosWo, float cosWi, float alpha) { /* :68-70 */
    return schlick_g(g_abs(cosWo), alpha) * schlick_g(g_abs(cosWi), alpha);
}
static inline float gtr2_distrib(float cosTheta, float alpha) {      /* :72-81 */
    if (cosTheta < 1e-6f) {
        return 0.f;
    }
    float aa = alpha * alpha;
    float nom = aa;
    float denom = cosTheta * cosTheta * (aa - 1.f) + 1.f;
    denom = denom * denom * PI_F;
    return nom / denom;
}
static inline v3 lambertian_bsdf(const orc_material* m) {            /* :122-124: baseColor * 1.f / Pi */
    return dvs(scl(ld3(m->baseColor), 1.f), PI_F);
}
static v3 metallic_workflow_bsdf(const orc_material* m, v3 n, v3 wo, v3 wi) { /* :171-186 */
    float alpha = m->roughness * m->roughness;
    v3 h = normalize3(add(wo, wi));

    float cosO = dot(n, wo);
    float cosI = dot(n, wi);
    if (cosI * cosO < 1e-7f) {
        return v3s(0.f);
    }
    v3 baseColor = ld3(m->baseColor);
    v3 f0 = mix3s(v3s(.08f), baseColor, m->metallic);
    v3 f = mix3s(f0, v3s(1.f), pow5(1.f - dot(h, wo)));             /* fresnelSchlick :39-41 */
    float g = smith_g(cosO, cosI, alpha);
    float d = gtr2_distrib(dot(n, h), alpha);

    v3 diff = scl(dvs(scl(baseColor, 1.f), PI_F), 1.f - m->metallic);
    return mix3v(diff, v3s(g * d / (4.f * cosI * cosO)), f);
}
static v3 material_bsdf(const orc_material* m, v3 n, v3 wo, v3 wi) { /* :218-228 */
    switch (m->type) {
    case MAT_LAMBERTIAN: return lambertian_bsdf(m);
    case MAT_METALLIC:   return metallic_workflow_bsdf(m, n, wo, wi);
    case MAT_DIELECTRIC: return v3s(0.f);
    }
    return v3s(0.f);
}

/* glm::inverse(mat3) (func_matrix.inl compute_inverse<tmat3x3>), columns in / columns out */
static inline void m3_inverse(v3 c0, v3 c1, v3 c2, v3* o0, v3* o1, v3* o2) {
    const float m[3][3] = { { c0.x, c0.y, c0.z }, { c1.x, c1.y, c1.z }, { c2.x, c2.y, c2.z } };
    float ood = 1.f / (
        + m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
        - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
        + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    *o0 = V3(+ (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood, - (m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood, + (m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood);
    *o1 = V3(- (m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood, + (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood, - (m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood);
    *o2 = V3(+ (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood, - (m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood, + (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood);
}

/* ------------------------------------------------------------------------------------------
 * libm calls of __device__ code (see orc_set_libm_mode in the header)
 * ---------------------------------------------------------------------------------------- */
static int g_libm_cr = 0;
void orc_set_libm_mode(int correctlyRounded) { g_libm_cr = correctlyRounded != 0; }
static inline float m_sin(float x) { return g_libm_cr ? (float)sin((double)x) : sinf(x); }
static inline float m_cos(float x) { return g_libm_cr ? (float)cos((double)x) : cosf(x); }
static inline float m_atan2(float y, float x) { return g_libm_cr ? (float)atan2((double)y, (double)x) : atan2f(y, x); }

static inline float g_fract(float x) { return x - floorf(x); }                 /* func_common.inl:318-321 */

/* mathUtil.h:134-137 */
static inline v3 to_sphere(v2 v) {
    v.x *= PI_TWO_F; v.y *= PI_F;
    return V3(m_cos(v.x) * m_sin(v.y), m_cos(v.y), m_sin(v.x) * m_sin(v.y));
}
/* mathUtil.h:139-144; PiInv is the macro "1.f / Pi" (Q10): x * PiInv * .5f == ((x * 1.f) / Pi) * .5f */
static inline v2 to_plane(v3 v) {
    v2 o;
    o.x = g_fract(((m_atan2(v.z, v.x) * 1.f) / PI_F) * .5f + 1.f);
    o.y = (m_atan2(sqrtf(v.x * v.x + v.z * v.z), v.y) * 1.f) / PI_F;      /* glm::length(vec2) = sqrt(dot) */
    return o;
}
/* mathUtil.h:146-155 */
static inline v3 local_to_world(v3 n, v3 v) {
    v3 t = (g_abs(n.y) > 0.9999f) ? V3(0.f, 0.f, 1.f) : V3(0.f, 1.f, 0.f);
    v3 b = normalize3(cross(n, t));
    t = cross(b, n);
    return normalize3(m3mul(t, b, n, v));
}

/* image.h:41-75 (T = glm::vec3); float -> int conversions truncate (device semantics via f2i) */
static v3 linear_sample(const orc_texture* tex, v2 uv) {
    const float Eps = FLT_MIN;
    const int width = tex->width, height = tex->height;
    uv.x = g_fract(uv.x); uv.y = g_fract(uv.y);
    float fx = uv.x * ((float)width - Eps) + .5f;
    float fy = uv.y * ((float)height - Eps) + .5f;
    int ix = f2i(g_fract(fx) > .5f ? fx : fx - 1);
    if (ix < 0) ix += width;
    int iy = f2i(g_fract(fy) > .5f ? fy : fy - 1);
    if (iy < 0) iy += height;
    int ux = ix + 1;
    if (ux >= width) ux -= width;
    int uy = iy + 1;
    if (uy >= height) uy -= height;
    float lx = g_fract(fx + .5f);
    float ly = g_fract(fy + .5f);
    const float* d = tex->data;
    v3 c1 = mix3s(ld3(d + ((size_t)iy * width + ix) * 3), ld3(d + ((size_t)iy * width + ux) * 3), lx);
    v3 c2 = mix3s(ld3(d + ((size_t)uy * width + ix) * 3), ld3(d + ((size_t)uy * width + ux) * 3), lx);
    return mix3s(c1, c2, ly);
}

/* ------------------------------------------------------------------------------------------
 * Material::sample / pdf (material.h:42-62,82-121,126-170,186-256), the sampling half of the BSDFs
 * ---------------------------------------------------------------------------------------- */
enum { BS_DIFFUSE = 1 << 0, BS_GLOSSY = 1 << 1, BS_SPECULAR = 1 << 2, BS_REFLECTION = 1 << 4, BS_TRANSMISSION = 1 << 5, BS_INVALID = 1 << 15 };
typedef struct { v3 dir, bsdf; float pdf; uint32_t type; } bsdf_sample_t;

/* mathUtil.h:128-132 with the libm mode of __device__ code (the spatial tap keeps to_concentric_disk) */
static inline v2 to_concentric_disk_m(float x, float y) {
    float r = sqrtf(x);
    float theta = y * PI_F * 2.0f;
    v2 o = { m_cos(theta) * r, m_sin(theta) * r };
    return o;
}
/* mathUtil.h:157-161 */
static inline v3 sample_hemisphere_cosine(v3 n, float rx, float ry) {
    v2 d = to_concentric_disk_m(rx, ry);
    float z = sqrtf(1.f - (d.x * d.x + d.y * d.y));
    return local_to_world(n, V3(d.x, d.y, z));
}
/* mathUtil.h:163-180 */
static inline int math_refract(v3 n, v3 wi, float ior, v3* wt) {
    float cosIn = dot(n, wi);
    if (cosIn < 0) ior = 1.f / ior;
    float sin2In = g_max(0.f, 1.f - cosIn * cosIn);
    float sin2Tr = sin2In / (ior * ior);
    if (sin2Tr >= 1.f) return 0;
    float cosTr = sqrtf(1.f - sin2Tr);
    if (cosIn < 0) cosTr = -cosTr;
    *wt = normalize3(add(dvs(neg(wi), ior), scl(n, cosIn / ior - cosTr)));
    return 1;
}
static inline v3 glm_reflect(v3 I, v3 N) { return sub(I, mul(scl(N, dot(N, I)), v3s(2.f))); }   /* func_geometric.inl:176-179 */
static inline float power_heuristic(float f, float g) { float f2 = f * f; return f2 / (f2 + g * g); }   /* mathUtil.h:81-84 */
static inline v3 hdr_to_ldr(v3 c) { return scl(dvv(c, adds(c, 1.f)), 1.f); }                         /* mathUtil.h:36-38 */

/* material.h:42-61 (MATERIAL_DIELECTRIC_USE_SCHLICK_APPROX is not defined -> the exact form) */
static float fresnel_dielectric(float cosIn, float ior) {
    if (cosIn < 0) { ior = 1.f / ior; cosIn = -cosIn; }
    float sinIn = sqrtf(1.f - cosIn * cosIn);
    float sinTr = sinIn / ior;
    if (sinTr >= 1.f) return 1.f;
    float cosTr = sqrtf(1.f - sinTr * sinTr);
    float a = (cosIn - ior * cosTr) / (cosIn + ior * cosTr), b = (ior * cosIn - cosTr) / (ior * cosIn + cosTr);
    return (a * a + b * b) * .5f;
}
/* material.h:82-85 */
static inline float gtr2_pdf(v3 n, v3 m, v3 wo, float alpha) {
    return gtr2_distrib(dot(n, m), alpha) * schlick_g(dot(n, wo), alpha) * abs_dot(m, wo) / abs_dot(n, wo);
}
/* material.h:94-112: GGX visible-normal sampling */
static v3 gtr2_sample(v3 n, v3 wo, float alpha, float rx, float ry) {
    v3 t0 = (g_abs(n.y) > 0.9999f) ? V3(0.f, 0.f, 1.f) : V3(0.f, 1.f, 0.f);      /* Math::localRefMatrix */
    v3 b0 = normalize3(cross(n, t0));
    t0 = cross(b0, n);
    v3 i0, i1, i2;
    m3_inverse(t0, b0, n, &i0, &i1, &i2);
    v3 vh = normalize3(mul(m3mul(i0, i1, i2, wo), V3(alpha, alpha, 1.f)));
    float lenSq = vh.x * vh.x + vh.y * vh.y;
    v3 t = lenSq > 0.f ? dvs(V3(-vh.y, vh.x, 0.f), sqrtf(lenSq)) : V3(1.f, 0.f, 0.f);
    v3 b = cross(vh, t);
    v2 p = to_concentric_disk_m(rx, ry);
    float s = 0.5f * (vh.z + 1.f);
    p.y = (1.f - s) * sqrtf(1.f - p.x * p.x) + s * p.y;
    v3 h = add(add(scl(t, p.x), scl(b, p.y)), scl(vh, sqrtf(g_max(0.f, 1.f - (p.x * p.x + p.y * p.y)))));
    h = V3(h.x * alpha, h.y * alpha, g_max(0.f, h.z));
    return normalize3(m3mul(t0, b0, n, h));
}
/* material.h:186-193 */
static float metallic_workflow_pdf(const orc_material* m, v3 n, v3 wo, v3 wi) {
    v3 h = normalize3(add(wo, wi));
    return mixf((sat_dot(n, wi) * 1.f) / PI_F,
                gtr2_pdf(n, h, wo, m->roughness * m->roughness) / (4.f * abs_dot(h, wo)),
                1.f / (2.f - m->metallic));
}
/* Material::pdf (material.h:230-240) */
static float material_pdf(const orc_material* m, v3 n, v3 wo, v3 wi) {
    switch (m->type) {
    case MAT_LAMBERTIAN: return (sat_dot(n, wi) * 1.f) / PI_F;                   /* :126-128 */
    case MAT_METALLIC:   return metallic_workflow_pdf(m, n, wo, wi);
    case MAT_DIELECTRIC: return 0.f;
    }
    return 0.f;
}
/* Material::sample (material.h:242-256).  Fields the reference leaves unwritten (pdf / dir of an Invalid sample) are 0 here. */
static void material_sample(const orc_material* m, v3 n, v3 wo, v3 r, bsdf_sample_t* sp) {
    sp->dir = v3s(0.f); sp->bsdf = v3s(0.f); sp->pdf = 0.f; sp->type = BS_INVALID;
    switch (m->type) {
    case MAT_LAMBERTIAN:                                                          /* :130-135 */
        sp->dir = sample_hemisphere_cosine(n, r.x, r.y);
        sp->bsdf = dvs(scl(ld3(m->baseColor), 1.f), PI_F);
        sp->pdf = (sat_dot(n, sp->dir) * 1.f) / PI_F;
        sp->type = BS_DIFFUSE | BS_REFLECTION;
        break;
    case MAT_METALLIC: {                                                          /* :195-213 */
        float alpha = m->roughness * m->roughness;
        if (r.z > (1.f / (2.f - m->metallic))) {
            sp->dir = sample_hemisphere_cosine(n, r.x, r.y);
        }
        else {
            v3 h = gtr2_sample(n, wo, alpha, r.x, r.y);
            sp->dir = neg(glm_reflect(wo, h));
        }
        if (dot(n, sp->dir) < 0.f) {
            sp->type = BS_INVALID;
        }
        else {
            sp->bsdf = metallic_workflow_bsdf(m, n, wo, sp->dir);
            sp->pdf = metallic_workflow_pdf(m, n, wo, sp->dir);
            sp->type = BS_GLOSSY | BS_REFLECTION;
        }
        break;
    }
    case MAT_DIELECTRIC: {                                                        /* :145-169 */
        float pdfRefl = fresnel_dielectric(dot(n, wo), m->ior);
        sp->bsdf = ld3(m->baseColor);
        if (r.z < pdfRefl) {
            sp->dir = glm_reflect(neg(wo), n);
            sp->type = BS_SPECULAR | BS_REFLECTION;
            sp->pdf = 1.f;
        }
        else {
            if (!math_refract(n, wo, m->ior, &sp->dir)) { sp->type = BS_INVALID; break; }
            float eta = m->ior;
            if (dot(n, wo) < 0) eta = 1.f / eta;
            sp->bsdf = dvs(sp->bsdf, eta * eta);
            sp->type = BS_SPECULAR | BS_TRANSMISSION;
            sp->pdf = 1.f;
        }
        break;
    }
    default:
        sp->type = BS_INVALID;
    }
}

/* ------------------------------------------------------------------------------------------
 * DevScene services (scene.h:101-198,245-316,394-459)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int primId, matId;
    v3 pos, norm;
    v2 uv;
    v3 wo;
} isect_t;

/* scene.h:101-119 */
static int get_mtbvh_id(v3 dir) {
    v3 a = V3(g_abs(dir.x), g_abs(dir.y), g_abs(dir.z));
    if (a.x > a.y) {
        if (a.x > a.z) return dir.x > 0 ? 0 : 1;
        else           return dir.z > 0 ? 4 : 5;
    }
    else {
        if (a.y > a.z) return dir.y > 0 ? 2 : 3;
        else           return dir.z > 0 ? 4 : 5;
    }
}

static inline void tri_verts(const orc_scene* s, int primId, v3* a, v3* b, v3* c) {
    const float* p = s->vertices + (size_t)primId * 9;
    *a = ld3(p); *b = ld3(p + 3); *c = ld3(p + 6);
}

/* scene.h:135-151 */
static void get_intersec_geom_info(const orc_scene* s, int primId, v2 bary, isect_t* it) {
    v3 va, vb, vc;
    tri_verts(s, primId, &va, &vb, &vc);
    const float* n = s->normals + (size_t)primId * 9;
    v3 na = ld3(n), nb = ld3(n + 3), nc = ld3(n + 6);
    const float* t = s->texcoords + (size_t)primId * 6;
    float w = 1.f - bary.x - bary.y;
    it->pos = add(add(scl(vb, bary.x), scl(vc, bary.y)), scl(va, w));
    it->norm = normalize3(add(add(scl(nb, bary.x), scl(nc, bary.y)), scl(na, w)));
    it->uv.x = t[2] * bary.x + t[4] * bary.y + t[0] * w;
    it->uv.y = t[3] * bary.x + t[5] * bary.y + t[1] * w;
}

/* scene.h:245-284; *walks counts BVH walks for the Mrays/s metric */
static void scene_intersect(const orc_scene* s, ray_t ray, isect_t* it) {
    float closestDist = FLT_MAX;
    int closestPrimId = NULL_PRIM;
    v2 closestBary = { 0.f, 0.f };

    const int* nodes = s->bvhNodes[get_mtbvh_id(neg(ray.direction))];
    int node = 0;
    while (node != s->bvhSize) {
        const int* nd = nodes + (size_t)node * 3;
        const float* bb = s->boundingBoxes + (size_t)nd[1] * 6;
        float boundDist;
        int boundHit = aabb_intersect(ld3(bb), ld3(bb + 3), ray, &boundDist);

        if (boundHit && boundDist < closestDist) {
            int primId = nd[0];
            if (primId != NULL_PRIM) {
                float dist;
                v2 bary;
                v3 va, vb, vc;
                tri_verts(s, primId, &va, &vb, &vc);
                int hit = intersect_triangle(ray, va, vb, vc, &bary, &dist);
                if (hit && dist < closestDist) {
                    closestDist = dist;
                    closestBary = bary;
                    closestPrimId = primId;
                }
            }
            node++;
        }
        else {
            node = nd[2];
        }
    }
    if (closestPrimId != NULL_PRIM) {
        get_intersec_geom_info(s, closestPrimId, closestBary, it);
        it->matId = s->materialIds[closestPrimId];
    }
    it->primId = closestPrimId;
}

/* scene.h:286-316 */
static int scene_test_occlusion(const orc_scene* s, v3 x, v3 y) {
    const float Eps = 1e-4f;
    v3 dir = sub(y, x);
    float dist = length3(dir);
    dir = dvs(dir, dist);
    ray_t ray = make_offseted_ray(x, dir);
    dist -= Eps * 2.f;

    const int* nodes = s->bvhNodes[get_mtbvh_id(neg(ray.direction))];
    int node = 0;
    while (node != s->bvhSize) {
        const int* nd = nodes + (size_t)node * 3;
        const float* bb = s->boundingBoxes + (size_t)nd[1] * 6;
        float boundDist;
        int boundHit = aabb_intersect(ld3(bb), ld3(bb + 3), ray, &boundDist);

        if (boundHit && boundDist < dist) {
            int primId = nd[0];
            if (primId != NULL_PRIM) {
                v3 va, vb, vc;
                v2 bary;
                float d;
                tri_verts(s, primId, &va, &vb, &vc);
                int hit = intersect_triangle(ray, va, vb, vc, &bary, &d);   /* scene.h:165-173 */
                if (hit && d < dist) {
                    return 1;
                }
            }
            node++;
        }
        else {
            node = nd[2];
        }
    }
    return 0;
}

/* scene.h:68-76 */
static v3 procedural_texture(v2 uv) {
    rng_t rng = rng_seed_raw((uint32_t)(f2i(uv.x * 1024.f) * 1024 + f2i(uv.y * 1024.f)));
    float rx = sample1D(&rng);
    float ry = sample1D(&rng);
    float f = (m_sin(uv.x * 10.f * PI_TWO_F + rx * PI_TWO_F) + 1.f) * .5f;
    float g = (m_sin(uv.y * 10.f * PI_TWO_F + ry * PI_TWO_F) + 1.f) * .5f;
    return v3s(f * g);
}

/* scene.h:78-99: material with its maps applied; a normal map also replaces intersec.norm */
static orc_material textured_material_and_surface(const orc_scene* s, isect_t* it) {
    orc_material mat = s->materials[it->matId];
    if (mat.baseColorMapId != NULL_TEXTURE) {
        st3(mat.baseColor, mat.baseColorMapId == PROCEDURAL_TEX ? procedural_texture(it->uv)
                                                                : linear_sample(&s->textures[mat.baseColorMapId], it->uv));
    }
    if (mat.metallicMapId > NULL_TEXTURE) {
        mat.metallic = linear_sample(&s->textures[mat.metallicMapId], it->uv).x;
    }
    if (mat.roughnessMapId > NULL_TEXTURE) {
        mat.roughness = linear_sample(&s->textures[mat.roughnessMapId], it->uv).x;
    }
    if (mat.normalMapId != NULL_TEXTURE) {
        v3 mapped = linear_sample(&s->textures[mat.normalMapId], it->uv);
        v3 localNorm = normalize3(adds(scl(mapped, 1.f), -0.5f));
        it->norm = local_to_world(it->norm, localNorm);
    }
    return mat;
}

/* radiance seen by a ray that leaves the scene: envMap->linearSample(Math::toPlane(dir)) (restir.cu:134-136) */
static inline int scene_has_env(const orc_scene* s) { return s->envMapTexId >= 0; }
static v3 env_radiance(const orc_scene* s, v3 dir) {
    return linear_sample(&s->textures[s->envMapTexId], to_plane(dir));
}

/* sampler.h:203-207 on the environment-map table */
static inline int env_sampler_sample(const orc_scene* s, float r1, float r2) {
    int passId = i_min(f2i((float)s->envMapSamplerLength * r1), s->envMapSamplerLength - 1);
    return (r2 < s->envMapProb[passId]) ? passId : s->envMapFailId[passId];
}

/* scene.h:364-376 (NoVisibility) / :378-392.  `PiInv * PiInv * .5f` is the macro quirk Q10:
 * x * 1.f / Pi * 1.f / Pi * .5f evaluated left to right. */
static float sample_environment_map_nv(const orc_scene* s, v2 r, v3* radiance, v3* wi) {
    const orc_texture* env = &s->textures[s->envMapTexId];
    int pixId = env_sampler_sample(s, r.x, r.y);
    int y = pixId / env->width;
    int x = pixId - y * env->width;
    *radiance = ld3(env->data + (size_t)pixId * 3);
    v2 uv = { (.5f + (float)x) / (float)env->width, (.5f + (float)y) / (float)env->height };
    *wi = to_sphere(uv);
    return ((((luminance(*radiance) * s->sumLightPowerInv * (float)env->width * (float)env->height * 1.f) / PI_F) * 1.f) / PI_F) * .5f;
}

/* sampler.h:203-207 */
static inline int light_sampler_sample(const orc_scene* s, float r1, float r2) {
    int passId = i_min(f2i((float)s->numLights * r1), s->numLights - 1);
    return (r2 < s->lightProb[passId]) ? passId : s->lightFailId[passId];
}

/* scene.h:394-425 */
static float sample_direct_light_nv(const orc_scene* s, v3 pos, v4 r, v3* radiance, v3* wi, float* dist) {
    if (s->numLights == 0) {
        return INVALID_PDF;
    }
    int lightId = light_sampler_sample(s, r.x, r.y);
    if (lightId == s->numLights - 1 && s->envMapSamplerLength != 0) {     /* :400-403 */
        *dist = 1e10f;
        v2 r2 = { r.z, r.w };
        return sample_environment_map_nv(s, r2, radiance, wi);
    }
    int primId = s->lightPrimIds[lightId];
    v3 v0, v1, v2_;
    tri_verts(s, primId, &v0, &v1, &v2_);
    v3 sampled = sample_triangle_uniform(v0, v1, v2_, r.z, r.w);

    v3 normal = triangle_normal(v0, v1, v2_);
    v3 posToSampled = sub(sampled, pos);

    if (dot(normal, posToSampled) > -1e-6f) {        /* SCENE_LIGHT_SINGLE_SIDED (common.h:6) */
        return INVALID_PDF;
    }
    float area = triangle_area(v0, v1, v2_);
    *radiance = ld3(s->lightUnitRadiance + (size_t)lightId * 3);
    *wi = normalize3(posToSampled);
    *dist = length3(posToSampled);
    float power = luminance(*radiance) / (area * 2.f * GLM_PI_F);
    return pdf_area_to_solid_angle(power * s->sumLightPowerInv, pos, sampled, normal);
}

/* scene.h:427-459 */
static float sample_direct_light(const orc_scene* s, v3 pos, v4 r, v3* radiance, v3* wi, int* walks) {
    if (s->numLights == 0) {
        return INVALID_PDF;
    }
    int lightId = light_sampler_sample(s, r.x, r.y);
    if (lightId == s->numLights - 1 && s->envMapSamplerLength != 0) {     /* :433-435 -> sampleEnvironmentMap :378-392 */
        v2 r2 = { r.z, r.w };
        float pdf = sample_environment_map_nv(s, r2, radiance, wi);
        (*walks)++;
        if (scene_test_occlusion(s, pos, add(pos, scl(*wi, 1e6f)))) {
            return INVALID_PDF;
        }
        return pdf;
    }
    int primId = s->lightPrimIds[lightId];
    v3 v0, v1, v2_;
    tri_verts(s, primId, &v0, &v1, &v2_);
    v3 sampled = sample_triangle_uniform(v0, v1, v2_, r.z, r.w);

    (*walks)++;
    if (scene_test_occlusion(s, pos, sampled)) {
        return INVALID_PDF;
    }
    v3 normal = triangle_normal(v0, v1, v2_);
    v3 posToSampled = sub(sampled, pos);
    if (dot(normal, posToSampled) > -1e-6f) {
        return INVALID_PDF;
    }
    float area = triangle_area(v0, v1, v2_);
    *radiance = ld3(s->lightUnitRadiance + (size_t)lightId * 3);
    *wi = normalize3(posToSampled);
    float power = luminance(*radiance) / (area * 2.f * GLM_PI_F);
    return pdf_area_to_solid_angle(power * s->sumLightPowerInv, pos, sampled, normal);
}

/* ------------------------------------------------------------------------------------------
 * Reservoir<DirectLiSample> (restir.h:29-117)
 * ---------------------------------------------------------------------------------------- */
typedef struct { v3 Li, wi; float dist; } li_sample_t;
typedef struct { li_sample_t sample; int numSamples; float weight; } resv_t;

static inline resv_t resv_default(void) { resv_t r; memset(&r, 0, sizeof r); return r; }
static inline resv_t resv_load(const orc_reservoir* p) {
    resv_t r;
    r.sample.Li = ld3(p->Li); r.sample.wi = ld3(p->wi); r.sample.dist = p->dist;
    r.numSamples = p->numSamples; r.weight = p->weight;
    return r;
}
static inline void resv_store(orc_reservoir* p, const resv_t* r) {
    st3(p->Li, r->sample.Li); st3(p->wi, r->sample.wi); p->dist = r->sample.dist;
    p->numSamples = r->numSamples; p->weight = r->weight;
}
static inline void resv_update(resv_t* r, const li_sample_t* ns, float newWeight, float rnd) { /* :38-44 */
    r->weight += newWeight;
    r->numSamples++;
    if (rnd * r->weight < newWeight) {
        r->sample = *ns;
    }
}
static inline int resv_invalid(const resv_t* r) {             /* :51-53 */
    return is_nan_or_inf(r->weight) || r->weight < 0.f;
}
static inline void resv_check_validity(resv_t* r) {           /* :55-59, clear :46-49 */
    if (resv_invalid(r)) {
        r->weight = 0.f;
        r->numSamples = 0;
    }
}
static inline void resv_merge(resv_t* r, const resv_t* rhs, float rnd) { /* :61-68 */
    r->weight += rhs->weight;
    r->numSamples += rhs->numSamples;
    if (rnd * r->weight < rhs->weight) {
        r->sample = rhs->sample;
    }
}
static inline void resv_clamp(resv_t* r, int val) {            /* :88-93 */
    if (r->numSamples > val) {
        r->weight *= (float)val / (float)r->numSamples;
        r->numSamples = val;
    }
}
static inline void resv_pre_clamped_merge(resv_t* r, resv_t rhs, int M, float rnd) { /* :95-102 */
    if (r->numSamples > 0) {
        resv_clamp(&rhs, (M - 1) * r->numSamples);
    }
    resv_merge(r, &rhs, rnd);
}

/* ------------------------------------------------------------------------------------------
 * G-buffer (gbuffer.cu:3-78)
 * ---------------------------------------------------------------------------------------- */
void orc_gbuffer_render(const orc_scene* s, const orc_camera* cam, orc_gbuffer* g, int y0, int y1) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    float* normal = g->normal[g->frameIdx];
    int* primIdPlane = g->primId[g->frameIdx];
    float* depth = g->depth[g->frameIdx];
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            int idx = y * W + x;
            float aspect = (float)W / (float)H;
            float tanFovY = tanf(g_radians(cam->fov[1]));
            v2 pixelSize = { 1.f / (float)W, 1.f / (float)H };
            v2 scr = { (float)x * pixelSize.x, (float)y * pixelSize.y };
            v2 ruv = { scr.x + pixelSize.x * .5f, scr.y + pixelSize.y * .5f };

            v3 pLens = v3s(0.f);
            v3 pFocusPlane = scl(V3((1.f - ruv.x * 2.f) * aspect * tanFovY,
                                    (1.f - ruv.y * 2.f) * 1.f * tanFovY, 1.f), cam->focalDist);
            v3 dir = sub(pFocusPlane, pLens);

            ray_t ray;
            ray.direction = normalize3(m3mul(ld3(cam->right), ld3(cam->up), ld3(cam->view), dir));
            ray.origin = add(add(ld3(cam->position), scl(ld3(cam->right), pLens.x)), scl(ld3(cam->up), pLens.y));

            isect_t it;
            scene_intersect(s, ray, &it);

            if (it.primId != NULL_PRIM) {
                int matId = it.matId;
                if (s->materials[it.matId].type == MAT_LIGHT) {
                    matId = NULL_PRIM - 1;
                    /* gbuffer.cu:32-36 only rewrites intersec.primId, which is not read again */
                }
                orc_material material = textured_material_and_surface(s, &it);   /* scene.h:78-99 */
                st3(g->albedo + (size_t)idx * 3, ld3(material.baseColor));
                st3(normal + (size_t)idx * 3, it.norm);
                primIdPlane[idx] = matId;
                depth[idx] = length3(sub(ray.origin, it.pos));      /* glm::distance(pos, origin) */

                int lx, ly;
                camera_raster_coord(&g->lastCamera, it.pos, &lx, &ly);
                if (lx >= 0 && lx < g->width && ly >= 0 && ly < g->height) {
                    g->motion[idx] = ly * W + lx;
                }
                else {
                    g->motion[idx] = -1;
                }
            }
            else {
                v3 albedo = v3s(0.f);
                if (scene_has_env(s)) {                                  /* gbuffer.cu:59-62 */
                    albedo = env_radiance(s, ray.direction);
                }
                st3(g->albedo + (size_t)idx * 3, albedo);
                st3(normal + (size_t)idx * 3, v3s(0.f));
                primIdPlane[idx] = NULL_PRIM;
                depth[idx] = 1.f;
                g->motion[idx] = 0;
            }
        }
    }
}

void orc_gbuffer_update(orc_gbuffer* g, const orc_camera* cam) {   /* gbuffer.cu:75-78 */
    g->lastCamera = *cam;
    g->frameIdx ^= 1;
}

/* ------------------------------------------------------------------------------------------
 * PTDirectKernel (pathtrace.cu:279-328)
 * ---------------------------------------------------------------------------------------- */
void orc_pt_direct(const orc_scene* s, const orc_camera* cam, float* directIllum,
                   int looper, int iter, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            v3 direct = v3s(0.f);
            int index = y * W + x;
            int walks = 0;
            rng_t rng = make_seeded_random_engine(looper, index, 0, s->sampleSequence);
            v4 r4 = sample4D(&rng);
            ray_t ray = camera_sample(cam, x, y, r4);
            isect_t it;
            scene_intersect(s, ray, &it);
            walks++;

            if (it.primId == NULL_PRIM) {
                if (scene_has_env(s)) {                                  /* pathtrace.cu:295-297 */
                    direct = env_radiance(s, ray.direction);
                }
            }
            else {
                orc_material mat = textured_material_and_surface(s, &it); /* :301 */
                const orc_material* material = &mat;
                if (material->type == MAT_LIGHT) {
                    direct = ld3(material->baseColor);
                }
                else {
                    it.wo = neg(ray.direction);
                    int deltaBSDF = (material->type == MAT_DIELECTRIC);
                    if (!deltaBSDF && dot(it.norm, it.wo) < 0.f) {
                        it.norm = neg(it.norm);
                    }
                    if (!deltaBSDF) {
                        v3 Li = v3s(0.f), wi = v3s(0.f);
                        v4 rl = sample4D(&rng);
                        float lightPdf = sample_direct_light(s, it.pos, rl, &Li, &wi, &walks);
                        if (lightPdf > 0.f) {
                            v3 f = material_bsdf(material, it.norm, it.wo, wi);
                            direct = dvs(scl(mul(Li, f), sat_dot(it.norm, wi)), lightPdf);
                        }
                    }
                }
            }
            float* o = directIllum + (size_t)index * 3;
            v3 prev = ld3(o);
            st3(o, dvs(add(scl(prev, (float)iter), direct), (float)(iter + 1)));
            total += (unsigned long long)walks;
        }
    }
    if (rays) *rays = total;
}

/* ------------------------------------------------------------------------------------------
 * Multi-bounce path tracing: singleKernelPT / PTIndirectKernel (pathtrace.cu:156-277,330-432)
 * ---------------------------------------------------------------------------------------- */
/* scene.h:358-362 */
static float environment_map_pdf(const orc_scene* s, v3 w) {
    const orc_texture* env = &s->textures[s->envMapTexId];
    v3 radiance = linear_sample(env, to_plane(w));
    return luminance(radiance) * s->sumLightPowerInv * (float)env->width * (float)env->height * .5f;
}
/* scene.h:121-126 */
static float primitive_area(const orc_scene* s, int primId) {
    v3 v0, v1, v2_;
    tri_verts(s, primId, &v0, &v1, &v2_);
    return length3(cross(sub(v1, v0), sub(v2_, v0))) * .5f;
}
static inline v3 sample3D(rng_t* r) { v3 o; o.x = sample1D(r); o.y = sample1D(r); o.z = sample1D(r); return o; }   /* sampler.h:55-57 */

/* One step that the three kernels share: next-event estimation at the current vertex
 * (pathtrace.cu:203-213 / 365-376, restir.cu:291-302).  Returns the contribution (zero if none). */
static v3 nee_contribution(const orc_scene* s, const orc_material* material, const isect_t* it, v3 throughput, rng_t* rng, int* walks) {
    v3 radiance = v3s(0.f), wi = v3s(0.f);
    v4 r = sample4D(rng);
    float lightPdf = sample_direct_light(s, it->pos, r, &radiance, &wi, walks);
    if (lightPdf > 0.f) {
        float BSDFPdf = material_pdf(material, it->norm, it->wo, wi);
        /* throughput * BSDF * radiance * satDot / lightPdf * powerHeuristic, left to right */
        v3 c = mul(mul(throughput, material_bsdf(material, it->norm, it->wo, wi)), radiance);
        c = scl(c, sat_dot(it->norm, wi));
        c = dvs(c, lightPdf);
        return scl(c, power_heuristic(lightPdf, BSDFPdf));
    }
    return v3s(0.f);
}

/* What happens when the continuation ray has been traced (pathtrace.cu:234-269 / 395-424, restir.cu:333-367):
 * returns 1 if the path ends here; *add receives the radiance picked up (environment map or an emitter). */
static int path_end_contribution(const orc_scene* s, isect_t* it, ray_t ray, v3 curPos, v3 throughput, const bsdf_sample_t* sample,
                                 int deltaSample, int firstBounceUnweighted, orc_material* material, v3* add_, int* hitLight) {
    *add_ = v3s(0.f);
    *hitLight = 0;
    if (it->primId == NULL_PRIM) {
        if (scene_has_env(s)) {
            v3 radiance = mul(env_radiance(s, ray.direction), throughput);
            float weight = deltaSample ? 1.f : power_heuristic(sample->pdf, environment_map_pdf(s, ray.direction));
            *add_ = scl(radiance, weight);
        }
        return 1;
    }
    *material = textured_material_and_surface(s, it);
    if (material->type == MAT_LIGHT) {
        if (dot(it->norm, ray.direction) < 0.f) {      /* SCENE_LIGHT_SINGLE_SIDED */
            return 1;
        }
        v3 radiance = ld3(material->baseColor);
        float weight = (deltaSample || firstBounceUnweighted) ? 1.f : power_heuristic(sample->pdf,
            pdf_area_to_solid_angle(luminance(radiance) * s->sumLightPowerInv * primitive_area(s, it->primId), curPos, it->pos, it->norm));
        *add_ = scl(mul(radiance, throughput), weight);
        *hitLight = 1;
        return 1;
    }
    return 0;
}

/* singleKernelPT (pathtrace.cu:156-277); DENOISER_DEMODULATE is true: the primary material's baseColor is 1 */
void orc_path_trace(const orc_scene* s, const orc_camera* cam, float* directIllum, float* indirectIllum,
                    int looper, int iter, int maxDepth, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            v3 direct = v3s(0.f), indirect = v3s(0.f);
            int index = y * W + x;
            int walks = 0;
            rng_t rng = make_seeded_random_engine(looper, index, 0, s->sampleSequence);
            ray_t ray = camera_sample(cam, x, y, sample4D(&rng));
            isect_t it;
            scene_intersect(s, ray, &it);
            walks++;
            if (it.primId == NULL_PRIM) {
                direct = v3s(1.f);
            }
            else {
                orc_material material = textured_material_and_surface(s, &it);
                st3(material.baseColor, v3s(1.f));
                if (material.type == MAT_LIGHT) {
                    direct = v3s(1.f);
                }
                else {
                    v3 throughput = v3s(1.f);
                    it.wo = neg(ray.direction);
                    for (int depth = 1; depth <= maxDepth; depth++) {
                        int deltaBSDF = (material.type == MAT_DIELECTRIC);
                        if (material.type != MAT_DIELECTRIC && dot(it.norm, it.wo) < 0.f) it.norm = neg(it.norm);
                        if (!deltaBSDF) {
                            v3 c = nee_contribution(s, &material, &it, throughput, &rng, &walks);
                            if (depth == 1) direct = add(direct, c); else indirect = add(indirect, c);
                        }
                        bsdf_sample_t sample;
                        material_sample(&material, it.norm, it.wo, sample3D(&rng), &sample);
                        if (sample.type == BS_INVALID) break;
                        else if (sample.pdf < 1e-8f) break;
                        int deltaSample = (sample.type & BS_SPECULAR) != 0;
                        throughput = mul(throughput, scl(dvs(sample.bsdf, sample.pdf), deltaSample ? 1.f : abs_dot(it.norm, sample.dir)));
                        ray = make_offseted_ray(it.pos, sample.dir);
                        v3 curPos = it.pos;
                        scene_intersect(s, ray, &it);
                        walks++;
                        it.wo = neg(ray.direction);
                        v3 c; int hitLight;
                        if (path_end_contribution(s, &it, ray, curPos, throughput, &sample, deltaSample, 0, &material, &c, &hitLight)) {
                            indirect = add(indirect, c);
                            break;
                        }
                    }
                }
            }
            if (has_nan_or_inf(direct)) direct = v3s(0.f);
            if (has_nan_or_inf(indirect)) indirect = v3s(0.f);
            direct = hdr_to_ldr(direct);
            indirect = hdr_to_ldr(indirect);
            float* od = directIllum + (size_t)index * 3; float* oi = indirectIllum + (size_t)index * 3;
            st3(od, dvs(add(scl(ld3(od), (float)iter), direct), (float)(iter + 1)));
            st3(oi, dvs(add(scl(ld3(oi), (float)iter), indirect), (float)(iter + 1)));
            total += (unsigned long long)walks;
        }
    }
    if (rays) *rays = total;
}

/* PTIndirectKernel (pathtrace.cu:330-432) */
void orc_pt_indirect(const orc_scene* s, const orc_camera* cam, float* indirectIllum,
                     int looper, int iter, int maxDepth, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            v3 indirect = v3s(0.f);
            int index = y * W + x;
            int walks = 0;
            rng_t rng = make_seeded_random_engine(looper, index, 0, s->sampleSequence);
            ray_t ray = camera_sample(cam, x, y, sample4D(&rng));
            isect_t it;
            scene_intersect(s, ray, &it);
            walks++;
            if (it.primId != NULL_PRIM) {
                orc_material material = textured_material_and_surface(s, &it);
                if (material.type != MAT_LIGHT) {
                    v3 throughput = v3s(1.f);
                    it.wo = neg(ray.direction);
                    for (int depth = 1; depth <= maxDepth; depth++) {
                        int deltaBSDF = (material.type == MAT_DIELECTRIC);
                        if (material.type != MAT_DIELECTRIC && dot(it.norm, it.wo) < 0.f) it.norm = neg(it.norm);
                        if (!deltaBSDF && depth > 1) {
                            indirect = add(indirect, nee_contribution(s, &material, &it, throughput, &rng, &walks));
                        }
                        bsdf_sample_t sample;
                        material_sample(&material, it.norm, it.wo, sample3D(&rng), &sample);
                        if (sample.type == BS_INVALID) break;
                        else if (sample.pdf < 1e-8f) break;
                        int deltaSample = (sample.type & BS_SPECULAR) != 0;
                        throughput = mul(throughput, scl(dvs(sample.bsdf, sample.pdf), deltaSample ? 1.f : abs_dot(it.norm, sample.dir)));
                        ray = make_offseted_ray(it.pos, sample.dir);
                        v3 curPos = it.pos;
                        scene_intersect(s, ray, &it);
                        walks++;
                        it.wo = neg(ray.direction);
                        v3 c; int hitLight;
                        if (path_end_contribution(s, &it, ray, curPos, throughput, &sample, deltaSample, 0, &material, &c, &hitLight)) {
                            indirect = add(indirect, c);
                            break;
                        }
                    }
                }
            }
            if (has_nan_or_inf(indirect)) indirect = v3s(0.f);
            float* oi = indirectIllum + (size_t)index * 3;
            st3(oi, dvs(add(scl(ld3(oi), (float)iter), indirect), (float)(iter + 1)));
            total += (unsigned long long)walks;
        }
    }
    if (rays) *rays = total;
}

/* ------------------------------------------------------------------------------------------
 * ReSTIRDirectKernel (restir.cu:20-100,111-231), two-phase contract
 * ---------------------------------------------------------------------------------------- */
/* restir.cu:20-45 */
static int temporal_neighbor_index(int idx, const orc_gbuffer* g);
static resv_t find_temporal_neighbor(const orc_reservoir* reservoir, int idx, const orc_gbuffer* g) {
    int lastIdx = temporal_neighbor_index(idx, g);
    return lastIdx < 0 ? resv_default() : resv_load(&reservoir[lastIdx]);
}
/* the neighbour test of findTemporalNeighbor (restir.cu:20-45): the index to reuse, or -1 for `T()` */
static int temporal_neighbor_index(int idx, const orc_gbuffer* g) {
    const int cur = g->frameIdx, last = g->frameIdx ^ 1;
    int primId = g->primId[cur][idx];
    int lastIdx = g->motion[idx];
    int diff = 0;

    if (lastIdx < 0) {
        diff = 1;
    }
    else if (primId <= NULL_PRIM) {
        diff = 1;
    }
    else if (g->primId[last][lastIdx] != primId) {
        diff = 1;
    }
    else {
        v3 norm = ld3(g->normal[cur] + (size_t)idx * 3);
        v3 lastNorm = ld3(g->normal[last] + (size_t)lastIdx * 3);
        float depth = g->depth[cur][idx];
        float pdepth = g->depth[last][lastIdx];
        if (abs_dot(norm, lastNorm) < .9f || g_abs(pdepth - depth) > depth * .1f) {
            diff = 1;
        }
    }
    return diff ? -1 : lastIdx;
}

/* restir.cu:47-85 */
static resv_t find_spatial_neighbor_disk(const orc_reservoir* reservoir, int x, int y,
                                         const orc_gbuffer* g, v2 r) {
    const float Radius = 5.f;
    const int cur = g->frameIdx;
    int idx = y * g->width + x;

    v2 p = to_concentric_disk_m(r.x, r.y);         /* cos / sin follow the libm mode (orc_set_libm_mode) */
    p.x *= Radius; p.y *= Radius;
    int px = f2i((float)x + .5f + p.x);
    int py = f2i((float)y + .5f + p.y);
    int pidx = py * g->width + px;
    int diff = 0;

    if (px < 0 || px >= g->width || py < 0 || py >= g->height || (px == x && py == y)) {
        diff = 1;
    }
    else if (g->primId[cur][pidx] != g->primId[cur][idx]) {
        diff = 1;
    }
    else {
        v3 norm = ld3(g->normal[cur] + (size_t)idx * 3);
        v3 pnorm = ld3(g->normal[cur] + (size_t)pidx * 3);
        if (dot(norm, pnorm) < .9f) {
            diff = 1;
        }
        float depth = g->depth[cur][idx];
        float pdepth = g->depth[cur][pidx];
        if (g_abs(depth - pdepth) > depth * .1f) {
            diff = 1;
        }
    }
    return diff ? resv_default() : resv_load(&reservoir[pidx]);
}

typedef struct {
    int   kind;          /* 0 = early exit (miss / light), 1 = shaded */
    v3    direct;        /* early-exit radiance */
    rng_t rng;
    resv_t reservoir;    /* post-temporal (validity-checked when spatial reuse is on) */
    v3    norm, wo;
    orc_material material;   /* with its maps applied and baseColor = 1 (restir.cu:140-141) */
} pixel_state_t;

#define RESERVOIR_SIZE 32   /* restir.cu:3 */

void* orc_restir_state_create(int width, int height) {
    return calloc((size_t)width * height, sizeof(pixel_state_t));
}
void orc_restir_state_destroy(void* st) { free(st); }

/* ---- phase A: restir.cu:119-194 (everything before the barrier) + :211-212, rows [y0,y1) ---- */
void orc_restir_phase_a(void* state, const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                        orc_reservoir* reservoirOut, const orc_reservoir* reservoirIn,
                        orc_reservoir* reservoirTemp, int looper, int first, int reuse,
                        int y0, int y1, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    pixel_state_t* st = (pixel_state_t*)state;
    unsigned long long total = 0;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            int index = y * W + x;
            pixel_state_t* ps = &st[index];
            ps->kind = 0;
            ps->direct = v3s(0.f);

            rng_t rng = make_seeded_random_engine(looper, index, 0, s->sampleSequence);
            v4 r4 = sample4D(&rng);
            ray_t ray = camera_sample(cam, x, y, r4);
            isect_t it;
            scene_intersect(s, ray, &it);
            total++;

            if (it.primId == NULL_PRIM) {
                if (scene_has_env(s)) {         /* restir.cu:134-136 */
                    ps->direct = env_radiance(s, ray.direction);
                }
                continue;
            }
            orc_material material = textured_material_and_surface(s, &it);   /* restir.cu:140 */
            st3(material.baseColor, v3s(1.f));  /* restir.cu:141 */

            if (material.type == MAT_LIGHT) {
                ps->direct = ld3(material.baseColor);
                continue;
            }
            it.wo = neg(ray.direction);
            int deltaBSDF = (material.type == MAT_DIELECTRIC);
            if (!deltaBSDF && dot(it.norm, it.wo) < 0.f) {
                it.norm = neg(it.norm);
            }

            resv_t reservoir = resv_default();
            for (int i = 0; i < RESERVOIR_SIZE; i++) {
                li_sample_t cand;
                cand.Li = v3s(0.f); cand.wi = v3s(0.f); cand.dist = 0.f;  /* dist: indeterminate in the reference when p<=0; never selected */
                v4 rl = sample4D(&rng);
                float p = sample_direct_light_nv(s, it.pos, rl, &cand.Li, &cand.wi, &cand.dist);
                v3 gg = scl(mul(cand.Li, material_bsdf(&material, it.norm, it.wo, cand.wi)), sat_dot(it.norm, cand.wi));
                float weight = luminance(dvs(gg, p));
                if (is_nan_or_inf(weight) || p <= 0.f) {
                    weight = 0.f;
                }
                float ru = sample1D(&rng);
                resv_update(&reservoir, &cand, weight, ru);
            }
            li_sample_t sample = reservoir.sample;

            total++;
            if (scene_test_occlusion(s, it.pos, add(it.pos, scl(sample.wi, sample.dist)))) {
                reservoir.weight = 0.f;
            }

            if (!first && (reuse & 1)) {
                resv_t temporal = find_temporal_neighbor(reservoirIn, index, g);
                if (!resv_invalid(&temporal)) {
                    float ru = sample1D(&rng);
                    resv_pre_clamped_merge(&reservoir, temporal, 20, ru);
                }
            }

            resv_t tempReservoir = reservoir;
            if (reuse & 2) {
                resv_check_validity(&reservoir);
                resv_store(&reservoirTemp[index], &reservoir);
            }
            resv_check_validity(&tempReservoir);
            resv_store(&reservoirOut[index], &tempReservoir);

            ps->kind = 1;
            ps->rng = rng;
            ps->reservoir = reservoir;
            ps->norm = it.norm;
            ps->wo = it.wo;
            ps->material = material;
        }
    }
    if (rays) *rays = total;
}

/* ---- phase B: restir.cu:196-230, rows [y0,y1) ---- */
void orc_restir_phase_b(void* state, const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                        float* directIllum, const orc_reservoir* reservoirTemp, int iter, int reuse,
                        int y0, int y1) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    pixel_state_t* st = (pixel_state_t*)state;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            int index = y * W + x;
            pixel_state_t* ps = &st[index];
            v3 direct = ps->direct;

            if (ps->kind == 1) {
                rng_t rng = ps->rng;
                resv_t reservoir = ps->reservoir;
                orc_material material = ps->material;

                if (reuse & 2) {
                    resv_t agg = resv_default();           /* mergeSpatialNeighborDirect :87-100 */
                    for (int i = 0; i < 5; i++) {
                        v2 r2 = sample2D(&rng);
                        resv_t spatial = find_spatial_neighbor_disk(reservoirTemp, x, y, g, r2);
                        if (!resv_invalid(&spatial)) {
                            float ru = sample1D(&rng);
                            resv_merge(&agg, &spatial, ru);
                        }
                    }
                    if (!resv_invalid(&agg) && !resv_invalid(&reservoir)) {
                        float ru = sample1D(&rng);
                        resv_merge(&reservoir, &agg, ru);
                    }
                }

                li_sample_t sample = reservoir.sample;
                direct = v3s(0.f);
                if (!resv_invalid(&reservoir)) {
                    v3 LiBSDF = mul(sample.Li, material_bsdf(&material, ps->norm, ps->wo, sample.wi));
                    direct = dvs(scl(dvs(LiBSDF, luminance(LiBSDF)), reservoir.weight), (float)reservoir.numSamples);
                }
                if (has_nan_or_inf(direct)) {
                    direct = v3s(0.f);
                }
            }
            direct = mul(direct, ld3(g->albedo + (size_t)index * 3));
            float* o = directIllum + (size_t)index * 3;
            st3(o, dvs(add(scl(ld3(o), (float)iter), direct), (float)(iter + 1)));
        }
    }
}

void orc_restir_direct(const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                       float* directIllum, orc_reservoir* reservoirOut,
                       const orc_reservoir* reservoirIn, orc_reservoir* reservoirTemp,
                       int looper, int iter, int first, int reuse, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    void* st = orc_restir_state_create(W, H);
    orc_restir_phase_a(st, s, cam, g, reservoirOut, reservoirIn, reservoirTemp, looper, first, reuse, 0, H, rays);
    orc_restir_phase_b(st, s, cam, g, directIllum, reservoirTemp, iter, reuse, 0, H);
    orc_restir_state_destroy(st);
}

/* ------------------------------------------------------------------------------------------
 * ReSTIRIndirectKernel (restir.cu:233-416): one path per pixel into a Reservoir<IndirectLiSample>, temporal reuse
 * ---------------------------------------------------------------------------------------- */
/* IndirectLiSample members the reference leaves uninitialised (xv, nv, xs, ns; restir.h:13-27) start at 0 here. */
static inline orc_indirect_reservoir ires_default(void) { orc_indirect_reservoir r; memset(&r, 0, sizeof r); return r; }
static inline int ires_invalid(const orc_indirect_reservoir* r) { return is_nan_or_inf(r->weight) || r->weight < 0.f; }

void orc_restir_indirect(const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g, float* indirectIllum,
                         orc_indirect_reservoir* temporalReservoir, const orc_indirect_reservoir* lastTemporalReservoir,
                         int looper, int iter, int maxDepth, int first, int reuse, unsigned long long* rays) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            orc_indirect_reservoir smp = ires_default();     /* indirectSample lives in .Lo/.xv/.nv/.xs/.ns of this record */
            int index = y * W + x;
            int walks = 0;
            rng_t rng = make_seeded_random_engine(looper, index, 0, s->sampleSequence);
            ray_t ray = camera_sample(cam, x, y, sample4D(&rng));
            isect_t it;
            scene_intersect(s, ray, &it);
            walks++;
            float primSamplePdf = 0.f;
            int primSampleDelta = 0;
            v3 primWo = neg(ray.direction);
            orc_material primMaterial; memset(&primMaterial, 0, sizeof primMaterial);
            v3 Lo = v3s(0.f);

            if (it.primId != NULL_PRIM) {
                orc_material material = textured_material_and_surface(s, &it);
                if (material.type != MAT_LIGHT) {
                    v3 throughput = v3s(1.f);
                    it.wo = neg(ray.direction);
                    primMaterial = material;
                    for (int depth = 1; depth <= maxDepth; depth++) {
                        int deltaBSDF = (material.type == MAT_DIELECTRIC);
                        if (material.type != MAT_DIELECTRIC && dot(it.norm, it.wo) < 0.f) it.norm = neg(it.norm);
                        if (!deltaBSDF && depth > 1) {
                            Lo = add(Lo, nee_contribution(s, &material, &it, throughput, &rng, &walks));
                        }
                        bsdf_sample_t sample;
                        material_sample(&material, it.norm, it.wo, sample3D(&rng), &sample);
                        if (sample.type == BS_INVALID) break;
                        else if (sample.pdf < 1e-8f) break;
                        int deltaSample = (sample.type & BS_SPECULAR) != 0;
                        if (depth > 1) {
                            throughput = mul(throughput, scl(dvs(sample.bsdf, sample.pdf), deltaSample ? 1.f : abs_dot(it.norm, sample.dir)));
                        }
                        else {
                            primSamplePdf = sample.pdf;
                            primSampleDelta = deltaSample;
                            st3(smp.xv, it.pos);
                            st3(smp.nv, it.norm);
                        }
                        ray = make_offseted_ray(it.pos, sample.dir);
                        v3 curPos = it.pos;
                        scene_intersect(s, ray, &it);
                        walks++;
                        it.wo = neg(ray.direction);
                        v3 c; int hitLight;
                        if (path_end_contribution(s, &it, ray, curPos, throughput, &sample, deltaSample, depth == 1, &material, &c, &hitLight)) {
                            Lo = add(Lo, c);
                            if (hitLight && depth == 1) { st3(smp.xs, it.pos); st3(smp.ns, it.norm); }     /* :360-363 */
                            break;
                        }
                        if (depth == 1) { st3(smp.xs, it.pos); st3(smp.ns, it.norm); }                     /* :367-370 */
                    }
                }
            }
            st3(smp.Lo, Lo);
            /* WriteSample (:372-416) */
            orc_indirect_reservoir reservoir = ires_default();
            float sampleWeight = 0.f;
            if (!(luminance(Lo) < 1e-8f)) {                                   /* !indirectSample.invalid() */
                sampleWeight = luminance(dvs(Lo, primSamplePdf));             /* toScalar(pHatIndirect(...) / primSamplePdf), pHat = Lo (:234) */
                if (isnan(sampleWeight) || sampleWeight < 0.f) sampleWeight = 0.f;
            }
            {   /* reservoir.update(indirectSample, sampleWeight, sample1D(rng)) */
                float r = sample1D(&rng);
                reservoir.weight += sampleWeight;
                reservoir.numSamples++;
                if (r * reservoir.weight < sampleWeight) { memcpy(&reservoir, &smp, 15 * sizeof(float)); }
            }
            if (!first && (reuse & 1)) {
                int lastIdx = temporal_neighbor_index(index, g);
                orc_indirect_reservoir temp = lastIdx < 0 ? ires_default() : lastTemporalReservoir[lastIdx];
                if (!ires_invalid(&temp)) {
                    float r = sample1D(&rng);                                  /* merge (restir.h:61-68) */
                    reservoir.weight += temp.weight;
                    reservoir.numSamples += temp.numSamples;
                    if (r * reservoir.weight < temp.weight) { memcpy(&reservoir, &temp, 15 * sizeof(float)); }
                }
            }
            v3 indirect = v3s(0.f);
            orc_indirect_reservoir picked = reservoir;                         /* IndirectLiSample sample = reservoir.sample */
            if (reservoir.numSamples > 20) {                                   /* clamp<20>() (restir.h:79-86) */
                reservoir.weight *= (float)20 / (float)reservoir.numSamples;
                reservoir.numSamples = 20;
            }
            if (!ires_invalid(&reservoir)) {
                v3 primWi = normalize3(sub(ld3(picked.xs), ld3(picked.xv)));
                v3 rl = ld3(reservoir.Lo);
                indirect = dvs(scl(dvs(rl, luminance(rl)), reservoir.weight), (float)reservoir.numSamples);
                indirect = mul(indirect, scl(material_bsdf(&primMaterial, ld3(picked.nv), primWo, primWi),
                                             primSampleDelta ? 1.f : sat_dot(ld3(picked.nv), primWi)));
            }
            if (has_nan_or_inf(indirect)) indirect = v3s(0.f);
            temporalReservoir[index] = reservoir;
            float* oi = indirectIllum + (size_t)index * 3;
            st3(oi, dvs(add(scl(ld3(oi), (float)iter), indirect), (float)(iter + 1)));
            total += (unsigned long long)walks;
        }
    }
    if (rays) *rays = total;
}

/* ------------------------------------------------------------------------------------------
 * sendImageToPBO (pathtrace.cu:30-56), tone-map operators (mathUtil.h:102-117)
 * ---------------------------------------------------------------------------------------- */
static inline float calc_filmic(float c) {
    return (c * (c * 0.22f + 0.03f) + 0.002f) / (c * (c * 0.22f + 0.3f) + 0.06f) - 1.f / 30.f;
}
static inline v3 tonemap(v3 color, int mode) {
    if (mode == 1) {        /* filmic */
        v3 c = scl(color, 1.6f);
        float d = calc_filmic(11.2f);
        color = V3(calc_filmic(c.x) / d, calc_filmic(c.y) / d, calc_filmic(c.z) / d);
    }
    else if (mode == 2) {   /* ACES */
        v3 a = mul(color, adds(scl(color, 2.51f), 0.03f));
        v3 b = adds(mul(color, adds(scl(color, 2.43f), 0.59f)), 0.14f);
        color = dvv(a, b);
    }
    const float e = 1.f / 2.2f;
    return V3(powf(color.x, e), powf(color.y, e), powf(color.z, e));
}

void orc_tonemap(int n, const float* in, int mode, float* out) {
    for (int i = 0; i < n; i++) st3(out + 3 * i, tonemap(ld3(in + 3 * i), mode));
}

void orc_send_image_to_pbo(int w, int h, const float* image, int toneMapping, float scale,
                           unsigned char* rgba) {
#pragma omp parallel for
    for (int i = 0; i < w * h; i++) {
        v3 color = tonemap(scl(ld3(image + (size_t)i * 3), scale), toneMapping);
        rgba[4 * i + 0] = (unsigned char)i_clamp(f2i(color.x * 255.f), 0, 255);
        rgba[4 * i + 1] = (unsigned char)i_clamp(f2i(color.y * 255.f), 0, 255);
        rgba[4 * i + 2] = (unsigned char)i_clamp(f2i(color.z * 255.f), 0, 255);
        rgba[4 * i + 3] = 0;
    }
}

/* pathtrace.cu:58-106: the vec2 / float / int overloads (kind 0 / 1 / 2): correctGamma only.  The int form turns a
 * pixel index into (idx % width, idx / HEIGHT) / (width, height) -- `/ height` as written in the reference (:100). */
void orc_send_debug_to_pbo(int w, int h, const void* image, int kind, unsigned char* rgba) {
    for (int i = 0; i < w * h; i++) {
        v3 color;
        if (kind == 0) color = V3(((const float*)image)[2 * i], ((const float*)image)[2 * i + 1], 0.f);
        else if (kind == 1) color = v3s(((const float*)image)[i]);
        else {
            int v = ((const int*)image)[i];
            int px = v % w, py = v / h;
            color = V3((float)px / (float)w, (float)py / (float)h, 0.f);
        }
        color = tonemap(color, 0);                      /* mode 0 = Math::correctGamma only */
        rgba[4 * i + 0] = (unsigned char)i_clamp(f2i(color.x * 255.f), 0, 255);
        rgba[4 * i + 1] = (unsigned char)i_clamp(f2i(color.y * 255.f), 0, 255);
        rgba[4 * i + 2] = (unsigned char)i_clamp(f2i(color.z * 255.f), 0, 255);
        rgba[4 * i + 3] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * EAW a-trous (denoiser.cu:18-24,64-134,463-477), modulate / add (denoiser.cu:218-248)
 * ---------------------------------------------------------------------------------------- */
static const float Gaussian5x5[5][5] = {
    { .0030f, .0133f, .0219f, .0133f, .0030f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0219f, .0983f, .1621f, .0983f, .0219f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0030f, .0133f, .0219f, .0133f, .0030f }
};

void orc_eaw_level(const orc_gbuffer* g, const orc_camera* cam, const float* colorIn,
                   float* colorOut, float sigDepth, float sigNormal, float sigLumin, int level) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    const int step = 1 << level;
    const int* primIdPlane = g->primId[g->frameIdx];
    const float* normal = g->normal[g->frameIdx];
    const float* depth = g->depth[g->frameIdx];
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            int idxP = y * W + x;
            int primIdP = primIdPlane[idxP];
            if (primIdP <= NULL_PRIM) {
                st3(colorOut + (size_t)idxP * 3, ld3(colorIn + (size_t)idxP * 3));
                continue;
            }
            v3 normP = ld3(normal + (size_t)idxP * 3);
            v3 colorP = ld3(colorIn + (size_t)idxP * 3);
            v3 posP = camera_get_position(cam, x, y, depth[idxP]);

            v3 sum = v3s(0.f);
            float sumWeight = 0.f;
            for (int i = -2; i <= 2; i++) {
                for (int j = -2; j <= 2; j++) {
                    int qx = x + j * step;
                    int qy = y + i * step;
                    int idxQ = qy * W + qx;
                    if (qx >= W || qy >= H || qx < 0 || qy < 0) {
                        continue;
                    }
                    if (primIdPlane[idxQ] != primIdP) {
                        continue;
                    }
                    v3 normQ = ld3(normal + (size_t)idxQ * 3);
                    v3 colorQ = ld3(colorIn + (size_t)idxQ * 3);
                    v3 posQ = camera_get_position(cam, qx, qy, depth[idxQ]);

                    v3 dc = sub(colorP, colorQ);
                    float distColor2 = dot(dc, dc);
                    float wColor = g_min(1.f, expf(-distColor2 / sigLumin));
                    v3 dn = sub(normP, normQ);
                    float distNorm2 = dot(dn, dn);
                    float wNorm = g_min(1.f, expf(-distNorm2 / sigNormal));
                    v3 dp = sub(posP, posQ);
                    float distPos2 = dot(dp, dp);
                    float wPos = g_min(1.f, expf(-distPos2 / sigDepth));

                    float weight = wColor * wNorm * wPos * Gaussian5x5[i + 2][j + 2];
                    sum = add(sum, scl(colorQ, weight));
                    sumWeight += weight;
                }
            }
            v3 o = (sumWeight == 0.f) ? ld3(colorIn + (size_t)idxP * 3) : dvs(sum, sumWeight);
            st3(colorOut + (size_t)idxP * 3, o);
        }
    }
}

float* orc_eaw_filter(const orc_gbuffer* g, const orc_camera* cam, const float* colorIn,
                      float* out, float* tmp) {
    const float sigLumin = 64.f, sigNormal = .2f, sigDepth = 1.f;   /* denoiser.cu:455 */
    float* a = out; float* b = tmp; float* t;
    orc_eaw_level(g, cam, colorIn, a, sigDepth, sigNormal, sigLumin, 0);
    for (int level = 1; level <= 4; level++) {
        orc_eaw_level(g, cam, a, b, sigDepth, sigNormal, sigLumin, level);
        t = a; a = b; b = t;
    }
    return a;
}

/* ------------------------------------------------------------------------------------------
 * SVGF: SpatioTemporalFilter (denoiser.cu:136-216,250-371,479-568)
 * ---------------------------------------------------------------------------------------- */
static const float Gaussian3x3[3][3] = {          /* denoiser.cu:11-15 */
    { .075f, .124f, .075f },
    { .124f, .204f, .124f },
    { .075f, .124f, .075f }
};

typedef struct {
    int width, height;
    float* accumColor[2]; float* accumMoment[2];
    float* variance; float* tempColor; float* tempVariance; float* filteredVariance;
    float* colorOut;     /* the caller's `devColorOut` of the reference: swapped with the filter's buffers by filter() */
    int firstTime, frameIdx;
} svgf_t;

void* orc_svgf_create(int width, int height) {        /* :479-493 */
    svgf_t* f = (svgf_t*)calloc(1, sizeof(svgf_t));
    size_t n = (size_t)width * height;
    f->width = width; f->height = height;
    for (int i = 0; i < 2; i++) { f->accumColor[i] = (float*)calloc(n * 3, 4); f->accumMoment[i] = (float*)calloc(n * 3, 4); }
    f->variance = (float*)calloc(n, 4); f->tempVariance = (float*)calloc(n, 4); f->filteredVariance = (float*)calloc(n, 4);
    f->tempColor = (float*)calloc(n * 3, 4); f->colorOut = (float*)calloc(n * 3, 4);
    f->firstTime = 1; f->frameIdx = 0;
    return f;
}
void orc_svgf_destroy(void* p) {
    svgf_t* f = (svgf_t*)p;
    if (!f) return;
    for (int i = 0; i < 2; i++) { free(f->accumColor[i]); free(f->accumMoment[i]); }
    free(f->variance); free(f->tempVariance); free(f->filteredVariance); free(f->tempColor); free(f->colorOut);
    free(f);
}
void orc_svgf_next_frame(void* p) { ((svgf_t*)p)->frameIdx ^= 1; }          /* :566-568 */
const float* orc_svgf_variance(void* p) { return ((svgf_t*)p)->variance; }
const float* orc_svgf_accum_color(void* p) { svgf_t* f = (svgf_t*)p; return f->accumColor[f->frameIdx]; }
const float* orc_svgf_accum_moment(void* p) { svgf_t* f = (svgf_t*)p; return f->accumMoment[f->frameIdx]; }

/* temporalAccumulate (:250-305).  lastColor / lastMoment are read at lastIdx before `diff` is looked at in the
 * reference (also at lastIdx = -1); they only matter when !diff. */
static void svgf_temporal(float* colorOut, const float* colorAccIn, float* momentOut, const float* momentAccIn,
                          const float* colorIn, const orc_gbuffer* g, int first) {
    const float Alpha = .2f;
    const int* primIdPlane = g->primId[g->frameIdx];
    const int* lastPrimId = g->primId[g->frameIdx ^ 1];
    const float* normal = g->normal[g->frameIdx];
    const float* lastNormal = g->normal[g->frameIdx ^ 1];
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < g->width * g->height; idx++) {
        int primId = primIdPlane[idx];
        int lastIdx = g->motion[idx];
        int diff = first;
        if (lastIdx < 0) diff = 1;
        else if (primId <= NULL_PRIM) diff = 1;
        else if (lastPrimId[lastIdx] != primId) diff = 1;
        else {
            v3 norm = ld3(normal + (size_t)idx * 3), lastNorm = ld3(lastNormal + (size_t)lastIdx * 3);
            if (g_abs(dot(norm, lastNorm)) < .1f) diff = 1;
        }
        v3 color = ld3(colorIn + (size_t)idx * 3);
        float lum = luminance(color);
        v3 accumColor, accumMoment;
        if (diff) {
            accumColor = color;
            accumMoment = V3(lum, lum * lum, 0.f);
        }
        else {
            v3 lastColor = ld3(colorAccIn + (size_t)lastIdx * 3), lastMoment = ld3(momentAccIn + (size_t)lastIdx * 3);
            accumColor = mix3s(lastColor, color, Alpha);
            accumMoment = V3(mixf(lastMoment.x, lum, Alpha), mixf(lastMoment.y, lum * lum, Alpha), lastMoment.z + 1.f);
        }
        st3(colorOut + (size_t)idx * 3, accumColor);
        st3(momentOut + (size_t)idx * 3, accumMoment);
    }
}

/* estimateVariance (:307-343) */
static void svgf_estimate_variance(float* variance, const float* moment, int width, int height) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            int idx = y * width + x;
            v3 m = ld3(moment + (size_t)idx * 3);
            if (m.z > 3.5f) { variance[idx] = m.y - m.x * m.x; continue; }
            v2 sumMoment = { 0.f, 0.f };
            int numPixel = 0;
            for (int i = -1; i <= 1; i++) {
                for (int j = -1; j <= 1; j++) {
                    int qx = x + j, qy = y + i;
                    if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
                    int idxQ = qy * width + qx;
                    sumMoment.x += moment[(size_t)idxQ * 3]; sumMoment.y += moment[(size_t)idxQ * 3 + 1];
                    numPixel++;
                }
            }
            sumMoment.x /= (float)numPixel; sumMoment.y /= (float)numPixel;
            variance[idx] = sumMoment.y - sumMoment.x * sumMoment.x;
        }
    }
}

/* filterVariance (:345-371): qx follows the outer loop variable */
static void svgf_filter_variance(float* out, const float* in, int width, int height) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            float sum = 0.f, sumWeight = 0.f;
            for (int i = -1; i <= 1; i++) {
                for (int j = -1; j <= 1; j++) {
                    int qx = x + i, qy = y + j;
                    if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
                    float weight = Gaussian3x3[i + 1][j + 1];
                    sum += in[qy * width + qx] * weight;
                    sumWeight += weight;
                }
            }
            out[y * width + x] = sum / sumWeight;
        }
    }
}

/* waveletFilter, SVGF form (:139-216) */
static void svgf_wavelet(float* colorOut, const float* colorIn, float* varOut, const float* varIn, const float* varFiltered,
                         const orc_gbuffer* g, const orc_camera* cam, float sigDepth, float sigNormal, float sigLuminance, int level) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    const int step = 1 << level;
    const int* primIdPlane = g->primId[g->frameIdx];
    const float* normal = g->normal[g->frameIdx];
    const float* depth = g->depth[g->frameIdx];
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            int idxP = y * W + x;
            int primIdP = primIdPlane[idxP];
            if (primIdP <= NULL_PRIM) {
                st3(colorOut + (size_t)idxP * 3, ld3(colorIn + (size_t)idxP * 3));
                varOut[idxP] = varIn[idxP];
                continue;
            }
            v3 normP = ld3(normal + (size_t)idxP * 3);
            v3 colorP = ld3(colorIn + (size_t)idxP * 3);
            v3 posP = camera_get_position(cam, x, y, depth[idxP]);
            v3 sumColor = v3s(0.f);
            float sumVariance = 0.f, sumWeight = 0.f, sumWeight2 = 0.f;
            for (int i = -2; i <= 2; i++) {
                for (int j = -2; j <= 2; j++) {
                    int qx = x + j * step, qy = y + i * step;
                    int idxQ = qy * W + qx;
                    if (qx >= W || qy >= H || qx < 0 || qy < 0) continue;
                    if (primIdPlane[idxQ] != primIdP) continue;
                    v3 normQ = ld3(normal + (size_t)idxQ * 3);
                    v3 colorQ = ld3(colorIn + (size_t)idxQ * 3);
                    v3 posQ = camera_get_position(cam, qx, qy, depth[idxQ]);
                    float varQ = varIn[idxQ];
                    v3 dp = sub(posP, posQ);
                    float distPos2 = dot(dp, dp);
                    float wPos = expf(-distPos2 / sigDepth) + 1e-4f;
                    float wNorm = powf(sat_dot(normP, normQ), sigNormal) + 1e-4f;
                    float denom = sigLuminance * sqrtf(g_max(varFiltered[idxQ], 0.f)) + 1e-4f;
                    float wColor = expf(-g_abs(luminance(colorP) - luminance(colorQ)) / denom) + 1e-4f;
                    float weight = wColor * wNorm * wPos * Gaussian5x5[i + 2][j + 2];
                    float weight2 = weight * weight;
                    sumColor = add(sumColor, scl(colorQ, weight));
                    sumVariance += varQ * weight2;
                    sumWeight += weight;
                    sumWeight2 += weight2;
                }
            }
            st3(colorOut + (size_t)idxP * 3, (sumWeight < FLT_EPSILON) ? ld3(colorIn + (size_t)idxP * 3) : dvs(sumColor, sumWeight));
            varOut[idxP] = (sumWeight2 < FLT_EPSILON) ? varIn[idxP] : sumVariance / sumWeight2;
        }
    }
}

#define SWAP_PTR(a, b) do { float* t_ = (a); (a) = (b); (b) = t_; } while (0)

/* SpatioTemporalFilter::filter (:532-564); the state's colorOut plays the caller's `glm::vec3*& devColorOut`
 * carried from call to call.  Returns the buffer that holds the result. */
const float* orc_svgf_filter(void* p, const float* colorIn, const orc_gbuffer* g, const orc_camera* cam) {
    svgf_t* f = (svgf_t*)p;
    const float sigLumin = 4.f, sigNormal = 128.f, sigDepth = 1.f;          /* :488 */
    const int W = f->width, H = f->height, fi = f->frameIdx;
    svgf_temporal(f->accumColor[fi], f->accumColor[fi ^ 1], f->accumMoment[fi], f->accumMoment[fi ^ 1], colorIn, g, f->firstTime);
    f->firstTime = 0;
    svgf_estimate_variance(f->variance, f->accumMoment[fi], W, H);

    svgf_filter_variance(f->filteredVariance, f->variance, W, H);
    svgf_wavelet(f->colorOut, f->accumColor[fi], f->tempVariance, f->variance, f->filteredVariance, g, cam, sigDepth, sigNormal, sigLumin, 0);
    SWAP_PTR(f->colorOut, f->accumColor[fi]);
    SWAP_PTR(f->tempVariance, f->variance);

    svgf_filter_variance(f->filteredVariance, f->variance, W, H);
    svgf_wavelet(f->colorOut, f->accumColor[fi], f->tempVariance, f->variance, f->filteredVariance, g, cam, sigDepth, sigNormal, sigLumin, 1);
    SWAP_PTR(f->tempVariance, f->variance);

    for (int level = 2; level <= 4; level++) {
        svgf_filter_variance(f->filteredVariance, f->variance, W, H);
        svgf_wavelet(f->tempColor, f->colorOut, f->tempVariance, f->variance, f->filteredVariance, g, cam, sigDepth, sigNormal, sigLumin, level);
        SWAP_PTR(f->tempColor, f->colorOut);
        SWAP_PTR(f->tempVariance, f->variance);
    }
    return f->colorOut;
}

void orc_modulate(int w, int h, float* image, const float* albedo) {   /* denoiser.cu:218-228 */
    for (int i = 0; i < w * h; i++) {
        v3 color = ld3(image + (size_t)i * 3);
        color = dvs(color, 1.f);                                  /* LDRToHDR mathUtil.h:40-43 */
        color = dvv(color, adds(sub(v3s(1.f), color), 1e-4f));
        v3 al = vmax(ld3(albedo + (size_t)i * 3), v3s(0.f));
        st3(image + (size_t)i * 3, mul(color, al));
    }
}
void orc_add(int w, int h, float* image, const float* in) {
    for (int i = 0; i < w * h * 3; i++) image[i] += in[i];
}
void orc_add3(int w, int h, float* out, const float* in1, const float* in2) {
    for (int i = 0; i < w * h * 3; i++) out[i] = in1[i] + in2[i];
}

/* ------------------------------------------------------------------------------------------
 * Host scene build: BVH (bvh.cpp:10-202), alias table (sampler.h:79-121), light table
 * (scene.cpp:159-190)
 * ---------------------------------------------------------------------------------------- */
typedef struct { v3 pMin, pMax; } aabb_t;
static inline aabb_t aabb_empty(void) { aabb_t a = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } }; return a; }
static inline aabb_t aabb_union_pt(aabb_t a, v3 p) { aabb_t r = { vmin(a.pMin, p), vmax(a.pMax, p) }; return r; }    /* bvh.h:26-28 */
static inline aabb_t aabb_union(aabb_t a, aabb_t b) { aabb_t r = { vmin(a.pMin, b.pMin), vmax(a.pMax, b.pMax) }; return r; } /* :30-32 */
static inline v3 aabb_center(aabb_t a) { return scl(add(a.pMin, a.pMax), .5f); }     /* :47-49 */
static inline float aabb_surface_area(aabb_t a) {                                     /* :51-54 */
    v3 size = sub(a.pMax, a.pMin);
    return 2.f * (size.x * size.y + size.y * size.z + size.z * size.x);
}
static inline int aabb_longest_axis(aabb_t a) {                                       /* :59-67 */
    v3 size = sub(a.pMax, a.pMin);
    if (size.x < size.y) return size.y > size.z ? 1 : 2;
    else                 return size.x > size.z ? 0 : 2;
}

typedef struct { int isLeaf; int primIdOrSize; } node_info_t;
typedef struct { int primId; aabb_t bound; v3 center; } prim_info_t;
typedef struct { int offset, start, end; } build_info_t;

int orc_bvh_build(int numPrims, const float* vertices, float* boxesOut, int* nodesOut[6]) {
    const int BVHSize = numPrims * 2 - 1;
    prim_info_t* primInfo = (prim_info_t*)malloc(sizeof(prim_info_t) * (size_t)numPrims);
    prim_info_t* temp = (prim_info_t*)malloc(sizeof(prim_info_t) * (size_t)numPrims);
    node_info_t* nodeInfo = (node_info_t*)malloc(sizeof(node_info_t) * (size_t)BVHSize);
    aabb_t* boxes = (aabb_t*)malloc(sizeof(aabb_t) * (size_t)BVHSize);
    build_info_t* stack = (build_info_t*)malloc(sizeof(build_info_t) * (size_t)BVHSize);

    for (int i = 0; i < numPrims; i++) {
        v3 va = ld3(vertices + (size_t)i * 9), vb = ld3(vertices + (size_t)i * 9 + 3), vc = ld3(vertices + (size_t)i * 9 + 6);
        primInfo[i].primId = i;
        primInfo[i].bound.pMin = vmin(vmin(va, vb), vc);        /* bvh.h:20-21 */
        primInfo[i].bound.pMax = vmax(vmax(va, vb), vc);
        primInfo[i].center = aabb_center(primInfo[i].bound);
    }

    int stackTop = 0;
    stack[stackTop].offset = 0; stack[stackTop].start = 0; stack[stackTop].end = numPrims - 1; stackTop++;
    enum { NumBuckets = 16 };

    while (stackTop) {
        stackTop--;
        int offset = stack[stackTop].offset;
        int start = stack[stackTop].start;
        int end = stack[stackTop].end;

        int numSubPrims = end - start + 1;
        int nodeSize = numSubPrims * 2 - 1;
        int isLeaf = nodeSize == 1;
        nodeInfo[offset].isLeaf = isLeaf;
        nodeInfo[offset].primIdOrSize = isLeaf ? primInfo[start].primId : nodeSize;

        aabb_t nodeBound = aabb_empty(), centerBound = aabb_empty();
        for (int i = start; i <= end; i++) {
            nodeBound = aabb_union(nodeBound, primInfo[i].bound);
            centerBound = aabb_union_pt(centerBound, primInfo[i].center);
        }
        boxes[offset] = nodeBound;
        if (isLeaf) {
            continue;
        }
        int splitAxis = aabb_longest_axis(centerBound);
        /* bvh.cpp:65-73 (nodeSize == 2) is dead: nodeSize is always odd */

        aabb_t bucketBounds[NumBuckets];
        int bucketCounts[NumBuckets];
        for (int i = 0; i < NumBuckets; i++) { bucketBounds[i] = aabb_empty(); bucketCounts[i] = 0; }

        float dimMin = comp(centerBound.pMin, splitAxis);
        float dimMax = comp(centerBound.pMax, splitAxis);

        for (int i = start; i <= end; i++) {
            int bid = i_clamp(f2i_host((comp(primInfo[i].center, splitAxis) - dimMin) / (dimMax - dimMin) * (float)NumBuckets),
                              0, NumBuckets - 1);
            bucketBounds[bid] = aabb_union(bucketBounds[bid], primInfo[i].bound);
            bucketCounts[bid]++;
        }

        aabb_t lBounds[NumBuckets], rBounds[NumBuckets];
        int countPrefix[NumBuckets];
        for (int i = 0; i < NumBuckets; i++) { lBounds[i] = aabb_empty(); rBounds[i] = aabb_empty(); }

        lBounds[0] = bucketBounds[0];
        rBounds[NumBuckets - 1] = bucketBounds[NumBuckets - 1];
        countPrefix[0] = bucketCounts[0];
        for (int i = 1, j = NumBuckets - 2; i < NumBuckets; i++, j--) {
            lBounds[i] = aabb_union(lBounds[i], bucketBounds[i - 1]);    /* sic: not cumulative (bvh.cpp:97-98) */
            rBounds[j] = aabb_union(rBounds[j], bucketBounds[j + 1]);
            countPrefix[i] = countPrefix[i - 1] + bucketCounts[i];
        }

        float minSAH = FLT_MAX;
        int divBucket = 0;
        for (int i = 0; i < NumBuckets - 1; i++) {
            float SAH = mixf(aabb_surface_area(lBounds[i]), aabb_surface_area(rBounds[i + 1]),
                             (float)countPrefix[i] / (float)numSubPrims);
            if (SAH < minSAH) {
                minSAH = SAH;
                divBucket = i;
            }
        }

        memcpy(temp, primInfo + start, (size_t)numSubPrims * sizeof(prim_info_t));
        int divPrim = start, divEnd = end;
        for (int i = 0; i < numSubPrims; i++) {
            int bid = i_clamp(f2i_host((comp(temp[i].center, splitAxis) - dimMin) / (dimMax - dimMin) * (float)NumBuckets),
                              0, NumBuckets - 1);
            if (bid <= divBucket) primInfo[divPrim++] = temp[i];
            else                  primInfo[divEnd--] = temp[i];
        }
        divPrim = i_clamp(divPrim - 1, start, end - 1);
        int lSize = 2 * (divPrim - start + 1) - 1;

        stack[stackTop].offset = offset + 1 + lSize; stack[stackTop].start = divPrim + 1; stack[stackTop].end = end; stackTop++;
        stack[stackTop].offset = offset + 1; stack[stackTop].start = start; stack[stackTop].end = divPrim; stackTop++;
    }

    for (int i = 0; i < BVHSize; i++) {
        st3(boxesOut + (size_t)i * 6, boxes[i].pMin);
        st3(boxesOut + (size_t)i * 6 + 3, boxes[i].pMax);
    }

    /* buildMTBVH (bvh.cpp:133-202) */
    int* istack = (int*)malloc(sizeof(int) * (size_t)BVHSize);
    for (int i = 0; i < 6; i++) {
        int* nodes = nodesOut[i];
        int top = 0;
        istack[top++] = 0;
        int nodeIdNew = 0;
        while (top) {
            int nodeIdOrig = istack[--top];
            int isLeaf = nodeInfo[nodeIdOrig].isLeaf;
            int nodeSize = isLeaf ? 1 : nodeInfo[nodeIdOrig].primIdOrSize;

            nodes[(size_t)nodeIdNew * 3 + 0] = isLeaf ? nodeInfo[nodeIdOrig].primIdOrSize : NULL_PRIM;
            nodes[(size_t)nodeIdNew * 3 + 1] = nodeIdOrig;
            nodes[(size_t)nodeIdNew * 3 + 2] = nodeIdNew + nodeSize;
            nodeIdNew++;

            if (isLeaf) {
                continue;
            }
            int isLeftLeaf = nodeInfo[nodeIdOrig + 1].isLeaf;
            int leftSize = isLeftLeaf ? 1 : nodeInfo[nodeIdOrig + 1].primIdOrSize;

            int left = nodeIdOrig + 1;
            int right = nodeIdOrig + 1 + leftSize;

            int dim = i / 2;
            int lesser = i & 1;
            if ((comp(aabb_center(boxes[left]), dim) < comp(aabb_center(boxes[right]), dim)) ^ lesser) {
                int t = left; left = right; right = t;
            }
            istack[top++] = right;
            istack[top++] = left;
        }
    }
    free(istack); free(stack); free(boxes); free(nodeInfo); free(temp); free(primInfo);
    return BVHSize;
}

void orc_alias_build(int n, const float* valuesIn, float* prob, int* failId, float* sumAllOut) {
    float* values = (float*)malloc(sizeof(float) * (size_t)n);
    float sumAll = 0.f;
    for (int i = 0; i < n; i++) { values[i] = valuesIn[i]; sumAll += values[i]; }
    float sumInv = (float)n / sumAll;
    for (int i = 0; i < n; i++) values[i] *= sumInv;

    typedef struct { float prob; int failId; } distrib_t;
    distrib_t* gtOne = (distrib_t*)malloc(sizeof(distrib_t) * (size_t)n * 2);
    distrib_t* lsOne = (distrib_t*)malloc(sizeof(distrib_t) * (size_t)n * 2);
    int topGt = 0, topLs = 0;

    for (int i = 0; i < n; i++) {
        distrib_t d = { values[i], i };
        if (values[i] > 1.f) gtOne[topGt++] = d; else lsOne[topLs++] = d;
    }
    while (topGt && topLs) {
        distrib_t gt = gtOne[--topGt];
        distrib_t ls = lsOne[--topLs];
        prob[ls.failId] = ls.prob; failId[ls.failId] = gt.failId;
        gt.prob -= (1.f - ls.prob);
        if (gt.prob > 1.f) gtOne[topGt++] = gt; else lsOne[topLs++] = gt;
    }
    for (int i = topGt - 1; i >= 0; i--) { prob[gtOne[i].failId] = gtOne[i].prob; failId[gtOne[i].failId] = gtOne[i].failId; }
    for (int i = topLs - 1; i >= 0; i--) { prob[lsOne[i].failId] = lsOne[i].prob; failId[lsOne[i].failId] = lsOne[i].failId; }
    *sumAllOut = sumAll;
    free(lsOne); free(gtOne); free(values);
}

int orc_light_table(int numPrims, const float* vertices, const int* materialIds,
                    const orc_material* mats, int* lightPrimIds, float* lightUnitRadiance,
                    float* lightPower) {
    int n = 0;
    for (int p = 0; p < numPrims; p++) {
        const orc_material* m = &mats[materialIds[p]];
        if (m->type != MAT_LIGHT) continue;
        v3 radianceUnitArea = ld3(m->baseColor);
        float powerUnitArea = luminance(radianceUnitArea) * 2.f * GLM_PI_F;
        v3 v0 = ld3(vertices + (size_t)p * 9), v1 = ld3(vertices + (size_t)p * 9 + 3), v2_ = ld3(vertices + (size_t)p * 9 + 6);
        float area = triangle_area(v0, v1, v2_);
        lightPrimIds[n] = p;
        st3(lightUnitRadiance + (size_t)n * 3, radianceUnitArea);
        lightPower[n] = powerUnitArea * area;
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Instance baking of Scene::buildDevData (scene.cpp:161-171): Math::buildTransformationMatrix (mathUtil.cpp:13-20) with
 * glm::translate / rotate / scale (gtc/matrix_transform.inl:40-134), mat4 product and inverse (type_mat4x4.inl:37-92,
 * 686-704), normalMat = transpose(mat3(inverse)) (scene.cpp:276-278).  Host code in the reference: glibc cosf / sinf.
 * Matrices are column-major float[16].
 * ---------------------------------------------------------------------------------------- */
typedef struct { float c[4][4]; } m4;     /* c[col][row] */
static m4 m4_identity(void) { m4 m; memset(&m, 0, sizeof m); m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.f; return m; }
static m4 m4_mul(m4 a, m4 b) {           /* type_mat4x4.inl:686-704: Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3] */
    m4 r;
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
            r.c[j][k] = ((a.c[0][k] * b.c[j][0] + a.c[1][k] * b.c[j][1]) + a.c[2][k] * b.c[j][2]) + a.c[3][k] * b.c[j][3];
    return r;
}
static m4 glm_translate(m4 m, v3 v) {     /* :40-49 */
    m4 r = m;
    for (int k = 0; k < 4; k++) r.c[3][k] = ((m.c[0][k] * v.x + m.c[1][k] * v.y) + m.c[2][k] * v.z) + m.c[3][k];
    return r;
}
static m4 glm_rotate(m4 m, float angle, v3 v) {   /* :52-85 */
    const float c = cosf(angle), s = sinf(angle);
    v3 axis = normalize3(v);
    v3 temp = scl(axis, 1.f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    m4 r;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++) r.c[j][k] = (m.c[0][k] * R[j][0] + m.c[1][k] * R[j][1]) + m.c[2][k] * R[j][2];
    for (int k = 0; k < 4; k++) r.c[3][k] = m.c[3][k];
    return r;
}
static m4 glm_scale(m4 m, v3 v) {         /* :122-134 */
    m4 r;
    for (int k = 0; k < 4; k++) { r.c[0][k] = m.c[0][k] * v.x; r.c[1][k] = m.c[1][k] * v.y; r.c[2][k] = m.c[2][k] * v.z; r.c[3][k] = m.c[3][k]; }
    return r;
}
static m4 build_transformation_matrix(v3 translation, v3 rotation, v3 scale) {      /* mathUtil.cpp:13-20 */
    m4 translationMat = glm_translate(m4_identity(), translation);
    m4 rotationMat = glm_rotate(m4_identity(), rotation.x * PI_F / 180.f, V3(1.f, 0.f, 0.f));
    rotationMat = m4_mul(rotationMat, glm_rotate(m4_identity(), rotation.y * PI_F / 180.f, V3(0.f, 1.f, 0.f)));
    rotationMat = m4_mul(rotationMat, glm_rotate(m4_identity(), rotation.z * PI_F / 180.f, V3(0.f, 0.f, 1.f)));
    m4 scaleMat = glm_scale(m4_identity(), scale);
    return m4_mul(m4_mul(translationMat, rotationMat), scaleMat);
}
static m4 m4_inverse(m4 mm) {             /* type_mat4x4.inl:37-92 */
#define M(c_, r_) mm.c[c_][r_]
    float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3), Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3), Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3), Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3), Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2), Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2), Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3), Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3), Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2), Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2), Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1), Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1), Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    const float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 }, Fac1[4] = { Coef04, Coef04, Coef06, Coef07 }, Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    const float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 }, Fac4[4] = { Coef16, Coef16, Coef18, Coef19 }, Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    const float Vec0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) }, Vec1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
    const float Vec2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) }, Vec3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };
    const float SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
    m4 inv;
    for (int k = 0; k < 4; k++) {
        inv.c[0][k] = ((Vec1[k] * Fac0[k] - Vec2[k] * Fac1[k]) + Vec3[k] * Fac2[k]) * SignA[k];
        inv.c[1][k] = ((Vec0[k] * Fac0[k] - Vec2[k] * Fac3[k]) + Vec3[k] * Fac4[k]) * SignB[k];
        inv.c[2][k] = ((Vec0[k] * Fac1[k] - Vec1[k] * Fac3[k]) + Vec3[k] * Fac5[k]) * SignA[k];
        inv.c[3][k] = ((Vec0[k] * Fac2[k] - Vec1[k] * Fac4[k]) + Vec2[k] * Fac5[k]) * SignB[k];
    }
    const float d0 = M(0,0) * inv.c[0][0], d1 = M(0,1) * inv.c[1][0], d2 = M(0,2) * inv.c[2][0], d3 = M(0,3) * inv.c[3][0];
    const float Dot1 = (d0 + d1) + (d2 + d3);
    const float ood = 1.f / Dot1;
    for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) inv.c[j][k] = inv.c[j][k] * ood;
#undef M
    return inv;
}

/* the float libm calls of Scene::loadCamera (scene.cpp:344-348), as this host's libm evaluates them */
float orc_tanf(float x) { return tanf(x); }
float orc_atanf(float x) { return atanf(x); }

void orc_build_transformation_matrix(const float* t, const float* r, const float* sc, float* out16) {
    m4 m = build_transformation_matrix(ld3(t), ld3(r), ld3(sc));
    memcpy(out16, &m, sizeof m);
}
/* transform, then scene.cpp:169-170 on n vertices / normals: vec3(transform * vec4(v, 1)), normalize(normalMat * n) */
void orc_bake_instance(const float* t, const float* r, const float* sc, int n, const float* vertsIn, const float* normalsIn,
                       float* vertsOut, float* normalsOut) {
    m4 tr = build_transformation_matrix(ld3(t), ld3(r), ld3(sc));
    m4 inv = m4_inverse(tr);
    /* normalMat = transpose(mat3(transfInv)): column j of normalMat = row j of the upper-left 3x3 of inv */
    float nm[3][3];
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) nm[j][k] = inv.c[k][j];
    for (int i = 0; i < n; i++) {
        v3 v = ld3(vertsIn + (size_t)i * 3), nn = ld3(normalsIn + (size_t)i * 3);
        /* mat4 * vec4 (type_mat4x4.inl:612-628): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
        v3 o;
        o.x = (tr.c[0][0] * v.x + tr.c[1][0] * v.y) + (tr.c[2][0] * v.z + tr.c[3][0] * 1.f);
        o.y = (tr.c[0][1] * v.x + tr.c[1][1] * v.y) + (tr.c[2][1] * v.z + tr.c[3][1] * 1.f);
        o.z = (tr.c[0][2] * v.x + tr.c[1][2] * v.y) + (tr.c[2][2] * v.z + tr.c[3][2] * 1.f);
        st3(vertsOut + (size_t)i * 3, o);
        v3 q = V3(nm[0][0] * nn.x + nm[1][0] * nn.y + nm[2][0] * nn.z,
                  nm[0][1] * nn.x + nm[1][1] * nn.y + nm[2][1] * nn.z,
                  nm[0][2] * nn.x + nm[1][2] * nn.y + nm[2][2] * nn.z);
        st3(normalsOut + (size_t)i * 3, normalize3(q));
    }
}

/* ------------------------------------------------------------------------------------------
 * function-level entry points
 * ---------------------------------------------------------------------------------------- */
static inline ray_t ld_ray(const float* p) { ray_t r; r.origin = ld3(p); r.direction = ld3(p + 3); return r; }

void orc_intersect_triangle(int n, const float* rays, const float* tris, int* hit, float* bary, float* dist) {
    for (int i = 0; i < n; i++) {
        v2 b = { 0.f, 0.f }; float d = 0.f;
        hit[i] = intersect_triangle(ld_ray(rays + 6 * i), ld3(tris + 9 * i), ld3(tris + 9 * i + 3), ld3(tris + 9 * i + 6), &b, &d);
        bary[2 * i] = b.x; bary[2 * i + 1] = b.y; dist[i] = d;
    }
}
void orc_aabb_intersect(int n, const float* rays, const float* boxes, int* hit, float* tMin) {
    for (int i = 0; i < n; i++) {
        float t = 0.f;
        hit[i] = aabb_intersect(ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3), ld_ray(rays + 6 * i), &t);
        tMin[i] = t;
    }
}
void orc_utilhash(int n, const uint32_t* in, uint32_t* out) { for (int i = 0; i < n; i++) out[i] = utilhash(in[i]); }
void orc_rng_stream(int n, const int* looper, const int* index, const int* dim, int m, float* out) {
    for (int i = 0; i < n; i++) {
        rng_t r = make_seeded_random_engine(looper[i], index[i], dim[i], 0);
        for (int k = 0; k < m; k++) out[(size_t)i * m + k] = sample1D(&r);
    }
}
/* src/sampler.h:9-36: m draws of the Sobol-branch sampler for each (looper, index, dim) over the table `data` */
void orc_sobol_stream(const uint32_t* data, int n, const int* looper, const int* index, const int* dim, int m, float* out) {
    for (int i = 0; i < n; i++) {
        rng_t r = make_seeded_random_engine(looper[i], index[i], dim[i], data);
        for (int k = 0; k < m; k++) out[(size_t)i * m + k] = sample1D(&r);
    }
}
void orc_rng_stream_raw(int n, const int* seeds, int m, float* out) {
    for (int i = 0; i < n; i++) {
        rng_t r = rng_seed_raw((uint32_t)seeds[i]);
        for (int k = 0; k < m; k++) out[(size_t)i * m + k] = sample1D(&r);
    }
}
void orc_triangle_misc(int n, const float* tris, const float* x, float* area, float* normal, float* pdf) {
    for (int i = 0; i < n; i++) {
        v3 v0 = ld3(tris + 9 * i), v1 = ld3(tris + 9 * i + 3), v2_ = ld3(tris + 9 * i + 6);
        area[i] = triangle_area(v0, v1, v2_);
        v3 nrm = triangle_normal(v0, v1, v2_);
        st3(normal + 3 * i, nrm);
        pdf[i] = pdf_area_to_solid_angle(luminance(v1), ld3(x + 3 * i), v0, nrm);
    }
}
void orc_bsdf(int n, const orc_material* mats, const float* nrm, const float* wo, const float* wi, float* out) {
    for (int i = 0; i < n; i++) st3(out + 3 * i, material_bsdf(&mats[i], ld3(nrm + 3 * i), ld3(wo + 3 * i), ld3(wi + 3 * i)));
}
void orc_camera_sample(const orc_camera* cam, int n, const int* xy, const float* r, float* rays) {
    for (int i = 0; i < n; i++) {
        v4 r4 = { r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3] };
        ray_t ray = camera_sample(cam, xy[2 * i], xy[2 * i + 1], r4);
        st3(rays + 6 * i, ray.origin); st3(rays + 6 * i + 3, ray.direction);
    }
}
void orc_camera_raster_coord(const orc_camera* cam, int n, const float* pos, int* xy) {
    for (int i = 0; i < n; i++) camera_raster_coord(cam, ld3(pos + 3 * i), &xy[2 * i], &xy[2 * i + 1]);
}
void orc_camera_position(const orc_camera* cam, int n, const int* xy, const float* dist, float* pos) {
    for (int i = 0; i < n; i++) st3(pos + 3 * i, camera_get_position(cam, xy[2 * i], xy[2 * i + 1], dist[i]));
}
void orc_sample_triangle_uniform(int n, const float* tris, const float* ruv, float* out) {
    for (int i = 0; i < n; i++)
        st3(out + 3 * i, sample_triangle_uniform(ld3(tris + 9 * i), ld3(tris + 9 * i + 3), ld3(tris + 9 * i + 6), ruv[2 * i], ruv[2 * i + 1]));
}
void orc_to_concentric_disk(int n, const float* xy, float* out) {
    for (int i = 0; i < n; i++) { v2 p = to_concentric_disk(xy[2 * i], xy[2 * i + 1]); out[2 * i] = p.x; out[2 * i + 1] = p.y; }
}
void orc_linear_sample(const orc_texture* tex, int n, const float* uv, float* out) {
    for (int i = 0; i < n; i++) { v2 u = { uv[i * 2], uv[i * 2 + 1] }; st3(out + (size_t)i * 3, linear_sample(tex, u)); }
}
void orc_to_sphere(int n, const float* uv, float* dir) {
    for (int i = 0; i < n; i++) { v2 u = { uv[i * 2], uv[i * 2 + 1] }; st3(dir + (size_t)i * 3, to_sphere(u)); }
}
void orc_to_plane(int n, const float* dir, float* uv) {
    for (int i = 0; i < n; i++) { v2 u = to_plane(ld3(dir + (size_t)i * 3)); uv[i * 2] = u.x; uv[i * 2 + 1] = u.y; }
}
/* Material::sample / pdf on n inputs: r3 = sample3D; out dir[3], bsdf[3], pdf, type */
void orc_material_sample(int n, const orc_material* mats, const float* nrm, const float* wo, const float* r3,
                         float* dir, float* bsdf, float* pdf, uint32_t* type) {
    for (int i = 0; i < n; i++) {
        bsdf_sample_t sp;
        material_sample(&mats[i], ld3(nrm + (size_t)i * 3), ld3(wo + (size_t)i * 3), ld3(r3 + (size_t)i * 3), &sp);
        if (sp.type == BS_INVALID) { sp.dir = v3s(0.f); sp.bsdf = v3s(0.f); sp.pdf = 0.f; }
        st3(dir + (size_t)i * 3, sp.dir); st3(bsdf + (size_t)i * 3, sp.bsdf); pdf[i] = sp.pdf; type[i] = sp.type;
    }
}
void orc_material_pdf(int n, const orc_material* mats, const float* nrm, const float* wo, const float* wi, float* pdf) {
    for (int i = 0; i < n; i++) pdf[i] = material_pdf(&mats[i], ld3(nrm + (size_t)i * 3), ld3(wo + (size_t)i * 3), ld3(wi + (size_t)i * 3));
}
void orc_local_to_world(int n, const float* nrm, const float* v, float* out) {
    for (int i = 0; i < n; i++) st3(out + (size_t)i * 3, local_to_world(ld3(nrm + (size_t)i * 3), ld3(v + (size_t)i * 3)));
}
void orc_procedural_texture(int n, const float* uv, float* out) {
    for (int i = 0; i < n; i++) { v2 u = { uv[i * 2], uv[i * 2 + 1] }; st3(out + (size_t)i * 3, procedural_texture(u)); }
}
/* scene.cpp:139-146 (host code: glibc sinf in every mode) */
void orc_envmap_pdf(int width, int height, const float* data, float* pdf) {
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            int idx = i * width + j;
            pdf[idx] = luminance(ld3(data + (size_t)idx * 3)) * sinf((.5f + (float)i) / (float)height * PI_F);
        }
    }
}

void orc_intersect(const orc_scene* s, int n, const float* rays, int* primId, int* matId, float* pos, float* norm, float* uv) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        isect_t it; memset(&it, 0, sizeof it);
        scene_intersect(s, ld_ray(rays + (size_t)6 * i), &it);
        primId[i] = it.primId;
        matId[i] = it.primId != NULL_PRIM ? it.matId : -1;
        st3(pos + (size_t)3 * i, it.pos); st3(norm + (size_t)3 * i, it.norm);
        uv[2 * (size_t)i] = it.uv.x; uv[2 * (size_t)i + 1] = it.uv.y;
    }
}
void orc_test_occlusion(const orc_scene* s, int n, const float* seg, int* occluded) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) occluded[i] = scene_test_occlusion(s, ld3(seg + (size_t)6 * i), ld3(seg + (size_t)6 * i + 3));
}
void orc_sample_direct_light_nv(const orc_scene* s, int n, const float* pos, const float* r,
                                float* pdf, float* Li, float* wi, float* dist) {
    for (int i = 0; i < n; i++) {
        v3 L = v3s(0.f), w = v3s(0.f); float d = 0.f;
        v4 r4 = { r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3] };
        pdf[i] = sample_direct_light_nv(s, ld3(pos + 3 * i), r4, &L, &w, &d);
        st3(Li + 3 * i, L); st3(wi + 3 * i, w); dist[i] = d;
    }
}
