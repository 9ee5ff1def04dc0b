// rs_math.h -- FP32 vector math shared by host and device code of librestir_hip.
//
// Parity contract: every helper evaluates in the operation order of the GLM 0.9.6.3 function the
// reference calls (external/include/glm/detail/func_geometric.inl, func_common.inl), so that with
// -ffp-contract=off and IEEE divide/sqrt the device results are bit-identical to a host evaluation.
// Nothing here may be "simplified" algebraically.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_HD __host__ __device__ __forceinline__

namespace rs {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

constexpr float kPi    = 3.1415926535897932384626422832795028841971f;   // mathUtil.h:11
constexpr float kGlmPi = 3.14159265358979323846264338327950288f;        // glm::pi<float>()
constexpr int   kNullPrim = -1;                                         // bvh.h:12
constexpr float kInvalidPdf = -1.f;                                     // material.h:14

RS_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RS_HD f3 splat(float s) { return mk3(s, s, s); }
RS_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
RS_HD void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

RS_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RS_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RS_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RS_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
RS_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RS_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RS_HD f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
RS_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// scalar helpers with GLM's exact comparison forms (they differ from fminf/fmaxf on NaN)
RS_HD float gabs(float x) { return x >= 0.f ? x : -x; }
RS_HD float gmin(float x, float y) { return x < y ? x : y; }
RS_HD float gmax(float x, float y) { return x > y ? x : y; }
RS_HD int   imin(int a, int b) { return a < b ? a : b; }
RS_HD int   imax(int a, int b) { return a > b ? a : b; }
RS_HD int   iclamp(int v, int lo, int hi) { return imin(imax(v, lo), hi); }
RS_HD f3    vmin(f3 a, f3 b) { return mk3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
RS_HD f3    vmax(f3 a, f3 b) { return mk3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }

RS_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RS_HD f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
}  // namespace rs
#define RS_MATH_H_BODY
#include "rs_exact.h"          // device: sqrt_exact / rcp_exact -- the IEEE results, fewer instructions for operands in [2^-60, 2^60)
namespace rs {
#if defined(__HIP_DEVICE_COMPILE__)
RS_HD float rsqrt_glm(float x) { return rcp_exact(sqrt_exact(x)); }     // glm::inversesqrt = 1 / sqrt(x)
RS_HD float length(f3 v) { return sqrt_exact(dot(v, v)); }
#else
RS_HD float rsqrt_glm(float x) { return 1.f / sqrtf(x); }            // glm::inversesqrt
RS_HD float length(f3 v) { return sqrtf(dot(v, v)); }
#endif
RS_HD f3 normalize(f3 v) { return v * rsqrt_glm(dot(v, v)); }
RS_HD f3 mix(f3 x, f3 y, float a) { return x + (y - x) * a; }          // x + a*(y-x)
RS_HD f3 mix(f3 x, f3 y, f3 a) { return x + a * (y - x); }
RS_HD float mixf(float x, float y, float a) { return x + a * (y - x); }
RS_HD float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// column-major 3x3 times vector (type_mat3x3.inl operator*)
RS_HD f3 mul_cols(f3 c0, f3 c1, f3 c2, f3 v) {
    return mk3(c0.x * v.x + c1.x * v.y + c2.x * v.z,
               c0.y * v.x + c1.y * v.y + c2.y * v.z,
               c0.z * v.x + c1.z * v.y + c2.z * v.z);
}

// float -> int as the kernels' hardware conversion does it: truncate, saturate, NaN -> 0
// (v_cvt_i32_f32; the same as CUDA's cvt.rzi.s32.f32 the reference runs under).
RS_HD int f2i(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rz(f);
#else
    if (f != f) return 0;
    if (f >= 2147483648.f) return 2147483647;
    if (f <= -2147483648.f) return (-2147483647 - 1);
    return (int)f;
#endif
}

RS_HD bool is_nan_or_inf(float x) { return (x != x) || (gabs(x) == __builtin_inff()); }
RS_HD bool any_nan_or_inf(f3 v) { return is_nan_or_inf(v.x) || is_nan_or_inf(v.y) || is_nan_or_inf(v.z); }

RS_HD float luminance(f3 c) { return dot(c, mk3(.2126f, .7152f, .0722f)); }   // mathUtil.h:119-123
RS_HD float sat_dot(f3 a, f3 b) { return gmax(dot(a, b), 0.f); }
RS_HD float abs_dot(f3 a, f3 b) { return gabs(dot(a, b)); }
RS_HD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }

// Math::utilhash (mathUtil.h:190-198)
RS_HD uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

// ---- RNG: thrust::minstd_rand + uniform_real_distribution<float>(0,1) as used by sampler.h:38-61.
// x <- 48271*x mod (2^31-1); the 64-bit product is folded with 2^31 == 1 (mod m), which yields
// the same residue as thrust's Schrage form.
struct Rng {
    uint32_t x;
    RS_HD uint32_t next() {
        uint64_t p = (uint64_t)x * 48271u;
        uint32_t s = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
        if (s >= 2147483647u) s -= 2147483647u;
        x = s;
        return s;
    }
    // float(x - min) / (1.f + float(max - min)) with min=1, max=2^31-2: the divisor is 2^31 exactly
    RS_HD float uniform() { return (float)(next() - 1u) / 2147483648.f; }
    RS_HD f2 uniform2() { f2 r; r.x = uniform(); r.y = uniform(); return r; }
    RS_HD f4 uniform4() { f4 r; r.x = uniform(); r.y = uniform(); r.z = uniform(); r.w = uniform(); return r; }
};

// makeSeededRandomEngine (sampler.h:41-44) followed by linear_congruential_engine::seed
RS_HD Rng seeded_rng(int iter, int index, int dim) {
    uint32_t h = utilhash(0x80000000u | ((uint32_t)dim << 22) | (uint32_t)iter) ^ utilhash((uint32_t)index);
    uint32_t v = h % 2147483647u;
    Rng r; r.x = v == 0u ? 1u : v;
    return r;
}

// ---- the Sobol branch of src/sampler.h:9-36 (SAMPLER_USE_SOBOL): Sampler{ptr, scramble, data} over the table
// DevScene::sampleSequence, SobolSampleNum x SobolSampleDim uint32 (scene.cpp:500-506).
constexpr int kSobolSampleDim = 200;          // sampler.h:11
struct SobolSampler {
    const uint32_t* data;
    uint32_t scramble;
    int ptr;
    RS_HD float uniform() {                   // Sampler::sample (sampler.h:19-23): `r * 0x1p-32f` converts r to float first
        const uint32_t r = data[ptr++] ^ scramble;
        scramble = utilhash(scramble);
        return (float)r * 0x1p-32f;
    }
    RS_HD f2 uniform2() { f2 r; r.x = uniform(); r.y = uniform(); return r; }
    RS_HD f4 uniform4() { f4 r; r.x = uniform(); r.y = uniform(); r.z = uniform(); r.w = uniform(); return r; }
};

// One name for both samplers, chosen by a kernel's template parameter.  `seeded` is makeSeededRandomEngine(iter, index, dim, data);
// `word` / `resume` carry a sampler from one pass of a frame to the next in ONE 32-bit word per pixel: the generator's state, or the
// scramble -- the table position is then iter * SobolSampleDim + dim + (draws made so far), which the passes of ReSTIRDirect know.
template <bool SOBOL> struct SamplerT;
template <> struct SamplerT<false> : Rng {
    static RS_HD SamplerT seeded(const uint32_t*, int iter, int index, int dim) { SamplerT s; s.x = seeded_rng(iter, index, dim).x; return s; }
    static RS_HD SamplerT resume(const uint32_t*, uint32_t word, int, int) { SamplerT s; s.x = word; return s; }
    RS_HD uint32_t word() const { return x; }
};
template <> struct SamplerT<true> : SobolSampler {
    static RS_HD SamplerT seeded(const uint32_t* table, int iter, int index, int dim) {           // sampler.h:30-32
        SamplerT s; s.data = table; s.scramble = utilhash((uint32_t)index); s.ptr = iter * kSobolSampleDim + dim; return s;
    }
    static RS_HD SamplerT resume(const uint32_t* table, uint32_t word, int iter, int drawn) {
        SamplerT s; s.data = table; s.scramble = word; s.ptr = iter * kSobolSampleDim + drawn; return s;
    }
    RS_HD uint32_t word() const { return scramble; }
};

}  // namespace rs
