// rs_surface.h -- texture maps and the environment map of DevScene (included at the end of rs_scene.h):
//   linearSample / DevTextureObj          src/image.h:41-97
//   proceduralTexture                     src/scene.h:68-76
//   getTexturedMaterialAndSurface         src/scene.h:78-99
//   Math::toSphere / toPlane / localToWorld   src/mathUtil.h:134-155
//   sampleEnvironmentMapNoVisibility      src/scene.h:364-376
//
// Texture filtering, the normal-map frame and the environment sampler's pdf are + - * / floor only, so
// they are bit-exact against a host evaluation like everything else.  The four libm calls of these paths
// (sin/cos in toSphere and proceduralTexture, atan2 in toPlane) are evaluated in double precision and
// rounded once: the correctly rounded float result, which is what the parity tests pin (the oracle's
// "correctly rounded" libm mode); glibc's and CUDA's float versions each differ from it by an ulp for a
// percent or so of arguments.  These functions run only in the TEX kernel variants (scenes with a map or
// an environment map), so their cost and registers never touch the untextured path.
#pragma once

namespace rs {

constexpr float kPiTwo = 6.2831853071795864769252867665590057683943f;    // mathUtil.h:12
constexpr int kNullTexture = -1, kProceduralTex = -2;                     // material.h:11-13

RS_HD float fractf_glm(float x) { return x - floorf(x); }                 // func_common.inl:318-321

#if defined(__HIPCC__)
__device__ __forceinline__ float cr_sin(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cr_cos(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float cr_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

// image.h:41-75 with T = glm::vec3
__device__ inline f3 linear_sample(const TexRec t, float u, float v) {
    const float Eps = 1.17549435e-38f;                                    // FLT_MIN
    u = fractf_glm(u); v = fractf_glm(v);
    const float fx = u * ((float)t.width - Eps) + .5f;
    const float fy = v * ((float)t.height - Eps) + .5f;
    int ix = f2i(fractf_glm(fx) > .5f ? fx : fx - 1);
    if (ix < 0) ix += t.width;
    int iy = f2i(fractf_glm(fy) > .5f ? fy : fy - 1);
    if (iy < 0) iy += t.height;
    int ux = ix + 1;
    if (ux >= t.width) ux -= t.width;
    int uy = iy + 1;
    if (uy >= t.height) uy -= t.height;
    const float lx = fractf_glm(fx + .5f);
    const float ly = fractf_glm(fy + .5f);
    const float* d = t.data;
    const f3 c1 = mix(ld3(d + ((size_t)iy * t.width + ix) * 3), ld3(d + ((size_t)iy * t.width + ux) * 3), lx);
    const f3 c2 = mix(ld3(d + ((size_t)uy * t.width + ix) * 3), ld3(d + ((size_t)uy * t.width + ux) * 3), lx);
    return mix(c1, c2, ly);
}

// scene.h:68-76
__device__ inline f3 procedural_texture(float u, float v) {
    const uint32_t seed = (uint32_t)(f2i(u * 1024.f) * 1024 + f2i(v * 1024.f));
    const uint32_t m = seed % 2147483647u;
    Rng rng; rng.x = m == 0u ? 1u : m;                                    // linear_congruential_engine::seed
    const float rx = rng.uniform();
    const float ry = rng.uniform();
    const float f = (cr_sin(u * 10.f * kPiTwo + rx * kPiTwo) + 1.f) * .5f;
    const float g = (cr_sin(v * 10.f * kPiTwo + ry * kPiTwo) + 1.f) * .5f;
    return splat(f * g);
}

// mathUtil.h:134-137
__device__ inline f3 to_sphere(float u, float v) {
    u *= kPiTwo; v *= kPi;
    return mk3(cr_cos(u) * cr_sin(v), cr_cos(v), cr_sin(u) * cr_sin(v));
}
// mathUtil.h:139-144; PiInv is the macro `1.f / Pi`: x * PiInv * .5f is ((x * 1.f) / Pi) * .5f
__device__ inline void to_plane(f3 d, float& u, float& v) {
    u = fractf_glm(((cr_atan2(d.z, d.x) * 1.f) / kPi) * .5f + 1.f);
    v = (cr_atan2(sqrtf(d.x * d.x + d.z * d.z), d.y) * 1.f) / kPi;
}
// mathUtil.h:146-155
__device__ inline f3 local_to_world(f3 n, f3 v) {
    f3 t = (gabs(n.y) > 0.9999f) ? mk3(0.f, 0.f, 1.f) : mk3(0.f, 1.f, 0.f);
    const f3 b = normalize(cross(n, t));
    t = cross(b, n);
    return normalize(mul_cols(t, b, n, v));
}

// what the kernels need of a Material after getTexturedMaterialAndSurface
struct SurfMat { int type; f3 baseColor; float metallic, roughness, ior; };

__device__ inline SurfMat plain_material(const DevScene& s, int matId) {
    const rs_material m = s.materials[matId];
    SurfMat o; o.type = m.type; o.baseColor = ld3(m.baseColor); o.metallic = m.metallic; o.roughness = m.roughness; o.ior = m.ior;
    return o;
}

// scene.h:135-151: uv part of getIntersecGeomInfo
__device__ inline void hit_uv(const DevScene& s, const Hit& h, float& u, float& v) {
    const float* t = s.texcoords + (size_t)h.primId * 6;
    const float w = 1.f - h.bx - h.by;
    u = t[2] * h.bx + t[4] * h.by + t[0] * w;
    v = t[3] * h.bx + t[5] * h.by + t[1] * w;
}

// scene.h:78-99; a normal map also replaces the interpolated normal
__device__ inline SurfMat textured_material(const DevScene& s, const Hit& h, f3& norm) {
    const rs_material m = s.materials[h.matId];
    SurfMat o; o.type = m.type; o.baseColor = ld3(m.baseColor); o.metallic = m.metallic; o.roughness = m.roughness; o.ior = m.ior;
    if (m.baseColorMapId == kNullTexture && m.metallicMapId <= kNullTexture && m.roughnessMapId <= kNullTexture && m.normalMapId == kNullTexture)
        return o;
    float u, v;
    hit_uv(s, h, u, v);
    if (m.baseColorMapId != kNullTexture)
        o.baseColor = m.baseColorMapId == kProceduralTex ? procedural_texture(u, v) : linear_sample(s.textures[m.baseColorMapId], u, v);
    if (m.metallicMapId > kNullTexture) o.metallic = linear_sample(s.textures[m.metallicMapId], u, v).x;
    if (m.roughnessMapId > kNullTexture) o.roughness = linear_sample(s.textures[m.roughnessMapId], u, v).x;
    if (m.normalMapId != kNullTexture) {
        const f3 mapped = linear_sample(s.textures[m.normalMapId], u, v);
        const f3 localNorm = normalize((mapped * 1.f) + (-0.5f));
        norm = local_to_world(norm, localNorm);
    }
    return o;
}

// envMap->linearSample(Math::toPlane(dir)) (restir.cu:134-136, gbuffer.cu:59-62, pathtrace.cu:295-297)
__device__ inline f3 env_radiance(const DevScene& s, f3 dir) {
    float u, v;
    to_plane(dir, u, v);
    return linear_sample(s.textures[s.envTex], u, v);
}

// sampleEnvironmentMapNoVisibility (scene.h:364-376); `PiInv * PiInv * .5f` is x * 1.f / Pi * 1.f / Pi * .5f
__device__ inline float sample_env_nv(const DevScene& s, float r1, float r2, f3& radiance, f3& wi) {
    const TexRec env = s.textures[s.envTex];
    const int pass = imin(f2i((float)s.envLen * r1), s.envLen - 1);       // DevDiscreteSampler1D::sample (sampler.h:203-207)
    const AliasRec al = s.envAlias[pass];
    const int pix = r2 < al.prob ? pass : al.failId;
    const int y = pix / env.width;
    const int x = pix - y * env.width;
    radiance = ld3(env.data + (size_t)pix * 3);
    wi = to_sphere((.5f + (float)x) / (float)env.width, (.5f + (float)y) / (float)env.height);
    return ((((luminance(radiance) * s.sumLightPowerInv * (float)env.width * (float)env.height * 1.f) / kPi) * 1.f) / kPi) * .5f;
}

// sampleDirectLightNoVisibility (scene.h:394-425); `lights`/`alias` may point to global memory or to an LDS copy.
// Bit-exact shortcuts: dot(x-y, x-y) == dot(y-x, y-x) and normalize(x-y) == -normalize(y-x), so
// the pdf conversion (mathUtil.h:182-185) reuses wi and dist instead of re-deriving them.
// ENV: the scene has an environment map, whose sampler entry is the last one (scene.h:400-403).
// The exactly rounded square root of a uniform variate.  Rng::uniform() is k * 2^-31 with an integer k, i.e. 0 or a normal number in
// [2^-31, 1]: the range scaling (x < 2^-96) and the zero / infinity / NaN class test of the general expansion are dead for such an
// argument (7 of its 16 vector instructions); what is left is the expansion's own refinement of the hardware estimate -- one ulp down
// or up, decided by the sign of the two exact residuals -- which returns 0 for 0 (the residual of the step down is NaN, of the step
// up +0: both comparisons fail).  tests/test_gpu_parity.py compares the RIS winners with the oracle's sqrtf bit for bit.
__device__ __forceinline__ float sqrt_of_uniform(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rDown = __builtin_fmaf(-down, s, x), rUp = __builtin_fmaf(-up, s, x);
    float root = rDown <= 0.f ? down : s;
    root = rUp > 0.f ? up : root;
    return root;
}

// the four 16-byte quarters of a light record: from the record array, or from an LDS copy laid out by quarter (restir.hip k_ris_lds)
__device__ __forceinline__ void load_light(const LightRec* lights, int id, float4& a, float4& b, float4& c, float4& d) {
    const float4* lp = reinterpret_cast<const float4*>(&lights[id]);
    a = lp[0]; b = lp[1]; c = lp[2]; d = lp[3];
}
template <int N>
struct LightQuarters {                      // quarter k of light i at q[k * N + i]
    const float4* q;
};
template <int N>
__device__ __forceinline__ void load_light(LightQuarters<N> lights, int id, float4& a, float4& b, float4& c, float4& d) {
    a = lights.q[id]; b = lights.q[N + id]; c = lights.q[2 * N + id]; d = lights.q[3 * N + id];
}

template <bool ENV, typename AliasPtr, typename LightPtr>
__device__ __forceinline__ LightSample sample_light_nv(const DevScene& s, AliasPtr alias, LightPtr lights, int numLights, f3 pos, f4 r) {
    LightSample o;
    o.pdf = kInvalidPdf; o.Li = splat(0.f); o.wi = splat(0.f); o.dist = 0.f; o.point = splat(0.f); o.id = 0; o.bu = o.bv = 0.f;
    if (numLights == 0) return o;
    int pass = imin(f2i((float)numLights * r.x), numLights - 1);      // DevDiscreteSampler1D::sample
    AliasRec al = alias[pass];
    int id = r.y < al.prob ? pass : al.failId;
    o.id = id;
    if (ENV && id == numLights - 1) {
        o.dist = 1e10f;
        o.pdf = sample_env_nv(s, r.z, r.w, o.Li, o.wi);
        o.point = pos + o.wi * 1e6f;                                 // sampleEnvironmentMap's occlusion target (scene.h:387)
        return o;
    }
    float4 a, b, c, d;
    load_light(lights, id, a, b, c, d);
    f3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
    f3 nrm = mk3(a.w, b.w, c.w);
    float sr = sqrt_of_uniform(r.w);               // sampleTriangleUniform(v0,v1,v2, ru=r.z, rv=r.w); r is always Rng::uniform4()
    float u = 1.f - sr;
    float v = r.z * sr;
    f3 sampled = v1 * u + v2 * v + v0 * (1.f - u - v);
    o.point = sampled; o.bu = u; o.bv = v;
    f3 toS = sampled - pos;
    if (dot(nrm, toS) > -1e-6f) return o;          // SCENE_LIGHT_SINGLE_SIDED
    float dd = dot(toS, toS);
    float len = sqrt_exact(dd);                    // rs_exact.h: the IEEE results, fewer instructions for operands in [2^-60, 2^60)
    o.Li = mk3(d.x, d.y, d.z);
    o.wi = toS * rcp_exact(len);
    o.dist = len;
    o.pdf = div_exact(d.w * dd, gabs(-dot(nrm, o.wi)));
    return o;
}

// The Li / wi / dist of an accepted triangle-light sample again, from its light and barycentric pair: the same expressions in the same
// order as above, hence the same bits.  Lets a reservoir loop carry three values per winner instead of seven.
template <typename LightPtr>
__device__ __forceinline__ void light_sample_again(LightPtr lights, int id, float u, float v, f3 pos, f3& Li, f3& wi, float& dist) {
    float4 a, b, c, d;
    load_light(lights, id, a, b, c, d);
    const f3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
    const f3 sampled = v1 * u + v2 * v + v0 * (1.f - u - v);
    const f3 toS = sampled - pos;
    const float len = sqrt_exact(dot(toS, toS));
    Li = mk3(d.x, d.y, d.z);
    wi = toS * rcp_exact(len);
    dist = len;
}

#endif  // __HIPCC__

}  // namespace rs
