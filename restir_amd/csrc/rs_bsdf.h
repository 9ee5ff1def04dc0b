// rs_bsdf.h -- the sampling half of the BSDFs, Material::sample / pdf (src/material.h:42-62,82-121,126-170,186-256),
// and the small Math:: helpers the path-tracing kernels need (src/mathUtil.h:36-38,81-84,157-180).
// Device only; used by gi.hip.  cos / sin of the concentric-disk map are evaluated correctly rounded
// (cr_cos / cr_sin, rs_surface.h) like the other libm calls outside the spatial tap.
#pragma once

#include "rs_scene.h"

namespace rs {

#if defined(__HIPCC__)
enum : uint32_t { kBsDiffuse = 1u << 0, kBsGlossy = 1u << 1, kBsSpecular = 1u << 2, kBsReflection = 1u << 4, kBsTransmission = 1u << 5, kBsInvalid = 1u << 15 };
struct BsdfSample { f3 dir, bsdf; float pdf; uint32_t type; };

__device__ __forceinline__ f3 hdr_to_ldr(f3 c) { return (c / (c + 1.f)) * 1.f; }                           // mathUtil.h:36-38
__device__ __forceinline__ float power_heuristic(float f, float g) { const float f2 = f * f; return f2 / (f2 + g * g); }   // :81-84
__device__ __forceinline__ f3 glm_reflect(f3 I, f3 N) { return I - (N * dot(N, I)) * splat(2.f); }        // func_geometric.inl:176-179

// mathUtil.h:128-132 (polar map)
__device__ inline void to_concentric_disk_cr(float x, float y, float& ox, float& oy) {
    const float r = sqrtf(x);
    const float theta = y * kPi * 2.0f;
    ox = cr_cos(theta) * r; oy = cr_sin(theta) * r;
}
// mathUtil.h:157-161
__device__ inline f3 sample_hemisphere_cosine(f3 n, float rx, float ry) {
    float dx, dy;
    to_concentric_disk_cr(rx, ry, dx, dy);
    const float z = sqrtf(1.f - (dx * dx + dy * dy));
    return local_to_world(n, mk3(dx, dy, z));
}
// mathUtil.h:163-180
__device__ inline bool math_refract(f3 n, f3 wi, float ior, f3& wt) {
    const float cosIn = dot(n, wi);
    if (cosIn < 0) ior = 1.f / ior;
    const float sin2In = gmax(0.f, 1.f - cosIn * cosIn);
    const float sin2Tr = sin2In / (ior * ior);
    if (sin2Tr >= 1.f) return false;
    float cosTr = sqrtf(1.f - sin2Tr);
    if (cosIn < 0) cosTr = -cosTr;
    wt = normalize((-wi) / ior + n * (cosIn / ior - cosTr));
    return true;
}
// material.h:42-61 (exact Fresnel: MATERIAL_DIELECTRIC_USE_SCHLICK_APPROX is not defined)
__device__ inline float fresnel_dielectric(float cosIn, float ior) {
    if (cosIn < 0) { ior = 1.f / ior; cosIn = -cosIn; }
    const float sinIn = sqrtf(1.f - cosIn * cosIn);
    const float sinTr = sinIn / ior;
    if (sinTr >= 1.f) return 1.f;
    const float cosTr = sqrtf(1.f - sinTr * sinTr);
    const float a = (cosIn - ior * cosTr) / (cosIn + ior * cosTr), b = (ior * cosIn - cosTr) / (ior * cosIn + cosTr);
    return (a * a + b * b) * .5f;
}
// material.h:82-85
__device__ inline float gtr2_pdf(f3 n, f3 m, f3 wo, float alpha) {
    return gtr2(dot(n, m), alpha) * schlick_g(dot(n, wo), alpha) * abs_dot(m, wo) / abs_dot(n, wo);
}
// glm::inverse(mat3) (func_matrix.inl compute_inverse<tmat3x3>): columns in, columns out
__device__ inline void m3_inverse(f3 c0, f3 c1, f3 c2, f3& o0, f3& o1, f3& o2) {
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
    const float ood = 1.f / (+ m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02));
    o0 = mk3(+ (m11 * m22 - m21 * m12) * ood, - (m01 * m22 - m21 * m02) * ood, + (m01 * m12 - m11 * m02) * ood);
    o1 = mk3(- (m10 * m22 - m20 * m12) * ood, + (m00 * m22 - m20 * m02) * ood, - (m00 * m12 - m10 * m02) * ood);
    o2 = mk3(+ (m10 * m21 - m20 * m11) * ood, - (m00 * m21 - m20 * m01) * ood, + (m00 * m11 - m10 * m01) * ood);
}
// material.h:94-112: GGX sampling of visible normals
__device__ inline f3 gtr2_sample(f3 n, f3 wo, float alpha, float rx, float ry) {
    f3 t0 = (gabs(n.y) > 0.9999f) ? mk3(0.f, 0.f, 1.f) : mk3(0.f, 1.f, 0.f);       // Math::localRefMatrix
    const f3 b0 = normalize(cross(n, t0));
    t0 = cross(b0, n);
    f3 i0, i1, i2;
    m3_inverse(t0, b0, n, i0, i1, i2);
    const f3 vh = normalize(mul_cols(i0, i1, i2, wo) * mk3(alpha, alpha, 1.f));
    const float lenSq = vh.x * vh.x + vh.y * vh.y;
    const f3 t = lenSq > 0.f ? mk3(-vh.y, vh.x, 0.f) / sqrtf(lenSq) : mk3(1.f, 0.f, 0.f);
    const f3 b = cross(vh, t);
    float px, py;
    to_concentric_disk_cr(rx, ry, px, py);
    const float s = 0.5f * (vh.z + 1.f);
    py = (1.f - s) * sqrtf(1.f - px * px) + s * py;
    f3 h = t * px + b * py + vh * sqrtf(gmax(0.f, 1.f - (px * px + py * py)));
    h = mk3(h.x * alpha, h.y * alpha, gmax(0.f, h.z));
    return normalize(mul_cols(t0, b0, n, h));
}
// material.h:186-193
__device__ inline float metallic_workflow_pdf(const SurfMat& m, f3 n, f3 wo, f3 wi) {
    const f3 h = normalize(wo + wi);
    return mixf((sat_dot(n, wi) * 1.f) / kPi, gtr2_pdf(n, h, wo, m.roughness * m.roughness) / (4.f * abs_dot(h, wo)), 1.f / (2.f - m.metallic));
}
// Material::pdf (material.h:230-240)
__device__ inline float material_pdf(const SurfMat& m, f3 n, f3 wo, f3 wi) {
    if (m.type == 0) return (sat_dot(n, wi) * 1.f) / kPi;
    if (m.type == 1) return metallic_workflow_pdf(m, n, wo, wi);
    return 0.f;
}
__device__ inline f3 material_bsdf(const SurfMat& m, f3 n, f3 wo, f3 wi) {
    return eval_bsdf(m.type, m.baseColor, m.metallic, m.roughness, n, wo, wi);
}
// Material::sample (material.h:242-256)
__device__ inline BsdfSample material_sample(const SurfMat& m, f3 n, f3 wo, f3 r) {
    BsdfSample sp;
    sp.dir = splat(0.f); sp.bsdf = splat(0.f); sp.pdf = 0.f; sp.type = kBsInvalid;
    if (m.type == 0) {                                                            // lambertianSample :130-135
        sp.dir = sample_hemisphere_cosine(n, r.x, r.y);
        sp.bsdf = (m.baseColor * 1.f) / kPi;
        sp.pdf = (sat_dot(n, sp.dir) * 1.f) / kPi;
        sp.type = kBsDiffuse | kBsReflection;
    }
    else if (m.type == 1) {                                                       // metallicWorkflowSample :195-213
        const float alpha = m.roughness * m.roughness;
        if (r.z > (1.f / (2.f - m.metallic))) sp.dir = sample_hemisphere_cosine(n, r.x, r.y);
        else sp.dir = -glm_reflect(wo, gtr2_sample(n, wo, alpha, r.x, r.y));
        if (!(dot(n, sp.dir) < 0.f)) {
            sp.bsdf = material_bsdf(m, n, wo, sp.dir);
            sp.pdf = metallic_workflow_pdf(m, n, wo, sp.dir);
            sp.type = kBsGlossy | kBsReflection;
        }
    }
    else if (m.type == 2) {                                                       // dielectricSample :145-169
        const float pdfRefl = fresnel_dielectric(dot(n, wo), m.ior);
        sp.bsdf = m.baseColor;
        if (r.z < pdfRefl) {
            sp.dir = glm_reflect(-wo, n);
            sp.type = kBsSpecular | kBsReflection;
            sp.pdf = 1.f;
        }
        else if (math_refract(n, wo, m.ior, sp.dir)) {
            float eta = m.ior;
            if (dot(n, wo) < 0) eta = 1.f / eta;
            sp.bsdf = sp.bsdf / (eta * eta);
            sp.type = kBsSpecular | kBsTransmission;
            sp.pdf = 1.f;
        }
    }
    return sp;
}
#endif  // __HIPCC__

}  // namespace rs
