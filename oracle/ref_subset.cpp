/*
 * ref_subset.cpp -- extern "C" wrappers over the REFERENCE'S OWN functions, compiled from the
 * sources where they lie under /root/reference (nothing is copied into this repo).
 *
 * TEST INFRASTRUCTURE ONLY (builds oracle/_ref/libref_subset.so, see oracle/Makefile).
 * Used in this container to pin oracle/restir_oracle.c bit-for-bit and to generate the golden
 * vectors under tests/golden/ (tests/golden/make_golden.py).  /root/reference does not exist on
 * the GPU box; nothing at run time there needs this file's output except as a prebuilt checker.
 *
 * What compiles: the thrust-free subset of the reference -- intersections.h, bvh.h, bvh.cpp,
 * sceneStructs.h, material.h, mathUtil.h, image.h -- with plain g++, the reference's vendored GLM 0.9.6.3
 * and the genuine CUDA runtime headers that ship inside this image's Triton wheel (only for the
 * `#include <cuda_runtime.h>` lines and the __host__/__device__ annotations, which g++ ignores).
 * What does not: scene.h / sampler.h / restir.h / gbuffer.h / *.cu need CUDA Thrust and nvcc
 * (the image's rocThrust cannot coexist with CUDA headers); see DESIGN.md "Oracle".
 */
#include <cstring>
#include <cstdint>
#include <vector>

#include "intersections.h"   // /root/reference/src (via -I)
#include "bvh.h"
#include "material.h"
#include "sceneStructs.h"
#include "mathUtil.h"
#include "image.h"           // linearSample / DevTextureObj (templates; Image's methods are only declared)
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/matrix_inverse.hpp>

static inline glm::vec3 ld3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }
static inline void st3(float* p, glm::vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline Ray ldRay(const float* p) { Ray r; r.origin = ld3(p); r.direction = ld3(p + 3); return r; }

static_assert(sizeof(Material) == 44, "Material layout");
static_assert(sizeof(Camera) == 196, "Camera layout");
static_assert(sizeof(AABB) == 24, "AABB layout");
static_assert(sizeof(MTBVHNode) == 12, "MTBVHNode layout");

extern "C" {

void ref_intersect_triangle(int n, const float* rays, const float* tris, int* hit, float* bary, float* dist) {
    for (int i = 0; i < n; i++) {
        glm::vec2 b(0.f); float d = 0.f;
        hit[i] = intersectTriangle(ldRay(rays + 6 * i), ld3(tris + 9 * i), ld3(tris + 9 * i + 3), ld3(tris + 9 * i + 6), b, d);
        bary[2 * i] = b.x; bary[2 * i + 1] = b.y; dist[i] = d;
    }
}

void ref_aabb_intersect(int n, const float* rays, const float* boxes, int* hit, float* tMin) {
    for (int i = 0; i < n; i++) {
        AABB box(ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3));
        float t = 0.f;
        hit[i] = box.intersect(ldRay(rays + 6 * i), t);
        tMin[i] = t;
    }
}

void ref_utilhash(int n, const uint32_t* in, uint32_t* out) {
    for (int i = 0; i < n; i++) out[i] = Math::utilhash(in[i]);
}

void ref_bsdf(int n, const void* mats, const float* nrm, const float* wo, const float* wi, float* out) {
    for (int i = 0; i < n; i++) {
        Material m;
        std::memcpy(&m, (const char*)mats + (size_t)i * sizeof(Material), sizeof(Material));
        st3(out + 3 * i, m.BSDF(ld3(nrm + 3 * i), ld3(wo + 3 * i), ld3(wi + 3 * i)));
    }
}

void ref_camera_sample(const void* cam, int n, const int* xy, const float* r, float* rays) {
    Camera c; std::memcpy(&c, cam, sizeof(Camera));
    for (int i = 0; i < n; i++) {
        Ray ray = c.sample(xy[2 * i], xy[2 * i + 1], glm::vec4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]));
        st3(rays + 6 * i, ray.origin); st3(rays + 6 * i + 3, ray.direction);
    }
}

void ref_camera_raster_coord(const void* cam, int n, const float* pos, int* xy) {
    Camera c; std::memcpy(&c, cam, sizeof(Camera));
    for (int i = 0; i < n; i++) {
        glm::ivec2 p = c.getRasterCoord(ld3(pos + 3 * i));
        xy[2 * i] = p.x; xy[2 * i + 1] = p.y;
    }
}

void ref_camera_position(const void* cam, int n, const int* xy, const float* dist, float* pos) {
    Camera c; std::memcpy(&c, cam, sizeof(Camera));
    for (int i = 0; i < n; i++) st3(pos + 3 * i, c.getPosition(xy[2 * i], xy[2 * i + 1], dist[i]));
}

void ref_camera_update(void* cam) {
    Camera c; std::memcpy(&c, cam, sizeof(Camera));
    c.update();
    std::memcpy(cam, &c, sizeof(Camera));
}

void ref_sample_triangle_uniform(int n, const float* tris, const float* ruv, float* out) {
    for (int i = 0; i < n; i++)
        st3(out + 3 * i, Math::sampleTriangleUniform(ld3(tris + 9 * i), ld3(tris + 9 * i + 3), ld3(tris + 9 * i + 6), ruv[2 * i], ruv[2 * i + 1]));
}

void ref_to_concentric_disk(int n, const float* xy, float* out) {
    for (int i = 0; i < n; i++) {
        glm::vec2 p = Math::toConcentricDisk(xy[2 * i], xy[2 * i + 1]);
        out[2 * i] = p.x; out[2 * i + 1] = p.y;
    }
}

/* triangleArea, triangleNormal, luminance, pdfAreaToSolidAngle: out = area, normal[3], lum(normal), pdf */
void ref_triangle_misc(int n, const float* tris, const float* x, float* area, float* normal, float* pdf) {
    for (int i = 0; i < n; i++) {
        glm::vec3 v0 = ld3(tris + 9 * i), v1 = ld3(tris + 9 * i + 3), v2 = ld3(tris + 9 * i + 6);
        area[i] = Math::triangleArea(v0, v1, v2);
        glm::vec3 nrm = Math::triangleNormal(v0, v1, v2);
        st3(normal + 3 * i, nrm);
        pdf[i] = Math::pdfAreaToSolidAngle(Math::luminance(v1) , ld3(x + 3 * i), v0, nrm);
    }
}

void ref_tonemap(int n, const float* in, int mode, float* out) {
    for (int i = 0; i < n; i++) {
        glm::vec3 color = ld3(in + 3 * i);
        switch (mode) {
        case ToneMapping::Filmic: color = Math::filmic(color); break;
        case ToneMapping::ACES:   color = Math::ACES(color); break;
        case ToneMapping::None:   break;
        }
        st3(out + 3 * i, Math::correctGamma(color));
    }
}

int ref_bvh_build(int numPrims, const float* vertices, float* boxesOut, int* nodesOut[6]) {
    std::vector<glm::vec3> verts((size_t)numPrims * 3);
    std::memcpy(verts.data(), vertices, sizeof(float) * 9 * (size_t)numPrims);
    std::vector<AABB> boxes;
    std::vector<std::vector<MTBVHNode>> nodes;
    int size = BVHBuilder::build(verts, boxes, nodes);
    std::memcpy(boxesOut, boxes.data(), sizeof(AABB) * boxes.size());
    for (int i = 0; i < 6; i++) std::memcpy(nodesOut[i], nodes[i].data(), sizeof(MTBVHNode) * nodes[i].size());
    return size;
}

// A host-side closest-hit loop over the reference's own intersections.h / bvh.h: the traversal is the loop of
// DevScene::intersect (scene.h:245-284; scene.h itself needs Thrust, so the dozen lines are restated here) and every
// geometric test in it is the reference's compiled code -- AABB::intersect (bvh.h:85-157), intersectTriangle
// (intersections.h:17-54) -- on the tree of the reference's BVHBuilder (ref_bvh_build).  bench.py times it on the host cores
// as the "reference loop" CPU baseline; tests check it against the oracle.  nodes[k]: MTBVHNode arrays of the six orders.
void ref_closest_hit_loop(int bvhSize, const float* boxes, const int* const nodes[6], const float* vertices,
                          int n, const float* rays, int* primOut, float* distOut) {
    const AABB* bb = reinterpret_cast<const AABB*>(boxes);
    const glm::vec3* verts = reinterpret_cast<const glm::vec3*>(vertices);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        const Ray ray = ldRay(rays + 6 * (size_t)i);
        const glm::vec3 dir = -ray.direction, absDir = glm::abs(dir);         // getMTBVHId(-ray.direction), scene.h:101-119
        int id;
        if (absDir.x > absDir.y) id = absDir.x > absDir.z ? (dir.x > 0 ? 0 : 1) : (dir.z > 0 ? 4 : 5);
        else id = absDir.y > absDir.z ? (dir.y > 0 ? 2 : 3) : (dir.z > 0 ? 4 : 5);
        const MTBVHNode* nd = reinterpret_cast<const MTBVHNode*>(nodes[id]);
        float closestDist = FLT_MAX;
        int closestPrimId = NullPrimitive;
        int node = 0;
        while (node != bvhSize) {
            AABB bound = bb[nd[node].boundingBoxId];
            float boundDist;
            const bool boundHit = bound.intersect(ray, boundDist);
            if (boundHit && boundDist < closestDist) {
                const int primId = nd[node].primitiveId;
                if (primId != NullPrimitive) {
                    float dist;
                    glm::vec2 bary;
                    const bool hit = intersectTriangle(ray, verts[primId * 3 + 0], verts[primId * 3 + 1], verts[primId * 3 + 2], bary, dist);
                    if (hit && dist < closestDist) { closestDist = dist; closestPrimId = primId; }
                }
                node++;
            }
            else node = nd[node].nextNodeIfMiss;
        }
        primOut[i] = closestPrimId;
        distOut[i] = closestDist;
    }
}

// material.h:230-256.  Fields the reference leaves unwritten for an Invalid sample are reported as 0.
void ref_material_sample(int n, const Material* mats, const float* nrm, const float* wo, const float* r3,
                         float* dir, float* bsdf, float* pdf, uint32_t* type) {
    for (int i = 0; i < n; i++) {
        BSDFSample sp; sp.dir = glm::vec3(0.f); sp.bsdf = glm::vec3(0.f); sp.pdf = 0.f; sp.type = Invalid;
        mats[i].sample(ld3(nrm + 3 * i), ld3(wo + 3 * i), ld3(r3 + 3 * i), sp);
        if (sp.type == Invalid) { sp.dir = glm::vec3(0.f); sp.bsdf = glm::vec3(0.f); sp.pdf = 0.f; }
        st3(dir + 3 * i, sp.dir); st3(bsdf + 3 * i, sp.bsdf); pdf[i] = sp.pdf; type[i] = sp.type;
    }
}
void ref_material_pdf(int n, const Material* mats, const float* nrm, const float* wo, const float* wi, float* pdf) {
    for (int i = 0; i < n; i++) pdf[i] = mats[i].pdf(ld3(nrm + 3 * i), ld3(wo + 3 * i), ld3(wi + 3 * i));
}

// mathUtil.cpp:13-20 and the per-vertex expressions of Scene::buildDevData (scene.cpp:169-170) / loadModel (:276-278),
// written with the same GLM calls (scene.cpp itself needs thrust and cannot be compiled here)
void ref_build_transformation_matrix(const float* t, const float* r, const float* s, float* out16) {
    glm::mat4 m = Math::buildTransformationMatrix(ld3(t), ld3(r), ld3(s));
    std::memcpy(out16, &m[0][0], sizeof(float) * 16);
}
void ref_bake_instance(const float* t, const float* r, const float* s, int n, const float* vertsIn, const float* normalsIn,
                       float* vertsOut, float* normalsOut) {
    glm::mat4 transform = Math::buildTransformationMatrix(ld3(t), ld3(r), ld3(s));
    glm::mat4 transfInv = glm::inverse(transform);
    glm::mat3 normalMat = glm::transpose(glm::mat3(transfInv));
    for (int i = 0; i < n; i++) {
        st3(vertsOut + 3 * i, glm::vec3(transform * glm::vec4(ld3(vertsIn + 3 * i), 1.f)));
        st3(normalsOut + 3 * i, glm::normalize(normalMat * ld3(normalsIn + 3 * i)));
    }
}

// image.h:41-75 through DevTextureObj::linearSample (image.h:89-91)
void ref_linear_sample(int width, int height, const float* data, int n, const float* uv, float* out) {
    DevTextureObj tex;
    tex.width = width; tex.height = height;
    tex.devData = reinterpret_cast<glm::vec3*>(const_cast<float*>(data));
    for (int i = 0; i < n; i++) st3(out + 3 * i, tex.linearSample(glm::vec2(uv[2 * i], uv[2 * i + 1])));
}
// mathUtil.h:134-155
void ref_to_sphere(int n, const float* uv, float* dir) {
    for (int i = 0; i < n; i++) st3(dir + 3 * i, Math::toSphere(glm::vec2(uv[2 * i], uv[2 * i + 1])));
}
void ref_to_plane(int n, const float* dir, float* uv) {
    for (int i = 0; i < n; i++) { glm::vec2 u = Math::toPlane(ld3(dir + 3 * i)); uv[2 * i] = u.x; uv[2 * i + 1] = u.y; }
}
void ref_local_to_world(int n, const float* nrm, const float* v, float* out) {
    for (int i = 0; i < n; i++) st3(out + 3 * i, Math::localToWorld(ld3(nrm + 3 * i), ld3(v + 3 * i)));
}

} // extern "C"
