// rs_scene.h -- device-resident scene (the DevScene of src/scene.h:64-481, re-laid-out for CDNA4)
// and the per-ray / per-sample services the kernels call.
//
// HBM layout (all arrays hipMalloc'ed once by rs_scene_create, read-only afterwards):
//   nodes[k]   BvhNode[bvhSize]   32 B: MTBVHNode (src/bvh.h:163-171) fused with the AABB it points to,
//                                 so one traversal step is ONE 32-byte fetch (two dwordx4 from one
//                                 sector) instead of the reference's two dependent loads
//                                 (12-B node -> 24-B box, src/scene.h:254).  6 threaded orders.
//   tris       TriRec[numPrims]   48 B: v0, e01 = v1-v0, e02 = v2-v0 (the first two lines of
//                                 intersectTriangle, src/intersections.h:20-21, hoisted to scene build;
//                                 IEEE subtraction, so bit-identical) -> three dwordx4 per leaf test.
//   vertices / normals            float[9] per triangle, as given (attribute fetch of the final hit only)
//   materialIds, materials        as given (44-B Material records)
//   lights     LightRec[numLights] 64 B: v0,v1,v2, geometric normal, unit radiance and the
//                                 per-light constant pdfArea = lum(Le)/(area*2*pi) * sumLightPowerInv
//                                 (src/scene.h:411,419-424: per-candidate in the reference, per-light here)
//   alias      AliasRec[numLights] 8 B: BinomialDistrib {prob, failId} (src/sampler.h:63-67)
#pragma once

#include "rs_math.h"
#include "../../include/restir_hip.h"

namespace rs {

struct __attribute__((aligned(16))) BvhNode {
    float bminx, bminy, bminz; int primId;
    float bmaxx, bmaxy, bmaxz; int next;
};
struct __attribute__((aligned(16))) TriRec {
    float v0x, v0y, v0z, pad0;
    float e1x, e1y, e1z, pad1;
    float e2x, e2y, e2z, pad2;
};
struct __attribute__((aligned(16))) LightRec {
    float v0x, v0y, v0z, nx;
    float v1x, v1y, v1z, ny;
    float v2x, v2y, v2z, nz;
    float Lx, Ly, Lz, pdfArea;
};
struct AliasRec { float prob; int failId; };

struct DevScene {
    const BvhNode* nodes[6];
    const TriRec*  tris;
    const float*   vertices;
    const float*   normals;
    const int*     materialIds;
    const rs_material* materials;
    const LightRec* lights;
    const AliasRec* alias;
    int bvhSize;
    int numPrims;
    int numLights;
    int numMaterials;
};

struct Ray { f3 o, d; };

struct Hit {
    int primId;
    int matId;
    f3  pos;
    f3  norm;
};

// ---- camera (src/sceneStructs.h:22-86) ---------------------------------------------------------
// The reference evaluates tan(radians(fov.y)) per thread; it is a per-frame constant, so the host
// evaluates the same expression once (same libm as any host evaluation) and passes it in.
struct CamParams {
    f3 position, right, up, view;
    f3 inv0, inv1, inv2;         // columns of rotationMatInv
    float aspect, tanFovY, focalDist, lensRadius;
    float pixelSizeX, pixelSizeY;
    int width, height;
};

// Camera::sample (sceneStructs.h:69-86); r = first two components of the 4-D jitter
RS_HD Ray camera_sample(const CamParams& c, int x, int y, float rx, float ry) {
    float scrx = (float)x * c.pixelSizeX, scry = (float)y * c.pixelSizeY;
    float ruvx = scrx + c.pixelSizeX * rx, ruvy = scry + c.pixelSizeY * ry;
    ruvx = 1.f - ruvx * 2.f;
    ruvy = 1.f - ruvy * 2.f;
    f3 pLens = mk3(0.f * c.lensRadius, 0.f * c.lensRadius, 0.f);
    f3 pFocus = mk3(ruvx * c.aspect * c.tanFovY, ruvy * 1.f * c.tanFovY, 1.f) * c.focalDist;
    f3 dir = pFocus - pLens;
    Ray r;
    r.d = normalize(mul_cols(c.right, c.up, c.view, dir));
    r.o = c.position + c.right * pLens.x + c.up * pLens.y;
    return r;
}

// pixel-centre ray of renderGBuffer (gbuffer.cu:11-24)
RS_HD Ray camera_center_ray(const CamParams& c, int x, int y) {
    float scrx = (float)x * c.pixelSizeX, scry = (float)y * c.pixelSizeY;
    float ruvx = scrx + c.pixelSizeX * .5f, ruvy = scry + c.pixelSizeY * .5f;
    f3 pLens = splat(0.f);
    f3 pFocus = mk3((1.f - ruvx * 2.f) * c.aspect * c.tanFovY, (1.f - ruvy * 2.f) * 1.f * c.tanFovY, 1.f) * c.focalDist;
    f3 dir = pFocus - pLens;
    Ray r;
    r.d = normalize(mul_cols(c.right, c.up, c.view, dir));
    r.o = c.position + c.right * pLens.x + c.up * pLens.y;
    return r;
}

// Camera::getPosition (sceneStructs.h:48-64)
RS_HD f3 camera_get_position(const CamParams& c, int x, int y, float dist) {
    float scrx = (float)x * c.pixelSizeX, scry = (float)y * c.pixelSizeY;
    float ruvx = scrx + c.pixelSizeX * .5f, ruvy = scry + c.pixelSizeY * .5f;
    ruvx = 1.f - ruvx * 2.f;
    ruvy = 1.f - ruvy * 2.f;
    f3 pLens = mk3(0.f * c.lensRadius, 0.f * c.lensRadius, 0.f);
    f3 pFocus = mk3(ruvx * c.aspect * c.tanFovY, ruvy * 1.f * c.tanFovY, 1.f) * c.focalDist;
    f3 dir = normalize(mul_cols(c.right, c.up, c.view, pFocus - pLens));
    f3 ori = c.position + c.right * pLens.x + c.up * pLens.y;
    return ori + dir * dist;
}

// Camera::getRasterCoord (sceneStructs.h:23-46)
RS_HD void camera_raster_coord(const CamParams& c, f3 pos, int& ox, int& oy) {
    f3 dir = normalize(pos - c.position);
    float d = 1.f / dot(dir, c.view);
    f3 p = mul_cols(c.inv0, c.inv1, c.inv2, dir * d);
    p = p / mk3(c.aspect * c.tanFovY, 1.f * c.tanFovY, 1.f);
    float ndcx = -p.x, ndcy = -p.y;
    ndcx = ndcx * .5f + .5f;
    ndcy = ndcy * .5f + .5f;
    ox = f2i((float)c.width * ndcx);
    oy = f2i((float)c.height * ndcy);
}

// ---- ray / box / triangle ----------------------------------------------------------------------
// Per-ray constants of AABB::intersect (src/bvh.h:85-157): which special case applies depends only
// on the ray direction, so it is classified once per ray instead of once per node.
struct RayBoxCtx {
    f3 o, d, dinv;
    int mode;          // 0 general, 1/2/3 axis-aligned along x/y/z (abs(d) > 1-1e-6, first match)
    bool zx, zy, zz;   // abs(d.c) < 1e-6
};

RS_HD RayBoxCtx make_box_ctx(const Ray& r) {
    const float Eps = 1e-6f;
    RayBoxCtx c;
    c.o = r.o; c.d = r.d;
    c.dinv = mk3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
    c.mode = gabs(r.d.x) > 1.f - Eps ? 1 : (gabs(r.d.y) > 1.f - Eps ? 2 : (gabs(r.d.z) > 1.f - Eps ? 3 : 0));
    c.zx = gabs(r.d.x) < Eps; c.zy = gabs(r.d.y) < Eps; c.zz = gabs(r.d.z) < Eps;
    return c;
}

RS_HD bool in_range(float x, float lo, float hi) { return x >= lo && x <= hi; }

RS_HD bool slab_max_min(float n1, float n2, float f1, float f2_, float& tMin) {   // getDistMaxMin bvh.h:75-79
    tMin = fmaxf(n1, n2);
    float tMax = fminf(f1, f2_);
    return tMax >= 0.f && tMax >= tMin;
}
RS_HD bool slab_min_max(float t1, float t2, float& tMin) {                        // getDistMinMax bvh.h:69-73
    tMin = fminf(t1, t2);
    float tMax = fmaxf(t1, t2);
    return tMax >= 0.f && tMax >= tMin;
}

// true only when the ray's coordinate on the ignored axis stays outside [lo-tol, hi+tol] for every
// t in [max(t0,0), t1]; NaN / infinite inputs never skip.
RS_HD bool skip_far_on_axis(float o, float d, float lo, float hi, float t0, float t1) {
    const float a = o + d * fmaxf(t0, 0.f), b = o + d * t1;
    const float tol = 1e-3f * (1.f + fmaxf(gabs(lo), gabs(hi)));
    const float mn = fminf(a, b), mx = fmaxf(a, b);
    return (mx < lo - tol) || (mn > hi + tol);
}

RS_HD bool box_hit(const RayBoxCtx& c, f3 bmin, f3 bmax, float& tMin) {
    if (c.mode != 0) {                         // axis-aligned rays (bvh.h:91-123), rare
        if (c.mode == 1) {
            if (in_range(c.o.y, bmin.y, bmax.y) && in_range(c.o.z, bmin.z, bmax.z))
                return slab_min_max((bmin.x - c.o.x) * c.dinv.x, (bmax.x - c.o.x) * c.dinv.x, tMin);
            return false;
        }
        if (c.mode == 2) {
            if (in_range(c.o.z, bmin.z, bmax.z) && in_range(c.o.x, bmin.x, bmax.x))
                return slab_min_max((bmin.y - c.o.y) * c.dinv.y, (bmax.y - c.o.y) * c.dinv.y, tMin);
            return false;
        }
        if (in_range(c.o.x, bmin.x, bmax.x) && in_range(c.o.y, bmin.y, bmax.y))
            return slab_min_max((bmin.z - c.o.z) * c.dinv.z, (bmax.z - c.o.z) * c.dinv.z, tMin);
        return false;
    }
    f3 t1 = (bmin - c.o) * c.dinv;
    f3 t2 = (bmax - c.o) * c.dinv;
    f3 tn = vmin(t1, t2);
    f3 tf = vmax(t1, t2);
    f3 td = tf - tn;
    float yz = tf.z - tn.y;
    float zx = tf.x - tn.z;
    float xy = tf.y - tn.x;
    bool oyz = td.y + td.z > yz, ozx = td.z + td.x > zx, oxy = td.x + td.y > xy;
    // Near-zero direction component: the reference tests only the other two slabs (bvh.h:136-146), so
    // such a ray "enters" every box its projection crosses and walks thousands of nodes (measured:
    // 2.5k-10k steps against a mean of 130; a handful of such rays per 1080p frame set the kernel's
    // tail).  skip_far_on_axis() adds a conservative cull on the ignored axis: a box is skipped only
    // if the ray stays farther than a generous tolerance from it over the interval it crosses the
    // other two slabs.  A skipped subtree cannot contain a triangle the ray hits (a Moeller-Trumbore
    // hit point lies inside its triangle's box up to rounding << tol), the visiting order of the
    // remaining nodes is unchanged, so closest hit, ties and occlusion results are identical.
    if (c.zx && oyz) return slab_max_min(tn.y, tn.z, tf.y, tf.z, tMin) && !skip_far_on_axis(c.o.x, c.d.x, bmin.x, bmax.x, tMin, fminf(tf.y, tf.z));
    if (c.zy && ozx) return slab_max_min(tn.z, tn.x, tf.z, tf.x, tMin) && !skip_far_on_axis(c.o.y, c.d.y, bmin.y, bmax.y, tMin, fminf(tf.z, tf.x));
    if (c.zz && oxy) return slab_max_min(tn.x, tn.y, tf.x, tf.y, tMin) && !skip_far_on_axis(c.o.z, c.d.z, bmin.z, bmax.z, tMin, fminf(tf.x, tf.y));
    if (oyz && ozx && oxy)
        return slab_max_min(fmaxf(tn.x, tn.y), tn.z, fminf(tf.x, tf.y), tf.z, tMin);
    return false;
}

// intersectTriangle (src/intersections.h:17-54) on a pre-differenced triangle record
RS_HD bool tri_hit(f3 o, f3 d, f3 v0, f3 e01, f3 e02, float& bx, float& by, float& dist) {
    f3 p = cross(d, e02);
    float det = dot(p, e01);
    if (gabs(det) < 1.1920928955078125e-7f) return false;       // FLT_EPSILON
    f3 t = o - v0;
    if (det < 0.f) { det = -det; t = -t; }
    bx = dot(t, p);
    if (bx < 0.f || bx > det) return false;
    f3 q = cross(t, e01);
    by = dot(d, q);
    if (by < 0.f || bx + by > det) return false;
    float inv = 1.f / det;
    dist = dot(e02, q) * inv;
    bx *= inv;
    by *= inv;
    return dist > 0.f;
}

// DevScene::getMTBVHId (src/scene.h:101-119)
RS_HD int mtbvh_order(f3 dir) {
    float ax = gabs(dir.x), ay = gabs(dir.y), az = gabs(dir.z);
    if (ax > ay) {
        if (ax > az) return dir.x > 0 ? 0 : 1;
        return dir.z > 0 ? 4 : 5;
    }
    if (ay > az) return dir.y > 0 ? 2 : 3;
    return dir.z > 0 ? 4 : 5;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void load_node(const BvhNode* nodes, int node, f3& bmin, f3& bmax, int& prim, int& next) {
    const float4* p = reinterpret_cast<const float4*>(nodes + node);
    float4 a = p[0], b = p[1];
    bmin = mk3(a.x, a.y, a.z); prim = __float_as_int(a.w);
    bmax = mk3(b.x, b.y, b.z); next = __float_as_int(b.w);
}
__device__ __forceinline__ void load_tri(const TriRec* tris, int prim, f3& v0, f3& e1, f3& e2) {
    const float4* p = reinterpret_cast<const float4*>(tris + prim);
    float4 a = p[0], b = p[1], c = p[2];
    v0 = mk3(a.x, a.y, a.z); e1 = mk3(b.x, b.y, b.z); e2 = mk3(c.x, c.y, c.z);
}

// DevScene::intersect (src/scene.h:245-284): closest hit, stackless threaded walk
__device__ inline Hit trace_closest(const DevScene& s, const Ray& ray) {
    float closest = 3.402823466e+38f;   // FLT_MAX
    int   cprim = kNullPrim;
    float cbx = 0.f, cby = 0.f;
    const BvhNode* nodes = s.nodes[mtbvh_order(-ray.d)];
    const RayBoxCtx ctx = make_box_ctx(ray);
    const int end = s.bvhSize;
    int node = 0;
    while (node != end) {
        f3 bmin, bmax; int prim, next;
        load_node(nodes, node, bmin, bmax, prim, next);
        float tb;
        bool bh = box_hit(ctx, bmin, bmax, tb);
        if (bh && tb < closest) {
            if (prim != kNullPrim) {
                f3 v0, e1, e2;
                load_tri(s.tris, prim, v0, e1, e2);
                float bx, by, dist;
                if (tri_hit(ray.o, ray.d, v0, e1, e2, bx, by, dist) && dist < closest) {
                    closest = dist; cbx = bx; cby = by; cprim = prim;
                }
            }
            node++;
        }
        else {
            node = next;
        }
    }
    Hit h;
    h.primId = cprim;
    h.matId = 0;
    h.pos = splat(0.f);
    h.norm = splat(0.f);
    if (cprim != kNullPrim) {             // getIntersecGeomInfo (scene.h:135-151)
        const float* v = s.vertices + (size_t)cprim * 9;
        const float* n = s.normals + (size_t)cprim * 9;
        float w = 1.f - cbx - cby;
        h.pos = ld3(v + 3) * cbx + ld3(v + 6) * cby + ld3(v) * w;
        h.norm = normalize(ld3(n + 3) * cbx + ld3(n + 6) * cby + ld3(n) * w);
        h.matId = s.materialIds[cprim];
    }
    return h;
}

// DevScene::testOcclusion (src/scene.h:286-316): any hit between x and y
__device__ inline bool trace_occluded(const DevScene& s, f3 x, f3 y) {
    f3 dir = y - x;
    float dist = length(dir);
    dir = dir / dist;
    Ray ray; ray.o = x + dir * 1e-5f; ray.d = dir;       // makeOffsetedRay (intersections.h:13-15)
    dist -= 1e-4f * 2.f;
    const BvhNode* nodes = s.nodes[mtbvh_order(-ray.d)];
    const RayBoxCtx ctx = make_box_ctx(ray);
    const int end = s.bvhSize;
    int node = 0;
    while (node != end) {
        f3 bmin, bmax; int prim, next;
        load_node(nodes, node, bmin, bmax, prim, next);
        float tb;
        bool bh = box_hit(ctx, bmin, bmax, tb);
        if (bh && tb < dist) {
            if (prim != kNullPrim) {
                f3 v0, e1, e2;
                load_tri(s.tris, prim, v0, e1, e2);
                float bx, by, d;
                if (tri_hit(ray.o, ray.d, v0, e1, e2, bx, by, d) && d < dist) return true;
            }
            node++;
        }
        else {
            node = next;
        }
    }
    return false;
}
#endif  // __HIPCC__

// ---- materials (src/material.h:34-124,171-186,218-228) -----------------------------------------
RS_HD float schlick_g(float c, float alpha) { float a = alpha * .5f; return c / (c * (1.f - a) + a); }
RS_HD float gtr2(float c, float alpha) {
    if (c < 1e-6f) return 0.f;
    float aa = alpha * alpha;
    float den = c * c * (aa - 1.f) + 1.f;
    den = den * den * kPi;
    return aa / den;
}

RS_HD f3 eval_bsdf(int type, f3 baseColor, float metallic, float roughness, f3 n, f3 wo, f3 wi) {
    if (type == 0) {                                   // lambertianBSDF: baseColor * 1.f / Pi
        return (baseColor * 1.f) / kPi;
    }
    if (type == 1) {                                   // metallicWorkflowBSDF
        float alpha = roughness * roughness;
        f3 h = normalize(wo + wi);
        float cosO = dot(n, wo);
        float cosI = dot(n, wi);
        if (cosI * cosO < 1e-7f) return splat(0.f);
        f3 f0 = mix(splat(.08f), baseColor, metallic);
        f3 f = mix(f0, splat(1.f), pow5(1.f - dot(h, wo)));
        float g = schlick_g(gabs(cosO), alpha) * schlick_g(gabs(cosI), alpha);
        float d = gtr2(dot(n, h), alpha);
        f3 diffuse = ((baseColor * 1.f) / kPi) * (1.f - metallic);
        return mix(diffuse, splat(g * d / (4.f * cosI * cosO)), f);
    }
    return splat(0.f);                                 // Dielectric, Disney, Light
}

// ---- light sampling (src/scene.h:394-459, src/sampler.h:203-207, src/mathUtil.h:94-100,182-185) --
struct LightSample { float pdf; f3 Li, wi; float dist; f3 point; };

#if defined(__HIPCC__)
// sampleDirectLightNoVisibility; `lights`/`alias` may point to global memory or to an LDS copy.
// Bit-exact shortcuts: dot(x-y, x-y) == dot(y-x, y-x) and normalize(x-y) == -normalize(y-x), so
// the pdf conversion (mathUtil.h:182-185) reuses wi and dist instead of re-deriving them.
template <typename AliasPtr, typename LightPtr>
__device__ __forceinline__ LightSample sample_light_nv(AliasPtr alias, LightPtr lights, int numLights, f3 pos, f4 r) {
    LightSample o;
    o.pdf = kInvalidPdf; o.Li = splat(0.f); o.wi = splat(0.f); o.dist = 0.f; o.point = splat(0.f);
    if (numLights == 0) return o;
    int pass = imin(f2i((float)numLights * r.x), numLights - 1);      // DevDiscreteSampler1D::sample
    AliasRec al = alias[pass];
    int id = r.y < al.prob ? pass : al.failId;
    const float4* lp = reinterpret_cast<const float4*>(&lights[id]);
    float4 a = lp[0], b = lp[1], c = lp[2], d = lp[3];
    f3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
    f3 nrm = mk3(a.w, b.w, c.w);
    float sr = sqrtf(r.w);                         // sampleTriangleUniform(v0,v1,v2, ru=r.z, rv=r.w)
    float u = 1.f - sr;
    float v = r.z * sr;
    f3 sampled = v1 * u + v2 * v + v0 * (1.f - u - v);
    o.point = sampled;
    f3 toS = sampled - pos;
    if (dot(nrm, toS) > -1e-6f) return o;          // SCENE_LIGHT_SINGLE_SIDED
    float dd = dot(toS, toS);
    float len = sqrtf(dd);
    o.Li = mk3(d.x, d.y, d.z);
    o.wi = toS * (1.f / len);
    o.dist = len;
    o.pdf = d.w * dd / gabs(-dot(nrm, o.wi));
    return o;
}
#endif

}  // namespace rs
