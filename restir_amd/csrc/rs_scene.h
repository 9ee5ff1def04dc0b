// rs_scene.h -- device-resident scene (the DevScene of src/scene.h:64-481, re-laid-out for CDNA4)
// and the per-ray / per-sample services the kernels call.
//
// HBM layout (all arrays hipMalloc'ed once by rs_scene_create, read-only afterwards):
//   nodesAll   BvhNode[6*bvhSize+1] 32 B: MTBVHNode (src/bvh.h:163-171) fused with the AABB it points to,
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
//   occNodes   uint4[occCount]    16 B: the shadow-ray tree (occlusion_bvh.cpp): box on a 16-bit grid over the
//                                 scene bounds {lo.x|lo.y<<16, lo.z|hi.x<<16, hi.y|hi.z<<16}, w = miss link of an
//                                 inner node (as a byte offset: index * 16) or ~(firstTriangle*8+count) of a leaf.  Null when the fast path is off.
//   occChain   BvhNode[bvhSize]   32 B: the reference's boxes by ORIGINAL node id with primId = next = parent id
//                                 (-1 at the root): the path a candidate occluder is verified against
//   occTris    TriRec[numPrims]   the same pre-differenced triangles in the shadow tree's leaf order,
//                                 pad0 = bit pattern of the triangle's reference leaf node id
#pragma once

#include "rs_math.h"
#include "rs_exact.h"
#include "../../include/restir_hip.h"

namespace rs {

// field order: the packet walk reads a record through the scalar cache and feeds SGPR pairs to packed FP32 instructions, which
// want (min.x, min.y), (min.z, max.z), (max.x, max.y) in aligned pairs; node_unpack() restores the (min | prim), (max | next)
// view the per-lane walks are written with (register renaming, no instructions)
struct __attribute__((aligned(16))) BvhNode {
    float bminx, bminy, bminz, bmaxz;
    float bmaxx, bmaxy; int primId; int next;
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
struct TexRec { const float* data; int width, height; };       // DevTextureObj (src/image.h:76-97): packed float[3] texels

struct DevScene {
    const BvhNode* nodesAll;      // 6 * bvhSize records (+1 padding record): order k starts at k * bvhSize
    const TriRec*  tris;
    const float*   vertices;
    const float*   normals;
    const float*   texcoords;     // 6 floats / triangle; null unless the scene has texture maps
    const int*     materialIds;
    const rs_material* materials;
    const LightRec* lights;
    const AliasRec* alias;
    const TexRec*  textures;      // null unless the scene has texture maps or an environment map
    const AliasRec* envAlias;     // envMapSampler (src/scene.h:364-376), envLen = width*height of the map, 0 = none
    int envTex, envLen;           // envMap = textures + envTex (src/scene.cpp:495-498), -1 = none
    float sumLightPowerInv;       // src/scene.cpp:489
    const uint4*   occNodes;
    const BvhNode* occChain;
    const TriRec*  occTris;
    f3 occBase, occScale;         // grid plane q on axis c = occBase.c + q * occScale.c
    f3 occRootLo, occRootHi;      // the reference's root box
    bool occNested;               // every reference box lies inside its parent's (enables the leaf shortcut)
    bool axisCull;                // boxes contain their children and triangles (enables skip_far_on_axis)
    bool linksNested;             // in every threaded order the miss links nest: c in (a, link(a)) => link(c) <= link(a)
    int occCount;
    // the emissive triangles alone, as a tree of the shadow tree's kind (scene.hip build_emissive_side; may_hit_emissive_wave):
    // emiState 0 = not built (every ray may hit one), 1 = built, emiCount nodes (0: the scene has no emissive triangle)
    const uint4*   emiNodes;
    const TriRec*  emiTris;
    f3 emiBase, emiScale;
    int emiCount, emiState;
    // closest hit of incoherent rays: six trees in the reference's six visiting orders (occlusion_bvh.cpp rs_build_ordered_bvh), 16-byte
    // records on the shadow tree's grid, order k at byte offset k * ordStride, its end record at (k + 1) * ordStride - 16;
    // ordTris: per axis numPrims triangles in the even order's sequence (pad0 = reference leaf node, pad1 = primitive id).  Null = off.
    const uint4*   ordNodes;
    const TriRec*  ordTris;
    unsigned ordStride;
    unsigned long long* walkStats;   // null unless built with -DRS_WALK_STATS (tools/walk_stats.py)
    const unsigned char* occDepth;   // depth of every occNodes record (-DRS_WALK_STATS builds only)
    int bvhSize;
    int numPrims;
    int numLights;
    int numMaterials;
    // DevScene::sampleSequence (src/scene.h:480, src/scene.cpp:500-506): the Sobol table, sampleCount x kSobolSampleDim uint32 followed
    // by a guard of kSobolGuard zeros; null = the default thrust engine (SAMPLER_USE_SOBOL false).  rs_scene_set_sample_sequence.
    const uint32_t* sampleSeq;
    int sampleCount;
};
constexpr int kSobolGuard = 4096;

typedef float vf2 __attribute__((ext_vector_type(2)));     // operands of the packed FP32 instructions (v_pk_add / mul / fma_f32)

struct Ray { f3 o, d; };

struct Hit {
    int primId;
    int matId;
    f3  pos;
    f3  norm;
    float bx, by;     // barycentrics of the hit (for the texture coordinates of textured scenes)
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
    bool cull;         // skip_far_on_axis allowed: the box table is a proper hierarchy (DevScene::axisCull)
};

RS_HD RayBoxCtx make_box_ctx(const Ray& r) {
    const float Eps = 1e-6f;
    RayBoxCtx c;
    c.o = r.o; c.d = r.d;
#if defined(__HIP_DEVICE_COMPILE__)
    c.dinv = rcp3_exact_signed(r.d);
#else
    c.dinv = mk3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
#endif
    c.mode = gabs(r.d.x) > 1.f - Eps ? 1 : (gabs(r.d.y) > 1.f - Eps ? 2 : (gabs(r.d.z) > 1.f - Eps ? 3 : 0));
    c.zx = gabs(r.d.x) < Eps; c.zy = gabs(r.d.y) < Eps; c.zz = gabs(r.d.z) < Eps;
    c.cull = true;
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
    // remaining nodes is unchanged, so closest hit, ties and occlusion results are identical.  That argument
    // needs boxes that contain their triangles and their children, which rs_scene_create checks
    // (DevScene::axisCull); for any other caller-supplied table the cull is off.
    if (c.zx && oyz) return slab_max_min(tn.y, tn.z, tf.y, tf.z, tMin) && !(c.cull && skip_far_on_axis(c.o.x, c.d.x, bmin.x, bmax.x, tMin, fminf(tf.y, tf.z)));
    if (c.zy && ozx) return slab_max_min(tn.z, tn.x, tf.z, tf.x, tMin) && !(c.cull && skip_far_on_axis(c.o.y, c.d.y, bmin.y, bmax.y, tMin, fminf(tf.z, tf.x)));
    if (c.zz && oxy) return slab_max_min(tn.x, tn.y, tf.x, tf.y, tMin) && !(c.cull && skip_far_on_axis(c.o.z, c.d.z, bmin.z, bmax.z, tMin, fminf(tf.x, tf.y)));
    if (oyz && ozx && oxy)
        return slab_max_min(fmaxf(tn.x, tn.y), tn.z, fminf(tf.x, tf.y), tf.z, tMin);
    return false;
}

// intersectTriangle (src/intersections.h:17-54) on a pre-differenced triangle record
// SIGNBIT: `if (det < 0) { det = -det; t = -t; }` as sign-bit arithmetic (5 vector instructions instead of 9 in the packet walks;
// the per-lane shadow-ray walk is faster with the branch-free selects, so it keeps them): |det| >= FLT_EPSILON at that point, so
// det < 0 is its sign bit, and a NaN determinant fails every comparison below whatever the sign of t.
template <bool SIGNBIT = false>
RS_HD bool tri_hit(f3 o, f3 d, f3 v0, f3 e01, f3 e02, float& bx, float& by, float& dist) {
    f3 p = cross(d, e02);
    float det = dot(p, e01);
    if (gabs(det) < 1.1920928955078125e-7f) return false;       // FLT_EPSILON
    f3 t = o - v0;
    if (SIGNBIT) {
        const unsigned flip = __builtin_bit_cast(unsigned, det) & 0x80000000u;
        det = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, det) ^ flip);
        t = mk3(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, t.x) ^ flip), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t.y) ^ flip),
                __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t.z) ^ flip));
    }
    else if (det < 0.f) { det = -det; t = -t; }
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
// ---- traversal ---------------------------------------------------------------------------------
// All six threaded orders live in ONE array (order k at records [k*bvhSize, (k+1)*bvhSize), one padding
// record at the very end), so a lane addresses its node with a 32-bit byte offset from a wave-uniform
// base: the loads compile to `global_load_dwordx4 v, v_off, s[base]` (no 64-bit address arithmetic),
// and the two possible successors of a node can be prefetched before its box test has finished:
//   * entered  -> the next record in memory (pre-order layout: first child / next sibling)
//   * rejected -> nextNodeIfMiss, which is part of the record just loaded
// Both are requested at the top of the step, so the ~40-instruction slab test of node n overlaps the
// memory latency of node n+1 whichever way the test goes.  The walk itself is unchanged: same nodes,
// same order, same arithmetic as DevScene::intersect / testOcclusion (src/scene.h:245-316).

__device__ __forceinline__ float4 ld16(const char* base, unsigned off) {
    return *reinterpret_cast<const float4*>(base + off);
}
// the two 16-byte halves of a BvhNode as loaded -> lo = {min.xyz, bits(primId)}, hi = {max.xyz, bits(next)}
__device__ __forceinline__ void node_unpack(const float4& ra, const float4& rb, float4& lo, float4& hi) {
    lo = make_float4(ra.x, ra.y, ra.z, rb.z);
    hi = make_float4(rb.x, rb.y, ra.w, rb.w);
}

// General-case slab test (bvh.h:124-156 with none of the special cases): valid when every
// |d.c| is in [1e-6, 1-1e-6].  Then all t are finite, so glm::min/max equal fminf/fmaxf up to the
// sign of a zero, which no comparison below can see.
__device__ __forceinline__ bool box_hit_general(f3 o, f3 dinv, float4 lo, float4 hi, float& tMin) {
    const float t1x = (lo.x - o.x) * dinv.x, t1y = (lo.y - o.y) * dinv.y, t1z = (lo.z - o.z) * dinv.z;
    const float t2x = (hi.x - o.x) * dinv.x, t2y = (hi.y - o.y) * dinv.y, t2z = (hi.z - o.z) * dinv.z;
    const float nx = fminf(t1x, t2x), ny = fminf(t1y, t2y), nz = fminf(t1z, t2z);
    const float fx = fmaxf(t1x, t2x), fy = fmaxf(t1y, t2y), fz = fmaxf(t1z, t2z);
    const float dx = fx - nx, dy = fy - ny, dz = fz - nz;
    const bool overlap = (dy + dz > fz - ny) & (dz + dx > fx - nz) & (dx + dy > fy - nx);
    tMin = fmaxf(fmaxf(nx, ny), nz);
    const float tMax = fminf(fminf(fx, fy), fz);
    return overlap & (tMax >= 0.f) & (tMax >= tMin);
}

__device__ __forceinline__ void load_tri(const TriRec* tris, int prim, f3& v0, f3& e1, f3& e2) {
    const float4* p = reinterpret_cast<const float4*>(tris + prim);
    float4 a = p[0], b = p[1], c = p[2];
    v0 = mk3(a.x, a.y, a.z); e1 = mk3(b.x, b.y, b.z); e2 = mk3(c.x, c.y, c.z);
}

struct WalkResult {
    float closest; int prim; float bx, by; bool any;
    unsigned nodes;            // packet walks: nodes the WAVE visited (the union of its lanes' walks), wave-uniform
#ifdef RS_WALK_STATS
    unsigned steps, nearSteps, enteredSteps, leafSteps, clearSteps;   // clearSteps: entered without evaluating the overlap part
    unsigned myVisits;                                                // nodes this lane's own walk visited (packet walks)
#endif
};

// One per-lane MTBVH walk.  ANYHIT: stop at the first triangle closer than `limit` (testOcclusion);
// otherwise keep the closest (intersect).  GENERAL: every lane of the wave is a general-case ray.
template <bool ANYHIT, bool GENERAL>
__device__ __forceinline__ WalkResult walk(const DevScene& s, const Ray& ray, const RayBoxCtx& ctx, float limit) {
    WalkResult r;
    r.closest = limit; r.prim = kNullPrim; r.bx = 0.f; r.by = 0.f; r.any = false;
    const char* base = reinterpret_cast<const char*>(s.nodesAll);
    const unsigned first = (unsigned)mtbvh_order(-ray.d) * (unsigned)s.bvhSize * 32u;
    const unsigned endOff = first + (unsigned)s.bvhSize * 32u;
    unsigned cur = first;
    while (cur != endOff) {
        float4 lo, hi;
        node_unpack(ld16(base, cur), ld16(base, cur + 16), lo, hi);
        float tb;
        bool bh;
        if (GENERAL) bh = box_hit_general(ctx.o, ctx.dinv, lo, hi, tb);
        else bh = box_hit(ctx, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), tb);
        if (bh && tb < r.closest) {
            const int prim = __float_as_int(lo.w);
            if (prim != kNullPrim) {
                f3 v0, e1, e2;
                load_tri(s.tris, prim, v0, e1, e2);
                float bx, by, dist;
                if (tri_hit(ray.o, ray.d, v0, e1, e2, bx, by, dist) && dist < r.closest) {
                    if (ANYHIT) { r.any = true; return r; }
                    r.closest = dist; r.bx = bx; r.by = by; r.prim = prim;
                }
            }
            cur += 32u;
        }
        else {
            cur = first + (unsigned)__float_as_int(hi.w) * 32u;
        }
    }
    return r;
}

// ---- pair-cooperative node fetch for incoherent rays ---------------------------------------------
// Measured on the per-lane walk with shadow rays (profiles/): the L1 can look up one cache line per
// clock, and a wave of incoherent rays touches ~64 different lines in EACH of the two 16-byte loads
// of a step (TD/TA busy 92 %).  Here lanes 2k and 2k+1 fetch together: one load instruction reads
// both halves of the even lane's node (one line), the next both halves of the odd lane's node, and a
// DPP quad-permute hands each lane the half it is missing -- the same 32 bytes per lane, half the
// line look-ups.  Every lane still walks exactly its own node sequence with the same arithmetic.
// Must be called by all 64 lanes (`active` false for lanes without a ray): finished lanes keep
// fetching for their partner.
__device__ __forceinline__ int dpp_swap1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float4 dpp_swap1(float4 v) {
    return make_float4(__int_as_float(dpp_swap1(__float_as_int(v.x))), __int_as_float(dpp_swap1(__float_as_int(v.y))),
                       __int_as_float(dpp_swap1(__float_as_int(v.z))), __int_as_float(dpp_swap1(__float_as_int(v.w))));
}

template <bool ANYHIT, bool GENERAL>
__device__ __forceinline__ WalkResult walk_paired(const DevScene& s, const Ray& ray, const RayBoxCtx& ctx, float limit, bool active) {
    WalkResult r;
    r.closest = limit; r.prim = kNullPrim; r.bx = 0.f; r.by = 0.f; r.any = false;
    const char* base = reinterpret_cast<const char*>(s.nodesAll);
    const unsigned first = (unsigned)mtbvh_order(-ray.d) * (unsigned)s.bvhSize * 32u;
    const unsigned endOff = first + (unsigned)s.bvhSize * 32u;
    const bool odd = (__lane_id() & 1u) != 0;
    const unsigned halfOff = odd ? 16u : 0u;
    unsigned cur = active ? first : endOff;            // endOff is a readable record (next order / padding)
#ifdef RS_WALK_STATS
    unsigned long long pst[4] = { 1, 0, 0, 0 };
#endif
    while (__any(cur != endOff)) {
#ifdef RS_WALK_STATS
        pst[1]++; pst[2] += __popcll(__ballot(cur != endOff));
#endif
        const unsigned partner = (unsigned)dpp_swap1((int)cur);
        const float4 r1 = ld16(base, (odd ? partner : cur) + halfOff);      // even lane's node, split over the pair
        const float4 r2 = ld16(base, (odd ? cur : partner) + halfOff);      // odd lane's node
        const float4 s1 = dpp_swap1(r1), s2 = dpp_swap1(r2);
        float4 lo, hi;
        node_unpack(odd ? s2 : r1, odd ? r2 : s1, lo, hi);
        if (cur != endOff) {
            float tb;
            bool bh;
            if (GENERAL) bh = box_hit_general(ctx.o, ctx.dinv, lo, hi, tb);
            else bh = box_hit(ctx, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), tb);
            if (bh && tb < r.closest) {
                const int prim = __float_as_int(lo.w);
                cur += 32u;
                if (prim != kNullPrim) {
                    f3 v0, e1, e2;
                    load_tri(s.tris, prim, v0, e1, e2);
                    float bx, by, dist;
                    if (tri_hit(ray.o, ray.d, v0, e1, e2, bx, by, dist) && dist < r.closest) {
                        if (ANYHIT) { r.any = true; cur = endOff; }
                        else { r.closest = dist; r.bx = bx; r.by = by; r.prim = prim; }
                    }
                }
            }
            else {
                cur = first + (unsigned)__float_as_int(hi.w) * 32u;
            }
        }
    }
#ifdef RS_WALK_STATS
    if (s.walkStats && __lane_id() == 0 && !ANYHIT) for (int i = 0; i < 3; i++) atomicAdd(&s.walkStats[80 + i], pst[i]);
#endif
    return r;
}

template <bool ANYHIT>
__device__ __forceinline__ WalkResult walk_dispatch(const DevScene& s, const Ray& ray, float limit) {
    RayBoxCtx ctx = make_box_ctx(ray);
    ctx.cull = s.axisCull;
    const bool special = ctx.mode != 0 || ctx.zx || ctx.zy || ctx.zz || !(ray.d.x == ray.d.x);
    // the special cases are ~1e-6 of the rays: a wave that has none runs the branch-free test
    if (__any(special)) return walk<ANYHIT, false>(s, ray, ctx, limit);
    return walk<ANYHIT, true>(s, ray, ctx, limit);
}

// ---- any-hit walk with deferred, batched leaf tests ------------------------------------------------
// Measured (profiles/): a vector-memory instruction occupies the return path for ~26 cycles however
// few lanes are active, and in the any-hit walk more than half of all load instructions were the
// three 16-byte triangle loads of a leaf visit, each issued for the one or two lanes that happened to
// sit on a leaf in that step.  testOcclusion only asks whether ANY visited triangle is hit closer than
// the limit, and the walk past a leaf does not depend on that leaf's outcome, so a lane may queue the
// leaf and keep walking.  Queued leaves are tested in rounds in which every lane with a pending leaf
// takes part; a lane that finds a hit is occluded and stops.  The set of triangles tested is the
// reference's set (src/scene.h:286-316) up to its first hit plus possibly a few later ones, so the
// boolean is identical.  Rounds run when a queue is full or when no lane can walk on (A/B: triggering
// earlier, once 12/24/40 lanes wait, was 2-5 % slower).  Node fetches are pair-cooperative as in walk_paired.
constexpr int kLeafQueue = 4;

template <bool GENERAL>
__device__ __forceinline__ bool walk_anyhit_deferred(const DevScene& s, const Ray& ray, const RayBoxCtx& ctx, float limit, bool active) {
    const char* base = reinterpret_cast<const char*>(s.nodesAll);
    const unsigned first = (unsigned)mtbvh_order(-ray.d) * (unsigned)s.bvhSize * 32u;
    const unsigned endOff = first + (unsigned)s.bvhSize * 32u;
    const bool odd = (__lane_id() & 1u) != 0;
    const unsigned halfOff = odd ? 16u : 0u;
    unsigned cur = active ? first : endOff;
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;        // LIFO of queued leaf primitives
    bool occluded = false;
    for (;;) {
        const bool walking = cur != endOff;
        const unsigned long long wmask = __ballot(walking);
        const unsigned long long pmask = __ballot(qn > 0);
        if (!(wmask | pmask)) break;
        const bool round = __any(qn == kLeafQueue) || wmask == 0;
        if (round) {
            if (qn > 0) {
                const int prim = q0;
                q0 = q1; q1 = q2; q2 = q3; qn--;
                f3 v0, e1, e2;
                load_tri(s.tris, prim, v0, e1, e2);
                float bx, by, dist;
                if (tri_hit(ray.o, ray.d, v0, e1, e2, bx, by, dist) && dist < limit) { occluded = true; cur = endOff; qn = 0; }
            }
            continue;
        }
        const unsigned partner = (unsigned)dpp_swap1((int)cur);
        const float4 r1 = ld16(base, (odd ? partner : cur) + halfOff);
        const float4 r2 = ld16(base, (odd ? cur : partner) + halfOff);
        const float4 s1 = dpp_swap1(r1), s2 = dpp_swap1(r2);
        float4 lo, hi;
        node_unpack(odd ? s2 : r1, odd ? r2 : s1, lo, hi);
        if (walking) {
            float tb;
            bool bh;
            if (GENERAL) bh = box_hit_general(ctx.o, ctx.dinv, lo, hi, tb);
            else bh = box_hit(ctx, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), tb);
            if (bh && tb < limit) {
                const int prim = __float_as_int(lo.w);
                cur += 32u;
                if (prim != kNullPrim) { q3 = q2; q2 = q1; q1 = q0; q0 = prim; qn++; }
            }
            else {
                cur = first + (unsigned)__float_as_int(hi.w) * 32u;
            }
        }
    }
    return occluded;
}

// ---- shadow rays through the second tree ------------------------------------------------------------
// testOcclusion (src/scene.h:286-316) is true iff some triangle T has (a) every node on the reference
// tree's path to T passing the reference's box test with tBox < range and (b) intersectTriangle(T) closer
// than range; the visiting order is irrelevant.  The reference's tree costs 86 node visits per shadow
// ray on the Sponza-class scene (its SAH sweep is not cumulative, src/bvh.cpp:92-100) at 32 bytes each,
// so general-case rays look for triangles with (b) in a well-built tree of 16-byte nodes over the SAME
// leaf boxes (occlusion_bvh.cpp shows why its relaxed slab test cannot miss a triangle whose reference
// leaf box the ray passes) and then evaluate (a) for such a candidate literally: the reference's box test
// on T's leaf and on each of its ancestors (parent links by original node id).  The first candidate that
// passes is what the reference's walk would also have reached and hit -> occluded; if none passes the
// reference reports no occlusion either.  A lane is in one of two modes:
// Three phases alternate until no lane has work left:
//   walk   : cur = byte offset in occNodes; relaxed test on the grid box, branch-free step; leaves are
//            queued (as in walk_anyhit_deferred); ends when a queue is full or all walks have ended
//   leaves : every lane tests the triangles of its newest queued leaf
//   verify : lanes with a candidate run the reference's test along occChain; pass -> occluded,
//            fail -> the lane walks on
// Only for rays that take none of AABB::intersect's special cases (all |d.c| in [1e-6, 1-1e-6]) and start
// within 4 grid extents of the scene (the error bound of the grid test, occlusion_bvh.cpp).
__device__ __forceinline__ bool occlusion_tree_usable(const DevScene& s, f3 o) {
    const float reach = 4.f * 65535.f;
    return gabs(o.x - s.occBase.x) <= reach * s.occScale.x && gabs(o.y - s.occBase.y) <= reach * s.occScale.y &&
           gabs(o.z - s.occBase.z) <= reach * s.occScale.z;
}

__device__ __forceinline__ bool walk_occlusion_tree(const DevScene& s, const Ray& ray, const RayBoxCtx& ctx, float limit, bool active) {
    const char* nodes = reinterpret_cast<const char*>(s.occNodes);
    const unsigned endOff = (unsigned)s.occCount * 16u;
    // slab distance of grid plane q: (base + q*scale - o) / d = q * A + B
    // A lane that enters without a ray of its own (outside the frame, a special-case or far-origin ray that takes the reference walk)
    // is parked on the sentinel record past the end for the whole walk; it evaluates that record like every other lane, so its
    // slab distances must fail the test whatever its ray is: q * 0 + (-1) gives tMax = -1 < 0.  (With its own A and B a ray
    // 2^24 grid extents away would absorb q * A in B, pass the empty box and step beyond the allocation.)
    const f3 A = active ? mk3(s.occScale.x * ctx.dinv.x, s.occScale.y * ctx.dinv.y, s.occScale.z * ctx.dinv.z) : splat(0.f);
    const f3 B = active ? mk3((s.occBase.x - ctx.o.x) * ctx.dinv.x, (s.occBase.y - ctx.o.y) * ctx.dinv.y, (s.occBase.z - ctx.o.z) * ctx.dinv.z) : splat(-1.f);
    // largest |slab distance| of the reference's root box
    const float tRoot = fmaxf(fmaxf(fmaxf(gabs((s.occRootLo.x - ctx.o.x) * ctx.dinv.x), gabs((s.occRootHi.x - ctx.o.x) * ctx.dinv.x)),
                                    fmaxf(gabs((s.occRootLo.y - ctx.o.y) * ctx.dinv.y), gabs((s.occRootHi.y - ctx.o.y) * ctx.dinv.y))),
                              fmaxf(gabs((s.occRootLo.z - ctx.o.z) * ctx.dinv.z), gabs((s.occRootHi.z - ctx.o.z) * ctx.dinv.z)));
    unsigned cur = active ? 0u : endOff;
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;                 // LIFO of queued leaf codes
    bool occluded = false;
    // Which of the two grid planes of an axis is the near one depends on the sign of A only (fma is monotone in q): a byte
    // permute with a per-ray selector puts {near plane, far plane} of an axis into one dword, and the six min / max of the slab
    // test are gone.  Node dwords: x = lo.x | lo.y << 16, y = lo.z | hi.x << 16, z = hi.y | hi.z << 16.
    const unsigned selX = A.x < 0.f ? 0x01000706u : 0x07060100u;      // v_perm_b32(n.y, n.x): bytes 0-3 = n.x, 4-7 = n.y
    const unsigned selY = A.y < 0.f ? 0x03020504u : 0x05040302u;      // v_perm_b32(n.z, n.x)
    const unsigned selZ = A.z < 0.f ? 0x01000706u : 0x07060100u;      // v_perm_b32(n.z, n.y)
    const vf2 Axy = { A.x, A.y }, Bxy = { B.x, B.y }, Azz = { A.z, A.z }, Bzz = { B.z, B.z };
#ifdef RS_WALK_STATS
    unsigned long long wst[10] = { 1, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int mySteps = 0, myTris = 0;         // of this lane's ray
#define RS_STAT(i, v) wst[i] += (v)
#else
#define RS_STAT(i, v)
#endif
    for (;;) {
        // walk phase: a tight loop until some lane's leaf queue is full or every walk has ended
        for (;;) {
            const unsigned long long walkers = __ballot(cur != endOff);
            if (!walkers) break;
            RS_STAT(1, 1); RS_STAT(5, 1); RS_STAT(6, __popcll(__ballot(cur != endOff)));
#ifdef RS_WALK_STATS
            if (cur != endOff) mySteps++;
#ifdef RS_WALK_STATS_TIME      // instead of the depth histogram: walking lanes and wave iterations by iteration index (buckets of 24)
            { const unsigned long long walkers = __ballot(cur != endOff);
              if (s.walkStats && __lane_id() == 0) { const int b = wst[5] / 24 < 9 ? (int)(wst[5] / 24) : 9; atomicAdd(&s.walkStats[44 + b], (unsigned long long)__popcll(walkers)); atomicAdd(&s.walkStats[54 + b], 1ull); } }
#endif
#endif
            {   // every lane, also one whose walk has ended: it reads the record past the end, an empty box linked to itself (scene.hip)
                const uint4 n = *reinterpret_cast<const uint4*>(nodes + cur);
#if defined(RS_WALK_STATS) && !defined(RS_WALK_STATS_TIME)
                if (s.walkStats && s.occDepth && cur != endOff) { const int dep = s.occDepth[cur >> 4]; atomicAdd(&s.walkStats[44 + (dep < 19 ? dep : 19)], 1ull); }
#endif
                const unsigned px = __builtin_amdgcn_perm(n.y, n.x, selX), py = __builtin_amdgcn_perm(n.z, n.x, selY), pz = __builtin_amdgcn_perm(n.z, n.y, selZ);
                const vf2 nearXY = __builtin_elementwise_fma(vf2{ (float)(px & 0xffffu), (float)(py & 0xffffu) }, Axy, Bxy);
                const vf2 farXY = __builtin_elementwise_fma(vf2{ (float)(px >> 16), (float)(py >> 16) }, Axy, Bxy);
                const vf2 zNF = __builtin_elementwise_fma(vf2{ (float)(pz & 0xffffu), (float)(pz >> 16) }, Azz, Bzz);
                const float tMin = fmaxf(fmaxf(nearXY.x, nearXY.y), zNF.x);
                const float tMax = fminf(fminf(farXY.x, farXY.y), zNF.y);
                const bool pass = (tMax >= fmaxf(tMin, 0.f)) && (tMin < limit);
                const int meta = (int)n.w;
                const bool leaf = meta < 0;
                const bool push = pass && leaf;
                q3 = push ? q2 : q3; q2 = push ? q1 : q2; q1 = push ? q0 : q1; q0 = push ? ~meta : q0; qn = push ? qn + 1 : qn;
                cur = (pass || leaf) ? cur + 16u : (unsigned)meta;
            }
            if (__any(qn == kLeafQueue)) break;
        }
        if (!__any(qn > 0)) break;
        // leaf round: every lane takes its newest queued leaf (so no queue is full when the walk resumes) and
        // tests its triangles; a hit becomes a candidate, and the rest of the leaf waits for its verdict
        RS_STAT(2, 1);
        int tri = 0, cnt = 0, verify = -1;
        if (qn > 0) { tri = q0 >> 3; cnt = q0 & 7; q0 = q1; q1 = q2; q2 = q3; qn--; }
        for (;;) {
            while (__any((cnt > 0) & (verify < 0))) {
                RS_STAT(3, 1); RS_STAT(9, __popcll(__ballot((cnt > 0) & (verify < 0))));
                if ((cnt > 0) & (verify < 0)) {
                    const float4* p = reinterpret_cast<const float4*>(s.occTris + tri);
                    const float4 a = p[0], b = p[1], c = p[2];
                    float bx, by, dist;
                    tri++; cnt--;
#ifdef RS_WALK_STATS
                    myTris++;
#endif
                    if (tri_hit(ray.o, ray.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), bx, by, dist) && dist < limit)
                        verify = __float_as_int(a.w);          // reference leaf of the candidate
                }
            }
            if (!__any(verify >= 0)) break;
            // candidates: the reference's own test along the path to the triangle's leaf (normally one step, see below)
            while (__any(verify >= 0)) {
                RS_STAT(4, 1); RS_STAT(7, __popcll(__ballot(verify >= 0)));
                if (verify >= 0) {
                    const float4* rec = reinterpret_cast<const float4*>(s.occChain + verify);
                    float4 lo, hi;
                    node_unpack(rec[0], rec[1], lo, hi);
                    // the general case of AABB::intersect (box_hit_general), spelled out for the margins below
                    const float t1x = (lo.x - ctx.o.x) * ctx.dinv.x, t1y = (lo.y - ctx.o.y) * ctx.dinv.y, t1z = (lo.z - ctx.o.z) * ctx.dinv.z;
                    const float t2x = (hi.x - ctx.o.x) * ctx.dinv.x, t2y = (hi.y - ctx.o.y) * ctx.dinv.y, t2z = (hi.z - ctx.o.z) * ctx.dinv.z;
                    const float nx = fminf(t1x, t2x), ny = fminf(t1y, t2y), nz = fminf(t1z, t2z);
                    const float fx = fmaxf(t1x, t2x), fy = fmaxf(t1y, t2y), fz = fmaxf(t1z, t2z);
                    const float dx = fx - nx, dy = fy - ny, dz = fz - nz;
                    const bool overlap = (dy + dz > fz - ny) & (dz + dx > fx - nz) & (dx + dy > fy - nx);
                    const float tMin = fmaxf(fmaxf(nx, ny), nz), tMax = fminf(fminf(fx, fy), fz);
                    const bool open = overlap & (tMax >= 0.f) & (tMax >= tMin) & (tMin < limit);
                    // Shortcut at the leaf (first record of a chain).  Every ancestor box contains the leaf box
                    // (checked at scene build), so by monotone rounding its near distances are <= and its far
                    // distances >= the leaf's: tMax >= 0, tMax >= tMin and tMin < range carry over exactly.  The
                    // three overlap conditions are, in real arithmetic, fy > nz, fz > nx, fx > ny, and those
                    // differences can only grow towards the root; evaluated in float they are off by less than
                    // 2^-20 * tRoot (four roundings of values below 4 * tRoot, tRoot = largest |slab distance| of
                    // the root box, which bounds every ancestor's).  A leaf that clears them by 2^-18 * tRoot
                    // therefore settles the whole path; otherwise the ancestors are tested one by one -- and the same
                    // argument holds from ANY node of the path upwards, so the first ancestor that clears them ends
                    // the walk (thin leaf boxes in a long scene: a Bistro-class chain took 30 steps to the root).
                    const bool clear = fminf(fminf(fy - nz, fz - nx), fx - ny) > tRoot * 3.814697265625e-6f;
                    const int parent = __float_as_int(lo.w);
                    const bool done = open & ((parent < 0) | (clear & s.occNested));
                    if (done) { occluded = true; cur = endOff; qn = 0; cnt = 0; }
                    verify = (open & !done) ? parent : -1;          // closed: the reference never reaches the triangle
                }
            }
        }
    }
#ifdef RS_WALK_STATS
    if (s.walkStats && __lane_id() == 0) for (int i = 0; i < 10; i++) atomicAdd(&s.walkStats[i], wst[i]);
    if (s.walkStats && active) {        // per ray: [10] occluded rays, [11] their steps, [12] unoccluded rays, [13] their steps, [14] triangle tests of all
        atomicAdd(&s.walkStats[occluded ? 10 : 12], 1ull); atomicAdd(&s.walkStats[occluded ? 11 : 13], (unsigned long long)mySteps);
        atomicAdd(&s.walkStats[14], (unsigned long long)myTris);
    }
#endif
#undef RS_STAT
    return occluded;
}

// May the reference's closest hit of this ray be an EMISSIVE triangle?  False only if the ray hits (intersectTriangle) no emissive
// triangle whose reference leaf box it passes -- then DevScene::intersect, which accepts a triangle only on those two conditions, cannot
// return one.  The walk is walk_occlusion_tree's on the tree of the emissive triangles (no range, no verification: any hit answers "maybe");
// special-case and far-origin rays answer "maybe" without walking.  Every lane of the wave must call it.
__device__ __forceinline__ bool may_hit_emissive_wave(const DevScene& s, const Ray& ray, bool active) {
    if (!s.emiState) return active;
    if (s.emiCount == 0) return false;
    RayBoxCtx ctx = make_box_ctx(ray);
    const bool special = ctx.mode != 0 || ctx.zx || ctx.zy || ctx.zz || !(ray.d.x == ray.d.x);
    const float reach = 4.f * 65535.f;
    const bool usable = gabs(ray.o.x - s.emiBase.x) <= reach * s.emiScale.x && gabs(ray.o.y - s.emiBase.y) <= reach * s.emiScale.y &&
                        gabs(ray.o.z - s.emiBase.z) <= reach * s.emiScale.z;
    const bool maybe = active && (special || !usable);          // answered without a walk
    const bool walks = active && !maybe;
    const char* nodes = reinterpret_cast<const char*>(s.emiNodes);
    const unsigned endOff = (unsigned)s.emiCount * 16u;
    const f3 A = walks ? mk3(s.emiScale.x * ctx.dinv.x, s.emiScale.y * ctx.dinv.y, s.emiScale.z * ctx.dinv.z) : splat(0.f);
    const f3 B = walks ? mk3((s.emiBase.x - ctx.o.x) * ctx.dinv.x, (s.emiBase.y - ctx.o.y) * ctx.dinv.y, (s.emiBase.z - ctx.o.z) * ctx.dinv.z) : splat(-1.f);
    unsigned cur = walks ? 0u : endOff;
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;
    bool found = false;
    const unsigned selX = A.x < 0.f ? 0x01000706u : 0x07060100u, selY = A.y < 0.f ? 0x03020504u : 0x05040302u, selZ = A.z < 0.f ? 0x01000706u : 0x07060100u;
    const vf2 Axy = { A.x, A.y }, Bxy = { B.x, B.y }, Azz = { A.z, A.z }, Bzz = { B.z, B.z };
    for (;;) {
        for (;;) {
            if (!__ballot(cur != endOff)) break;
            const uint4 n = *reinterpret_cast<const uint4*>(nodes + cur);
            const unsigned px = __builtin_amdgcn_perm(n.y, n.x, selX), py = __builtin_amdgcn_perm(n.z, n.x, selY), pz = __builtin_amdgcn_perm(n.z, n.y, selZ);
            const vf2 nearXY = __builtin_elementwise_fma(vf2{ (float)(px & 0xffffu), (float)(py & 0xffffu) }, Axy, Bxy);
            const vf2 farXY = __builtin_elementwise_fma(vf2{ (float)(px >> 16), (float)(py >> 16) }, Axy, Bxy);
            const vf2 zNF = __builtin_elementwise_fma(vf2{ (float)(pz & 0xffffu), (float)(pz >> 16) }, Azz, Bzz);
            const float tMin = fmaxf(fmaxf(nearXY.x, nearXY.y), zNF.x);
            const float tMax = fminf(fminf(farXY.x, farXY.y), zNF.y);
            const bool pass = tMax >= fmaxf(tMin, 0.f);
            const int meta = (int)n.w;
            const bool leaf = meta < 0;
            const bool push = pass && leaf;
            q3 = push ? q2 : q3; q2 = push ? q1 : q2; q1 = push ? q0 : q1; q0 = push ? ~meta : q0; qn = push ? qn + 1 : qn;
            cur = (pass || leaf) ? cur + 16u : (unsigned)meta;
            if (__any(qn == kLeafQueue)) break;
        }
        if (!__any(qn > 0)) break;
        int tri = 0, cnt = 0;
        if (qn > 0) { tri = q0 >> 3; cnt = q0 & 7; q0 = q1; q1 = q2; q2 = q3; qn--; }
        while (__any(cnt > 0)) {
            if (cnt > 0) {
                const float4* p = reinterpret_cast<const float4*>(s.emiTris + tri);
                const float4 a = p[0], b = p[1], c = p[2];
                float bx, by, dist;
                tri++; cnt--;
                if (tri_hit(ray.o, ray.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), bx, by, dist)) { found = true; cur = endOff; qn = 0; cnt = 0; }
            }
        }
    }
    return maybe || found;
}

// ---- closest hit of incoherent rays through the trees that keep the reference's order ---------------------------------------
// DevScene::intersect (src/scene.h:245-284) accepts a triangle iff its leaf is entered -- the reference's box test passes on the
// leaf (and, before it, on every ancestor) with tBox < closest AT THAT MOMENT -- and intersectTriangle hits closer than closest at
// that moment: the result depends on the order in which the walk meets the triangles.  The tree walked here (occlusion_bvh.cpp
// rs_build_ordered_bvh; one per threaded order) meets them in the reference's order through better boxes on the shadow tree's
// grid, and the rule is applied literally:
//   * a node is entered iff the relaxed test passes with tBox' < closest.  tBox' <= tLeaf of every triangle below (conservative
//     boxes, occlusion_bvh.cpp), so a skipped node holds only leaves the reference would not enter at this `closest` or any later one;
//   * a triangle hit closer than `closest` is a candidate; it is accepted iff the reference's own test passes along the path to
//     its leaf with tLeaf < closest (the chain check of the shadow rays, leaf shortcut included) -- exactly when the reference
//     reaches it.  Boxes are nested (occNested is a precondition), so tLeaf bounds the ancestors' entry distances.
// Leaves are queued and tested in wave-wide rounds as in walk_occlusion_tree, but FIRST IN, FIRST OUT and one candidate at a
// time, so that a lane's triangles are judged in the reference's order with the reference's `closest`; a walk that runs ahead of
// its queue only uses a staler (larger) `closest`, i.e. enters more, never less.  Same primitive, same barycentrics, same bits as
// walk<false, ...>; tested against it and against the oracle on the full scenes.
// Only for general-case rays that start within the grid's reach (as the shadow walk); every lane of the wave must call it.
#ifndef RS_ORD_QUEUE
#define RS_ORD_QUEUE 4          // leaves a lane may queue before the wave runs a leaf round (1..4)
#endif
__device__ __forceinline__ WalkResult walk_ordered_tree(const DevScene& s, const Ray& ray, const RayBoxCtx& ctx, bool active) {
    WalkResult r;
    r.closest = 3.402823466e+38f; r.prim = kNullPrim; r.bx = 0.f; r.by = 0.f; r.any = false;
    const char* nodes = reinterpret_cast<const char*>(s.ordNodes);
    const unsigned k = active ? (unsigned)mtbvh_order(-ray.d) : 0u;
    const unsigned endOff = (k + 1u) * s.ordStride - 16u;
    const TriRec* tris = s.ordTris + (size_t)(k >> 1) * (size_t)s.numPrims;
    const int triStep = (k & 1u) ? -1 : 1;
    // slab distance of grid plane q: q * A + B; a lane without a ray rests on the end record with distances that fail whatever the record is
    const f3 A = active ? mk3(s.occScale.x * ctx.dinv.x, s.occScale.y * ctx.dinv.y, s.occScale.z * ctx.dinv.z) : splat(0.f);
    const f3 B = active ? mk3((s.occBase.x - ctx.o.x) * ctx.dinv.x, (s.occBase.y - ctx.o.y) * ctx.dinv.y, (s.occBase.z - ctx.o.z) * ctx.dinv.z) : splat(-1.f);
    const float tRoot = fmaxf(fmaxf(fmaxf(gabs((s.occRootLo.x - ctx.o.x) * ctx.dinv.x), gabs((s.occRootHi.x - ctx.o.x) * ctx.dinv.x)),
                                    fmaxf(gabs((s.occRootLo.y - ctx.o.y) * ctx.dinv.y), gabs((s.occRootHi.y - ctx.o.y) * ctx.dinv.y))),
                              fmaxf(gabs((s.occRootLo.z - ctx.o.z) * ctx.dinv.z), gabs((s.occRootHi.z - ctx.o.z) * ctx.dinv.z)));
    unsigned cur = active ? k * s.ordStride : endOff;
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;                 // FIFO of queued leaf codes, q0 the oldest
    const unsigned selX = A.x < 0.f ? 0x01000706u : 0x07060100u;      // near / far plane of an axis by a per-ray byte permute (walk_occlusion_tree)
    const unsigned selY = A.y < 0.f ? 0x03020504u : 0x05040302u;
    const unsigned selZ = A.z < 0.f ? 0x01000706u : 0x07060100u;
    const vf2 Axy = { A.x, A.y }, Bxy = { B.x, B.y }, Azz = { A.z, A.z }, Bzz = { B.z, B.z };
#ifdef RS_WALK_STATS
    unsigned long long st[12] = { 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int mySteps = 0;
#define RS_OSTAT(i, v) st[i] += (v)
#else
#define RS_OSTAT(i, v)
#endif
    for (;;) {
        for (;;) {          // walk phase: until some lane's queue is full or every walk has ended
            if (!__ballot(cur != endOff)) break;
            RS_OSTAT(1, 1); RS_OSTAT(2, __popcll(__ballot(cur != endOff)));
#ifdef RS_WALK_STATS
            if (cur != endOff) mySteps++;
#endif
            const uint4 n = *reinterpret_cast<const uint4*>(nodes + cur);
            const unsigned px = __builtin_amdgcn_perm(n.y, n.x, selX), py = __builtin_amdgcn_perm(n.z, n.x, selY), pz = __builtin_amdgcn_perm(n.z, n.y, selZ);
            const vf2 nearXY = __builtin_elementwise_fma(vf2{ (float)(px & 0xffffu), (float)(py & 0xffffu) }, Axy, Bxy);
            const vf2 farXY = __builtin_elementwise_fma(vf2{ (float)(px >> 16), (float)(py >> 16) }, Axy, Bxy);
            const vf2 zNF = __builtin_elementwise_fma(vf2{ (float)(pz & 0xffffu), (float)(pz >> 16) }, Azz, Bzz);
            const float tMin = fmaxf(fmaxf(nearXY.x, nearXY.y), zNF.x);
            const float tMax = fminf(fminf(farXY.x, farXY.y), zNF.y);
            const bool pass = (tMax >= fmaxf(tMin, 0.f)) && (tMin < r.closest);
            const int meta = (int)n.w;
            const bool leaf = meta < 0;
            const bool push = pass && leaf;
            const int code = ~meta;
            q0 = (push && qn == 0) ? code : q0; q1 = (push && qn == 1) ? code : q1; q2 = (push && qn == 2) ? code : q2; q3 = (push && qn == 3) ? code : q3;
            qn = push ? qn + 1 : qn;
            cur = (pass || leaf) ? cur + 16u : (unsigned)meta;
            if (__any(qn == RS_ORD_QUEUE)) break;
        }
        if (!__any(qn > 0)) break;
        {
        // leaf round: every lane takes its OLDEST queued leaf and judges its triangles one after the other
        RS_OSTAT(3, 1);
        int tri = 0, cnt = 0, verify = -1;
        if (qn > 0) { tri = q0 >> 3; cnt = q0 & 7; q0 = q1; q1 = q2; q2 = q3; qn--; }
        float cd = 0.f, cbx = 0.f, cby = 0.f; int cprim = kNullPrim;
        for (;;) {
            while (__any((cnt > 0) & (verify < 0))) {
                RS_OSTAT(4, 1); RS_OSTAT(5, __popcll(__ballot((cnt > 0) & (verify < 0))));
                if ((cnt > 0) & (verify < 0)) {
                    const float4* p = reinterpret_cast<const float4*>(tris + tri);
                    const float4 a = p[0], b = p[1], c = p[2];
                    float bx, by, dist;
                    tri += triStep; cnt--;
                    if (tri_hit(ray.o, ray.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), bx, by, dist) && dist < r.closest) {
                        cd = dist; cbx = bx; cby = by; cprim = __float_as_int(b.w);
                        verify = __float_as_int(a.w);          // reference leaf of the candidate
                    }
                }
            }
            if (!__any(verify >= 0)) break;
            while (__any(verify >= 0)) {
                RS_OSTAT(6, 1); RS_OSTAT(7, __popcll(__ballot(verify >= 0)));
                if (verify >= 0) {
                    const float4* rec = reinterpret_cast<const float4*>(s.occChain + verify);
                    float4 lo, hi;
                    node_unpack(rec[0], rec[1], lo, hi);
                    const float t1x = (lo.x - ctx.o.x) * ctx.dinv.x, t1y = (lo.y - ctx.o.y) * ctx.dinv.y, t1z = (lo.z - ctx.o.z) * ctx.dinv.z;
                    const float t2x = (hi.x - ctx.o.x) * ctx.dinv.x, t2y = (hi.y - ctx.o.y) * ctx.dinv.y, t2z = (hi.z - ctx.o.z) * ctx.dinv.z;
                    const float nx = fminf(t1x, t2x), ny = fminf(t1y, t2y), nz = fminf(t1z, t2z);
                    const float fx = fmaxf(t1x, t2x), fy = fmaxf(t1y, t2y), fz = fmaxf(t1z, t2z);
                    const float dx = fx - nx, dy = fy - ny, dz = fz - nz;
                    const bool overlap = (dy + dz > fz - ny) & (dz + dx > fx - nz) & (dx + dy > fy - nx);
                    const float tMin = fmaxf(fmaxf(nx, ny), nz), tMax = fminf(fminf(fx, fy), fz);
                    const bool open = overlap & (tMax >= 0.f) & (tMax >= tMin) & (tMin < r.closest);
                    // the leaf shortcut of walk_occlusion_tree: a leaf that clears the overlap conditions by 2^-18 * tRoot settles its whole path
                    const bool clear = fminf(fminf(fy - nz, fz - nx), fx - ny) > tRoot * 3.814697265625e-6f;
                    const int parent = __float_as_int(lo.w);
                    const bool done = open & ((parent < 0) | clear);
                    if (done) { r.closest = cd; r.bx = cbx; r.by = cby; r.prim = cprim; }
                    verify = (open & !done) ? parent : -1;          // closed: the reference never reaches the triangle
                }
            }
        }
        }
    }
#ifdef RS_WALK_STATS
    if (s.walkStats && __lane_id() == 0) for (int i = 0; i < 8; i++) atomicAdd(&s.walkStats[64 + i], st[i]);
    if (s.walkStats && active) { atomicAdd(&s.walkStats[72], 1ull); atomicAdd(&s.walkStats[73], (unsigned long long)mySteps); }
#endif
#undef RS_OSTAT
    return r;
}

// all 64 lanes of the wave must call this
template <bool ANYHIT>
__device__ __forceinline__ WalkResult walk_dispatch_paired(const DevScene& s, const Ray& ray, float limit, bool active) {
    RayBoxCtx ctx = make_box_ctx(ray);
    ctx.cull = s.axisCull;
    const bool special = active && (ctx.mode != 0 || ctx.zx || ctx.zy || ctx.zz || !(ray.d.x == ray.d.x));
    if (ANYHIT) {
        WalkResult r;
        r.closest = limit; r.prim = kNullPrim; r.bx = 0.f; r.by = 0.f;
        if (s.occNodes) {
            const bool slow = active && (special || !occlusion_tree_usable(s, ray.o));
            r.any = walk_occlusion_tree(s, ray, ctx, limit, active && !slow);
            if (__any(slow)) r.any = walk_anyhit_deferred<false>(s, ray, ctx, limit, slow) || r.any;
        }
        else
            r.any = __any(special) ? walk_anyhit_deferred<false>(s, ray, ctx, limit, active)
                                   : walk_anyhit_deferred<true>(s, ray, ctx, limit, active);
        return r;
    }
    if (s.ordNodes) {                      // closest hit (limit = FLT_MAX) through the tree of the ray's order; the rare other rays walk the reference's
        const bool slow = active && (special || !occlusion_tree_usable(s, ray.o));
        WalkResult r = walk_ordered_tree(s, ray, ctx, active && !slow);
        if (__any(slow)) {
            const WalkResult r2 = walk_paired<false, false>(s, ray, ctx, limit, slow);
            if (slow) r = r2;
        }
        return r;
    }
    if (__any(special)) return walk_paired<ANYHIT, false>(s, ray, ctx, limit, active);
    return walk_paired<ANYHIT, true>(s, ray, ctx, limit, active);
}

// DevScene::intersect (src/scene.h:245-284): closest hit, stackless threaded walk
__device__ inline Hit trace_closest(const DevScene& s, const Ray& ray) {
    const WalkResult w = walk_dispatch<false>(s, ray, 3.402823466e+38f);   // FLT_MAX
    Hit h;
    h.primId = w.prim;
    h.matId = 0;
    h.pos = splat(0.f);
    h.norm = splat(0.f);
    h.bx = w.bx; h.by = w.by;
    if (w.prim != kNullPrim) {             // getIntersecGeomInfo (scene.h:135-151)
        const float* v = s.vertices + (size_t)w.prim * 9;
        const float* n = s.normals + (size_t)w.prim * 9;
        float wgt = 1.f - w.bx - w.by;
        h.pos = ld3(v + 3) * w.bx + ld3(v + 6) * w.by + ld3(v) * wgt;
        h.norm = normalize(ld3(n + 3) * w.bx + ld3(n + 6) * w.by + ld3(n) * wgt);
        h.matId = s.materialIds[w.prim];
    }
    return h;
}

// ---- wave-cooperative ("packet") closest-hit walk for coherent rays ------------------------------
// Measured on the per-lane walk above (rocprofv3, profiles/): the G-buffer and primary-ray kernels
// are bound by the vector-memory return path (TD busy 87-89 %): every lane fetches its own 32-byte
// node, 2 KiB per wave-step, although the 64 rays of an 8x8 pixel tile visit almost the same nodes
// (union of visited nodes 158 vs 142 for the slowest single ray, one threaded order per tile).
//
// Here the WAVE walks the union once.  All walks of one order move forward through the same array,
// so the wave visits c = min over lanes of "the node I want next"; the node record is fetched ONCE
// through the scalar cache (s_load_dwordx8: 32 B per wave-step instead of 2 KiB), every lane whose
// own walk is at c runs its slab / triangle test against the SGPR-resident record, the others wait.
// Each lane still visits exactly the nodes of DevScene::intersect (src/scene.h:245-284), in the same
// order with the same arithmetic, so results are bit-identical to the per-lane walk.
//
// Must be called by all 64 lanes of the wave (`active` false for lanes without a ray).

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    // butterfly inside rows of 16 (quad_perm xor1, xor2, row_half_mirror, row_mirror), then the two
    // row broadcasts of gfx9; lane 63 ends up with the minimum of all 64 lanes
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- the fast form of the packet walk ---------------------------------------------------------------------------------------
// For general-case rays (none of AABB::intersect's special cases) whose direction has the same component signs in every lane of
// the wave (all but the ~1 % of 8x8 tiles that straddle a sign change), on a proper box table with nested miss links
// (DevScene::axisCull && linksNested).  Same nodes per lane, same order, same arithmetic on the values that decide -- what
// changes is how little the wave does per node.  Measured on the r01 loop: 54 VALU + ~30 SALU instructions per union node, and
// both units issue one instruction per SIMD every four cycles, so the scalar side counts as much as the vector side:
//   * near / far plane of each axis: with lo <= hi, (lo - o) * dinv <= (hi - o) * dinv for dinv > 0 and the other way round for
//     dinv < 0, because IEEE subtraction and multiplication round monotonically -- the reference's glm::min / glm::max
//     (bvh.h:128-129) pick a known operand (same floats up to the sign of a zero, which no comparison sees).  The sign pattern
//     NEG is a template parameter (eight loop bodies, one runs), so the choice costs no instruction at all;
//   * AABB::intersect's general case is overlap & tMax >= 0 & tMax >= tMin (bvh.h:147-153) and the walk then asks
//     tMin < closest.  A conjunction can be evaluated in any order: the distance part comes first (tMax >= max(tMin, 0)), and
//     the three overlap comparisons, which cost as much as everything else, only run when some lane passed it;
//   * no reduction for the next node.  The walks of a wave all move forward through one pre-order array whose miss links nest
//     (link(c) <= link(a) for c inside (a, link(a)), checked by rs_scene_create).  A lane that is not at c waits at a node
//     p > c which it reached by rejecting some a < c, so p = link(a) with c inside a's span, hence link(c) <= p; the lanes
//     at c go to c + 1 (entered) or link(c).  The minimum of all pending targets is therefore c + 1 if any lane entered and
//     link(c) otherwise: one ballot instead of a 64-lane DPP minimum.  For the same reason a lane's own target after a node
//     nobody entered is max(myNext, link(c)) -- one instruction, no mask.
// Slab distances of one record held in SGPRs {min.x, min.y, min.z, max.z | max.x, max.y, prim, next}: three packed subtractions
// and three packed multiplications, each the reference's (p - ori) * dirInv on two components
struct SlabT { vf2 xy1, xy2, z12; };      // (t1.x, t1.y), (t2.x, t2.y), (t1.z, t2.z)
__device__ __forceinline__ SlabT slabs(vf2 oxy, vf2 ozz, vf2 dxy, vf2 dzz, const float4& ra, const float4& rb) {
    SlabT t;
    t.xy1 = (vf2{ ra.x, ra.y } - oxy) * dxy;
    t.xy2 = (vf2{ rb.x, rb.y } - oxy) * dxy;
    t.z12 = (vf2{ ra.z, ra.w } - ozz) * dzz;
    return t;
}
template <int NEG> __device__ __forceinline__ float near_x(const SlabT& t) { return (NEG & 1) ? t.xy2.x : t.xy1.x; }
template <int NEG> __device__ __forceinline__ float far_x(const SlabT& t) { return (NEG & 1) ? t.xy1.x : t.xy2.x; }
template <int NEG> __device__ __forceinline__ float near_y(const SlabT& t) { return (NEG & 2) ? t.xy2.y : t.xy1.y; }
template <int NEG> __device__ __forceinline__ float far_y(const SlabT& t) { return (NEG & 2) ? t.xy1.y : t.xy2.y; }
template <int NEG> __device__ __forceinline__ float near_z(const SlabT& t) { return (NEG & 4) ? t.z12.y : t.z12.x; }
template <int NEG> __device__ __forceinline__ float far_z(const SlabT& t) { return (NEG & 4) ? t.z12.x : t.z12.y; }
// Both parts end in ONE float comparison whose operand carries the other conditions (a lane that already failed compares against
// an infinity): the result of a comparison is a lane mask in SGPRs that a ballot can use as it is, where a boolean combined
// from several would first be turned into 0 / 1 per lane and compared again.
template <int NEG>
__device__ __forceinline__ bool slab_distance_part(const SlabT& t, bool part, float closest, float& chord) {
    const float tMin = fmaxf(fmaxf(near_x<NEG>(t), near_y<NEG>(t)), near_z<NEG>(t));
    const float tMax = fminf(fminf(far_x<NEG>(t), far_y<NEG>(t)), far_z<NEG>(t));
    const bool ok = part && (tMax >= fmaxf(tMin, 0.f));
    chord = tMax - tMin;
    return tMin < (ok ? closest : -__builtin_inff());
}
// The overlap part without evaluating it.  In real arithmetic its three comparisons are fy > nz, fz > nx, fx > ny, and each of
// those differences is at least tMax - tMin (fy >= tMax = min f, nz <= tMin = max n).  Evaluated in float as the reference
// does -- (fy - ny) + (fz - nz) > fz - ny: three subtractions and a sum of values below 4 T in magnitude -- the two sides keep
// their order whenever the real difference exceeds 10 * 2^-24 * T, T = the largest |slab distance| at the node.  Every box lies
// inside the root box (DevScene::axisCull) and subtraction and multiplication round monotonically, so T <= tRoot, the largest
// |slab distance| of the ROOT box: a per-ray constant.  A lane with tMax - tMin > 2^-19 * tRoot (three times the bound) therefore
// passes the overlap part whatever its bits; any other lane (a box the ray only grazes, a flat box: tMax == tMin) sends the wave
// to the exact evaluation.  On the benchmark view 98 % of the entered nodes take the shortcut (tools/walk_stats.py).
__device__ __forceinline__ float overlap_margin(f3 o, f3 dinv, const float4& rootA, const float4& rootB) {
    const float ax = gabs((rootA.x - o.x) * dinv.x), bx = gabs((rootB.x - o.x) * dinv.x);
    const float ay = gabs((rootA.y - o.y) * dinv.y), by = gabs((rootB.y - o.y) * dinv.y);
    const float az = gabs((rootA.z - o.z) * dinv.z), bz = gabs((rootA.w - o.z) * dinv.z);
    return fmaxf(fmaxf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.9073486328125e-6f;       // 2^-19 * tRoot
}
template <int NEG>
__device__ __forceinline__ bool slab_overlap_part(const SlabT& t, bool near) {
    const float nx = near_x<NEG>(t), ny = near_y<NEG>(t), nz = near_z<NEG>(t), fx = far_x<NEG>(t), fy = far_y<NEG>(t), fz = far_z<NEG>(t);
    const float dx = fx - nx, dy = fy - ny, dz = fz - nz;
    const bool ok = near & (dy + dz > fz - ny) & (dz + dx > fx - nz);     // plain "and": the short-circuit form compiles to nested exec-mask regions
    return dx + dy > (ok ? fy - nx : __builtin_inff());
}

template <int NEG, bool COUNT>
__device__ __forceinline__ void packet_walk_fast(const DevScene& s, int order, bool mine, const Ray& ray, const RayBoxCtx& ctx, WalkResult& r) {
    const char* __restrict__ base = reinterpret_cast<const char*>(s.nodesAll + (size_t)order * (size_t)s.bvhSize);
    const unsigned end = (unsigned)s.bvhSize;
    const vf2 oxy = { ctx.o.x, ctx.o.y }, ozz = { ctx.o.z, ctx.o.z }, dxy = { ctx.dinv.x, ctx.dinv.y }, dzz = { ctx.dinv.z, ctx.dinv.z };
    unsigned myNext = mine ? 0u : end;
    unsigned c = 0;                                           // wave-uniform
    float4 ra = *reinterpret_cast<const float4*>(base), rb = *reinterpret_cast<const float4*>(base + 16);      // uniform addresses -> scalar loads
    const float margin = overlap_margin(ctx.o, ctx.dinv, ra, rb);        // the first record is the root
    while (c != end) {
        if (COUNT) r.nodes++;                                 // (a scalar instruction per node: only for launches that split their heavy tiles)
#ifdef RS_WALK_STATS
        r.steps++; if (myNext == c) r.myVisits++;
#endif
        const int prim = __float_as_int(rb.z);
        const unsigned nxt = (unsigned)__float_as_int(rb.w), cNext = c + 1u;
        const SlabT t = slabs(oxy, ozz, dxy, dzz, ra, rb);
        float chord;
        const bool near = slab_distance_part<NEG>(t, myNext == c, r.closest, chord);
        const unsigned long long nearMask = __builtin_amdgcn_ballot_w64(near);
        unsigned target = nxt;                                // wave-uniform: the record the wave reads next
        float4 pa, pb;
        bool took = false;
        if (nearMask != 0ull) {
            // the record after this one is requested as soon as some lane may enter (it is the successor then), before the
            // overlap part and the triangle test; a node every lane rejects on the distance part does not pay for it
            pa = *reinterpret_cast<const float4*>(base + cNext * 32u); pb = *reinterpret_cast<const float4*>(base + cNext * 32u + 16u);
#ifdef RS_WALK_STATS
            r.nearSteps++;
#endif
            // (ballots of plain comparisons, combined on the scalar unit: a ballot of a combined boolean costs two vector instructions)
            bool entered = near;
            unsigned long long enteredMask = nearMask;
            if ((nearMask & __builtin_amdgcn_ballot_w64(!(chord > margin))) != 0ull) {
                entered = slab_overlap_part<NEG>(t, near);
                enteredMask = __builtin_amdgcn_ballot_w64(entered);
            }
#ifdef RS_WALK_STATS
            else r.clearSteps++;
#endif
            if (enteredMask != 0ull) {
#ifdef RS_WALK_STATS
                r.enteredSteps++; if (prim != kNullPrim) r.leafSteps++;
#endif
                if (prim != kNullPrim) {                      // uniform branch
                    const float4* tp = reinterpret_cast<const float4*>(s.tris + prim);      // uniform -> scalar
                    const float4 a = tp[0], b = tp[1], e = tp[2];
                    float bx, by, dist;
                    const bool hit = tri_hit<true>(ray.o, ray.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(e.x, e.y, e.z), bx, by, dist);
                    if (entered && hit && dist < r.closest) { r.closest = dist; r.bx = bx; r.by = by; r.prim = prim; }
                }
                myNext = entered ? cNext : max(myNext, nxt);
                target = cNext; took = true;
            }
        }
        if (!took) {
            myNext = max(myNext, nxt);
            pa = *reinterpret_cast<const float4*>(base + nxt * 32u); pb = *reinterpret_cast<const float4*>(base + nxt * 32u + 16u);     // (requesting it at the top of every step as well: frame 1.20 -> 1.27 ms)
        }
        c = target; ra = pa; rb = pb;       // (one tail for both outcomes: with `continue` in the entered branch the compiler carried an undefined record index through the other, one v_readfirstlane per node)
    }
}

template <bool GENERAL, bool COUNT>
__device__ __forceinline__ void packet_walk_order(const DevScene& s, int order, bool mine, const Ray& ray,
                                                  const RayBoxCtx& ctx, WalkResult& r) {
    const BvhNode* __restrict__ nodes = s.nodesAll + (size_t)order * (size_t)s.bvhSize;
    const unsigned end = (unsigned)s.bvhSize;
    unsigned myNext = mine ? 0u : end;
    unsigned c = 0;                                           // wave-uniform
    // uniform addresses -> scalar loads.  The record after the current one is requested before the current
    // one is tested: c+1 is the successor whenever any lane enters the node (about half of the steps), and
    // then the scalar-load latency is off the wave's critical path.  nodes[end] is readable (next order / padding).
    const float4* np0 = reinterpret_cast<const float4*>(nodes);
    float4 lo, hi;
    node_unpack(np0[0], np0[1], lo, hi);
    while (c != end) {
        if (COUNT) r.nodes++;
#ifdef RS_WALK_STATS
        r.steps++;
#endif
        const float4* nq = reinterpret_cast<const float4*>(nodes + c + 1);
        float4 plo, phi;
        node_unpack(nq[0], nq[1], plo, phi);
        const int prim = __float_as_int(lo.w);
        const unsigned nxt = (unsigned)__float_as_int(hi.w);
        const bool part = myNext == c;
        float tb;
        bool bh;
        if (GENERAL) bh = box_hit_general(ctx.o, ctx.dinv, lo, hi, tb);
        else bh = box_hit(ctx, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), tb);
        const bool entered = part & bh & (tb < r.closest);
        if (prim != kNullPrim) {                              // uniform branch
            if (__any(entered)) {
                const float4* tp = reinterpret_cast<const float4*>(s.tris + prim);      // uniform -> scalar
                const float4 a = tp[0], b = tp[1], e = tp[2];
                float bx, by, dist;
                const bool hit = tri_hit(ray.o, ray.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(e.x, e.y, e.z), bx, by, dist);
                if (entered && hit && dist < r.closest) { r.closest = dist; r.bx = bx; r.by = by; r.prim = prim; }
            }
        }
        myNext = part ? (entered ? c + 1u : nxt) : myNext;
        // every pending target is > c; if some lane wants c+1 that is the minimum
        if (__any(myNext == c + 1u)) { c = c + 1u; lo = plo; hi = phi; }
        else {
            c = wave_min_u32(myNext);
            const float4* np = reinterpret_cast<const float4*>(nodes + c);
            node_unpack(np[0], np[1], lo, hi);
        }
    }
}

// the bits of `neg`: a direction component is negative in the lanes that take part (the same in all of them)
template <bool COUNT>
__device__ __forceinline__ void packet_walk_fast_dispatch(int neg, const DevScene& s, int order, bool mine, const Ray& ray, const RayBoxCtx& ctx, WalkResult& r) {
    switch (neg) {
        case 0: packet_walk_fast<0, COUNT>(s, order, mine, ray, ctx, r); break;
        case 1: packet_walk_fast<1, COUNT>(s, order, mine, ray, ctx, r); break;
        case 2: packet_walk_fast<2, COUNT>(s, order, mine, ray, ctx, r); break;
        case 3: packet_walk_fast<3, COUNT>(s, order, mine, ray, ctx, r); break;
        case 4: packet_walk_fast<4, COUNT>(s, order, mine, ray, ctx, r); break;
        case 5: packet_walk_fast<5, COUNT>(s, order, mine, ray, ctx, r); break;
        case 6: packet_walk_fast<6, COUNT>(s, order, mine, ray, ctx, r); break;
        default: packet_walk_fast<7, COUNT>(s, order, mine, ray, ctx, r); break;
    }
}

// closest hit for a wave of coherent rays; lanes with active == false carry no ray
// unionNodes (may be null): the number of nodes the wave visited, the length of its chain of dependent fetches (tile splitting, rs_tilesplit.h)
template <bool COUNT = false>
__device__ inline Hit trace_closest_packet(const DevScene& s, const Ray& ray, bool active, unsigned* unionNodes = nullptr) {
    WalkResult w;
    w.closest = 3.402823466e+38f; w.prim = kNullPrim; w.bx = 0.f; w.by = 0.f; w.any = false; w.nodes = 0;
    RayBoxCtx ctx = make_box_ctx(ray);
    ctx.cull = s.axisCull;
    const bool special = active && (ctx.mode != 0 || ctx.zx || ctx.zy || ctx.zz || !(ray.d.x == ray.d.x));
    const bool anySpecial = __any(special);
    const int order = mtbvh_order(-ray.d);
    unsigned long long todo = __ballot(active);
#ifdef RS_WALK_STATS
    w.steps = w.nearSteps = w.enteredSteps = w.leafSteps = w.clearSteps = 0; w.myVisits = 0;
    unsigned norders = 0;
#endif
    while (todo) {                                            // one pass per threaded order present in the wave
#ifdef RS_WALK_STATS
        norders++;
#endif
        const int lead = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(order, lead);
        const bool mine = active && order == k;
        const unsigned long long mm = __ballot(mine);
        todo &= ~mm;
        const unsigned long long sx = __ballot(mine && ray.d.x < 0.f), sy = __ballot(mine && ray.d.y < 0.f), sz = __ballot(mine && ray.d.z < 0.f);
        const bool uniformSigns = (sx == 0 || sx == mm) && (sy == 0 || sy == mm) && (sz == 0 || sz == mm);
        if (anySpecial) packet_walk_order<false, COUNT>(s, k, mine, ray, ctx, w);
        else if (uniformSigns && s.axisCull && s.linksNested) packet_walk_fast_dispatch<COUNT>((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0), s, k, mine, ray, ctx, w);
        else packet_walk_order<true, COUNT>(s, k, mine, ray, ctx, w);
    }
#ifdef RS_WALK_STATS
    if (s.walkStats && __lane_id() == 0) {
        atomicAdd(&s.walkStats[16], 1ull); atomicAdd(&s.walkStats[17], (unsigned long long)w.steps);
        atomicMax(&s.walkStats[18], (unsigned long long)w.steps); atomicAdd(&s.walkStats[19], (unsigned long long)norders);
        atomicAdd(&s.walkStats[20], anySpecial ? 1ull : 0ull);
        atomicAdd(&s.walkStats[21], (unsigned long long)w.nearSteps); atomicAdd(&s.walkStats[22], (unsigned long long)w.enteredSteps);
        atomicAdd(&s.walkStats[23], (unsigned long long)w.leafSteps); atomicAdd(&s.walkStats[15], (unsigned long long)w.clearSteps);
        atomicAdd(&s.walkStats[24 + (w.steps ? 31 - __clz((int)w.steps) : 0)], 1ull);
    }
    {   // is a heavy tile heavy because its rays diverge (large union) or because single rays visit that many nodes?
        unsigned mv = w.myVisits;
        for (int off = 32; off > 0; off >>= 1) mv = max(mv, (unsigned)__shfl_xor((int)mv, off));
        if (s.walkStats && __lane_id() == 0) {
            atomicAdd(&s.walkStats[87], (unsigned long long)mv);
            if (w.steps >= 1024u) { atomicAdd(&s.walkStats[84], 1ull); atomicAdd(&s.walkStats[85], (unsigned long long)w.steps); atomicAdd(&s.walkStats[86], (unsigned long long)mv); }
        }
    }
#endif
    if (unionNodes) *unionNodes = w.nodes;
    Hit h;
    h.primId = w.prim;
    h.matId = 0;
    h.pos = splat(0.f);
    h.norm = splat(0.f);
    h.bx = w.bx; h.by = w.by;
    if (w.prim != kNullPrim) {             // getIntersecGeomInfo (scene.h:135-151)
        const float* v = s.vertices + (size_t)w.prim * 9;
        const float* n = s.normals + (size_t)w.prim * 9;
        float wgt = 1.f - w.bx - w.by;
        h.pos = ld3(v + 3) * w.bx + ld3(v + 6) * w.by + ld3(v) * wgt;
        h.norm = normalize(ld3(n + 3) * w.bx + ld3(n + 6) * w.by + ld3(n) * wgt);
        h.matId = s.materialIds[w.prim];
    }
    return h;
}

// closest hit for a whole wave of INCOHERENT rays (bounce rays): per-lane walks of the reference's tree with the
// pair-cooperative node fetch; every lane of the wave must call it, `active` false where there is no ray
__device__ inline Hit trace_closest_wave(const DevScene& s, const Ray& ray, bool active) {
    const WalkResult w = walk_dispatch_paired<false>(s, ray, 3.402823466e+38f, active);
    Hit h;
    h.primId = active ? w.prim : kNullPrim;
    h.matId = 0;
    h.pos = splat(0.f);
    h.norm = splat(0.f);
    h.bx = w.bx; h.by = w.by;
    if (h.primId != kNullPrim) {           // getIntersecGeomInfo (scene.h:135-151)
        const float* v = s.vertices + (size_t)w.prim * 9;
        const float* n = s.normals + (size_t)w.prim * 9;
        float wgt = 1.f - w.bx - w.by;
        h.pos = ld3(v + 3) * w.bx + ld3(v + 6) * w.by + ld3(v) * wgt;
        h.norm = normalize(ld3(n + 3) * w.bx + ld3(n + 6) * w.by + ld3(n) * wgt);
        h.matId = s.materialIds[w.prim];
    }
    return h;
}

// testOcclusion for a whole wave of (incoherent) segments with the pair-cooperative fetch; every lane of
// the wave must call it, `active` false where there is no segment
__device__ inline bool trace_occluded_wave(const DevScene& s, f3 x, f3 y, bool active) {
    f3 dir = y - x;
    float dist = length(dir);
    dir = div3_exact_signed(dir, dist);
    Ray ray; ray.o = x + dir * 1e-5f; ray.d = dir;       // makeOffsetedRay (intersections.h:13-15)
    dist -= 1e-4f * 2.f;
    return walk_dispatch_paired<true>(s, ray, dist, active).any;
}

// DevScene::testOcclusion (src/scene.h:286-316): any hit between x and y
__device__ inline bool trace_occluded(const DevScene& s, f3 x, f3 y) {
    f3 dir = y - x;
    float dist = length(dir);
    dir = div3_exact_signed(dir, dist);
    Ray ray; ray.o = x + dir * 1e-5f; ray.d = dir;       // makeOffsetedRay (intersections.h:13-15)
    dist -= 1e-4f * 2.f;
    return walk_dispatch<true>(s, ray, dist).any;
}
#endif  // __HIPCC__

// ---- materials (src/material.h:34-124,171-186,218-228) -----------------------------------------
// x / d, 1 / sqrt(x): on the device the short forms of rs_exact.h (the same IEEE results, fewer instructions for operands in [2^-60, 2^60))
RS_HD float bsdf_div(float x, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return div_exact(x, d);
#else
    return x / d;
#endif
}
RS_HD f3 bsdf_normalize(f3 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v * rcp_exact(sqrt_exact(dot(v, v)));
#else
    return normalize(v);
#endif
}
RS_HD float schlick_g(float c, float alpha) { float a = alpha * .5f; return bsdf_div(c, c * (1.f - a) + a); }
RS_HD float gtr2(float c, float alpha) {
    if (c < 1e-6f) return 0.f;
    float aa = alpha * alpha;
    float den = c * c * (aa - 1.f) + 1.f;
    den = den * den * kPi;
    return bsdf_div(aa, den);
}

RS_HD f3 eval_bsdf(int type, f3 baseColor, float metallic, float roughness, f3 n, f3 wo, f3 wi) {
    if (type == 0) {                                   // lambertianBSDF: baseColor * 1.f / Pi
        return (baseColor * 1.f) / kPi;
    }
    if (type == 1) {                                   // metallicWorkflowBSDF
        float alpha = roughness * roughness;
        f3 h = bsdf_normalize(wo + wi);
        float cosO = dot(n, wo);
        float cosI = dot(n, wi);
        if (cosI * cosO < 1e-7f) return splat(0.f);
        f3 f0 = mix(splat(.08f), baseColor, metallic);
        f3 f = mix(f0, splat(1.f), pow5(1.f - dot(h, wo)));
        float g = schlick_g(gabs(cosO), alpha) * schlick_g(gabs(cosI), alpha);
        float d = gtr2(dot(n, h), alpha);
        f3 diffuse = ((baseColor * 1.f) / kPi) * (1.f - metallic);
        return mix(diffuse, splat(bsdf_div(g * d, 4.f * cosI * cosO)), f);
    }
    return splat(0.f);                                 // Dielectric, Disney, Light
}

// ---- light sampling (src/scene.h:394-459, src/sampler.h:203-207, src/mathUtil.h:94-100,182-185) --
struct LightSample { float pdf; f3 Li, wi; float dist; f3 point; int id; float bu, bv; };     // bu, bv: the barycentric pair of sampleTriangleUniform

#if defined(__HIPCC__)
// sampleDirectLightNoVisibility; `lights`/`alias` may point to global memory or to an LDS copy.
// Bit-exact shortcuts: dot(x-y, x-y) == dot(y-x, y-x) and normalize(x-y) == -normalize(y-x), so
// the pdf conversion (mathUtil.h:182-185) reuses wi and dist instead of re-deriving them.
template <bool ENV, typename AliasPtr, typename LightPtr>
__device__ __forceinline__ LightSample sample_light_nv(const DevScene& s, AliasPtr alias, LightPtr lights, int numLights, f3 pos, f4 r);
#endif

}  // namespace rs

#include "rs_surface.h"

