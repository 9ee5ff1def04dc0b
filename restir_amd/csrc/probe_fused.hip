// probe_fused.hip -- MEASUREMENT ONLY (not on the product path): is it worth walking the G-buffer ray (pixel centre) and the
// shading ray (jittered) of a pixel in ONE packet walk?  Both belong to the same 8x8 tile and visit nearly the same nodes, so a
// fused walk fetches every node once and shares the loop control and the next-node reduction; each ray still makes exactly the
// visits of DevScene::intersect.  rs_debug_probe_fused times nothing itself: tools/probe_fused_walk.py launches the two
// variants under HIP events and compares their outputs.
//   mode 0: two separate packet walks per tile (what k_render_gbuffer + k_primary do today, in one kernel)
//   mode 1: the fused walk
#include "rs_internal.h"

using namespace rs;

namespace {

template <bool GENERAL>
__device__ __forceinline__ void packet_walk_order2(const DevScene& s, int order, bool mineA, bool mineB, const Ray& ra, const RayBoxCtx& ca,
                                                   const Ray& rb, const RayBoxCtx& cb, WalkResult& wa, WalkResult& wb) {
    const BvhNode* __restrict__ nodes = s.nodesAll + (size_t)order * (size_t)s.bvhSize;
    const unsigned end = (unsigned)s.bvhSize;
    unsigned nextA = mineA ? 0u : end, nextB = mineB ? 0u : end;
    unsigned c = 0;
    const float4* np0 = reinterpret_cast<const float4*>(nodes);
    float4 lo = np0[0], hi = np0[1];
    while (c != end) {
        const float4* nq = reinterpret_cast<const float4*>(nodes + c + 1);
        const float4 plo = nq[0], phi = nq[1];
        const int prim = __float_as_int(lo.w);
        const unsigned nxt = (unsigned)__float_as_int(hi.w);
        const bool partA = nextA == c, partB = nextB == c;
        float ta, tb;
        bool ha, hb;
        if (GENERAL) { ha = box_hit_general(ca.o, ca.dinv, lo, hi, ta); hb = box_hit_general(cb.o, cb.dinv, lo, hi, tb); }
        else { ha = box_hit(ca, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), ta); hb = box_hit(cb, mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), tb); }
        const bool inA = partA & ha & (ta < wa.closest), inB = partB & hb & (tb < wb.closest);
        if (prim != kNullPrim) {
            if (__any(inA | inB)) {
                const float4* tp = reinterpret_cast<const float4*>(s.tris + prim);
                const float4 a = tp[0], b = tp[1], e = tp[2];
                float bx, by, dist;
                if (inA) { if (tri_hit(ra.o, ra.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(e.x, e.y, e.z), bx, by, dist) && dist < wa.closest) { wa.closest = dist; wa.bx = bx; wa.by = by; wa.prim = prim; } }
                if (inB) { if (tri_hit(rb.o, rb.d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(e.x, e.y, e.z), bx, by, dist) && dist < wb.closest) { wb.closest = dist; wb.bx = bx; wb.by = by; wb.prim = prim; } }
            }
        }
        nextA = partA ? (inA ? c + 1u : nxt) : nextA;
        nextB = partB ? (inB ? c + 1u : nxt) : nextB;
        const unsigned want = min(nextA, nextB);
        if (__any(want == c + 1u)) { c = c + 1u; lo = plo; hi = phi; }
        else {
            c = wave_min_u32(want);
            const float4* np = reinterpret_cast<const float4*>(nodes + c);
            lo = np[0]; hi = np[1];
        }
    }
}

__device__ __forceinline__ void trace_two(const DevScene& s, const Ray& ra, const Ray& rb, bool active, WalkResult& wa, WalkResult& wb) {
    wa.closest = wb.closest = 3.402823466e+38f; wa.prim = wb.prim = kNullPrim; wa.bx = wa.by = wb.bx = wb.by = 0.f; wa.any = wb.any = false;
    RayBoxCtx ca = make_box_ctx(ra), cb = make_box_ctx(rb);
    ca.cull = cb.cull = s.axisCull;
    const bool special = active && (ca.mode != 0 || ca.zx || ca.zy || ca.zz || cb.mode != 0 || cb.zx || cb.zy || cb.zz);
    const bool anySpecial = __any(special);
    const int oa = mtbvh_order(-ra.d), ob = mtbvh_order(-rb.d);
    unsigned long long todoA = __ballot(active), todoB = todoA;
    while (todoA | todoB) {
        int k;
        if (todoA) k = __builtin_amdgcn_readlane(oa, __ffsll((long long)todoA) - 1);
        else k = __builtin_amdgcn_readlane(ob, __ffsll((long long)todoB) - 1);
        const bool mineA = active && oa == k && ((todoA >> __lane_id()) & 1ull), mineB = active && ob == k && ((todoB >> __lane_id()) & 1ull);
        todoA &= ~__ballot(mineA); todoB &= ~__ballot(mineB);
        if (anySpecial) packet_walk_order2<false>(s, k, mineA, mineB, ra, ca, rb, cb, wa, wb);
        else packet_walk_order2<true>(s, k, mineA, mineB, ra, ca, rb, cb, wa, wb);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(DevScene s, CamParams cam, int looper, int tilesX, int* primA, int* primB, float* distA, float* distB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    const int x = bx * 32 + wave * 8 + (lane & 7), y = by * 8 + (lane >> 3);
    const bool inside = x < cam.width && y < cam.height;
    const int index = y * cam.width + x;
    Rng rng = seeded_rng(looper, index, 0);
    const f4 r = rng.uniform4();
    const Ray ra = camera_center_ray(cam, x, y), rb = camera_sample(cam, x, y, r.x, r.y);
    int pa, pb; float da, db;
    if (MODE == 0) {
        const Hit a = trace_closest_packet(s, ra, inside), b = trace_closest_packet(s, rb, inside);
        pa = a.primId; pb = b.primId;
        da = a.primId != kNullPrim ? length(ra.o - a.pos) : 0.f; db = b.primId != kNullPrim ? length(rb.o - b.pos) : 0.f;
    }
    else {
        WalkResult wa, wb;
        trace_two(s, ra, rb, inside, wa, wb);
        pa = wa.prim; pb = wb.prim;
        da = wa.prim != kNullPrim ? wa.closest : 0.f; db = wb.prim != kNullPrim ? wb.closest : 0.f;
    }
    if (inside) { primA[index] = pa; primB[index] = pb; if (MODE == 1) { distA[index] = da; distB[index] = db; } else { distA[index] = da; distB[index] = db; } }
}

}  // namespace

extern "C" int rs_debug_probe_fused(const rs_scene* scene, const rs_camera* cam, int mode, int looper, int* primA, int* primB, float* distA, float* distB) {
    if (!scene || !cam || !primA || !primB || !distA || !distB) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_probe_fused: null argument");
    const int tilesX = (cam->resolution[0] + 31) / 32, tilesY = (cam->resolution[1] + 7) / 8;
    const CamParams cp = rs_make_cam_params(cam);
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(tilesX * tilesY), dim3(256), 0, rs_stream(), scene->dev, cp, looper, tilesX, primA, primB, distA, distB);
    else hipLaunchKernelGGL(k_probe<1>, dim3(tilesX * tilesY), dim3(256), 0, rs_stream(), scene->dev, cp, looper, tilesX, primA, primB, distA, distB);
    return rs_after_launch("probe");
}
