// probe_fused.hip -- MEASUREMENT ONLY: is it worth walking the G-buffer ray (pixel centre) and the
// shading ray (jittered) of a pixel in ONE packet walk?  Both belong to the same 8x8 tile and visit nearly the same nodes, so a
// fused walk fetches every node once and shares the loop control and the next-node reduction; each ray still makes exactly the
// visits of DevScene::intersect.  rs_debug_probe_fused times nothing itself: tools/probe_fused_walk.py launches the two
// variants under HIP events and compares their outputs.
//   mode 0: two separate packet walks per tile (what k_render_gbuffer + k_primary do today, in one kernel)
//   mode 1: the fused walk
#include "rs_internal.h"

using namespace rs;

namespace {

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
        walk_two_packet(s, ra, rb, inside, inside, wa, wb);
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
