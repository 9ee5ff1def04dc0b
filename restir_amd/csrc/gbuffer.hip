// gbuffer.hip -- GBuffer::create/destroy (src/denoiser.cu:373-403), GBuffer::render / update and the
// renderGBuffer kernel (src/gbuffer.cu:3-86).
//
// Kernel: one lane per pixel, each 64-lane wave owns an 8x8 pixel tile so that the rays of a wave
// stay coherent during the closest-hit walk (the walk is the cost; the 36 B/px of plane writes are
// coalesced in 8-pixel row segments).  HBM per pixel: write albedo 12 + normal 12 + id 4 + depth 4
// + motion 4 = 36 B; BVH node/triangle reads are data dependent and mostly served by L2 / MALL.
#include <algorithm>
#include <mutex>

#include "rs_internal.h"

#ifndef RS_WALK_WAVES
#define RS_WALK_WAVES 8        // waves per SIMD the walk kernels are held to (launch bound; holding them to exactly that many was A/B'd in round 3: no gain)
#endif
using namespace rs;

// TEX: the scene has texture maps or an environment map (getTexturedMaterialAndSurface, gbuffer.cu:38,59-62)
// 8 blocks per CU: without the bound the kernel takes 100+ SGPRs and runs at 7 waves per SIMD (0.392 -> 0.370 ms at 1080p)
template <bool TEX, bool SPLIT>
__device__ __forceinline__ void render_gbuffer_body(const DevScene& s, const CamParams& cam, const CamParams& lastCam, const GBufWrite& g,
                                                    int y0, int y1, int tilesX, const TileSplit& ts) {
    // block = 4 waves, each an 8x8 tile; the block covers 32x8 pixels (a tile that was heavy last time: four waves of 4x4, rs_tilesplit.h)
    RS_SETPRIO(RS_PRIO_WALK);
    int x, py, tile;
    bool mine, helper;
    if (!tile_split_map<8, 8>(ts, tilesX, threadIdx.x & 63, x, py, mine, tile, helper)) return;
    const int y = y0 + py;
    const bool inside = mine && x < cam.width && y < y1;
    const int idx = y * cam.width + x;

    Ray ray = camera_center_ray(cam, x, y);
    unsigned unionNodes = 0;
    Hit h = trace_closest_packet<SPLIT>(s, ray, inside, &unionNodes);       // all 64 lanes take part in the wave's walk
    if (inside) gbuffer_store<TEX>(s, cam, lastCam, g, idx, ray, h);
    if (SPLIT) tile_split_report(ts.base, ts.rot, tile, helper, !helper && !mine, unionNodes);
}

template <bool TEX>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_render_gbuffer(DevScene s, CamParams cam, CamParams lastCam, GBufWrite g, int y0, int y1, int tilesX) {
    render_gbuffer_body<TEX, false>(s, cam, lastCam, g, y0, y1, tilesX, TileSplit{ nullptr, 0, 0 });
}
template <bool TEX>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_render_gbuffer_split(DevScene s, CamParams cam, CamParams lastCam, GBufWrite g, int y0, int y1, int tilesX, TileSplit ts) {
    render_gbuffer_body<TEX, true>(s, cam, lastCam, g, y0, y1, tilesX, ts);
}

namespace {

// G-buffers that hold a render which has been requested but not launched yet.  The record keeps a pointer to the scene, so
// rs_scene_destroy launches (or drops) every pending render of the scene it is about to free (rs_gbuffer_release_scene).
std::mutex g_deferredMutex;
std::vector<const rs_gbuffer*> g_deferred;
void deferred_register(const rs_gbuffer* g) {
    std::lock_guard<std::mutex> lock(g_deferredMutex);
    if (std::find(g_deferred.begin(), g_deferred.end(), g) == g_deferred.end()) g_deferred.push_back(g);
}
void deferred_unregister(const rs_gbuffer* g) {
    std::lock_guard<std::mutex> lock(g_deferredMutex);
    g_deferred.erase(std::remove(g_deferred.begin(), g_deferred.end(), g), g_deferred.end());
}

int launch_render(const rs_gbuffer* g, const rs_scene* scene, const rs_camera* cam, const rs_camera* lastCam, int y0, int y1, hipStream_t st) {
    const int c = g->cur();
    GBufWrite w{ g->albedo[c], g->motion[c], g->normal[c], g->primId[c], g->depth[c] };
    const int tilesX = (g->width + 31) / 32, tilesY = (y1 - y0 + 7) / 8;
    TileSplit ts; int helpers = 0;
    RS_TRY(rs_tile_split_prepare(&g->split[st == rs_stream() ? 0 : 1], (((long long)y0 << 20 | y1) << 12 | tilesX), tilesX * 4 * tilesY, tilesX * tilesY,
                                 (st == rs_stream() && (rs_sync_enabled() || !rs_aux_stream(0))) ? 1 : ((long long)tilesX * tilesY * 4 < kSplitSmallWaves ? 2 : 0), st, &ts, &helpers));
    const CamParams cp = rs_make_cam_params(cam), lp = rs_make_cam_params(lastCam);
    if (ts.base) {
        if (scene->textured) hipLaunchKernelGGL(k_render_gbuffer_split<true>, dim3(helpers + tilesX * tilesY), dim3(256), 0, st, scene->dev, cp, lp, w, y0, y1, tilesX, ts);
        else hipLaunchKernelGGL(k_render_gbuffer_split<false>, dim3(helpers + tilesX * tilesY), dim3(256), 0, st, scene->dev, cp, lp, w, y0, y1, tilesX, ts);
    }
    else if (scene->textured) hipLaunchKernelGGL(k_render_gbuffer<true>, dim3(tilesX * tilesY), dim3(256), 0, st, scene->dev, cp, lp, w, y0, y1, tilesX);
    else hipLaunchKernelGGL(k_render_gbuffer<false>, dim3(tilesX * tilesY), dim3(256), 0, st, scene->dev, cp, lp, w, y0, y1, tilesX);
    return 0;
}

// launches a deferred render on the auxiliary stream
int flush_deferred(const rs_gbuffer* g) {
    if (!g->deferred.valid) return 0;
    g->deferred.valid = false;
    deferred_unregister(g);
    hipStream_t aux = rs_aux_stream(0);
    if (aux) RS_TRY(rs_gbuffer_order_before_render(g, aux));
    else aux = rs_stream();                                 // (the mode was switched in between: plain launch on the library stream)
    RS_TRY(launch_render(g, g->deferred.scene, &g->deferred.cam, &g->deferred.lastCam, g->deferred.y0, g->deferred.y1, aux));
    RS_TRY(rs_check_hip(hipGetLastError(), "renderGBuffer"));
    if (aux != rs_stream()) { RS_HIP(hipEventRecord(g->doneEv, aux)); g->pending = true; }
    return 0;
}

}  // namespace

// Nothing that runs before the temporal pass of a frame reads the G-buffer, and the set written is not the one the previous
// frame's passes read: the render is ordered after the last readers of its set (or, for a second render without an update in
// between, after everything enqueued so far on the library stream).
int rs_gbuffer_order_before_render(const rs_gbuffer* g, hipStream_t stream) {
    if (g->deferred.rerender) {
        RS_HIP(hipEventRecord(g->forkEv, rs_stream()));
        RS_HIP(hipStreamWaitEvent(stream, g->forkEv, 0));
    }
    else if (g->useOf[g->cur()] >= 0) RS_HIP(hipStreamWaitEvent(stream, g->useEv[g->useOf[g->cur()]], 0));
    if (g->denoiseValid[g->cur()]) {                    // (five frames old by now: a wait that has long been satisfied)
        RS_HIP(hipStreamWaitEvent(stream, g->denoiseEv[g->cur()], 0));
        g->denoiseValid[g->cur()] = false;
    }
    return 0;
}

int rs_gbuffer_denoise_mark(const rs_gbuffer* g) {
    const int c = g->cur();
    if (!g->denoiseEv[c]) RS_HIP(hipEventCreateWithFlags(&g->denoiseEv[c], hipEventDisableTiming));
    RS_HIP(hipEventRecord(g->denoiseEv[c], rs_stream()));
    g->denoiseValid[c] = true;
    return 0;
}

// ReSTIRDirect has launched the deferred render itself (restir.hip k_gbuffer_primary)
void rs_gbuffer_deferred_taken(const rs_gbuffer* g) {
    g->deferred.valid = false;
    deferred_unregister(g);
}

// rs_scene_destroy: every render of `scene` that is still only recorded is launched now, while the scene's arrays exist (the
// caller then waits for all streams before freeing them), so that no G-buffer keeps a pointer to a dead scene.
int rs_gbuffer_release_scene(const rs_scene* scene) {
    std::vector<const rs_gbuffer*> mine;
    {
        std::lock_guard<std::mutex> lock(g_deferredMutex);
        for (const rs_gbuffer* g : g_deferred) if (g->deferred.valid && g->deferred.scene == scene) mine.push_back(g);
    }
    int e = 0;
    for (const rs_gbuffer* g : mine) { const int r = flush_deferred(g); if (r && !e) e = r; }
    return e;
}

// the library stream waits for a render still running on an auxiliary stream; a deferred render is launched first
int rs_gbuffer_join(const rs_gbuffer* g) {
    if (!g) return 0;
    RS_TRY(flush_deferred(g));
    if (!g->pending) return 0;
    g->pending = false;
    return rs_check_hip(hipStreamWaitEvent(rs_stream(), g->doneEv, 0), "G-buffer join");
}

extern "C" {

int rs_gbuffer_destroy(rs_gbuffer* g) {
    RS_SCOPE(g);
    if (!g) return 0;
    g->deferred.valid = false;                          // a render nobody asked the result of
    deferred_unregister(g);
    (void)rs_synchronize();                             // also a render still running on the auxiliary stream
    for (int i = 0; i < rs_gbuffer::kSets; i++) {
        rs_dev_free(g->albedo[i]); rs_dev_free(g->motion[i]); rs_dev_free(g->normal[i]); rs_dev_free(g->primId[i]); rs_dev_free(g->depth[i]);
        if (g->useEv[i]) (void)hipEventDestroy(g->useEv[i]);
        if (g->denoiseEv[i]) (void)hipEventDestroy(g->denoiseEv[i]);
    }
    if (g->forkEv) (void)hipEventDestroy(g->forkEv);
    if (g->doneEv) (void)hipEventDestroy(g->doneEv);
    for (auto& t : g->split) rs_tile_split_free(&t);
    delete g;
    return 0;
}

int rs_gbuffer_create(int width, int height, rs_gbuffer** out) {
    if (!out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_create: bad size");
    *out = nullptr;
    rs_gbuffer* g = new rs_gbuffer();
    g->ctx = rs_ctx();
    rs_ctx_scope scope(g->ctx);
    g->width = width; g->height = height;
    const size_t n = (size_t)width * height;
    int e = 0;
    // The reference leaves the planes uninitialised (cudaMalloc only).  They are zeroed here so that
    // first-frame reads of the "last" planes are deterministic; no reference-visible value changes.
    for (int i = 0; i < rs_gbuffer::kSets && !e; i++) {
        if (!e) e = rs_dev_alloc(&g->albedo[i], n * 3);
        if (!e) e = rs_dev_alloc(&g->motion[i], n);
        if (!e) e = rs_dev_alloc(&g->normal[i], n * 3);
        if (!e) e = rs_dev_alloc(&g->primId[i], n);
        if (!e) e = rs_dev_alloc(&g->depth[i], n);
        if (!e) e = rs_check_hip(hipMemset(g->albedo[i], 0, n * 12), "memset");
        if (!e) e = rs_check_hip(hipMemset(g->motion[i], 0, n * 4), "memset");
        if (!e) e = rs_check_hip(hipMemset(g->normal[i], 0, n * 12), "memset");
        if (!e) e = rs_check_hip(hipMemset(g->primId[i], 0, n * 4), "memset");
        if (!e) e = rs_check_hip(hipMemset(g->depth[i], 0, n * 4), "memset");
        if (!e) e = rs_check_hip(hipEventCreateWithFlags(&g->useEv[i], hipEventDisableTiming), "hipEventCreate");
    }
    if (!e) e = rs_check_hip(hipEventCreateWithFlags(&g->forkEv, hipEventDisableTiming), "hipEventCreate");
    if (!e) e = rs_check_hip(hipEventCreateWithFlags(&g->doneEv, hipEventDisableTiming), "hipEventCreate");
    // the clears above are enqueued on the default stream, which the auxiliary streams are not ordered after: finish them
    // before the first render can be launched there
    if (!e) e = rs_check_hip(hipDeviceSynchronize(), "rs_gbuffer_create");
    if (e) { rs_gbuffer_destroy(g); return e; }
    *out = g;
    return 0;
}

int rs_gbuffer_render_rows(rs_gbuffer* g, const rs_scene* scene, const rs_camera* cam, int y0, int y1) {
    RS_SCOPE(g);
    if (!g || !scene || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_render: null argument");
    if (cam->resolution[0] != g->width || cam->resolution[1] != g->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_render: camera resolution differs from the G-buffer size");
    if (y0 < 0) y0 = 0;
    if (y1 > g->height) y1 = g->height;
    if (y1 <= y0) return 0;
    // Asynchronous mode: the render is only recorded here.  ReSTIRDirect launches it together with its primary rays (the two rays
    // of a pixel in one packet walk); any other reader of the planes launches it on the auxiliary stream first, where it is
    // ordered after the last readers of the set it writes and overlaps the other frame's passes (rs_gbuffer_join).
    const hipStream_t aux = rs_aux_stream(0);
    if (aux) {
        RS_TRY(flush_deferred(g));                          // an earlier render of this frame goes first
        g->deferred.valid = true; g->deferred.rerender = g->renderedSinceUpdate;
        g->deferred.scene = scene; g->deferred.cam = *cam; g->deferred.lastCam = g->lastCamera; g->deferred.y0 = y0; g->deferred.y1 = y1;
        deferred_register(g);
        g->renderedSinceUpdate = true;
        if (!rs_fuse_enabled()) RS_TRY(flush_deferred(g));
        return 0;
    }
    RS_TRY(rs_gbuffer_join(g));                             // an earlier render of this frame may still be on the auxiliary stream
    RS_TRY(launch_render(g, scene, cam, &g->lastCamera, y0, y1, rs_stream()));
    g->renderedSinceUpdate = true;
    return rs_after_launch("renderGBuffer");
}

int rs_gbuffer_render(rs_gbuffer* g, const rs_scene* scene, const rs_camera* cam) {
    RS_SCOPE(g);
    return rs_gbuffer_render_rows(g, scene, cam, 0, g ? g->height : 0);
}

int rs_gbuffer_update(rs_gbuffer* g, const rs_camera* cam) {
    RS_SCOPE(g);
    if (!g || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_update: null argument");
    RS_TRY(rs_gbuffer_join(g));                         // a render nobody consumed (possibly still deferred) belongs to the frame that ends here
    g->lastCamera = *cam;
    // the frame that read sets cur (as current) and prev (as last) has been enqueued on the library stream up to here
    const int c = g->cur(), l = g->prev();
    if (!rs_sync_enabled()) {
        const int k = g->updates % rs_gbuffer::kSets;
        RS_HIP(hipEventRecord(g->useEv[k], rs_stream()));
        g->useOf[c] = g->useOf[l] = k;
    }
    else g->useOf[c] = g->useOf[l] = -1;                // synchronous mode: those readers have finished
    g->updates++;
    g->ring = (g->ring + 1) % rs_gbuffer::kSets;
    g->frameIdx ^= 1;
    g->renderedSinceUpdate = false;
    return 0;
}

// rows of the id / normal / depth planes, packed [id rows][normal rows][depth rows] (20 B / px);
// sel 0 = the planes of the current frameIdx, 1 = the "last" planes (what findTemporalNeighbor reads)
size_t rs_gbuffer_rows_bytes(const rs_gbuffer* g, int rows) { RS_SCOPE(g); return g ? (size_t)g->width * (size_t)(rows > 0 ? rows : 0) * 20u : 0; }

int rs_gbuffer_rows_pack(const rs_gbuffer* g, int sel, int y0, int rows, void* devBuffer) {
    RS_SCOPE(g);
    if (!g || !devBuffer || (sel != 0 && sel != 1) || y0 < 0 || rows < 0 || y0 + rows > g->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_rows_pack: bad argument");
    RS_TRY(rs_gbuffer_join(g));
    const int f = sel ? g->prev() : g->cur();
    const size_t n = (size_t)g->width * rows, off = (size_t)y0 * g->width;
    char* b = (char*)devBuffer;
    RS_HIP(hipMemcpyAsync(b, g->primId[f] + off, n * 4, hipMemcpyDeviceToDevice, rs_stream()));
    RS_HIP(hipMemcpyAsync(b + n * 4, g->normal[f] + off * 3, n * 12, hipMemcpyDeviceToDevice, rs_stream()));
    RS_HIP(hipMemcpyAsync(b + n * 16, g->depth[f] + off, n * 4, hipMemcpyDeviceToDevice, rs_stream()));
    return rs_after_launch("rs_gbuffer_rows_pack");
}

int rs_gbuffer_rows_unpack(rs_gbuffer* g, int sel, int y0, int rows, const void* devBuffer) {
    RS_SCOPE(g);
    if (!g || !devBuffer || (sel != 0 && sel != 1) || y0 < 0 || rows < 0 || y0 + rows > g->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_rows_unpack: bad argument");
    RS_TRY(rs_gbuffer_join(g));
    const int f = sel ? g->prev() : g->cur();
    const size_t n = (size_t)g->width * rows, off = (size_t)y0 * g->width;
    const char* b = (const char*)devBuffer;
    RS_HIP(hipMemcpyAsync(g->primId[f] + off, b, n * 4, hipMemcpyDeviceToDevice, rs_stream()));
    RS_HIP(hipMemcpyAsync(g->normal[f] + off * 3, b + n * 4, n * 12, hipMemcpyDeviceToDevice, rs_stream()));
    RS_HIP(hipMemcpyAsync(g->depth[f] + off, b + n * 16, n * 4, hipMemcpyDeviceToDevice, rs_stream()));
    return rs_after_launch("rs_gbuffer_rows_unpack");
}

int rs_gbuffer_get_view(const rs_gbuffer* g, rs_gbuffer_view* v) {
    RS_SCOPE(g);
    if (!g || !v) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_gbuffer_get_view: null argument");
    RS_TRY(rs_gbuffer_join(g));                         // the caller is about to use the planes on the library stream
    const int c = g->cur(), l = g->prev(), f = g->frameIdx;
    v->devAlbedo = g->albedo[g->latest()]; v->devMotion = g->motion[g->latest()];
    v->devNormal[f] = g->normal[c]; v->devNormal[f ^ 1] = g->normal[l];
    v->devPrimId[f] = g->primId[c]; v->devPrimId[f ^ 1] = g->primId[l];
    v->devDepth[f] = g->depth[c]; v->devDepth[f ^ 1] = g->depth[l];
    v->frameIdx = g->frameIdx; v->width = g->width; v->height = g->height;
    return 0;
}

}  // extern "C"
