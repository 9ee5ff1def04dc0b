// rs_internal.h -- host-side objects behind the opaque handles of include/restir_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#ifndef RS_AUX_STREAMS
#define RS_AUX_STREAMS 3
#endif
#ifndef RS_GBUF_SETS
#define RS_GBUF_SETS (RS_AUX_STREAMS + 2)
#endif
#ifndef RS_SURF_SETS
#define RS_SURF_SETS (RS_AUX_STREAMS + 1)
#endif
#include <vector>

#include "rs_scene.h"
#include "rs_tilesplit.h"

// ---- contexts ----------------------------------------------------------------------------------
// What used to be process-wide settings: the device, the stream the library enqueues on, synchronous / asynchronous
// launches, the auxiliary streams of the asynchronous mode and how GBuffer::render is launched.  Every object
// (scene, G-buffer, reservoirs, filters, strip driver) belongs to the context that was current in its thread when it
// was created, and every entry point that takes an object runs under that object's context (rs_ctx_scope), so two
// contexts -- two host threads, two devices, two streams, a synchronous and an asynchronous caller -- do not see each
// other's settings.  A thread that never asks for one uses the default context (what rs_init / rs_set_* configure).
struct rs_context {
    static constexpr int kAux = RS_AUX_STREAMS;   // 0: GBuffer::render, 1 + k: the primary -> RIS -> shadow chain of every second frame (rs_restir::kChains); small launches: all of them chains in turn
    int device = 0;
    hipStream_t stream = nullptr;
    bool sync = true;
    hipStream_t aux[kAux] = {};
    int auxPriority = 0;                  // the level the auxiliary streams were created at
    int auxWant = -99;                    // cached rs_internal_stream_priority(); -99: recompute
    bool auxLevelSet = false; int auxLevel = 0;   // rs_set_internal_stream_priority
    bool auxStale = false;                // the auxiliary streams were chosen for another caller stream / preference: choose again on next use
    bool auxPlain = false;                // the choice by measurement failed once: plain streams from then on
    bool auxForget = false;               // rs_choose_internal_streams_again: the kept choices are dropped as well
    hipStream_t auxForStream = nullptr; int auxForLevel = 2;     // the caller stream and the preference the current three were chosen for
    struct AuxChoice { hipStream_t caller; int level; hipStream_t aux[RS_AUX_STREAMS]; int priority; double chosenUs, fastestUs; };
    std::vector<AuxChoice> auxKept;       // choices made for other caller streams / preferences (rs_set_stream back to one of them takes its three again)
    double auxCalibratedUs = 0, auxFastestUs = 0;   // the chosen triple's calibration time and the fastest of all candidates (rs_internal_streams_info)
    int auxMode = -1;                     // -1: not decided yet (RS_SIDE_STREAM); 0 off; 1 on
    int risGlobalBelow = -1;              // launches of fewer pixels read the RIS light table from global memory (rs_set_ris_table_pixels): -1 = the default, 64 Ki
    int fuseMode = -1;                    // deferred G-buffer render walked with the primary rays: -1 = not resolved yet (rs_set_side_stream; default: measured per rs_restir)
    int chainStreams = -1, smallChains = -1, shadowOnMain = -1;   // rs_set_stream_plan; -1: not resolved yet (environment or default)
    int ownCommStreams = 0;               // strip drivers of this context that keep a transfer stream of their own busy (rs_strips_set_comm_stream(s, 1))
    // The denoise stream (rs_set_denoise_stream): LeveledEAWFilter of frame f -- and the tone map that reads its result -- on auxiliary
    // stream 0, ordered after phase B of f by an event, next to the temporal / spatial passes of frame f + 1 on the library stream.  Four
    // streams that hand events to each other is what the device runs side by side, so the chains then take turns on
    // TWO streams (rs_chains_in_flight).  What that stream writes is ordered for the library stream by events, not by stream order:
    // denoiseBufs lists those buffers with the event that covers their last use there, and every entry point that is handed a raw image
    // pointer asks rs_denoise_order() first.
    int denoiseMode = 0;                  // 0: filters run on the library stream; 1: on auxiliary stream 0, the chains on two streams (asynchronous mode only)
    hipStream_t streamOverride = nullptr; // inside an rs_denoise_scope: what rs_stream() returns
    struct DenoiseBuf { const char* base; size_t bytes; hipEvent_t ev; bool readOnly, pending; };
    std::vector<DenoiseBuf> denoiseBufs;  // (an entry keeps its event for the buffer's next use)
    bool denoiseUsed = false;             // the denoise stream has carried work since the last rs_synchronize
    hipEvent_t denoiseFork = nullptr;     // library stream -> denoise stream
    unsigned long long* ptRayCount = nullptr;   // pathTraceDirect's walk counter (pathtrace.hip)
    int tileSplit = 0; bool tileSplitSet = false;   // union nodes from which a tile of a closest-hit kernel is traced by four waves (rs_tilesplit.h): rs_set_tile_split, default 768; 0 off; negative: |value|, also for launches that overlap others
};
rs_context* rs_ctx();                                   // the context this thread's library code runs under right now
struct rs_ctx_scope {                                   // entry points: run under the context of the object they were handed
    rs_context* prev;
    int prevDevice = -1;                                // the caller's device, restored on exit when the scope had to change it
    explicit rs_ctx_scope(rs_context* c);               // null: keep the thread's current context
    ~rs_ctx_scope();
};
#define RS_SCOPE(obj) rs_ctx_scope rs_scope_guard_((obj) ? (obj)->ctx : nullptr)

// ---- error plumbing --------------------------------------------------------------------------
int rs_fail(int code, const char* msg);                 // records msg, returns code
int rs_check_hip(hipError_t e, const char* what);       // 0 on success
hipStream_t rs_stream();
bool rs_sync_enabled();
// after a launch: hipGetLastError (+ stream sync when sync mode is on), like checkCUDAError
int rs_after_launch(const char* what);
// auxiliary streams of the asynchronous mode (api_common.hip): 0 = GBuffer::render, 1 + k = primary rays + RIS + shadow rays of every
// kChains-th frame; nullptr = not in use
hipStream_t rs_aux_stream(int i);
hipStream_t rs_aux_stream_any(int i);                  // the same stream whatever the launch mode (null only when the side streams are switched off)
int rs_aux_synchronize();
int rs_internal_stream_priority();                     // the priority level the library's own streams are created at: not the caller's stream's
int rs_chains_in_flight();                             // how many auxiliary streams the frames' chains take in turn: 3, less one per other stream of the context with work in flight (a strip driver's transfer stream, the denoise stream)
// ---- the denoise stream (api_common.hip) ----
hipStream_t rs_denoise_stream();                       // auxiliary stream 0 when rs_set_denoise_stream(1) is in force and launches are asynchronous, else null
// Work for the denoise stream: the constructor orders that stream after everything enqueued on the library stream so far (`fork`) and makes
// rs_stream() return it; the destructor restores the library stream.  Inactive (everything stays on the library stream) when there is no
// denoise stream.
struct rs_denoise_scope {
    rs_context* c = nullptr;
    hipStream_t prev = nullptr;
    bool active = false;
    int err = 0;
    explicit rs_denoise_scope(bool fork, bool enable = true);
    ~rs_denoise_scope();
};
int rs_denoise_mark(const void* base, size_t bytes, bool readOnly);   // inside a scope: `base` is in use on the denoise stream up to here (readOnly: only read there)
int rs_denoise_order(const void* p, bool write = true);               // the library stream waits for the denoise stream's last use of the buffer p points into (no-op for buffers it never touched; reads do not wait for reads)
bool rs_denoise_owns(const void* p);                                  // p points into a buffer the denoise stream has written and the library stream has not been ordered after
int rs_denoise_join();                                                // the library stream waits for everything enqueued on the denoise stream

#define RS_TRY(expr)                                       \
    do {                                                   \
        int _rs_e = (expr);                                \
        if (_rs_e != 0) return _rs_e;                      \
    } while (0)
#define RS_HIP(expr) RS_TRY(rs_check_hip((expr), #expr))

template <typename T>
static inline int rs_dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    return rs_check_hip(hipMalloc((void**)p, sizeof(T) * count), "hipMalloc");
}
template <typename T>
static inline void rs_dev_free(T*& p) {
    if (p) { (void)hipFree((void*)p); p = nullptr; }
}

// kernel<A, B> for two run-time booleans (TEX / ENV x SOBOL variants)
#define RS_LAUNCH2(kernel, a, b, grid, block, stream, ...)                                                          \
    do {                                                                                                            \
        if (a) { if (b) hipLaunchKernelGGL((kernel<true, true>), grid, block, 0, stream, __VA_ARGS__);              \
                 else   hipLaunchKernelGGL((kernel<true, false>), grid, block, 0, stream, __VA_ARGS__); }           \
        else   { if (b) hipLaunchKernelGGL((kernel<false, true>), grid, block, 0, stream, __VA_ARGS__);             \
                 else   hipLaunchKernelGGL((kernel<false, false>), grid, block, 0, stream, __VA_ARGS__); }          \
    } while (0)
#define RS_LAUNCH1(kernel, a, grid, block, stream, ...)                                                             \
    do {                                                                                                            \
        if (a) hipLaunchKernelGGL((kernel<true>), grid, block, 0, stream, __VA_ARGS__);                             \
        else   hipLaunchKernelGGL((kernel<false>), grid, block, 0, stream, __VA_ARGS__);                            \
    } while (0)
// The Sobol branch indexes its table by the caller's looper (State::looper, kept below SobolSampleNum by the reference's
// `(looper + 1) % SobolSampleNum`, restir.cu:441-445): a looper outside the table is refused instead of read.
int rs_check_looper(const struct rs_scene* scene, int looper, const char* what);

// ---- tile-split hints (rs_tilesplit.h) -------------------------------------------------------------
// One per launch site and stream: the lists / flags the last launch of that geometry on that stream left for the next one.
struct rs_tile_split {
    static constexpr int kCapacity = 4096;
    int* base = nullptr;                             // header + three hints + three flag arrays (layout: rs_tilesplit.h)
    size_t bytes = 0;
    int rot = 0;                                     // the hint the next launch reads
    long long key = -1;                              // the launch geometry (and threshold) the hints belong to
    int numTiles = 0, capacity = 0;                  // ... and what the device header says (a changed value re-initialises the hints whatever the key)
    // small overlapped launches (mode 2): the device's report word and the host's view of it
    unsigned long long* report = nullptr;            // pinned host memory (rs_tilesplit.h)
    unsigned issued = 0, wake = 0;                   // split launches issued so far; the count when the site last (re)started splitting
    int sleep = 0;                                   // plain launches left before the site looks again
    bool lastNone = false;                           // the last fresh report named no heavy tile
};
// fills *ts for a launch of `regularBlocks` blocks of four tiles each on stream st (all zero when the feature is off) and
// returns through *helperBlocks how many blocks the grid gets in front of them.  mode:
//   1  the kernel runs with nothing next to it (synchronous mode, per-pass timing): the launch lasts as long as its longest chain;
//   0  a launch of several rounds of wave slots next to other frames' kernels: they fill in around a long tile, and splitting
//      measured 2.5 % SLOWER (extra waves, extra blocks) -- no splitting;
//   2  a launch of less than three rounds of wave slots next to other frames' kernels (a strip): a single round of waves ends with
//      its slowest tile whatever runs next to it -- the 72- and 48-row strips of a cost-balanced 8-way split of the Bistro-class
//      frame 0.42 / 0.46 -> 0.29 / 0.26 ms per frame (profiles/r05_ab_tile_split_on_strips.log).  Split at 4/3 of the threshold (768
//      -> 1024: measured better than 768 on strips), and only while it finds heavy tiles: the kernels that count cost a scene
//      without any (the Sponza-class strips) 1.3 %, so a site whose last fresh report says "none" runs the plain kernels for
//      kTileSplitSleep launches before it looks again.
int rs_tile_split_prepare(rs_tile_split* t, long long key, int numTiles, int regularBlocks, int mode, hipStream_t st, rs::TileSplit* ts, int* helperBlocks);
constexpr long long kSmallLaunchWaves = 3 * 8192;   // three rounds of the chip's 8 192 wave slots (256 CUs x 4 SIMDs x 8 waves)
#ifndef RS_SPLIT_SMALL_ROUNDS
#define RS_SPLIT_SMALL_ROUNDS 3
#endif
constexpr long long kSplitSmallWaves = (long long)RS_SPLIT_SMALL_ROUNDS * 8192;   // launches below this many waves split their heavy tiles also when other kernels run next to them
void rs_tile_split_free(rs_tile_split* t);
int rs_tile_split_threshold();                      // of the current context

// ---- scene -------------------------------------------------------------------------------------
struct rs_scene {
    rs_context* ctx = nullptr;
    rs::DevScene dev{};              // passed to kernels by value (pointers into the arrays below)
    // owned device arrays
    rs::BvhNode* dNodesAll = nullptr;
    rs::TriRec* dTris = nullptr;
    float* dVertices = nullptr;
    float* dNormals = nullptr;
    int* dMaterialIds = nullptr;
    rs_material* dMaterials = nullptr;
    rs::LightRec* dLights = nullptr;
    rs::AliasRec* dAlias = nullptr;
    rs::TexRec* dTextures = nullptr;
    float* dTexcoords = nullptr;
    rs::AliasRec* dEnvAlias = nullptr;
    std::vector<float*> dTexData;                 // one device array per texture
    std::vector<std::vector<float>> hTexData;     // host copies (rs_scene_host_desc)
    std::vector<rs_texture> hTextures;
    std::vector<float> hEnvProb;
    std::vector<int> hEnvFail;
    int envMapTexId = -1;
    bool textured = false;                        // any material map or an environment map: kernels take the textured variant
    uint32_t* dSampleSeq = nullptr;  // the Sobol table (rs_scene_set_sample_sequence); null: the default thrust engine
    uint4* dOccNodes = nullptr;      // shadow-ray tree (occlusion_bvh.cpp)
    rs::BvhNode* dOccChain = nullptr;   // reference boxes + parent links by original node id
    rs::TriRec* dOccTris = nullptr;
    uint4* dEmiNodes = nullptr;      // tree of the emissive triangles alone (scene.hip build_emissive_side)
    rs::TriRec* dEmiTris = nullptr;
    uint4* dOrdNodes = nullptr;      // closest-hit trees in the reference's visiting orders (occlusion_bvh.cpp rs_build_ordered_bvh)
    rs::TriRec* dOrdTris = nullptr;
    unsigned long long* dWalkStats = nullptr;   // -DRS_WALK_STATS builds only
    // host copies of the source arrays (rs_scene_host_desc)
    std::vector<float> hVertices, hNormals, hTexcoords, hBoxes, hLightRadiance, hLightProb;
    std::vector<int> hMaterialIds, hNodes[6], hLightPrimIds, hLightFailId;
    std::vector<int> hParent, hLeafOf;   // reference tree: parent by original node id, leaf node of each primitive
    std::vector<rs_material> hMaterials;
    float sumLightPower = 0.f;
    int numPrims = 0, bvhSize = 0, numLights = 0;
    unsigned long long id = 0;       // unique per rs_scene_create (a freed scene's address can come back; its id cannot)
};

// ---- G-buffer (src/gbuffer.h:41-58) ------------------------------------------------------------
// The reference keeps two sets of the id / normal / depth planes and toggles frameIdx.  Here all planes live in a ring of
// five sets: the set of the previous frame ("last" planes), of the frame whose passes the library stream is at, and of up to
// three frames whose renders run ahead on the auxiliary streams (a strip has three chains in flight, each with the frame's render
// in its first launch), so that no render writes what an earlier frame's temporal pass still reads.  frameIdx is still toggled
// and reported by rs_gbuffer_get_view, whose devNormal[frameIdx] / [frameIdx ^ 1] are the current / last sets.
struct rs_gbuffer {
    rs_context* ctx = nullptr;
    static constexpr int kSets = RS_GBUF_SETS;
    float* albedo[kSets] = {};
    int* motion[kSets] = {};
    float* normal[kSets] = {};
    int* primId[kSets] = {};
    float* depth[kSets] = {};
    int ring = 0;                // set of the current frame; the previous frame's is (ring + kSets - 1) % kSets
    int frameIdx = 0;
    rs_camera lastCamera{};      // uninitialised in the reference until the first update (Q14); zero here
    int width = 0, height = 0;
    // ordering of a render on the auxiliary stream (asynchronous mode)
    hipEvent_t forkEv = nullptr;             // "everything enqueued on the library stream so far"
    hipEvent_t doneEv = nullptr;             // the render
    hipEvent_t useEv[kSets] = {};   // recorded by update(): the frame that ended there has been enqueued
    int useOf[kSets];       // per set: which useEv covers its last readers (-1: none outstanding)
    rs_gbuffer() { for (int i = 0; i < kSets; i++) useOf[i] = -1; }
    // a filter on the denoise stream reads (a strip's: also writes the rows just outside the strip of) the current set after the library
    // stream has moved on: the next render into that set waits for this event as well (rs_gbuffer_order_before_render)
    mutable hipEvent_t denoiseEv[kSets] = {};
    mutable bool denoiseValid[kSets] = {};
    int updates = 0;
    bool renderedSinceUpdate = false;
    mutable bool pending = false;            // a render on the auxiliary stream has not been joined yet
    // asynchronous mode: a render that has been requested but not launched yet.  ReSTIRDirect launches it together with its
    // primary rays (one packet walk for the two rays of a pixel, restir.hip k_gbuffer_primary); any other reader of the planes
    // launches it by itself first (rs_gbuffer_join).
    struct Deferred {
        bool valid = false, rerender = false;
        const rs_scene* scene = nullptr;
        rs_camera cam{}, lastCam{};
        int y0 = 0, y1 = 0;
    };
    mutable Deferred deferred;
    mutable rs_tile_split split[2];          // k_render_gbuffer on the library stream / on the auxiliary stream
    int cur() const { return ring; }
    int prev() const { return (ring + kSets - 1) % kSets; }
    // albedo / motion are single planes in the reference: they show the most recent render, also after update()
    int latest() const { return (renderedSinceUpdate || updates == 0) ? cur() : prev(); }
};
// the library stream waits for a render that is still on the auxiliary stream (and a deferred one is launched first); every
// reader of the planes calls it first
int rs_gbuffer_join(const rs_gbuffer* g);
// a deferred render was launched by someone else (ReSTIRDirect, fused with its primary rays): forget the record
void rs_gbuffer_deferred_taken(const rs_gbuffer* g);
// launches every render of `scene` that is still only recorded (rs_scene_destroy calls it before it frees the arrays)
int rs_gbuffer_release_scene(const rs_scene* scene);
// orders `stream` after the last readers of the set a (deferred) render is about to write
int rs_gbuffer_order_before_render(const rs_gbuffer* g, hipStream_t stream);
// inside an rs_denoise_scope: the current set is in use on the denoise stream up to here
int rs_gbuffer_denoise_mark(const rs_gbuffer* g);
bool rs_fuse_enabled();
int rs_ris_global_below();
const rs_context* rs_stream_plan();   // the current context with chainStreams / smallChains / shadowOnMain resolved
int rs_fuse_mode();     // 0 never, 1 always (large launches), 2 always, 3 measured per rs_restir

// device view of the planes the kernels read
struct GBufView {
    const float* albedo;
    const int* motion;
    const float* normal; const float* lastNormal;
    const int* primId;   const int* lastPrimId;
    const float* depth;  const float* lastDepth;
    int width, height;
};
static inline GBufView gbuf_view(const rs_gbuffer* g) {
    GBufView v;
    const int c = g->cur(), l = g->prev();
    v.albedo = g->albedo[g->latest()]; v.motion = g->motion[g->latest()];
    v.normal = g->normal[c]; v.lastNormal = g->normal[l];
    v.primId = g->primId[c]; v.lastPrimId = g->primId[l];
    v.depth = g->depth[c]; v.lastDepth = g->depth[l];
    v.width = g->width; v.height = g->height;
    return v;
}

#if defined(__HIPCC__)
// Wave priorities (s_setprio, 0-3: the SIMD's arbiter issues the ready wave of the highest priority first).  In the overlapped mode the
// streaming kernels of the library stream (temporal merge, border-row copies, spatial pass, tone map) share every SIMD with the walks
// and the RIS loop of other frames, and a GPU-paced kernel trace of a 1/8 strip shows them stretched seven-fold (k_temporal 13 -> 95 us:
// the library stream busy 96 % of the frame period, tools/strip_trace_c.py).  Their waves are few and short, so they go first: full
// frame 1.047 -> 1.027 ms, light strips -7 %, heavy strips -2 % (priority 1 and 3 measure the same).  The walks at ANY raised
// priority cost a third of the frame (1.05 -> 1.42 ms: they take the issue slots of the RIS loop, which is what bounds the frame), and
// the RIS loop raised gains nothing (profiles/r05_ab_wave_priorities.log).
#ifndef RS_PRIO_WALK
#define RS_PRIO_WALK 0
#endif
#ifndef RS_PRIO_STREAM
#define RS_PRIO_STREAM 1
#endif
#ifndef RS_PRIO_RIS
#define RS_PRIO_RIS 0
#endif
#ifndef RS_PRIO_EAW
#define RS_PRIO_EAW 0
#endif
#define RS_SETPRIO(p) do { if ((p) > 0) __builtin_amdgcn_s_setprio(p); } while (0)
// what renderGBuffer writes for one pixel (src/gbuffer.cu:21-72); shared by k_render_gbuffer and the kernel that walks the
// G-buffer ray together with the shading ray (restir.hip)
struct GBufWrite {
    float* albedo; int* motion; float* normal; int* primId; float* depth;
};

template <bool TEX>
__device__ __forceinline__ void gbuffer_store(const rs::DevScene& s, const rs::CamParams& cam, const rs::CamParams& lastCam, const GBufWrite& g,
                                              int idx, const rs::Ray& ray, const rs::Hit& h) {
    using namespace rs;
    if (h.primId != kNullPrim) {
        int matId = h.matId;
        f3 norm = h.norm;
        const SurfMat m = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
        if (m.type == 4) matId = kNullPrim - 1;          // lights -> -2 (gbuffer.cu:30-31; the map lookup cannot change the type)
        st3(g.albedo + (size_t)idx * 3, m.baseColor);
        st3(g.normal + (size_t)idx * 3, norm);
        g.primId[idx] = matId;
        g.depth[idx] = length(ray.o - h.pos);            // glm::distance(pos, origin) = length(origin - pos)
        int lx, ly;
        camera_raster_coord(lastCam, h.pos, lx, ly);
        g.motion[idx] = (lx >= 0 && lx < cam.width && ly >= 0 && ly < cam.height) ? ly * cam.width + lx : -1;
    }
    else {
        st3(g.albedo + (size_t)idx * 3, (TEX && s.envTex >= 0) ? env_radiance(s, ray.d) : splat(0.f));
        st3(g.normal + (size_t)idx * 3, splat(0.f));
        g.primId[idx] = kNullPrim;
        g.depth[idx] = 1.f;
        g.motion[idx] = 0;
    }
}
#endif

// ---- reservoirs ----------------------------------------------------------------------------------
// One Reservoir<DirectLiSample> (36 B AoS in the reference, src/restir.h:114-116) is stored as four
// planes so that streaming passes read 16 B / lane and the spatial pass can stage only what its
// neighbour tests need:
//   li  float4[N]  { Li.x, Li.y, Li.z, dist }
//   wi  float4[N]  { wi.x, wi.y, wi.z, 0    }
//   w   float [N]  weight
//   m   int   [N]  numSamples
struct ResvPlanes {
    float4* li = nullptr;
    float4* wi = nullptr;
    float*  w = nullptr;
    int*    m = nullptr;
};

// devDirectTemp (src/restir.cu:10), the buffer the spatial pass gathers from.  Weight and M travel
// together with the two G-buffer values every tap test needs, so that the spatial pass stages ONE
// 16-byte record (+ the 12-byte normal) per neighbour instead of four 4-byte planes:
//   tap float4[N] { weight, bits(numSamples), bits(G-buffer id), depth }
// The reservoir half of a record is only written for pixels that published this frame (stale
// otherwise, Q1); the G-buffer half is refreshed for every pixel.
struct TempPlanes {
    float4* li = nullptr;
    float4* wi = nullptr;
    float4* tap = nullptr;
};

struct rs_restir {
    rs_context* ctx = nullptr;
    // The chain primary rays -> RIS -> shadow rays of a frame depends on no other frame.  Frames put theirs on kChains auxiliary
    // streams in turn, so that consecutive frames' chains overlap: on a 1/8 strip a chain lasts 0.4 ms however few rows it has.
    // A chain writes one of kSurfSets sets of surface planes, which the frame's temporal / spatial passes read afterwards; with as
    // many sets as chains a chain would have to wait until those passes of the frame two back have finished, with one more it
    // starts as soon as its stream is free.  A third chain next to the render's stream would be a fifth stream on the runtime's
    // four hardware queues (slower).  A small launch -- a strip, whose kernels last as long as their slowest wave -- therefore
    // takes the fused launch (render + primary rays, k_gbuffer_primary), after which nothing runs on the render's stream, and
    // gives that stream to a third chain (kSmallChains; restir.hip phase_a_impl): 8 strips of 1080p 5.96x -> 6.5x.
    static constexpr int kChains = 2, kSmallChains = rs_context::kAux, kSurfSets = RS_SURF_SETS;
    int width = 0, height = 0;
    ResvPlanes cur;      // devDirectReservoir      (written this frame)
    ResvPlanes last;     // devLastDirectReservoir  (read by the temporal merge)
    TempPlanes temp;     // devDirectTemp           (published for the spatial pass)
    rs_indirect_reservoir* indResv[2] = { nullptr, nullptr };   // devIndTemporalReservoir / devIndLastTemporalReservoir (gi.hip), allocated on first use
    bool firstFrame = true;
    int looper = 0;                  // of the last phase-A call (phase B resumes the per-pixel samplers it left)
    // per-pixel state carried between the passes of one frame (implementation bytes, not in the
    // reference: its single fused kernel keeps these in registers).  kSurfSets sets, used in turn: the primary-ray,
    // RIS and shadow-ray kernels of the next frames (auxiliary streams) fill theirs while the temporal / spatial passes of frame f read this one.
    struct Surf {
        float4* posKind = nullptr;   // hit position xyz, w = bit pattern of (matId | kind<<24)
        float4* norm = nullptr;      // shading normal xyz (flipped to wo side)
        float4* wo = nullptr;        // wo xyz (read only for non-Lambertian materials)
        uint2*  rngMat = nullptr;    // { RNG state, matId | kind<<24 }
        float4* candLi = nullptr;    // RIS winner: Li xyz, w = dist
        float4* candWi = nullptr;    // RIS winner: wi xyz, w = weight (sum of candidate weights)
    } surf[kSurfSets];
    int surfSet = 0, chain = 0, smallChain = 0;      // of the frame in flight
    hipEvent_t surfFree[kSurfSets] = {};             // recorded by end_frame: the frame that used the set has been enqueued
    bool surfFreeValid[kSurfSets] = {};
    hipEvent_t auxFork = nullptr, auxDone = nullptr;
    int phaseACalls = 0;             // since the last end_frame
    bool idleFrame = false;          // this frame's first phase-A call found the previous frames finished (restir.hip phase_a_impl)
    int idleStreak = 0;
    // one traversal for the G-buffer ray and the shading ray of a pixel, or two?  Measured once per scene (rs_fuse_mode() == 3):
    // frames 6..17 with two launches against 22..33 with one (the four frames after each switch are not timed), by events on the library stream at the frame ends
    unsigned long long tuneSceneId = 0;
    int tuneFrame = 0;               // frames with a fusable launch since tuning began
    int tuneChoice = -1;             // -1 measuring, 0 separate, 1 fused
    int lastFused = -1, lastChains = 0;   // form of the last phase-A launch: render fused with the primary rays? how many chain streams in turn? (rs_restir_launch_choice)
    bool tuneCounted = false;        // this frame had a launch the choice applies to
    hipEvent_t tuneEv[4] = { nullptr, nullptr, nullptr, nullptr };
    unsigned long long* dRayCount = nullptr;   // ring of per-frame counters (1024 slots)
    rs_tile_split split[1 + rs_context::kAux][3];   // primary-ray launches: per stream (library, auxiliary 0..2) and per call within a frame (strips: interior rows, border rows)
    int raySlot = 0;
    // timing
    int timing = 0;                  // 0 off, 1 every pass on the library stream, 2 the spatial pass only, launches as in the overlapped mode
    hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    static constexpr int kSpatialRing = 256;
    hipEvent_t spatialEv[kSpatialRing][2] = {};     // timing 2: the spatial pass of the last frames between two events each (rs_restir_spatial_times)
    int spatialNext = 0;
    bool probe = false;              // the spatial pass goes out as k_spatial_shade_probe (rs_restir_set_probe)
};

static_assert(rs_context::kAux >= 1 + rs_restir::kChains, "one auxiliary stream for GBuffer::render and one per chain");

struct rs_eaw {
    rs_context* ctx = nullptr;
    int width = 0, height = 0, level = 0;
    float sigLumin = 64.f, sigNormal = .2f, sigDepth = 1.f;     // src/denoiser.cu:455
    float* devTempImg = nullptr;
    float* devPos = nullptr;        // per-pixel cam.getPosition(x,y,depth), computed once per filter call
    bool tiled = true;              // levels of step 1, 2, 4 from an LDS tile (k_wavelet_tiled); rs_eaw_set_tiled
    bool fused = true;              // taps in fused arithmetic (denoiser.hip kFusedTaps); rs_eaw_set_fused
};

// SpatioTemporalFilter (src/denoiser.h:45-70); EAWaveletFilter(width, height, 4, 128, 1) (src/denoiser.cu:488)
struct rs_svgf {
    rs_context* ctx = nullptr;
    int width = 0, height = 0, level = 0;
    float sigLumin = 4.f, sigNormal = 128.f, sigDepth = 1.f;
    float* devAccumColor[2] = { nullptr, nullptr };
    float* devAccumMoment[2] = { nullptr, nullptr };
    float* devVariance = nullptr;
    float* devTempColor = nullptr;
    float* devTempVariance = nullptr;
    float* devFilteredVariance = nullptr;
    float* devPos = nullptr;                  // per-pixel cam.getPosition(x,y,depth), once per filter call
    bool firstTime = true;
    int frameIdx = 0;
    bool tiled = true;                        // the a-trous levels from the row-phase LDS tile (k_svgf_wavelet_tiled); rs_svgf_set_tiled
    bool fused = true;                        // taps in fused arithmetic (denoiser.hip svgf_tap_weight FUSED); rs_svgf_set_fused
};

// SpatioTemporalFilter on the rows of a strip (denoiser.hip): `exchange` swaps the first / last `rows` rows of the strip's part of
// image a (ca floats per pixel) -- and of image b when it is not null -- with the strips above and below, into the rows just outside
struct rs_svgf_row_hooks {
    void* ctx;
    int (*exchange)(void* ctx, float* a, int ca, float* b, int cb, int rows);
};
int rs_svgf_filter_rows(rs_svgf* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam, int y0, int y1,
                        const rs_svgf_row_hooks* hooks);

rs::CamParams rs_make_cam_params(const rs_camera* cam);

// occlusion_bvh.cpp
int rs_build_occlusion_bvh(int numPrims, const float* primBoxes, std::vector<rs::BvhNode>& nodes, std::vector<int>& leafPrims);
int rs_quantize_occlusion_bvh(const std::vector<rs::BvhNode>& nodes, float base[3], float scale[3], std::vector<unsigned>& out);
int rs_build_ordered_bvh(int numPrims, const float* primBoxes, const int* seq, std::vector<rs::BvhNode>& forward, std::vector<rs::BvhNode>& mirrored);
int rs_reference_chain_tables(int bvhSize, const int* order0, std::vector<int>& parent, std::vector<int>& leafOfPrim, int numPrims);
