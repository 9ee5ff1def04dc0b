// api_common.hip -- library state (device, stream, sync mode, last error) of librestir_hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "rs_internal.h"

namespace {
std::mutex g_errMutex;
std::string g_lastError;
rs_context g_default;                                   // what rs_init / rs_set_* configure for threads that never create a context
thread_local rs_context* t_current = nullptr;           // rs_context_set_current
thread_local rs_context* t_scoped = nullptr;            // the object's context while an entry point runs
}  // namespace

rs_context* rs_ctx() { return t_scoped ? t_scoped : (t_current ? t_current : &g_default); }

// An entry point runs with the context's device current and leaves the caller's device as it found it: the host (torch, another
// library, a second context of this one on another GPU) may call hipSetDevice between two library calls, so the device is asked
// for, not remembered.
rs_ctx_scope::rs_ctx_scope(rs_context* c) : prev(t_scoped) {
    if (c) t_scoped = c;
    const int dev = rs_ctx()->device;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    if (cur != dev && hipSetDevice(dev) == hipSuccess) prevDevice = cur;
}
rs_ctx_scope::~rs_ctx_scope() {
    if (prevDevice >= 0) (void)hipSetDevice(prevDevice);
    t_scoped = prev;
}

int rs_fail(int code, const char* msg) {
    std::lock_guard<std::mutex> lock(g_errMutex);
    g_lastError = msg ? msg : "";
    return code;
}

int rs_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    std::string m = std::string(what ? what : "hip") + ": " + hipGetErrorString(e);
    return rs_fail((int)e, m.c_str());
}

hipStream_t rs_stream() { rs_context* c = rs_ctx(); return c->streamOverride ? c->streamOverride : c->stream; }
bool rs_sync_enabled() { return rs_ctx()->sync; }

// ---- which streams the library makes for itself ------------------------------------------------------------------------------------
// In overlapped mode four streams carry work at any time: the caller's (temporal / spatial passes, tone map, the strip driver's
// transfers) and three of the library's own (the chains of consecutive frames).  Whether four streams of a process really run side by
// side is decided below the API: the runtime maps streams onto four hardware queues PER PRIORITY LEVEL in the order they are made, the
// legacy default stream holds one of the normal level's, and how hardware queues share the command processor's pipes depends on every
// queue the process has made before -- torch's stream pool, an ncclComm's own streams.  Measured on 1/8 strips of 1080p
// (profiles/r05_ab_stream_priority_pools.log, r05_ab_stream_levels_with_rccl.log): three normal-priority streams next to an ordinary caller
// stream 0.29 ms per frame instead of 0.19 (five streams on four queues); the same three at high priority 0.19 -- until a one-rank RCCL
// communicator exists in the process, then 0.39, and only the normal level gives 0.17.  No fixed rule survives all of that, so the
// streams are CHOSEN BY MEASUREMENT when they are first needed (and again after rs_set_stream): four candidates per level, and for
// every triple of a level eight rounds of [a 20-us spin on each of the four streams, an event hand-over to the neighbour] are timed
// (tools/micro/queue_quads.hip is the same test stand-alone: 280-320 us when the four run side by side, 450-860 when two of them take
// turns).  The best triple within 8 % of the fastest is kept, levels in the order of rs_set_internal_stream_priority's preference
// (default: above the caller's stream -- the chains are the frame's critical path, config 3 1.071 -> 1.045 ms; with a denoiser on the
// library stream below it, config 5 1.87 -> 1.78 ms); the other nine streams are destroyed.  About 25 ms, once, with the device idle:
// the one place where the overlapped mode waits on the host.
namespace {
__global__ void k_calibration_spin(long long ticks) {
    const long long t0 = wall_clock64();                                // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
struct Calibration {
    static constexpr int kMax = 8;
    hipEvent_t begin = nullptr, end[kMax] = {}, hand[8][kMax] = {};
    bool ok = true;
    Calibration() {
        ok = hipEventCreate(&begin) == hipSuccess;
        for (auto& e : end) ok = ok && hipEventCreate(&e) == hipSuccess;
        for (auto& r : hand) for (auto& e : r) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    }
    ~Calibration() {
        if (begin) (void)hipEventDestroy(begin);
        for (auto& e : end) if (e) (void)hipEventDestroy(e);
        for (auto& r : hand) for (auto& e : r) if (e) (void)hipEventDestroy(e);
        (void)hipGetLastError();
    }
    // microseconds until all n streams have finished their eight rounds (best of three), or a negative value on error
    double chained(const hipStream_t* s, int n = 4) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            for (int k = 0; k < n; k++) if (hipStreamSynchronize(s[k]) != hipSuccess) return -1;      // (these streams only: other contexts' work goes on)
            (void)hipEventRecord(begin, s[0]);
            for (int r = 0; r < 8; r++) {
                for (int k = 0; k < n; k++) { hipLaunchKernelGGL(k_calibration_spin, dim3(1), dim3(64), 0, s[k], 2000); (void)hipEventRecord(hand[r][k], s[k]); }
                for (int k = 0; k < n; k++) (void)hipStreamWaitEvent(s[k], hand[r][(k + 1) % n], 0);
            }
            for (int k = 0; k < n; k++) (void)hipEventRecord(end[k], s[k]);
            for (int k = 0; k < n; k++) if (hipEventSynchronize(end[k]) != hipSuccess) return -1;
            float worst = 0.f;
            for (int k = 0; k < n; k++) { float t = 0.f; if (hipEventElapsedTime(&t, begin, end[k]) != hipSuccess) return -1; worst = t > worst ? t : worst; }
            best = worst < best ? worst : best;
        }
        return best * 1e3;
    }
};
}  // namespace

// fills c->aux with the three streams chosen by measurement; false: nothing could be measured (the caller falls back to plain streams)
static bool calibrate_internal_streams(rs_context* c) {
    static_assert(rs_context::kAux >= 3 && rs_context::kAux < Calibration::kMax, "three auxiliary streams (measurement builds: up to seven)");
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (rs_context::kAux != 3) {
        // measurement builds with more chains in flight (-DRS_AUX_STREAMS=4): twelve candidates, chosen greedily -- the stream that runs
        // best next to the ones already chosen, one at a time
        Calibration cal;
        if (!cal.ok) { (void)hipGetLastError(); return false; }
        std::vector<hipStream_t> cand;
        for (int p : { greatest, 0, least }) for (int i = 0; i < 4; i++) { hipStream_t st = nullptr; if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, p) == hipSuccess) cand.push_back(st); }
        (void)hipGetLastError();
        hipStream_t chosen[Calibration::kMax] = { c->stream };
        double last = 0;
        for (int k = 1; k <= rs_context::kAux; k++) {
            int best = -1; double bestT = 1e30;
            for (size_t i = 0; i < cand.size(); i++) {
                if (!cand[i]) continue;
                chosen[k] = cand[i];
                const double t = cal.chained(chosen, k + 1);
                if (t >= 0 && t < bestT) { bestT = t; best = (int)i; }
            }
            if (best < 0) { for (hipStream_t st : cand) if (st) (void)hipStreamDestroy(st); return false; }
            chosen[k] = cand[(size_t)best]; cand[(size_t)best] = nullptr; last = bestT;
        }
        for (hipStream_t st : cand) if (st) (void)hipStreamDestroy(st);
        for (int k = 0; k < rs_context::kAux; k++) c->aux[k] = chosen[k + 1];
        c->auxPriority = 0; c->auxCalibratedUs = last; c->auxFastestUs = last;
        return true;
    }
    const int levels[3] = { greatest, 0, least };
    const int nLevels = (greatest < 0 ? 1 : 0) + 1 + (least > 0 ? 1 : 0);
    int order[3], n = 0;                                               // preference: high, normal, low -- or low first when asked for
    if (c->auxLevelSet && c->auxLevel > 0) { if (least > 0) order[n++] = 2; order[n++] = 1; if (greatest < 0) order[n++] = 0; }
    else if (c->auxLevelSet && c->auxLevel == 0) { order[n++] = 1; if (greatest < 0) order[n++] = 0; if (least > 0) order[n++] = 2; }
    else { if (greatest < 0) order[n++] = 0; order[n++] = 1; if (least > 0) order[n++] = 2; }
    (void)nLevels;
    // not the caller's own level: there the library's streams share the level's four hardware queues with the caller's stream and whatever
    // else the process keeps at it, and the eight-round test does not always see it (RCCL in the process, ordinary caller stream, normal
    // level: a calibration time as good as any and a config 5 frame of 2.14 ms instead of 1.79).  The default stream is no such caller: it
    // holds one queue of the normal level, and next to it the normal level measures as well as any.
    if (c->stream) {
        int own = 0;
        if (hipStreamGetPriority(c->stream, &own) != hipSuccess) { (void)hipGetLastError(); own = 0; }
        int m = 0;
        for (int q = 0; q < n; q++) if (levels[order[q]] != own) order[m++] = order[q];
        if (m > 0) n = m;
    }
    Calibration cal;
    if (!cal.ok) { (void)hipGetLastError(); return false; }
    hipStream_t cand[3][4] = {};
    bool made = true;
    for (int q = 0; q < n && made; q++)
        for (int i = 0; i < 4 && made; i++) made = hipStreamCreateWithPriority(&cand[order[q]][i], hipStreamNonBlocking, levels[order[q]]) == hipSuccess;
    double t[3][4];
    double fastest = 1e30;
    for (int q = 0; q < n && made; q++)
        for (int skip = 0; skip < 4; skip++) {
            hipStream_t s[4] = { c->stream };
            int k = 1;
            for (int i = 0; i < 4; i++) if (i != skip) s[k++] = cand[order[q]][i];
            t[order[q]][skip] = cal.chained(s);
            if (t[order[q]][skip] < 0) { made = false; break; }
            fastest = t[order[q]][skip] < fastest ? t[order[q]][skip] : fastest;
        }
    int level = -1, skip = -1;
    for (int q = 0; q < n && made && level < 0; q++) {
        int bestSkip = 0;
        for (int i = 1; i < 4; i++) if (t[order[q]][i] < t[order[q]][bestSkip]) bestSkip = i;
        if (t[order[q]][bestSkip] <= fastest * 1.08) { level = order[q]; skip = bestSkip; }
    }
    int k = 0;
    for (int p = 0; p < 3; p++)
        for (int i = 0; i < 4; i++) {
            if (!cand[p][i]) continue;
            if (p == level && i != skip) c->aux[k++] = cand[p][i];
            else (void)hipStreamDestroy(cand[p][i]);
        }
    (void)hipGetLastError();
    if (level < 0) { for (hipStream_t& st : c->aux) st = nullptr; return false; }
    c->auxPriority = levels[level];
    c->auxCalibratedUs = t[level][skip];
    c->auxFastestUs = fastest;
    return true;
}

// (kept for the streams the strip driver makes for itself, and as the fallback when nothing can be measured)
int rs_internal_stream_priority() {
    rs_context* c = rs_ctx();
    if (c->auxWant != -99) return c->auxWant;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int own = 0;
    if (c->stream && hipStreamGetPriority(c->stream, &own) != hipSuccess) { (void)hipGetLastError(); own = 0; }
    int want = (greatest < 0 && own > greatest) ? greatest : 0;        // the caller's stream is not in the high pool: ours are
    if (c->auxLevelSet) {                                               // the caller's choice, unless it is the caller's own level
        const int asked = c->auxLevel < 0 ? greatest : c->auxLevel > 0 ? least : 0;
        if (asked != own || !c->stream) want = asked;
    }
    c->auxWant = want;
    return want;
}
// Auxiliary streams (asynchronous mode only): 0 carries GBuffer::render, 1 + k the primary-ray + RIS + shadow-ray kernels of every
// kChains-th frame (frames take the chains in turn, so that these chains of consecutive frames overlap each other
// as well as the passes of the frames before); for small launches, whose render is part of the chain's first launch, stream 0 is
// a third chain.
// Neither reads what the temporal / spatial passes of the previous frame write, so with the per-frame surface planes
// in four sets and the G-buffer planes in a ring of five they run next to those passes; the objects own the events
// that order them (rs_gbuffer, rs_restir).
// The auxiliary stream i of the current context, or nullptr when launches are synchronous (rs_set_sync(1): nothing to overlap)
// or the feature is off (rs_set_side_stream(0) / RS_SIDE_STREAM=0).
hipStream_t rs_aux_stream_any(int i) {
    rs_context* c = rs_ctx();
    const bool sync = c->sync;
    c->sync = false;
    const hipStream_t st = rs_aux_stream(i);
    c->sync = sync;
    return st;
}
hipStream_t rs_aux_stream(int i) {
    rs_context* c = rs_ctx();
    if (c->auxMode < 0) {
        // RS_SIDE_STREAM=0 pre-sets rs_set_side_stream(0) for a process that cannot call it: the profiling scripts in tools/ time every
        // kernel alone on one stream under rocprofv3
        const char* e = std::getenv("RS_SIDE_STREAM");
        c->auxMode = (e && e[0] == '0') ? 0 : 1;
    }
    if (c->sync || !c->auxMode || i < 0 || i >= rs_context::kAux) return nullptr;
    const int levelKey = c->auxLevelSet ? c->auxLevel : 2;
    if (c->auxStale) {
        // rs_set_stream handed the library another stream (or rs_set_internal_stream_priority another preference) after the streams were
        // made: they are chosen for the new stream.  The old ones are idle (rs_set_stream waited for them) and are KEPT under the stream and
        // preference they were chosen for -- a caller that alternates between two streams measures each once, not at every switch (at most
        // four choices are kept; rs_choose_internal_streams_again forgets them all).
        bool any = false;
        for (hipStream_t st : c->aux) any = any || st != nullptr;
        if (any && !c->auxPlain && !c->auxForget) {
            rs_context::AuxChoice keep{ c->auxForStream, c->auxForLevel, {}, c->auxPriority, c->auxCalibratedUs, c->auxFastestUs };
            for (int k = 0; k < rs_context::kAux; k++) keep.aux[k] = c->aux[k];
            if (c->auxKept.size() >= 4) { for (hipStream_t st : c->auxKept.front().aux) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } c->auxKept.erase(c->auxKept.begin()); }
            c->auxKept.push_back(keep);
        }
        else {
            for (hipStream_t& st : c->aux) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
            if (c->auxForget) { for (auto& k : c->auxKept) for (hipStream_t st : k.aux) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } c->auxKept.clear(); }
        }
        for (hipStream_t& st : c->aux) st = nullptr;
        c->auxStale = false; c->auxForget = false; c->auxPlain = false;
        for (size_t k = 0; k < c->auxKept.size(); k++)
            if (c->auxKept[k].caller == c->stream && c->auxKept[k].level == levelKey) {
                for (int j = 0; j < rs_context::kAux; j++) c->aux[j] = c->auxKept[k].aux[j];
                c->auxPriority = c->auxKept[k].priority; c->auxCalibratedUs = c->auxKept[k].chosenUs; c->auxFastestUs = c->auxKept[k].fastestUs;
                c->auxForStream = c->stream; c->auxForLevel = levelKey;
                c->auxKept.erase(c->auxKept.begin() + (long)k);
                break;
            }
    }
    if (!c->aux[i]) {
        bool any = false;
        for (hipStream_t st : c->aux) any = any || st != nullptr;
        // a stream that is being captured into a graph cannot be timed (the spins and the waits would end the capture): plain streams
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (c->stream && hipStreamIsCapturing(c->stream, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
        if (!any && !c->auxPlain && capturing == hipStreamCaptureStatusNone && calibrate_internal_streams(c)) { c->auxForStream = c->stream; c->auxForLevel = levelKey; return c->aux[i]; }
        c->auxPlain = true;                                 // nothing could be measured: plain streams at the level the rule names
        c->auxPriority = rs_internal_stream_priority();
        int prio = c->auxPriority;
#ifdef RS_AUX_PRIORITY_ENV                              // measurement builds: RS_AUX_PRIORITIES=-1,-1,0,... per auxiliary stream
        if (const char* e = std::getenv("RS_AUX_PRIORITIES")) { for (int k = 0; k < i && e; k++) { e = std::strchr(e, ','); if (e) e++; } if (e) prio = std::atoi(e); }
#endif
        if (hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); c->aux[i] = nullptr; c->auxMode = 0; return nullptr; }
    }
    return c->aux[i];
}
// In asynchronous mode GBuffer::render can be deferred and launched by ReSTIRDirect together with its primary rays (the two rays
// of a pixel in one packet walk).  That saves walk work on a full frame, but the slowest tile of the launch takes longer, which
// costs on scenes whose closest-hit kernels are tails of a few long tiles: by default every rs_restir measures the
// frame period both ways once and keeps the faster (restir.hip); rs_set_side_stream(1) / (2) force it off / on.
// 0 never, 1 always (launches of at least three rounds of wave slots), 2 always (any size), 3 decided per rs_restir by measuring
int rs_fuse_mode() {
    rs_context* c = rs_ctx();
    if (c->fuseMode < 0) c->fuseMode = 3;
    return c->fuseMode;
}
bool rs_fuse_enabled() { return rs_fuse_mode() != 0; }
const rs_context* rs_stream_plan() {
    rs_context* c = rs_ctx();
    if (c->chainStreams < 0) c->chainStreams = 2;          // the defaults of rs_set_stream_plan
    if (c->smallChains < 0) c->smallChains = 1;
    if (c->shadowOnMain < 0) c->shadowOnMain = 2;
    return c;
}
int rs_ris_global_below() {
    rs_context* c = rs_ctx();
    if (c->risGlobalBelow < 0) c->risGlobalBelow = 64 * 1024;
    return c->risGlobalBelow;
}
int rs_aux_synchronize() {
    for (hipStream_t st : rs_ctx()->aux) if (st) RS_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- the denoise stream ---------------------------------------------------------------------------------------------------------------
// With a filter in the loop the library stream is the one chain that links consecutive frames AND carries the most work (config 5 at
// N = 1: shadow rays 794 us + temporal 208 + spatial 481 + five a-trous levels 243 + tone map, 98 % busy while the chain streams idle at
// 51-57 %, profiles/r05_config5_strip_gpu_paced_traces.txt).  rs_set_denoise_stream(1) gives the filter -- and the tone map that reads
// its result -- auxiliary stream 0: the levels of frame f wait for phase B of f by an event and run next to the temporal / spatial passes
// of f + 1.  A fifth stream with work in flight halves the frame rate, so the chains then take turns on two streams.
int rs_chains_in_flight() {
    const rs_context* c = rs_ctx();
    const int n = rs_context::kAux - (c->ownCommStreams > 0 ? 1 : 0) - (c->denoiseMode == 1 ? 1 : 0);
    return n < 1 ? 1 : n;
}
hipStream_t rs_denoise_stream() {
    rs_context* c = rs_ctx();
    if (c->denoiseMode == 0 || c->sync) return nullptr;
    return rs_aux_stream(0);                            // (null: the auxiliary streams are switched off)
}
rs_denoise_scope::rs_denoise_scope(bool fork, bool enable) {
    c = rs_ctx();
    if (!enable || c->streamOverride) return;           // nested: already there
    const hipStream_t d = rs_denoise_stream();
    if (!d) return;
    if (!c->denoiseFork) err = rs_check_hip(hipEventCreateWithFlags(&c->denoiseFork, hipEventDisableTiming), "hipEventCreate");
    if (!err && fork) {
        err = rs_check_hip(hipEventRecord(c->denoiseFork, c->stream), "denoise stream: fork");
        if (!err) err = rs_check_hip(hipStreamWaitEvent(d, c->denoiseFork, 0), "denoise stream: fork");
    }
    if (err) return;
    c->streamOverride = d; active = true;
}
rs_denoise_scope::~rs_denoise_scope() {
    if (!active) return;
    c->denoiseUsed = true;
    c->streamOverride = nullptr;
}

namespace {
inline bool inside(const rs_context::DenoiseBuf& b, const void* p) { return (const char*)p >= b.base && (const char*)p < b.base + b.bytes; }
}
int rs_denoise_mark(const void* base, size_t bytes, bool readOnly) {
    rs_context* c = rs_ctx();
    if (!c->streamOverride || !base || bytes == 0) return 0;
    rs_context::DenoiseBuf* e = nullptr;
    for (auto& b : c->denoiseBufs) if (b.base == (const char*)base) { e = &b; break; }
    if (!e) {
        if (c->denoiseBufs.size() >= 32) {              // a caller that hands over ever new buffers: order the library stream after the oldest and reuse its entry
            e = &c->denoiseBufs[0];
            for (auto& b : c->denoiseBufs) if (!b.pending) { e = &b; break; }
            if (e->pending) RS_HIP(hipStreamWaitEvent(c->stream, e->ev, 0));
        }
        else {
            hipEvent_t ev = nullptr;
            RS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->denoiseBufs.push_back({ nullptr, 0, ev, true, false });
            e = &c->denoiseBufs.back();
        }
        e->base = (const char*)base; e->bytes = 0; e->readOnly = true; e->pending = false;
    }
    e->bytes = bytes > e->bytes ? bytes : e->bytes;
    e->readOnly = e->pending ? (e->readOnly && readOnly) : readOnly;
    e->pending = true;
    RS_HIP(hipEventRecord(e->ev, c->streamOverride));
    return 0;
}
bool rs_denoise_owns(const void* p) {
    const rs_context* c = rs_ctx();
    for (const auto& b : c->denoiseBufs) if (b.pending && !b.readOnly && inside(b, p)) return true;
    return false;
}
int rs_denoise_order(const void* p, bool write) {
    rs_context* c = rs_ctx();
    if (c->denoiseBufs.empty() || c->streamOverride || !p) return 0;      // (inside a scope: in that stream's order anyway)
    for (auto& b : c->denoiseBufs) {
        if (!b.pending || !inside(b, p) || (b.readOnly && !write)) continue;
        RS_HIP(hipStreamWaitEvent(c->stream, b.ev, 0));
        b.pending = false;
    }
    return 0;
}
int rs_denoise_join() {
    rs_context* c = rs_ctx();
    if (c->streamOverride || !c->denoiseUsed || !c->aux[0]) return 0;    // (the streams are only ever replaced behind an rs_synchronize, which clears denoiseUsed)
    if (!c->denoiseFork) RS_HIP(hipEventCreateWithFlags(&c->denoiseFork, hipEventDisableTiming));
    RS_HIP(hipEventRecord(c->denoiseFork, c->aux[0]));
    RS_HIP(hipStreamWaitEvent(c->stream, c->denoiseFork, 0));
    for (auto& b : c->denoiseBufs) b.pending = false;
    return 0;
}

int rs_after_launch(const char* what) {
    RS_TRY(rs_check_hip(hipGetLastError(), what));
    if (rs_ctx()->sync) RS_TRY(rs_check_hip(hipStreamSynchronize(rs_ctx()->stream), what));
    return 0;
}

rs::CamParams rs_make_cam_params(const rs_camera* cam) {
    using namespace rs;
    CamParams c;
    c.position = ld3(cam->position);
    c.right = ld3(cam->right);
    c.up = ld3(cam->up);
    c.view = ld3(cam->view);
    c.inv0 = ld3(cam->rotationMatInv);
    c.inv1 = ld3(cam->rotationMatInv + 3);
    c.inv2 = ld3(cam->rotationMatInv + 6);
    c.width = cam->resolution[0];
    c.height = cam->resolution[1];
    c.aspect = (float)cam->resolution[0] / (float)cam->resolution[1];
    c.tanFovY = tanf(radians(cam->fov[1]));            // glm::tan(glm::radians(fov.y)), sceneStructs.h:72
    c.focalDist = cam->focalDist;
    c.lensRadius = cam->lensRadius;
    c.pixelSizeX = 1.f / (float)cam->resolution[0];
    c.pixelSizeY = 1.f / (float)cam->resolution[1];
    return c;
}

// ---- tile-split hints ---------------------------------------------------------------------------------------------------------
int rs_tile_split_threshold() {
    rs_context* c = rs_ctx();
    if (!c->tileSplitSet) { c->tileSplit = 768; c->tileSplitSet = true; }
    return c->tileSplit;
}
void rs_tile_split_free(rs_tile_split* t) {
    rs_dev_free(t->base);
    if (t->report) { (void)hipHostFree(t->report); t->report = nullptr; }
    t->bytes = 0; t->key = -1; t->rot = 0; t->numTiles = t->capacity = 0;
    t->issued = t->wake = 0; t->sleep = 0; t->lastNone = false;
}
constexpr int kTileSplitSleep = 29;
int rs_tile_split_prepare(rs_tile_split* t, long long key, int numTiles, int regularBlocks, int mode, hipStream_t st, rs::TileSplit* ts, int* helperBlocks) {
    *ts = rs::TileSplit{ nullptr, 0, 0 };
    *helperBlocks = 0;
    int threshold = rs_tile_split_threshold();
    if (threshold == 0 || numTiles <= 0 || (threshold > 0 && mode == 0)) return 0;
    const bool adaptive = threshold > 0 && mode == 2;
    bool fresh = false;
    unsigned found = 0;
    if (adaptive) {
        threshold += threshold / 3;
        // The launch with sequence number q reports what launch q - 1 found, and the first launch after a (re)start reads a stale list:
        // the first fresh report is that of launch wake + 2.  Never waited for -- the host may be a hundred frames ahead of the device
        // (bench.py enqueues all its timed frames before it waits) -- so a site whose last fresh report said "none" goes back to the
        // plain kernels after its two probing launches without waiting for their report, and a report that does name heavy tiles ends
        // that sleep when it arrives.
        if (t->report) {
            const unsigned long long rep = *reinterpret_cast<volatile unsigned long long*>(t->report);
            const unsigned seq = (unsigned)(rep >> 32);
            fresh = rep != ~0ull && seq >= t->wake + 2 && seq <= t->issued;
            found = (unsigned)rep;
            if (fresh) t->lastNone = found == 0;
        }
        if (t->sleep > 0) {
            if (fresh && found > 0) { t->sleep = 0; t->wake = t->issued; fresh = false; }
            else {
                if (--t->sleep == 0) t->wake = t->issued;
                return 0;                               // a plain launch
            }
        }
    }
    if (threshold < 0) threshold = -threshold;      // negative: for every launch, also next to other kernels (tests, measurements)
    const int capacity = std::min(rs_tile_split::kCapacity, std::max(64, regularBlocks / 2));
    key = key * 1048573 + threshold;
    const size_t hintInts = 2 + (size_t)capacity, flagOffset = (8 + 3 * hintInts) * sizeof(int), flagStride = ((size_t)numTiles + 15) & ~(size_t)15;
    const size_t bytes = flagOffset + 3 * flagStride;
    if (t->bytes < bytes) {                         // (hipFree waits for the device: nothing in flight reads the old arrays)
        rs_dev_free(t->base);
        t->bytes = 0; t->key = -1;                  // (if the allocation below fails, the next launch tries again instead of clearing a null array)
        unsigned char* p = nullptr;
        RS_TRY(rs_dev_alloc(&p, bytes));
        t->base = reinterpret_cast<int*>(p); t->bytes = bytes; t->key = -1;
    }
    if (!t->report) {
        RS_HIP(hipHostMalloc((void**)&t->report, sizeof(unsigned long long), hipHostMallocDefault));
        *t->report = ~0ull;
    }
    if (t->key != key || t->numTiles != numTiles || t->capacity != capacity) {      // another geometry: no hints
        RS_HIP(hipMemsetAsync(t->base, 0, bytes, st));
        hipLaunchKernelGGL(rs::k_tile_split_init, dim3(1), dim3(1), 0, st, t->base, capacity, threshold, numTiles, (int)flagOffset, (int)flagStride, t->report);
        t->key = key; t->rot = 0; t->numTiles = numTiles; t->capacity = capacity;
        t->wake = t->issued;                        // (the sequence numbers go on: reports of launches before this point stay behind `wake`)
    }
    t->issued = (t->issued + 1) & 0x0fffffffu;
    if (t->issued == 0) t->wake = 0;
    ts->base = t->base; ts->rot = t->rot + 3 * (int)t->issued; ts->helperBlocks = capacity;
    *helperBlocks = capacity;
    t->rot = (t->rot + 1) % 3;
    if (adaptive && ((fresh && found == 0) || (!fresh && t->lastNone && t->issued - t->wake >= 2))) t->sleep = kTileSplitSleep;
    return 0;
}

extern "C" {

int rs_set_tile_split(int threshold) {
    rs_ctx()->tileSplit = threshold; rs_ctx()->tileSplitSet = true;
    return 0;
}


const char* rs_last_error(void) {
    std::lock_guard<std::mutex> lock(g_errMutex);
    static thread_local std::string copy;
    copy = g_lastError;
    return copy.c_str();
}

int rs_init(int device) {
    int n = 0;
    RS_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_init: no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= n) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_init: device index out of range");
    RS_HIP(hipSetDevice(device));
    g_default.device = device;
    return 0;
}

int rs_context_create(int device, rs_context** out) {
    if (!out) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_context_create: null output");
    *out = nullptr;
    int n = 0;
    RS_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_context_create: device index out of range");
    rs_context* c = new rs_context();
    c->device = device;
    *out = c;
    return 0;
}
int rs_context_destroy(rs_context* c) {
    if (!c || c == &g_default) return 0;
    {
        rs_ctx_scope scope(c);
        (void)rs_synchronize();
        for (hipStream_t& st : c->aux) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
        for (auto& k : c->auxKept) for (hipStream_t st : k.aux) if (st) (void)hipStreamDestroy(st);
        c->auxKept.clear();
        if (c->ptRayCount) { (void)hipFree(c->ptRayCount); c->ptRayCount = nullptr; }
        for (auto& b : c->denoiseBufs) if (b.ev) (void)hipEventDestroy(b.ev);
        c->denoiseBufs.clear();
        if (c->denoiseFork) { (void)hipEventDestroy(c->denoiseFork); c->denoiseFork = nullptr; }
    }
    if (t_current == c) t_current = nullptr;
    delete c;
    return 0;
}
int rs_context_set_current(rs_context* c) {
    t_current = c;
    return rs_check_hip(hipSetDevice(rs_ctx()->device), "rs_context_set_current");      // the context's device becomes this thread's device

}

int rs_set_stream(void* hipStream) {
    rs_context* c = rs_ctx();
    const bool changed = (hipStream_t)hipStream != c->stream;
    if (changed) RS_TRY(rs_synchronize());              // events recorded on the old stream order the auxiliary ones
    c->stream = (hipStream_t)hipStream;
    c->auxWant = -99;
    if (changed) c->auxStale = true;                    // the library's own streams were chosen next to the old stream: chosen again on next use
    return 0;
}
int rs_set_sync(int sync) { rs_ctx()->sync = sync != 0; return 0; }
// what the choice by measurement came to (for logs): the level of the three streams (-1 high / 0 normal / 1 low), the chosen triple's
// calibration time and the fastest time any candidate triple reached, in microseconds (0 / 0: not chosen by measurement -- none made yet,
// or the measurement failed and plain streams are in use)
int rs_internal_streams_info(int* priority, double* chosenUs, double* fastestUs) {
    rs_context* c = rs_ctx();
    if (priority) *priority = c->auxPriority;
    if (chosenUs) *chosenUs = c->auxPlain ? 0.0 : c->auxCalibratedUs;
    if (fastestUs) *fastestUs = c->auxPlain ? 0.0 : c->auxFastestUs;
    return 0;
}
// Makes the choice again at the next overlapped launch: for a caller whose process has gained streams since the library chose (an
// ncclComm created after the first frames, a torch stream pool touched for the first time).  Waits for the library's streams.
int rs_choose_internal_streams_again(void) {
    rs_context* c = rs_ctx();
    RS_TRY(rs_synchronize());
    c->auxStale = true; c->auxForget = true; c->auxPlain = false; c->auxWant = -99;
    return 0;
}
// Makes the choice NOW (about 25 ms: spins on the caller's stream and on twelve candidate streams, waits for those streams only) instead of
// at the first overlapped launch -- for a caller that wants no such pause inside its first frame.  No-op when launches are synchronous,
// the side streams are off, or a choice for the current stream and preference is in force.
int rs_prepare_streams(void) {
    rs_ctx_scope scope(nullptr);
    (void)rs_aux_stream(0);
    return 0;
}
int rs_set_internal_stream_priority(int level) {
    rs_context* c = rs_ctx();
    if (level < -1 || level > 2) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_set_internal_stream_priority: -1 high, 0 normal, 1 low, 2 automatic");
    const bool changed = c->auxLevelSet != (level != 2) || (level != 2 && c->auxLevel != level);
    c->auxLevelSet = level != 2; c->auxLevel = level == 2 ? 0 : level;
    c->auxWant = -99;
    if (changed) { RS_TRY(rs_synchronize()); c->auxStale = true; }
    return 0;
}
int rs_set_side_stream(int enable) {
    rs_context* c = rs_ctx();
    c->auxMode = enable ? 1 : 0;                        // work already enqueued on the auxiliary streams is still joined by its consumers
    c->fuseMode = enable == 2 ? 1 : enable == 3 ? 2 : enable == 4 ? 3 : 0;      // 2 always, 3 always and at any size (tests), 4 measured
    return 0;
}
// How the asynchronous mode spreads a frame's kernels over the auxiliary streams; every value -1 = keep.
//   chainStreams  1: one stream for every frame's primary -> RIS -> shadow chain; 2: frames alternate between two (default)
//   smallChains   0 / 1: a launch below three rounds of wave slots (a strip) fuses the render with the primary rays and rotates
//                 its chains over three streams (default 1)
//   shadowOnMain  0 never / 1 always / 2 for launches that fill the chip three times over (default): the shadow rays on the library stream
int rs_set_stream_plan(int chainStreams, int smallChains, int shadowOnMain) {
    rs_context* c = rs_ctx();
    if (chainStreams > 2 || smallChains > 1 || shadowOnMain > 2 || chainStreams == 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_set_stream_plan: value out of range");
    (void)rs_stream_plan();                             // resolve the defaults first
    if (chainStreams >= 1) c->chainStreams = chainStreams;
    if (smallChains >= 0) c->smallChains = smallChains;
    if (shadowOnMain >= 0) c->shadowOnMain = shadowOnMain;
    return 0;
}
int rs_set_ris_table_pixels(int pixels) {
    if (pixels < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_set_ris_table_pixels: negative");
    rs_ctx()->risGlobalBelow = pixels;
    return 0;
}
int rs_synchronize(void) {
    rs_ctx_scope scope(nullptr);
    RS_TRY(rs_aux_synchronize());
    RS_TRY(rs_check_hip(hipStreamSynchronize(rs_ctx()->stream), "rs_synchronize"));
    RS_TRY(rs_aux_synchronize());                       // (an auxiliary launch may have been waiting for the library stream)
    for (auto& b : rs_ctx()->denoiseBufs) b.pending = false;      // everything the denoise stream was handed has finished
    rs_ctx()->denoiseUsed = false;
    return 0;
}
// LeveledEAWFilter (rs_eaw_filter, rs_strips_eaw_filter) and an rs_copy_image_to_pbo that reads its result on a stream of the library
// (asynchronous launches only): 1 on, 0 off (default).  Their results are then ordered for the caller's stream by events -- every library
// call that is handed one of those buffers waits as needed, and rs_join_denoise_stream() / rs_synchronize() do for the caller's own work.
int rs_set_denoise_stream(int enable) {
    rs_ctx_scope scope(nullptr);
    rs_context* c = rs_ctx();
    if (enable != 0 && enable != 1) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_set_denoise_stream: 0 or 1");
    if (c->denoiseMode == enable) return 0;
    RS_TRY(rs_synchronize());
    c->denoiseMode = enable;
    return 0;
}
int rs_join_denoise_stream(void) {
    rs_ctx_scope scope(nullptr);
    return rs_denoise_join();
}

}  // extern "C"
