// strips.hip -- the row-strip frame of one rank for a C / C++ caller (include/restir_hip.h rs_comm_*, rs_strips_*):
// the framebuffer of runCuda (src/main.cpp:146-185) cut into `world` row strips, one process per GPU.
//
//     GBuffer::render on the strip's rows
//     ReSTIRDirect phase A on the strip's rows                      (restir.cu:127-194)
//     5 border rows of published reservoirs + G-buffer id / normal / depth  ->  the strip above / below      <- the only exchange
//     phase B (restir.cu:196-230): in one launch after the rows have arrived when the transfers share the library stream (default); with
//     the transfers on a stream of the driver on the interior rows while the border rows travel, then on the two 5-row bands
//
// The exchange goes through a transport of two operations, send and recv of a device buffer, grouped:
//   * RCCL: ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd on a stream of this driver that is ordered after the packing
//     copies and before the unpacking copies by events, so that the interior rows of phase B (library stream) run while the rows
//     travel over xGMI.  librccl is opened at run time (dlopen): a single-GPU caller does not need it.
//   * caller-supplied callbacks (rs_comm_create): tests drive the same frame code with torch.distributed / gloo underneath.
// restir_amd/tiling.py is the Python form of the same schedule (test harness, bench.py); tests compare the two.
//
// Around the frame, with the same transport: rs_strips_eaw_filter (LeveledEAWFilter on the strip: 32 G-buffer rows once, then the
// 2 << level border rows of every level's input), rs_strips_exchange_history (moving camera: the rows the next temporal merge may
// reproject into travel to every rank) and rs_strips_gather (image assembly).  Rows of a row-major image are contiguous, so the
// colour rows and the assembled image travel from and into place without packing.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "rs_internal.h"

struct rs_comm {
    rs_transport t{};
    int rank = 0, world = 1;
    // RCCL transport
    void* lib = nullptr;
    void* nccl = nullptr;                          // ncclComm_t
    int (*pSend)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*pRecv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*pGroupStart)() = nullptr;
    int (*pGroupEnd)() = nullptr;
    const char* (*pErr)(int) = nullptr;
};

struct Xfer { bool send; void* buf; size_t bytes; int peer; };

struct rs_strips {
    rs_context* ctx = nullptr;
    rs_comm* comm = nullptr;
    int width = 0, height = 0;
    std::vector<int> bounds;                       // world + 1 row offsets
    int y0 = 0, y1 = 0;
    size_t haloBytes = 0;                          // one edge: reservoirs + G-buffer rows
    // How many rows of the G-buffer id / normal / depth planes travel with the 5 reservoir rows of an edge (rs_strips_set_gbuffer_halo): 5 is
    // what the spatial taps need; a caller whose frame goes on with a denoiser sets 32, and the filter finds its G-buffer rows already there
    // (one packing launch, one group and one unpacking launch less per frame than exchanging them again: gbufFresh)
    int gReach = RS_SPATIAL_HALO_ROWS;
    bool gbufFresh = false;                        // the current G-buffer set holds the neighbours' gReach rows (set by rs_strips_frame, consumed by the filters)
    char* sendUp = nullptr; char* recvUp = nullptr; char* sendDown = nullptr; char* recvDown = nullptr;
    // Where the transfers are enqueued.  commOnMain (default): on the library stream itself, in order with the packing copies before
    // and the unpacking copies after them -- no extra stream, no events.  The overlapped mode already keeps four streams busy (the
    // library stream and three chains), which is the number of hardware queues: a fifth stream shares a queue with one of the chains
    // and serialises two of the three (a 1/8 strip of 1080p through this driver with a transport that does nothing: 0.289 ms per
    // frame with a separate stream against 0.19 without; tools/host_enqueue_strips.py, tools/strip_trace_c.py).  The price: the
    // library stream does not run the interior rows of phase B while the rows travel -- a few microseconds of kernel on a strip.
    // rs_strips_set_comm_stream(s, 1): the transfers on a stream of their own, ordered by events (round 2's form).
    bool commOnMain = true;
    hipStream_t commStream = nullptr;              // carries the transfers when !commOnMain: ONE stream, so that every rank issues its groups in one order
    // rs_strips_gather_begin with the transfers in a stream's own order: they ride in the next group on that stream -- [0] the library
    // stream (the next frame's border rows), [1] the denoise stream (rs_set_denoise_stream: the tone map of a filtered image runs there,
    // and the next group there is the next frame's filter exchanging G-buffer rows)
    std::vector<Xfer> deferred[2];
    int deferredSlots[2] = { 0, 0 };               // bit per gather slot whose transfers are still in `deferred`
    bool gatherOnDenoise[4] = {};                  // per slot: its rows were written, and its transfers are ordered, on the denoise stream
    bool ownStreamCounted = false;                 // this driver is one of rs_context::ownCommStreams
    hipEvent_t packed = nullptr, arrived = nullptr;
    static constexpr int kGatherSlots = 4;
    hipEvent_t gathered[kGatherSlots] = {};        // rs_strips_gather_begin / _end: the gather of a slot has finished
    bool gatherPending[kGatherSlots] = {};
    // rs_strips_enable_timing: how long the library stream sat idle in join() of the last frame's halo exchange (events on the library stream around the wait)
    bool timing = false;
    hipEvent_t waitFrom = nullptr, waitTo = nullptr;
    bool waitValid = false;
    // rs_strips_eaw_filter: the two full-frame buffers the levels alternate between, staging for the 32 G-buffer rows of an edge
    float* eawBuf[2] = { nullptr, nullptr };
    char* eawSend[2] = { nullptr, nullptr }; char* eawRecv[2] = { nullptr, nullptr };
    // rs_strips_exchange_history: this strip's rows packed, every other strip's rows as they arrive (rank order)
    char* histSend = nullptr; char* histRecv = nullptr;
    size_t histSendBytes = 0, histRecvBytes = 0;
};

namespace {

constexpr int kHalo = RS_SPATIAL_HALO_ROWS;

int rccl_fail(rs_comm* c, int e, const char* what) {
    std::string m = std::string(what) + ": " + (c->pErr ? c->pErr(e) : "RCCL error");
    return rs_fail(RS_ERR_UNSUPPORTED, m.c_str());
}
int rccl_group_begin(void* ctx) { rs_comm* c = (rs_comm*)ctx; const int e = c->pGroupStart(); return e ? rccl_fail(c, e, "ncclGroupStart") : 0; }
int rccl_group_end(void* ctx) { rs_comm* c = (rs_comm*)ctx; const int e = c->pGroupEnd(); return e ? rccl_fail(c, e, "ncclGroupEnd") : 0; }
int rccl_send(void* ctx, const void* buf, size_t bytes, int peer, void* stream) {
    rs_comm* c = (rs_comm*)ctx;
    const int e = c->pSend(buf, bytes, /*ncclUint8*/ 1, peer, c->nccl, (hipStream_t)stream);
    return e ? rccl_fail(c, e, "ncclSend") : 0;
}
int rccl_recv(void* ctx, void* buf, size_t bytes, int peer, void* stream) {
    rs_comm* c = (rs_comm*)ctx;
    const int e = c->pRecv(buf, bytes, /*ncclUint8*/ 1, peer, c->nccl, (hipStream_t)stream);
    return e ? rccl_fail(c, e, "ncclRecv") : 0;
}

// Packing and unpacking the border rows of a frame: 6 planes (published reservoirs li / wi / tap, G-buffer id / normal / depth) x 2
// edges.  As 24 hipMemcpyAsync calls per frame they cost the HOST 0.12 ms -- more than half of what a 1/8 strip's kernels last
// (tools/host_enqueue_strips.py: 0.210 ms of host time per frame against 0.18 ms of kernels) -- so all segments of a direction go
// through ONE launch of a copy kernel that finds its segment from a table passed by value.
constexpr int kMaxSegs = 12;
struct CopyTable { const char* src[kMaxSegs]; char* dst[kMaxSegs]; unsigned start[kMaxSegs + 1]; int n; };      // start: in units of `unit` bytes
template <typename T>
__global__ void __launch_bounds__(256) k_copy_segments(CopyTable t) {
    RS_SETPRIO(RS_PRIO_STREAM);
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= t.start[t.n]) return;
    int k = 0;
    while (i >= t.start[k + 1]) k++;
    reinterpret_cast<T*>(t.dst[k])[i - t.start[k]] = reinterpret_cast<const T*>(t.src[k])[i - t.start[k]];
}
struct SegList {
    const char* a[kMaxSegs]; char* b[kMaxSegs]; size_t bytes[kMaxSegs]; int n = 0;
    void add(const void* plane, void* packed, size_t nbytes) { a[n] = (const char*)plane; b[n] = (char*)packed; bytes[n] = nbytes; n++; }
};
// pack: plane -> packed buffer; unpack: packed buffer -> plane
int copy_segments(const SegList& l, bool pack) {
    if (l.n == 0) return 0;
    bool wide = true;
    for (int k = 0; k < l.n; k++) wide = wide && l.bytes[k] % 16 == 0 && ((size_t)l.a[k] % 16 == 0) && ((size_t)l.b[k] % 16 == 0);
    const unsigned unit = wide ? 16u : 4u;
    CopyTable t; t.n = l.n; t.start[0] = 0;
    for (int k = 0; k < l.n; k++) {
        t.src[k] = pack ? l.a[k] : l.b[k];
        t.dst[k] = pack ? l.b[k] : const_cast<char*>(l.a[k]);
        t.start[k + 1] = t.start[k] + (unsigned)(l.bytes[k] / unit);
    }
    const unsigned total = t.start[t.n];
    if (total == 0) return 0;
    if (wide) hipLaunchKernelGGL(k_copy_segments<uint4>, dim3((total + 255) / 256), dim3(256), 0, rs_stream(), t);
    else hipLaunchKernelGGL(k_copy_segments<unsigned>, dim3((total + 255) / 256), dim3(256), 0, rs_stream(), t);
    return rs_check_hip(hipGetLastError(), "strip border rows");
}
// the border rows of one edge in the packed layout of rs_restir_halo_pack followed by rs_gbuffer_rows_pack: li, wi, tap, id, normal, depth
// (yResv: first of the kHalo reservoir rows; yG: first of the gRows G-buffer rows -- the same edge of the strip, so the two ranges end or begin together)
void halo_segments(SegList& l, rs_restir* r, rs_gbuffer* g, int yResv, int yG, int gRows, char* packed) {
    const size_t n = (size_t)r->width * kHalo, off = (size_t)yResv * r->width;
    const int c = g->cur();
    l.add(r->temp.li + off, packed, n * 16); l.add(r->temp.wi + off, packed + n * 16, n * 16); l.add(r->temp.tap + off, packed + n * 32, n * 16);
    char* gb = packed + n * 48;
    const size_t m = (size_t)r->width * gRows, offG = (size_t)yG * r->width;
    l.add(g->primId[c] + offG, gb, m * 4); l.add(g->normal[c] + offG * 3, gb + m * 4, m * 12); l.add(g->depth[c] + offG, gb + m * 16, m * 4);
}

// One grouped exchange, ordered after everything enqueued on the library stream so far: on the driver's stream for a stream-ordered
// transport (RCCL), from the host side of a finished library stream otherwise.  join() makes the library stream continue after it.
int post(rs_strips* s, const Xfer* opsIn, size_t nIn, int list = -1) {
    const rs_comm* c = s->comm;
    // transfers of a gather that was begun in this stream's order travel in this group (one RCCL launch per frame instead of two)
    const int which = list >= 0 ? list : (rs_ctx()->streamOverride ? 1 : 0);      // posting from inside a denoise scope: that stream's list (or the list the caller names)
    std::vector<Xfer> merged;
    const Xfer* ops = opsIn; size_t n = nIn;
    int carried = 0;
    if (!s->deferred[which].empty()) {
        merged.assign(opsIn, opsIn + nIn);
        merged.insert(merged.end(), s->deferred[which].begin(), s->deferred[which].end());
        s->deferred[which].clear(); carried = s->deferredSlots[which]; s->deferredSlots[which] = 0;
        ops = merged.data(); n = merged.size();
    }
    if (n == 0) return 0;
    hipStream_t ts = s->commOnMain ? rs_stream() : s->commStream;
    if (c->t.stream_ordered) { if (!s->commOnMain) { RS_HIP(hipEventRecord(s->packed, rs_stream())); RS_HIP(hipStreamWaitEvent(ts, s->packed, 0)); } }
    else RS_TRY(rs_synchronize());
    if (c->t.group_begin) RS_TRY(c->t.group_begin(c->t.ctx));
    // a group that was begun is always ended (an open ncclGroup would swallow every later call of this thread); the first error is reported
    int err = 0;
    for (size_t i = 0; i < n && !err; i++) {
        if (ops[i].send) err = c->t.send(c->t.ctx, ops[i].buf, ops[i].bytes, ops[i].peer, ts);
        else err = c->t.recv(c->t.ctx, ops[i].buf, ops[i].bytes, ops[i].peer, ts);
    }
    std::string firstError;
    if (err) firstError = rs_last_error();
    if (c->t.group_end) {                                        // RCCL: the grouped transfers are enqueued here; a host-side transport completes here
        const int e2 = c->t.group_end(c->t.ctx);
        if (err) return rs_fail(err, firstError.c_str());
        if (e2) return e2;
    }
    else if (err) return err;
    // the gathers this group carried have travelled when the carrying stream gets here (rs_strips_gather_end)
    if (c->t.stream_ordered)
        for (int slot = 0; slot < rs_strips::kGatherSlots; slot++)
            if (carried & (1 << slot)) RS_HIP(hipEventRecord(s->gathered[slot], ts));
    return 0;
}
int join(rs_strips* s, bool timed);
// the `reach` rows of the current G-buffer id / normal / depth planes beyond each edge of the strip, from the neighbouring strips
// (what the taps of a denoiser compare against; rs_strips_eaw_filter, rs_strips_svgf_filter)
int exchange_gbuffer_rows(rs_strips* s, rs_gbuffer* g, int reach) {
    const rs_comm* c = s->comm;
    const int y0 = s->y0, y1 = s->y1;
    const bool up = c->rank > 0, down = c->rank + 1 < c->world;
    if (!up && !down) return 0;
    const size_t gBytes = rs_gbuffer_rows_bytes(g, reach);
    for (int i = 0; i < 2; i++)
        if ((i == 0 ? up : down) && !s->eawSend[i]) { RS_TRY(rs_dev_alloc(&s->eawSend[i], gBytes)); RS_TRY(rs_dev_alloc(&s->eawRecv[i], gBytes)); }
    // the packed layout of rs_gbuffer_rows_pack: id rows, normal rows, depth rows -- all six segments of a direction in ONE copy launch
    // (as 12 hipMemcpyAsync calls they were a dozen launches of ~8 us on the library stream of every filtered frame)
    RS_TRY(rs_gbuffer_join(g));
    const int cur = g->cur();
    const auto segments = [&](SegList& l, int y, char* packed) {
        const size_t n = (size_t)s->width * reach, off = (size_t)y * s->width;
        l.add(g->primId[cur] + off, packed, n * 4); l.add(g->normal[cur] + off * 3, packed + n * 4, n * 12); l.add(g->depth[cur] + off, packed + n * 16, n * 4);
    };
    Xfer ops[4]; size_t n = 0;
    {
        SegList l;
        if (up) { segments(l, y0, s->eawSend[0]); ops[n++] = { true, s->eawSend[0], gBytes, c->rank - 1 }; ops[n++] = { false, s->eawRecv[0], gBytes, c->rank - 1 }; }
        if (down) { segments(l, y1 - reach, s->eawSend[1]); ops[n++] = { true, s->eawSend[1], gBytes, c->rank + 1 }; ops[n++] = { false, s->eawRecv[1], gBytes, c->rank + 1 }; }
        RS_TRY(copy_segments(l, true));
    }
    RS_TRY(post(s, ops, n));
    RS_TRY(join(s, false));
    {
        SegList l;
        if (up) segments(l, y0 - reach, s->eawRecv[0]);
        if (down) segments(l, y1, s->eawRecv[1]);
        RS_TRY(copy_segments(l, false));
    }
    return 0;
}
// the first / last `rows` rows of the strip's part of a row-major image (and of a second one) to the strips above / below, theirs
// into the rows just outside the strip: sent from and received into place
int exchange_image_rows(rs_strips* s, float* a, int ca, float* b, int cb, int rows) {
    const rs_comm* c = s->comm;
    const int W = s->width, y0 = s->y0, y1 = s->y1;
    const bool up = c->rank > 0, down = c->rank + 1 < c->world;
    if (!up && !down) return 0;
    Xfer ops[8]; size_t n = 0;
    float* img[2] = { a, b }; const int comp[2] = { ca, cb };
    for (int k = 0; k < 2; k++) {
        if (!img[k]) continue;
        const size_t row = (size_t)W * comp[k], bytes = row * rows * sizeof(float);
        if (up) { ops[n++] = { true, img[k] + (size_t)y0 * row, bytes, c->rank - 1 }; ops[n++] = { false, img[k] + (size_t)(y0 - rows) * row, bytes, c->rank - 1 }; }
        if (down) { ops[n++] = { true, img[k] + (size_t)(y1 - rows) * row, bytes, c->rank + 1 }; ops[n++] = { false, img[k] + (size_t)y1 * row, bytes, c->rank + 1 }; }
    }
    RS_TRY(post(s, ops, n));
    return join(s, false);
}
int svgf_exchange_hook(void* ctx, float* a, int ca, float* b, int cb, int rows) { return exchange_image_rows((rs_strips*)ctx, a, ca, b, cb, rows); }

int join(rs_strips* s, bool timed) {
    if (s->comm->t.stream_ordered && !s->commOnMain) {
        RS_HIP(hipEventRecord(s->arrived, s->commStream));
        if (timed && s->timing) RS_HIP(hipEventRecord(s->waitFrom, rs_stream()));
        RS_HIP(hipStreamWaitEvent(rs_stream(), s->arrived, 0));
        if (timed && s->timing) { RS_HIP(hipEventRecord(s->waitTo, rs_stream())); s->waitValid = true; }
    }
    return 0;
}

}  // namespace

extern "C" {

int rs_comm_destroy(rs_comm* c) {
    if (!c) return 0;
    if (c->lib) dlclose(c->lib);
    delete c;
    return 0;
}

int rs_comm_create(const rs_transport* t, int rank, int world, rs_comm** out) {
    if (!t || !out || world < 1 || rank < 0 || rank >= world || !t->send || !t->recv)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_comm_create: bad argument");
    rs_comm* c = new rs_comm();
    c->t = *t; c->rank = rank; c->world = world;
    *out = c;
    return 0;
}

// The communicator belongs to ONE copy of RCCL -- the one whose ncclCommInitRank made it -- and its handle means nothing to another
// copy (a process can hold two: PyTorch wheels bundle their own librccl.so next to /opt/rocm/lib's).  Binding order: the path the
// caller names; else what the process has linked or loaded globally (dlsym in the global scope); else a copy that is already
// loaded under RCCL's soname (RTLD_NOLOAD: nothing new is mapped); only if the process has no RCCL at all a fresh, local open.
int rs_comm_create_rccl_lib(void* ncclComm, int rank, int world, const char* librcclPath, rs_comm** out) {
    if (!ncclComm || !out || world < 1 || rank < 0 || rank >= world) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_comm_create_rccl: bad argument");
    rs_comm* c = new rs_comm();
    c->rank = rank; c->world = world; c->nccl = ncclComm;
    void* from = nullptr;                                        // handle to look the symbols up in; RTLD_DEFAULT when they are global
    if (librcclPath && librcclPath[0]) {
        c->lib = dlopen(librcclPath, RTLD_NOW | RTLD_LOCAL);
        if (!c->lib) { delete c; return rs_fail(RS_ERR_UNSUPPORTED, "rs_comm_create_rccl_lib: the named librccl cannot be opened"); }
        from = c->lib;
    }
    else if (dlsym(RTLD_DEFAULT, "ncclSend")) from = RTLD_DEFAULT;
    else {
        for (const char* name : { "librccl.so.1", "librccl.so" }) { c->lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (c->lib) break; }
        if (!c->lib) for (const char* name : { "librccl.so.1", "librccl.so" }) { c->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (c->lib) break; }
        if (!c->lib) { delete c; return rs_fail(RS_ERR_UNSUPPORTED, "rs_comm_create_rccl: librccl.so cannot be opened"); }
        from = c->lib;
    }
    c->pSend = (decltype(c->pSend))dlsym(from, "ncclSend");
    c->pRecv = (decltype(c->pRecv))dlsym(from, "ncclRecv");
    c->pGroupStart = (decltype(c->pGroupStart))dlsym(from, "ncclGroupStart");
    c->pGroupEnd = (decltype(c->pGroupEnd))dlsym(from, "ncclGroupEnd");
    c->pErr = (decltype(c->pErr))dlsym(from, "ncclGetErrorString");
    if (!c->pSend || !c->pRecv || !c->pGroupStart || !c->pGroupEnd) { rs_comm_destroy(c); return rs_fail(RS_ERR_UNSUPPORTED, "rs_comm_create_rccl: librccl.so lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd"); }
    c->t.ctx = c; c->t.group_begin = rccl_group_begin; c->t.group_end = rccl_group_end; c->t.send = rccl_send; c->t.recv = rccl_recv;
    c->t.stream_ordered = 1;
    *out = c;
    return 0;
}
int rs_comm_create_rccl(void* ncclComm, int rank, int world, rs_comm** out) { return rs_comm_create_rccl_lib(ncclComm, rank, world, nullptr, out); }

// Sends `bytes` of devSend to this rank itself and receives them into devRecv, through the transport exactly as rs_strips_frame
// uses it (group, stream order, events): a one-rank check of a transport, e.g. of the run-time binding to librccl on a single GPU.
int rs_comm_self_exchange(rs_comm* c, const void* devSend, void* devRecv, size_t bytes) {
    if (!c || !devSend || !devRecv) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_comm_self_exchange: null argument");
    hipStream_t ts = nullptr;
    RS_HIP(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
    hipEvent_t before = nullptr, after = nullptr;
    RS_HIP(hipEventCreateWithFlags(&before, hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&after, hipEventDisableTiming));
    int e = 0;
    if (c->t.stream_ordered) { e = rs_check_hip(hipEventRecord(before, rs_stream()), "event"); if (!e) e = rs_check_hip(hipStreamWaitEvent(ts, before, 0), "wait"); }
    else e = rs_synchronize();
    if (!e && c->t.group_begin) e = c->t.group_begin(c->t.ctx);
    if (!e) e = c->t.send(c->t.ctx, devSend, bytes, c->rank, ts);
    if (!e) e = c->t.recv(c->t.ctx, devRecv, bytes, c->rank, ts);
    if (!e && c->t.group_end) e = c->t.group_end(c->t.ctx);
    if (!e && c->t.stream_ordered) { e = rs_check_hip(hipEventRecord(after, ts), "event"); if (!e) e = rs_check_hip(hipStreamWaitEvent(rs_stream(), after, 0), "wait"); }
    if (!e) e = rs_check_hip(hipStreamSynchronize(ts), "rs_comm_self_exchange");
    if (!e) e = rs_check_hip(hipStreamSynchronize(rs_stream()), "rs_comm_self_exchange");
    (void)hipEventDestroy(before); (void)hipEventDestroy(after); (void)hipStreamDestroy(ts);
    return e;
}

int rs_strips_destroy(rs_strips* s) {
    RS_SCOPE(s);
    if (!s) return 0;
    (void)rs_synchronize();
    if (!s->deferred[0].empty()) (void)post(s, nullptr, 0);   // (every rank reaches this with the same deferred gathers)
    if (!s->deferred[1].empty()) { rs_denoise_scope onDenoiseStream(false); (void)post(s, nullptr, 0, 1); }      // (on that stream if it still exists, else on the library stream: everything has finished)
    (void)rs_synchronize();
    if (s->commStream) { (void)hipStreamSynchronize(s->commStream); (void)hipStreamDestroy(s->commStream); }
    if (s->ownStreamCounted) { rs_ctx()->ownCommStreams--; s->ownStreamCounted = false; }      // (the transfer stream is gone: its chain is free again)
    if (s->packed) (void)hipEventDestroy(s->packed);
    if (s->arrived) (void)hipEventDestroy(s->arrived);
    for (hipEvent_t& e : s->gathered) if (e) (void)hipEventDestroy(e);
    if (s->waitFrom) (void)hipEventDestroy(s->waitFrom);
    if (s->waitTo) (void)hipEventDestroy(s->waitTo);
    rs_dev_free(s->sendUp); rs_dev_free(s->recvUp); rs_dev_free(s->sendDown); rs_dev_free(s->recvDown);
    for (int i = 0; i < 2; i++) { rs_dev_free(s->eawBuf[i]); rs_dev_free(s->eawSend[i]); rs_dev_free(s->eawRecv[i]); }
    rs_dev_free(s->histSend); rs_dev_free(s->histRecv);
    delete s;
    return 0;
}

int rs_strips_create(rs_comm* comm, int width, int height, const int* bounds, rs_strips** out) {
    if (!comm || !out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_create: bad argument");
    *out = nullptr;
    rs_strips* s = new rs_strips();
    s->ctx = rs_ctx();
    rs_ctx_scope scope(s->ctx);
    s->comm = comm; s->width = width; s->height = height;
    s->bounds.resize((size_t)comm->world + 1);
    for (int r = 0; r <= comm->world; r++) {
        if (bounds) s->bounds[(size_t)r] = bounds[r];
        else {                                                   // heights that differ by at most one row (tiling.strip_bounds)
            const int base = height / comm->world, rem = height % comm->world;
            s->bounds[(size_t)r] = r * base + (r < rem ? r : rem);
        }
    }
    bool ok = s->bounds[0] == 0 && s->bounds[(size_t)comm->world] == height;
    for (int r = 0; r < comm->world && ok; r++) ok = s->bounds[(size_t)r + 1] - s->bounds[(size_t)r] >= (comm->world > 1 ? kHalo : 1);
    if (!ok) { delete s; return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_create: the bounds must tile [0, height) in rank order with strips of at least 5 rows"); }
    s->y0 = s->bounds[(size_t)comm->rank]; s->y1 = s->bounds[(size_t)comm->rank + 1];
    s->haloBytes = (size_t)width * kHalo * (48u + 20u);          // rs_restir_halo_bytes + rs_gbuffer_rows_bytes
    int e = 0;
    if (comm->rank > 0) { e = rs_dev_alloc(&s->sendUp, s->haloBytes); if (!e) e = rs_dev_alloc(&s->recvUp, s->haloBytes); }
    if (!e && comm->rank + 1 < comm->world) { e = rs_dev_alloc(&s->sendDown, s->haloBytes); if (!e) e = rs_dev_alloc(&s->recvDown, s->haloBytes); }
    if (!e && comm->world > 1) {
        if (!e) e = rs_check_hip(hipEventCreateWithFlags(&s->packed, hipEventDisableTiming), "hipEventCreate");
        if (!e) e = rs_check_hip(hipEventCreateWithFlags(&s->arrived, hipEventDisableTiming), "hipEventCreate");
        for (hipEvent_t& ev : s->gathered) if (!e) e = rs_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        if (!e) e = rs_check_hip(hipEventCreate(&s->waitFrom), "hipEventCreate");
        if (!e) e = rs_check_hip(hipEventCreate(&s->waitTo), "hipEventCreate");
    }
    if (e) { rs_strips_destroy(s); return e; }
    *out = s;
    return 0;
}

// Where the transfers of a stream-ordered transport are enqueued: 0 (default) on the library stream itself, 1 on a stream of the
// driver ordered against the library stream by events (the interior rows of phase B then run while the border rows travel, at the
// price of a fifth stream).  Between frames only: no gather may be in flight.
int rs_strips_set_comm_stream(rs_strips* s, int ownStream) {
    RS_SCOPE(s);
    if (!s) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_set_comm_stream: null");
    for (bool pending : s->gatherPending) if (pending) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_set_comm_stream: a gather is in flight");
    RS_TRY(rs_synchronize());
    if (s->commStream) RS_HIP(hipStreamSynchronize(s->commStream));
    // (In the caller's priority pool.  Whichever pool it is in, this is a FIFTH stream with work in flight next to the library stream and
    // three chains, and the device runs four queues at a time (a 1/8 strip 0.166 -> 0.48 ms per frame with a transport that moves
    // nothing, exactly like a fourth chain -- profiles/r05_ab_four_chains.log), so with it the frames keep two chains in flight: 0.28 ms.
    // The form exists for a machine where RCCL's kernel on the library stream costs more than that difference; bench.py times both and
    // keeps this one only if it wins by 3 %.)
    if (ownStream && !s->commStream && s->comm->world > 1) {
        int prio = 0;
        if (rs_stream() && hipStreamGetPriority(rs_stream(), &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
        RS_HIP(hipStreamCreateWithPriority(&s->commStream, hipStreamNonBlocking, prio));
    }
    s->commOnMain = !(ownStream && s->commStream);
    // the transfer stream is a stream with work in flight next to the library stream: with it the frames keep one chain less in flight
    // (five streams that hand events to each other: 0.17 -> 0.48 ms per frame on a 1/8 strip; with two chains 0.28).  Counted per driver, so that a second
    // driver of the context (bench.py's parity check makes one) neither takes the chain back nor gives it away (rs_chains_in_flight).
    if (!s->commOnMain && !s->ownStreamCounted) { rs_ctx()->ownCommStreams++; s->ownStreamCounted = true; }
    if (s->commOnMain && s->ownStreamCounted) { rs_ctx()->ownCommStreams--; s->ownStreamCounted = false; }
    return 0;
}

// How many rows of the G-buffer id / normal / depth planes travel to the neighbours with the reservoir rows of rs_strips_frame: 5 (default) is what
// the spatial taps compare against; 32 makes the rows a denoiser's taps reach arrive in the same group, and rs_strips_eaw_filter /
// rs_strips_svgf_filter then skip their own exchange of them.  Every rank must set the same value (it is the size of the messages); strips of at
// least `rows` rows; between frames.
int rs_strips_set_gbuffer_halo(rs_strips* s, int rows) {
    RS_SCOPE(s);
    if (!s || rows < kHalo || rows > 64) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_set_gbuffer_halo: 5 to 64 rows");
    const rs_comm* c = s->comm;
    if (c->world > 1)
        for (int r = 0; r < c->world; r++)
            if (s->bounds[(size_t)r + 1] - s->bounds[(size_t)r] < rows) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_set_gbuffer_halo: a strip is shorter than the rows asked for");
    if (rows == s->gReach) return 0;
    RS_TRY(rs_synchronize());
    s->gReach = rows; s->gbufFresh = false;
    s->haloBytes = (size_t)s->width * ((size_t)kHalo * 48u + (size_t)rows * 20u);
    rs_dev_free(s->sendUp); rs_dev_free(s->recvUp); rs_dev_free(s->sendDown); rs_dev_free(s->recvDown);
    if (c->rank > 0) { RS_TRY(rs_dev_alloc(&s->sendUp, s->haloBytes)); RS_TRY(rs_dev_alloc(&s->recvUp, s->haloBytes)); }
    if (c->rank + 1 < c->world) { RS_TRY(rs_dev_alloc(&s->sendDown, s->haloBytes)); RS_TRY(rs_dev_alloc(&s->recvDown, s->haloBytes)); }
    return 0;
}

int rs_strips_rows(const rs_strips* s, int* y0, int* y1) {
    RS_SCOPE(s);
    if (!s || !y0 || !y1) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_rows: null argument");
    *y0 = s->y0; *y1 = s->y1;
    return 0;
}

// One runCuda frame of this rank: GBuffer::render, ReSTIRDirect, (the caller tone-maps / gathers its rows), GBuffer::update is
// the caller's, as in the reference.  Radiance rows [y0, y1) of devDirectIllum are valid afterwards.
int rs_strips_frame(rs_strips* s, rs_restir* r, const rs_scene* scene, const rs_camera* cam, rs_gbuffer* g,
                    float* devDirectIllum, int iter, int looper, int reuse) {
    RS_SCOPE(s);
    if (!s || !r || !scene || !cam || !g || !devDirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_frame: null argument");
    if (g->width != s->width || g->height != s->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_frame: G-buffer size differs from the strips' frame");
    const int y0 = s->y0, y1 = s->y1;
    const rs_comm* c = s->comm;
    s->gbufFresh = false;
    RS_TRY(rs_gbuffer_render_rows(g, scene, cam, y0, y1));
    RS_TRY(rs_restir_phase_a(r, scene, cam, g, looper, reuse, y0, y1));
    const bool up = c->rank > 0, down = c->rank + 1 < c->world;
    if (!(reuse & 2) || (!up && !down)) {
        RS_TRY(rs_restir_phase_b(r, scene, cam, g, devDirectIllum, iter, reuse, y0, y1));
        return rs_restir_end_frame(r);
    }
    // pack the border rows (library stream): one launch for the six planes of both edges
    RS_TRY(rs_gbuffer_join(g));
    {
        SegList l;
        if (up) halo_segments(l, r, g, y0, y0, s->gReach, s->sendUp);
        if (down) halo_segments(l, r, g, y1 - kHalo, y1 - s->gReach, s->gReach, s->sendDown);
        RS_TRY(copy_segments(l, true));
    }
    // the transfers: after the packing copies
    Xfer ops[4]; size_t n = 0;
    if (up) { ops[n++] = { true, s->sendUp, s->haloBytes, c->rank - 1 }; ops[n++] = { false, s->recvUp, s->haloBytes, c->rank - 1 }; }
    if (down) { ops[n++] = { true, s->sendDown, s->haloBytes, c->rank + 1 }; ops[n++] = { false, s->recvDown, s->haloBytes, c->rank + 1 }; }
    const bool timeMain = s->commOnMain && c->t.stream_ordered && s->timing;
    if (timeMain) RS_HIP(hipEventRecord(s->waitFrom, rs_stream()));
    RS_TRY(post(s, ops, n));
    if (timeMain) { RS_HIP(hipEventRecord(s->waitTo, rs_stream())); s->waitValid = true; }
    if (s->commOnMain && c->t.stream_ordered) {
        // The transfers are in the library stream's own order: nothing runs next to them there, so splitting phase B into interior rows
        // and two 5-row bands would only buy two more launches -- and the bands are 5 rows in 16-row tiles.  Unpack, then ONE launch over
        // the strip (a 1/8 strip of 1080p: three launches of 14.6 + 8.5 + 8.8 us -> one of 15; results do not depend on the partition).
        SegList l;
        if (up) halo_segments(l, r, g, y0 - kHalo, y0 - s->gReach, s->gReach, s->recvUp);
        if (down) halo_segments(l, r, g, y1, y1, s->gReach, s->recvDown);
        s->gbufFresh = true;
        RS_TRY(copy_segments(l, false));
        RS_TRY(rs_restir_phase_b(r, scene, cam, g, devDirectIllum, iter, reuse, y0, y1));
        return rs_restir_end_frame(r);
    }
    // interior rows (their taps stay inside the strip) while the border rows travel
    const int topEnd = up ? (y0 + kHalo < y1 ? y0 + kHalo : y1) : y0;
    const int botStart = down ? (y1 - kHalo > topEnd ? y1 - kHalo : topEnd) : y1;
    if (botStart > topEnd) RS_TRY(rs_restir_phase_b(r, scene, cam, g, devDirectIllum, iter, reuse, topEnd, botStart));
    RS_TRY(join(s, true));
    {
        SegList l;
        if (up) halo_segments(l, r, g, y0 - kHalo, y0 - s->gReach, s->gReach, s->recvUp);
        if (down) halo_segments(l, r, g, y1, y1, s->gReach, s->recvDown);
        s->gbufFresh = true;
        RS_TRY(copy_segments(l, false));
    }
    if (topEnd > y0) RS_TRY(rs_restir_phase_b(r, scene, cam, g, devDirectIllum, iter, reuse, y0, topEnd));
    if (y1 > botStart) RS_TRY(rs_restir_phase_b(r, scene, cam, g, devDirectIllum, iter, reuse, botStart, y1));
    return rs_restir_end_frame(r);
}

// LeveledEAWFilter::filter (src/denoiser.cu:453-477) on this strip's rows of devColor.  The taps of level l reach 2 << l rows
// beyond the strip: the G-buffer rows they compare against (32 at most) come from the neighbouring strips once, and before each
// level the strips swap the 2 << l border rows of that level's input -- the values a full-frame filter reads there, so rows
// [y0, y1) of *devResult (a buffer of the driver) equal the full-frame filter's bit for bit.  Call between rs_strips_frame and
// rs_gbuffer_update.  The rows of devColor and of the current G-buffer planes just outside the strip are overwritten.
int rs_strips_eaw_filter(rs_strips* s, rs_eaw* f, rs_gbuffer* g, const rs_camera* cam, float* devColor, float** devResult) {
    RS_SCOPE(s);
    if (!s || !f || !g || !cam || !devColor || !devResult) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_eaw_filter: null argument");
    if (g->width != s->width || g->height != s->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_eaw_filter: G-buffer size differs from the strips' frame");
    const rs_comm* c = s->comm;
    const int W = s->width, y0 = s->y0, y1 = s->y1, kLevels = 5, reach = 2 << (kLevels - 1);
    const bool up = c->rank > 0, down = c->rank + 1 < c->world;
    if (c->world > 1)
        for (int r = 0; r < c->world; r++)
            if (s->bounds[(size_t)r + 1] - s->bounds[(size_t)r] < reach) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_eaw_filter: strips must be at least 32 rows tall");
    const size_t image = (size_t)W * s->height * 3 * sizeof(float);
    // rs_set_denoise_stream(1): everything below -- the exchanges included -- goes to the denoise stream, ordered after this frame's phase B
    // by an event; the library stream goes on with the next frame.  The transfers of that stream's groups are enqueued on it (or on the
    // driver's transfer stream, ordered against it), and the display gather of a tone map that ran there rides in its next group.
    RS_TRY(rs_gbuffer_join(g));                                  // (the library stream's own join, before the stream changes)
    rs_denoise_scope onDenoiseStream(true);
    RS_TRY(onDenoiseStream.err);
    RS_TRY(rs_denoise_order(devColor)); RS_TRY(rs_denoise_order(s->eawBuf[0])); RS_TRY(rs_denoise_order(s->eawBuf[1]));      // (no-ops on the denoise stream)
    for (int i = 0; i < 2; i++)
        if (!s->eawBuf[i]) { RS_TRY(rs_dev_alloc(&s->eawBuf[i], image / sizeof(float))); RS_HIP(hipMemsetAsync(s->eawBuf[i], 0, image, rs_stream())); }
    if (!(s->gbufFresh && s->gReach >= reach)) RS_TRY(exchange_gbuffer_rows(s, g, reach));      // (rs_strips_set_gbuffer_halo(s, 32): they came with the reservoir rows)
    s->gbufFresh = false;
    RS_TRY(rs_eaw_positions_rows(f, g, cam, y0 - reach > 0 ? y0 - reach : 0, y1 + reach < s->height ? y1 + reach : s->height));
    for (int level = 0; level < kLevels; level++) {
        float* in = level == 0 ? devColor : s->eawBuf[(level - 1) % 2];
        float* out = s->eawBuf[level % 2];
        if (up || down) {
            const int rows = 2 << level;
            const size_t bytes = (size_t)rows * W * 3 * sizeof(float);
            Xfer ops[4]; size_t n = 0;
            if (up) { ops[n++] = { true, in + (size_t)y0 * W * 3, bytes, c->rank - 1 }; ops[n++] = { false, in + (size_t)(y0 - rows) * W * 3, bytes, c->rank - 1 }; }
            if (down) { ops[n++] = { true, in + (size_t)(y1 - rows) * W * 3, bytes, c->rank + 1 }; ops[n++] = { false, in + (size_t)y1 * W * 3, bytes, c->rank + 1 }; }
            RS_TRY(post(s, ops, n));
            RS_TRY(join(s, false));
        }
        RS_TRY(rs_eaw_level_rows(f, out, in, g, level, y0, y1));
        // (the image: read by level 0, its rows just outside the strip written by that level's exchange; the next frame's phase B writes it)
        if (level == 0 && onDenoiseStream.active) RS_TRY(rs_denoise_mark(devColor, image, false));
    }
    if (onDenoiseStream.active) {
        RS_TRY(rs_denoise_mark(s->eawBuf[0], image, false)); RS_TRY(rs_denoise_mark(s->eawBuf[1], image, false));
        RS_TRY(rs_gbuffer_denoise_mark(g));
    }
    *devResult = s->eawBuf[(kLevels - 1) % 2];
    return 0;
}

// SpatioTemporalFilter::filter (src/denoiser.cu:532-564) on this strip's rows.  What a strip reads beyond its rows travels from the
// neighbouring strips: the 32 G-buffer rows the taps compare against (once), one row of the accumulated moments (the 3x3 variance
// estimate), and before every level the 2 * step + 1 border rows of that level's input colour and of the variance -- the values a
// full-frame filter reads there, so rows [y0, y1) of the result equal the full-frame rs_svgf_filter's bit for bit.  *devColorOut is
// handed over exactly as by rs_svgf_filter (every rank performs the same buffer swaps).  Strips of at least 33 rows; call between
// rs_strips_frame and rs_gbuffer_update, then rs_svgf_next_frame; with a moving camera rs_strips_exchange_svgf_history in between.
int rs_strips_svgf_filter(rs_strips* s, rs_svgf* f, rs_gbuffer* g, const rs_camera* cam, const float* devColorIn, float** devColorOut) {
    RS_SCOPE(s);
    if (!s || !f || !g || !cam || !devColorIn || !devColorOut || !*devColorOut) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_svgf_filter: null argument");
    if (g->width != s->width || g->height != s->height || f->width != s->width || f->height != s->height || cam->resolution[0] != s->width || cam->resolution[1] != s->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_svgf_filter: size mismatch");
    const rs_comm* c = s->comm;
    RS_TRY(rs_gbuffer_join(g));
    if (c->world == 1) return rs_svgf_filter_rows(f, devColorOut, devColorIn, g, cam, 0, s->height, nullptr);
    const int reach = 2 << 4;
    for (int r = 0; r < c->world; r++)
        if (s->bounds[(size_t)r + 1] - s->bounds[(size_t)r] < reach + 1) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_svgf_filter: strips must be at least 33 rows tall");
    if (!(s->gbufFresh && s->gReach >= reach)) RS_TRY(exchange_gbuffer_rows(s, g, reach));
    s->gbufFresh = false;
    const rs_svgf_row_hooks hooks{ s, svgf_exchange_hook };
    return rs_svgf_filter_rows(f, devColorOut, devColorIn, g, cam, s->y0, s->y1, &hooks);
}

// Moving camera: the temporal accumulation of the next frame reads the filter's history -- accumulated colour and moments of this
// frame -- at the reprojected pixel (src/denoiser.cu:250-305), which may lie in another strip: every rank's rows of both planes travel
// to every other rank, from and into place.  Call after rs_strips_svgf_filter and before rs_svgf_next_frame.
int rs_strips_exchange_svgf_history(rs_strips* s, rs_svgf* f) {
    RS_SCOPE(s);
    if (!s || !f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_exchange_svgf_history: null argument");
    if (f->width != s->width || f->height != s->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_exchange_svgf_history: size mismatch");
    const rs_comm* c = s->comm;
    if (c->world == 1) return 0;
    const size_t row = (size_t)s->width * 3;
    float* planes[2] = { f->devAccumColor[f->frameIdx], f->devAccumMoment[f->frameIdx] };
    std::vector<Xfer> ops;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        for (float* p : planes) {
            ops.push_back({ true, p + (size_t)s->y0 * row, (size_t)(s->y1 - s->y0) * row * sizeof(float), k });
            ops.push_back({ false, p + (size_t)s->bounds[(size_t)k] * row, (size_t)(s->bounds[(size_t)k + 1] - s->bounds[(size_t)k]) * row * sizeof(float), k });
        }
    }
    RS_TRY(post(s, ops.data(), ops.size()));
    return join(s, false);
}

// Moving camera: the temporal merge of the next frame reads last-frame reservoirs and G-buffer planes at the reprojected pixel
// (restir.cu:20-45), which may lie in another strip.  Every rank sends the rows it produced -- the reservoirs the next merge reads
// (which = 1 after rs_restir_end_frame) and the "last" G-buffer planes (after rs_gbuffer_update) -- to every other rank.  Call
// after rs_gbuffer_update.  A static camera reprojects into its own pixel and does not need this.
int rs_strips_exchange_history(rs_strips* s, rs_restir* r, rs_gbuffer* g) {
    RS_SCOPE(s);
    if (!s || !r || !g) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_exchange_history: null argument");
    if (g->width != s->width || g->height != s->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_exchange_history: G-buffer size differs from the strips' frame");
    const rs_comm* c = s->comm;
    if (c->world == 1) return 0;
    RS_TRY(rs_denoise_join());                                   // a filter on the denoise stream writes rows of these planes just outside the strip
    auto resvBytes = [&](int rank) { return rs_restir_rows_bytes(r, 1, s->bounds[(size_t)rank + 1] - s->bounds[(size_t)rank]); };
    auto bytesOf = [&](int rank) { return resvBytes(rank) + rs_gbuffer_rows_bytes(g, s->bounds[(size_t)rank + 1] - s->bounds[(size_t)rank]); };
    size_t others = 0;
    for (int k = 0; k < c->world; k++) if (k != c->rank) others += bytesOf(k);
    const size_t mine = bytesOf(c->rank);
    if (s->histSendBytes < mine) { rs_dev_free(s->histSend); s->histSend = nullptr; RS_TRY(rs_dev_alloc(&s->histSend, mine)); s->histSendBytes = mine; }
    if (s->histRecvBytes < others) { rs_dev_free(s->histRecv); s->histRecv = nullptr; RS_TRY(rs_dev_alloc(&s->histRecv, others)); s->histRecvBytes = others; }
    RS_TRY(rs_restir_rows_pack(r, 1, s->y0, s->y1 - s->y0, s->histSend));
    RS_TRY(rs_gbuffer_rows_pack(g, 1, s->y0, s->y1 - s->y0, s->histSend + resvBytes(c->rank)));
    std::vector<Xfer> ops;
    size_t off = 0;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        ops.push_back({ true, s->histSend, mine, k });
        ops.push_back({ false, s->histRecv + off, bytesOf(k), k });
        off += bytesOf(k);
    }
    RS_TRY(post(s, ops.data(), ops.size()));
    RS_TRY(join(s, false));
    off = 0;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        const int a = s->bounds[(size_t)k], rows = s->bounds[(size_t)k + 1] - a;
        RS_TRY(rs_restir_rows_unpack(r, 1, a, rows, s->histRecv + off));
        RS_TRY(rs_gbuffer_rows_unpack(g, 1, a, rows, s->histRecv + off + resvBytes(k)));
        off += bytesOf(k);
    }
    return 0;
}

// Image assembly: rows [y0, y1) of every rank's devImage (bytesPerPixel bytes per pixel, row-major, full-frame sized: the radiance
// image at 12, the display image at 4) travel into the same rows of devImage on `root`, or on every rank for root = -1.
int rs_strips_gather(rs_strips* s, void* devImage, size_t bytesPerPixel, int root) {
    RS_SCOPE(s);
    const rs_comm* c = s ? s->comm : nullptr;
    if (!s || !devImage || bytesPerPixel == 0 || root < -1 || root >= c->world) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_gather: bad argument");
    if (c->world == 1) return 0;
    const size_t row = (size_t)s->width * bytesPerPixel;
    char* base = (char*)devImage;
    RS_TRY(rs_denoise_order(base + (size_t)s->y0 * row));        // rows the denoise stream wrote: the library stream after it
    std::vector<Xfer> ops;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        if (root < 0 || root == k) ops.push_back({ true, base + (size_t)s->y0 * row, (size_t)(s->y1 - s->y0) * row, k });
        if (root < 0 || root == c->rank) ops.push_back({ false, base + (size_t)s->bounds[(size_t)k] * row, (size_t)(s->bounds[(size_t)k + 1] - s->bounds[(size_t)k]) * row, k });
    }
    RS_TRY(post(s, ops.data(), ops.size()));
    return join(s, false);
}

// The same assembly without making the library stream wait for it: _begin posts the transfers (after everything enqueued on the
// library stream so far) and returns; _end makes the library stream wait for them.  A caller with two display buffers begins the
// gather of frame f into one and ends it before frame f + 2 writes that buffer again (or before root reads it): the rows travel
// while the next frame's kernels run.  `slot` in [0, 4) names the gather; every rank must begin the same gathers in the same order.
int rs_strips_gather_begin(rs_strips* s, void* devImage, size_t bytesPerPixel, int root, int slot) {
    RS_SCOPE(s);
    const rs_comm* c = s ? s->comm : nullptr;
    if (!s || !devImage || bytesPerPixel == 0 || root < -1 || root >= c->world || slot < 0 || slot >= rs_strips::kGatherSlots)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_gather_begin: bad argument");
    if (c->world == 1) return 0;
    if (s->gatherPending[slot]) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_gather_begin: the slot's previous gather has not been ended");
    const size_t row = (size_t)s->width * bytesPerPixel;
    char* base = (char*)devImage;
    std::vector<Xfer> ops;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        if (root < 0 || root == k) ops.push_back({ true, base + (size_t)s->y0 * row, (size_t)(s->y1 - s->y0) * row, k });
        if (root < 0 || root == c->rank) ops.push_back({ false, base + (size_t)s->bounds[(size_t)k] * row, (size_t)(s->bounds[(size_t)k + 1] - s->bounds[(size_t)k]) * row, k });
    }
    // rows that the denoise stream wrote (the tone map of a filtered image, rs_set_denoise_stream): the gather is ordered on that stream
    const bool there = rs_denoise_owns(base + (size_t)s->y0 * row) && rs_denoise_stream() != nullptr;
    s->gatherOnDenoise[slot] = there;
    if (s->commOnMain && c->t.stream_ordered) {
        // in the stream's own order a gather is in order with everything else anyway: its transfers wait for the next group there (the next
        // frame's border rows; on the denoise stream the next frame's filter), or for rs_strips_gather_end if that comes first
        const int which = there ? 1 : 0;
        s->deferred[which].insert(s->deferred[which].end(), ops.begin(), ops.end());
        s->deferredSlots[which] |= 1 << slot;
        s->gatherPending[slot] = true;
        return 0;
    }
    {
        rs_denoise_scope onDenoiseStream(false, there);
        RS_TRY(onDenoiseStream.err);
        RS_TRY(post(s, ops.data(), ops.size()));
        if (there && root == c->rank) RS_TRY(rs_denoise_mark(base, (size_t)s->height * row, false));      // (root: the other ranks' rows arrive in that stream's order)
    }
    if (c->t.stream_ordered) RS_HIP(hipEventRecord(s->gathered[slot], s->commStream));
    s->gatherPending[slot] = true;
    return 0;
}
int rs_strips_gather_end(rs_strips* s, int slot) {
    RS_SCOPE(s);
    if (!s || slot < 0 || slot >= rs_strips::kGatherSlots) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_gather_end: bad argument");
    if (!s->gatherPending[slot]) return 0;
    s->gatherPending[slot] = false;
    if (s->commOnMain && s->comm->t.stream_ordered) {
        if (s->gatherOnDenoise[slot]) {
            if (s->deferredSlots[1] & (1 << slot)) {                            // nothing has carried them yet: a group of their own, on that stream
                rs_denoise_scope onDenoiseStream(false);         // (inactive if the mode was switched off since: rs_set_denoise_stream waited for everything, the library stream will do)
                RS_TRY(onDenoiseStream.err);
                RS_TRY(post(s, nullptr, 0, 1));
            }
            // (the next tone map into the buffer runs on the denoise stream, in order; the library stream waits for the group that carried the
            // rows -- the previous frame's, long finished in a running sequence)
            RS_HIP(hipStreamWaitEvent(rs_stream(), s->gathered[slot], 0));
            return 0;
        }
        if (s->deferredSlots[0] & (1 << slot)) RS_TRY(post(s, nullptr, 0));   // nothing has carried them yet: a group of their own
        return 0;                                                               // (in order on the library stream: nothing to wait for)
    }
    if (s->comm->t.stream_ordered) RS_HIP(hipStreamWaitEvent(rs_stream(), s->gathered[slot], 0));
    return 0;
}

// Measurement: with timing on, rs_strips_frame brackets the library stream's wait for the neighbours' border rows with two
// events; rs_strips_halo_wait_ms returns the span of the last frame that exchanged rows -- the part of the halo latency the
// interior rows of phase B did not hide (0 when the rows had arrived by the time the interior rows were done).  Waits for that frame.
int rs_strips_enable_timing(rs_strips* s, int enable) {
    RS_SCOPE(s);
    if (!s) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_enable_timing: null");
    s->timing = enable != 0 && s->waitFrom != nullptr;
    s->waitValid = false;
    return 0;
}
int rs_strips_halo_wait_ms(rs_strips* s, float* ms) {
    RS_SCOPE(s);
    if (!s || !ms) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_strips_halo_wait_ms: null argument");
    *ms = 0.f;
    if (!s->waitValid) return 0;
    RS_HIP(hipEventSynchronize(s->waitTo));
    RS_HIP(hipEventElapsedTime(ms, s->waitFrom, s->waitTo));
    return 0;
}

}  // extern "C"
