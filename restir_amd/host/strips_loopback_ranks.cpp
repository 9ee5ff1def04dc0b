// strips_loopback_ranks.cpp -- the strip driver's STREAM-ORDERED path with N ranks carrying real data on ONE GPU.
//
// RCCL refuses two ranks on one device, so on a one-GPU box the RCCL transport only ever runs with one rank, where rs_strips_frame
// returns before it posts anything; the gloo rehearsal transport is host-side (not stream-ordered), so it never enters the branch in
// which rs_strips_gather_begin DEFERS its transfers into the next frame's border-row group.  This check closes that gap: N host
// threads, each a rank with its own library context and its own non-blocking stream on GPU 0, over a transport that has RCCL's
// contract -- send / recv are only ENQUEUED on the stream they are given, grouped, matched in order per pair of ranks, the send
// buffer is free again in stream order -- implemented as device-to-device copies through a mailbox of staging buffers with events.
// The per-frame calls are bench.py's: rs_strips_frame, [rs_strips_eaw_filter,] rs_gbuffer_update, [rs_strips_exchange_history,]
// rs_strips_gather_end(k), tone map into display buffer k, rs_strips_gather_begin(k) -- two display buffers in flight -- plus the
// synchronous rs_strips_gather of the radiance (or filtered) image.  Rank 0 renders every frame once more as a full frame and compares
//   * the gathered radiance / filtered image after every frame, bit for bit,
//   * EVERY frame's asynchronously gathered RGBA8 display image (copied out right after its rs_strips_gather_end), byte for byte,
// for a static and an orbiting camera, with the transfers on the library stream (the default: deferred gathers ride in the next
// frame's group) and on the driver's own stream (rs_strips_set_comm_stream), without and with the EAW filter.
//
//     strips_loopback_ranks [N [SECONDS]]     N ranks (default 3, at most 8: the split BASELINE's multi-GPU configs name); the process ends itself after SECONDS (default 240)
#include <condition_variable>
#include <deque>

#include "strips_ranks_common.h"

namespace {
using namespace ranks;

constexpr int kMaxRanks = 8, kSlots = 8;
constexpr size_t kSlotBytes = 2u << 20;
constexpr int kFrames = 6;

struct Slot { char* buf = nullptr; size_t bytes = 0; hipEvent_t ready = nullptr, consumed = nullptr; bool posted = false, used = false; };
struct Channel { Slot slots[kSlots]; int head = 0, tail = 0; };           // messages src -> dst, matched first in, first out
struct Mailbox {
    std::mutex m;
    std::condition_variable cv;
    Channel ch[kMaxRanks][kMaxRanks];
};
struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };
struct Endpoint { Mailbox* box; int rank; bool inGroup = false; std::vector<Op> ops; };

// enqueue on `st`: wait until the slot's last reader is done, copy the payload into it, mark it ready
int do_send(Endpoint* e, const Op& op) {
    if (op.bytes > kSlotBytes) { std::fprintf(stderr, "loopback: message of %zu bytes exceeds the staging slots\n", op.bytes); return 1; }
    Channel& c = e->box->ch[e->rank][op.peer];
    std::unique_lock<std::mutex> lock(e->box->m);
    Slot& s = c.slots[c.tail % kSlots];
    if (!e->box->cv.wait_for(lock, std::chrono::seconds(60), [&] { return !s.posted; })) { std::fprintf(stderr, "loopback: rank %d -> %d: no free slot (receiver stuck)\n", e->rank, op.peer); return 1; }
    if (s.used && hipStreamWaitEvent(op.stream, s.consumed, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(s.buf, op.buf, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return 1;
    if (hipEventRecord(s.ready, op.stream) != hipSuccess) return 1;
    s.bytes = op.bytes; s.posted = true; s.used = true;
    c.tail++;
    e->box->cv.notify_all();
    return 0;
}
// enqueue on `st`: wait for the matching message to be ready, copy it out, mark the slot consumed
int do_recv(Endpoint* e, const Op& op) {
    Channel& c = e->box->ch[op.peer][e->rank];
    std::unique_lock<std::mutex> lock(e->box->m);
    Slot& s = c.slots[c.head % kSlots];
    if (!e->box->cv.wait_for(lock, std::chrono::seconds(60), [&] { return s.posted; })) { std::fprintf(stderr, "loopback: rank %d <- %d: nothing was sent\n", e->rank, op.peer); return 1; }
    if (s.bytes != op.bytes) { std::fprintf(stderr, "loopback: rank %d <- %d: %zu bytes expected, %zu sent (send / recv pairing broken)\n", e->rank, op.peer, op.bytes, s.bytes); return 1; }
    if (hipStreamWaitEvent(op.stream, s.ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(op.buf, s.buf, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return 1;
    if (hipEventRecord(s.consumed, op.stream) != hipSuccess) return 1;
    s.posted = false;
    c.head++;
    e->box->cv.notify_all();
    return 0;
}
int run_group(Endpoint* e) {           // as ncclGroupEnd: all sends are issued before any receive waits, so no order of calls can deadlock
    int err = 0;
    for (const Op& op : e->ops) if (op.send && !err) err = do_send(e, op);
    for (const Op& op : e->ops) if (!op.send && !err) err = do_recv(e, op);
    e->ops.clear();
    return err;
}
int lb_group_begin(void* ctx) { ((Endpoint*)ctx)->inGroup = true; return 0; }
int lb_group_end(void* ctx) { Endpoint* e = (Endpoint*)ctx; e->inGroup = false; return run_group(e); }
int lb_send(void* ctx, const void* buf, size_t bytes, int peer, void* stream) {
    Endpoint* e = (Endpoint*)ctx;
    e->ops.push_back({ true, const_cast<void*>(buf), bytes, peer, (hipStream_t)stream });
    return e->inGroup ? 0 : run_group(e);
}
int lb_recv(void* ctx, void* buf, size_t bytes, int peer, void* stream) {
    Endpoint* e = (Endpoint*)ctx;
    e->ops.push_back({ false, buf, bytes, peer, (hipStream_t)stream });
    return e->inGroup ? 0 : run_group(e);
}

std::atomic<int> mismatches{ 0 };
#define CHECK(x) RANKS_CHECK(x)

void run_rank(int rank, int world, Mailbox* box) {
    RANKS_HIP(hipSetDevice(0));
    rs_context* ctx = nullptr;
    CHECK(rs_context_create(0, &ctx));
    CHECK(rs_context_set_current(ctx));
    hipStream_t lib = nullptr;
    RANKS_HIP(hipStreamCreateWithFlags(&lib, hipStreamNonBlocking));
    CHECK(rs_set_stream(lib));
    CHECK(rs_set_sync(0));                                         // asynchronous launches, frames overlapped, as the benchmark runs them
    Endpoint ep{ box, rank };
    rs_transport t{};
    t.ctx = &ep; t.group_begin = lb_group_begin; t.group_end = lb_group_end; t.send = lb_send; t.recv = lb_recv; t.stream_ordered = 1;
    rs_comm* comm = nullptr;
    CHECK(rs_comm_create(&t, rank, world, &comm));
    rs_scene* scene = build_scene(rank);
    const size_t px = (size_t)W * H;

    for (int mode = 0; mode < 12; mode++) {
        // modes 8-11: the EAW filter, the tone map of its result and the display gather on the library's denoise stream
        // (rs_set_denoise_stream(1)) -- frames are enqueued without the library stream ever waiting for that stream, so the display images
        // also check that no later frame overwrites what the filter of an earlier one still reads
        const bool orbit = mode & 1, ownStream = mode & 2, denoiseStream = mode >= 8, denoise = (mode & 4) || denoiseStream;
        CHECK(rs_set_denoise_stream(denoiseStream ? 1 : 0));
        rs_strips* strips = nullptr;
        CHECK(rs_strips_create(comm, W, H, nullptr, &strips));
        CHECK(rs_strips_set_comm_stream(strips, ownStream ? 1 : 0));
        if (denoise && orbit) CHECK(rs_strips_set_gbuffer_halo(strips, 32));      // (half of the filter modes: its 32 G-buffer rows travel with the reservoir rows)
        int y0 = 0, y1 = 0;
        CHECK(rs_strips_rows(strips, &y0, &y1));
        const int sets = rank == 0 ? 2 : 1;                        // rank 0: the strips' objects and a full-frame renderer of its own
        rs_gbuffer* g[2] = {}; rs_restir* r[2] = {}; rs_eaw* eaw[2] = {}; float* img[2] = {}; float* filtOut = nullptr;
        for (int k = 0; k < sets; k++) {
            CHECK(rs_gbuffer_create(W, H, &g[k])); CHECK(rs_restir_init(W, H, &r[k]));
            if (denoise) CHECK(rs_eaw_create(W, H, 5, &eaw[k]));
            RANKS_HIP(hipMalloc((void**)&img[k], px * 12)); RANKS_HIP(hipMemset(img[k], 0, px * 12));
        }
        float* gathered = nullptr; unsigned char* pbo[2] = {}; unsigned char* shownCopy = nullptr; unsigned char* refPbo = nullptr;
        RANKS_HIP(hipMalloc((void**)&gathered, px * 12)); RANKS_HIP(hipMemset(gathered, 0, px * 12));
        for (auto& p : pbo) { RANKS_HIP(hipMalloc((void**)&p, px * 4)); RANKS_HIP(hipMemset(p, 0, px * 4)); }
        RANKS_HIP(hipMalloc((void**)&shownCopy, px * 4 * kFrames));
        if (rank == 0) { RANKS_HIP(hipMalloc((void**)&refPbo, px * 4)); RANKS_HIP(hipMalloc((void**)&filtOut, px * 12)); }
        std::vector<std::vector<unsigned char>> refDisplay;        // rank 0: the full frame's display image of every frame
        bool same = true;
        float* refOut = filtOut;                                   // rs_eaw_filter swaps this with the filter's buffer, like the reference's vec3*&
        for (int frame = 0; frame < kFrames; frame++) {
            rs_camera cam;
            make_camera(cam, frame, orbit);
            CHECK(rs_camera_update(&cam));
            CHECK(rs_strips_frame(strips, r[0], scene, &cam, g[0], img[0], 0, frame, 3));
            float* shown = img[0];
            if (denoise) CHECK(rs_strips_eaw_filter(strips, eaw[0], g[0], &cam, img[0], &shown));
            CHECK(rs_gbuffer_update(g[0], &cam));
            if (orbit) CHECK(rs_strips_exchange_history(strips, r[0], g[0]));
            const int k = frame % 2;
            CHECK(rs_strips_gather_end(strips, k));                 // the gather that read display buffer k two frames ago
            if (frame >= 2) RANKS_HIP(hipMemcpyAsync(shownCopy + px * 4 * (frame - 2), pbo[k], px * 4, hipMemcpyDeviceToDevice, lib));
            CHECK(rs_copy_image_to_pbo(pbo[k] + (size_t)y0 * W * 4, shown + (size_t)y0 * W * 3, W, y1 - y0, 2, 1.f));
            // the radiance (or filtered) rows of every rank on rank 0, the synchronous form -- BEFORE the display gather is begun, so that
            // on the library stream that one stays deferred until the next frame's border-row group carries it (the merged group).
            // With the denoise stream only after the last frame: a caller's own copy of the filtered image needs rs_join_denoise_stream
            // first, and a join in every frame would hide the hazards the display images are there to catch.
            const bool checkRadiance = !denoiseStream || frame == kFrames - 1;
            if (checkRadiance) {
                if (denoiseStream) CHECK(rs_join_denoise_stream());
                RANKS_HIP(hipMemcpyAsync(gathered + (size_t)y0 * W * 3, shown + (size_t)y0 * W * 3, (size_t)(y1 - y0) * W * 12, hipMemcpyDeviceToDevice, lib));
                CHECK(rs_strips_gather(strips, gathered, 12, 0));
            }
            CHECK(rs_strips_gather_begin(strips, pbo[k], 4, 0, k));
            if (rank == 0) {
                CHECK(rs_gbuffer_render(g[1], scene, &cam));
                CHECK(rs_restir_direct(r[1], scene, &cam, g[1], img[1], 0, frame, 3));
                const float* ref = img[1];
                if (denoise) { CHECK(rs_eaw_filter(eaw[1], &refOut, img[1], g[1], &cam)); ref = refOut; }
                CHECK(rs_copy_image_to_pbo(refPbo, ref, W, H, 2, 1.f));
                CHECK(rs_gbuffer_update(g[1], &cam));
                CHECK(rs_synchronize());
                std::vector<float> x(3 * px), y(3 * px);
                RANKS_HIP(hipMemcpy(x.data(), gathered, px * 12, hipMemcpyDeviceToHost));
                RANKS_HIP(hipMemcpy(y.data(), ref, px * 12, hipMemcpyDeviceToHost));
                refDisplay.emplace_back(px * 4);
                RANKS_HIP(hipMemcpy(refDisplay.back().data(), refPbo, px * 4, hipMemcpyDeviceToHost));
                double sum = 0; for (float f : y) sum += f;
                if (checkRadiance && (std::memcmp(x.data(), y.data(), px * 12) != 0 || !(sum > 0))) {
                    size_t bad = 0; for (size_t i = 0; i < x.size(); i++) bad += std::memcmp(&x[i], &y[i], 4) != 0;
                    std::fprintf(stderr, "mode %d, frame %d: gathered strips differ from the full frame in %zu values (sum %g)\n", mode, frame, bad, sum);
                    same = false;
                }
            }
        }
        // the last two display images are still in flight
        for (int frame = kFrames; frame < kFrames + 2; frame++) {
            const int k = frame % 2;
            CHECK(rs_strips_gather_end(strips, k));
            RANKS_HIP(hipMemcpyAsync(shownCopy + px * 4 * (frame - 2), pbo[k], px * 4, hipMemcpyDeviceToDevice, lib));
        }
        CHECK(rs_synchronize());
        RANKS_HIP(hipStreamSynchronize(lib));
        if (rank == 0) {
            std::vector<unsigned char> got(px * 4);
            for (int frame = 0; frame < kFrames; frame++) {
                RANKS_HIP(hipMemcpy(got.data(), shownCopy + px * 4 * frame, px * 4, hipMemcpyDeviceToHost));
                if (std::memcmp(got.data(), refDisplay[(size_t)frame].data(), px * 4) != 0) {
                    size_t bad = 0; for (size_t i = 0; i < px; i++) bad += std::memcmp(&got[i * 4], &refDisplay[(size_t)frame][i * 4], 4) != 0;
                    std::fprintf(stderr, "mode %d, frame %d: asynchronously gathered display image differs from the full frame's in %zu pixels\n", mode, frame, bad);
                    same = false;
                }
            }
            std::printf("world %d, %s camera, transfers on %s%s: gathered strips and display images == full frame over %d frames: %s\n", world,
                        orbit ? "orbiting" : "static", ownStream ? "the driver's stream" : "the library stream (deferred gathers)",
                        denoiseStream ? ", EAW filter + tone map + display gather on the denoise stream" : denoise ? ", EAW filter" : "", kFrames, same ? "True" : "False");
            std::fflush(stdout);
            if (!same) mismatches++;
        }
        CHECK(rs_strips_destroy(strips));
        for (int k = 0; k < sets; k++) { rs_restir_free(r[k]); rs_gbuffer_destroy(g[k]); if (eaw[k]) rs_eaw_destroy(eaw[k]); (void)hipFree(img[k]); }
        (void)hipFree(gathered); (void)hipFree(pbo[0]); (void)hipFree(pbo[1]); (void)hipFree(shownCopy); (void)hipFree(refPbo); (void)hipFree(refOut);
    }
    rs_comm_destroy(comm);
    rs_scene_destroy(scene);
    (void)rs_set_stream(nullptr);
    (void)hipStreamDestroy(lib);
    (void)rs_context_set_current(nullptr);
    (void)rs_context_destroy(ctx);
}

}  // namespace

int main(int argc, char** argv) {
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) { std::fprintf(stderr, "no GPU\n"); return 1; }
    const int world = argc > 1 ? std::atoi(argv[1]) : 3;
    if (world < 1 || world > kMaxRanks) { std::fprintf(stderr, "1..%d ranks\n", kMaxRanks); return 2; }
    if (world > 6) H = 36 * world;                                 // eight ranks: 288 rows, 36 per strip (the EAW levels reach 32 rows into the neighbours)
    if (H / world < 32) { std::fprintf(stderr, "strips of fewer than 32 rows (the EAW levels reach that far)\n"); return 2; }
    start_watchdog(argc > 2 ? std::atoi(argv[2]) : 240);
    static Mailbox box;
    if (hipSetDevice(0) != hipSuccess) return 1;
    for (int a = 0; a < world; a++) for (int b = 0; b < world; b++) {
        if (a == b) continue;
        for (Slot& s : box.ch[a][b].slots) {
            if (hipMalloc((void**)&s.buf, kSlotBytes) != hipSuccess || hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) != hipSuccess) { std::fprintf(stderr, "mailbox allocation failed\n"); return 1; }
        }
    }
    std::vector<std::thread> threads;
    for (int k = 0; k < world; k++) threads.emplace_back(run_rank, k, world, &box);
    for (auto& t : threads) t.join();
    if (mismatches) { std::fprintf(stderr, "strips_loopback_ranks: %d mode(s) with mismatches\n", mismatches.load()); return 1; }
    std::printf("strips_loopback_ranks ok (%d ranks)\n", world);
    return 0;
}
