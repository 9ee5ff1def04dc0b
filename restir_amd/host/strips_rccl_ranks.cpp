// strips_rccl_ranks.cpp -- the strip driver over RCCL with N ranks in ONE process: N host threads, thread k on GPU k with its own
// library context (rs_context_create), one ncclComm_t per thread from one ncclUniqueId (ncclCommInitRank), and the calls of
// INTEGRATION.md section 2 per frame: rs_strips_frame, rs_gbuffer_update, rs_strips_gather.  Rank 0 also renders the full frame by
// itself and compares the gathered image with it bit for bit, static camera and then an orbiting one (rs_strips_exchange_history).
//
//     strips_rccl_ranks [N [SECONDS]]   N = number of ranks = number of GPUs used (default: every GPU of the node); the process ends
//                                       itself after SECONDS (default 240) -- and at once, communicators aborted, on the first error
//
// With N = 1 this runs on a one-GPU box (tests/test_gpu_parity.py runs it that way: threads, contexts, communicator, driver, comparison);
// N > 1 needs N GPUs -- RCCL refuses two ranks on one device -- and is the check to run once on a multi-GPU node.
#include <rccl/rccl.h>

#include "strips_ranks_common.h"

namespace {
using namespace ranks;

// every communicator created so far: the first failure aborts them (ncclCommAbort), so that no rank keeps spinning in an RCCL kernel or
// in ncclCommInitRank for a peer that has given up
std::mutex g_comm_mutex;
std::vector<ncclComm_t> g_comms;
void abort_comms() {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    for (ncclComm_t c : g_comms) (void)ncclCommAbort(c);
    g_comms.clear();
}
#define CHECK(x) RANKS_CHECK(x)
std::atomic<int> failures{ 0 };        // image mismatches (errors end the process at once)

void run_rank(int rank, int world, ncclUniqueId id) {
    RANKS_HIP(hipSetDevice(rank));
    rs_context* ctx = nullptr;
    CHECK(rs_context_create(rank, &ctx));
    CHECK(rs_context_set_current(ctx));
    CHECK(rs_set_sync(0));                                         // asynchronous launches, as the benchmark runs them
    ncclComm_t nccl = nullptr;
    { const ncclResult_t e = ncclCommInitRank(&nccl, world, id, rank); if (e != ncclSuccess) fail(rank, "ncclCommInitRank", (int)e); }
    { std::lock_guard<std::mutex> lock(g_comm_mutex); g_comms.push_back(nccl); }
    rs_comm* comm = nullptr;
    CHECK(rs_comm_create_rccl(nccl, rank, world, &comm));

    rs_scene* scene = build_scene(rank);

    for (int orbit = 0; orbit < 2; orbit++) {
        rs_strips* strips = nullptr;
        CHECK(rs_strips_create(comm, W, H, nullptr, &strips));
        rs_gbuffer* g[2] = { nullptr, nullptr }; rs_restir* r[2] = { nullptr, nullptr }; float* img[2] = { nullptr, nullptr };
        const int sets = rank == 0 ? 2 : 1;                        // rank 0: the strips' objects and a full-frame renderer of its own
        for (int k = 0; k < sets; k++) {
            CHECK(rs_gbuffer_create(W, H, &g[k])); CHECK(rs_restir_init(W, H, &r[k]));
            RANKS_HIP(hipMalloc((void**)&img[k], sizeof(float) * 3 * W * H));
            (void)hipMemset(img[k], 0, sizeof(float) * 3 * W * H);
        }
        bool same = true;
        for (int frame = 0; frame < FRAMES; frame++) {
            rs_camera cam;
            make_camera(cam, frame, orbit != 0);
            CHECK(rs_camera_update(&cam));
            CHECK(rs_strips_frame(strips, r[0], scene, &cam, g[0], img[0], 0, frame, 3));
            CHECK(rs_gbuffer_update(g[0], &cam));
            if (orbit) CHECK(rs_strips_exchange_history(strips, r[0], g[0]));
            CHECK(rs_strips_gather(strips, img[0], 12, 0));
            if (rank == 0) {
                CHECK(rs_gbuffer_render(g[1], scene, &cam));
                CHECK(rs_restir_direct(r[1], scene, &cam, g[1], img[1], 0, frame, 3));
                CHECK(rs_gbuffer_update(g[1], &cam));
                CHECK(rs_synchronize());
                std::vector<float> x(3 * W * H), y(3 * W * H);
                (void)hipMemcpy(x.data(), img[0], sizeof(float) * x.size(), hipMemcpyDeviceToHost);
                (void)hipMemcpy(y.data(), img[1], sizeof(float) * y.size(), hipMemcpyDeviceToHost);
                double sum = 0; for (float f : y) sum += f;
                if (std::memcmp(x.data(), y.data(), sizeof(float) * x.size()) != 0 || !(sum > 0)) {
                    size_t bad = 0; for (size_t i = 0; i < x.size(); i++) bad += std::memcmp(&x[i], &y[i], 4) != 0;
                    std::fprintf(stderr, "%s camera, frame %d: gathered strips differ from the full frame in %zu values (radiance sum %g)\n", orbit ? "orbiting" : "static", frame, bad, sum);
                    same = false;
                }
            }
        }
        CHECK(rs_synchronize());
        if (rank == 0) {
            std::printf("world %d, %s camera: gathered strips == full frame over %d frames: %s\n", world, orbit ? "orbiting" : "static", FRAMES, same ? "True" : "False");
            if (!same) failures++;
        }
        rs_strips_destroy(strips);
        for (int k = 0; k < sets; k++) { rs_restir_free(r[k]); rs_gbuffer_destroy(g[k]); (void)hipFree(img[k]); }
    }
    rs_comm_destroy(comm);
    rs_scene_destroy(scene);
    { std::lock_guard<std::mutex> lock(g_comm_mutex); for (auto& c : g_comms) if (c == nccl) { c = g_comms.back(); g_comms.pop_back(); break; } }
    ncclCommDestroy(nccl);
    (void)rs_context_set_current(nullptr);
    (void)rs_context_destroy(ctx);
}

}  // namespace

int main(int argc, char** argv) {
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) { std::fprintf(stderr, "no GPU\n"); return 1; }
    const int world = argc > 1 ? std::atoi(argv[1]) : devices;
    g_abort_hook = abort_comms;
    start_watchdog(argc > 2 ? std::atoi(argv[2]) : 240);
    if (world < 1 || world > devices) { std::fprintf(stderr, "%d ranks asked for, %d GPUs here (one rank per GPU)\n", world, devices); return 2; }
    if (H / world < 5) { std::fprintf(stderr, "strips of fewer than 5 rows\n"); return 2; }
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
    std::vector<std::thread> threads;
    for (int k = 0; k < world; k++) threads.emplace_back(run_rank, k, world, id);
    for (auto& t : threads) t.join();
    if (failures) { std::fprintf(stderr, "strips_rccl_ranks: %d failure(s)\n", failures.load()); return 1; }
    std::printf("strips_rccl_ranks ok (%d rank%s)\n", world, world == 1 ? "" : "s");
    return 0;
}
