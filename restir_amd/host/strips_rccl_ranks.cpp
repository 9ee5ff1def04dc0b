// strips_rccl_ranks.cpp -- the strip driver over RCCL with N ranks in ONE process: N host threads, thread k on GPU k with its own
// library context (rs_context_create), one ncclComm_t per thread from one ncclUniqueId (ncclCommInitRank), and the calls of
// INTEGRATION.md section 2 per frame: rs_strips_frame, rs_gbuffer_update, rs_strips_gather.  Rank 0 also renders the full frame by
// itself and compares the gathered image with it bit for bit, static camera and then an orbiting one (rs_strips_exchange_history).
//
//     strips_rccl_ranks [N]        N = number of ranks = number of GPUs used; default: every GPU of the node
//
// With N = 1 this runs on a one-GPU box (tests/test_gpu_parity.py runs it that way: threads, contexts, communicator, driver, comparison);
// N > 1 needs N GPUs -- RCCL refuses two ranks on one device -- and is the check to run once on a multi-GPU node.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/restir_hip.h"

namespace {

constexpr int W = 320, H = 200, FRAMES = 4;

// a floor, a back wall of 24 x 12 slanted facets (so that strips see different geometry) and four small lights facing down
void make_scene(std::vector<float>& v, std::vector<float>& n, std::vector<int>& matIds, std::vector<rs_material>& mats) {
    auto tri = [&](const float* a, const float* b, const float* c, int mat) {
        const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
        float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float l = std::sqrt(nx * nx + ny * ny + nz * nz); nx /= l; ny /= l; nz /= l;
        for (const float* p : { a, b, c }) { v.insert(v.end(), p, p + 3); n.push_back(nx); n.push_back(ny); n.push_back(nz); }
        matIds.push_back(mat);
    };
    const float f0[3] = { -4, 0, 1 }, f1[3] = { 4, 0, 1 }, f2[3] = { 4, 0, -6 }, f3[3] = { -4, 0, -6 };
    tri(f0, f1, f2, 0); tri(f0, f2, f3, 0);
    for (int j = 0; j < 12; j++) for (int i = 0; i < 24; i++) {
        const float x0 = -3.f + i * .25f, x1 = x0 + .25f, y0 = j * .25f, y1 = y0 + .25f;
        const float z00 = -5.f + .15f * std::sin(1.7f * i + .9f * j), z10 = -5.f + .15f * std::sin(1.7f * (i + 1) + .9f * j);
        const float z01 = -5.f + .15f * std::sin(1.7f * i + .9f * (j + 1)), z11 = -5.f + .15f * std::sin(1.7f * (i + 1) + .9f * (j + 1));
        const float a[3] = { x0, y0, z00 }, b[3] = { x1, y0, z10 }, c[3] = { x1, y1, z11 }, d[3] = { x0, y1, z01 };
        tri(a, b, c, 1 + ((i + j) & 1)); tri(a, c, d, 1 + ((i + j) & 1));
    }
    for (int k = 0; k < 4; k++) {
        const float cx = -2.25f + 1.5f * k, a[3] = { cx - .2f, 3.2f, -3.2f }, b[3] = { cx, 3.2f, -2.8f }, c[3] = { cx + .2f, 3.2f, -3.2f };
        tri(a, b, c, 3);                                           // counter-clockwise seen from below: the normal points down
    }
    mats.assign(4, rs_material{});
    const float col[3][3] = { { .7f, .7f, .7f }, { .8f, .3f, .3f }, { .3f, .5f, .8f } };
    for (int m = 0; m < 3; m++) { mats[m].type = 0; for (int c = 0; c < 3; c++) mats[m].baseColor[c] = col[m][c]; }
    mats[3].type = 4; mats[3].baseColor[0] = 14.f; mats[3].baseColor[1] = 12.f; mats[3].baseColor[2] = 9.f;
    for (auto& m : mats) m.baseColorMapId = m.metallicMapId = m.roughnessMapId = m.normalMapId = -1;
}

void make_camera(rs_camera& cam, int frame, bool orbit) {
    std::memset(&cam, 0, sizeof cam);
    cam.resolution[0] = W; cam.resolution[1] = H;
    cam.position[0] = orbit ? .3f * std::sin(.4f * frame) : 0.f; cam.position[1] = 1.4f; cam.position[2] = orbit ? .8f + .1f * frame : .8f;
    cam.rotation[0] = -90.f;
    cam.fov[1] = 28.f; cam.focalDist = 1.f;
}

std::atomic<int> failures{ 0 };
#define CHECK(x) do { int e_ = (x); if (e_) { std::fprintf(stderr, "rank %d: %s failed: %d (%s)\n", rank, #x, e_, rs_last_error()); failures++; return; } } while (0)

void run_rank(int rank, int world, ncclUniqueId id) {
    if (hipSetDevice(rank) != hipSuccess) { std::fprintf(stderr, "rank %d: hipSetDevice failed\n", rank); failures++; return; }
    rs_context* ctx = nullptr;
    CHECK(rs_context_create(rank, &ctx));
    CHECK(rs_context_set_current(ctx));
    CHECK(rs_set_sync(0));                                         // asynchronous launches, as the benchmark runs them
    ncclComm_t nccl = nullptr;
    if (ncclCommInitRank(&nccl, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); failures++; return; }
    rs_comm* comm = nullptr;
    CHECK(rs_comm_create_rccl(nccl, rank, world, &comm));

    std::vector<float> v, n; std::vector<int> matIds; std::vector<rs_material> mats;
    make_scene(v, n, matIds, mats);
    const std::vector<float> uv(matIds.size() * 6, 0.f);
    rs_scene* scene = nullptr;
    CHECK(rs_scene_build((int)matIds.size(), v.data(), n.data(), uv.data(), matIds.data(), (int)mats.size(), mats.data(), &scene));

    for (int orbit = 0; orbit < 2; orbit++) {
        rs_strips* strips = nullptr;
        CHECK(rs_strips_create(comm, W, H, nullptr, &strips));
        rs_gbuffer* g[2] = { nullptr, nullptr }; rs_restir* r[2] = { nullptr, nullptr }; float* img[2] = { nullptr, nullptr };
        const int sets = rank == 0 ? 2 : 1;                        // rank 0: the strips' objects and a full-frame renderer of its own
        for (int k = 0; k < sets; k++) {
            CHECK(rs_gbuffer_create(W, H, &g[k])); CHECK(rs_restir_init(W, H, &r[k]));
            if (hipMalloc((void**)&img[k], sizeof(float) * 3 * W * H) != hipSuccess) { failures++; return; }
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
    ncclCommDestroy(nccl);
    (void)rs_context_set_current(nullptr);
    (void)rs_context_destroy(ctx);
}

}  // namespace

int main(int argc, char** argv) {
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) { std::fprintf(stderr, "no GPU\n"); return 1; }
    const int world = argc > 1 ? std::atoi(argv[1]) : devices;
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
