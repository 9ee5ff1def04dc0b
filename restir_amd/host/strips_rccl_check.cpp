// strips_rccl_check.cpp -- the C++ caller of the strip driver over RCCL (include/restir_hip.h rs_comm_create_rccl, rs_strips_*),
// reduced to what one GPU can run: a communicator of ONE rank (ncclCommInitRank), the run-time binding of the library to
// librccl checked with a transfer to the rank itself (ncclSend / ncclRecv inside a group, as a frame issues them), and one
// strip frame with world = 1 (no neighbours) compared with rs_restir_direct.  With N ranks the only changes are the unique id
// handed to the other processes and `world`; INTEGRATION.md shows that form.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/restir_hip.h"

#define CHECK(x) do { int e_ = (x); if (e_) { std::fprintf(stderr, "%s failed: %d (%s)\n", #x, e_, rs_last_error()); return 1; } } while (0)

// `strips_rccl_check latency`: what ONE RCCL group of a strip frame costs where one GPU can measure it -- the rank sends to itself, so
// nothing crosses xGMI and the copy runs at HBM speed: a LOWER BOUND of the group's duration on a multi-GPU node (there the 653 KB per
// edge travel over one 153 GB/s link: + >= 4.3 us per direction, and the ranks' skew is added), and the real figure for what does not
// depend on the wire: RCCL's kernel launch, its host-side cost per group, and the gap it leaves on the stream.
//   frame-shaped group      two sends + two receives of 1920 x 5 rows x 68 B = 652 800 B (a middle strip's two neighbours)
//   + deferred gather       one more send + receive of 1920 x 135 rows x 4 B = 1 036 800 B (a rank's RGBA8 strip to rank 0)
// each 200 times back to back between two HIP events on the stream that carries them (GPU time per group incl. the launch gap) and
// against the host clock (host time per group: what rs_strips_frame's caller pays in ncclGroupEnd), on an ordinary stream as the
// library stream and on a second stream ordered against it by events per group (rs_strips_set_comm_stream(s, 1)'s form).
static int latency_mode() {
    if (hipSetDevice(0) != hipSuccess) return 1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    ncclComm_t nccl = nullptr;
    if (ncclCommInitRank(&nccl, 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
    const size_t edge = 1920u * 5u * 68u, strip = 1920u * 135u * 4u;
    char *sendBuf[3], *recvBuf[3];
    const size_t bytes[3] = { edge, edge, strip };
    for (int i = 0; i < 3; i++)
        if (hipMalloc((void**)&sendBuf[i], bytes[i]) != hipSuccess || hipMalloc((void**)&recvBuf[i], bytes[i]) != hipSuccess) return 1;
    hipStream_t lib = nullptr, own = nullptr;
    if (hipStreamCreateWithFlags(&lib, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) return 1;
    hipEvent_t e0, e1, packed, arrived;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventCreateWithFlags(&packed, hipEventDisableTiming); (void)hipEventCreateWithFlags(&arrived, hipEventDisableTiming);
    const int reps = 200;
    std::printf("# RCCL group to self on one MI355X (lower bound: no xGMI, no second rank); %d groups back to back per line\n", reps);
    for (int ops = 2; ops <= 3; ops++) {
        for (int ownStream = 0; ownStream <= 1; ownStream++) {
            hipStream_t st = ownStream ? own : lib;
            auto group = [&]() {
                if (ownStream) { (void)hipEventRecord(packed, lib); (void)hipStreamWaitEvent(own, packed, 0); }
                ncclGroupStart();
                for (int i = 0; i < ops; i++) { ncclSend(sendBuf[i], bytes[i], ncclUint8, 0, nccl, st); ncclRecv(recvBuf[i], bytes[i], ncclUint8, 0, nccl, st); }
                const ncclResult_t r = ncclGroupEnd();
                if (ownStream) { (void)hipEventRecord(arrived, own); (void)hipStreamWaitEvent(lib, arrived, 0); }
                return r;
            };
            for (int i = 0; i < 20; i++) if (group() != ncclSuccess) { std::fprintf(stderr, "ncclGroupEnd failed\n"); return 1; }
            (void)hipStreamSynchronize(lib); (void)hipStreamSynchronize(own);
            (void)hipEventRecord(e0, lib);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) (void)group();
            const auto t1 = std::chrono::steady_clock::now();
            (void)hipEventRecord(e1, lib);
            (void)hipStreamSynchronize(lib); (void)hipStreamSynchronize(own);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            // one group alone, the stream idle before it: its latency from enqueue to completion as the host sees it
            double alone = 0;
            for (int i = 0; i < 20; i++) {
                (void)hipStreamSynchronize(lib); (void)hipStreamSynchronize(own);
                const auto a0 = std::chrono::steady_clock::now();
                (void)group();
                (void)hipStreamSynchronize(lib); (void)hipStreamSynchronize(own);
                alone += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a0).count();
            }
            std::printf("%s, transfers on %s: GPU %.1f us per group (events on the library stream), host %.1f us per group call, one group alone enqueue-to-done %.1f us\n",
                        ops == 2 ? "frame-shaped group (2 x 652 800 B each way)  " : "frame-shaped group + deferred gather (1 036 800 B)",
                        ownStream ? "a stream of their own (2 events per group)" : "the library stream                       ",
                        ms * 1e3 / reps, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, alone / 20);
        }
    }
    // for scale: the same bytes as plain device-to-device copies on the library stream
    (void)hipEventRecord(e0, lib);
    for (int i = 0; i < reps; i++) for (int k = 0; k < 3; k++) (void)hipMemcpyAsync(recvBuf[k], sendBuf[k], bytes[k], hipMemcpyDeviceToDevice, lib);
    (void)hipEventRecord(e1, lib);
    (void)hipStreamSynchronize(lib);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("for scale: the three buffers as hipMemcpyAsync device-to-device on the library stream: %.1f us per set\n", ms * 1e3 / reps);
    ncclCommDestroy(nccl);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "latency") == 0) return latency_mode();
    CHECK(rs_init(0));
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
    ncclComm_t nccl = nullptr;
    if (ncclCommInitRank(&nccl, 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
    rs_comm* comm = nullptr;
    CHECK(rs_comm_create_rccl(nccl, 0, 1, &comm));

    // 1. the transport: 1 MiB to ourselves
    const size_t n = 1 << 20;
    std::vector<unsigned char> h(n), back(n);
    for (size_t i = 0; i < n; i++) h[i] = (unsigned char)(i * 131u + 7u);
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n) != hipSuccess || hipMalloc(&b, n) != hipSuccess) return 1;
    (void)hipMemcpy(a, h.data(), n, hipMemcpyHostToDevice);
    (void)hipMemset(b, 0, n);
    CHECK(rs_comm_self_exchange(comm, a, b, n));
    (void)hipMemcpy(back.data(), b, n, hipMemcpyDeviceToHost);
    if (std::memcmp(h.data(), back.data(), n) != 0) { std::fprintf(stderr, "self exchange over RCCL: data differs\n"); return 1; }
    std::printf("ncclSend / ncclRecv through rs_comm: 1 MiB to self ok\n");

    // 2. a strip frame of a one-rank world equals ReSTIRDirect
    const int W = 96, H = 64;
    const float tri[2 * 9] = { -1, 0, -3,  1, 0, -3,  0, 1.5f, -3,     -.5f, 1.2f, -2,  0, 1.2f, -2.6f,  .5f, 1.2f, -2 };      // the second one (the light) faces down
    float nrm[2 * 9]; for (int i = 0; i < 6; i++) { nrm[i * 3] = 0; nrm[i * 3 + 1] = i < 3 ? 0.f : -1.f; nrm[i * 3 + 2] = i < 3 ? 1.f : 0.f; }
    const float uv[2 * 6] = {};
    const int matIds[2] = { 0, 1 };
    rs_material mats[2];
    std::memset(mats, 0, sizeof mats);
    mats[0].type = 0; mats[0].baseColor[0] = mats[0].baseColor[1] = mats[0].baseColor[2] = .8f;
    mats[1].type = 4; mats[1].baseColor[0] = mats[1].baseColor[1] = mats[1].baseColor[2] = 12.f;          // Light
    for (auto& m : mats) m.baseColorMapId = m.metallicMapId = m.roughnessMapId = m.normalMapId = -1;
    rs_scene* scene = nullptr;
    CHECK(rs_scene_build(2, tri, nrm, uv, matIds, 2, mats, &scene));
    rs_camera cam;
    std::memset(&cam, 0, sizeof cam);
    cam.resolution[0] = W; cam.resolution[1] = H;
    cam.position[0] = 0; cam.position[1] = .6f; cam.position[2] = 1.f;
    cam.rotation[0] = -90.f;
    cam.fov[1] = 25.f; cam.focalDist = 1.f;
    CHECK(rs_camera_update(&cam));
    rs_strips* strips = nullptr;
    CHECK(rs_strips_create(comm, W, H, nullptr, &strips));
    float* img[2] = { nullptr, nullptr };
    rs_gbuffer* g[2]; rs_restir* r[2];
    for (int k = 0; k < 2; k++) {
        CHECK(rs_gbuffer_create(W, H, &g[k])); CHECK(rs_restir_init(W, H, &r[k]));
        if (hipMalloc((void**)&img[k], sizeof(float) * 3 * W * H) != hipSuccess) return 1;
        (void)hipMemset(img[k], 0, sizeof(float) * 3 * W * H);
    }
    for (int frame = 0; frame < 3; frame++) {
        CHECK(rs_strips_frame(strips, r[0], scene, &cam, g[0], img[0], 0, frame, 3));
        if (frame == 2) {                                         // LeveledEAWFilter through the strip driver and directly
            rs_eaw* f[2]; float* out[2] = { nullptr, nullptr };
            CHECK(rs_eaw_create(W, H, 5, &f[0])); CHECK(rs_eaw_create(W, H, 5, &f[1]));
            if (hipMalloc((void**)&out[1], sizeof(float) * 3 * W * H) != hipSuccess) return 1;
            CHECK(rs_strips_eaw_filter(strips, f[0], g[0], &cam, img[0], &out[0]));
            CHECK(rs_eaw_filter(f[1], &out[1], img[0], g[0], &cam));
            CHECK(rs_synchronize());
            std::vector<float> u(3 * W * H), v(3 * W * H);
            (void)hipMemcpy(u.data(), out[0], sizeof(float) * u.size(), hipMemcpyDeviceToHost);
            (void)hipMemcpy(v.data(), out[1], sizeof(float) * v.size(), hipMemcpyDeviceToHost);
            if (std::memcmp(u.data(), v.data(), sizeof(float) * u.size()) != 0) { std::fprintf(stderr, "rs_strips_eaw_filter differs from rs_eaw_filter\n"); return 1; }
            std::printf("rs_strips_eaw_filter (world 1) == rs_eaw_filter\n");
            rs_eaw_destroy(f[0]); rs_eaw_destroy(f[1]);
            (void)hipFree(out[1]);                                // whichever buffer the reference-style pointer swap left with the caller
        }
        CHECK(rs_gbuffer_update(g[0], &cam));
        CHECK(rs_strips_exchange_history(strips, r[0], g[0]));    // a world of one rank has nobody to tell
        CHECK(rs_strips_gather(strips, img[0], 12, -1));
        CHECK(rs_gbuffer_render(g[1], scene, &cam));
        CHECK(rs_restir_direct(r[1], scene, &cam, g[1], img[1], 0, frame, 3));
        CHECK(rs_gbuffer_update(g[1], &cam));
    }
    CHECK(rs_synchronize());
    std::vector<float> x(3 * W * H), y(3 * W * H);
    (void)hipMemcpy(x.data(), img[0], sizeof(float) * x.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(y.data(), img[1], sizeof(float) * y.size(), hipMemcpyDeviceToHost);
    double sum = 0; for (float v : x) sum += v;
    if (std::memcmp(x.data(), y.data(), sizeof(float) * x.size()) != 0 || !(sum > 0)) { std::fprintf(stderr, "strip frame differs from ReSTIRDirect (sum %g)\n", sum); return 1; }
    std::printf("rs_strips_frame (world 1) == rs_restir_direct, radiance sum %.3f\n", sum);
    rs_strips_destroy(strips); rs_comm_destroy(comm);
    for (int k = 0; k < 2; k++) { rs_restir_free(r[k]); rs_gbuffer_destroy(g[k]); (void)hipFree(img[k]); }
    rs_scene_destroy(scene);
    (void)hipFree(a); (void)hipFree(b);
    ncclCommDestroy(nccl);
    std::printf("strips_rccl_check ok\n");
    return 0;
}
