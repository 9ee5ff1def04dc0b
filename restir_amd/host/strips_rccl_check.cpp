// strips_rccl_check.cpp -- the C++ caller of the strip driver over RCCL (include/restir_hip.h rs_comm_create_rccl, rs_strips_*),
// reduced to what one GPU can run: a communicator of ONE rank (ncclCommInitRank), the run-time binding of the library to
// librccl checked with a transfer to the rank itself (ncclSend / ncclRecv inside a group, as a frame issues them), and one
// strip frame with world = 1 (no neighbours) compared with rs_restir_direct.  With N ranks the only changes are the unique id
// handed to the other processes and `world`; INTEGRATION.md shows that form.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/restir_hip.h"

#define CHECK(x) do { int e_ = (x); if (e_) { std::fprintf(stderr, "%s failed: %d (%s)\n", #x, e_, rs_last_error()); return 1; } } while (0)

int main() {
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
