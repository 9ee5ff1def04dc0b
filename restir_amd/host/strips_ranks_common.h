// strips_ranks_common.h -- what the two "N ranks as N host threads" checks of the strip driver share (strips_rccl_ranks.cpp: one GPU per
// rank over RCCL; strips_loopback_ranks.cpp: every rank on one GPU over an in-process stream-ordered transport): the test scene and
// camera, and how a failure or a hang ends the process.
//
// A rank thread that returned on an error would leave the other ranks waiting for it forever -- in ncclCommInitRank, in an RCCL kernel
// that spins on the GPU for the missing peer, in a mailbox wait -- with main() blocked in join().  So the FIRST failure ends the whole
// process: fail() prints, runs the hook the binary registered (ncclCommAbort on every communicator), and leaves through std::_Exit(1)
// without unwinding; a watchdog thread does the same after `seconds` (a hang without an error).  Nothing is restarted or re-exec'd.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/restir_hip.h"

namespace ranks {

constexpr int W = 320, FRAMES = 4;
inline int H = 200;                                   // (strips_loopback_ranks raises it for more than six ranks: 32 rows per strip at least with the filter)

inline void (*g_abort_hook)() = nullptr;             // e.g. ncclCommAbort on every communicator created so far
inline std::mutex g_fail_mutex;
[[noreturn]] inline void fail(int rank, const char* what, int code) {
    {
        std::lock_guard<std::mutex> lock(g_fail_mutex);       // one report; the other ranks' follow-up failures stay silent
        std::fprintf(stderr, "rank %d: %s failed: %d (%s)\n", rank, what, code, rs_last_error());
        std::fflush(stderr);
        if (g_abort_hook) g_abort_hook();
        std::_Exit(1);
    }
}
#define RANKS_CHECK(x) do { int e_ = (x); if (e_) ranks::fail(rank, #x, e_); } while (0)
#define RANKS_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); ranks::fail(rank, #x, (int)e_); } } while (0)

inline void start_watchdog(int seconds) {
    std::thread([seconds] {
        std::this_thread::sleep_for(std::chrono::seconds(seconds));
        std::fprintf(stderr, "watchdog: still running after %d s -- a rank is waiting for a peer that never answers; ending the process\n", seconds);
        std::fflush(stderr);
        if (g_abort_hook) g_abort_hook();
        std::_Exit(3);
    }).detach();
}

// a floor, a back wall of 24 x 12 slanted facets (so that strips see different geometry) and four small lights facing down
inline void make_scene(std::vector<float>& v, std::vector<float>& n, std::vector<int>& matIds, std::vector<rs_material>& mats) {
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

inline void make_camera(rs_camera& cam, int frame, bool orbit) {
    std::memset(&cam, 0, sizeof cam);
    cam.resolution[0] = W; cam.resolution[1] = H;
    cam.position[0] = orbit ? .3f * std::sin(.4f * frame) : 0.f; cam.position[1] = 1.4f; cam.position[2] = orbit ? .8f + .1f * frame : .8f;
    cam.rotation[0] = -90.f;
    cam.fov[1] = 28.f; cam.focalDist = 1.f;
}

inline rs_scene* build_scene(int rank) {
    std::vector<float> v, n; std::vector<int> matIds; std::vector<rs_material> mats;
    make_scene(v, n, matIds, mats);
    const std::vector<float> uv(matIds.size() * 6, 0.f);
    rs_scene* scene = nullptr;
    RANKS_CHECK(rs_scene_build((int)matIds.size(), v.data(), n.data(), uv.data(), matIds.data(), (int)mats.size(), mats.data(), &scene));
    return scene;
}

}  // namespace ranks
