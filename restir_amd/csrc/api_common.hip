// api_common.hip -- library state (device, stream, sync mode, last error) of librestir_hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "rs_internal.h"

namespace {
std::mutex g_errMutex;
std::string g_lastError;
hipStream_t g_stream = nullptr;
bool g_sync = true;
// Auxiliary streams (asynchronous mode only): 0 carries GBuffer::render, 1 + k the primary-ray + RIS + shadow-ray kernels of every
// kChains-th frame (frames take the chains in turn, so that these chains of consecutive frames overlap each other
// as well as the passes of the frames before).
// Neither reads what the temporal / spatial passes of the previous frame write, so with the per-frame surface planes
// double-buffered and the G-buffer planes in a ring of three they run next to those passes; the objects own the events
// that order them (rs_gbuffer, rs_restir).
hipStream_t g_aux[1 + rs_restir::kChains] = {};
int g_auxMode = -1;                                     // -1: not decided yet; 0 off; 1 on
int g_fuseMode = -1;                                    // deferred G-buffer render walked with the primary rays: -1 from the environment
}  // namespace

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

hipStream_t rs_stream() { return g_stream; }
bool rs_sync_enabled() { return g_sync; }

// The auxiliary stream i, or nullptr when launches are synchronous (rs_set_sync(1): nothing to overlap) or the feature is
// off (rs_set_side_stream(0) / RS_SIDE_STREAM=0).
hipStream_t rs_aux_stream(int i) {
    if (g_auxMode < 0) {
        const char* e = std::getenv("RS_SIDE_STREAM");
        g_auxMode = (e && e[0] == '0') ? 0 : 1;
    }
    if (g_sync || !g_auxMode) return nullptr;
    if (!g_aux[i] && hipStreamCreateWithFlags(&g_aux[i], hipStreamNonBlocking) != hipSuccess) { g_aux[i] = nullptr; g_auxMode = 0; return nullptr; }
    return g_aux[i];
}
// In asynchronous mode GBuffer::render can be deferred and launched by ReSTIRDirect together with its primary rays (the two rays
// of a pixel in one packet walk).  That saves 10 % of the walk work, 4 % of a Sponza-class 1080p frame, but the slowest tile of
// the launch takes almost twice as long, which costs 28 % on the Bistro-class scene whose closest-hit kernels are tails of a
// few long tiles: by default every rs_restir measures the frame period both ways once and keeps the
// faster (restir.hip).  RS_FUSE_GBUFFER=0 / 1 force it off / on.
// 0 never, 1 always (launches of at least three rounds of wave slots), 2 always (any size), 3 decided per rs_restir by measuring
int rs_fuse_mode() {
    if (g_fuseMode < 0) {
        const char* e = std::getenv("RS_FUSE_GBUFFER");
        g_fuseMode = !e ? 3 : e[0] == '1' ? 1 : e[0] == '0' ? 0 : 3;
    }
    return g_fuseMode;
}
bool rs_fuse_enabled() { return rs_fuse_mode() != 0; }
int rs_aux_synchronize() {
    for (hipStream_t st : g_aux) if (st) RS_HIP(hipStreamSynchronize(st));
    return 0;
}

int rs_after_launch(const char* what) {
    RS_TRY(rs_check_hip(hipGetLastError(), what));
    if (g_sync) RS_TRY(rs_check_hip(hipStreamSynchronize(g_stream), what));
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

extern "C" {

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
    return 0;
}

int rs_set_stream(void* hipStream) {
    if ((hipStream_t)hipStream != g_stream) RS_TRY(rs_synchronize());   // events recorded on the old stream order the auxiliary ones
    g_stream = (hipStream_t)hipStream;
    return 0;
}
int rs_set_sync(int sync) { g_sync = sync != 0; return 0; }
int rs_set_side_stream(int enable) {
    g_auxMode = enable ? 1 : 0;                         // work already enqueued on the auxiliary streams is still joined by its consumers
    g_fuseMode = enable == 2 ? 1 : enable == 3 ? 2 : enable == 4 ? 3 : 0;      // 2 always, 3 always and at any size (tests), 4 measured
    return 0;
}
int rs_synchronize(void) {
    RS_TRY(rs_aux_synchronize());
    RS_TRY(rs_check_hip(hipStreamSynchronize(g_stream), "rs_synchronize"));
    return rs_aux_synchronize();                        // (an auxiliary launch may have been waiting for the library stream)
}

}  // extern "C"
