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
// side stream: work that the next kernels on the library stream do not depend on (the G-buffer render next to the
// primary-ray and RIS kernels) runs here and is joined where it is first consumed
hipStream_t g_side = nullptr;
hipEvent_t g_sideFork = nullptr, g_sideDone = nullptr;
bool g_sidePending = false;
int g_sideMode = -1;                                    // -1: not decided yet; 0 off; 1 on
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

// Returns the side stream, ordered after everything enqueued on the library stream so far, or nullptr when launches are
// synchronous (rs_set_sync(1): nothing to overlap) or RS_SIDE_STREAM=0.
hipStream_t rs_side_fork() {
    if (g_sideMode < 0) {
        const char* e = std::getenv("RS_SIDE_STREAM");
        g_sideMode = (e && e[0] == '0') ? 0 : 1;
    }
    if (g_sync || !g_sideMode) return nullptr;
    if (!g_side) {
        if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) { g_side = nullptr; g_sideMode = 0; return nullptr; }
        if (hipEventCreateWithFlags(&g_sideFork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g_sideDone, hipEventDisableTiming) != hipSuccess) { g_sideMode = 0; return nullptr; }
    }
    if (hipEventRecord(g_sideFork, g_stream) != hipSuccess || hipStreamWaitEvent(g_side, g_sideFork, 0) != hipSuccess) return nullptr;
    return g_side;
}
// after the launches on the side stream
int rs_side_submitted() {
    RS_HIP(hipEventRecord(g_sideDone, g_side));
    g_sidePending = true;
    return 0;
}
// the library stream waits for the side stream's work; called by every consumer of what was produced there
int rs_side_join() {
    if (!g_sidePending) return 0;
    g_sidePending = false;
    return rs_check_hip(hipStreamWaitEvent(g_stream, g_sideDone, 0), "side-stream join");
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
    RS_TRY(rs_side_join());                             // pending side work is ordered into the stream being left
    g_stream = (hipStream_t)hipStream;
    return 0;
}
int rs_set_sync(int sync) { g_sync = sync != 0; return 0; }
int rs_set_side_stream(int enable) {
    RS_TRY(rs_side_join());
    g_sideMode = enable ? 1 : 0;
    return 0;
}
int rs_synchronize(void) {
    RS_TRY(rs_side_join());
    return rs_check_hip(hipStreamSynchronize(g_stream), "rs_synchronize");
}

}  // extern "C"
