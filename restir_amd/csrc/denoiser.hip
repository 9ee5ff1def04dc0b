// denoiser.hip -- the EAW a-trous path of src/denoiser.cu:
//   waveletFilter (colour)         src/denoiser.cu:64-134    (one level)
//   EAWaveletFilter::filter        src/denoiser.cu:427-437
//   LeveledEAWFilter               src/denoiser.cu:453-477   (5 levels, sigma 64 / .2 / 1, ping-pong)
//   modulate / add                 src/denoiser.cu:218-248,405-425
//
// The reference reconstructs the world position of every tap with cam.getPosition(qx,qy,depth) --
// a normalize per tap, 25 taps, 5 levels.  getPosition is a pure function of the pixel, so it is
// evaluated once per pixel into a position plane at the start of a filter call (same expression,
// same bits) and the levels gather from it.
#include "rs_internal.h"

using namespace rs;

namespace {

__constant__ float kGaussian5x5[5][5] = {        // src/denoiser.cu:18-24
    { .0030f, .0133f, .0219f, .0133f, .0030f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0219f, .0983f, .1621f, .0983f, .0219f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0030f, .0133f, .0219f, .0133f, .0030f }
};

__global__ void __launch_bounds__(256) k_positions(CamParams cam, const float* __restrict__ depth, const int* __restrict__ primId,
                                                   float* __restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cam.width * cam.height) return;
    const int x = i % cam.width, y = i / cam.width;
    f3 p = splat(0.f);
    if (primId[i] > kNullPrim) p = camera_get_position(cam, x, y, depth[i]);
    st3(pos + (size_t)i * 3, p);
}

__global__ void __launch_bounds__(256) k_wavelet(float* __restrict__ colorOut, const float* __restrict__ colorIn,
                                                 const int* __restrict__ primId, const float* __restrict__ normal,
                                                 const float* __restrict__ pos, int W, int H,
                                                 float sigDepth, float sigNormal, float sigLumin, int level) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const int step = 1 << level;
    const int idxP = y * W + x;
    const int idP = primId[idxP];
    const f3 colorP = ld3(colorIn + (size_t)idxP * 3);
    if (idP <= kNullPrim) { st3(colorOut + (size_t)idxP * 3, colorP); return; }
    const f3 normP = ld3(normal + (size_t)idxP * 3);
    const f3 posP = ld3(pos + (size_t)idxP * 3);

    f3 sum = splat(0.f);
    float sumW = 0.f;
#pragma unroll
    for (int i = -2; i <= 2; i++) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const int qx = x + j * step, qy = y + i * step;
            if (qx >= W || qy >= H || qx < 0 || qy < 0) continue;
            const int idxQ = qy * W + qx;
            if (primId[idxQ] != idP) continue;
            const f3 normQ = ld3(normal + (size_t)idxQ * 3);
            const f3 colorQ = ld3(colorIn + (size_t)idxQ * 3);
            const f3 posQ = ld3(pos + (size_t)idxQ * 3);
            const f3 dc = colorP - colorQ, dn = normP - normQ, dp = posP - posQ;
            const float wC = gmin(1.f, expf(-dot(dc, dc) / sigLumin));
            const float wN = gmin(1.f, expf(-dot(dn, dn) / sigNormal));
            const float wP = gmin(1.f, expf(-dot(dp, dp) / sigDepth));
            const float w = wC * wN * wP * kGaussian5x5[i + 2][j + 2];
            sum = sum + colorQ * w;
            sumW += w;
        }
    }
    st3(colorOut + (size_t)idxP * 3, sumW == 0.f ? colorP : sum / sumW);
}

__global__ void __launch_bounds__(256) k_modulate(float* __restrict__ image, const float* __restrict__ albedo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 c = ld3(image + (size_t)i * 3);
    c = c / 1.f;                                                   // Math::LDRToHDR (mathUtil.h:40-43)
    c = c / ((splat(1.f) - c) + 1e-4f);
    st3(image + (size_t)i * 3, c * vmax(ld3(albedo + (size_t)i * 3), splat(0.f)));
}
__global__ void __launch_bounds__(256) k_add(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

int wavelet_level(const rs_eaw* f, float* out, const float* in, const rs_gbuffer* g, int level) {
    dim3 grid((f->width + 31) / 32, (f->height + 7) / 8);
    hipLaunchKernelGGL(k_wavelet, grid, dim3(256), 0, rs_stream(), out, in, g->devPrimId[g->frameIdx], g->devNormal[g->frameIdx],
                       f->devPos, f->width, f->height, f->sigDepth, f->sigNormal, f->sigLumin, level);
    return rs_after_launch("EAW Filter");
}

}  // namespace

extern "C" {

int rs_eaw_destroy(rs_eaw* f) {
    if (!f) return 0;
    rs_dev_free(f->devTempImg); rs_dev_free(f->devPos);
    delete f;
    return 0;
}

int rs_eaw_create(int width, int height, int level, rs_eaw** out) {
    if (!out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_create: bad size");
    *out = nullptr;
    rs_eaw* f = new rs_eaw();
    f->width = width; f->height = height; f->level = level;
    int e = rs_dev_alloc(&f->devTempImg, (size_t)width * height * 3);
    if (!e) e = rs_dev_alloc(&f->devPos, (size_t)width * height * 3);
    if (e) { rs_eaw_destroy(f); return e; }
    *out = f;
    return 0;
}

int rs_eaw_filter(rs_eaw* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam) {
    if (!f || !devColorOut || !*devColorOut || !devColorIn || !g || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: null argument");
    if (g->width != f->width || g->height != f->height || cam->resolution[0] != f->width || cam->resolution[1] != f->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: size mismatch");
    const int n = f->width * f->height;
    hipLaunchKernelGGL(k_positions, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), rs_make_cam_params(cam),
                       g->devDepth[g->frameIdx], g->devPrimId[g->frameIdx], f->devPos);
    RS_TRY(rs_after_launch("EAW positions"));
    // LeveledEAWFilter::filter (denoiser.cu:463-477): level 0 into out, then four ping-pongs with the
    // internal buffer; the caller's pointer and the internal one are swapped after each
    RS_TRY(wavelet_level(f, *devColorOut, devColorIn, g, 0));
    for (int level = 1; level <= 4; level++) {
        RS_TRY(wavelet_level(f, f->devTempImg, *devColorOut, g, level));
        float* t = *devColorOut; *devColorOut = f->devTempImg; f->devTempImg = t;
    }
    return 0;
}

int rs_modulate_albedo(float* devImage, const rs_gbuffer* g) {
    if (!devImage || !g) return rs_fail(RS_ERR_INVALID_ARGUMENT, "modulateAlbedo: null argument");
    const int n = g->width * g->height;
    hipLaunchKernelGGL(k_modulate, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devImage, g->devAlbedo, n);
    return rs_after_launch("modulate");
}

int rs_add_image(float* devImage, const float* devIn, int width, int height) {
    if (!devImage || !devIn || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "addImage: bad argument");
    const int n = width * height * 3;
    hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devImage, devImage, devIn, n);
    return rs_after_launch("addImage");
}

int rs_add_image3(float* devOut, const float* devIn1, const float* devIn2, int width, int height) {
    if (!devOut || !devIn1 || !devIn2 || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "addImage: bad argument");
    const int n = width * height * 3;
    hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devOut, devIn1, devIn2, n);
    return rs_after_launch("addImage");
}

}  // extern "C"
