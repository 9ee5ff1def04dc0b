// pathtrace.hip -- the non-ReSTIR baseline pass and the display conversion of src/pathtrace.cu:
//   PTDirectKernel / pathTraceDirect   (src/pathtrace.cu:279-328,457-476)
//   sendImageToPBO / copyImageToPBO    (src/pathtrace.cu:30-56,108-113; tone-map ops mathUtil.h:102-117)
#include <cstdlib>

#include "rs_internal.h"

using namespace rs;

namespace {

// TEX: the scene has texture maps or an environment map
// 8 waves per SIMD (64 VGPRs instead of 70-80, 28-108 B of scratch): 2.04 -> 1.94 ms per 1080p frame on the bench scene
#ifndef RS_PT_BLOCKS
#define RS_PT_BLOCKS 8
#endif
template <bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_PT_BLOCKS) k_pt_direct(DevScene s, CamParams cam, float* __restrict__ directIllum,
                                                   int looper, int iter, int tilesX, unsigned long long* rayCount) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    const int x = bx * 32 + wave * 8 + (lane & 7);
    const int y = by * 8 + (lane >> 3);
    int walks = 0;
    const bool inside = x < cam.width && y < cam.height;
    const int index = y * cam.width + x;
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::seeded(s.sampleSeq, looper, inside ? index : 0, 0);       // pathtrace.cu:288
    f4 r = rng.uniform4();
    Ray ray = camera_sample(cam, x, y, r.x, r.y);
    Hit h = trace_closest_packet(s, ray, inside);       // all 64 lanes take part in the wave's walk
    // The shadow ray of sampleDirectLight (scene.h:427-459) is walked by the WHOLE wave through the shadow tree (trace_occluded_wave), as
    // in gi.hip: the light sample first (sampleDirectLight tests occlusion towards the sampled point before it looks at the side or
    // the pdf, so every sampling lane has a segment), then one cooperative walk, then the shading.
    f3 direct = splat(0.f), norm = splat(0.f), wo = splat(0.f);
    SurfMat m = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    LightSample c;
    c.pdf = kInvalidPdf; c.Li = splat(0.f); c.wi = splat(0.f); c.dist = 0.f; c.point = h.pos; c.id = 0; c.bu = c.bv = 0.f;
    bool nee = false;
    if (inside) {
        walks = 1;
        if (h.primId == kNullPrim) {
            if (TEX && s.envTex >= 0) direct = env_radiance(s, ray.d);          // pathtrace.cu:295-297
        }
        else {
            norm = h.norm;
            m = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);    // pathtrace.cu:301
            if (m.type == 4) {
                direct = m.baseColor;
            }
            else {
                wo = -ray.d;
                const bool delta = m.type == 2;
                if (!delta && dot(norm, wo) < 0.f) norm = -norm;
                if (!delta) {
                    const f4 rl = rng.uniform4();
                    nee = s.numLights > 0;
                    if (nee) c = (TEX && s.envTex >= 0) ? sample_light_nv<true, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, h.pos, rl)
                                                        : sample_light_nv<false, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, h.pos, rl);
                }
            }
        }
    }
    // The segment's visibility decides whether `value` becomes the pixel's radiance.  It is not asked (the segment is counted as the
    // reference's testOcclusion call, not walked) where the answer cannot matter: a sample without a valid pdf (a single-sided light that
    // faces away, scene.h:448-452: InvalidPdf either way) and a value whose three components are all +0 (the light is below the surface's
    // horizon): `direct` is that very +0 already.  As in gi.hip.
    f3 value = splat(0.f);
    const bool valid = nee && c.pdf > 0.f;
    if (valid) value = ((c.Li * eval_bsdf(m.type, m.baseColor, m.metallic, m.roughness, norm, wo, c.wi)) * sat_dot(norm, c.wi)) / c.pdf;
    const bool matters = valid && (__float_as_uint(value.x) | __float_as_uint(value.y) | __float_as_uint(value.z)) != 0u;
    const bool occluded = trace_occluded_wave(s, h.pos, c.point, matters);
    if (nee) {
        walks++;
        if (matters && !occluded) direct = value;
    }
    if (inside) {
        float* o = directIllum + (size_t)index * 3;
        st3(o, (ld3(o) * (float)iter + direct) / (float)(iter + 1));
    }
    // wave-level sum of walks, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) walks += __shfl_down(walks, off);
    if (lane == 0 && walks) atomicAdd(rayCount, (unsigned long long)walks);
}

__device__ __forceinline__ float filmic_curve(float c) {
    return (c * (c * 0.22f + 0.03f) + 0.002f) / (c * (c * 0.22f + 0.3f) + 0.06f) - 1.f / 30.f;
}
// correctGamma: glm::pow(c, 1/2.2f).  Evaluated in double and rounded once so that the 8-bit
// quantisation below sees the correctly rounded float power.
__device__ __forceinline__ float gamma_pow(float c) { return (float)pow((double)c, (double)(1.f / 2.2f)); }

// clamp(int(pow(c, 1/2.2) * 255), 0, 255) -- the only use of the power in sendImageToPBO (pathtrace.cu:50-55).  Three
// double-precision pows per pixel made this streaming kernel VALU-bound at 0.9 TB/s; the integer only changes at the
// 255 boundaries, so exp2(log2(c) / 2.2) (v_log_f32 / v_exp_f32, tests: test_fast_gamma_equals_exact) decides the byte
// unless c^(1/2.2) * 255 lies within 2e-3 of an integer (8x the error bound), and only then the exact power is evaluated.
template <bool EXACT>
__device__ __forceinline__ int gamma_byte(float c) {
    if (!EXACT && c >= 0.f && c < 1e-6f) return 0;              // (1e-6)^(1/2.2) * 255 = 0.48: the power is monotone, the byte is 0
    if (!EXACT && c >= 1e-6f && c < 1e6f) {                       // |log2 c| <= 20: total relative error < 1e-6, 2.6e-4 in v
        const float v = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.f / 2.2f)) * 255.f;
        if (v >= 256.f) return 255;
        if (gabs(v - rintf(v)) > 2e-3f) return f2i(v);          // v < 256: no clamp needed; truncation as in the reference
    }
    return iclamp(f2i(gamma_pow(c) * 255.f), 0, 255);
}

template <bool EXACT>
__global__ void __launch_bounds__(256) k_send_image_to_pbo(uchar4* __restrict__ pbo, const float* __restrict__ image,
                                                           int n, int toneMapping, float scale) {
    RS_SETPRIO(RS_PRIO_STREAM);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 c = ld3(image + (size_t)i * 3) * scale;
    if (toneMapping == 1) {                                            // Math::filmic
        f3 t = c * 1.6f;
        const float d = filmic_curve(11.2f);
        c = mk3(filmic_curve(t.x) / d, filmic_curve(t.y) / d, filmic_curve(t.z) / d);
    }
    else if (toneMapping == 2) {                                       // Math::ACES
        c = (c * (c * 2.51f + 0.03f)) / (c * (c * 2.43f + 0.59f) + 0.14f);
    }
    pbo[i] = make_uchar4((unsigned char)gamma_byte<EXACT>(c.x), (unsigned char)gamma_byte<EXACT>(c.y), (unsigned char)gamma_byte<EXACT>(c.z), 0);
}

// The debug-view overloads (pathtrace.cu:58-106): gamma only, no tone map.  kind 0: vec2 image -> (x, y, 0);
// kind 1: float image -> grey; kind 2: int image (a pixel index, e.g. devMotion) -> (idx % width, idx / HEIGHT) / (width, height)
// -- the reference divides by height where the row length is width (pathtrace.cu:100); kept as is.
__global__ void __launch_bounds__(256) k_send_debug_to_pbo(uchar4* __restrict__ pbo, const void* __restrict__ image,
                                                            int width, int height, int kind) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    f3 c;
    if (kind == 0) { const float* p = (const float*)image + (size_t)i * 2; c = mk3(p[0], p[1], 0.f); }
    else if (kind == 1) c = splat(((const float*)image)[i]);
    else {
        const int v = ((const int*)image)[i];
        const int px = v % width, py = v / height;
        c = mk3((float)px / (float)width, (float)py / (float)height, 0.f);
    }
    c = mk3(gamma_pow(c.x), gamma_pow(c.y), gamma_pow(c.z));
    pbo[i] = make_uchar4((unsigned char)iclamp(f2i(c.x * 255.f), 0, 255), (unsigned char)iclamp(f2i(c.y * 255.f), 0, 255),
                         (unsigned char)iclamp(f2i(c.z * 255.f), 0, 255), 0);
}


int copy_debug(void* devPBO, const void* devImage, int width, int height, int kind) {
    rs_ctx_scope scope(nullptr);                         // no object: the thread's current context
    if (!devPBO || !devImage || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "copyImageToPBO: bad argument");
    const int n = width * height;
    hipLaunchKernelGGL(k_send_debug_to_pbo, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), (uchar4*)devPBO, devImage, width, height, kind);
    return rs_after_launch("copyImageToPBO");
}

}  // namespace

extern "C" {

int rs_path_trace_init(void) {
    rs_ctx_scope scope(nullptr);
    if (!rs_ctx()->ptRayCount) RS_TRY(rs_dev_alloc(&rs_ctx()->ptRayCount, 1));
    return 0;
}
int rs_path_trace_free(void) { rs_ctx_scope scope(nullptr); rs_dev_free(rs_ctx()->ptRayCount); return 0; }

int rs_path_trace_direct(const rs_scene* scene, const rs_camera* cam, float* devDirectIllum, int iter, int looper, unsigned long long* rays) {
    RS_SCOPE(scene);
    if (!scene || !cam || !devDirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "pathTraceDirect: null argument");
    RS_TRY(rs_check_looper(scene, looper, "pathTraceDirect"));
    RS_TRY(rs_path_trace_init());
    RS_TRY(rs_denoise_order(devDirectIllum));           // an image a filter on the denoise stream may still be reading
    RS_HIP(hipMemsetAsync(rs_ctx()->ptRayCount, 0, 8, rs_stream()));
    const int W = cam->resolution[0], H = cam->resolution[1];
    const int tilesX = (W + 31) / 32, tilesY = (H + 7) / 8;
    RS_LAUNCH2(k_pt_direct, scene->textured, scene->dev.sampleSeq != nullptr, dim3(tilesX * tilesY), dim3(256), rs_stream(), scene->dev,
               rs_make_cam_params(cam), devDirectIllum, looper, iter, tilesX, rs_ctx()->ptRayCount);
    RS_TRY(rs_after_launch("pathTrace"));
    if (rays) {
        RS_HIP(hipStreamSynchronize(rs_stream()));
        RS_HIP(hipMemcpy(rays, rs_ctx()->ptRayCount, 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int rs_copy_image_to_pbo(void* devPBO, const float* devImage, int width, int height, int toneMapping, float scale) {
    rs_ctx_scope scope(nullptr);                         // no object: the thread's current context
    if (!devPBO || !devImage || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "copyImageToPBO: bad argument");
    const int n = width * height;
    // An image that the denoise stream has written (rs_set_denoise_stream(1): the result of LeveledEAWFilter) is converted there, in that
    // stream's order -- the library stream does not wait for the filter; the display buffer is then ordered by an event like the
    // filter's own buffers (rs_join_denoise_stream, rs_synchronize, any library call that is handed it).
    rs_denoise_scope onDenoiseStream(!rs_denoise_owns(devPBO), rs_denoise_owns(devImage));      // (fork for a display buffer that stream has never written: after its last use on the library stream)
    RS_TRY(onDenoiseStream.err);
    if (!onDenoiseStream.active) { RS_TRY(rs_denoise_order(devImage, false)); RS_TRY(rs_denoise_order(devPBO)); }
    if (std::getenv("RS_EXACT_GAMMA"))      // test switch: every pixel through the double-precision power
        hipLaunchKernelGGL(k_send_image_to_pbo<true>, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), (uchar4*)devPBO, devImage, n, toneMapping, scale);
    else
        hipLaunchKernelGGL(k_send_image_to_pbo<false>, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), (uchar4*)devPBO, devImage, n, toneMapping, scale);
    if (onDenoiseStream.active) RS_TRY(rs_denoise_mark(devPBO, (size_t)n * 4, false));
    return rs_after_launch("copyImageToPBO");
}

int rs_copy_image2_to_pbo(void* devPBO, const float* devImage, int width, int height) { return copy_debug(devPBO, devImage, width, height, 0); }
int rs_copy_imagef_to_pbo(void* devPBO, const float* devImage, int width, int height) { return copy_debug(devPBO, devImage, width, height, 1); }
int rs_copy_imagei_to_pbo(void* devPBO, const int* devImage, int width, int height) { return copy_debug(devPBO, devImage, width, height, 2); }

}  // extern "C"
