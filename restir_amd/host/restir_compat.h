// restir_compat.h -- the reference's host-side names on top of the C ABI (include/restir_hip.h).
//
// A maintainer of HummaWhite/ReSTIR replaces src/restir.cu, src/pathtrace.cu, src/gbuffer.cu,
// src/denoiser.cu (and the device side of scene.h / intersections.h) by this header + librestir_hip.so;
// runCuda() (src/main.cpp:146-185) and main() (:78-100) then compile against the same calls:
//
//     gBuffer.render(scene->devScene, scene->camera);
//     ReSTIRDirect(devDirectIllum, iteration, gBuffer);        // or pathTraceDirect(devDirectIllum, iteration)
//     copyImageToPBO(devPBO, devImage, width, height, Settings::toneMapping);
//     gBuffer.update(scene->camera);
//
// Differences that remain visible to the caller:
//   * image pointers are hipMalloc memory; the CUDA-GL PBO interop of main.cpp:176-181 is replaced by
//     a plain device buffer (display interop is outside this path);
//   * Scene(filename) + buildDevData() go through rs_scene_file_load / rs_scene_build_textured (parser, OBJ reader, instance
//     baking, BVH, light table and alias tables are built by the library with the reference's exact results); image files
//     named in a scene must be PNG, JPEG, TGA, BMP, binary PPM or Radiance HDR, other formats are decoded by the caller and passed as arrays.
// Errors keep the reference's convention: print and exit (checkCUDAError, src/cudaUtil.h:13-31).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/restir_hip.h"

#ifdef RESTIR_COMPAT_USE_GLM
#include <glm/glm.hpp>
namespace rsc { using vec3 = glm::vec3; }
#else
namespace rsc { struct vec3 { float x, y, z; }; }
#endif
static_assert(sizeof(rsc::vec3) == 12, "glm::vec3 must be 3 packed floats");

namespace rsc {
inline void check(int code, const char* msg) {          // checkCUDAError(msg)
    if (code == 0) return;
    std::fprintf(stderr, "HIP error: %s: %s\n", msg, rs_last_error());
    std::exit(EXIT_FAILURE);
}
}  // namespace rsc

// ---- src/common.h:4, src/sampler.h:9-11 -----------------------------------------------------------
// The reference picks its sampler at compile time; a caller built with -DSAMPLER_USE_SOBOL=true gets the Sobol branch: buildDevData
// reads "sobol_10k_200.bin" from the working directory as DevScene::create does (src/scene.cpp:500-506; tools/make_sobol_table.py
// writes one) and the launchers wrap State::looper at SobolSampleNum (src/restir.cu:441-445, src/pathtrace.cu:450-454).
#ifndef SAMPLER_USE_SOBOL
#define SAMPLER_USE_SOBOL false
#endif
#define SobolSampleNum 10000
#define SobolSampleDim 200

// ---- src/common.h:18-67 -------------------------------------------------------------------------
struct ToneMapping { enum { None = 0, Filmic = 1, ACES = 2 }; };
struct ReservoirReuse { enum { None = 0, Temporal = 1, Spatial = 2, Spatiotemporal = 3 }; };

struct Scene;
struct Settings {
    static inline int traceDepth = 0;
    static inline int toneMapping = ToneMapping::ACES;
    static inline bool useReservoir = true;
    static inline int reservoirReuse = ReservoirReuse::Temporal;
    static inline bool accumulate = false;
};
struct State {
    static inline bool camChanged = true;
    static inline int looper = 0;
    static inline Scene* scene = nullptr;
};
namespace rsc {
inline void advanceLooper() {                           // the tail of every launcher (src/restir.cu:441-445)
#if SAMPLER_USE_SOBOL
    State::looper = (State::looper + 1) % SobolSampleNum;
#else
    State::looper++;
#endif
}
// DevScene::create's last step (src/scene.cpp:500-506).  A missing or short file leaves the rest of the table zero, as the
// reference's unchecked ifstream::read into a zero-filled vector does.
inline void uploadSampleSequence(rs_scene* devScene) {
#if SAMPLER_USE_SOBOL
    std::vector<uint32_t> sobolData((size_t)SobolSampleNum * SobolSampleDim, 0u);
    std::ifstream sobolFile("sobol_10k_200.bin", std::ios::in | std::ios::binary);
    sobolFile.read(reinterpret_cast<char*>(sobolData.data()), (std::streamsize)(sobolData.size() * sizeof(uint32_t)));
    check(rs_scene_set_sample_sequence(devScene, sobolData.data(), SobolSampleNum, SobolSampleDim), "Dev Scene");
#else
    (void)devScene;
#endif
}
}  // namespace rsc

// ---- src/sceneStructs.h:22-126 -------------------------------------------------------------------
struct Camera : rs_camera {
    void update() { rsc::check(rs_camera_update(this), "Camera::update"); }
};
static_assert(sizeof(Camera) == 196, "Camera layout (src/sceneStructs.h:104-117)");

using Material = rs_material;                           // src/material.h:113-268 (data members)
using DevScene = rs_scene;                              // src/scene.h:64-481

// ---- src/scene.h:483-531 (only what the path needs) -----------------------------------------------
struct RenderState {                                    // src/sceneStructs.h:128-133 (the members main.cpp reads)
    unsigned int iterations = 0;
    std::string imageName;
};

struct Scene {
    Camera camera{};
    RenderState state;
    DevScene* devScene = nullptr;
    rs_scene_file* file = nullptr;                      // what Scene(filename) parsed (materials, instances, textures, camera)

    Scene() = default;
    // Scene::Scene(filename) (src/scene.cpp:96-131): materials, objects (OBJ, baked per instance), camera, environment map.
    // Image files must be PNG, JPEG, TGA, BMP, binary PPM or Radiance HDR; see include/restir_hip.h "scene files".
    explicit Scene(const std::string& filename) {
        rsc::check(rs_scene_file_load(filename.c_str(), &file), "Scene loading");
        rs_scene_file_view v;
        rsc::check(rs_scene_file_get(file, &v), "Scene loading");
        static_cast<rs_camera&>(camera) = v.camera;
        state.iterations = (unsigned int)v.iterations;
        state.imageName = v.imageName;
        Settings::traceDepth = v.traceDepth;            // loadCamera's "Depth" line (src/scene.cpp:321-323)
    }
    ~Scene() { rs_scene_file_free(file); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    // buildDevData (src/scene.cpp:159-215) of the scene the constructor parsed
    void buildDevData() {
        rs_scene_file_view v;
        rsc::check(rs_scene_file_get(file, &v), "Dev Scene");
        rsc::check(rs_scene_build_textured(v.numPrims, v.vertices, v.normals, v.texcoords, v.materialIds, v.numMaterials, v.materials,
                                           v.numTextures, v.textures, v.envMapTexId, &devScene), "Dev Scene");
        rsc::uploadSampleSequence(devScene);
    }
    // buildDevData on an already baked, de-indexed triangle soup
    void buildDevData(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                      const int* materialIds, int numMaterials, const Material* materials) {
        rsc::check(rs_scene_build(numPrims, vertices, normals, texcoords, materialIds, numMaterials, materials, &devScene), "Dev Scene");
        rsc::uploadSampleSequence(devScene);
    }
    // the same with the decoded images of Scene::textures and Scene::envMapTexId (src/scene.h:518-521;
    // createLightSampler's environment-map sampler, src/scene.cpp:136-152, is built by the library)
    void buildDevData(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                      const int* materialIds, int numMaterials, const Material* materials,
                      int numTextures, const rs_texture* textures, int envMapTexId) {
        rsc::check(rs_scene_build_textured(numPrims, vertices, normals, texcoords, materialIds, numMaterials, materials,
                                           numTextures, textures, envMapTexId, &devScene), "Dev Scene");
        rsc::uploadSampleSequence(devScene);
    }
    void clear() { rs_scene_destroy(devScene); devScene = nullptr; }     // src/scene.cpp:217-220
};

// ---- src/gbuffer.h:15-59 ---------------------------------------------------------------------------
struct GBuffer {
    rs_gbuffer* impl = nullptr;
    int width = 0, height = 0;
    void create(int w, int h) { width = w; height = h; rsc::check(rs_gbuffer_create(w, h, &impl), "GBuffer::create"); }
    void destroy() { rs_gbuffer_destroy(impl); impl = nullptr; }
    void render(DevScene* scene, const Camera& cam) { rsc::check(rs_gbuffer_render(impl, scene, &cam), "renderGBuffer"); }
    void update(const Camera& cam) { rsc::check(rs_gbuffer_update(impl, &cam), "GBuffer::update"); }
    rs_gbuffer_view planes() const { rs_gbuffer_view v; rsc::check(rs_gbuffer_get_view(impl, &v), "GBuffer"); return v; }
};

// ---- src/restir.h:128-133 ---------------------------------------------------------------------------
namespace rsc { inline rs_restir* g_restir = nullptr; }

inline void ReSTIRInit() {                               // src/restir.cu:478-504
    const Camera& cam = State::scene->camera;
    rsc::check(rs_restir_init(cam.resolution[0], cam.resolution[1], &rsc::g_restir), "ReSTIRInit");
}
inline void ReSTIRFree() { rs_restir_free(rsc::g_restir); rsc::g_restir = nullptr; }
inline void ReSTIRReset() { rsc::check(rs_restir_reset(rsc::g_restir), "ReSTIRReset"); }
inline void ReSTIRDirect(rsc::vec3* devDirectIllum, int iter, const GBuffer& gBuffer) {   // src/restir.cu:418-446
    rsc::check(rs_restir_direct(rsc::g_restir, State::scene->devScene, &State::scene->camera, gBuffer.impl,
                                reinterpret_cast<float*>(devDirectIllum), iter, State::looper, Settings::reservoirReuse),
               "ReSTIR Direct");
    rsc::advanceLooper();
}

// ---- src/pathtrace.h:8-16 ----------------------------------------------------------------------------
inline void pathTraceInit() { rsc::check(rs_path_trace_init(), "pathTraceInit"); }
inline void pathTraceFree() { rs_path_trace_free(); }
inline void pathTraceDirect(rsc::vec3* devDirectIllum, int iter) {                        // src/pathtrace.cu:457-476
    rsc::check(rs_path_trace_direct(State::scene->devScene, &State::scene->camera, reinterpret_cast<float*>(devDirectIllum),
                                    iter, State::looper, nullptr), "pathTrace");
    rsc::advanceLooper();
}
// multi-bounce kernels (src/pathtrace.h:12-16, src/restir.h:133); Settings::traceDepth is the reference's global
inline void pathTrace(rsc::vec3* devDirectIllum, rsc::vec3* devIndirectIllum, int iter) {  // src/pathtrace.cu:434-455
    rsc::check(rs_path_trace(State::scene->devScene, &State::scene->camera, reinterpret_cast<float*>(devDirectIllum),
                             reinterpret_cast<float*>(devIndirectIllum), iter, State::looper, Settings::traceDepth, nullptr), "pathTrace");
    rsc::advanceLooper();
}
inline void pathTraceIndirect(rsc::vec3* devIndirectIllum, int iter) {                     // src/pathtrace.cu:478-497
    rsc::check(rs_path_trace_indirect(State::scene->devScene, &State::scene->camera, reinterpret_cast<float*>(devIndirectIllum),
                                      iter, State::looper, Settings::traceDepth, nullptr), "pathTrace");
    rsc::advanceLooper();
}
inline void ReSTIRIndirect(rsc::vec3* devIndirectIllum, int iter, const GBuffer& gBuffer) {   // src/restir.cu:448-476
    rsc::check(rs_restir_indirect(rsc::g_restir, State::scene->devScene, &State::scene->camera, gBuffer.impl,
                                  reinterpret_cast<float*>(devIndirectIllum), iter, State::looper, Settings::reservoirReuse,
                                  Settings::traceDepth, nullptr), "ReSTIR Indirect");
    rsc::advanceLooper();
}
struct uchar4_t { unsigned char x, y, z, w; };
inline void copyImageToPBO(void* devPBO, rsc::vec3* devImage, int width, int height, int toneMapping, float scale = 1.f) {
    rsc::check(rs_copy_image_to_pbo(devPBO, reinterpret_cast<const float*>(devImage), width, height, toneMapping, scale), "copyImageToPBO");
}
// the debug-view overloads (src/pathtrace.h:9-11); vec2 is two packed floats
namespace rsc { struct vec2 { float x, y; }; }
inline void copyImageToPBO(void* devPBO, rsc::vec2* devImage, int width, int height) {
    rsc::check(rs_copy_image2_to_pbo(devPBO, reinterpret_cast<const float*>(devImage), width, height), "copyImageToPBO");
}
inline void copyImageToPBO(void* devPBO, float* devImage, int width, int height) {
    rsc::check(rs_copy_imagef_to_pbo(devPBO, devImage, width, height), "copyImageToPBO");
}
inline void copyImageToPBO(void* devPBO, int* devImage, int width, int height) {
    rsc::check(rs_copy_imagei_to_pbo(devPBO, devImage, width, height), "copyImageToPBO");
}

// ---- the viewer's pixel-buffer object and screenshot (src/preview.cpp:88,112-133; src/main.cpp:105-144,176-181) -------------------
// The reference addresses its PBO by the GLuint alone; HIP's interop hands back a resource at registration, kept here per GLuint.
namespace rsc {
inline std::map<unsigned, rs_pbo*>& pbos() { static std::map<unsigned, rs_pbo*> m; return m; }
}
inline void cudaGLSetGLDevice(int device) { rsc::check(rs_init(device), "cudaGLSetGLDevice"); }
inline void cudaGLRegisterBufferObject(unsigned pbo) { rsc::check(rs_pbo_register(pbo, &rsc::pbos()[pbo]), "cudaGLRegisterBufferObject"); }
inline void cudaGLMapBufferObject(void** devPtr, unsigned pbo) { rsc::check(rs_pbo_map(rsc::pbos()[pbo], devPtr, nullptr), "cudaGLMapBufferObject"); }
inline void cudaGLUnmapBufferObject(unsigned pbo) { rsc::check(rs_pbo_unmap(rsc::pbos()[pbo]), "cudaGLUnmapBufferObject"); }
inline void cudaGLUnregisterBufferObject(unsigned pbo) { rsc::check(rs_pbo_unregister(rsc::pbos()[pbo]), "cudaGLUnregisterBufferObject"); rsc::pbos().erase(pbo); }
// saveImage(jpg) (src/main.cpp:105-144) for a device image: "<imageName>.<time>.<samples>samp.png" or ".jpg" (Image::saveJPG, quality 90)
inline std::string saveImage(const std::string& imageName, const std::string& timeString, int samples, rsc::vec3* devImage, int width, int height, int toneMapping, bool jpg = false) {
    const std::string filename = imageName + "." + timeString + "." + std::to_string(samples) + (jpg ? "samp.jpg" : "samp.png");
    rsc::check((jpg ? rs_save_image_jpg : rs_save_image)(filename.c_str(), reinterpret_cast<const float*>(devImage), width, height, toneMapping), "saveImage");
    return filename;
}

// ---- src/denoiser.h:33-43,72-74 -------------------------------------------------------------------------
// src/denoiser.h:15-31: the public members the viewer edits (src/preview.cpp:262-286); the kernels live behind rs_eaw / rs_svgf
struct EAWaveletFilter {
    EAWaveletFilter() = default;
    EAWaveletFilter(int width, int height, float sigLumin, float sigNormal, float sigDepth) :
        sigLumin(sigLumin), sigNormal(sigNormal), sigDepth(sigDepth), width(width), height(height) {}
    float sigLumin = 0.f;
    float sigNormal = 0.f;
    float sigDepth = 0.f;
    int width = 0;
    int height = 0;
};
struct LeveledEAWFilter {
    rs_eaw* impl = nullptr;
    EAWaveletFilter waveletFilter;
    int level = 0;
    void create(int width, int height, int lv) {
        level = lv;
        waveletFilter = EAWaveletFilter(width, height, 64.f, .2f, 1.f);                     // src/denoiser.cu:455
        rsc::check(rs_eaw_create(width, height, lv, &impl), "EAW create");
    }
    void destroy() { rs_eaw_destroy(impl); impl = nullptr; }
    void filter(rsc::vec3*& devColorOut, rsc::vec3* devColorIn, const GBuffer& gBuffer, const Camera& cam) {
        // the members may have been edited since the last call (ImGui sliders write them directly)
        rsc::check(rs_eaw_set_params(impl, waveletFilter.sigLumin, waveletFilter.sigNormal, waveletFilter.sigDepth, level), "EAW Filter");
        float* out = reinterpret_cast<float*>(devColorOut);
        rsc::check(rs_eaw_filter(impl, &out, reinterpret_cast<const float*>(devColorIn), gBuffer.impl, &cam), "EAW Filter");
        devColorOut = reinterpret_cast<rsc::vec3*>(out);
    }
};
// src/denoiser.h:45-70 (filter / nextFrame; the three sub-steps are internal to rs_svgf_filter)
struct SpatioTemporalFilter {
    rs_svgf* impl = nullptr;
    EAWaveletFilter waveletFilter;
    int level = 0;
    void create(int width, int height, int lv) {
        level = lv;
        waveletFilter = EAWaveletFilter(width, height, 4.f, 128.f, 1.f);                    // src/denoiser.cu:488
        rsc::check(rs_svgf_create(width, height, lv, &impl), "SVGF create");
    }
    void destroy() { rs_svgf_destroy(impl); impl = nullptr; }
    void filter(rsc::vec3*& devColorOut, rsc::vec3* devColorIn, const GBuffer& gBuffer, const Camera& cam) {
        rsc::check(rs_svgf_set_params(impl, waveletFilter.sigLumin, waveletFilter.sigNormal, waveletFilter.sigDepth, level), "SpatioTemporalFilter::filter");
        float* out = reinterpret_cast<float*>(devColorOut);
        rsc::check(rs_svgf_filter(impl, &out, reinterpret_cast<const float*>(devColorIn), gBuffer.impl, &cam), "SpatioTemporalFilter::filter");
        devColorOut = reinterpret_cast<rsc::vec3*>(out);
    }
    void nextFrame() { rsc::check(rs_svgf_next_frame(impl), "SpatioTemporalFilter::nextFrame"); }
};
inline void modulateAlbedo(rsc::vec3* devImage, const GBuffer& gBuffer) {
    rsc::check(rs_modulate_albedo(reinterpret_cast<float*>(devImage), gBuffer.impl), "modulate");
}
inline void addImage(rsc::vec3* devImage, rsc::vec3* devIn, int width, int height) {
    rsc::check(rs_add_image(reinterpret_cast<float*>(devImage), reinterpret_cast<const float*>(devIn), width, height), "addImage");
}
inline void addImage(rsc::vec3* devOut, rsc::vec3* devIn1, rsc::vec3* devIn2, int width, int height) {
    rsc::check(rs_add_image3(reinterpret_cast<float*>(devOut), reinterpret_cast<const float*>(devIn1),
                             reinterpret_cast<const float*>(devIn2), width, height), "addImage");
}
