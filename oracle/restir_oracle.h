/*
 * restir_oracle.h -- CPU restatement of the ReSTIR-DI hot path of HummaWhite/ReSTIR.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (restir_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the reported CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pinned bit-for-bit against the reference's own code compiled here (oracle/_ref, built by
 *     oracle/Makefile from /root/reference/src with g++): intersectTriangle, AABB::intersect,
 *     BVHBuilder::build/buildMTBVH, Camera::{update,sample,getRasterCoord,getPosition},
 *     Material::BSDF / sample / pdf, Math::* helpers (incl. toSphere / toPlane / localToWorld), linearSample,
 *     tone-map operators;
 *   - pinned against the third-party dependency itself (rocThrust 2.8.5 minstd_rand +
 *     uniform_real_distribution compiled by hipcc host-only): the RNG stream;
 *   - PARITY UNPINNED (restated from source text only; the reference's scene.h / sampler.h /
 *     restir.h / *.cu cannot be compiled in this image without CUDA Thrust + nvcc):
 *     DevScene::intersect / testOcclusion loops, sampleDirectLight*, DiscreteSampler1D,
 *     Reservoir<>, and the kernel glue of restir.cu / gbuffer.cu / pathtrace.cu / denoiser.cu (DI, GI, path tracing,
 *     EAW, SVGF).
 *
 * All arithmetic is FP32, compiled with -ffp-contract=off; operation order follows GLM 0.9.6.3.
 * Layout-compatible with include/restir_hip.h (rs_material / rs_camera / rs_reservoir).
 */
#ifndef RESTIR_ORACLE_H
#define RESTIR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/material.h:258-267 (44 bytes) */
typedef struct orc_material {
    int   type;              /* 0 Lambertian, 1 MetallicWorkflow, 2 Dielectric, 3 Disney, 4 Light */
    float baseColor[3];
    float metallic;
    float roughness;
    float ior;
    int   baseColorMapId;    /* -1 = none (NullTextureId), -2 = procedural (ProceduralTexId), else texture index */
    int   metallicMapId;
    int   roughnessMapId;
    int   normalMapId;
} orc_material;

/* reference: src/sceneStructs.h:104-117 (196 bytes); mat3/mat4 column-major */
typedef struct orc_camera {
    int   resolution[2];
    float position[3];
    float rotation[3];
    float view[3];
    float up[3];
    float right[3];
    float fov[2];
    float pixelLength[2];
    float rotationMatInv[9];
    float viewProjection[16];
    float lensRadius;
    float focalDist;
    float tanFovY;
} orc_camera;

/* reference: src/restir.h:7-11,29-117 -- Reservoir<DirectLiSample> (36 bytes) */
typedef struct orc_reservoir {
    float Li[3];
    float wi[3];
    float dist;
    int   numSamples;
    float weight;
} orc_reservoir;

/* reference: src/image.h:76-97 DevTextureObj -- linear RGB float texels, row-major */
typedef struct orc_texture {
    int          width, height;
    const float* data;             /* 3 floats / texel */
} orc_texture;

/* reference: src/restir.h:13-27,114-116 -- Reservoir<IndirectLiSample> (68 bytes) */
typedef struct orc_indirect_reservoir {
    float Lo[3];
    float xv[3], nv[3];
    float xs[3], ns[3];
    int   numSamples;
    float weight;
} orc_indirect_reservoir;

/* Host-memory image of DevScene (src/scene.h:461-480). */
typedef struct orc_scene {
    int           numPrims;
    const float*  vertices;        /* 9 floats / prim */
    const float*  normals;         /* 9 floats / prim */
    const float*  texcoords;       /* 6 floats / prim */
    const int*    materialIds;     /* 1 / prim */
    int           numMaterials;
    const orc_material* materials;
    int           bvhSize;
    const float*  boundingBoxes;   /* 6 floats / node: pMin, pMax */
    const int*    bvhNodes[6];     /* 3 ints / node: primitiveId, boundingBoxId, nextNodeIfMiss */
    int           numLights;
    const int*    lightPrimIds;
    const float*  lightUnitRadiance; /* 3 / light */
    const float*  lightProb;       /* alias table: BinomialDistrib.prob   */
    const int*    lightFailId;     /* alias table: BinomialDistrib.failId */
    float         sumLightPowerInv;
    /* textures and environment map (scene.h:78-99,358-392; scene.cpp:136-152,479-498).  With an environment
     * map the light sampler has one more entry than there are light primitives: numLights counts it, and
     * lightPrimIds / lightUnitRadiance hold numLights-1 records (scene.cpp:151, scene.h:400-403). */
    int           numTextures;
    const orc_texture* textures;
    int           envMapTexId;         /* -1 = none */
    int           envMapSamplerLength; /* width*height of the environment map, 0 = none */
    const float*  envMapProb;          /* alias table over lum(texel)*sin(theta) */
    const int*    envMapFailId;
    /* DevScene::sampleSequence (scene.h:480, scene.cpp:500-506): the Sobol table, SobolSampleNum x SobolSampleDim uint32
     * (sampler.h:10-11); NULL selects the default thrust engine (SAMPLER_USE_SOBOL false, common.h:4). */
    const uint32_t* sampleSequence;
} orc_scene;

/* Host-memory image of GBuffer (src/gbuffer.h:41-58). */
typedef struct orc_gbuffer {
    float* albedo;       /* 3 / px */
    int*   motion;
    float* normal[2];    /* 3 / px */
    int*   primId[2];
    float* depth[2];
    int    frameIdx;
    orc_camera lastCamera;
    int    width, height;
} orc_gbuffer;

/* ---- function-level entry points (vectorised over n) ------------------------------ */

/* src/intersections.h:17-54.  rays: 6 floats (origin, direction); tris: 9 floats. */
void orc_intersect_triangle(int n, const float* rays, const float* tris,
                            int* hit, float* bary, float* dist);
/* src/bvh.h:85-157.  boxes: 6 floats (pMin, pMax). */
void orc_aabb_intersect(int n, const float* rays, const float* boxes, int* hit, float* tMin);
/* src/mathUtil.h:190-198 */
void orc_utilhash(int n, const uint32_t* in, uint32_t* out);
/* src/sampler.h:41-48 + thrust minstd_rand: m draws of sample1D for each (looper,index,dim) */
void orc_rng_stream(int n, const int* looper, const int* index, const int* dim, int m, float* out);
/* src/sampler.h:9-36 (SAMPLER_USE_SOBOL): m draws of Sampler::sample for each (looper, index, dim) over `data` */
void orc_sobol_stream(const uint32_t* data, int n, const int* looper, const int* index, const int* dim, int m, float* out);
/* src/material.h:218-228 */
void orc_bsdf(int n, const orc_material* mats, const float* nrm, const float* wo, const float* wi,
              float* out);
/* src/sceneStructs.h:69-86.  r: 4 floats / ray; out rays 6 floats */
void orc_camera_sample(const orc_camera* cam, int n, const int* xy, const float* r, float* rays);
/* src/sceneStructs.h:23-46 */
void orc_camera_raster_coord(const orc_camera* cam, int n, const float* pos, int* xy);
/* src/sceneStructs.h:48-64 */
void orc_camera_position(const orc_camera* cam, int n, const int* xy, const float* dist, float* pos);
/* src/sceneStructs.h:88-102 (host).  viewProjection is left untouched (not on the path). */
void orc_camera_update(orc_camera* cam);
/* src/mathUtil.h:94-100,86-92,182-185,128-132 */
void orc_sample_triangle_uniform(int n, const float* tris, const float* ruv, float* out);
void orc_to_concentric_disk(int n, const float* xy, float* out);
/* mathUtil.h:86-92,119-123,182-185: area, normal, pdfAreaToSolidAngle(lum(v1), x, v0, normal) */
void orc_triangle_misc(int n, const float* tris, const float* x, float* area, float* normal, float* pdf);
/* raw-seed RNG stream: thrust::default_random_engine(seed) then m draws of uniform_real<float>(0,1) */
void orc_rng_stream_raw(int n, const int* seeds, int m, float* out);
/* src/mathUtil.h:102-117: mode 0 none, 1 filmic, 2 ACES; then correctGamma; float output */
void orc_tonemap(int n, const float* in, int mode, float* out);

/* How the libm calls of __device__ code are evaluated (sin/cos/atan2 in toSphere / toPlane / proceduralTexture
 * / the spatial tap; the reference's CUDA build uses libdevice, which cannot be reproduced here):
 *   0  glibc sinf/cosf/atan2f -- what a host build of the reference computes (default)
 *   1  the correctly rounded value (evaluated in double, rounded once) -- what librestir_hip computes wherever a
 *      libm result matters (spatial tap position, environment map, procedural texture, BSDF sampling).
 * In mode 1 the device and the oracle agree bit for bit everywhere; in mode 0 a pixel differs where glibc's result
 * is one ulp off the correctly rounded one AND that ulp changes an integer (spatial tap: ~2e-8 of the pixels) or a
 * path. */
void orc_set_libm_mode(int correctlyRounded);

/* src/image.h:41-75 linearSample on n uv pairs */
void orc_linear_sample(const orc_texture* tex, int n, const float* uv, float* out);
/* src/mathUtil.h:134-144: toSphere (2 -> 3), toPlane (3 -> 2) */
void orc_to_sphere(int n, const float* uv, float* dir);
void orc_to_plane(int n, const float* dir, float* uv);
/* src/material.h:230-256: Material::sample (r3 = sample3D; an Invalid sample reports dir = bsdf = 0, pdf = 0) and ::pdf */
void orc_material_sample(int n, const orc_material* mats, const float* nrm, const float* wo, const float* r3,
                         float* dir, float* bsdf, float* pdf, uint32_t* type);
void orc_material_pdf(int n, const orc_material* mats, const float* nrm, const float* wo, const float* wi, float* pdf);
/* src/mathUtil.h:146-155 localToWorld(n, v) */
void orc_local_to_world(int n, const float* nrm, const float* v, float* out);
/* src/scene.h:68-76 proceduralTexture (value replicated to 3 channels) */
void orc_procedural_texture(int n, const float* uv, float* out);

/* src/mathUtil.cpp:13-20 buildTransformationMatrix (column-major float[16]) and the instance baking of
 * Scene::buildDevData (src/scene.cpp:161-171, normalMat from :276-278): host code, glibc cosf / sinf */
void orc_build_transformation_matrix(const float* t, const float* r, const float* s, float* out16);
void orc_bake_instance(const float* t, const float* r, const float* s, int n, const float* vertsIn, const float* normalsIn,
                       float* vertsOut, float* normalsOut);
/* tan / atan of Scene::loadCamera (src/scene.cpp:344-348) as the host libm evaluates them in float */
float orc_tanf(float x);
float orc_atanf(float x);

/* ---- host scene build -------------------------------------------------------------- */

/* src/scene.cpp:139-146: pdf[i*w+j] = lum(texel) * sin((.5f + i) / h * Pi) -- host code, glibc sinf */
void orc_envmap_pdf(int width, int height, const float* data, float* pdf);


/* src/bvh.cpp:10-202.  boxes: 6*(2n-1) floats; nodes: 6 arrays of 3*(2n-1) ints. Returns BVHSize. */
int  orc_bvh_build(int numPrims, const float* vertices, float* boxes, int* nodes[6]);
/* src/sampler.h:79-121 */
void orc_alias_build(int n, const float* values, float* prob, int* failId, float* sumAll);
/* src/scene.cpp:159-190 light table part.  Returns number of lights. */
int  orc_light_table(int numPrims, const float* vertices, const int* materialIds,
                     const orc_material* mats, int* lightPrimIds, float* lightUnitRadiance,
                     float* lightPower);

/* ---- scene services ---------------------------------------------------------------- */

/* src/scene.h:245-284.  out: primId, matId, pos[3], norm[3], uv[2] (only valid when primId>=0) */
void orc_intersect(const orc_scene* s, int n, const float* rays, int* primId, int* matId,
                   float* pos, float* norm, float* uv);
/* src/scene.h:286-316.  seg: 6 floats (x, y) */
void orc_test_occlusion(const orc_scene* s, int n, const float* seg, int* occluded);
/* src/scene.h:394-425.  r: 4 floats.  out pdf, Li[3], wi[3], dist */
void orc_sample_direct_light_nv(const orc_scene* s, int n, const float* pos, const float* r,
                                float* pdf, float* Li, float* wi, float* dist);

/* ---- frame-level passes ------------------------------------------------------------ */

/* src/gbuffer.cu:3-73 (writes the frameIdx planes + albedo + motion). rows [y0,y1). */
void orc_gbuffer_render(const orc_scene* s, const orc_camera* cam, orc_gbuffer* g, int y0, int y1);
/* src/gbuffer.cu:75-78 */
void orc_gbuffer_update(orc_gbuffer* g, const orc_camera* cam);

/* src/pathtrace.cu:279-328 */
void orc_pt_direct(const orc_scene* s, const orc_camera* cam, float* directIllum,
                   int looper, int iter, unsigned long long* rays);

/* src/pathtrace.cu:156-277 singleKernelPT (pathTrace) and :330-432 PTIndirectKernel (pathTraceIndirect) */
void orc_path_trace(const orc_scene* s, const orc_camera* cam, float* directIllum, float* indirectIllum,
                    int looper, int iter, int maxDepth, unsigned long long* rays);
void orc_pt_indirect(const orc_scene* s, const orc_camera* cam, float* indirectIllum,
                     int looper, int iter, int maxDepth, unsigned long long* rays);
/* src/restir.cu:233-416 ReSTIRIndirectKernel: temporalReservoir is written, lastTemporalReservoir read (the launcher
 * swaps them afterwards, :463).  reuse bit0 = temporal. */
void orc_restir_indirect(const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g, float* indirectIllum,
                         orc_indirect_reservoir* temporalReservoir, const orc_indirect_reservoir* lastTemporalReservoir,
                         int looper, int iter, int maxDepth, int first, int reuse, unsigned long long* rays);

/* src/restir.cu:111-231 with the two-phase contract of SURVEY.md Q1:
 * phase A (primary, RIS, shadow, temporal, publish) for all pixels, grid barrier,
 * phase B (spatial, shade, accumulate).  reuse: bit0 temporal, bit1 spatial.
 * rays (optional): number of BVH walks performed (intersect + testOcclusion). */
void orc_restir_direct(const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                       float* directIllum, orc_reservoir* reservoirOut,
                       const orc_reservoir* reservoirIn, orc_reservoir* reservoirTemp,
                       int looper, int iter, int first, int reuse, unsigned long long* rays);

/* The same pass split at the barrier, on a row range, with the per-pixel carried state (RNG, surface,
 * post-temporal reservoir) kept in an opaque buffer: the decomposition the multi-GPU row-strip tiling
 * uses (phase A on own rows, exchange of reservoirTemp halo rows, phase B on own rows). */
void* orc_restir_state_create(int width, int height);
void  orc_restir_state_destroy(void* state);
void  orc_restir_phase_a(void* state, const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                         orc_reservoir* reservoirOut, const orc_reservoir* reservoirIn,
                         orc_reservoir* reservoirTemp, int looper, int first, int reuse,
                         int y0, int y1, unsigned long long* rays);
void  orc_restir_phase_b(void* state, const orc_scene* s, const orc_camera* cam, const orc_gbuffer* g,
                         float* directIllum, const orc_reservoir* reservoirTemp, int iter, int reuse,
                         int y0, int y1);

/* src/pathtrace.cu:30-56: rgba8 out (4 bytes / px, a = 0) */
void orc_send_image_to_pbo(int w, int h, const float* image, int toneMapping, float scale,
                           unsigned char* rgba);

/* src/pathtrace.cu:58-106: the vec2 (kind 0) / float (1) / int (2) overloads of sendImageToPBO */
void orc_send_debug_to_pbo(int w, int h, const void* image, int kind, unsigned char* rgba);

/* src/denoiser.cu:64-134: one EAW level */
void orc_eaw_level(const orc_gbuffer* g, const orc_camera* cam, const float* colorIn,
                   float* colorOut, float sigDepth, float sigNormal, float sigLumin, int level);
/* src/denoiser.cu:463-477: 5 levels, sigma 64 / .2 / 1; result in out, tmp is scratch. Returns
 * pointer (out or tmp) that holds the final image, mirroring the pointer swap of the reference. */
float* orc_eaw_filter(const orc_gbuffer* g, const orc_camera* cam, const float* colorIn,
                      float* out, float* tmp);
/* src/denoiser.cu:136-216,250-371,479-568: SpatioTemporalFilter (SVGF).  The state object also holds the caller's
 * `devColorOut` buffer of the reference (it is swapped with the filter's buffers by every filter() call);
 * orc_svgf_filter returns the buffer that holds the result of this call. */
void* orc_svgf_create(int width, int height);
void  orc_svgf_destroy(void* f);
const float* orc_svgf_filter(void* f, const float* colorIn, const orc_gbuffer* g, const orc_camera* cam);
void  orc_svgf_next_frame(void* f);
const float* orc_svgf_variance(void* f);
const float* orc_svgf_accum_color(void* f);
const float* orc_svgf_accum_moment(void* f);

/* src/denoiser.cu:218-248 */
void orc_modulate(int w, int h, float* image, const float* albedo);
void orc_add(int w, int h, float* image, const float* in);
void orc_add3(int w, int h, float* out, const float* in1, const float* in2);

#ifdef __cplusplus
}
#endif
#endif
