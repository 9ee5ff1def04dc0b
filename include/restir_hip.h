/*
 * restir_hip.h -- C ABI of librestir_hip.so: the MI355X (gfx950) implementation of the ReSTIR-DI
 * per-pixel pipeline and MTBVH traversal of HummaWhite/ReSTIR, behind the reference's own render-pass
 * entry points.  Plain pointers and sizes only; every image / G-buffer pointer is DEVICE memory
 * (hipMalloc), row-major y*W+x, tightly packed float[3] exactly as the reference's glm::vec3 arrays.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository).  The C++ header restir_amd/host/restir_compat.h re-declares the reference's names
 * (GBuffer::render, ReSTIRDirect, copyImageToPBO, ...) on top of this ABI; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Return convention: 0 = success, otherwise a hipError_t value (or RS_ERR_*); rs_last_error() gives
 * the message.  (The reference prints and exit()s from checkCUDAError, src/cudaUtil.h:13-31; the
 * compat header reproduces that on top of these codes.)
 *
 * Implicit inputs of the reference launchers (State::looper, Settings::reservoirReuse, the
 * file-static ReSTIRFirstFrame; src/restir.cu:418-446) are explicit parameters / object state here.
 */
#ifndef RESTIR_HIP_H
#define RESTIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ERR_INVALID_ARGUMENT 10001
#define RS_ERR_UNSUPPORTED      10002   /* e.g. an image or mesh format the scene-file reader does not decode */
#define RS_ERR_INTERNAL         10003   /* a self-check of the library failed (rs_ordered_bvh_host_check) */

/* src/material.h:258-267 -- identical 44-byte layout */
typedef struct rs_material {
    int   type;            /* Material::Type: 0 Lambertian 1 MetallicWorkflow 2 Dielectric 3 Disney 4 Light */
    float baseColor[3];
    float metallic;
    float roughness;
    float ior;
    int   baseColorMapId;  /* -1 NullTextureId, -2 ProceduralTexId (src/material.h:11-13), else index into rs_scene_desc::textures */
    int   metallicMapId;
    int   roughnessMapId;
    int   normalMapId;
} rs_material;

/* src/sceneStructs.h:104-117 -- identical 196-byte layout (glm mat3/mat4 are column-major) */
typedef struct rs_camera {
    int   resolution[2];
    float position[3];
    float rotation[3];
    float view[3];
    float up[3];
    float right[3];
    float fov[2];
    float pixelLength[2];
    float rotationMatInv[9];
    float viewProjection[16];   /* not read by any kernel on this path */
    float lensRadius;
    float focalDist;
    float tanFovY;
} rs_camera;

/* src/restir.h:7-11,114-116 -- Reservoir<DirectLiSample>, identical 36-byte layout */
typedef struct rs_reservoir {
    float Li[3];
    float wi[3];
    float dist;
    int   numSamples;
    float weight;
} rs_reservoir;

/* src/image.h:7-38,76-97 -- a decoded image: linear-RGB float texels, row-major (what Image / DevTextureObj hold;
 * decoding files stays with the caller, as it does with stb_image in the reference, src/image.cpp:14-31) */
typedef struct rs_texture {
    int          width, height;
    const float* data;            /* 3 floats / texel */
} rs_texture;

/* src/restir.h:13-27,114-116 -- Reservoir<IndirectLiSample>, identical 68-byte layout */
typedef struct rs_indirect_reservoir {
    float Lo[3];
    float xv[3], nv[3];
    float xs[3], ns[3];
    int   numSamples;
    float weight;
} rs_indirect_reservoir;

/* Host arrays that define a device scene = the inputs of DevScene::create (src/scene.cpp:435-509),
 * in the layouts Scene::buildDevData leaves them (src/scene.cpp:159-215). */
typedef struct rs_scene_desc {
    int                numPrims;
    const float*       vertices;          /* 9 floats / triangle  (meshData.vertices)  */
    const float*       normals;           /* 9 floats / triangle  (meshData.normals)   */
    const float*       texcoords;         /* 6 floats / triangle  (meshData.texcoords), may be NULL */
    const int*         materialIds;       /* 1 / triangle */
    int                numMaterials;
    const rs_material* materials;
    int                bvhSize;           /* 2*numPrims-1 */
    const float*       boundingBoxes;     /* 6 floats / node: AABB pMin,pMax (src/bvh.h:159-160) */
    const int*         bvhNodes[6];       /* 3 ints / node: MTBVHNode (src/bvh.h:163-171), 6 orders */
    int                numLights;
    const int*         lightPrimIds;
    const float*       lightUnitRadiance; /* 3 / light */
    const float*       lightProb;         /* DiscreteSampler1D::binomDistribs[i].prob   */
    const int*         lightFailId;       /* DiscreteSampler1D::binomDistribs[i].failId */
    float              sumLightPower;     /* lightSampler.sumAll */
    /* Textures and environment map (src/scene.h:78-99,358-392; src/scene.cpp:136-152,479-498).  With an environment
     * map the light sampler has one more entry than there are light primitives (src/scene.cpp:151): numLights
     * counts it, lightProb/lightFailId have numLights entries, lightPrimIds/lightUnitRadiance numLights-1. */
    int                numTextures;
    const rs_texture*  textures;
    int                envMapTexId;        /* Scene::envMapTexId, -1 = none */
    const float*       envMapProb;         /* envMapSampler alias table, width*height entries (NULL without a map) */
    const int*         envMapFailId;
} rs_scene_desc;

typedef struct rs_scene   rs_scene;    /* = Scene::devScene / DevScene      (src/scene.h:64-481)  */
typedef struct rs_gbuffer rs_gbuffer;  /* = GBuffer                          (src/gbuffer.h:15-59) */
typedef struct rs_restir  rs_restir;   /* = module statics of restir.cu      (src/restir.cu:8-18)  */
typedef struct rs_eaw     rs_eaw;      /* = LeveledEAWFilter                 (src/denoiser.h:33-43) */
typedef struct rs_svgf    rs_svgf;     /* = SpatioTemporalFilter             (src/denoiser.h:45-70) */

/* Device pointers of a GBuffer's planes (src/gbuffer.h:41-58): [frameIdx] = this frame's, [frameIdx ^ 1] = last frame's.
 * The library keeps the planes in a ring of five sets (so that the renders of the next frames, which run ahead of the library
 * stream, never overwrite what this frame's temporal pass reads): the pointers are those of the current frame and change at every rs_gbuffer_update -- fetch the
 * view again after it. */
typedef struct rs_gbuffer_view {
    float* devAlbedo;        /* float[3] / px */
    int*   devMotion;
    float* devNormal[2];     /* float[3] / px */
    int*   devPrimId[2];     /* holds the MATERIAL id (lights -2, miss -1), src/gbuffer.cu:29-42 */
    float* devDepth[2];
    int    frameIdx;
    int    width, height;
} rs_gbuffer_view;

/* ---- library ------------------------------------------------------------------------- */
const char* rs_last_error(void);
/* Contexts.  The reference keeps its state in globals (State::scene, Settings::*, one device, the default stream); here that
 * state -- device, stream, synchronous / asynchronous launches, the internal streams of the asynchronous mode -- is a context.
 * Every object (scene, G-buffer, reservoirs, filters, strip driver) belongs to the context that was current in its host thread
 * when it was created, and calls on it run under that context whatever thread makes them, so several host threads, devices or
 * launch modes can use the library side by side.  A caller that never creates one uses the default context (the reference's
 * single-threaded viewer does); rs_init and the rs_set_* calls below configure the CURRENT context of the calling thread.
 * Objects handed to one call (scene + G-buffer + reservoirs) must belong to one context. */
typedef struct rs_context rs_context;
int  rs_context_create(int device, rs_context** ctx);
int  rs_context_destroy(rs_context* ctx);           /* after the objects created under it */
int  rs_context_set_current(rs_context* ctx);       /* for the calling thread; NULL = back to the default context */
/* Selects the HIP device of the default context (hipSetDevice). */
int  rs_init(int device);
/* All work is enqueued on this hipStream_t (NULL = default stream). */
int  rs_set_stream(void* hipStream);
/* The streams the library makes for itself in overlapped mode (three: the chains of consecutive frames and the render) are CHOSEN BY
 * MEASUREMENT when first needed, and again after rs_set_stream: whether four streams of a process run side by side depends on every
 * stream the process has made before (torch's pool, an ncclComm's own), so candidates of every priority level are timed next to the
 * caller's stream and the best three kept (about 25 ms, once, with the device idle; api_common.hip).  This call names the PREFERRED level
 * among those that measure equal: 2 (default) = automatic, above the caller's stream first; -1 high / 0 normal / 1 low first.  A caller
 * that runs a denoiser on the library stream every frame does 5 % better with 1 (config 5: 1.87 -> 1.78 ms).  Before the first frame. */
int  rs_set_internal_stream_priority(int level);
/* What the measurement chose: level, the chosen streams' calibration time and the fastest candidate's (us; 0 = plain streams in use). */
int  rs_internal_streams_info(int* priority, double* chosenUs, double* fastestUs);
/* Chooses them again at the next overlapped launch (waits for the library's work): call after the process has made streams of its own
 * since the library's first overlapped frame -- in particular after ncclCommInitRank, which makes several. */
int  rs_choose_internal_streams_again(void);
/* Makes the choice now instead of at the first overlapped launch (about 25 ms; waits for the caller's stream and the candidates, not for
 * the device) -- for a caller that wants no pause inside its first frame.  A choice is kept per caller stream and preference: rs_set_stream
 * back to a stream the library has measured next to takes that choice again.  While the caller's stream is being captured into a graph
 * nothing is measured: plain streams. */
int  rs_prepare_streams(void);
/* 1 (default): every entry point synchronises and checks errors before returning, like
 * checkCUDAError after each launch in the reference.  0: launches are only enqueued. */
int  rs_set_sync(int sync);
/* With rs_set_sync(0), frames overlap: GBuffer::render goes to an internal stream, the primary-ray -> RIS -> shadow-ray kernels of
 * ReSTIRDirect to two more, frames taking them in turn (a full-size frame's shadow rays stay on the library stream; a frame that
 * fills the chip less than three times over -- a strip -- renders in the same launch as its primary rays and has three such chains); each is
 * ordered only after the work that last used its buffers (G-buffer planes in a ring of five, per-frame surface planes in
 * four sets) and joined into the library stream where its results are first read (the temporal pass, the denoisers,
 * rs_gbuffer_get_view, rs_synchronize).  They then run next to the previous frames' temporal / spatial
 * passes.  Results are identical; output buffers are valid in library-stream order as before.  GBuffer::render can in addition
 * be deferred until ReSTIRDirect, which then walks the pixel-centre ray and the jittered ray of every pixel in one traversal:
 * faster or slower by a few percent depending on the scene and the launch size, so by default
 * every rs_restir measures the frame period both ways once per scene (frames 6..17 against 22..33) and keeps the faster.
 *   0 = everything on the library stream (RS_SIDE_STREAM=0 in the environment pre-sets this: the profiling scripts use it)
 *   1 = overlapped frames, the render always its own launch
 *   2 = overlapped frames, the render always deferred for launches of at least three rounds of the chip's wave slots
 *   3 = like 2 for launches of any size
 *   4 = overlapped frames, measured choice (the default) */
int  rs_set_side_stream(int enable);
/* The RIS pass keeps the light table (up to 1 024 lights) in LDS, one copy per 1 024-thread block; a launch of fewer than `pixels`
 * pixels reads it from global memory in 256-thread blocks instead, which spread evenly over the CUs (default 64 Ki pixels: a quarter
 * of the CUs with a block; rounds 2-4 had 384 Ki, which cost every rank of an 8-way split of 1080p 1-13 % of its frame period once the
 * library's streams no longer shared hardware queues).  Same results either way; 0 = always LDS. */
int  rs_set_ris_table_pixels(int pixels);
/* How the overlapped mode spreads a frame's kernels over the internal streams (a setting of the current context; every argument
 * -1 = keep).  chainStreams 1 / 2: the primary-ray -> RIS -> shadow-ray chains of all frames on one stream, or of alternating frames on
 * two (default 2).  smallChains 0 / 1: a launch below three rounds of the chip's wave slots (a strip) takes the fused render and rotates
 * its chains over three streams (default 1).  shadowOnMain 0 / 1 / 2: the shadow rays never / always / for launches of at least three
 * rounds of wave slots (default 2) on the library stream.  Results are identical in every setting (tools/soak_async.py); the defaults
 * are what measured fastest.
 * No entry point waits on the host in overlapped mode: the measurement of rs_set_side_stream's mode 4 polls its last time stamp
 * (hipEventQuery at the frame ends) and frames take two launches until it has arrived. */
int  rs_set_stream_plan(int chainStreams, int smallChains, int shadowOnMain);
/* A denoiser in the loop (src/main.cpp:160-170: LeveledEAWFilter::filter between ReSTIRDirect and copyImageToPBO) makes the library
 * stream both the one chain that links consecutive frames and the busiest stream (config 5: 98 % busy, the chain streams 51-57 %).
 * 1: with rs_set_sync(0), rs_eaw_filter / rs_strips_eaw_filter -- and an rs_copy_image_to_pbo / rs_strips_gather_begin that reads their
 * result -- are enqueued on a stream of the library, ordered after everything enqueued on the library stream so far, and run next to the
 * NEXT frame's temporal and spatial passes; the chains of the frames then take turns on two internal streams instead of three (four
 * streams with work in flight is what the device runs side by side).  Results are identical.  What changes is the ORDER CONTRACT of
 * those calls' outputs (the filtered image, the display buffer, the gathered image): they are ordered for the library stream by events,
 * which every library call that is handed one of those buffers waits for by itself, and which rs_join_denoise_stream() (enqueue-only)
 * or rs_synchronize() hand to the caller's own work on that stream -- a caller that reads the display buffer with its own kernels or
 * copies calls rs_join_denoise_stream() first (rs_pbo_unmap does).  0 (default): everything in library-stream order. */
int  rs_set_denoise_stream(int enable);
/* The library stream waits (on the device; the host does not) for everything enqueued on the denoise stream so far. */
int  rs_join_denoise_stream(void);
/* Closest-hit kernels (GBuffer::render, the primary rays of ReSTIRDirect): a tile whose packet walk visited at least `threshold`
 * nodes the last time the same launch ran is traced by four waves of 16 rays instead of one of 64 -- a launch that runs alone lasts
 * as long as its longest chain of node fetches; results do not depend on the grouping.  Applies where launches run one after the
 * other (synchronous mode, per-pass timing); 0 = off; negative = |threshold| for every launch, also with the frames overlapped
 * (measured slower there); default 768. */
int  rs_set_tile_split(int threshold);
int  rs_synchronize(void);

/* ---- host scene build: replaces Scene::buildDevData (src/scene.cpp:159-215) ----------- */
/* BVHBuilder::build + buildMTBVH (src/bvh.cpp:10-202).  boundingBoxes: 6*(2n-1) floats;
 * bvhNodes[k]: 3*(2n-1) ints each.  Returns BVHSize via *bvhSize. */
int  rs_build_bvh(int numPrims, const float* vertices, float* boundingBoxes, int* const bvhNodes[6], int* bvhSize);
/* light table of buildDevData (src/scene.cpp:161-190); arrays sized numPrims. */
int  rs_build_light_table(int numPrims, const float* vertices, const int* materialIds,
                          int numMaterials, const rs_material* materials, int* numLights,
                          int* lightPrimIds, float* lightUnitRadiance, float* lightPower);
/* DiscreteSampler1D<float> constructor (src/sampler.h:79-121). */
int  rs_build_alias_table(int n, const float* values, float* prob, int* failId, float* sumAll);
/* The whole of buildDevData on a baked triangle soup: light table, alias table, BVH, upload. */
int  rs_scene_build(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                    const int* materialIds, int numMaterials, const rs_material* materials,
                    rs_scene** scene);
/* The same with textures and an environment map: adds Scene::createLightSampler's environment-map sampler
 * (src/scene.cpp:136-152).  envMapTexId: index into textures or -1. */
int  rs_scene_build_textured(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                             const int* materialIds, int numMaterials, const rs_material* materials,
                             int numTextures, const rs_texture* textures, int envMapTexId, rs_scene** scene);
/* pdf of the environment-map sampler: lum(texel) * sin((.5 + row) / height * Pi) (src/scene.cpp:139-146). */
int  rs_build_envmap_pdf(int width, int height, const float* data, float* pdf);
/* DevScene::create (src/scene.cpp:435-509) from prebuilt host arrays. */
int  rs_scene_create(const rs_scene_desc* desc, rs_scene** scene);
/* Host copies of the arrays the scene was created from (valid until rs_scene_destroy). */
int  rs_scene_host_desc(const rs_scene* scene, rs_scene_desc* desc);
/* DevScene::sampleSequence (src/scene.h:480; src/scene.cpp:500-506 reads "sobol_10k_200.bin" into it): the table of the Sobol
 * sampler, numSamples x numDims uint32 in row order (sample index, then dimension), numDims = 200 (SobolSampleDim, src/sampler.h:11;
 * the reference's table has SobolSampleNum = 10 000 rows).  The reference picks its sampler at compile time (SAMPLER_USE_SOBOL,
 * src/common.h:4); here the scene carries the choice: once a table is set, every pass that draws random numbers (ReSTIRDirect,
 * ReSTIRIndirect, pathTrace*) runs the Sobol branch of src/sampler.h:9-36 -- Sampler(looper * 200 + dim, utilhash(index), data),
 * sample() = (data[ptr++] ^ scramble) * 2^-32, scramble = utilhash(scramble) -- and data == NULL selects the default thrust engine
 * (src/sampler.h:38-49) again.  In Sobol mode `looper` must stay in [0, numSamples): the caller wraps it as the reference does,
 * State::looper = (State::looper + 1) % SobolSampleNum (src/restir.cu:441-445; restir_compat.h does).  The table is copied. */
int  rs_scene_set_sample_sequence(rs_scene* scene, const uint32_t* data, int numSamples, int numDims);
/* Scene::clear / DevScene::destroy (src/scene.cpp:217-220,511-532). */
int  rs_scene_destroy(rs_scene* scene);

/* Camera::update (src/sceneStructs.h:88-102): view/right/up/rotationMatInv from rotation. */
int  rs_camera_update(rs_camera* cam);

/* ---- scene files (host only; no GPU call) ------------------------------------------------------
 * Scene::Scene(filename) (src/scene.cpp:96-131) with loadMaterial (:371-433), loadModel (:222-283), loadCamera
 * (:285-354), Resource::loadOBJMesh (:27-61) and the instance baking of buildDevData (:161-176): parses the
 * reference's text scene format and returns the flat arrays rs_scene_build_textured takes.  Image files named in
 * the scene must be PNG, JPEG (Huffman-coded, 8 bit), TGA, BMP, binary PPM (P6, 8 bit) or Radiance HDR (.hdr, RGBE) -- the other stb_image
 * formats stay with the caller, who can build the arrays directly; glTF meshes are not read.  Pointers in the view stay valid until
 * rs_scene_file_free. */
typedef struct rs_scene_file rs_scene_file;
typedef struct rs_scene_file_view {
    int                numPrims;
    const float*       vertices;      /* 9 floats / triangle, transformed (scene.cpp:167) */
    const float*       normals;       /* 9 floats / triangle, normalize(normalMat * n) (:168) */
    const float*       texcoords;     /* 6 floats / triangle */
    const int*         materialIds;   /* 1 / triangle */
    int                numMaterials;
    const rs_material* materials;
    int                numTextures;
    const rs_texture*  textures;
    int                envMapTexId;   /* -1: none */
    rs_camera          camera;        /* after Camera::update */
    int                iterations;    /* "Sample" */
    int                traceDepth;    /* "Depth" (Settings::traceDepth) */
    const char*        imageName;     /* "File" */
    int                numSkippedObjects;   /* objects whose mesh file could not be opened ("[Fail to load, skipped]", scene.cpp:234-240) */
} rs_scene_file_view;
int  rs_scene_file_load(const char* path, rs_scene_file** file);
int  rs_scene_file_get(const rs_scene_file* file, rs_scene_file_view* view);
int  rs_scene_file_free(rs_scene_file* file);
/* Math::buildTransformationMatrix (src/mathUtil.cpp:13-20): column-major 4x4, rotation in degrees. */
int  rs_build_transformation_matrix(const float* translation, const float* rotation, const float* scale, float* out16);
/* scene.cpp:167-168 for one instance: vec3(transform * vec4(v, 1)) and normalize(transpose(mat3(inverse(transform))) * n). */
int  rs_bake_instance(const float* translation, const float* rotation, const float* scale, int n,
                      const float* vertsIn, const float* normalsIn, float* vertsOut, float* normalsOut);

/* ---- scene services, batched (for parity tests of DevScene::intersect / testOcclusion) */
/* DevScene::intersect (src/scene.h:245-284): n rays of 6 floats (origin, direction), device ptrs.
 * Outputs (device): primId[n], matId[n], pos[3n], norm[3n]. */
int  rs_trace_closest(const rs_scene* scene, int n, const float* devRays,
                      int* devPrimId, int* devMatId, float* devPos, float* devNorm);
/* The same query through the wave-level service the multi-bounce kernels use for their bounce rays (general-case rays walk the
 * closest-hit tree of their threaded order, occlusion_bvh.cpp; same outputs, same bits). */
int  rs_trace_closest_wave(const rs_scene* scene, int n, const float* devRays,
                           int* devPrimId, int* devMatId, float* devPos, float* devNorm);
/* on = 0: bounce rays walk the reference's own tree (src/scene.h:245-284 literally); 1: the closest-hit trees again (the default
 * when the scene's tables allow them).  *was (may be NULL) receives the previous setting.  Results are identical either way. */
int  rs_scene_set_ordered_tree(rs_scene* scene, int on, int* was);
/* Host only (no device call): builds the closest-hit trees of the bounce rays from a reference table (rs_build_bvh's outputs) and
 * verifies what their walk relies on: the leaves list the triangles in the order the threaded orders of src/bvh.cpp:156-193 meet
 * them, boxes contain what is below them, miss links nest.  nodeCounts[3] / maxDepth[3] (may be NULL): per axis. */
int  rs_ordered_bvh_host_check(int numPrims, int bvhSize, const float* boundingBoxes, const int* const bvhNodes[6],
                               int* nodeCounts, int* maxDepth);
/* DevScene::testOcclusion (src/scene.h:286-316): n segments of 6 floats (x, y). */
int  rs_trace_occlusion(const rs_scene* scene, int n, const float* devSegments, int* devOccluded);

/* ---- GBuffer (src/gbuffer.h:24-27; create/destroy are defined in src/denoiser.cu:373-403) */
int  rs_gbuffer_create(int width, int height, rs_gbuffer** g);
int  rs_gbuffer_destroy(rs_gbuffer* g);
/* GBuffer::render (src/gbuffer.cu:80-86) */
int  rs_gbuffer_render(rs_gbuffer* g, const rs_scene* scene, const rs_camera* cam);
/* row-strip variant for framebuffer tiling: only rows [y0,y1) are rendered */
int  rs_gbuffer_render_rows(rs_gbuffer* g, const rs_scene* scene, const rs_camera* cam, int y0, int y1);
/* GBuffer::update (src/gbuffer.cu:75-78): lastCamera = cam; frameIdx ^= 1 */
int  rs_gbuffer_update(rs_gbuffer* g, const rs_camera* cam);
int  rs_gbuffer_get_view(const rs_gbuffer* g, rs_gbuffer_view* view);
/* Rows [y0,y0+rows) of the id / normal / depth planes as one packed device buffer (20 B/px); sel 0 =
 * planes of the current frame index, 1 = the "last" planes.  Used by the multi-GPU tiling to share
 * G-buffer history when the camera moves (findTemporalNeighbor reads lastPrimId/lastNormal/lastDepth
 * at a reprojected pixel that may belong to another strip). */
size_t rs_gbuffer_rows_bytes(const rs_gbuffer* g, int rows);
int  rs_gbuffer_rows_pack(const rs_gbuffer* g, int sel, int y0, int rows, void* devBuffer);
int  rs_gbuffer_rows_unpack(rs_gbuffer* g, int sel, int y0, int rows, const void* devBuffer);

/* ---- ReSTIR (src/restir.h:128-133) ------------------------------------------------------ */
/* ReSTIRInit (src/restir.cu:478-504): reservoir buffers for width*height pixels, zero-filled. */
int  rs_restir_init(int width, int height, rs_restir** r);
/* ReSTIRFree (src/restir.cu:506-514) */
int  rs_restir_free(rs_restir* r);
/* ReSTIRReset (src/restir.cu:516-518): re-arms the first-frame flag */
int  rs_restir_reset(rs_restir* r);
/* ReSTIRDirect (src/restir.cu:418-446).  looper = State::looper (the caller increments it, as the
 * reference launcher does at :441-445); reuse = Settings::reservoirReuse (bit0 temporal, bit1 spatial).
 * Semantics: the two-phase contract (phase A for every pixel, then phase B), DESIGN.md "Q1". */
int  rs_restir_direct(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                      float* devDirectIllum, int iter, int looper, int reuse);
/* Row-strip variants for framebuffer tiling across GPUs: phase A (primary hit, RIS, shadow ray,
 * temporal merge, publish) and phase B (spatial reuse, shade, accumulate) on rows [y0,y1).
 * Between them the caller exchanges `halo` rows of published reservoirs with its neighbours
 * (rs_restir_halo_pack / _unpack).  rs_restir_end_frame swaps the ping-pong buffers and clears the
 * first-frame flag (src/restir.cu:434-438). */
int  rs_restir_phase_a(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                       int looper, int reuse, int y0, int y1);
int  rs_restir_phase_b(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                       float* devDirectIllum, int iter, int reuse, int y0, int y1);
int  rs_restir_end_frame(rs_restir* r);
/* What the measurement of rs_set_side_stream's mode 4 decided for this object and its current scene: 0 GBuffer::render and the
 * primary rays as two launches, 1 as one, -1 still measuring (decided once 14 frames with such a launch have been enqueued; a
 * caller that times frames runs those first), -2 nothing was measured (synchronous launches; a forced mode; launches below three
 * rounds of the chip's wave slots, which are always one launch -- rs_restir_last_launch tells what a frame actually ran). */
int  rs_restir_launch_choice(const rs_restir* r, int* choice);
/* What the last rs_restir_phase_a / rs_restir_direct call launched, measured or not: *fused = 1 GBuffer::render in the primary rays'
 * launch, 0 its own launch (-1 before the first call); *chains = how many internal streams the frames' chains take in turn (0 in
 * synchronous mode).  A launch below three rounds of wave slots (a strip) is fused and on three chains without a measurement. */
int  rs_restir_last_launch(const rs_restir* r, int* fused, int* chains);
#define RS_SPATIAL_HALO_ROWS 5           /* taps reach y-4..y+5 (src/restir.cu:49-56) */
/* bytes needed for `rows` rows of published reservoirs */
size_t rs_restir_halo_bytes(const rs_restir* r, int rows);
int  rs_restir_halo_pack(const rs_restir* r, int y0, int rows, void* devBuffer);
int  rs_restir_halo_unpack(rs_restir* r, int y0, int rows, const void* devBuffer);
/* Same packing for any of the three reservoir buffers (which: as rs_restir_download); which = 1 after
 * rs_restir_end_frame is the history the next frame's temporal merge reads. */
size_t rs_restir_rows_bytes(const rs_restir* r, int which, int rows);
int  rs_restir_rows_pack(const rs_restir* r, int which, int y0, int rows, void* devBuffer);
int  rs_restir_rows_unpack(rs_restir* r, int which, int y0, int rows, const void* devBuffer);
/* Debug / parity: copy a reservoir buffer to host as the reference's AoS records.
 * which: 0 = devDirectReservoir (next frame's output slot), 1 = devLastDirectReservoir (last
 * written), 2 = devDirectTemp. */
int  rs_restir_download(const rs_restir* r, int which, rs_reservoir* host);
int  rs_restir_upload(rs_restir* r, int which, const rs_reservoir* host);
/* BVH walks (intersect + testOcclusion calls) performed by the last rs_restir_direct / phase_a,
 * for the Mrays/s metric. Synchronises. */
int  rs_restir_ray_count(rs_restir* r, unsigned long long* rays);
/* Sum of the walk counters of the last `frames` (<= 1024) frames. Synchronises. */
int  rs_restir_ray_total(rs_restir* r, int frames, unsigned long long* rays);
/* Per-pass GPU time (ms) of the last frame, measured with hipEvents on the library's stream:
 * ms[0] primary hit, ms[1] RIS, ms[2] shadow+temporal, ms[3] spatial+shade.  Synchronises. */
int  rs_restir_pass_times(rs_restir* r, float ms[4]);
/* Enables the hipEvent bracketing above (off by default: it adds 5 event records per frame).  enable = 2: only the spatial pass is
 * bracketed (ms[3]) and the launches stay where the overlapped mode puts them: the pass's duration while the kernels of other
 * frames share the CUs with it. */
int  rs_restir_enable_timing(rs_restir* r, int enable);
/* enable_timing(r, 2) keeps the two events of the last 256 frames: their spans (ms, oldest first; *count of them, at most `capacity`).
 * Nothing waits inside the frames -- what the pass costs while other frames' kernels share the CUs.  Waits for the library stream. */
int  rs_restir_spatial_times(rs_restir* r, float* ms, int capacity, int* count);
/* 1: the spatial pass is launched under the name k_spatial_shade_probe (the same code) until 0 -- for the launches a measurement
 * makes for itself, so that a kernel trace of the process tells them from the launches of the frames. */
int  rs_restir_set_probe(rs_restir* r, int enable);
/* Test hook: the spatial pass estimates tap positions with the hardware sqrt/sin/cos and falls back
 * to the exact evaluation inside an error band; this returns the largest estimate error over n
 * pseudo-random samples so a test can assert the band really covers it. */
int  rs_debug_tap_estimate_error(int n, float* maxErr);
/* Test hook: the light sampler takes the square root of a uniform variate without the general expansion's range scaling and
 * class test (rs_surface.h sqrt_of_uniform); this compares it with the exactly rounded sqrtf on every value the generator can
 * return (2^31 - 2 of them) and returns the number of differing results. */
int  rs_debug_sqrt_of_uniform_mismatches(unsigned long long* mismatches);
/* Test hook: the EAW filter divides by a sigma that is not a power of two in three instructions (Markstein's form, denoiser.hip
 * div_sigma); this compares it with the IEEE division on every float in [2^-100, 2^100] and returns the number of differing quotients. */
int  rs_debug_div_sigma_mismatches(float sigma, unsigned long long* mismatches);
/* The same for every value the Sobol sampler can return: 0 and all floats in [2^-32, 1] (2^28 + 2 of them). */
int  rs_debug_sqrt_of_unit_floats_mismatches(unsigned long long* mismatches);
/* Test hook for restir_amd/csrc/rs_exact.h (the short forms of 1 / d, x / d and sqrt(x) the kernels use for operands in [2^-60, 2^60)),
 * each compared with the compiler's correctly rounded operator.  op 0: the reciprocal of EVERY float in the range and of its negative; op 1: the square root
 * of every float in the range; op 2: the quotient x / d for the denominators 1.s * 2^expD, s in [firstSig, firstSig + countSig) (at most
 * 65 536 per call), each against ALL 2^23 numerators 1.t * 2^expX; op 3 / 4 / 5: the same with negative numerators / a negative denominator /
 * both.  out3[0] = results that differ (must be 0), out3[1] = operands
 * (pairs) compared, out3[2] reserved.  tools/verify_exact_division.py walks op 2 over all 2^23 denominators: every pair of significands. */
int  rs_debug_exact_ops_mismatches(int op, unsigned firstSig, unsigned countSig, int expX, int expD, unsigned long long* out3);

/* ---- framebuffer tiling across the GPUs of one node: the row-strip frame of one rank ------------------------------------------
 * One process per GPU, the scene replicated, the framebuffer of runCuda (src/main.cpp:146-185) cut into `world` row strips.  A rank
 * renders the G-buffer and phase A on its rows, sends its 5 border rows of published reservoirs and of the G-buffer id / normal /
 * depth planes to the strip above and below (68 B/px, the only exchange; taps reach y-4..y+5, src/restir.cu:49-56), runs phase B on
 * its interior rows while they travel and on the two border bands after they have arrived.  Results equal the full-frame
 * rs_restir_direct bit for bit.  The EAW filter on the strip (config 5), the history exchange a moving camera needs and the
 * assembly of the image go through the same transport (rs_strips_eaw_filter / _exchange_history / _gather).  One frame of a rank:
 *     rs_strips_frame;  [rs_strips_eaw_filter;]  rs_gbuffer_update;  [rs_strips_exchange_history;]  [tone map;  rs_strips_gather]
 * (restir_amd/tiling.py drives the same calls from Python.) */
typedef struct rs_comm rs_comm;
typedef struct rs_strips rs_strips;
/* The exchange, as two grouped operations on device buffers.  stream_ordered != 0: send / recv enqueue on `hipStream` and the
 * buffers are touched in stream order (RCCL); 0: they are host calls that complete at group_end at the latest (the frame then
 * hands over a finished library stream and waits for group_end before it unpacks). */
typedef struct rs_transport {
    void* ctx;
    int (*group_begin)(void* ctx);                 /* may be null */
    int (*send)(void* ctx, const void* devBuffer, size_t bytes, int peer, void* hipStream);
    int (*recv)(void* ctx, void* devBuffer, size_t bytes, int peer, void* hipStream);
    int (*group_end)(void* ctx);                   /* may be null */
    int stream_ordered;
} rs_transport;
/* ncclComm: an ncclComm_t of `world` ranks created by the caller (ncclCommInitRank); librccl.so is opened at run time, the
 * transfers are ncclSend / ncclRecv (ncclUint8) inside ncclGroupStart / ncclGroupEnd on the library stream (rs_set_stream), between
 * the launch that packs the border rows and the launch that unpacks the neighbours'; rs_strips_set_comm_stream(strips, 1) puts them on
 * a stream of the strip driver instead, with phase B's interior rows next to them (a fifth stream that hands events to the others, and the
 * device runs four of those side by side: the frames then keep two chains in flight instead of three -- 0.17 -> 0.28 ms per frame on a 1/8
 * strip of 1080p with a transport that moves nothing, DESIGN.md sections 4 and 5).  Create the communicator before the first overlapped frame or call rs_choose_internal_streams_again() after it. */
int  rs_comm_create_rccl(void* ncclComm, int rank, int world, rs_comm** comm);
/* The same, naming the copy of RCCL that created ncclComm (a process can hold two: PyTorch wheels bundle their own librccl.so).
 * librcclPath NULL or "" = rs_comm_create_rccl's search: symbols the process has linked or loaded globally, else a copy already
 * loaded under the soname librccl.so.1 (nothing new is mapped), else a fresh local open of librccl.so.1. */
int  rs_comm_create_rccl_lib(void* ncclComm, int rank, int world, const char* librcclPath, rs_comm** comm);
int  rs_comm_create(const rs_transport* transport, int rank, int world, rs_comm** comm);
int  rs_comm_destroy(rs_comm* comm);
/* one-rank check of a transport: `bytes` of devSend travel to this rank itself into devRecv, grouped and ordered as in a frame */
int  rs_comm_self_exchange(rs_comm* comm, const void* devSend, void* devRecv, size_t bytes);
/* bounds: world + 1 row offsets (bounds[0] = 0, bounds[world] = height, strips of at least 5 rows), or null for equal heights */
int  rs_strips_create(rs_comm* comm, int width, int height, const int* bounds, rs_strips** strips);
int  rs_strips_destroy(rs_strips* strips);
int  rs_strips_rows(const rs_strips* strips, int* y0, int* y1);
/* Stream-ordered transports (RCCL): 0 (default) = the transfers are enqueued on the library stream, in order with the packing and
 * unpacking copies; 1 = on a stream of the driver, ordered by events, so that the interior rows of phase B run while the border rows
 * travel.  Same results.  Call between frames with no gather in flight. */
int  rs_strips_set_comm_stream(rs_strips* strips, int ownStream);
/* Rows of the G-buffer id / normal / depth planes that travel with the 5 reservoir rows of an edge: 5 (default) for the spatial taps; 32 when a
 * denoiser follows (src/main.cpp:160-170), whose own exchange of those rows -- a packing launch, a group and an unpacking launch per frame -- is
 * then skipped.  The same value on every rank; strips of at least that many rows; between frames. */
int  rs_strips_set_gbuffer_halo(rs_strips* strips, int rows);
/* GBuffer::render + ReSTIRDirect of this rank's rows; GBuffer::update stays with the caller, as in runCuda.  Afterwards rows
 * [y0, y1) of devDirectIllum hold the frame's radiance. */
int  rs_strips_frame(rs_strips* strips, rs_restir* r, const rs_scene* scene, const rs_camera* cam, rs_gbuffer* g,
                     float* devDirectIllum, int iter, int looper, int reuse);
/* LeveledEAWFilter::filter (src/denoiser.cu:453-477) on the strip's rows of devColor (a full-frame sized image): the 32 G-buffer
 * rows beyond each edge arrive once, the 2 << level border rows of every level's input before that level.  *devResult is a buffer
 * of the driver whose rows [y0, y1) equal the full-frame rs_eaw_filter's.  Strips of at least 32 rows; call before
 * rs_gbuffer_update; the rows of devColor and of the current G-buffer planes just outside the strip are overwritten. */
int  rs_strips_eaw_filter(rs_strips* strips, rs_eaw* f, rs_gbuffer* g, const rs_camera* cam, float* devColor, float** devResult);
/* SpatioTemporalFilter::filter (src/denoiser.cu:532-564) on the strip: the 32 G-buffer rows beyond each edge arrive once, one row of
 * the accumulated moments after the temporal accumulation, the 2 * step + 1 border rows of every level's input colour and of the
 * variance before that level.  Rows [y0, y1) of the result equal the full-frame rs_svgf_filter's; *devColorOut is handed over as by
 * rs_svgf_filter.  Strips of at least 33 rows; call before rs_gbuffer_update, then rs_svgf_next_frame.  With a moving camera the
 * filter's history (accumulated colour and moments, 24 B/px) travels to every rank by rs_strips_exchange_svgf_history, called after
 * the filter and before rs_svgf_next_frame. */
int  rs_strips_svgf_filter(rs_strips* strips, rs_svgf* f, rs_gbuffer* g, const rs_camera* cam, const float* devColorIn, float** devColorOut);
int  rs_strips_exchange_svgf_history(rs_strips* strips, rs_svgf* f);
/* Moving camera (findTemporalNeighbor reads the reprojected pixel of the last frame, src/restir.cu:20-45, which may belong to
 * another strip): every rank's rows of the reservoirs the next temporal merge reads and of the "last" G-buffer planes travel to
 * every other rank.  Call after rs_gbuffer_update.  Not needed for a static camera (the reference's default). */
int  rs_strips_exchange_history(rs_strips* strips, rs_restir* r, rs_gbuffer* g);
/* Image assembly: rows [y0, y1) of every rank's devImage (full-frame sized, bytesPerPixel bytes per pixel: 12 for the radiance
 * image, 4 for the display image) arrive in the same rows on rank `root`, or on every rank for root = -1. */
int  rs_strips_gather(rs_strips* strips, void* devImage, size_t bytesPerPixel, int root);
/* The same, overlapped with the next frame: _begin records the transfers, which then travel in the next transfer group the driver
 * posts (the border rows of the next rs_strips_frame: one RCCL group per frame) or, failing that, in a group of their own posted by
 * _end; _end returns with the transfers ordered before whatever is enqueued on the library stream next (call it before devImage is
 * written again or read on root).  devImage's rows must be final on the library stream when the next group is posted, i.e. enqueue
 * their producer before rs_strips_frame.  slot in [0, 4) names a gather in flight; every rank begins the same gathers in the same
 * order (all transfers of a strip driver share one stream). */
int  rs_strips_gather_begin(rs_strips* strips, void* devImage, size_t bytesPerPixel, int root, int slot);
int  rs_strips_gather_end(rs_strips* strips, int slot);
/* Measurement: with timing enabled rs_strips_frame brackets the library stream's wait for the neighbours' border rows with two
 * events; rs_strips_halo_wait_ms returns that span for the last frame -- the part of the exchange's latency that the interior rows
 * of phase B did not hide (about 0 when the rows had arrived in time).  It waits for that frame. */
int  rs_strips_enable_timing(rs_strips* strips, int enable);
int  rs_strips_halo_wait_ms(rs_strips* strips, float* ms);

/* ---- path-trace baseline (src/pathtrace.h:12-16) ---------------------------------------- */
int  rs_path_trace_init(void);            /* pathTraceInit (src/pathtrace.cu:23-25) */
int  rs_path_trace_free(void);            /* pathTraceFree (:27-28) */
/* pathTraceDirect (src/pathtrace.cu:457-476) -> PTDirectKernel (:279-328) */
int  rs_path_trace_direct(const rs_scene* scene, const rs_camera* cam, float* devDirectIllum,
                          int iter, int looper, unsigned long long* rays);

/* pathTrace(direct, indirect, iter) -> singleKernelPT (src/pathtrace.cu:156-277,434-455) and pathTraceIndirect ->
 * PTIndirectKernel (:330-432,478-497); maxDepth = Settings::traceDepth */
int  rs_path_trace(const rs_scene* scene, const rs_camera* cam, float* devDirectIllum, float* devIndirectIllum,
                   int iter, int looper, int maxDepth, unsigned long long* rays);
int  rs_path_trace_indirect(const rs_scene* scene, const rs_camera* cam, float* devIndirectIllum,
                            int iter, int looper, int maxDepth, unsigned long long* rays);
/* ReSTIRIndirect(devIndirectIllum, iter, gBuffer) (src/restir.cu:233-416,448-476): one path per pixel into a
 * Reservoir<IndirectLiSample>, temporal reuse (reuse bit 0), clamp<20>; shares the first-frame flag with rs_restir_direct. */
int  rs_restir_indirect(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g, float* devIndirectIllum,
                        int iter, int looper, int reuse, int maxDepth, unsigned long long* rays);
/* which: 0 = the buffer the next call writes, 1 = the one the last call wrote (devIndLastTemporalReservoir) */
int  rs_restir_download_indirect(rs_restir* r, int which, rs_indirect_reservoir* host);

/* ---- display conversion (src/pathtrace.h:8) ---------------------------------------------- */
/* copyImageToPBO(uchar4*, glm::vec3*, w, h, toneMapping, scale) (src/pathtrace.cu:108-113) */
int  rs_copy_image_to_pbo(void* devPBO, const float* devImage, int width, int height, int toneMapping, float scale);
/* the debug-view overloads copyImageToPBO(uchar4*, glm::vec2* / float* / int*, w, h) (src/pathtrace.h:9-11,
 * src/pathtrace.cu:58-106,115-134): gamma only; the int form shows a pixel-index plane such as devMotion */
int  rs_copy_image2_to_pbo(void* devPBO, const float* devImage, int width, int height);
int  rs_copy_imagef_to_pbo(void* devPBO, const float* devImage, int width, int height);
int  rs_copy_imagei_to_pbo(void* devPBO, const int* devImage, int width, int height);

/* ---- viewer interop (src/main.cpp:105-144,176-181, src/preview.cpp:88,112-133) ---------------------------------------------
 * The OpenGL pixel-buffer object of the interactive viewer through HIP's graphics interop, one call per reference call:
 *     cudaGLRegisterBufferObject(pbo)            -> rs_pbo_register(pbo, &p)      (an OpenGL context must be current on the thread)
 *     cudaGLMapBufferObject((void**)&devPBO,pbo) -> rs_pbo_map(p, &devPBO, NULL)  (ordered on the library stream)
 *     copyImageToPBO(devPBO, devImage, ...)      -> rs_copy_image_to_pbo(devPBO, devImage, ...)
 *     cudaGLUnmapBufferObject(pbo)               -> rs_pbo_unmap(p)
 *     cudaGLUnregisterBufferObject(pbo)          -> rs_pbo_unregister(p)
 * (cudaGLSetGLDevice(0) is rs_init(0).)  On a node without a GL context rs_pbo_register fails with the runtime's error. */
typedef struct rs_pbo rs_pbo;
int  rs_pbo_register(unsigned glBuffer, rs_pbo** pbo);
int  rs_pbo_map(rs_pbo* pbo, void** devPBO, size_t* bytes /* may be null */);
int  rs_pbo_unmap(rs_pbo* pbo);
int  rs_pbo_unregister(rs_pbo* pbo);
/* saveImage(false) (src/main.cpp:105-144 with Image::setPixel / savePNG, src/image.cpp:36-58): tone map (0 none, 1 filmic, 2 ACES) and
 * gamma per pixel, stored mirrored in x, clamped, scaled by 255 and truncated; an 8-bit RGB PNG at `path` (the complete file name: the
 * viewer composes "<imageName>.<time>.<samples>samp.png").  The JPEG form (saveImage(true)) is not provided.  Synchronises. */
int  rs_save_image(const char* path, const float* devImage, int width, int height, int toneMapping);
/* the PNG writer by itself (host only): rgb = height rows of width * 3 bytes */
int  rs_write_png(const char* path, const unsigned char* rgb, int width, int height);
/* saveImage(true) (src/main.cpp:138-140): the same tone-mapped, mirrored 8-bit picture through Image::saveJPG (src/image.cpp:60-74:
 * stb_image_write's stbi_write_jpg at quality 90) -- a baseline JFIF file, 4:4:4, Annex-K tables scaled to the quality, AAN float DCT;
 * byte-identical to the reference's file (pinned against its compiled writer).  `path` is the complete file name. */
int  rs_save_image_jpg(const char* path, const float* devImage, int width, int height, int toneMapping);
/* Image::saveJPG's file from bytes that are already clamped and scaled: rgb = height rows of width * 3 bytes. */
int  rs_write_jpg(const char* path, const unsigned char* rgb, int width, int height);

/* ---- EAW denoiser (src/denoiser.h:33-43,72-74) -------------------------------------------- */
int  rs_eaw_create(int width, int height, int level, rs_eaw** f);   /* LeveledEAWFilter::create */
int  rs_eaw_destroy(rs_eaw* f);
/* The members the viewer edits between frames (src/preview.cpp:262-265: LeveledEAWFilter::level and
 * waveletFilter.sigLumin / sigNormal / sigDepth, src/denoiser.h:18-27,41).  create() sets 64 / .2 / 1 (src/denoiser.cu:455).
 * `level` is stored and reported like the reference's member; LeveledEAWFilter::filter runs its five levels whatever it
 * holds (src/denoiser.cu:463-477), and so does rs_eaw_filter. */
int  rs_eaw_set_params(rs_eaw* f, float sigLumin, float sigNormal, float sigDepth, int level);
int  rs_eaw_get_params(const rs_eaw* f, float* sigLumin, float* sigNormal, float* sigDepth, int* level);
/* The levels of step 1, 2 and 4 read their 25 taps from an LDS tile of the block's pixels plus halo (default, 15 % faster per
 * filter call at 1080p) or gather them from memory like the levels of step 8 and 16 (0).  Same arithmetic in the same order: the
 * results are identical bit for bit (tests/test_gpu_parity.py test_eaw_tiled_levels_equal_plain_gathers). */
int  rs_eaw_set_tiled(rs_eaw* f, int tiled);
/* The arithmetic of a tap (waveletFilter's loop body, src/denoiser.cu:98-128).  0: every operation rounded separately in the reference's
 * order, as a host build of the reference without contraction computes it.  1: fused multiply-adds for the three squared distances, for
 * the exponent -(dc.dc / sigLumin + dn.dn / sigNormal + dp.dp / sigDepth) * log2(e) as one chain over coefficients -log2(e) / sigma, and
 * for sum += colour * w -- what a contracting compiler (nvcc's default) is free to produce; 36 vector instructions per tap for 50.  The
 * results agree to a few ulp (all sums are of non-negative terms); both stay inside the filter's stated rtol 1e-5 against the oracle.
 * DEFAULT 1 (a deviation from the reference's operation order that the parity contract states: the filters are compared within rtol,
 * never bit for bit); a caller that wants the reference's order sets 0. */
int  rs_eaw_set_fused(rs_eaw* f, int fused);
/* LeveledEAWFilter::filter (src/denoiser.cu:463-477): *devColorOut is in/out exactly like the
 * reference's `glm::vec3*& devColorOut` (it is swapped with the filter's internal buffer). */
int  rs_eaw_filter(rs_eaw* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam);
/* Row-strip form for framebuffer tiling: positions of rows [y0, y1) (EAWaveletFilter's cam.getPosition per tap, computed once),
 * then one wavelet level (src/denoiser.cu:64-134) on rows [y0, y1).  Level l reads rows up to 2 << l outside the strip from
 * devColorIn, the G-buffer and the positions; the caller exchanges those colour rows between the levels and owns the buffers. */
int  rs_eaw_positions_rows(rs_eaw* f, const rs_gbuffer* g, const rs_camera* cam, int y0, int y1);
int  rs_eaw_level_rows(rs_eaw* f, float* devColorOut, const float* devColorIn, const rs_gbuffer* g, int level, int y0, int y1);
/* ---- SVGF (src/denoiser.h:45-70) ---------------------------------------------------------- */
/* SpatioTemporalFilter::create / destroy (src/denoiser.cu:479-504); wavelet sigmas 4 / 128 / 1 as in the reference */
int  rs_svgf_create(int width, int height, int level, rs_svgf** f);
int  rs_svgf_destroy(rs_svgf* f);
/* SpatioTemporalFilter::level and waveletFilter.sig* (src/preview.cpp:278-286, src/denoiser.h:59); create() sets 4 / 128 / 1
 * (src/denoiser.cu:488).  As in the reference the filter always runs five levels. */
int  rs_svgf_set_params(rs_svgf* f, float sigLumin, float sigNormal, float sigDepth, int level);
int  rs_svgf_get_params(const rs_svgf* f, float* sigLumin, float* sigNormal, float* sigDepth, int* level);
/* 1 (default): the a-trous levels read their taps from an LDS tile (the reference's default sigmas only); 0: plain gathers.  Same bits. */
int  rs_svgf_set_tiled(rs_svgf* f, int tiled);
/* The arithmetic of a tap, as rs_eaw_set_fused: fused dot products and accumulation, one multiplication per exponent (the colour
 * weight's -log2(e) / denominator and the luminance staged once per pixel).  Acts with the reference's defaults sigNormal 128 and a
 * power-of-two sigDepth; other sigmas keep the separately rounded operations.  Stated tolerance against the oracle as before: rtol 3e-5.
 * DEFAULT 1, as rs_eaw_set_fused. */
int  rs_svgf_set_fused(rs_svgf* f, int fused);
/* SpatioTemporalFilter::filter (src/denoiser.cu:532-564): temporal accumulation (alpha .2), variance estimate, five
 * variance-guided a-trous levels.  *devColorOut is the reference's `glm::vec3*& devColorOut`: it is swapped with the
 * filter's buffers (the level-0 result becomes the history), so the caller continues with the pointer it gets back
 * and, as in the reference, buffers on both sides must come from hipMalloc: rs_svgf_destroy frees what the filter
 * holds at that time, the caller frees the pointer it holds. */
int  rs_svgf_filter(rs_svgf* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam);
int  rs_svgf_next_frame(rs_svgf* f);                                                  /* ::nextFrame (:566-568) */
/* Device pointers of the filter's state (devAccumColor / devAccumMoment / devVariance, src/denoiser.h:60-63). */
typedef struct rs_svgf_view {
    float* devAccumColor[2];
    float* devAccumMoment[2];
    float* devVariance;
    int    frameIdx;
    int    width, height;
} rs_svgf_view;
int  rs_svgf_get_view(const rs_svgf* f, rs_svgf_view* view);

int  rs_modulate_albedo(float* devImage, const rs_gbuffer* g);                        /* src/denoiser.cu:405-411 */
int  rs_add_image(float* devImage, const float* devIn, int width, int height);       /* :413-418 */
int  rs_add_image3(float* devOut, const float* devIn1, const float* devIn2, int width, int height); /* :420-425 */

#ifdef __cplusplus
}
#endif
#endif
