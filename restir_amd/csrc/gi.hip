// gi.hip -- the multi-bounce half of the reference:
//   singleKernelPT / pathTrace             src/pathtrace.cu:156-277,434-455
//   PTIndirectKernel / pathTraceIndirect   src/pathtrace.cu:330-432,478-497
//   ReSTIRIndirectKernel / ReSTIRIndirect  src/restir.cu:233-416,448-476   (Reservoir<IndirectLiSample>, temporal reuse)
//
// One lane per pixel, 8x8 pixel tile per wave.  The primary ray is coherent and uses the wave-cooperative packet
// walk; after the first bounce rays are incoherent, so continuation rays use the pair-cooperative per-lane walk of
// the reference's tree and shadow rays the shadow tree -- both wave-level services, which is why the path loop is
// run by the whole wave with per-lane `alive` flags instead of lanes breaking out of it.
// The three kernels share one path loop (path_loop below); what differs is cited at each switch.
#include "rs_internal.h"
#include "rs_bsdf.h"

using namespace rs;

namespace {

enum { kModePT = 0, kModePTIndirect = 1, kModeReSTIR = 2 };

// scene.h:358-362
__device__ inline float environment_map_pdf(const DevScene& s, f3 w) {
    const TexRec env = s.textures[s.envTex];
    float u, v;
    to_plane(w, u, v);
    return luminance(linear_sample(env, u, v)) * s.sumLightPowerInv * (float)env.width * (float)env.height * .5f;
}
// scene.h:121-126
__device__ inline float primitive_area(const DevScene& s, int prim) {
    const float* t = s.vertices + (size_t)prim * 9;
    const f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
    return length(cross(v1 - v0, v2 - v0)) * .5f;
}

struct PathState {
    f3 direct, indirect;          // kModePT: direct / indirect; others: indirect only (ReSTIR: the sample's Lo)
    // ReSTIR-GI bookkeeping (restir.cu:273-281,316-321)
    float primSamplePdf; bool primSampleDelta; f3 primWo; SurfMat primMaterial;
    f3 xv, nv, xs, ns;
    int walks;
};

// The loop of the three kernels from the first shaded hit on, run by the WHOLE wave: lanes whose path has ended
// (or never started: `alive` false) stay in the loop with their flag down, so that the shadow rays can use the
// wave-level shadow-tree walk (trace_occluded_wave) and the continuation rays the pair-cooperative walk
// (trace_closest_wave) instead of per-lane walks inside a divergent loop.  Per lane the sequence of
// random draws and arithmetic is that of the reference.
template <int MODE, bool TEX, typename Sampler>
__device__ inline void path_loop(const DevScene& s, Hit h, SurfMat material, Ray ray, Sampler& rng, int maxDepth, bool alive, PathState& st) {
    f3 throughput = splat(1.f);
    f3 norm = h.norm, pos = h.pos;
    f3 wo = -ray.d;
    const bool env = TEX && s.envTex >= 0;
    for (int depth = 1; depth <= maxDepth; depth++) {
        if (!__any(alive)) break;
        const bool deltaBSDF = material.type == 2;
        if (alive && material.type != 2 && dot(norm, wo) < 0.f) norm = -norm;

        // next-event estimation (pathtrace.cu:203-213 / 365-376, restir.cu:291-302): sampleDirectLight = light sample,
        // occlusion test towards it, then the single-sided / pdf part
        const bool nee = alive && !deltaBSDF && (MODE == kModePT || depth > 1) && s.numLights > 0;    // pathtrace.cu:203 vs :365, restir.cu:291
        LightSample c;
        c.pdf = kInvalidPdf; c.Li = splat(0.f); c.wi = splat(0.f); c.dist = 0.f; c.point = pos; c.id = 0;
        if (alive && !deltaBSDF && (MODE == kModePT || depth > 1)) {
            const f4 r = rng.uniform4();                                        // drawn even without lights (sample4D is an argument)
            if (nee) c = env ? sample_light_nv<true, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, pos, r)
                             : sample_light_nv<false, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, pos, r);
        }
        // (a sample without a valid pdf contributes nothing either way: counted as the reference's testOcclusion call, not walked)
        const bool occluded = trace_occluded_wave(s, pos, c.point, nee && c.pdf > 0.f);
        if (nee) {
            st.walks++;
            const float lightPdf = occluded ? kInvalidPdf : c.pdf;
            if (lightPdf > 0.f) {
                const float bsdfPdf = material_pdf(material, norm, wo, c.wi);
                const f3 add = ((((throughput * material_bsdf(material, norm, wo, c.wi)) * c.Li) * sat_dot(norm, c.wi)) / lightPdf) * power_heuristic(lightPdf, bsdfPdf);
                if (MODE == kModePT && depth == 1) st.direct = st.direct + add; else st.indirect = st.indirect + add;
            }
        }

        BsdfSample sample;
        sample.dir = splat(0.f); sample.bsdf = splat(0.f); sample.pdf = 0.f; sample.type = kBsInvalid;
        bool deltaSample = false;
        if (alive) {
            const f3 r3 = mk3(rng.uniform(), rng.uniform(), rng.uniform());    // sample3D
            sample = material_sample(material, norm, wo, r3);
            if (sample.type == kBsInvalid) alive = false;
            else if (sample.pdf < 1e-8f) alive = false;
        }
        const f3 curPos = pos;
        if (alive) {
            deltaSample = (sample.type & kBsSpecular) != 0;
            if (MODE != kModeReSTIR || depth > 1) {                             // restir.cu:315-325
                throughput = throughput * ((sample.bsdf / sample.pdf) * (deltaSample ? 1.f : abs_dot(norm, sample.dir)));
            }
            else {
                st.primSamplePdf = sample.pdf;
                st.primSampleDelta = deltaSample;
                st.xv = pos; st.nv = norm;
            }
            ray.o = pos + sample.dir * 1e-5f; ray.d = sample.dir;               // makeOffsetedRay
        }
        h = trace_closest_wave(s, ray, alive);
        if (alive) {
            st.walks++;
            wo = -ray.d;
            if (h.primId == kNullPrim) {
                if (env) {
                    const f3 radiance = env_radiance(s, ray.d) * throughput;
                    const float weight = deltaSample ? 1.f : power_heuristic(sample.pdf, environment_map_pdf(s, ray.d));
                    st.indirect = st.indirect + radiance * weight;
                }
                alive = false;
            }
            else {
                pos = h.pos; norm = h.norm;
                material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
                if (material.type == 4) {
                    if (!(dot(norm, ray.d) < 0.f)) {                            // SCENE_LIGHT_SINGLE_SIDED: the back side ends the path silently
                        const f3 radiance = material.baseColor;
                        const bool unweighted = deltaSample || (MODE == kModeReSTIR && depth == 1);      // restir.cu:353
                        const float weight = unweighted ? 1.f : power_heuristic(sample.pdf,
                            (luminance(radiance) * s.sumLightPowerInv * primitive_area(s, h.primId)) * dot(curPos - pos, curPos - pos) /
                                abs_dot(norm, normalize(curPos - pos)));         // Math::pdfAreaToSolidAngle (mathUtil.h:182-185)
                        st.indirect = st.indirect + (radiance * throughput) * weight;
                        if (MODE == kModeReSTIR && depth == 1) { st.xs = pos; st.ns = norm; }
                    }
                    alive = false;
                }
                else if (MODE == kModeReSTIR && depth == 1) { st.xs = pos; st.ns = norm; }
            }
        }
    }
}

__device__ __forceinline__ void accumulate(float* image, int index, f3 v, int iter) {
    float* o = image + (size_t)index * 3;
    st3(o, (ld3(o) * (float)iter + v) / (float)(iter + 1));
}

struct IndResv { f3 Lo, xv, nv, xs, ns; int M; float W; };            // Reservoir<IndirectLiSample>, 68 B at the boundary
__device__ inline IndResv ind_load(const rs_indirect_reservoir* p) {
    const float* f = reinterpret_cast<const float*>(p);
    IndResv r;
    r.Lo = ld3(f); r.xv = ld3(f + 3); r.nv = ld3(f + 6); r.xs = ld3(f + 9); r.ns = ld3(f + 12);
    r.M = __float_as_int(f[15]); r.W = f[16];
    return r;
}
__device__ inline void ind_store(rs_indirect_reservoir* p, const IndResv& r) {
    float* f = reinterpret_cast<float*>(p);
    st3(f, r.Lo); st3(f + 3, r.xv); st3(f + 6, r.nv); st3(f + 9, r.xs); st3(f + 12, r.ns);
    f[15] = __int_as_float(r.M); f[16] = r.W;
}
__device__ inline bool ind_invalid(float W) { return is_nan_or_inf(W) || W < 0.f; }

// 7 blocks per CU = 7 waves per SIMD caps the kernel at 72 VGPRs (it wants 110-140; ~200 B of scratch per lane, outside the
// walk): measured on the bench scene at depth 4, pathTrace 12.98 -> 10.6 ms, pathTraceIndirect 12.20 -> 10.0 ms,
// ReSTIRIndirect 15.26 -> 10.3 ms; 5 blocks 11.5 / 10.8 / 11.0 ms, 6 blocks 11.0 / 10.3 / 10.6 ms, 8 blocks (64 VGPRs) 12.3 / 12.5 / 14.1 ms
#ifndef RS_PATH_BLOCKS
#define RS_PATH_BLOCKS 7
#endif
template <int MODE, bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_PATH_BLOCKS) k_path(DevScene s, CamParams cam, float* __restrict__ directIllum, float* __restrict__ indirectIllum,
                                              rs_indirect_reservoir* __restrict__ resvOut, const rs_indirect_reservoir* __restrict__ resvIn,
                                              GBufView g, int looper, int iter, int maxDepth, int first, int reuse, int tilesX,
                                              unsigned long long* rayCount) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    const int x = bx * 32 + wave * 8 + (lane & 7);
    const int y = by * 8 + (lane >> 3);
    const bool inside = x < cam.width && y < cam.height;
    const int index = y * cam.width + x;
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::seeded(s.sampleSeq, looper, index, 0);     // pathtrace.cu:170,339, restir.cu:256
    const f4 r = rng.uniform4();
    const Ray ray = camera_sample(cam, x, y, r.x, r.y);
    const Hit h = trace_closest_packet(s, ray, inside);                // all 64 lanes take part in the wave's walk
    PathState st;
    st.direct = splat(0.f); st.indirect = splat(0.f); st.primSamplePdf = 0.f; st.primSampleDelta = false; st.primWo = -ray.d;
    st.primMaterial = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    st.xv = st.nv = st.xs = st.ns = splat(0.f);
    st.walks = 0;
    // primary hit (pathtrace.cu:172-190 / 343-350, restir.cu:259-270); lanes that end here keep alive = false
    bool alive = false;
    Hit hh = h;
    SurfMat material = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    if (inside) {
        st.walks = 1;
        if (h.primId == kNullPrim) {
            if (MODE == kModePT) st.direct = splat(1.f);                               // pathtrace.cu:175-178
        }
        else {
            f3 norm = h.norm;
            material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
            if (MODE == kModePT) material.baseColor = splat(1.f);                      // DENOISER_DEMODULATE (:181-185)
            if (material.type == 4) {
                if (MODE == kModePT) st.direct = splat(1.f);                           // :187-190
            }
            else {
                hh.norm = norm;
                st.primMaterial = material;
                alive = true;
            }
        }
    }
    path_loop<MODE, TEX, SamplerT<SOBOL>>(s, hh, material, ray, rng, maxDepth, alive, st);                // every lane of the wave takes part
    if (inside) {
        if (MODE == kModePT) {
            if (any_nan_or_inf(st.direct)) st.direct = splat(0.f);
            if (any_nan_or_inf(st.indirect)) st.indirect = splat(0.f);
            accumulate(directIllum, index, hdr_to_ldr(st.direct), iter);                // Math::HDRToLDR (:273-276)
            accumulate(indirectIllum, index, hdr_to_ldr(st.indirect), iter);
        }
        else if (MODE == kModePTIndirect) {
            if (any_nan_or_inf(st.indirect)) st.indirect = splat(0.f);
            accumulate(indirectIllum, index, st.indirect, iter);
        }
        else {
            // WriteSample (restir.cu:372-416)
            IndResv smp; smp.Lo = st.indirect; smp.xv = st.xv; smp.nv = st.nv; smp.xs = st.xs; smp.ns = st.ns; smp.M = 0; smp.W = 0.f;
            IndResv rv; rv.Lo = rv.xv = rv.nv = rv.xs = rv.ns = splat(0.f); rv.M = 0; rv.W = 0.f;
            float sampleWeight = 0.f;
            if (!(luminance(smp.Lo) < 1e-8f)) {                                         // !indirectSample.invalid()
                sampleWeight = luminance(smp.Lo / st.primSamplePdf);                    // toScalar(pHatIndirect / primSamplePdf), pHat = Lo
                if ((sampleWeight != sampleWeight) || sampleWeight < 0.f) sampleWeight = 0.f;
            }
            {
                const float u = rng.uniform();                                          // Reservoir::update
                rv.W += sampleWeight; rv.M++;
                if (u * rv.W < sampleWeight) { rv.Lo = smp.Lo; rv.xv = smp.xv; rv.nv = smp.nv; rv.xs = smp.xs; rv.ns = smp.ns; }
            }
            if (!first && (reuse & 1)) {                                                // findTemporalNeighbor (restir.cu:20-45)
                const int primId = g.primId[index];
                const int lastIdx = g.motion[index];
                bool diff = false;
                if (lastIdx < 0) diff = true;
                else if (primId <= kNullPrim) diff = true;
                else if (g.lastPrimId[lastIdx] != primId) diff = true;
                else {
                    const f3 n = ld3(g.normal + (size_t)index * 3), ln = ld3(g.lastNormal + (size_t)lastIdx * 3);
                    const float depth = g.depth[index], pdepth = g.lastDepth[lastIdx];
                    if (abs_dot(n, ln) < .9f || gabs(pdepth - depth) > depth * .1f) diff = true;
                }
                IndResv t; t.Lo = t.xv = t.nv = t.xs = t.ns = splat(0.f); t.M = 0; t.W = 0.f;
                if (!diff) t = ind_load(resvIn + lastIdx);
                if (!ind_invalid(t.W)) {
                    const float u = rng.uniform();                                      // Reservoir::merge (restir.h:61-68)
                    rv.W += t.W; rv.M += t.M;
                    if (u * rv.W < t.W) { rv.Lo = t.Lo; rv.xv = t.xv; rv.nv = t.nv; rv.xs = t.xs; rv.ns = t.ns; }
                }
            }
            f3 indirect = splat(0.f);
            if (rv.M > 20) { rv.W *= (float)20 / (float)rv.M; rv.M = 20; }            // clamp<20>() (restir.h:79-86)
            if (!ind_invalid(rv.W)) {
                const f3 primWi = normalize(rv.xs - rv.xv);
                indirect = ((rv.Lo / luminance(rv.Lo)) * rv.W) / (float)rv.M;
                indirect = indirect * (material_bsdf(st.primMaterial, rv.nv, st.primWo, primWi) * (st.primSampleDelta ? 1.f : sat_dot(rv.nv, primWi)));
            }
            if (any_nan_or_inf(indirect)) indirect = splat(0.f);
            ind_store(resvOut + index, rv);
            accumulate(indirectIllum, index, indirect, iter);
        }
    }
    // BVH walks for the Mrays/s metric: wave-level sum, one atomic per wave
    int walks = st.walks;
    for (int off = 32; off > 0; off >>= 1) walks += __shfl_down(walks, off);
    if (lane == 0 && walks) atomicAdd(rayCount + (blockIdx.x % 64) * 8, (unsigned long long)walks);
}

// ---- the wavefront form: one launch per stage and bounce over queues of live paths -----------------------------------------------------
// k_path above keeps a path in one lane from the camera to its end: after the first bounce a wave's lanes die one by one (31 of 64 walk
// on average), every lane carries the whole path state across both walks (110-140 VGPRs wanted, capped at 72 with scratch), and the rays
// of a wave start anywhere and go anywhere.  Here a path's state lives in memory (SoA planes indexed by queue slot) between the stages
//     primary   camera ray (packet walk), first hit                                   -> hit queue of depth 1
//     per depth d = 1 .. maxDepth:
//       shade    light sample + BSDF sample of every path in the hit queue              -> shadow queue, ray queue (one bucket per threaded order)
//       shadow   any-hit walk of every shadow segment; the visible ones add their contribution to the pixel's sum
//       extend   closest-hit walk of every bounce ray, the hit's own contribution (emissive surface, environment) -> hit queue of depth d + 1
//     finish    per pixel: the tail of the three kernels (accumulate; ReSTIR: reservoir update, temporal merge, shade)
// so that every wave of a walk kernel starts with 64 rays (queues are compacted with a ballot + one atomic per wave), the bounce rays of a
// wave share one threaded order (getMTBVHId, src/scene.h:101-119: six buckets) and the walk kernels carry nothing but the ray across the walk.
// Per path the draws, their order and every arithmetic expression are those of path_loop; the per-pixel sums receive their terms in the
// path's own order (shadow of depth d before extend of depth d on one stream), so the images, reservoirs and ray counts are bit-identical.
struct GiQueues {
    int n;                                             // capacity of every queue = pixels
    int* counters;                                     // ctr(q, depth, k): 0 hit-queue entries, 1 shadow-queue entries, 2..7 ray-queue entries per order
    // hit queue: a path at a surface that will be shaded
    int* hPixel; uint32_t* hRng; int* hPtr;
    float4 *hPos, *hNorm, *hWo, *hMatA, *hMatB;       // pos | thr.x, norm | thr.y, wo | thr.z, baseColor | type, metallic roughness ior 0
    // ray queue, slot = order * n + i
    int* rPixel; uint32_t* rRng; int* rPtr;
    float4 *rPos, *rDir, *rThr;                        // surface point | pdf of the sample, direction | sample is specular, throughput
    // shadow queue
    float4 *sX, *sY; float2* sAdd;                     // surface point | pixel and flags, light point | add.x, add.y add.z
    // per pixel: the sums, and what ReSTIRIndirect keeps of a path (restir.cu:273-281,316-321)
    float4 *accD, *accI;
    float4 *pWo, *pMatA, *pMatB, *pXv, *pNv, *pXs, *pNs;       // primWo | primSamplePdf, primMaterial (pMatB.w: primSampleDelta), ...
    uint32_t* endRng; int* endPtr;                     // the sampler as the path left it (the reservoir update draws from it)
};
constexpr int kFlagAdd = 1 << 29, kFlagDirect = 1 << 30, kPixelMask = kFlagAdd - 1;
// Every counter has a 128-byte line of its own, and a block adds to it ONCE (block_append): with one atomic per wave and the seven counters
// of a depth in one line, the shade stage spent 2.3 ms per bounce in 230 000 serialised atomics (profiles/r04_gi_wavefront_first.log).
constexpr int kCtrStride = 32;
__device__ __forceinline__ int* ctr(const GiQueues& q, int depth, int k) { return q.counters + (depth * 8 + k) * kCtrStride; }

template <bool SOBOL> __device__ __forceinline__ void rng_save(const SamplerT<SOBOL>& r, uint32_t* word, int* ptr, int i);
template <> __device__ __forceinline__ void rng_save<false>(const SamplerT<false>& r, uint32_t* word, int*, int i) { word[i] = r.x; }
template <> __device__ __forceinline__ void rng_save<true>(const SamplerT<true>& r, uint32_t* word, int* ptr, int i) { word[i] = r.scramble; ptr[i] = r.ptr; }
template <bool SOBOL> __device__ __forceinline__ SamplerT<SOBOL> rng_load(const uint32_t* table, const uint32_t* word, const int* ptr, int i);
template <> __device__ __forceinline__ SamplerT<false> rng_load<false>(const uint32_t*, const uint32_t* word, const int*, int i) { SamplerT<false> r; r.x = word[i]; return r; }
template <> __device__ __forceinline__ SamplerT<true> rng_load<true>(const uint32_t* table, const uint32_t* word, const int* ptr, int i) {
    SamplerT<true> r; r.data = table; r.scramble = word[i]; r.ptr = ptr[i]; return r;
}

// Slot of this thread in queue `which` (0 .. NQ-1; -1: nothing to append) out of NQ queues the BLOCK appends to: ballots inside the waves,
// the waves' counts through LDS, one atomic per queue and block.  Every thread of the block must call it (it synchronises the block).
template <int NQ, int WAVES>
__device__ __forceinline__ int block_append(int which, int* const counter[NQ], int (&lds)[WAVES + 1][NQ]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank = 0, mine = 0;
    for (int k = 0; k < NQ; k++) {
        const unsigned long long m = __ballot(which == k);
        if (which == k) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == k) mine = __popcll(m);
    }
    __syncthreads();                                   // (the previous use of `lds` is over)
    if (lane < NQ) lds[wave][lane] = mine;
    __syncthreads();
    if (threadIdx.x < NQ) {
        int total = 0;
        for (int w = 0; w < WAVES; w++) { const int c = lds[w][threadIdx.x]; lds[w][threadIdx.x] = total; total += c; }
        lds[WAVES][threadIdx.x] = total ? atomicAdd(counter[threadIdx.x], total) : 0;
    }
    __syncthreads();
    return which >= 0 ? lds[WAVES][which] + lds[wave][which] + rank : 0;
}

template <int MODE, bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256) k_wf_primary(DevScene s, CamParams cam, GiQueues q, int looper, int tilesX) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    const int x = bx * 32 + wave * 8 + (lane & 7);
    const int y = by * 8 + (lane >> 3);
    const bool inside = x < cam.width && y < cam.height;
    const int index = y * cam.width + x;
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::seeded(s.sampleSeq, looper, index, 0);     // pathtrace.cu:170,339, restir.cu:256
    const f4 r = rng.uniform4();
    const Ray ray = camera_sample(cam, x, y, r.x, r.y);
    const Hit h = trace_closest_packet(s, ray, inside);                // all 64 lanes take part in the wave's walk
    // primary hit (pathtrace.cu:172-190 / 343-350, restir.cu:259-270)
    bool alive = false;
    f3 direct = splat(0.f), norm = h.norm;
    SurfMat material = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    if (inside) {
        if (h.primId == kNullPrim) {
            if (MODE == kModePT) direct = splat(1.f);                                  // pathtrace.cu:175-178
        }
        else {
            material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
            if (MODE == kModePT) material.baseColor = splat(1.f);                      // DENOISER_DEMODULATE (:181-185)
            if (material.type == 4) {
                if (MODE == kModePT) direct = splat(1.f);                              // :187-190
            }
            else alive = true;
        }
        if (MODE == kModePT) q.accD[index] = make_float4(direct.x, direct.y, direct.z, 0.f);
        q.accI[index] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == kModeReSTIR) {
            const SurfMat pm = alive ? material : SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
            q.pWo[index] = make_float4(-ray.d.x, -ray.d.y, -ray.d.z, 0.f);
            q.pMatA[index] = make_float4(pm.baseColor.x, pm.baseColor.y, pm.baseColor.z, __int_as_float(pm.type));
            q.pMatB[index] = make_float4(pm.metallic, pm.roughness, pm.ior, 0.f);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            q.pXv[index] = z; q.pNv[index] = z; q.pXs[index] = z; q.pNs[index] = z;
            if (!alive) rng_save<SOBOL>(rng, q.endRng, q.endPtr, index);
        }
    }
    __shared__ int lds[4 + 1][1];
    int* const counters[1] = { ctr(q, 1, 0) };
    const int slot = block_append<1, 4>(alive ? 0 : -1, counters, lds);
    if (alive) {
        q.hPixel[slot] = index;
        rng_save<SOBOL>(rng, q.hRng, q.hPtr, slot);
        q.hPos[slot] = make_float4(h.pos.x, h.pos.y, h.pos.z, 1.f);                   // throughput (1, 1, 1)
        q.hNorm[slot] = make_float4(norm.x, norm.y, norm.z, 1.f);
        q.hWo[slot] = make_float4(-ray.d.x, -ray.d.y, -ray.d.z, 1.f);
        q.hMatA[slot] = make_float4(material.baseColor.x, material.baseColor.y, material.baseColor.z, __int_as_float(material.type));
        q.hMatB[slot] = make_float4(material.metallic, material.roughness, material.ior, 0.f);
    }
}

// light sample and BSDF sample of every path in the hit queue of `depth` (the first half of path_loop's body)
constexpr int kShadeThreads = 1024;
template <int MODE, bool TEX, bool SOBOL>
__global__ void __launch_bounds__(kShadeThreads) k_wf_shade(DevScene s, GiQueues q, int depth) {
    const int count = *ctr(q, depth, 0);
    if ((int)(blockIdx.x * (unsigned)kShadeThreads) >= count) return;
    const int i = blockIdx.x * kShadeThreads + threadIdx.x;
    bool alive = i < count;
    const int j = alive ? i : 0;
    const int pixel = q.hPixel[j];
    SamplerT<SOBOL> rng = rng_load<SOBOL>(s.sampleSeq, q.hRng, q.hPtr, j);
    const float4 a0 = q.hPos[j], a1 = q.hNorm[j], a2 = q.hWo[j], m0 = q.hMatA[j], m1 = q.hMatB[j];
    const f3 pos = mk3(a0.x, a0.y, a0.z), wo = mk3(a2.x, a2.y, a2.z);
    f3 norm = mk3(a1.x, a1.y, a1.z), throughput = mk3(a0.w, a1.w, a2.w);
    const SurfMat material = SurfMat{ __float_as_int(m0.w), mk3(m0.x, m0.y, m0.z), m1.x, m1.y, m1.z };
    const bool env = TEX && s.envTex >= 0;
    const bool deltaBSDF = material.type == 2;
    if (alive && material.type != 2 && dot(norm, wo) < 0.f) norm = -norm;

    // next-event estimation (pathtrace.cu:203-213 / 365-376, restir.cu:291-302): the light sample here, the occlusion test in k_wf_shadow
    const bool nee = alive && !deltaBSDF && (MODE == kModePT || depth > 1) && s.numLights > 0;
    LightSample c;
    c.pdf = kInvalidPdf; c.Li = splat(0.f); c.wi = splat(0.f); c.dist = 0.f; c.point = pos; c.id = 0;
    if (alive && !deltaBSDF && (MODE == kModePT || depth > 1)) {
        const f4 r = rng.uniform4();                                            // drawn even without lights (sample4D is an argument)
        if (nee) c = env ? sample_light_nv<true, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, pos, r)
                         : sample_light_nv<false, const AliasRec*, const LightRec*>(s, s.alias, s.lights, s.numLights, pos, r);
    }
    f3 add = splat(0.f);
    const bool contributes = nee && c.pdf > 0.f;                                // (visible: lightPdf = c.pdf)
    if (contributes) {
        const float bsdfPdf = material_pdf(material, norm, wo, c.wi);
        add = ((((throughput * material_bsdf(material, norm, wo, c.wi)) * c.Li) * sat_dot(norm, c.wi)) / c.pdf) * power_heuristic(c.pdf, bsdfPdf);
    }

    BsdfSample sample;
    sample.dir = splat(0.f); sample.bsdf = splat(0.f); sample.pdf = 0.f; sample.type = kBsInvalid;
    bool deltaSample = false;
    const bool was = alive;
    if (alive) {
        const f3 r3 = mk3(rng.uniform(), rng.uniform(), rng.uniform());        // sample3D
        sample = material_sample(material, norm, wo, r3);
        if (sample.type == kBsInvalid) alive = false;
        else if (sample.pdf < 1e-8f) alive = false;
    }
    if (alive) {
        deltaSample = (sample.type & kBsSpecular) != 0;
        if (MODE != kModeReSTIR || depth > 1)                                   // restir.cu:315-325
            throughput = throughput * ((sample.bsdf / sample.pdf) * (deltaSample ? 1.f : abs_dot(norm, sample.dir)));
        else {
            float4 w = q.pWo[pixel]; w.w = sample.pdf; q.pWo[pixel] = w;       // primSamplePdf
            float4 b = q.pMatB[pixel]; b.w = deltaSample ? 1.f : 0.f; q.pMatB[pixel] = b;
            q.pXv[pixel] = make_float4(pos.x, pos.y, pos.z, 0.f); q.pNv[pixel] = make_float4(norm.x, norm.y, norm.z, 0.f);
        }
    }
    if (MODE == kModeReSTIR && was && !alive) rng_save<SOBOL>(rng, q.endRng, q.endPtr, pixel);
    // the shadow segment into the shadow queue, the bounce ray into the bucket of its threaded order (what walk_ordered_tree derives from
    // the direction): queue 0 and queues 1..6 of one block-level append
    __shared__ int lds[kShadeThreads / 64 + 1][8];
    int* const counters[8] = { ctr(q, depth, 1), ctr(q, depth, 2), ctr(q, depth, 3), ctr(q, depth, 4), ctr(q, depth, 5), ctr(q, depth, 6), ctr(q, depth, 7),
                               ctr(q, depth, 1) + 16 };       // the last one: shadow segments counted, not walked
    const int order = alive ? mtbvh_order(-sample.dir) : -1;
    // (two appends per thread: the ranks of the shadow queue and of the ray buckets are computed in two passes over one table)
    // A light sample with no valid pdf (the light faces away: scene.h:448-452) contributes nothing whether its segment is occluded or not
    // (sampleDirectLight returns InvalidPdf either way): it is counted as the reference's testOcclusion call that it is, and not walked.
    const int sSlot = block_append<8, kShadeThreads / 64>(contributes ? 0 : (nee ? 7 : -1), counters, lds);
    if (contributes) {
        const int flags = pixel | kFlagAdd | ((MODE == kModePT && depth == 1) ? kFlagDirect : 0);
        q.sX[sSlot] = make_float4(pos.x, pos.y, pos.z, __int_as_float(flags));
        q.sY[sSlot] = make_float4(c.point.x, c.point.y, c.point.z, add.x);
        q.sAdd[sSlot] = make_float2(add.y, add.z);
    }
    const int rSlot = block_append<8, kShadeThreads / 64>(order >= 0 ? 1 + order : -1, counters, lds);
    if (order >= 0) {
        const int slot = order * q.n + rSlot;
        q.rPixel[slot] = pixel;
        rng_save<SOBOL>(rng, q.rRng, q.rPtr, slot);
        q.rPos[slot] = make_float4(pos.x, pos.y, pos.z, sample.pdf);
        q.rDir[slot] = make_float4(sample.dir.x, sample.dir.y, sample.dir.z, deltaSample ? 1.f : 0.f);
        q.rThr[slot] = make_float4(throughput.x, throughput.y, throughput.z, 0.f);
    }
}

// any-hit walk of every shadow segment of `depth`; a visible one adds its contribution to its pixel's sum
__global__ void __launch_bounds__(256) k_wf_shadow(DevScene s, GiQueues q, int depth) {
    const int count = *ctr(q, depth, 1);
    if ((int)(blockIdx.x * 256u) >= count) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < count;
    const int j = active ? i : 0;
    const float4 x = q.sX[j], y = q.sY[j];
    const bool occluded = trace_occluded_wave(s, mk3(x.x, x.y, x.z), mk3(y.x, y.y, y.z), active);
    const int flags = __float_as_int(x.w);
    if (active && !occluded && (flags & kFlagAdd)) {
        const float2 yz = q.sAdd[j];
        float4* acc = ((flags & kFlagDirect) ? q.accD : q.accI) + (flags & kPixelMask);
        float4 v = *acc;
        v.x += y.w; v.y += yz.x; v.z += yz.y;
        *acc = v;
    }
}

// (A streaming form of the closest-hit walk -- a resident grid of waves that replace finished rays from their share of the queue, 87 % of
// the lanes walking instead of 48 % -- was built and measured: bit-exact and slower, 1.35 ms against 1.25 per bounce.  The walk is bound by
// the CU's vector-memory path, ~2 cycles per distinct 128-byte line and 155 line accesses per ray, busy 80-85 % either way; idle lanes
// cost it nothing.  EXPERIMENTS.md, commit d0141a4, profiles/r04_gi_wavefront_streaming_walk_counters.txt.)
// closest-hit walk of every bounce ray of `depth` and the hit's own contribution (the second half of path_loop's body)
template <int MODE, bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256) k_wf_extend(DevScene s, GiQueues q, int depth, int maxDepth) {
    // blocks in bucket order: bucket k takes ceil(count_k / 256) blocks
    int k = 0, first = 0, count = 0;
    {
        int b = blockIdx.x;
        for (k = 0; k < 6; k++) {
            count = *ctr(q, depth, 2 + k);
            const int blocks = (count + 255) >> 8;
            if (b < blocks) { first = b * 256; break; }
            b -= blocks;
        }
        if (k == 6) return;
    }
    const int i = first + threadIdx.x;
    bool alive = i < count;
    const int j = k * q.n + (alive ? i : 0);
    const float4 a0 = q.rPos[j], a1 = q.rDir[j];
    const f3 curPos = mk3(a0.x, a0.y, a0.z);
    Ray ray; ray.d = mk3(a1.x, a1.y, a1.z); ray.o = curPos + ray.d * 1e-5f;   // makeOffsetedRay
    const Hit h = trace_closest_wave(s, ray, alive);
    const bool mine = alive;                                                    // this lane carries a ray
    const float samplePdf = a0.w;
    const bool deltaSample = a1.w != 0.f;
    const int pixel = q.rPixel[j];
    const float4 a2 = q.rThr[j];
    const f3 throughput = mk3(a2.x, a2.y, a2.z);
    const bool env = TEX && s.envTex >= 0;
    f3 pos = curPos, norm = splat(0.f);
    SurfMat material = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    if (!mine) {}
    else if (h.primId == kNullPrim) {
        if (env) {
            const f3 radiance = env_radiance(s, ray.d) * throughput;
            const float weight = deltaSample ? 1.f : power_heuristic(samplePdf, environment_map_pdf(s, ray.d));
            float4 v = q.accI[pixel];
            v.x += radiance.x * weight; v.y += radiance.y * weight; v.z += radiance.z * weight;
            q.accI[pixel] = v;
        }
        alive = false;
    }
    else {
        pos = h.pos; norm = h.norm;
        material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
        if (material.type == 4) {
            if (!(dot(norm, ray.d) < 0.f)) {                                    // SCENE_LIGHT_SINGLE_SIDED: the back side ends the path silently
                const f3 radiance = material.baseColor;
                const bool unweighted = deltaSample || (MODE == kModeReSTIR && depth == 1);      // restir.cu:353
                const float weight = unweighted ? 1.f : power_heuristic(samplePdf,
                    (luminance(radiance) * s.sumLightPowerInv * primitive_area(s, h.primId)) * dot(curPos - pos, curPos - pos) /
                        abs_dot(norm, normalize(curPos - pos)));                 // Math::pdfAreaToSolidAngle (mathUtil.h:182-185)
                const f3 add = (radiance * throughput) * weight;
                float4 v = q.accI[pixel];
                v.x += add.x; v.y += add.y; v.z += add.z;
                q.accI[pixel] = v;
                if (MODE == kModeReSTIR && depth == 1) { q.pXs[pixel] = make_float4(pos.x, pos.y, pos.z, 0.f); q.pNs[pixel] = make_float4(norm.x, norm.y, norm.z, 0.f); }
            }
            alive = false;
        }
        else if (MODE == kModeReSTIR && depth == 1) { q.pXs[pixel] = make_float4(pos.x, pos.y, pos.z, 0.f); q.pNs[pixel] = make_float4(norm.x, norm.y, norm.z, 0.f); }
    }
    const bool goesOn = alive && depth < maxDepth;
    if (MODE == kModeReSTIR && mine && !goesOn) { q.endRng[pixel] = q.rRng[j]; if (SOBOL) q.endPtr[pixel] = q.rPtr[j]; }
    __shared__ int lds[4 + 1][1];
    int* const counters[1] = { ctr(q, depth + 1, 0) };
    const int slot = block_append<1, 4>(goesOn ? 0 : -1, counters, lds);
    if (goesOn) {
        q.hPixel[slot] = pixel;
        q.hRng[slot] = q.rRng[j]; if (SOBOL) q.hPtr[slot] = q.rPtr[j];
        q.hPos[slot] = make_float4(pos.x, pos.y, pos.z, throughput.x);
        q.hNorm[slot] = make_float4(norm.x, norm.y, norm.z, throughput.y);
        q.hWo[slot] = make_float4(-ray.d.x, -ray.d.y, -ray.d.z, throughput.z);
        q.hMatA[slot] = make_float4(material.baseColor.x, material.baseColor.y, material.baseColor.z, __int_as_float(material.type));
        q.hMatB[slot] = make_float4(material.metallic, material.roughness, material.ior, 0.f);
    }
}


// per pixel: what follows the path loop in the three kernels
template <int MODE, bool SOBOL>
__global__ void __launch_bounds__(256) k_wf_finish(DevScene s, GiQueues q, float* __restrict__ directIllum, float* __restrict__ indirectIllum,
                                                   rs_indirect_reservoir* __restrict__ resvOut, const rs_indirect_reservoir* __restrict__ resvIn,
                                                   GBufView g, int iter, int maxDepth, int first, int reuse, int pixels, unsigned long long* rayCount) {
    const int index = blockIdx.x * 256 + threadIdx.x;
    if (index == 0) {      // BVH walks for the Mrays/s metric: one camera ray per pixel, every shadow segment, every bounce ray
        unsigned long long walks = (unsigned long long)pixels;
        for (int d = 1; d <= maxDepth; d++) {
            for (int k = 1; k < 8; k++) walks += (unsigned long long)*ctr(q, d, k);
            walks += (unsigned long long)ctr(q, d, 1)[16];                                         // shadow segments counted, not walked
        }
        rayCount[0] = walks;
    }
    if (index >= pixels) return;
    const float4 ai = q.accI[index];
    f3 ind = mk3(ai.x, ai.y, ai.z);
    if (MODE == kModePT) {
        const float4 ad = q.accD[index];
        f3 dir = mk3(ad.x, ad.y, ad.z);
        if (any_nan_or_inf(dir)) dir = splat(0.f);
        if (any_nan_or_inf(ind)) ind = splat(0.f);
        accumulate(directIllum, index, hdr_to_ldr(dir), iter);                          // Math::HDRToLDR (:273-276)
        accumulate(indirectIllum, index, hdr_to_ldr(ind), iter);
    }
    else if (MODE == kModePTIndirect) {
        if (any_nan_or_inf(ind)) ind = splat(0.f);
        accumulate(indirectIllum, index, ind, iter);
    }
    else {
        // WriteSample (restir.cu:372-416)
        SamplerT<SOBOL> rng = rng_load<SOBOL>(s.sampleSeq, q.endRng, q.endPtr, index);
        const float4 w0 = q.pWo[index], m0 = q.pMatA[index], m1 = q.pMatB[index], xv = q.pXv[index], nv = q.pNv[index], xs = q.pXs[index], ns = q.pNs[index];
        const f3 primWo = mk3(w0.x, w0.y, w0.z);
        const float primSamplePdf = w0.w;
        const bool primSampleDelta = m1.w != 0.f;
        const SurfMat primMaterial = SurfMat{ __float_as_int(m0.w), mk3(m0.x, m0.y, m0.z), m1.x, m1.y, m1.z };
        IndResv smp; smp.Lo = ind; smp.xv = mk3(xv.x, xv.y, xv.z); smp.nv = mk3(nv.x, nv.y, nv.z); smp.xs = mk3(xs.x, xs.y, xs.z); smp.ns = mk3(ns.x, ns.y, ns.z); smp.M = 0; smp.W = 0.f;
        IndResv rv; rv.Lo = rv.xv = rv.nv = rv.xs = rv.ns = splat(0.f); rv.M = 0; rv.W = 0.f;
        float sampleWeight = 0.f;
        if (!(luminance(smp.Lo) < 1e-8f)) {                                         // !indirectSample.invalid()
            sampleWeight = luminance(smp.Lo / primSamplePdf);                       // toScalar(pHatIndirect / primSamplePdf), pHat = Lo
            if ((sampleWeight != sampleWeight) || sampleWeight < 0.f) sampleWeight = 0.f;
        }
        {
            const float u = rng.uniform();                                          // Reservoir::update
            rv.W += sampleWeight; rv.M++;
            if (u * rv.W < sampleWeight) { rv.Lo = smp.Lo; rv.xv = smp.xv; rv.nv = smp.nv; rv.xs = smp.xs; rv.ns = smp.ns; }
        }
        if (!first && (reuse & 1)) {                                                // findTemporalNeighbor (restir.cu:20-45)
            const int primId = g.primId[index];
            const int lastIdx = g.motion[index];
            bool diff = false;
            if (lastIdx < 0) diff = true;
            else if (primId <= kNullPrim) diff = true;
            else if (g.lastPrimId[lastIdx] != primId) diff = true;
            else {
                const f3 n = ld3(g.normal + (size_t)index * 3), ln = ld3(g.lastNormal + (size_t)lastIdx * 3);
                const float depth = g.depth[index], pdepth = g.lastDepth[lastIdx];
                if (abs_dot(n, ln) < .9f || gabs(pdepth - depth) > depth * .1f) diff = true;
            }
            IndResv t; t.Lo = t.xv = t.nv = t.xs = t.ns = splat(0.f); t.M = 0; t.W = 0.f;
            if (!diff) t = ind_load(resvIn + lastIdx);
            if (!ind_invalid(t.W)) {
                const float u = rng.uniform();                                      // Reservoir::merge (restir.h:61-68)
                rv.W += t.W; rv.M += t.M;
                if (u * rv.W < t.W) { rv.Lo = t.Lo; rv.xv = t.xv; rv.nv = t.nv; rv.xs = t.xs; rv.ns = t.ns; }
            }
        }
        f3 indirect = splat(0.f);
        if (rv.M > 20) { rv.W *= (float)20 / (float)rv.M; rv.M = 20; }            // clamp<20>() (restir.h:79-86)
        if (!ind_invalid(rv.W)) {
            const f3 primWi = normalize(rv.xs - rv.xv);
            indirect = ((rv.Lo / luminance(rv.Lo)) * rv.W) / (float)rv.M;
            indirect = indirect * (material_bsdf(primMaterial, rv.nv, primWo, primWi) * (primSampleDelta ? 1.f : sat_dot(rv.nv, primWi)));
        }
        if (any_nan_or_inf(indirect)) indirect = splat(0.f);
        ind_store(resvOut + index, rv);
        accumulate(indirectIllum, index, indirect, iter);
    }
}

unsigned long long* g_giRayCount = nullptr;     // 64 partial counters, 64 B apart
int g_pathForm = 1;        // 0: one kernel per path (k_path); 1: the wavefront form

}  // namespace
// measurement switch while both forms exist (tools/bench_gi.py --ab-form)
extern "C" int rs_debug_set_path_form(int form) { g_pathForm = form; return 0; }
namespace {

int gi_counters() {
    if (!g_giRayCount) RS_TRY(rs_dev_alloc(&g_giRayCount, 64 * 8));
    RS_HIP(hipMemsetAsync(g_giRayCount, 0, 64 * 8 * sizeof(unsigned long long), rs_stream()));
    return 0;
}
int gi_read_rays(unsigned long long* rays) {
    if (!rays) return 0;
    unsigned long long h[64 * 8];
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(h, g_giRayCount, sizeof h, hipMemcpyDeviceToHost));
    *rays = 0;
    for (int i = 0; i < 64; i++) *rays += h[i * 8];
    return 0;
}

}  // namespace

// the queues of the wavefront form: one set per context, sized for the largest frame it has traced
struct rs_gi_scratch {
    GiQueues q{};
    size_t pixels = 0;
    int depths = 0;
    bool restir = false, sobol = false, direct = false;
    std::vector<void*> owned;
};
void rs_gi_scratch_free(rs_context* c) {
    if (!c || !c->gi) return;
    for (void* p : c->gi->owned) (void)hipFree(p);
    delete c->gi;
    c->gi = nullptr;
}

namespace {

template <typename T> int gi_plane(rs_gi_scratch* g, T** p, size_t count) {
    RS_TRY(rs_dev_alloc(p, count));
    g->owned.push_back((void*)*p);
    return 0;
}
int gi_scratch(size_t pixels, int maxDepth, bool restir, bool sobol, bool direct, GiQueues* out) {
    rs_context* c = rs_ctx();
    rs_gi_scratch* g = c->gi;
    if (g && (g->pixels < pixels || g->depths < maxDepth + 2 || (restir && !g->restir) || (sobol && !g->sobol) || (direct && !g->direct))) {
        RS_TRY(rs_synchronize());
        restir = restir || g->restir; sobol = sobol || g->sobol; direct = direct || g->direct;
        if (pixels < g->pixels) pixels = g->pixels;
        if (maxDepth + 2 < g->depths) maxDepth = g->depths - 2;
        rs_gi_scratch_free(c);
        g = nullptr;
    }
    if (!g) {
        g = c->gi = new rs_gi_scratch();
        GiQueues& q = g->q;
        const size_t n = pixels;
        q.n = (int)n;
        g->pixels = n; g->depths = maxDepth + 2; g->restir = restir; g->sobol = sobol; g->direct = direct;
        RS_TRY(gi_plane(g, &q.counters, (size_t)g->depths * 8 * kCtrStride));
        RS_TRY(gi_plane(g, &q.hPixel, n)); RS_TRY(gi_plane(g, &q.hRng, n));
        RS_TRY(gi_plane(g, &q.hPos, n)); RS_TRY(gi_plane(g, &q.hNorm, n)); RS_TRY(gi_plane(g, &q.hWo, n)); RS_TRY(gi_plane(g, &q.hMatA, n)); RS_TRY(gi_plane(g, &q.hMatB, n));
        RS_TRY(gi_plane(g, &q.rPixel, 6 * n)); RS_TRY(gi_plane(g, &q.rRng, 6 * n));
        RS_TRY(gi_plane(g, &q.rPos, 6 * n)); RS_TRY(gi_plane(g, &q.rDir, 6 * n)); RS_TRY(gi_plane(g, &q.rThr, 6 * n));
        RS_TRY(gi_plane(g, &q.sX, n)); RS_TRY(gi_plane(g, &q.sY, n)); RS_TRY(gi_plane(g, &q.sAdd, n));
        RS_TRY(gi_plane(g, &q.accI, n));
        if (direct) RS_TRY(gi_plane(g, &q.accD, n));
        if (sobol) { RS_TRY(gi_plane(g, &q.hPtr, n)); RS_TRY(gi_plane(g, &q.rPtr, 6 * n)); }
        if (restir) {
            RS_TRY(gi_plane(g, &q.pWo, n)); RS_TRY(gi_plane(g, &q.pMatA, n)); RS_TRY(gi_plane(g, &q.pMatB, n)); RS_TRY(gi_plane(g, &q.pXv, n));
            RS_TRY(gi_plane(g, &q.pNv, n)); RS_TRY(gi_plane(g, &q.pXs, n)); RS_TRY(gi_plane(g, &q.pNs, n)); RS_TRY(gi_plane(g, &q.endRng, n));
            if (sobol) RS_TRY(gi_plane(g, &q.endPtr, n));
        }
    }
    *out = g->q;
    return 0;
}


template <int MODE, bool TEX, bool SOBOL>
int launch_wavefront_t(const rs_scene* scene, const CamParams& cp, const GiQueues& q, float* direct, float* indirect, rs_indirect_reservoir* out,
                       const rs_indirect_reservoir* in, const GBufView& g, int looper, int iter, int maxDepth, int first, int reuse) {
    const int W = cp.width, H = cp.height, n = W * H;
    const int tilesX = (W + 31) / 32, tilesY = (H + 7) / 8;
    hipStream_t st = rs_stream();
    RS_HIP(hipMemsetAsync(q.counters, 0, sizeof(int) * 8 * kCtrStride * (size_t)(maxDepth + 2), st));
    hipLaunchKernelGGL((k_wf_primary<MODE, TEX, SOBOL>), dim3(tilesX * tilesY), dim3(256), 0, st, scene->dev, cp, q, looper, tilesX);
    const int blocks = (n + 255) / 256;
    for (int depth = 1; depth <= maxDepth; depth++) {
        hipLaunchKernelGGL((k_wf_shade<MODE, TEX, SOBOL>), dim3((n + kShadeThreads - 1) / kShadeThreads), dim3(kShadeThreads), 0, st, scene->dev, q, depth);
        if (MODE == kModePT || depth > 1) hipLaunchKernelGGL(k_wf_shadow, dim3(blocks), dim3(256), 0, st, scene->dev, q, depth);
        hipLaunchKernelGGL((k_wf_extend<MODE, TEX, SOBOL>), dim3(blocks + 6), dim3(256), 0, st, scene->dev, q, depth, maxDepth);
    }
    hipLaunchKernelGGL((k_wf_finish<MODE, SOBOL>), dim3(blocks), dim3(256), 0, st, scene->dev, q, direct, indirect, out, in, g, iter, maxDepth, first, reuse, n, g_giRayCount);
    return rs_check_hip(hipGetLastError(), "pathTrace (wavefront)");
}

template <int MODE>
int launch_wavefront(const rs_scene* scene, const rs_camera* cam, float* direct, float* indirect, rs_indirect_reservoir* out,
                     const rs_indirect_reservoir* in, const GBufView& g, int looper, int iter, int maxDepth, int first, int reuse) {
    const CamParams cp = rs_make_cam_params(cam);
    const bool sobol = scene->dev.sampleSeq != nullptr;
    GiQueues q;
    RS_TRY(gi_scratch((size_t)cp.width * cp.height, maxDepth, MODE == kModeReSTIR, sobol, MODE == kModePT, &q));
#define RS_WF_ARGS scene, cp, q, direct, indirect, out, in, g, looper, iter, maxDepth, first, reuse
    if (scene->textured) return sobol ? launch_wavefront_t<MODE, true, true>(RS_WF_ARGS) : launch_wavefront_t<MODE, true, false>(RS_WF_ARGS);
    return sobol ? launch_wavefront_t<MODE, false, true>(RS_WF_ARGS) : launch_wavefront_t<MODE, false, false>(RS_WF_ARGS);
#undef RS_WF_ARGS
}


template <int MODE>
int launch_path(const rs_scene* scene, const rs_camera* cam, float* direct, float* indirect, rs_indirect_reservoir* out,
                const rs_indirect_reservoir* in, const GBufView& g, int looper, int iter, int maxDepth, int first, int reuse) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    const int tilesX = (W + 31) / 32, tilesY = (H + 7) / 8;
    const CamParams cp = rs_make_cam_params(cam);
    // The Sobol branch reads data[ptr++] without a bound (sampler.h:20); a path draws at most 4 + 7 per bounce + 2 numbers, and the
    // device table ends in a guard of kSobolGuard zeros: refuse what could read beyond it.
    const bool sobol = scene->dev.sampleSeq != nullptr;
    if (sobol) {
        RS_TRY(rs_check_looper(scene, looper, "pathTrace / ReSTIRIndirect"));
        if (6 + 7LL * maxDepth > kSobolSampleDim + kSobolGuard) return rs_fail(RS_ERR_INVALID_ARGUMENT, "pathTrace / ReSTIRIndirect: trace depth too large for the Sobol table's guard");
    }
    if (g_pathForm == 1 && maxDepth >= 1) return launch_wavefront<MODE>(scene, cam, direct, indirect, out, in, g, looper, iter, maxDepth, first, reuse);
    const dim3 grid(tilesX * tilesY), block(256);
#define RS_PATH_ARGS scene->dev, cp, direct, indirect, out, in, g, looper, iter, maxDepth, first, reuse, tilesX, g_giRayCount
    if (scene->textured) { if (sobol) hipLaunchKernelGGL((k_path<MODE, true, true>), grid, block, 0, rs_stream(), RS_PATH_ARGS);
                           else       hipLaunchKernelGGL((k_path<MODE, true, false>), grid, block, 0, rs_stream(), RS_PATH_ARGS); }
    else                 { if (sobol) hipLaunchKernelGGL((k_path<MODE, false, true>), grid, block, 0, rs_stream(), RS_PATH_ARGS);
                           else       hipLaunchKernelGGL((k_path<MODE, false, false>), grid, block, 0, rs_stream(), RS_PATH_ARGS); }
#undef RS_PATH_ARGS
    return 0;
}

}  // namespace

extern "C" {

int rs_path_trace(const rs_scene* scene, const rs_camera* cam, float* devDirectIllum, float* devIndirectIllum,
                  int iter, int looper, int maxDepth, unsigned long long* rays) {
    RS_SCOPE(scene);
    if (!scene || !cam || !devDirectIllum || !devIndirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "pathTrace: null argument");
    RS_TRY(gi_counters());
    GBufView none{};
    RS_TRY(launch_path<kModePT>(scene, cam, devDirectIllum, devIndirectIllum, nullptr, nullptr, none, looper, iter, maxDepth, 0, 0));
    RS_TRY(rs_after_launch("pathTrace"));
    return gi_read_rays(rays);
}

int rs_path_trace_indirect(const rs_scene* scene, const rs_camera* cam, float* devIndirectIllum, int iter, int looper, int maxDepth,
                           unsigned long long* rays) {
    RS_SCOPE(scene);
    if (!scene || !cam || !devIndirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "pathTraceIndirect: null argument");
    RS_TRY(gi_counters());
    GBufView none{};
    RS_TRY(launch_path<kModePTIndirect>(scene, cam, nullptr, devIndirectIllum, nullptr, nullptr, none, looper, iter, maxDepth, 0, 0));
    RS_TRY(rs_after_launch("pathTrace"));
    return gi_read_rays(rays);
}

int rs_restir_indirect(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g, float* devIndirectIllum,
                       int iter, int looper, int reuse, int maxDepth, unsigned long long* rays) {
    RS_SCOPE(r);
    RS_TRY(rs_gbuffer_join(g));                         // the render may still be on the auxiliary stream
    if (!r || !scene || !cam || !g || !devIndirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRIndirect: null argument");
    if (cam->resolution[0] != r->width || cam->resolution[1] != r->height || g->width != r->width || g->height != r->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRIndirect: size mismatch");
    const size_t n = (size_t)r->width * r->height;
    for (int i = 0; i < 2; i++)
        if (!r->indResv[i]) {                                   // devIndTemporalReservoir / devIndLastTemporalReservoir (restir.cu:13-14,491-494)
            RS_TRY(rs_dev_alloc(&r->indResv[i], n));
            RS_HIP(hipMemsetAsync(r->indResv[i], 0, n * sizeof(rs_indirect_reservoir), rs_stream()));
        }
    RS_TRY(gi_counters());
    RS_TRY(launch_path<kModeReSTIR>(scene, cam, nullptr, devIndirectIllum, r->indResv[0], r->indResv[1], gbuf_view(g), looper, iter, maxDepth,
                                    r->firstFrame ? 1 : 0, reuse));
    { rs_indirect_reservoir* t = r->indResv[0]; r->indResv[0] = r->indResv[1]; r->indResv[1] = t; }      // std::swap (:463)
    r->firstFrame = false;                                                                                    // ReSTIRFirstFrame (:465-467)
    RS_TRY(rs_after_launch("ReSTIR Indirect"));
    return gi_read_rays(rays);
}

int rs_restir_download_indirect(rs_restir* r, int which, rs_indirect_reservoir* host) {
    RS_SCOPE(r);
    if (!r || !host || which < 0 || which > 1 || !r->indResv[which]) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_download_indirect: bad argument");
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(host, r->indResv[which], (size_t)r->width * r->height * sizeof(rs_indirect_reservoir), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
