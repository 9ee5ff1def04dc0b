// restir.hip -- ReSTIRDirect (src/restir.cu:20-231,418-518) for CDNA4.
//
// The reference is one fused kernel per frame with a block-level __syncthreads() standing in for a
// grid-wide dependency.  Here the frame is five launches in stream order, which gives
// the two-phase contract by construction (stream order is the grid barrier):
//
//   phase A  k_primary          jittered primary ray, wave-cooperative packet walk of
//                               the MTBVH, material after its texture maps            restir.cu:127-153
//            k_ris / k_ris_lds  32-candidate RIS over the light table (no rays)       restir.cu:155-170
//            k_shadow           shadow ray on the RIS winner (shadow tree)            restir.cu:172-176
//            k_temporal         temporal merge, publish reservoirs                    restir.cu:178-194,211-212
//   phase B  k_spatial_shade    5-tap spatial reuse from an LDS-staged tile+halo,
//                               shade, accumulate                                     restir.cu:196-230
//
// Splitting the walk-bound passes from the ALU-bound RIS loop keeps the traversal kernels at low
// register counts (8 waves per SIMD) and lets the RIS kernel run without divergence.  The price is
// ~90 B/px of per-pixel state between passes (SurfPlanes below), which is small next to the walks.
//
// Reservoir storage is four planes (rs_internal.h ResvPlanes); what the spatial pass gathers from is
// one 16-byte tap record per pixel {W, M, G-buffer id, depth} (TempPlanes).  The spatial pass stages
// tap record + normal for a 32x16 tile plus a 5-pixel halo into LDS (35 KB), tracks the SOURCE PIXEL
// of the surviving sample through the merges, and gathers that one sample (32 B) at the end instead
// of staging Li / wi for 1092 pixels.
#include <cstdlib>
#include <cstring>

#include "rs_internal.h"

#ifndef RS_WALK_WAVES
#define RS_WALK_WAVES 8        // waves per SIMD the walk kernels are held to (launch bound; holding them to exactly that many was A/B'd in round 3: no gain)
#endif
using namespace rs;

namespace {

constexpr int kReservoirSize = 32;   // restir.cu:3
constexpr int kHalo = RS_SPATIAL_HALO_ROWS;

// kind of a pixel after the primary hit
constexpr int kKindMiss = 0, kKindLight = 1, kKindShaded = 2;
// per-pixel tag: material id (24 bits) | kind << 24 (2 bits) | Material::Type << 26 (3 bits)
__device__ __forceinline__ int mk_kind(int mk) { return (mk >> 24) & 3; }
__device__ __forceinline__ int mk_type(int mk) { return (mk >> 26) & 7; }
// Sobol sampler (src/sampler.h:9-36): the passes carry the scramble in rngMat.x and know the table position from the number of
// draws made so far -- 4 by the primary ray (restir.cu:129), 5 per RIS candidate (:158,168), and the temporal merge's one draw if it
// happened (bit 29 of the tag as k_temporal leaves it in rngMat.y).
constexpr int kDrawsPrimary = 4, kDrawsRis = kDrawsPrimary + 5 * 32;
constexpr unsigned kTemporalDrewBit = 1u << 29;
constexpr int kRaySlots = 1024;   // ring of per-frame BVH-walk counters
constexpr int kRaySub = 64, kRayStride = 8;   // per frame: 64 partial counters, 64 B apart (one hot address cost ~85 us/frame)

struct SurfPlanes {
    float4* posMat;     // pos.xyz (radiance of the environment map for a miss pixel of an env-mapped scene), bits(tag)
    float4* norm;       // shading normal xyz (normal-mapped, flipped to the wo side), w = metallic (after its map)
    float4* wo;         // wo.xyz, w = roughness (after its map); written and read only for the metallic BSDF
    uint2*  rngMat;     // { RNG state carried from pass to pass, matId | kind << 24 }: all the spatial pass needs for a Lambertian pixel
    float4* candLi;     // RIS winner Li.xyz, dist
    float4* candWi;     // RIS winner wi.xyz, reservoir weight
};

__device__ __forceinline__ void pixel_of_lane(int tilesX, int y0, int& x, int& y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    x = bx * 32 + wave * 8 + (lane & 7);
    y = y0 + by * 8 + (lane >> 3);
}

// ---- phase A.1: primary hit ---------------------------------------------------------------------
// what ReSTIRDirectKernel keeps of its primary hit (restir.cu:127-153) for the later passes; returns whether the pixel is shaded
template <bool TEX>
__device__ __forceinline__ int primary_store(const DevScene& s, const SurfPlanes& sp, int index, const Ray& ray, const Hit& h, unsigned rngWord) {
    int shaded = 0;
    int kind = kKindMiss, matId = 0, type = 0;
    f3 norm = splat(0.f), wo = splat(0.f), p = h.pos;
    float metallic = 0.f, roughness = 0.f;
    if (h.primId != kNullPrim) {
        matId = h.matId;
        norm = h.norm;
        const SurfMat m = TEX ? textured_material(s, h, norm) : plain_material(s, matId);    // restir.cu:140 (baseColor is then forced to 1)
        type = m.type; metallic = m.metallic; roughness = m.roughness;
        if (type == 4) {
            kind = kKindLight;
            norm = splat(0.f);
        }
        else {
            kind = kKindShaded;
            wo = -ray.d;
            if (type != 2 && dot(norm, wo) < 0.f) norm = -norm;     // restir.cu:150-153
            shaded = 1;
        }
    }
    else if (TEX && s.envTex >= 0) {
        p = env_radiance(s, ray.d);                                 // restir.cu:134-136: the pixel's radiance
    }
    const int mk = matId | (kind << 24) | (type << 26);
    sp.posMat[index] = make_float4(p.x, p.y, p.z, __int_as_float(mk));
    sp.norm[index] = make_float4(norm.x, norm.y, norm.z, metallic);
    if (type == 1) sp.wo[index] = make_float4(wo.x, wo.y, wo.z, roughness);     // only the metallic BSDF reads wo (k_ris, k_spatial_shade)
    sp.rngMat[index] = make_uint2(rngWord, (unsigned)mk);
    return shaded;
}

// SPLIT: the launch splits its heavy tiles (rs_tilesplit.h) and counts the nodes of every walk for it; the plain kernel carries none of that
template <bool TEX, bool SOBOL, bool SPLIT>
__device__ __forceinline__ void primary_body(const DevScene& s, const CamParams& cam, const SurfPlanes& sp, int looper,
                                             int y0, int y1, int tilesX, unsigned long long* rayCount, const TileSplit& ts) {
    RS_SETPRIO(RS_PRIO_WALK);
    int x, py, tile;
    bool mine, helper;
    if (!tile_split_map<8, 8>(ts, tilesX, threadIdx.x & 63, x, py, mine, tile, helper)) return;     // (a helper block without a tile)
    const int y = y0 + py;
    const bool inside = mine && x < cam.width && y < y1;
    int shaded = 0;
    const int index = y * cam.width + x;
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::seeded(s.sampleSeq, looper, index, 0);     // restir.cu:127
    f4 r = rng.uniform4();                              // sample4D: all four are drawn, two are used
    Ray ray = camera_sample(cam, x, y, r.x, r.y);
    unsigned unionNodes = 0;
    Hit h = trace_closest_packet<SPLIT>(s, ray, inside, &unionNodes);       // all 64 lanes take part in the wave's walk
    if (inside) shaded = primary_store<TEX>(s, sp, index, ray, h, rng.word());
    // BVH walks for the Mrays/s metric: one per pixel here, one more per shaded pixel (shadow ray)
    const unsigned long long ballotIn = __ballot(inside), ballotSh = __ballot(shaded);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long c = (unsigned long long)__popcll(ballotIn) + (unsigned long long)__popcll(ballotSh);
        if (c) atomicAdd(rayCount + (blockIdx.x % kRaySub) * kRayStride, c);
    }
    if (SPLIT) tile_split_report(ts.base, ts.rot, tile, helper, !helper && !mine, unionNodes);
}

template <bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_primary(DevScene s, CamParams cam, SurfPlanes sp, int looper,
                                                 int y0, int y1, int tilesX, unsigned long long* rayCount) {
    primary_body<TEX, SOBOL, false>(s, cam, sp, looper, y0, y1, tilesX, rayCount, TileSplit{ nullptr, 0, 0 });
}
template <bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_primary_split(DevScene s, CamParams cam, SurfPlanes sp, int looper,
                                                 int y0, int y1, int tilesX, unsigned long long* rayCount, TileSplit ts) {
    primary_body<TEX, SOBOL, true>(s, cam, sp, looper, y0, y1, tilesX, rayCount, ts);
}

// GBuffer::render and the primary rays of ReSTIRDirect in one launch (asynchronous mode, when the render of this frame is
// still pending -- rs_gbuffer_render_rows defers it), each ray stored as k_render_gbuffer / k_primary store it.
// Frames at which the measured launch choice takes its time stamps: two launches until kTuneB, one fused launch from there to kTuneD; the
// first span is frames kTuneA..kTuneB, the second kTuneC..kTuneD -- twelve frames each, and the four frames after every switch are not
// timed: chains run up to four frames ahead of the library stream, so the frames around a switch carry the other form's kernels next to
// them.  (Rounds 1-5 timed 2..8 against 8..14: on the Bistro-class scene, where the forms differ by 18 %, one run in four took the
// slower one -- 2.10 instead of 1.78 ms per frame, profiles/r06_fuse_tuner_flips.log.)
constexpr int kTuneA = 6, kTuneB = 18, kTuneC = 22, kTuneD = 34;
constexpr long long kFuseMinWaves = kSmallLaunchWaves;     // three rounds of the chip's 8 192 wave slots (256 CUs x 4 SIMDs x 8 waves)

// The two rays of a pixel sit in two LANES: a wave takes an 8x4 block of pixels, lanes 0-31 walk their pixel-centre rays and lanes
// 32-63 their jittered rays -- 64 rays in the ordinary one-ray-per-lane packet walk.  The two rays of a pixel visit almost the same
// nodes, so the wave's union of nodes is that of 32 pixels, not 64: fewer node visits per pixel than two launches over 8x8 tiles, at
// the single walk's cost per visit.  (Round 1's form walked both rays in one lane, one after the other at every node of an 8x8 tile's
// union: it paid both slab tests per visit and, after the round-2 walk, measured 1.29 ms per frame against 1.20 for two launches and
// 1.193 for this form.)  Tiles are 8x4 from the G-buffer rows [gy0, gy1), blocks 32x4 pixels; the shading ray is active on rows [y0, y1).
template <bool TEX, bool SOBOL, bool SPLIT>
__device__ __forceinline__ void gbuffer_primary_body(const DevScene& s, const CamParams& cam, const CamParams& lastCam, const GBufWrite& g, const SurfPlanes& sp, int looper,
                                                     int gy0, int gy1, int y0, int y1, int tilesX, unsigned long long* rayCount, const TileSplit& ts) {
    RS_SETPRIO(RS_PRIO_WALK);
    const int lane = threadIdx.x & 63;
    const bool shading = lane >= 32;                            // which of the pixel's two rays this lane carries
    int x, py, tile;
    bool mine, helper;
    if (!tile_split_map<8, 4>(ts, tilesX, lane & 31, x, py, mine, tile, helper)) return;
    const int y = gy0 + py;
    const bool inside = mine && x < cam.width && (shading ? (y >= y0 && y < y1) : y < gy1);
    const int index = y * cam.width + x;
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::seeded(s.sampleSeq, looper, index, 0);
    const f4 r = rng.uniform4();
    const Ray ray = shading ? camera_sample(cam, x, y, r.x, r.y) : camera_center_ray(cam, x, y);
    unsigned unionNodes = 0;
    const Hit h = trace_closest_packet<SPLIT>(s, ray, inside, &unionNodes);
    int shaded = 0;
    if (inside) {
        if (shading) shaded = primary_store<TEX>(s, sp, index, ray, h, rng.word());
        else gbuffer_store<TEX>(s, cam, lastCam, g, index, ray, h);
    }
    const unsigned long long ballotIn = __ballot(inside && shading), ballotSh = __ballot(shaded);
    if (lane == 0) {
        unsigned long long c = (unsigned long long)__popcll(ballotIn) + (unsigned long long)__popcll(ballotSh);
        if (c) atomicAdd(rayCount + (blockIdx.x % kRaySub) * kRayStride, c);
    }
    if (SPLIT) tile_split_report(ts.base, ts.rot, tile, helper, !helper && !mine, unionNodes);
}

template <bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_gbuffer_primary(DevScene s, CamParams cam, CamParams lastCam, GBufWrite g, SurfPlanes sp, int looper,
                                                                  int gy0, int gy1, int y0, int y1, int tilesX, unsigned long long* rayCount) {
    gbuffer_primary_body<TEX, SOBOL, false>(s, cam, lastCam, g, sp, looper, gy0, gy1, y0, y1, tilesX, rayCount, TileSplit{ nullptr, 0, 0 });
}
template <bool TEX, bool SOBOL>
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_gbuffer_primary_split(DevScene s, CamParams cam, CamParams lastCam, GBufWrite g, SurfPlanes sp, int looper,
                                                                  int gy0, int gy1, int y0, int y1, int tilesX, unsigned long long* rayCount, TileSplit ts) {
    gbuffer_primary_body<TEX, SOBOL, true>(s, cam, lastCam, g, sp, looper, gy0, gy1, y0, y1, tilesX, rayCount, ts);
}

// ---- phase A.2: RIS over the light table ----------------------------------------------------------
// Measured with the table in global memory: 32 candidates x (8-byte alias record + 64-byte light record) per
// pixel are five 64-address gathers per candidate, and the texture addresser was as busy (88 %) as the VALU.
// Up to kRisLdsLights lights the whole table (72 B per light) is copied into LDS by a 1024-thread block --
// two such blocks per CU keep 8 waves per SIMD -- and the gathers become ds_read_b128 / ds_read_b64.
#ifndef RS_RIS_THREADS
#define RS_RIS_THREADS 1024
#endif
constexpr int kRisThreads = RS_RIS_THREADS;
constexpr int kRisLdsLights = 1024;
constexpr int kRisAliasLdsLights = 16384;        // alias records only: 128 KB of the CU's 160 KB at most

template <bool ENV, bool SOBOL, typename AliasPtr, typename LightPtr>
__device__ __forceinline__ void ris_pixel(const DevScene& s, const SurfPlanes& sp, AliasPtr alias, LightPtr lights, int index, int looper) {
    RS_SETPRIO(RS_PRIO_RIS);
    const float4 pm = sp.posMat[index];
    const int mk = __float_as_int(pm.w);
    if (mk_kind(mk) != kKindShaded) return;
    const float4 nr = sp.norm[index];
    const f3 pos = mk3(pm.x, pm.y, pm.z), norm = mk3(nr.x, nr.y, nr.z);
    struct { int type; float metallic, roughness; } m = { mk_type(mk), nr.w, 0.f };     // the material after its maps (k_primary)
    const f3 baseColor = splat(1.f);                       // material.baseColor = 1 (restir.cu:141)
    f3 wo = splat(0.f);
    if (m.type == 1) { const float4 w4 = sp.wo[index]; wo = mk3(w4.x, w4.y, w4.z); m.roughness = w4.w; }

    // Sobol: the table position looper * 200 + 4 + 5 i + k is the same for every pixel -- the table words come through the scalar
    // cache, one fetch per wave, and a draw is one xor and one utilhash per lane
    SamplerT<SOBOL> rng = SamplerT<SOBOL>::resume(s.sampleSeq, sp.rngMat[index].x, looper, kDrawsPrimary);
    f3 selLi = splat(0.f), selWi = splat(0.f);
    float selDist = 0.f, wsum = 0.f;
    // Without an environment map every winner is a triangle-light sample: the loop carries its light and barycentric pair (three
    // conditional moves per candidate instead of seven) and the sample is evaluated again after the loop (light_sample_again).
    int selId = -1;
    float selU = 0.f, selV = 0.f;
    for (int i = 0; i < kReservoirSize; i++) {
        f4 r = rng.uniform4();
        LightSample c = sample_light_nv<ENV, AliasPtr, LightPtr>(s, alias, lights, s.numLights, pos, r);
        f3 g = c.Li * eval_bsdf(m.type, baseColor, m.metallic, m.roughness, norm, wo, c.wi) * sat_dot(norm, c.wi);
        float weight = luminance(div3_exact(g, c.pdf, c.pdf <= 0.f));             // pdf <= 0: the weight is replaced by 0 below
        if (is_nan_or_inf(weight) || c.pdf <= 0.f) weight = 0.f;
#ifdef RS_WALK_STATS        // measurement builds (tools/ris_stats.py): how many candidates end without a valid pdf / with a zero weight
        {
            const unsigned long long all = __ballot(true), inv = __ballot(c.pdf <= 0.f), zero = __ballot(c.pdf > 0.f && weight == 0.f);
            if (s.walkStats && __lane_id() == (unsigned)__ffsll((long long)all) - 1u) {
                atomicAdd(&s.walkStats[88], (unsigned long long)__popcll(all)); atomicAdd(&s.walkStats[89], (unsigned long long)__popcll(inv));
                atomicAdd(&s.walkStats[90], (unsigned long long)__popcll(zero)); atomicAdd(&s.walkStats[91], 1ull);
                if (inv == all) atomicAdd(&s.walkStats[92], 1ull);
                if ((inv | zero) == all) atomicAdd(&s.walkStats[93], 1ull);
            }
        }
#endif
        float u = rng.uniform();
        wsum += weight;                                    // Reservoir::update (restir.h:38-44)
        if (u * wsum < weight) {
            if (ENV) { selLi = c.Li; selWi = c.wi; selDist = c.dist; }
            else { selId = c.id; selU = c.bu; selV = c.bv; }
        }
    }
    if (!ENV && selId >= 0) light_sample_again(lights, selId, selU, selV, pos, selLi, selWi, selDist);
    sp.candLi[index] = make_float4(selLi.x, selLi.y, selLi.z, selDist);
    sp.candWi[index] = make_float4(selWi.x, selWi.y, selWi.z, wsum);
    reinterpret_cast<unsigned*>(sp.rngMat + index)[0] = rng.word();
}

template <bool ENV, bool SOBOL>
__global__ void __launch_bounds__(256) k_ris(DevScene s, SurfPlanes sp, int width, int y0, int y1, int looper) {
    const int n0 = y0 * width, n1 = y1 * width;
    const int index = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= n1) return;
    ris_pixel<ENV, SOBOL, const AliasRec*, const LightRec*>(s, sp, s.alias, s.lights, index, looper);
}

template <bool SOBOL>
__global__ void __launch_bounds__(kRisThreads) k_ris_lds(DevScene s, SurfPlanes sp, int width, int y0, int y1, int looper) {
    // The copy is laid out by quarter: the lanes of a wave read the same quarter of 64 random records, and in record order those
    // 16 bytes lie in 2 of the 8 four-bank groups whatever the light (a 4-fold bank conflict; SQ_LDS_BANK_CONFLICT was 80 % of the
    // LDS cycles); by quarter, light i's lies in group i mod 8.
    __shared__ float4 sQuarters[4 * kRisLdsLights];
    __shared__ AliasRec sAlias[kRisLdsLights];
    {
        const float4* src = reinterpret_cast<const float4*>(s.lights);
        for (int i = threadIdx.x; i < s.numLights * 4; i += kRisThreads) sQuarters[(i & 3) * kRisLdsLights + (i >> 2)] = src[i];
        for (int i = threadIdx.x; i < s.numLights; i += kRisThreads) sAlias[i] = s.alias[i];
    }
    __syncthreads();
    const int n0 = y0 * width, n1 = y1 * width;
    const int index = n0 + blockIdx.x * kRisThreads + threadIdx.x;
    if (index >= n1) return;
    ris_pixel<false, SOBOL, const AliasRec*, LightQuarters<kRisLdsLights>>(s, sp, sAlias, LightQuarters<kRisLdsLights>{ sQuarters }, index, looper);
}

// More lights than the LDS copy of the whole table holds (config 5: 10 240): the alias records alone (8 B per light) still fit -- one
// LDS read and four 16-byte gathers of the light record per candidate instead of five gathers; the light records stay in L2.
// Dynamic LDS: numLights * 8 bytes.
template <bool SOBOL>
__global__ void __launch_bounds__(kRisThreads) k_ris_alias_lds(DevScene s, SurfPlanes sp, int width, int y0, int y1, int looper) {
    extern __shared__ AliasRec sAliasDyn[];
    for (int i = threadIdx.x; i < s.numLights; i += kRisThreads) sAliasDyn[i] = s.alias[i];
    __syncthreads();
    const int n0 = y0 * width, n1 = y1 * width;
    const int index = n0 + blockIdx.x * kRisThreads + threadIdx.x;
    if (index >= n1) return;
    ris_pixel<false, SOBOL, const AliasRec*, const LightRec*>(s, sp, sAliasDyn, s.lights, index, looper);
}

// ---- phase A.3: shadow ray, temporal merge, publish -------------------------------------------------
struct Resv { f3 Li, wi; float dist; int M; float W; };

__device__ __forceinline__ bool resv_invalid(float W) { return is_nan_or_inf(W) || W < 0.f; }   // restir.h:51-53

// Streaming accesses: data that is read or written once per frame and not again before it has left every cache.  Marked
// non-temporal so that it does not push out what the next pass re-reads -- the published tap records and normals that the
// spatial pass stages with a 5-pixel halo (measured: spatial pass 59 -> 56 us with the reservoir stores alone).
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) { const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ld_stream(const int* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ f3 ld3_stream(const float* p) { return mk3(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2)); }
__device__ __forceinline__ void st_stream(float4* p, float x, float y, float z, float w) { __builtin_nontemporal_store(vf4{ x, y, z, w }, reinterpret_cast<vf4*>(p)); }

__device__ __forceinline__ void resv_store(const ResvPlanes& p, int i, const Resv& r) {      // read by the NEXT frame's temporal merge
    st_stream(p.li + i, r.Li.x, r.Li.y, r.Li.z, r.dist);
    st_stream(p.wi + i, r.wi.x, r.wi.y, r.wi.z, 0.f);
    __builtin_nontemporal_store(r.W, p.w + i);
    __builtin_nontemporal_store(r.M, p.m + i);
}

// Phase A.3 is two launches.  The shadow ray (restir.cu:172-176) depends on this frame's RIS winner only, so it belongs to the
// chain primary rays -> RIS -> shadow rays that no other frame feeds -- the auxiliary streams run it next to the previous frame's
// passes -- and its whole effect is to clear the weight of an occluded winner, which it does in place (candWi.w).  What is left on
// the library stream, the only chain that links consecutive frames, is the streaming half: temporal merge, validity check, publish.
// (As one kernel the pass lasted as long as its slowest wave's walk, 0.2 ms on a 1/8 strip however few rows it has.)
// 8 blocks per CU = 8 waves per SIMD.
// (A hand-off of each wave's last rays to one wave per block was built and measured in round 3 -- same results, a quarter fewer wave
// iterations, no gain: the pass is bound by per-lane L1 look-ups; EXPERIMENTS.md, commit 933552f.)
__global__ void __launch_bounds__(256, RS_WALK_WAVES) k_shadow(DevScene s, SurfPlanes sp, int width, int y0, int y1, int tilesX) {
    RS_SETPRIO(RS_PRIO_WALK);
    int x, y;
    pixel_of_lane(tilesX, y0, x, y);
    const bool inside = x < width && y < y1;
    const int index = inside ? y * width + x : 0;
    const float4 pm = inside ? sp.posMat[index] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool shaded = inside && mk_kind(__float_as_int(pm.w)) == kKindShaded;
    float4 cl = make_float4(0.f, 0.f, 0.f, 0.f), cw = cl;
    if (shaded) { cl = sp.candLi[index]; cw = sp.candWi[index]; }
    const f3 pos = mk3(pm.x, pm.y, pm.z), wi = mk3(cw.x, cw.y, cw.z);
    // every lane of the wave takes part in the cooperative any-hit walk
    // (the tree's top levels as an LDS-resident table -- 54 % of a ray's node steps -- were built and measured in round 4: bit-exact and
    // slower, 0.478 -> 0.528 ms; EXPERIMENTS.md, commit 6f00700)
    const bool occluded = trace_occluded_wave(s, pos, pos + wi * cl.w, shaded);
    if (shaded && occluded) reinterpret_cast<float*>(sp.candWi + index)[3] = 0.f;      // `if (testOcclusion(...)) reservoir.weight = 0`
}

// (The shadow rays as a resident grid of waves that replace finished rays from a run of tiles -- 87 % of the lanes walking instead of 54 % --
// were built and measured in round 4: bit-exact and THREE TIMES slower, 0.475 -> 1.38 ms.  The 64 rays of an 8x8 tile start next to each
// other and walk in lockstep through the same cache lines; replacement desynchronises them, and what the wave then fetches per step is 64
// different lines.  EXPERIMENTS.md, commit 5febf79, profiles/r04_ab_shadow_streaming_replacement.log.)
template <bool SOBOL>
__global__ void __launch_bounds__(256) k_temporal(SurfPlanes sp, GBufView g, ResvPlanes last, ResvPlanes cur, TempPlanes temp, const uint32_t* sampleSeq, int looper,
                                                  int first, int reuse, int n0, int n1, unsigned long long* rayWork, unsigned long long* rayDone) {
    RS_SETPRIO(RS_PRIO_STREAM);
    // this call's BVH-walk counters are complete (the launch is ordered after the chain that counted): publish them and leave the
    // working slot zero for its next user, so that the chain itself needs no clearing launch
    if (blockIdx.x == 0 && threadIdx.x < kRaySub) {
        const int k = threadIdx.x * kRayStride;
        rayDone[k] = rayWork[k];
        rayWork[k] = 0;
    }
    const int index = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= n1) return;
    const uint2 rm = sp.rngMat[index];
    const bool shaded = mk_kind((int)rm.y) == kKindShaded;
    const int gid = ld_stream(g.primId + index);
    const float gdepth = ld_stream(g.depth + index);
    if (!shaded) {
        // early-exit pixels publish no reservoir (Q1: their slot keeps its stale value); only the
        // G-buffer half of the tap record is refreshed
        if (reuse & 2) reinterpret_cast<float2*>(temp.tap + index)[1] = make_float2(__int_as_float(gid), gdepth);
        return;
    }
    const float4 cl = ld_stream(sp.candLi + index), cw = ld_stream(sp.candWi + index);
    Resv r;
    r.Li = mk3(cl.x, cl.y, cl.z); r.wi = mk3(cw.x, cw.y, cw.z); r.dist = cl.w;
    r.M = kReservoirSize; r.W = cw.w;                             // 0 if the shadow ray was blocked (k_shadow)

    if (!first && (reuse & 1)) {                                  // findTemporalNeighbor, restir.cu:20-45
        const int primId = gid;
        const int lastIdx = ld_stream(g.motion + index);
        bool diff = false;
        if (lastIdx < 0) diff = true;
        else if (primId <= kNullPrim) diff = true;
        else if (ld_stream(g.lastPrimId + lastIdx) != primId) diff = true;
        else {
            f3 n = ld3(g.normal + (size_t)index * 3), ln = ld3_stream(g.lastNormal + (size_t)lastIdx * 3);
            float depth = gdepth, pdepth = ld_stream(g.lastDepth + lastIdx);
            if (abs_dot(n, ln) < .9f || gabs(pdepth - depth) > depth * .1f) diff = true;
        }
        Resv t;
        t.Li = splat(0.f); t.wi = splat(0.f); t.dist = 0.f; t.M = 0; t.W = 0.f;
        if (!diff) {
            const float4 a = ld_stream(last.li + lastIdx), b = ld_stream(last.wi + lastIdx);
            t.Li = mk3(a.x, a.y, a.z); t.dist = a.w; t.wi = mk3(b.x, b.y, b.z);
            t.W = ld_stream(last.w + lastIdx); t.M = ld_stream(last.m + lastIdx);
        }
        if (!resv_invalid(t.W)) {
            SamplerT<SOBOL> rng = SamplerT<SOBOL>::resume(sampleSeq, rm.x, looper, kDrawsRis);
            const float u = rng.uniform();
            if (SOBOL) sp.rngMat[index] = make_uint2(rng.word(), rm.y | kTemporalDrewBit);      // the spatial pass resumes one table word further
            else reinterpret_cast<unsigned*>(sp.rngMat + index)[0] = rng.word();
            // preClampedMerge<20> (restir.h:95-102)
            if (r.M > 0) {
                const int cap = (20 - 1) * r.M;
                if (t.M > cap) { t.W *= (float)cap / (float)t.M; t.M = cap; }
            }
            r.W += t.W;
            r.M += t.M;
            if (u * r.W < t.W) { r.Li = t.Li; r.wi = t.wi; r.dist = t.dist; }
        }
    }
    // checkValidity (restir.h:55-59); the temp copy and the stored copy are the same value
    if (resv_invalid(r.W)) { r.W = 0.f; r.M = 0; }
    if (reuse & 2) {
        temp.li[index] = make_float4(r.Li.x, r.Li.y, r.Li.z, r.dist);
        temp.wi[index] = make_float4(r.wi.x, r.wi.y, r.wi.z, 0.f);
        temp.tap[index] = make_float4(r.W, __int_as_float(r.M), __int_as_float(gid), gdepth);
    }
    resv_store(cur, index, r);
}

// ---- phase B: spatial reuse + shade --------------------------------------------------------------
// Tap position: Math::toConcentricDisk (mathUtil.h:128-132) * Radius, then int(x + .5f + p.x)
// (restir.cu:53-55).  Only the TRUNCATED pixel coordinate matters, so the position is first estimated
// with the hardware sqrt / sin / cos (v_sqrt_f32, v_sin_f32, v_cos_f32: a handful of instructions
// against ~90 for correctly rounded sqrtf + sincosf).  The estimate can differ from the exact value by
// at most kTapErr (trig + sqrt error, measured by rs_debug_tap_estimate_error and asserted in the
// tests) plus one rounding of the sum; if an integer lies within that band the exact path runs, and
// the exact path itself re-does the trigonometry in double when its own 2-ulp band straddles an
// integer.  Result: the same pixel as the host evaluation, at a fraction of the instruction count.
constexpr float kTapErr = 2e-5f;

// sin and cos of theta in [0, 2*pi] in double precision: quadrant reduction with a two-term pi/2 and
// the classic degree-13/14 kernels (fdlibm k_sin / k_cos coefficients), < 1 ulp(double).  Rounded to
// float this is the correctly rounded sinf/cosf for all but ~1e-9 of the arguments.  A hand-rolled
// routine instead of OCML's sincos(double) because the latter costs 26 VGPRs (6 instead of 8 waves per
// SIMD for the whole spatial pass) for a path taken by ~0.03 % of the taps.
__device__ __forceinline__ void sincos_2pi(double t, double& sn, double& cs) {
    const double k = rint(t * 6.36619772367581382433e-01);                 // 2/pi
    double r = fma(-k, 1.57079632673412561417e+00, t);                      // pi/2 high (33 bits)
    r = fma(-k, 6.07710050650619224932e-11, r);                             // pi/2 low
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s0 = r + r * z * ps;
    const double c0 = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    sn = (q == 0) ? s0 : (q == 1) ? c0 : (q == 2) ? -s0 : -c0;
    cs = (q == 0) ? c0 : (q == 1) ? -s0 : (q == 2) ? -c0 : s0;
}

// exact evaluation of the tap: correctly rounded cos/sin, then the reference's float arithmetic
__device__ __forceinline__ void disk_tap_exact(float rx, float ry, int x, int y, int& px, int& py) {
    const float Radius = 5.f;
    const float rr = sqrtf(rx);
    const float theta = ry * kPi * 2.0f;
    double sd, cd;
    sincos_2pi((double)theta, sd, cd);
    px = f2i((float)x + .5f + ((float)cd * rr) * Radius);
    py = f2i((float)y + .5f + ((float)sd * rr) * Radius);
}

__device__ __forceinline__ void disk_tap_estimate(float rx, float ry, int x, int y, float& fx, float& fy) {
    const float rr5 = __builtin_amdgcn_sqrtf(rx) * 5.f;
    // v_sin/v_cos take revolutions: theta / 2pi = ry (theta = ry * Pi * 2)
    fx = (float)x + .5f + __builtin_amdgcn_cosf(ry) * rr5;
    fy = (float)y + .5f + __builtin_amdgcn_sinf(ry) * rr5;
}

// The estimate and the exact chain differ by at most kTapErr before the final addition, and that
// addition rounds each to the float grid, so their truncations can differ only if an integer lies
// within kTapErr + ulp(f) of the estimate.
__device__ __forceinline__ bool tap_ambiguous(float f) {
    const float ulp = __uint_as_float((__float_as_uint(f) & 0x7f800000u) - (23u << 23));   // f is >= 0.5 in magnitude here or tiny
    return gabs(f - rintf(f)) <= kTapErr + ((gabs(f) >= 1.f) ? ulp : 1.2e-7f);
}

__device__ __forceinline__ void disk_tap(float rx, float ry, int x, int y, int& px, int& py) {
    float fx, fy;
    disk_tap_estimate(rx, ry, x, y, fx, fy);
    px = f2i(fx);
    py = f2i(fy);
    if (tap_ambiguous(fx) || tap_ambiguous(fy)) disk_tap_exact(rx, ry, x, y, px, py);
}

// debug / test hook: every state k of the generator gives uniform() = (float)(k - 1) / 2^31, k - 1 in [0, 2^31 - 3]
__global__ void k_sqrt_of_uniform_check(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k <= 0x7ffffffdull; k += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = (float)(unsigned)k / 2147483648.f;
        if (__float_as_int(sqrt_of_uniform(x)) != __float_as_int(sqrtf(x))) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}
// ... and every value the Sobol sampler can return: (float)r * 2^-32 for a 32-bit r is 0 or a float in [2^-32, 1] -- all of them
__global__ void k_sqrt_of_unit_floats_check(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    const unsigned lo = 0x2f800000u, hi = 0x3f800000u;                 // 2^-32 .. 1.0
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k <= (unsigned long long)(hi - lo) + 1ull; k += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = k == (unsigned long long)(hi - lo) + 1ull ? 0.f : __uint_as_float(lo + (unsigned)k);
        if (__float_as_int(sqrt_of_uniform(x)) != __float_as_int(sqrtf(x))) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// debug / test hook: largest |estimate - exact(double)| of the tap offset over n samples
__global__ void k_tap_estimate_error(int n, float* maxErr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float e = 0.f;
    if (i < n) {
        Rng rng = seeded_rng(i, 12345, 0);
        const float rx = rng.uniform(), ry = rng.uniform();
        float fx, fy;
        disk_tap_estimate(rx, ry, 0, 0, fx, fy);
        const double th = (double)(ry * kPi * 2.0f), rr = sqrt((double)rx) * 5.0;
        e = fmaxf(gabs((float)((double)fx - (0.5 + cos(th) * rr))), gabs((float)((double)fy - (0.5 + sin(th) * rr))));
    }
    for (int off = 32; off > 0; off >>= 1) e = fmaxf(e, __shfl_down(e, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(maxErr), __float_as_uint(e));   // e >= 0
}

// one staged pixel: tap record {W, M, id, depth} + G-buffer normal, 32 B
struct __attribute__((aligned(16))) Staged { float4 tap; float nx, ny, nz, pad; };

#ifndef RS_K4_WAVES
#define RS_K4_WAVES 6
#endif
#ifndef RS_K4_TILE_W
#define RS_K4_TILE_W 32
#endif
#ifndef RS_K4_TILE_H
#define RS_K4_TILE_H 16
#endif
constexpr int kBTileW = RS_K4_TILE_W, kBTileH = RS_K4_TILE_H, kBThreads = kBTileW * kBTileH;
constexpr int kBStageW = kBTileW + 2 * kHalo, kBStageH = kBTileH + 2 * kHalo, kBStageN = kBStageW * kBStageH;

__device__ __forceinline__ Staged fetch_staged_global(const GBufView& g, const TempPlanes& temp, int gi) {
    Staged v;
    v.tap = temp.tap[gi];
    const f3 n = ld3(g.normal + (size_t)gi * 3);
    v.nx = n.x; v.ny = n.y; v.nz = n.z; v.pad = 0.f;
    return v;
}

// The per-pixel body of phase B (restir.cu:196-230).  STAGED: neighbour records come from the LDS tile
// `stage` (origin sox, soy); otherwise from global memory.
template <bool STAGED, bool SOBOL>
__device__ __forceinline__ void spatial_pixel(const DevScene& s, const SurfPlanes& sp, const GBufView& g, const ResvPlanes& own,
                                              const TempPlanes& temp, const Staged* stage, int sox, int soy,
                                              float* __restrict__ directIllum, int iter, int looper, bool spatial,
                                              int x, int y, int index, uint2 rm, f3 albedo, f3 prev) {
    const int W = g.width, H = g.height;
    const int mk = (int)rm.y;
    const int kind = mk_kind(mk);

    f3 direct = splat(0.f);
    if (kind == kKindLight) direct = splat(1.f);               // restir.cu:143-146 (baseColor forced to 1)
    if (kind == kKindMiss && s.envTex >= 0) {                  // restir.cu:134-136: looked up by k_primary
        const float4 e = sp.posMat[index];
        direct = mk3(e.x, e.y, e.z);
    }
    if (kind == kKindShaded) {
        SamplerT<SOBOL> rng = SamplerT<SOBOL>::resume(s.sampleSeq, rm.x, looper, kDrawsRis + ((rm.y & kTemporalDrewBit) ? 1 : 0));

        // own reservoir = what phase A published (post-temporal, validity-checked)
        float W0; int M0; int src = index;
        if (spatial) {
            const int cy = y - soy;
            const Staged c = STAGED ? stage[cy * kBStageW + (x - sox)] : fetch_staged_global(g, temp, index);
            W0 = c.tap.x; M0 = __float_as_int(c.tap.y);
            const int idC = __float_as_int(c.tap.z);
            const float dC = c.tap.w;
            const f3 nC = mk3(c.nx, c.ny, c.nz);

            // mergeSpatialNeighborDirect (restir.cu:87-100), written branch-free: a rejected tap merges the
            // default reservoir (W = 0, M = 0), exactly what the reference's `diff ? T() : ...` does
            float aW = 0.f; int aM = 0; int aSrc = -1;
#pragma unroll 1
            for (int i = 0; i < 5; i++) {
                const f2 r2 = rng.uniform2();
                int px, py;
                disk_tap(r2.x, r2.y, x, y, px, py);
                const bool inb = (px >= 0) & (px < W) & (py >= 0) & (py < H) & !((px == x) & (py == y));
                Staged q;
                if (STAGED) {       // taps reach x-4..x+5, y-4..y+5: always inside the staged halo; the clamp keeps garbage in bounds
                    const int lx = iclamp(px - sox, 0, kBStageW - 1);
                    const int ly = iclamp(py - soy, 0, kBStageH - 1);
                    q = stage[ly * kBStageW + lx];
                }
                else q = fetch_staged_global(g, temp, iclamp(py, 0, H - 1) * W + iclamp(px, 0, W - 1));
                const bool same = inb & (__float_as_int(q.tap.z) == idC) & !(dot(nC, mk3(q.nx, q.ny, q.nz)) < .9f) &
                                  !(gabs(dC - q.tap.w) > dC * .1f);
                const float tW = same ? q.tap.x : 0.f;
                const int tM = same ? __float_as_int(q.tap.y) : 0;
                const int tSrc = same ? py * W + px : -1;
                const bool valid = !resv_invalid(tW);                  // `if (!spatial.invalid())`: the draw happens only then
                SamplerT<SOBOL> adv = rng;
                const float u = adv.uniform();
                if (valid) rng = adv;
                const float nW = aW + tW;                              // Reservoir::merge (restir.h:61-68)
                aSrc = (valid & (u * nW < tW)) ? tSrc : aSrc;
                aW = valid ? nW : aW;
                aM = valid ? aM + tM : aM;
            }
            if (!resv_invalid(aW) && !resv_invalid(W0)) {
                const float u = rng.uniform();
                W0 += aW; M0 += aM;
                if (u * W0 < aW) src = aSrc;
            }
        }
        else {
            // No spatial pass: `own` is the buffer phase A published this frame (validity-checked copy of
            // the post-temporal reservoir).  The reference shades the unchecked one; they differ only
            // when it was invalid, and then both give direct = 0 (0/0 -> NaN -> cleared below).
            W0 = own.w[index]; M0 = own.m[index];
        }

        if (!resv_invalid(W0)) {
            f3 Li = splat(0.f), wi = splat(0.f);
            if (src >= 0) {                                            // the surviving sample: one 32-byte gather
                const float4 a = spatial ? temp.li[src] : own.li[src], b = spatial ? temp.wi[src] : own.wi[src];      // (cached loads: neighbouring pixels share lines)
                Li = mk3(a.x, a.y, a.z); wi = mk3(b.x, b.y, b.z);
            }
            const int type = mk_type(mk);
            f3 wo = splat(0.f), norm = splat(0.f);
            float metallic = 0.f, roughness = 0.f;
            if (type == 1) {                                           // only the metallic BSDF looks at n, wo and the two scalars
                const float4 w4 = sp.wo[index], n4 = sp.norm[index];
                wo = mk3(w4.x, w4.y, w4.z); norm = mk3(n4.x, n4.y, n4.z);
                metallic = n4.w; roughness = w4.w;
            }
            const f3 LiBSDF = Li * eval_bsdf(type, splat(1.f), metallic, roughness, norm, wo, wi);
            direct = ((LiBSDF / luminance(LiBSDF)) * W0) / (float)M0;       // restir.cu:220-221
        }
        if (any_nan_or_inf(direct)) direct = splat(0.f);
    }
    direct = direct * albedo;
    // (prev * iter + direct) / (iter + 1), restir.cu:230; for iter == 0 the division by 1.f is the identity
    const f3 acc = prev * (float)iter + direct;
    st3(directIllum + (size_t)index * 3, iter == 0 ? acc : acc / (float)(iter + 1));
}

#ifndef RS_K4_BAND_TILES
#define RS_K4_BAND_TILES 48
#endif
constexpr int kBandTiles = RS_K4_BAND_TILES;
template <bool SOBOL>
__device__ __forceinline__ void spatial_shade_block(Staged* stage, const DevScene& s, const SurfPlanes& sp, const GBufView& g, const ResvPlanes& own, const TempPlanes& temp,
                                                    float* __restrict__ directIllum, int iter, int looper, int reuse,
                                                    int y0, int y1, int tilesX, int numTiles) {
    RS_SETPRIO(RS_PRIO_STREAM);

    // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a
    // contiguous run of tiles so that neighbouring tiles' halos are served by the same L2.
    int tile = blockIdx.x;
    {
        const int q = numTiles / 8, rem = numTiles % 8, xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }

    // ... and inside that run the tiles go band by band: a tile re-reads the five halo rows its upper neighbour staged one ROW OF TILES
    // earlier, and a row of tiles is 30.6 KB x tilesX of staged records -- 1.8 MB at 1920 pixels, which the XCD's 4 MB L2 still holds, 3.7 MB
    // at 3840, which it does not (counter traffic 1.46 x the algorithmic bytes against 1.21 x at 1080p).  Frames wider than kBandTiles tiles
    // are therefore swept in vertical bands of at most that many tiles, each band top to bottom (the order changes nothing but cache hits).
    int tcol = tile % tilesX, trow = tile / tilesX;
    if (tilesX > kBandTiles) {
        const int tilesY = numTiles / tilesX, bands = (tilesX + kBandTiles - 1) / kBandTiles, bw = (tilesX + bands - 1) / bands;
        const int full = bw * tilesY, last = (bands - 1) * full;
        const int b = tile >= last ? bands - 1 : tile / full, r = tile - b * full, w = tile >= last ? tilesX - (bands - 1) * bw : bw;
        trow = r / w; tcol = b * bw + r % w;
    }
    const int ox = tcol * kBTileW, oy = y0 + trow * kBTileH;
    const int W = g.width, H = g.height;
    const bool spatial = (reuse & 2) != 0;

    // own-pixel loads are issued before the staging so that their latency overlaps it
    const int tx = threadIdx.x % kBTileW, ty = threadIdx.x / kBTileW;
    const int x = ox + tx, y = oy + ty;
    const bool inside = x < W && y < y1;
    const int index = inside ? y * W + x : 0;
    uint2 rm = make_uint2(0u, 0u);
    f3 albedo = splat(0.f), prev = splat(0.f);
    if (inside) {                                           // read once: streaming, the staged records below are what should stay cached
        const unsigned long long rmBits = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(sp.rngMat + index));
        rm = make_uint2((unsigned)rmBits, (unsigned)(rmBits >> 32));
        albedo = ld3_stream(g.albedo + (size_t)index * 3);
        prev = ld3_stream(directIllum + (size_t)index * 3);
    }

    if (spatial) {
        for (int e = threadIdx.x; e < kBStageN; e += kBThreads) {
            const int sx = ox - kHalo + (e % kBStageW), sy = oy - kHalo + (e / kBStageW);
            Staged v;
            v.tap = make_float4(0.f, 0.f, __int_as_float(-3), 0.f);     // id -3 matches nothing
            v.nx = v.ny = v.nz = v.pad = 0.f;
            if (sx >= 0 && sx < W && sy >= 0 && sy < H) v = fetch_staged_global(g, temp, sy * W + sx);
            stage[e] = v;
        }
        __syncthreads();
    }
    if (!inside) return;
    spatial_pixel<true, SOBOL>(s, sp, g, own, temp, stage, ox - kHalo, oy - kHalo, directIllum, iter, looper, spatial, x, y, index, rm, albedo, prev);
}
template <bool SOBOL>
__global__ void __launch_bounds__(kBThreads, RS_K4_WAVES) k_spatial_shade(DevScene s, SurfPlanes sp, GBufView g, ResvPlanes own, TempPlanes temp,
                                                             float* __restrict__ directIllum, int iter, int looper, int reuse,
                                                             int y0, int y1, int tilesX, int numTiles) {
    __shared__ Staged stage[kBStageN];
    spatial_shade_block<SOBOL>(stage, s, sp, g, own, temp, directIllum, iter, looper, reuse, y0, y1, tilesX, numTiles);
}
// The same pass under another name: the launches a measurement makes for itself (bench.py's twenty back-to-back launches behind
// `roofline.kernel_us`; rs_restir_set_probe), so that a kernel trace of the run tells them from the launches of the frames.
template <bool SOBOL>
__global__ void __launch_bounds__(kBThreads, RS_K4_WAVES) k_spatial_shade_probe(DevScene s, SurfPlanes sp, GBufView g, ResvPlanes own, TempPlanes temp,
                                                                   float* __restrict__ directIllum, int iter, int looper, int reuse,
                                                                   int y0, int y1, int tilesX, int numTiles) {
    __shared__ Staged stage[kBStageN];
    spatial_shade_block<SOBOL>(stage, s, sp, g, own, temp, directIllum, iter, looper, reuse, y0, y1, tilesX, numTiles);
}

// (A rolling-window form of this pass -- a block walks a column of tiles and keeps the shared halo rows in a ring, 1.45 staged records
// per pixel instead of 2.13 -- was built and measured in round 3: bit-exact and slower, 54-63 us against 48; EXPERIMENTS.md, commit da6e82f.)

}  // namespace

// ================================================================================================
// host side: ReSTIRInit / Free / Reset / Direct (src/restir.cu:418-446,478-518)
// ================================================================================================
namespace {

int alloc_planes(ResvPlanes& p, size_t n) {
    RS_TRY(rs_dev_alloc(&p.li, n)); RS_TRY(rs_dev_alloc(&p.wi, n));
    RS_TRY(rs_dev_alloc(&p.w, n));  RS_TRY(rs_dev_alloc(&p.m, n));
    RS_HIP(hipMemset(p.li, 0, n * 16)); RS_HIP(hipMemset(p.wi, 0, n * 16));      // cudaMemset(.., 0, ..) restir.cu:483-489
    RS_HIP(hipMemset(p.w, 0, n * 4));   RS_HIP(hipMemset(p.m, 0, n * 4));
    return 0;
}
void free_planes(ResvPlanes& p) { rs_dev_free(p.li); rs_dev_free(p.wi); rs_dev_free(p.w); rs_dev_free(p.m); }

ResvPlanes* pick(rs_restir* r, int which) { return which == 0 ? &r->cur : (which == 1 ? &r->last : nullptr); }

SurfPlanes surf_of(rs_restir* r) {                       // the set of the frame in flight
    const rs_restir::Surf& f = r->surf[r->surfSet];
    SurfPlanes sp;
    sp.posMat = f.posKind; sp.norm = f.norm; sp.wo = f.wo; sp.rngMat = f.rngMat;
    sp.candLi = f.candLi; sp.candWi = f.candWi;
    return sp;
}

int check_frame_args(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g) {
    if (!r || !scene || !cam || !g) return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRDirect: null argument");
    if (scene->ctx != r->ctx || g->ctx != r->ctx) return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRDirect: the scene, the G-buffer and the reservoirs belong to different contexts");
    if (cam->resolution[0] != r->width || cam->resolution[1] != r->height || g->width != r->width || g->height != r->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRDirect: camera / G-buffer size differs from the size given to rs_restir_init");
    return 0;
}

// timing 1: every pass bracketed, all kernels on the library stream; 2: only the spatial pass bracketed (events 3 and 4), the launches
// stay where the overlapped mode puts them -- the pass's duration while other frames' kernels share the CUs
void mark(rs_restir* r, int i) { if (r->timing == 1 || (r->timing == 2 && i >= 3)) (void)hipEventRecord(r->ev[i], rs_stream()); }

}  // namespace

extern "C" {

int rs_restir_free(rs_restir* r) {
    RS_SCOPE(r);
    if (!r) return 0;
    (void)rs_synchronize();                                     // also kernels still running on the auxiliary stream
    free_planes(r->cur); free_planes(r->last);
    rs_dev_free(r->temp.li); rs_dev_free(r->temp.wi); rs_dev_free(r->temp.tap);
    for (auto& f : r->surf) {
        rs_dev_free(f.posKind); rs_dev_free(f.norm); rs_dev_free(f.wo); rs_dev_free(f.rngMat); rs_dev_free(f.candLi); rs_dev_free(f.candWi);
    }
    rs_dev_free(r->dRayCount);
    for (auto& perStream : r->split) for (auto& t : perStream) rs_tile_split_free(&t);
    rs_dev_free(r->indResv[0]); rs_dev_free(r->indResv[1]);
    for (auto& e : r->ev) if (e) (void)hipEventDestroy(e);
    for (auto& pair : r->spatialEv) for (hipEvent_t& e : pair) if (e) (void)hipEventDestroy(e);
    for (auto& e : r->surfFree) if (e) (void)hipEventDestroy(e);
    for (auto& e : r->tuneEv) if (e) (void)hipEventDestroy(e);
    if (r->auxFork) (void)hipEventDestroy(r->auxFork);
    if (r->auxDone) (void)hipEventDestroy(r->auxDone);
    delete r;
    return 0;
}

int rs_restir_init(int width, int height, rs_restir** out) {
    if (!out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_init: bad size");
    *out = nullptr;
    rs_restir* r = new rs_restir();
    r->ctx = rs_ctx();
    rs_ctx_scope scope(r->ctx);
    r->width = width; r->height = height;
    const size_t n = (size_t)width * height;
    int e = 0;
    if (!e) e = alloc_planes(r->cur, n);
    if (!e) e = alloc_planes(r->last, n);
    if (!e) e = rs_dev_alloc(&r->temp.li, n);
    if (!e) e = rs_dev_alloc(&r->temp.wi, n);
    if (!e) e = rs_dev_alloc(&r->temp.tap, n);
    if (!e) e = rs_check_hip(hipMemset(r->temp.li, 0, n * 16), "memset");
    if (!e) e = rs_check_hip(hipMemset(r->temp.wi, 0, n * 16), "memset");
    if (!e) e = rs_check_hip(hipMemset(r->temp.tap, 0, n * 16), "memset");
    for (auto& f : r->surf) {
        if (!e) e = rs_dev_alloc(&f.posKind, n);
        if (!e) e = rs_dev_alloc(&f.norm, n);
        if (!e) e = rs_dev_alloc(&f.wo, n);
        if (!e) e = rs_dev_alloc(&f.rngMat, n);
        if (!e) e = rs_dev_alloc(&f.candLi, n);
        if (!e) e = rs_dev_alloc(&f.candWi, n);
        if (!e) e = rs_check_hip(hipMemset(f.rngMat, 0, n * 8), "memset");
    }
    for (auto& ev : r->surfFree) if (!e) e = rs_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    for (auto& ev : r->tuneEv) if (!e) e = rs_check_hip(hipEventCreate(&ev), "hipEventCreate");
    if (!e) e = rs_check_hip(hipEventCreateWithFlags(&r->auxFork, hipEventDisableTiming), "hipEventCreate");
    if (!e) e = rs_check_hip(hipEventCreateWithFlags(&r->auxDone, hipEventDisableTiming), "hipEventCreate");
    if (!e) e = rs_dev_alloc(&r->dRayCount, 2 * (size_t)kRaySlots * kRaySub * kRayStride);     // working slots, then published slots
    if (!e) e = rs_check_hip(hipMemset(r->dRayCount, 0, 16 * (size_t)kRaySlots * kRaySub * kRayStride), "memset");
    for (auto& ev : r->ev) if (!e) e = rs_check_hip(hipEventCreate(&ev), "hipEventCreate");
    // the clears above are enqueued on the default stream, which the auxiliary streams are not ordered after: finish them
    // before the first primary-ray kernel can be launched there
    if (!e) e = rs_check_hip(hipDeviceSynchronize(), "rs_restir_init");
    if (e) { rs_restir_free(r); return e; }
    *out = r;
    return 0;
}

int rs_restir_reset(rs_restir* r) {
    RS_SCOPE(r);
    if (!r) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_reset: null");
    r->firstFrame = true;
    return 0;
}

int rs_restir_enable_timing(rs_restir* r, int enable) {
    RS_SCOPE(r);
    if (!r) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_enable_timing: null");
    r->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    if (r->timing == 2) {
        for (auto& pair : r->spatialEv) for (hipEvent_t& e : pair) if (!e) RS_HIP(hipEventCreate(&e));
        r->spatialNext = 0;
    }
    return 0;
}

// the launches of the spatial pass from now on go out as k_spatial_shade_probe (same code, another name in a kernel trace): 1 on, 0 off
int rs_restir_set_probe(rs_restir* r, int enable) {
    RS_SCOPE(r);
    if (!r) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_set_probe: null");
    r->probe = enable != 0;
    return 0;
}

// rs_restir_enable_timing(r, 2): the durations (ms) of the spatial pass in the last frames, oldest first, at most `capacity` and at most the
// ring's 256; *count = how many.  Waits for the library stream.
int rs_restir_spatial_times(rs_restir* r, float* ms, int capacity, int* count) {
    RS_SCOPE(r);
    if (!r || !ms || !count || capacity < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_spatial_times: bad argument");
    *count = 0;
    if (r->timing != 2) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_spatial_times: rs_restir_enable_timing(r, 2) is not in force");
    RS_HIP(hipStreamSynchronize(rs_stream()));
    int n = r->spatialNext < rs_restir::kSpatialRing ? r->spatialNext : rs_restir::kSpatialRing;
    if (n > capacity) n = capacity;
    for (int i = 0; i < n; i++) {
        const int slot = (r->spatialNext - n + i) % rs_restir::kSpatialRing;
        RS_HIP(hipEventElapsedTime(&ms[i], r->spatialEv[slot][0], r->spatialEv[slot][1]));
    }
    *count = n;
    return 0;
}

}  // extern "C"

namespace {
// RIS over the light table for rows [y0, y1) on stream st; alone: nothing runs next to it (picks the alias-in-LDS form for large tables)
int launch_ris(const rs_scene* scene, const SurfPlanes& sp, int W, int y0, int y1, int looper, bool sobol, hipStream_t st, bool alone) {
    const int npx = (y1 - y0) * W;
    // The LDS form runs one 1024-thread block per copy of the table: a launch of a few dozen blocks leaves most CUs idle and lasts as
    // long as one block.  Below 64 Ki pixels the table is read from global memory by 256-thread blocks, which spread evenly.  (Round 2
    // drew the line at 384 Ki pixels -- a 1/8 strip of 1080p 0.241 -> 0.231 ms per frame with the global table; measured again in round 5
    // through rs_strips_frame the LDS form wins on every rank of that split, 0.193 -> 0.191 ms on the heaviest strip and 0.149 -> 0.130 on
    // the lightest, whose chain is mostly RIS: profiles/r05_ab_strip_knobs.log.)
    // Alone the alias-in-LDS form is a third faster (config 5: 645 -> 455 us); inside overlapped frames it is slower (1.88 -> 1.95 ms per
    // frame: one 1024-thread block with 82 KB of LDS per CU keeps the other streams' kernels off that CU), so it is taken when the
    // kernels run one after the other on the library stream only (`alone`; A/B in profiles/r03_ab_config5_ris_alias_lds.log).
    const int risGlobalBelow = rs_ris_global_below();           // 64 Ki pixels unless rs_set_ris_table_pixels says otherwise
    if (scene->numLights > 0 && scene->numLights <= kRisLdsLights && npx >= risGlobalBelow && scene->envMapTexId < 0)
        // (one block per CU instead of two -- half of the wave slots left to the latency-bound kernels of the other streams -- measured
        // slower: frame 1.088 -> 1.142 ms, profiles/r03_ab_ris_blocks_per_cu.log)
        RS_LAUNCH1(k_ris_lds, sobol, dim3((npx + kRisThreads - 1) / kRisThreads), dim3(kRisThreads), st, scene->dev, sp, W, y0, y1, looper);
    else if (scene->envMapTexId < 0 && scene->numLights > kRisLdsLights && scene->numLights <= kRisAliasLdsLights && npx >= risGlobalBelow && alone) {
        const size_t lds = (size_t)scene->numLights * sizeof(AliasRec);
        static const bool ldsAllowed = []{      // more than 64 KB of dynamic LDS is opt-in
            const bool a = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ris_alias_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kRisAliasLdsLights * (int)sizeof(AliasRec)) == hipSuccess;
            const bool b = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ris_alias_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kRisAliasLdsLights * (int)sizeof(AliasRec)) == hipSuccess;
            (void)hipGetLastError();
            return a && b; }();
        (void)ldsAllowed;
        if (sobol) hipLaunchKernelGGL(k_ris_alias_lds<true>, dim3((npx + kRisThreads - 1) / kRisThreads), dim3(kRisThreads), lds, st, scene->dev, sp, W, y0, y1, looper);
        else hipLaunchKernelGGL(k_ris_alias_lds<false>, dim3((npx + kRisThreads - 1) / kRisThreads), dim3(kRisThreads), lds, st, scene->dev, sp, W, y0, y1, looper);
    }
    else                               // the environment map is one more light (scene.h:400-403)
        RS_LAUNCH2(k_ris, scene->envMapTexId >= 0, sobol, dim3((npx + 255) / 256), dim3(256), st, scene->dev, sp, W, y0, y1, looper);
    return 0;
}

// `last`: the call ends with rs_after_launch (which synchronises in synchronous mode); ReSTIRDirect passes false for its two
// inner calls and synchronises once at its end
int phase_a_impl(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g, int looper, int reuse, int y0, int y1, bool last) {
    RS_TRY(check_frame_args(r, scene, cam, g));
    RS_TRY(rs_check_looper(scene, looper, "ReSTIRDirect"));
    const bool sobol = scene->dev.sampleSeq != nullptr;
    r->looper = looper;                                         // phase B resumes the samplers of this call's pixels (Sobol: the table row)
    if (y0 < 0) y0 = 0;
    if (y1 > r->height) y1 = r->height;
    r->raySlot = (r->raySlot + 1) % kRaySlots;                  // one counter slot per call (ring): zero here, k_temporal moves it out
    unsigned long long* rayCounter = r->dRayCount + (size_t)r->raySlot * kRaySub * kRayStride;
    unsigned long long* rayDone = rayCounter + (size_t)kRaySlots * kRaySub * kRayStride;
    // The primary-ray and RIS kernels read nothing the previous frame's temporal / spatial passes write and fill this
    // frame's own set of surface planes: in asynchronous mode they go to an auxiliary stream, ordered after the frame
    // that last used the set (or, for a second call within one frame, after everything enqueued so far), and the library
    // stream joins them before the temporal pass.  Their heavy-tile tails then overlap the other frame's passes.
    const rs_context* plan = rs_stream_plan();                  // rs_set_stream_plan (defaults: two chain streams, small launches on three, shadow rays of large launches on the library stream)
    const bool parityStreams = plan->chainStreams == 2;
    const bool asyncMode = r->timing != 1 && rs_aux_stream(1) != nullptr;
    const int W = r->width;
    const int tilesX = (W + 31) / 32, tilesY = (y1 > y0 ? y1 - y0 + 7 : 0) / 8;
    // A render of this frame that rs_gbuffer_render_rows deferred (asynchronous mode) can be launched here, in ONE launch with the
    // primary rays (k_gbuffer_primary): same scene and camera, rows that contain the rows shaded here.
    //  * A launch that fills the chip at least three times over: ~5 % less work than two launches, longer waves; the frame period is
    //    measured both ways once per scene and the faster form kept (full 1080p frame: the fused launch, by 0.5 %).
    //  * A smaller launch -- a strip -- lasts as long as its slowest wave, and what bounds its frame rate is the length of the chain
    //    primary rays -> RIS -> shadow rays over the number of chains in flight.  With two chains the fused launch loses (its slowest
    //    wave: 0.25 ms against 0.18 on a 1/8 strip), but it leaves the render's stream idle, and with that stream as a THIRD chain
    //    it wins: 8 strips of 1080p 5.96x -> 6.5x (rs_set_stream_plan(-1, 0, -1): two chains and a separate render).
    // Whenever the launch is fused the frame's chain is one of three (a full frame gains another 0.9 % from the third).
    const bool smallChains = plan->smallChains != 0;
    const rs_gbuffer::Deferred& d = g->deferred;
    const int fuseMode = rs_fuse_mode();
    const bool fusable = asyncMode && fuseMode != 0 && y1 > y0 && d.valid && d.scene == scene && std::memcmp(&d.cam, cam, sizeof(rs_camera)) == 0 &&
                         d.y0 <= y0 && d.y1 >= y1;
    const bool large = fusable && (long long)tilesX * ((d.y1 - d.y0 + 7) / 8) * 4 >= kFuseMinWaves;
    const bool small = fusable && !large && smallChains && parityStreams && fuseMode == 3 && r->phaseACalls == 0;
    bool fuse = fusable && (small || large || fuseMode == 2);
    // (with a denoise stream the render's own stream is that stream: a separate render would queue behind the previous frame's filter)
    const bool denoiseStream = asyncMode && rs_ctx()->denoiseMode == 1;
    if (fuse && large && fuseMode == 3 && !denoiseStream) {        // measured choice (end_frame advances the measurement)
        if (r->tuneSceneId != scene->id) { r->tuneSceneId = scene->id; r->tuneFrame = 0; r->tuneChoice = -1; }
        r->tuneCounted = true;
        fuse = r->tuneChoice >= 0 ? r->tuneChoice == 1 : (r->tuneFrame >= kTuneB && r->tuneFrame < kTuneD);
    }
    int kThreeStreams[rs_restir::kSmallChains];                 // the chain streams first, the render's stream (idle after a fused launch) last
    for (int i = 0; i < rs_restir::kSmallChains; i++) kThreeStreams[i] = i < 2 ? 1 + i : i == 2 ? 0 : i;
    const bool three = fuse && parityStreams && r->phaseACalls == 0;
    // (a context that keeps another stream busy next to the frames -- the strip driver with its transfers on a stream of their own, the
    // denoise stream -- leaves room for two chains, or one: four streams that hand events to each other is what the device runs side by
    // side, rs_chains_in_flight)
    const int inFlight = rs_chains_in_flight();
    const int chainSlot = three ? kThreeStreams[inFlight >= rs_restir::kSmallChains ? r->smallChain : inFlight == 2 ? r->chain : 0] : 0;
    const hipStream_t aux = asyncMode ? rs_aux_stream(three ? chainSlot : (parityStreams && inFlight >= 2) ? 1 + r->chain : 1) : nullptr;
    r->lastFused = fuse ? 1 : 0;
    r->lastChains = !aux ? 0 : three ? (inFlight < rs_restir::kSmallChains ? inFlight : rs_restir::kSmallChains) : (parityStreams && inFlight >= 2) ? rs_restir::kChains : 1;
    const hipStream_t st = aux ? aux : rs_stream();
    const int splitSlot = !aux ? 0 : 1 + (three ? chainSlot : (parityStreams && inFlight >= 2) ? 1 + r->chain : 1);     // the hints of the stream this launch goes to (rs_tilesplit.h)
    const int splitCall = r->phaseACalls < 2 ? r->phaseACalls : 2;
    // One frame at a time -- a caller that waits for every frame before it enqueues the next (preview.cpp:337-361) -- has nothing running
    // next to this frame's kernels, like the synchronous mode: a launch lasts as long as its longest tile and RIS has the CUs to itself, so
    // it takes that mode's forms (heavy tiles split four ways, the alias table in LDS for large light sets).  Asked of the previous frame's
    // end event, never waited for: config 5 one frame in flight 3.34 -> 2.6 ms (synchronous 2.84).
    bool idle = false;
    if (aux && r->phaseACalls == 0) {
        const int prevSet = (r->surfSet + rs_restir::kSurfSets - 1) % rs_restir::kSurfSets;
        const bool finished = !r->surfFreeValid[prevSet] || hipEventQuery(r->surfFree[prevSet]) == hipSuccess;
        r->idleStreak = finished ? (r->idleStreak < 2 ? r->idleStreak + 1 : 2) : 0;      // (two frames in a row: a pipelined caller whose device catches up once keeps its forms)
        r->idleFrame = idle = r->idleStreak >= 2;
    }
    else if (aux) idle = r->idleFrame;
    if (aux) {
        if (r->phaseACalls > 0) {
            RS_HIP(hipEventRecord(r->auxFork, rs_stream()));
            RS_HIP(hipStreamWaitEvent(aux, r->auxFork, 0));
        }
        else if (r->surfFreeValid[r->surfSet]) RS_HIP(hipStreamWaitEvent(aux, r->surfFree[r->surfSet], 0));
    }
    r->phaseACalls++;
    if (y1 <= y0) {
        RS_HIP(hipMemsetAsync(rayDone, 0, 8 * kRaySub * kRayStride, st));
        if (aux) { RS_HIP(hipEventRecord(r->auxDone, aux)); RS_HIP(hipStreamWaitEvent(rs_stream(), r->auxDone, 0)); }
        return 0;
    }
    const SurfPlanes sp = surf_of(r);
    const CamParams cp = rs_make_cam_params(cam);
    mark(r, 0);
    if (fuse) {
        RS_TRY(rs_gbuffer_order_before_render(g, aux));
        rs_gbuffer_deferred_taken(g);
        const int c = g->cur();
        const GBufWrite gw{ g->albedo[c], g->motion[c], g->normal[c], g->primId[c], g->depth[c] };
        const int gTilesY = (d.y1 - d.y0 + 3) / 4;                // 8x4-pixel tiles: two rays per pixel fill the wave
        const CamParams lp = rs_make_cam_params(&d.lastCam);
        TileSplit ts; int helpers = 0;
        RS_TRY(rs_tile_split_prepare(&r->split[splitSlot][splitCall], ((((long long)1 << 20 | d.y0) << 20 | d.y1) << 12 | tilesX) ^ ((long long)(y0 * 4099 + y1) << 44), tilesX * 4 * gTilesY, tilesX * gTilesY, (!aux || idle) ? 1 : ((long long)tilesX * gTilesY * 4 < kSplitSmallWaves ? 2 : 0), st, &ts, &helpers));
        if (ts.base) RS_LAUNCH2(k_gbuffer_primary_split, scene->textured, sobol, dim3(helpers + tilesX * gTilesY), dim3(256), st, scene->dev, cp, lp, gw, sp, looper, d.y0, d.y1, y0, y1, tilesX, rayCounter, ts);
        else RS_LAUNCH2(k_gbuffer_primary, scene->textured, sobol, dim3(tilesX * gTilesY), dim3(256), st, scene->dev, cp, lp, gw, sp, looper, d.y0, d.y1, y0, y1, tilesX, rayCounter);
        RS_HIP(hipEventRecord(g->doneEv, aux));              // the planes are ready when this kernel is
        g->pending = true;
    }
    else {
        TileSplit ts; int helpers = 0;
        RS_TRY(rs_tile_split_prepare(&r->split[splitSlot][splitCall], (((long long)y0 << 20 | y1) << 12 | tilesX), tilesX * 4 * tilesY, tilesX * tilesY, (!aux || idle) ? 1 : ((long long)tilesX * tilesY * 4 < kSplitSmallWaves ? 2 : 0), st, &ts, &helpers));
        if (ts.base) RS_LAUNCH2(k_primary_split, scene->textured, sobol, dim3(helpers + tilesX * tilesY), dim3(256), st, scene->dev, cp, sp, looper, y0, y1, tilesX, rayCounter, ts);
        else RS_LAUNCH2(k_primary, scene->textured, sobol, dim3(tilesX * tilesY), dim3(256), st, scene->dev, cp, sp, looper, y0, y1, tilesX, rayCounter);
    }
    mark(r, 1);
    const int npx = (y1 - y0) * W;
    RS_TRY(launch_ris(scene, sp, W, y0, y1, looper, sobol, st, !aux || idle));
    mark(r, 2);
    // The shadow rays of a launch that fills the chip several times over go to the library stream, behind the previous frame's
    // spatial pass: every stream then has slack against the frame period and three or four kernels are in flight at any time,
    // which is what a frame bound by VALU issue needs (1080p: 1.277 -> 1.245 ms).  A small launch -- a strip -- lasts as long as its
    // slowest wave, and there the library stream is the one chain that links consecutive frames: its shadow rays stay on the
    // frame's own chain (8 strips of 1080p: 0.235 ms against 0.270).  rs_set_stream_plan(-1, -1, 0 / 1): never / always.
    const int shadowOnMain = plan->shadowOnMain;
    const bool shadowMain = aux && (shadowOnMain == 1 || (shadowOnMain == 2 && (long long)tilesX * tilesY * 4 >= kFuseMinWaves));
    if (!shadowMain) hipLaunchKernelGGL(k_shadow, dim3(tilesX * tilesY), dim3(256), 0, st, scene->dev, sp, W, y0, y1, tilesX);
    if (aux) {
        RS_TRY(rs_check_hip(hipGetLastError(), "ReSTIR Direct (primary / RIS / shadow rays)"));
        RS_HIP(hipEventRecord(r->auxDone, aux));
        RS_HIP(hipStreamWaitEvent(rs_stream(), r->auxDone, 0));
    }
    if (shadowMain) hipLaunchKernelGGL(k_shadow, dim3(tilesX * tilesY), dim3(256), 0, rs_stream(), scene->dev, sp, W, y0, y1, tilesX);
    RS_TRY(rs_gbuffer_join(g));                                 // first consumer of the G-buffer planes
    RS_LAUNCH1(k_temporal, sobol, dim3((npx + 255) / 256), dim3(256), rs_stream(), sp, gbuf_view(g),
               r->last, r->cur, r->temp, scene->dev.sampleSeq, looper, r->firstFrame ? 1 : 0, reuse, y0 * W, y1 * W, rayCounter, rayDone);
    mark(r, 3);
    return last ? rs_after_launch("ReSTIR Direct (phase A)") : rs_check_hip(hipGetLastError(), "ReSTIR Direct (phase A)");
}

int phase_b_impl(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                 float* devDirectIllum, int iter, int reuse, int y0, int y1, bool last) {
    RS_TRY(check_frame_args(r, scene, cam, g));
    if (!devDirectIllum) return rs_fail(RS_ERR_INVALID_ARGUMENT, "ReSTIRDirect: null radiance buffer");
    if (y0 < 0) y0 = 0;
    if (y1 > r->height) y1 = r->height;
    if (y1 <= y0) return 0;
    const int tilesX = (r->width + kBTileW - 1) / kBTileW, tilesY = (y1 - y0 + kBTileH - 1) / kBTileH;
    const int numTiles = tilesX * tilesY;
    RS_TRY(rs_gbuffer_join(g));
    RS_TRY(rs_denoise_order(devDirectIllum));                   // the previous frame's filter may still be reading the image on the denoise stream
    // timing 2: the pass of every frame between two events of a ring, never waited for here (rs_restir_spatial_times reads them later)
    const int slot = r->timing == 2 ? r->spatialNext % rs_restir::kSpatialRing : -1;
    if (slot >= 0) RS_HIP(hipEventRecord(r->spatialEv[slot][0], rs_stream()));
    if (r->probe)
        RS_LAUNCH1(k_spatial_shade_probe, scene->dev.sampleSeq != nullptr, dim3(numTiles), dim3(kBThreads), rs_stream(), scene->dev, surf_of(r), gbuf_view(g),
                   r->cur, r->temp, devDirectIllum, iter, r->looper, reuse, y0, y1, tilesX, numTiles);
    else
        RS_LAUNCH1(k_spatial_shade, scene->dev.sampleSeq != nullptr, dim3(numTiles), dim3(kBThreads), rs_stream(), scene->dev, surf_of(r), gbuf_view(g),
                   r->cur, r->temp, devDirectIllum, iter, r->looper, reuse, y0, y1, tilesX, numTiles);
    if (slot >= 0) { RS_HIP(hipEventRecord(r->spatialEv[slot][1], rs_stream())); r->spatialNext++; }
    mark(r, 4);
    return last ? rs_after_launch("ReSTIR Direct (phase B)") : rs_check_hip(hipGetLastError(), "ReSTIR Direct (phase B)");
}
}  // namespace

extern "C" {

int rs_restir_phase_a(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                      int looper, int reuse, int y0, int y1) {
    RS_SCOPE(r);
    return phase_a_impl(r, scene, cam, g, looper, reuse, y0, y1, true);
}

int rs_restir_phase_b(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                      float* devDirectIllum, int iter, int reuse, int y0, int y1) {
    RS_SCOPE(r);
    return phase_b_impl(r, scene, cam, g, devDirectIllum, iter, reuse, y0, y1, true);
}

// what the last phase-A call actually launched: *fused = 1 the render in the primary rays' launch, 0 its own launch (or no render
// pending); *chains = number of auxiliary streams the frames' chains take in turn (0: library stream only)
int rs_restir_last_launch(const rs_restir* r, int* fused, int* chains) {
    RS_SCOPE(r);
    if (!r || !fused || !chains) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_last_launch: null argument");
    *fused = r->lastFused; *chains = r->lastChains;
    return 0;
}

// 0 two launches, 1 one fused launch, -1 still measuring, -2 nothing measured (no frame so far had a launch the measurement applies
// to: synchronous launches, a forced mode, launches below three rounds of wave slots -- those are fused without one)
int rs_restir_launch_choice(const rs_restir* r, int* choice) {
    RS_SCOPE(r);
    if (!r || !choice) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_launch_choice: null argument");
    *choice = r->tuneChoice >= 0 ? r->tuneChoice : (r->tuneFrame > 0 || r->tuneCounted) ? -1 : -2;
    return 0;
}

int rs_restir_end_frame(rs_restir* r) {
    RS_SCOPE(r);
    if (!r) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_end_frame: null");
    ResvPlanes t = r->cur; r->cur = r->last; r->last = t;       // std::swap(devDirectReservoir, devLastDirectReservoir)
    r->firstFrame = false;
    // every reader of this frame's surface planes has been enqueued: the set is free for the frame after the next one
    if (!rs_sync_enabled()) { RS_HIP(hipEventRecord(r->surfFree[r->surfSet], rs_stream())); r->surfFreeValid[r->surfSet] = true; }
    else r->surfFreeValid[r->surfSet] = false;
    r->surfSet = (r->surfSet + 1) % rs_restir::kSurfSets;
    r->chain = (r->chain + 1) % rs_restir::kChains;
    r->smallChain = (r->smallChain + 1) % rs_restir::kSmallChains;
    r->phaseACalls = 0;
    // the measurement of rs_fuse_mode() == 3: time stamps on the library stream where frames kTuneA .. kTuneD end (two launches up to
    // kTuneB, one fused launch from there to kTuneD; the spans kTuneA-kTuneB and kTuneC-kTuneD are compared); from frame kTuneD on every
    // frame end asks (hipEventQuery, no wait) whether the last stamp has been reached, and the shorter span decides; until then frames take
    // two launches.  A caller that times frames runs kTuneD + 2 frames and a synchronisation first (bench.py does, before its warm-up) and
    // then sees one launch form only.
    if (r->tuneCounted && r->tuneChoice < 0) {
        const int f = ++r->tuneFrame;
        if (f == kTuneA || f == kTuneB || f == kTuneC || f == kTuneD) RS_HIP(hipEventRecord(r->tuneEv[f == kTuneA ? 0 : f == kTuneB ? 1 : f == kTuneC ? 2 : 3], rs_stream()));
        if (f >= kTuneD) {                                          // never a host wait: the stamp is asked for at every frame end until it is there
            float separate = 0.f, fused = 0.f;
            const hipError_t q = hipEventQuery(r->tuneEv[3]);
            if (q == hipSuccess) {
                if (hipEventElapsedTime(&separate, r->tuneEv[0], r->tuneEv[1]) == hipSuccess && hipEventElapsedTime(&fused, r->tuneEv[2], r->tuneEv[3]) == hipSuccess)
                    r->tuneChoice = fused < separate ? 1 : 0;
                else { (void)hipGetLastError(); r->tuneChoice = 0; }
            }
            else if (q != hipErrorNotReady) { (void)hipGetLastError(); r->tuneChoice = 0; }
        }
    }
    r->tuneCounted = false;
    return 0;
}

// (A synchronous ReSTIRDirect as a software pipeline over bands of rows -- primary rays / RIS / shadow rays of consecutive bands on three
// streams -- was built and measured in round 3: bit-exact and slower, 1.40 -> 1.56-1.63 ms; EXPERIMENTS.md, commit 9bc6c62.)

int rs_restir_direct(rs_restir* r, const rs_scene* scene, const rs_camera* cam, const rs_gbuffer* g,
                     float* devDirectIllum, int iter, int looper, int reuse) {
    RS_SCOPE(r);
    RS_TRY(check_frame_args(r, scene, cam, g));
    // one synchronisation for the whole call (synchronous mode); the library's mode itself is not touched, so the two phases
    // stay on the library stream in synchronous mode and use the auxiliary streams in asynchronous mode only
    RS_TRY(phase_a_impl(r, scene, cam, g, looper, reuse, 0, r->height, false));
    RS_TRY(phase_b_impl(r, scene, cam, g, devDirectIllum, iter, reuse, 0, r->height, false));
    RS_TRY(rs_restir_end_frame(r));
    return rs_after_launch("ReSTIR Direct");
}

size_t rs_restir_halo_bytes(const rs_restir* r, int rows) {
    RS_SCOPE(r);
    return r ? (size_t)r->width * (size_t)(rows > 0 ? rows : 0) * 48u : 0;      // li 16 + wi 16 + tap 16
}
size_t rs_restir_rows_bytes(const rs_restir* r, int which, int rows) {
    RS_SCOPE(r);
    if (!r || rows < 0) return 0;
    return (size_t)r->width * (size_t)rows * (which == 2 ? 48u : 40u);         // cur/last: li 16 + wi 16 + w 4 + m 4
}

namespace {
struct Span { void* ptr; size_t bytesPerPixel; };
int spans_of(rs_restir* r, int which, Span out[4]) {
    if (which == 2) {
        out[0] = { r->temp.li, 16 }; out[1] = { r->temp.wi, 16 }; out[2] = { r->temp.tap, 16 };
        return 3;
    }
    ResvPlanes* p = pick(r, which);
    if (!p) return 0;
    out[0] = { p->li, 16 }; out[1] = { p->wi, 16 }; out[2] = { p->w, 4 }; out[3] = { p->m, 4 };
    return 4;
}
int copy_rows(rs_restir* r, int which, int y0, int rows, char* buf, bool pack, const char* what) {
    Span sp[4];
    const int k = r ? spans_of(r, which, sp) : 0;
    if (!k || !buf || y0 < 0 || rows < 0 || y0 + rows > r->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, what);
    const size_t n = (size_t)r->width * rows, off = (size_t)y0 * r->width;
    size_t pos = 0;
    for (int i = 0; i < k; i++) {
        char* plane = (char*)sp[i].ptr + off * sp[i].bytesPerPixel;
        const size_t bytes = n * sp[i].bytesPerPixel;
        if (bytes) RS_HIP(hipMemcpyAsync(pack ? (void*)(buf + pos) : (void*)plane, pack ? (const void*)plane : (const void*)(buf + pos), bytes,
                                         hipMemcpyDeviceToDevice, rs_stream()));
        pos += bytes;
    }
    return rs_after_launch(what);
}
}  // namespace

// packed layout of `rows` rows: the planes of the buffer one after the other (cur/last: li, wi, w, m;
// temp: li, wi, tap)
int rs_restir_rows_pack(const rs_restir* r, int which, int y0, int rows, void* devBuffer) {
    RS_SCOPE(r);
    return copy_rows(const_cast<rs_restir*>(r), which, y0, rows, (char*)devBuffer, true, "rs_restir_rows_pack: bad argument");
}
int rs_restir_rows_unpack(rs_restir* r, int which, int y0, int rows, const void* devBuffer) {
    RS_SCOPE(r);
    return copy_rows(r, which, y0, rows, (char*)const_cast<void*>(devBuffer), false, "rs_restir_rows_unpack: bad argument");
}
int rs_restir_halo_pack(const rs_restir* r, int y0, int rows, void* devBuffer) { RS_SCOPE(r); return rs_restir_rows_pack(r, 2, y0, rows, devBuffer); }
int rs_restir_halo_unpack(rs_restir* r, int y0, int rows, const void* devBuffer) { RS_SCOPE(r); return rs_restir_rows_unpack(r, 2, y0, rows, devBuffer); }

int rs_restir_download(const rs_restir* rc, int which, rs_reservoir* host) {
    RS_SCOPE(rc);
    rs_restir* r = const_cast<rs_restir*>(rc);
    if (!r || !host || which < 0 || which > 2) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_download: bad argument");
    const size_t n = (size_t)r->width * r->height;
    std::vector<float4> li(n), wi(n), tap; std::vector<float> w; std::vector<int> m;
    RS_HIP(hipStreamSynchronize(rs_stream()));
    if (which == 2) {
        tap.resize(n);
        RS_HIP(hipMemcpy(li.data(), r->temp.li, n * 16, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(wi.data(), r->temp.wi, n * 16, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(tap.data(), r->temp.tap, n * 16, hipMemcpyDeviceToHost));
    }
    else {
        ResvPlanes* p = pick(r, which);
        w.resize(n); m.resize(n);
        RS_HIP(hipMemcpy(li.data(), p->li, n * 16, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(wi.data(), p->wi, n * 16, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(w.data(), p->w, n * 4, hipMemcpyDeviceToHost));
        RS_HIP(hipMemcpy(m.data(), p->m, n * 4, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; i++) {
        host[i].Li[0] = li[i].x; host[i].Li[1] = li[i].y; host[i].Li[2] = li[i].z;
        host[i].wi[0] = wi[i].x; host[i].wi[1] = wi[i].y; host[i].wi[2] = wi[i].z;
        host[i].dist = li[i].w;
        if (which == 2) { host[i].weight = tap[i].x; std::memcpy(&host[i].numSamples, &tap[i].y, 4); }
        else { host[i].weight = w[i]; host[i].numSamples = m[i]; }
    }
    return 0;
}

int rs_restir_upload(rs_restir* r, int which, const rs_reservoir* host) {
    RS_SCOPE(r);
    if (!r || !host || which < 0 || which > 2) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_upload: bad argument");
    const size_t n = (size_t)r->width * r->height;
    std::vector<float4> li(n), wi(n);
    for (size_t i = 0; i < n; i++) {
        li[i] = make_float4(host[i].Li[0], host[i].Li[1], host[i].Li[2], host[i].dist);
        wi[i] = make_float4(host[i].wi[0], host[i].wi[1], host[i].wi[2], 0.f);
    }
    RS_HIP(hipStreamSynchronize(rs_stream()));
    if (which == 2) {
        std::vector<float4> tap(n);
        RS_HIP(hipMemcpy(tap.data(), r->temp.tap, n * 16, hipMemcpyDeviceToHost));      // keep the G-buffer half
        for (size_t i = 0; i < n; i++) { tap[i].x = host[i].weight; std::memcpy(&tap[i].y, &host[i].numSamples, 4); }
        RS_HIP(hipMemcpy(r->temp.li, li.data(), n * 16, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(r->temp.wi, wi.data(), n * 16, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(r->temp.tap, tap.data(), n * 16, hipMemcpyHostToDevice));
    }
    else {
        ResvPlanes* p = pick(r, which);
        std::vector<float> w(n); std::vector<int> m(n);
        for (size_t i = 0; i < n; i++) { w[i] = host[i].weight; m[i] = host[i].numSamples; }
        RS_HIP(hipMemcpy(p->li, li.data(), n * 16, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(p->wi, wi.data(), n * 16, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(p->w, w.data(), n * 4, hipMemcpyHostToDevice));
        RS_HIP(hipMemcpy(p->m, m.data(), n * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

// test hook: sqrt_of_uniform (rs_surface.h) against the compiler's exactly rounded sqrtf on EVERY value Rng::uniform() can return
int rs_debug_sqrt_of_uniform_mismatches(unsigned long long* mismatches) {
    rs_ctx_scope scope(nullptr);
    if (!mismatches) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_sqrt_of_uniform_mismatches: null");
    unsigned long long* d = nullptr;
    RS_TRY(rs_dev_alloc(&d, 1));
    RS_HIP(hipMemsetAsync(d, 0, 8, rs_stream()));
    hipLaunchKernelGGL(k_sqrt_of_uniform_check, dim3(4096), dim3(256), 0, rs_stream(), d);
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost));
    rs_dev_free(d);
    return 0;
}

// test hook: the same for every value the Sobol sampler can return (0 and all floats in [2^-32, 1])
int rs_debug_sqrt_of_unit_floats_mismatches(unsigned long long* mismatches) {
    rs_ctx_scope scope(nullptr);
    if (!mismatches) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_sqrt_of_unit_floats_mismatches: null");
    unsigned long long* d = nullptr;
    RS_TRY(rs_dev_alloc(&d, 1));
    RS_HIP(hipMemsetAsync(d, 0, 8, rs_stream()));
    hipLaunchKernelGGL(k_sqrt_of_unit_floats_check, dim3(4096), dim3(256), 0, rs_stream(), d);
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost));
    rs_dev_free(d);
    return 0;
}

// test hook: largest error of the hardware-trig tap estimate against a double evaluation (see disk_tap)
int rs_debug_tap_estimate_error(int n, float* maxErr) {
    rs_ctx_scope scope(nullptr);
    if (n <= 0 || !maxErr) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_tap_estimate_error: bad argument");
    float* d = nullptr;
    RS_TRY(rs_dev_alloc(&d, 1));
    RS_HIP(hipMemsetAsync(d, 0, 4, rs_stream()));
    hipLaunchKernelGGL(k_tap_estimate_error, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), n, d);
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(maxErr, d, 4, hipMemcpyDeviceToHost));
    rs_dev_free(d);
    return 0;
}

int rs_restir_ray_count(rs_restir* r, unsigned long long* rays) {
    RS_SCOPE(r);
    if (!r || !rays) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_ray_count: null");
    RS_HIP(hipStreamSynchronize(rs_stream()));
    unsigned long long h[kRaySub * kRayStride];
    RS_HIP(hipMemcpy(h, r->dRayCount + (size_t)(kRaySlots + r->raySlot) * kRaySub * kRayStride, sizeof h, hipMemcpyDeviceToHost));
    *rays = 0;
    for (int i = 0; i < kRaySub; i++) *rays += h[i * kRayStride];
    return 0;
}

int rs_restir_ray_total(rs_restir* r, int frames, unsigned long long* rays) {
    RS_SCOPE(r);
    if (!r || !rays || frames < 0 || frames > kRaySlots) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_ray_total: frames must be in [0, 1024]");
    RS_HIP(hipStreamSynchronize(rs_stream()));
    const size_t per = (size_t)kRaySub * kRayStride;
    std::vector<unsigned long long> h((size_t)kRaySlots * per);
    RS_HIP(hipMemcpy(h.data(), r->dRayCount + (size_t)kRaySlots * per, 8 * h.size(), hipMemcpyDeviceToHost));
    unsigned long long t = 0;
    for (int i = 0; i < frames; i++) {
        const size_t slot = (size_t)((r->raySlot - i) % kRaySlots + kRaySlots) % kRaySlots;
        for (int k = 0; k < kRaySub; k++) t += h[slot * per + (size_t)k * kRayStride];
    }
    *rays = t;
    return 0;
}

int rs_restir_pass_times(rs_restir* r, float ms[4]) {
    RS_SCOPE(r);
    if (!r || !ms) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_pass_times: null");
    if (!r->timing) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_restir_pass_times: timing is not enabled");
    RS_HIP(hipEventSynchronize(r->ev[4]));
    for (int i = 0; i < 4; i++) {
        ms[i] = 0.f;
        if (r->timing == 1 || i == 3) RS_HIP(hipEventElapsedTime(&ms[i], r->ev[i], r->ev[i + 1]));
    }
    return 0;
}

}  // extern "C"
