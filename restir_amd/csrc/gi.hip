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
// (A wavefront form -- path state in SoA queues, one launch per stage and bounce, ray queue bucketed by threaded order, a streaming walk
// that replaces finished rays -- was built in round 4: bit-exact, a tie on the Sponza-class scene and 15-20 % slower on the Bistro-class
// one; the walks cost the same either way.  EXPERIMENTS.md, commits d0141a4 and 220fdf1.)
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

// A three-vector kept in LDS, one column per thread (component k of thread t at p[k * 256 + t]: conflict-free).  The path loop's cold
// state -- the radiance sums, touched once or twice per bounce, and ReSTIR-GI's four recorded points, written once per path -- lived in
// registers the compiler had to spill around the walks (72 VGPRs at 7 blocks per CU; 172-268 bytes of scratch per lane, 310 scratch
// instructions per bounce and wave, every one of them a trip to L2: a CU's 28 waves keep 360 KB of scratch behind a 32 KB L1).  Round 5: 6 / 9
// / 21 columns (pathTraceIndirect / pathTrace / ReSTIR-GI: 21.5 KB per block, 150 of 160 KB at 7 blocks per CU); round 6: with the surface
// state below 19 / 22 / 22 columns = 19.5 / 22.5 / 22.5 KB per block, 136 / 158 / 158 KB per CU -- the kernel uses no other LDS.
#ifndef RS_PATH_COLD_LDS
#define RS_PATH_COLD_LDS 1
#endif
#if RS_PATH_COLD_LDS
struct Cold3 {
    float* p;
    __device__ __forceinline__ f3 get() const { return mk3(p[0], p[256], p[512]); }
    __device__ __forceinline__ void set(f3 v) const { p[0] = v.x; p[256] = v.y; p[512] = v.z; }
    __device__ __forceinline__ void add(f3 v) const { set(get() + v); }
};
#else
struct Cold3 {
    f3 v;
    __device__ __forceinline__ f3 get() const { return v; }
    __device__ __forceinline__ void set(f3 x) { v = x; }
    __device__ __forceinline__ void add(f3 x) { v = v + x; }
};
#endif
// The surface state a bounce carries across its shadow-ray walk -- the material (7 words), the shading normal and wo -- is dead weight
// inside the walk and was the bulk of what the compiler parked in scratch around it.  Path tracing keeps it in LDS as well (13 more
// columns: 22 floats per lane for pathTrace, 19 for pathTraceIndirect and ReSTIR-GI, which is what 7 blocks per CU leave of the 160 KB;
// ReSTIR-GI's recorded points, which took 12 columns in round 5, wait in the output array instead: Glob3 below).
template <bool IN_LDS> struct ColdSurf;
template <> struct ColdSurf<true> {
    float* p;
    __device__ __forceinline__ SurfMat mat() const { SurfMat m; m.type = __float_as_int(p[0]); m.baseColor = mk3(p[256], p[512], p[768]); m.metallic = p[1024]; m.roughness = p[1280]; m.ior = p[1536]; return m; }
    __device__ __forceinline__ int type() const { return __float_as_int(p[0]); }
    __device__ __forceinline__ void set_mat(const SurfMat& m) { p[0] = __int_as_float(m.type); p[256] = m.baseColor.x; p[512] = m.baseColor.y; p[768] = m.baseColor.z; p[1024] = m.metallic; p[1280] = m.roughness; p[1536] = m.ior; }
    __device__ __forceinline__ f3 norm() const { return mk3(p[1792], p[2048], p[2304]); }
    __device__ __forceinline__ void set_norm(f3 n) { p[1792] = n.x; p[2048] = n.y; p[2304] = n.z; }
    __device__ __forceinline__ f3 wo() const { return mk3(p[2560], p[2816], p[3072]); }
    __device__ __forceinline__ void set_wo(f3 w) { p[2560] = w.x; p[2816] = w.y; p[3072] = w.z; }
};
template <> struct ColdSurf<false> {
    SurfMat m; f3 n, w;
    __device__ __forceinline__ SurfMat mat() const { return m; }
    __device__ __forceinline__ int type() const { return m.type; }
    __device__ __forceinline__ void set_mat(const SurfMat& x) { m = x; }
    __device__ __forceinline__ f3 norm() const { return n; }
    __device__ __forceinline__ void set_norm(f3 x) { n = x; }
    __device__ __forceinline__ f3 wo() const { return w; }
    __device__ __forceinline__ void set_wo(f3 x) { w = x; }
};
#ifndef RS_PATH_SURF_LDS
#define RS_PATH_SURF_LDS 1
#endif
// ReSTIR-GI's four recorded points (xv, nv at the first hit, xs, ns at the second: restir.cu:316-321,345-360) are written once per path and read
// once after it: they wait in the pixel's slot of the OUTPUT reservoir array (which the kernel overwrites at its end anyway, and which is not
// the array the temporal neighbour is read from), so that the LDS columns they took hold the surface state instead.
struct Glob3 {
    float* p;
    __device__ __forceinline__ f3 get() const { return ld3(p); }
    __device__ __forceinline__ void set(f3 v) const { st3(p, v); }
};
struct PathState {
    Cold3 direct, indirect;       // kModePT: direct / indirect; others: indirect only (ReSTIR: the sample's Lo)
    Cold3 throughput;
    // ReSTIR-GI bookkeeping (restir.cu:273-281,316-321)
    float primSamplePdf; bool primSampleDelta; SurfMat primMaterial;     // (primMaterial: textured scenes only; a plain material is read again from its id)
    Cold3 primWo;                 // (ReSTIR-GI: live from the primary hit to the end of the path)
    int primMatId;
#if RS_PATH_COLD_LDS && RS_PATH_SURF_LDS
    Glob3 xv, nv, xs, ns;
#else
    Cold3 xv, nv, xs, ns;
#endif
    int walks;
};

// The loop of the three kernels from the first shaded hit on, run by the WHOLE wave: lanes whose path has ended
// (or never started: `alive` false) stay in the loop with their flag down, so that the shadow rays can use the
// wave-level shadow-tree walk (trace_occluded_wave) and the continuation rays the pair-cooperative walk
// (trace_closest_wave) instead of per-lane walks inside a divergent loop.  Per lane the sequence of
// random draws and arithmetic is that of the reference.
template <int MODE, bool TEX, typename Sampler, typename Surf>
__device__ inline void path_loop(const DevScene& s, Hit h, Surf& surf, Ray ray, Sampler& rng, int maxDepth, bool alive, PathState& st) {
    Cold3& throughputC = st.throughput;
    throughputC.set(splat(1.f));
#define throughput throughputC.get()
    f3 pos = h.pos;
    surf.set_norm(h.norm); surf.set_wo(-ray.d);
    const bool env = TEX && s.envTex >= 0;
    for (int depth = 1; depth <= maxDepth; depth++) {
        if (!__any(alive)) break;
        const bool deltaBSDF = surf.type() == 2;
        {
            const f3 n0 = surf.norm();
            if (alive && !deltaBSDF && dot(n0, surf.wo()) < 0.f) surf.set_norm(-n0);
        }

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
        // What the segment's visibility decides is whether `add` is added (pathtrace.cu:205-212).  It is not asked -- the segment is counted
        // as the reference's testOcclusion call and not walked -- where the answer cannot matter: a sample without a valid pdf (the light faces
        // away, scene.h:448-452: sampleDirectLight returns InvalidPdf either way; 42 % of the segments on the Sponza-class scene), and a
        // contribution whose three components are all +0 (the light is below the surface's horizon, sat_dot = 0, or the BSDF is zero):
        // x + (+0) = x for every x but -0, and the sums, which start at +0, never become -0.  A NaN or -0 component: the segment is walked.
        f3 add = splat(0.f);
        const bool valid = nee && c.pdf > 0.f;
        if (valid) {
            const SurfMat material = surf.mat();
            const f3 norm = surf.norm(), wo = surf.wo();
            const float bsdfPdf = material_pdf(material, norm, wo, c.wi);
            add = ((((throughput * material_bsdf(material, norm, wo, c.wi)) * c.Li) * sat_dot(norm, c.wi)) / c.pdf) * power_heuristic(c.pdf, bsdfPdf);
        }
        const bool matters = valid && (__float_as_uint(add.x) | __float_as_uint(add.y) | __float_as_uint(add.z)) != 0u;
        const bool occluded = trace_occluded_wave(s, pos, c.point, matters);
        if (nee) {
            st.walks++;
            if (matters && !occluded) {
                if (MODE == kModePT && depth == 1) st.direct.add(add); else st.indirect.add(add);
            }
        }

        BsdfSample sample;
        sample.dir = splat(0.f); sample.bsdf = splat(0.f); sample.pdf = 0.f; sample.type = kBsInvalid;
        bool deltaSample = false;
        f3 norm = splat(0.f);
        if (alive) {
            const f3 r3 = mk3(rng.uniform(), rng.uniform(), rng.uniform());    // sample3D
            norm = surf.norm();
            sample = material_sample(surf.mat(), norm, surf.wo(), r3);
            if (sample.type == kBsInvalid) alive = false;
            else if (sample.pdf < 1e-8f) alive = false;
        }
        const f3 curPos = pos;
        if (alive) {
            deltaSample = (sample.type & kBsSpecular) != 0;
            if (MODE != kModeReSTIR || depth > 1) {                             // restir.cu:315-325
                throughputC.set(throughput * ((sample.bsdf / sample.pdf) * (deltaSample ? 1.f : abs_dot(norm, sample.dir))));
            }
            else {
                st.primSamplePdf = sample.pdf;
                st.primSampleDelta = deltaSample;
                st.xv.set(pos); st.nv.set(norm);
            }
            ray.o = pos + sample.dir * 1e-5f; ray.d = sample.dir;               // makeOffsetedRay
        }
        // The last bounce only asks whether its closest hit is an emissive triangle (a surface would need another bounce to contribute,
        // and without an environment map neither does a miss): rays that cannot hit one the way the reference accepts triangles
        // (may_hit_emissive_wave, a walk of the few emissive triangles' own tree) are counted as the reference's intersect call and not walked.
        // (ReSTIRIndirect at depth 1 records the hit point whatever it is: restir.cu:345-360.)
        bool walk = alive;
        if (depth == maxDepth && !env && !(MODE == kModeReSTIR && depth == 1)) walk = may_hit_emissive_wave(s, ray, alive);
        h = trace_closest_wave(s, ray, walk);
        if (alive) {
            st.walks++;
            surf.set_wo(-ray.d);
            if (h.primId == kNullPrim) {
                if (env) {
                    const f3 radiance = env_radiance(s, ray.d) * throughput;
                    const float weight = deltaSample ? 1.f : power_heuristic(sample.pdf, environment_map_pdf(s, ray.d));
                    st.indirect.add(radiance * weight);
                }
                alive = false;
            }
            else {
                pos = h.pos; norm = h.norm;
                const SurfMat material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
                surf.set_mat(material); surf.set_norm(norm);
                if (material.type == 4) {
                    if (!(dot(norm, ray.d) < 0.f)) {                            // SCENE_LIGHT_SINGLE_SIDED: the back side ends the path silently
                        const f3 radiance = material.baseColor;
                        const bool unweighted = deltaSample || (MODE == kModeReSTIR && depth == 1);      // restir.cu:353
                        const float weight = unweighted ? 1.f : power_heuristic(sample.pdf,
                            (luminance(radiance) * s.sumLightPowerInv * primitive_area(s, h.primId)) * dot(curPos - pos, curPos - pos) /
                                abs_dot(norm, normalize(curPos - pos)));         // Math::pdfAreaToSolidAngle (mathUtil.h:182-185)
                        st.indirect.add((radiance * throughput) * weight);
                        if (MODE == kModeReSTIR && depth == 1) { st.xs.set(pos); st.ns.set(norm); }
                    }
                    alive = false;
                }
                else if (MODE == kModeReSTIR && depth == 1) { st.xs.set(pos); st.ns.set(norm); }
            }
        }
    }
#undef throughput

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
#if RS_PATH_COLD_LDS
    constexpr bool kSurfLds = RS_PATH_SURF_LDS != 0;
    constexpr int kColdCols = MODE == kModePT ? 9 : (MODE == kModeReSTIR && !kSurfLds) ? 21 : 6;
    constexpr int kPrimWoCols = (MODE == kModeReSTIR && kSurfLds) ? 3 : 0;
    __shared__ float sCold[(kColdCols + (kSurfLds ? 13 : 0) + kPrimWoCols) * 256];
    st.indirect.p = sCold + threadIdx.x;
    st.throughput.p = sCold + (kColdCols - 3) * 256 + threadIdx.x;
    st.direct.p = MODE == kModePT ? sCold + 3 * 256 + threadIdx.x : st.indirect.p;                 // (only kModePT has a direct sum)
#if RS_PATH_SURF_LDS
    {
        float* slot = reinterpret_cast<float*>(resvOut + (MODE == kModeReSTIR && inside ? index : 0));      // (other modes: never touched)
        st.xv.p = slot + 3; st.nv.p = slot + 6; st.xs.p = slot + 9; st.ns.p = slot + 12;
    }
#else
    st.xv.p = st.nv.p = st.xs.p = st.ns.p = st.indirect.p;
    if (MODE == kModeReSTIR) { st.xv.p = sCold + 6 * 256 + threadIdx.x; st.nv.p = st.xv.p + 3 * 256; st.xs.p = st.xv.p + 6 * 256; st.ns.p = st.xv.p + 9 * 256; }
#endif
#endif
#if RS_PATH_COLD_LDS
    st.primWo.p = kPrimWoCols ? sCold + (kColdCols + 13) * 256 + threadIdx.x : st.indirect.p;
#endif
    st.direct.set(splat(0.f)); st.indirect.set(splat(0.f)); st.primSamplePdf = 0.f; st.primSampleDelta = false;
    if (MODE == kModeReSTIR) st.primWo.set(-ray.d);
    st.primMaterial = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f }; st.primMatId = -1;
    if (MODE == kModeReSTIR && (inside || !(RS_PATH_COLD_LDS && RS_PATH_SURF_LDS))) { st.xv.set(splat(0.f)); st.nv.set(splat(0.f)); st.xs.set(splat(0.f)); st.ns.set(splat(0.f)); }
    st.walks = 0;
    // primary hit (pathtrace.cu:172-190 / 343-350, restir.cu:259-270); lanes that end here keep alive = false
    bool alive = false;
    Hit hh = h;
    SurfMat material = SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f };
    if (inside) {
        st.walks = 1;
        if (h.primId == kNullPrim) {
            if (MODE == kModePT) st.direct.set(splat(1.f));                            // pathtrace.cu:175-178
        }
        else {
            f3 norm = h.norm;
            material = TEX ? textured_material(s, h, norm) : plain_material(s, h.matId);
            if (MODE == kModePT) material.baseColor = splat(1.f);                      // DENOISER_DEMODULATE (:181-185)
            if (material.type == 4) {
                if (MODE == kModePT) st.direct.set(splat(1.f));                        // :187-190
            }
            else {
                hh.norm = norm;
                if (TEX) st.primMaterial = material; else st.primMatId = h.matId;       // seven registers across the whole path loop, or one
                alive = true;
            }
        }
    }
#if RS_PATH_COLD_LDS
    ColdSurf<kSurfLds> surf;
    if constexpr (kSurfLds) surf.p = sCold + kColdCols * 256 + threadIdx.x;
#else
    ColdSurf<false> surf;
#endif
    surf.set_mat(material);
    path_loop<MODE, TEX, SamplerT<SOBOL>>(s, hh, surf, ray, rng, maxDepth, alive, st);                // every lane of the wave takes part
    if (inside) {
        if (MODE == kModePT) {
            f3 direct = st.direct.get(), indirect = st.indirect.get();
            if (any_nan_or_inf(direct)) direct = splat(0.f);
            if (any_nan_or_inf(indirect)) indirect = splat(0.f);
            accumulate(directIllum, index, hdr_to_ldr(direct), iter);                   // Math::HDRToLDR (:273-276)
            accumulate(indirectIllum, index, hdr_to_ldr(indirect), iter);
        }
        else if (MODE == kModePTIndirect) {
            f3 indirect = st.indirect.get();
            if (any_nan_or_inf(indirect)) indirect = splat(0.f);
            accumulate(indirectIllum, index, indirect, iter);
        }
        else {
            // WriteSample (restir.cu:372-416)
            IndResv smp; smp.Lo = st.indirect.get(); smp.xv = st.xv.get(); smp.nv = st.nv.get(); smp.xs = st.xs.get(); smp.ns = st.ns.get(); smp.M = 0; smp.W = 0.f;
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
                const SurfMat primMaterial = TEX ? st.primMaterial : (st.primMatId >= 0 ? plain_material(s, st.primMatId) : SurfMat{ 0, splat(0.f), 0.f, 0.f, 0.f });
                indirect = indirect * (material_bsdf(primMaterial, rv.nv, st.primWo.get(), primWi) * (st.primSampleDelta ? 1.f : sat_dot(rv.nv, primWi)));
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

unsigned long long* g_giRayCount = nullptr;     // 64 partial counters, 64 B apart

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
    RS_TRY(rs_denoise_order(devDirectIllum)); RS_TRY(rs_denoise_order(devIndirectIllum));      // images a filter on the denoise stream may still be reading
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
    RS_TRY(rs_denoise_order(devIndirectIllum));
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
    RS_TRY(rs_denoise_order(devIndirectIllum));
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
