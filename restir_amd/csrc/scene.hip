// scene.hip -- DevScene::create / destroy (src/scene.cpp:435-532) re-laid-out for CDNA4 (see
// rs_scene.h), Scene::buildDevData as one call, and batched ray entry points for parity tests.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "rs_internal.h"

using namespace rs;

namespace {

// Map ids as getTexturedMaterialAndSurface reads them (src/scene.h:78-99): baseColor -1 none / -2 procedural / index;
// metallic and roughness are used only when > -1; a normal map id other than -1 indexes `textures` directly.
int check_materials(int n, const rs_material* m, int numTextures, bool* anyMap) {
    *anyMap = false;
    for (int i = 0; i < n; i++) {
        const int ids[4] = { m[i].baseColorMapId, m[i].metallicMapId, m[i].roughnessMapId, m[i].normalMapId };
        for (int k = 0; k < 4; k++)
            if (ids[k] >= numTextures) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: material map id beyond the texture table");
        if (ids[0] < -2 || ids[3] < -1) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: invalid material map id");
        if (ids[0] != -1 || ids[1] > -1 || ids[2] > -1 || ids[3] != -1) *anyMap = true;
    }
    return 0;
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    RS_TRY(rs_dev_alloc(dst, src.size()));
    if (!src.empty()) RS_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return 0;
}

// The shadow-ray tree and the reference chain records (rs_scene.h walk_occlusion_tree).  Needs every
// reference box finite with min <= max (always true for boxes from rs_build_bvh; a caller-supplied table
// that is not leaves the fast path off and shadow rays walk the reference's tree).
int build_occlusion_side(rs_scene* s) {
    const size_t np = (size_t)s->numPrims, nn = (size_t)s->bvhSize;
    if (std::getenv("RS_NO_OCCLUSION_TREE")) return 0;          // A/B switch for measurements
    for (size_t i = 0; i < nn * 6; i++) if (!std::isfinite(s->hBoxes[i])) return 0;
    for (size_t i = 0; i < nn; i++)
        for (int k = 0; k < 3; k++) if (s->hBoxes[i * 6 + k] > s->hBoxes[i * 6 + 3 + k]) return 0;
    const std::vector<int>& parent = s->hParent;
    const std::vector<int>& leafOf = s->hLeafOf;
    std::vector<float> primBoxes(np * 6);
    for (size_t p = 0; p < np; p++) std::memcpy(&primBoxes[p * 6], &s->hBoxes[(size_t)leafOf[p] * 6], 6 * sizeof(float));
    std::vector<BvhNode> nodes;
    std::vector<int> leafPrims;
    // a scene the second tree cannot be built for (RS_ERR_UNSUPPORTED: e.g. all vertices in one point, so that the grid has no
    // extent) is still a scene the reference renders: the fast path stays off and shadow rays walk the reference's tree
    if (int e = rs_build_occlusion_bvh(s->numPrims, primBoxes.data(), nodes, leafPrims)) return e == RS_ERR_UNSUPPORTED ? 0 : e;
    const size_t no = nodes.size();
    if (no * 16 >= 0x7fffffffull || nn * sizeof(BvhNode) >= 0xffffffffull || np >= (1u << 27)) return 0;
    float base[3], scale[3];
    std::vector<unsigned> packed;
    if (int e = rs_quantize_occlusion_bvh(nodes, base, scale, packed)) return e == RS_ERR_UNSUPPORTED ? 0 : e;
    std::vector<BvhNode> chain(nn);
    for (size_t i = 0; i < nn; i++) {
        const float* b = &s->hBoxes[i * 6];
        BvhNode& r = chain[i];
        r.bminx = b[0]; r.bminy = b[1]; r.bminz = b[2]; r.primId = parent[i];
        r.bmaxx = b[3]; r.bmaxy = b[4]; r.bmaxz = b[5]; r.next = parent[i];
    }
    std::vector<TriRec> rec(np);
    for (size_t i = 0; i < np; i++) {
        const size_t p = (size_t)leafPrims[i];
        const float* t = &s->hVertices[p * 9];
        f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
        f3 e1 = v1 - v0, e2 = v2 - v0;
        float leafBits;
        std::memcpy(&leafBits, &leafOf[p], 4);
        rec[i] = TriRec{ v0.x, v0.y, v0.z, leafBits, e1.x, e1.y, e1.z, 0.f, e2.x, e2.y, e2.z, 0.f };
    }
    int root = 0;
    for (size_t i = 0; i < nn; i++) if (parent[i] < 0) root = (int)i;
    s->dev.occNested = s->dev.axisCull;          // a proper hierarchy is in particular nested
    s->dev.occRootLo = ld3(&s->hBoxes[(size_t)root * 6]);
    s->dev.occRootHi = ld3(&s->hBoxes[(size_t)root * 6 + 3]);
    // one record past the end: an empty box (lo = 65535 > hi = 0 on every axis fails the slab test for either sign of the direction)
    // whose link is its own offset -- a lane whose walk has ended stays there, so the walk loop needs no "has this lane ended" region
    packed.push_back(0xffffffffu); packed.push_back(0x0000ffffu); packed.push_back(0u); packed.push_back((unsigned)(no * 16));
    RS_TRY(rs_dev_alloc(&s->dOccNodes, no + 1));
    RS_HIP(hipMemcpy(s->dOccNodes, packed.data(), (no + 1) * 16, hipMemcpyHostToDevice));
    RS_TRY(upload(&s->dOccChain, chain));
    RS_TRY(upload(&s->dOccTris, rec));
#ifdef RS_WALK_STATS
    {   // depth of every record of the pre-order array (spans nest: a stack of span ends)
        std::vector<unsigned char> depth(no);
        std::vector<int> ends;
        for (size_t i = 0; i < no; i++) {
            while (!ends.empty() && ends.back() <= (int)i) ends.pop_back();
            depth[i] = (unsigned char)std::min<size_t>(ends.size(), 255);
            ends.push_back(nodes[i].next);
        }
        unsigned char* d = nullptr;
        RS_TRY(rs_dev_alloc(&d, no));
        RS_HIP(hipMemcpy(d, depth.data(), no, hipMemcpyHostToDevice));
        s->dev.occDepth = d;                              // (measurement builds: never freed)
        size_t perDepth[20] = {};
        for (size_t i = 0; i < no; i++) perDepth[std::min<int>(depth[i], 19)]++;
        std::fprintf(stderr, "occlusion tree: %zu nodes; per depth:", no);
        for (int k = 0; k < 20; k++) std::fprintf(stderr, " %zu", perDepth[k]);
        std::fprintf(stderr, "\n");
    }
#endif
    s->dev.occNodes = s->dOccNodes;
    s->dev.occChain = s->dOccChain;
    s->dev.occTris = s->dOccTris;
    s->dev.occCount = (int)no;
    s->dev.occBase = mk3(base[0], base[1], base[2]);
    s->dev.occScale = mk3(scale[0], scale[1], scale[2]);
    return 0;
}

// The tree of the EMISSIVE triangles alone (rs_scene.h may_hit_emissive_wave).  The last bounce of a path (gi.hip) only asks whether its
// closest hit is an emissive triangle; the reference accepts a triangle only if intersectTriangle hits it and its leaf box was entered, so a
// ray that hits no emissive triangle whose reference leaf box it passes cannot have an emissive closest hit -- and that question is asked
// of a tree over the reference leaf boxes of the few emissive triangles (the shadow tree's builder and grid: its relaxed box test cannot
// miss a triangle whose reference leaf box the ray passes), which stays in L1, instead of the scene's.  Needs the shadow side's preconditions.
int build_emissive_side(rs_scene* s) {
    if (!s->dev.occNodes) return 0;
    std::vector<int> prims;
    for (int p = 0; p < s->numPrims; p++) {
        const int m = s->hMaterialIds[(size_t)p];
        if (m >= 0 && m < (int)s->hMaterials.size() && s->hMaterials[(size_t)m].type == 4) prims.push_back(p);
    }
    if (prims.empty()) { s->dev.emiState = 1; s->dev.emiCount = 0; return 0; }      // no emissive triangle: no closest hit is one
    std::vector<float> boxes(prims.size() * 6);
    for (size_t i = 0; i < prims.size(); i++) std::memcpy(&boxes[i * 6], &s->hBoxes[(size_t)s->hLeafOf[(size_t)prims[i]] * 6], 6 * sizeof(float));
    std::vector<BvhNode> nodes;
    std::vector<int> leafPrims;
    if (int e = rs_build_occlusion_bvh((int)prims.size(), boxes.data(), nodes, leafPrims)) return e == RS_ERR_UNSUPPORTED ? 0 : e;
    float base[3], scale[3];
    std::vector<unsigned> packed;
    if (int e = rs_quantize_occlusion_bvh(nodes, base, scale, packed)) return e == RS_ERR_UNSUPPORTED ? 0 : e;
    const size_t no = nodes.size();
    packed.push_back(0xffffffffu); packed.push_back(0x0000ffffu); packed.push_back(0u); packed.push_back((unsigned)(no * 16));      // the end record
    std::vector<TriRec> rec(prims.size());
    for (size_t i = 0; i < prims.size(); i++) {
        const float* t = &s->hVertices[(size_t)prims[(size_t)leafPrims[i]] * 9];
        const f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6), e1 = v1 - v0, e2 = v2 - v0;
        rec[i] = TriRec{ v0.x, v0.y, v0.z, 0.f, e1.x, e1.y, e1.z, 0.f, e2.x, e2.y, e2.z, 0.f };
    }
    RS_TRY(rs_dev_alloc(&s->dEmiNodes, no + 1));
    RS_HIP(hipMemcpy(s->dEmiNodes, packed.data(), (no + 1) * 16, hipMemcpyHostToDevice));
    RS_TRY(upload(&s->dEmiTris, rec));
    s->dev.emiNodes = s->dEmiNodes; s->dev.emiTris = s->dEmiTris; s->dev.emiCount = (int)no;
    s->dev.emiBase = mk3(base[0], base[1], base[2]); s->dev.emiScale = mk3(scale[0], scale[1], scale[2]);
    s->dev.emiState = 1;
    return 0;
}

// The closest-hit trees of the incoherent rays (occlusion_bvh.cpp rs_build_ordered_bvh, rs_scene.h walk_ordered_tree): per axis
// one tree over the leaf sequence of the even threaded order, emitted in forward and mirrored pre-order -> six arrays of 16-byte
// grid-box records like the shadow tree's, laid out at a common stride in ONE allocation so that a lane addresses its order by
// k * stride; links are absolute byte offsets; the last slot of every stride is the self-linked empty record a finished walk
// rests on, and the slots between a shorter tree's end and that record are empty inner records linked to it.  Needs the shadow
// side (grid, chain records) and a proper hierarchy; any table whose odd orders are not the mirror images of the even ones
// (never one from rs_build_bvh) leaves the path off, and those rays walk the reference's tree.
int build_ordered_side(rs_scene* s) {
    if (!s->dev.occNodes || !s->dev.occNested || !s->dev.linksNested) return 0;
    if (std::getenv("RS_NO_ORDERED_TREE")) return 0;            // A/B switch for measurements
    const size_t np = (size_t)s->numPrims, nn = (size_t)s->bvhSize;
    std::vector<int> seq[3];
    for (int a = 0; a < 3; a++) {
        seq[a].reserve(np);
        for (size_t i = 0; i < nn; i++) if (s->hNodes[2 * a][i * 3] >= 0) seq[a].push_back(s->hNodes[2 * a][i * 3]);
        if (seq[a].size() != np) return 0;
        size_t j = np;
        for (size_t i = 0; i < nn; i++) { const int p = s->hNodes[2 * a + 1][i * 3]; if (p >= 0) { if (j == 0 || seq[a][--j] != p) return 0; } }
        if (j != 0) return 0;
    }
    std::vector<float> primBoxes(np * 6);
    for (size_t p = 0; p < np; p++) std::memcpy(&primBoxes[p * 6], &s->hBoxes[(size_t)s->hLeafOf[p] * 6], 6 * sizeof(float));
    std::vector<unsigned> packed[6];
    size_t maxCount = 0;
    {   // the three axes are independent: one host thread each (2.8 M triangles: 2.5 s -> 0.9 s of scene set-up)
        int err[3] = { 0, 0, 0 };
        bool sameGrid[3] = { true, true, true };
        size_t counts[3] = { 0, 0, 0 };
        auto work = [&](int a) {
          try {                                              // (an exception must not leave a host thread: the build's vectors can run out of memory)
            std::vector<BvhNode> tree[2];
            if ((err[a] = rs_build_ordered_bvh(s->numPrims, primBoxes.data(), seq[a].data(), tree[0], tree[1])) != 0) return;
            for (int m = 0; m < 2; m++) {
                float base[3], scale[3];
                if ((err[a] = rs_quantize_occlusion_bvh(tree[m], base, scale, packed[2 * a + m])) != 0) return;
                // every tree's root box is the union of all leaf boxes, so all of them share the shadow tree's grid
                if (std::memcmp(base, &s->dev.occBase, 12) != 0 || std::memcmp(scale, &s->dev.occScale, 12) != 0) sameGrid[a] = false;
                counts[a] = std::max(counts[a], tree[m].size());
            }
          } catch (...) { err[a] = RS_ERR_UNSUPPORTED; }     // the scene then renders without these trees (the reference walk)
        };
        std::thread t1(work, 1), t2(work, 2);
        work(0);
        t1.join(); t2.join();
        for (int a = 0; a < 3; a++) {
            if (err[a]) return err[a] == RS_ERR_UNSUPPORTED ? 0 : err[a];
            if (!sameGrid[a]) return 0;
            maxCount = std::max(maxCount, counts[a]);
        }
    }
    const size_t stride = (maxCount + 1) * 16;
    if (stride * 6 >= 0x7fffffffull) return 0;                  // links are signed 32-bit byte offsets (the sign marks a leaf)
    std::vector<unsigned> all(stride * 6 / 4);
    for (int k = 0; k < 6; k++) {
        const size_t count = packed[k].size() / 4, first = (size_t)k * stride, endOff = first + maxCount * 16;
        unsigned* o = &all[first / 4];
        std::memcpy(o, packed[k].data(), count * 16);
        for (size_t i = 0; i < count; i++) {
            unsigned& w = o[i * 4 + 3];
            if ((int)w >= 0) w = (w == count * 16) ? (unsigned)endOff : (unsigned)(first + w);       // inner: miss link, relative -> absolute
        }
        for (size_t i = count; i <= maxCount; i++) { o[i * 4] = 0xffffffffu; o[i * 4 + 1] = 0x0000ffffu; o[i * 4 + 2] = 0u; o[i * 4 + 3] = (unsigned)endOff; }
    }
    // triangles in the even orders' sequence: pad0 = reference leaf node (the chain check starts there), pad1 = primitive id
    std::vector<TriRec> rec(np * 3);
    for (int a = 0; a < 3; a++)
        for (size_t i = 0; i < np; i++) {
            const size_t p = (size_t)seq[a][i];
            const float* t = &s->hVertices[p * 9];
            f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
            f3 e1 = v1 - v0, e2 = v2 - v0;
            float leafBits, primBits;
            const int prim = (int)p;
            std::memcpy(&leafBits, &s->hLeafOf[p], 4); std::memcpy(&primBits, &prim, 4);
            rec[(size_t)a * np + i] = TriRec{ v0.x, v0.y, v0.z, leafBits, e1.x, e1.y, e1.z, primBits, e2.x, e2.y, e2.z, 0.f };
        }
    RS_TRY(rs_dev_alloc(&s->dOrdNodes, all.size() / 4));
    RS_HIP(hipMemcpy(s->dOrdNodes, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    RS_TRY(upload(&s->dOrdTris, rec));
    s->dev.ordNodes = s->dOrdNodes;
    s->dev.ordTris = s->dOrdTris;
    s->dev.ordStride = (unsigned)stride;
    return 0;
}

}  // namespace

extern "C" int rs_scene_destroy(rs_scene* s) {
    RS_SCOPE(s);
    if (!s) return 0;
    (void)rs_gbuffer_release_scene(s);                  // asynchronous mode: a GBuffer::render of this scene that has only been recorded so far
    (void)rs_synchronize();                             // ... and kernels on the library / auxiliary streams may still read the scene
    rs_dev_free(s->dNodesAll); rs_dev_free(s->dWalkStats); rs_dev_free(s->dOccNodes); rs_dev_free(s->dEmiNodes); rs_dev_free(s->dEmiTris); rs_dev_free(s->dOccChain); rs_dev_free(s->dOccTris); rs_dev_free(s->dOrdNodes); rs_dev_free(s->dOrdTris);
    rs_dev_free(s->dTris); rs_dev_free(s->dVertices); rs_dev_free(s->dNormals);
    rs_dev_free(s->dMaterialIds); rs_dev_free(s->dMaterials); rs_dev_free(s->dLights); rs_dev_free(s->dAlias);
    rs_dev_free(s->dTextures); rs_dev_free(s->dEnvAlias); rs_dev_free(s->dTexcoords); rs_dev_free(s->dSampleSeq);
    for (float*& p : s->dTexData) rs_dev_free(p);
    delete s;
    return 0;
}

// DevScene::sampleSequence (src/scene.cpp:500-506 reads sobol_10k_200.bin into it).  The reference chooses its sampler at compile
// time (SAMPLER_USE_SOBOL, src/common.h:4); here the scene carries the choice: with a table every kernel that draws random numbers
// runs its Sobol instantiation (src/sampler.h:9-36), with data == NULL the default thrust engine (:38-49) again.
extern "C" int rs_scene_set_sample_sequence(rs_scene* s, const uint32_t* data, int numSamples, int numDims) {
    RS_SCOPE(s);
    if (!s) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_set_sample_sequence: null scene");
    if (data && (numSamples <= 0 || numDims != kSobolSampleDim))
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_set_sample_sequence: the table must be numSamples x 200 (SobolSampleDim, src/sampler.h:11)");
    (void)rs_gbuffer_release_scene(s);
    RS_TRY(rs_synchronize());                           // kernels in flight may still read the previous table
    rs_dev_free(s->dSampleSeq);
    s->dev.sampleSeq = nullptr; s->dev.sampleCount = 0;
    if (!data) return 0;
    const size_t n = (size_t)numSamples * numDims;
    // Sampler::sample reads data[ptr++] without a bound (sampler.h:20): a guard of zeros behind the table keeps the long paths of the
    // multi-bounce kernels inside the allocation for the last rows too (ReSTIRDirect draws at most 181 < 200 numbers and never gets there)
    RS_TRY(rs_dev_alloc(&s->dSampleSeq, n + kSobolGuard));
    RS_HIP(hipMemcpy(s->dSampleSeq, data, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    RS_HIP(hipMemset(s->dSampleSeq + n, 0, kSobolGuard * sizeof(uint32_t)));
    s->dev.sampleSeq = s->dSampleSeq; s->dev.sampleCount = numSamples;
    return 0;
}

int rs_check_looper(const rs_scene* scene, int looper, const char* what) {
    if (scene && scene->dev.sampleSeq && (looper < 0 || looper >= scene->dev.sampleCount)) {
        static thread_local std::string msg;
        msg = std::string(what) + ": looper outside the Sobol table (the reference keeps State::looper below SobolSampleNum, restir.cu:441-445)";
        return rs_fail(RS_ERR_INVALID_ARGUMENT, msg.c_str());
    }
    return 0;
}

extern "C" int rs_scene_create(const rs_scene_desc* d, rs_scene** out) {
    if (!d || !out) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: null argument");
    *out = nullptr;
    if (d->numPrims <= 0 || d->bvhSize != 2 * d->numPrims - 1 || !d->vertices || !d->normals || !d->materialIds ||
        !d->materials || d->numMaterials <= 0 || !d->boundingBoxes)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: inconsistent scene description");
    for (int k = 0; k < 6; k++)
        if (!d->bvhNodes[k]) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: missing BVH order");
    bool anyMap = false;
    if (d->numTextures < 0 || (d->numTextures > 0 && !d->textures) || d->envMapTexId < -1 || d->envMapTexId >= d->numTextures)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: inconsistent texture table");
    for (int i = 0; i < d->numTextures; i++)
        if (d->textures[i].width <= 0 || d->textures[i].height <= 0 || !d->textures[i].data)
            return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: empty texture");
    RS_TRY(check_materials(d->numMaterials, d->materials, d->numTextures, &anyMap));
    const bool hasEnv = d->envMapTexId >= 0;
    if (hasEnv && (!d->envMapProb || !d->envMapFailId || d->numLights < 1))
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: an environment map needs its sampler table and its light-sampler entry");
    if (anyMap && !d->texcoords) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: texture maps need texcoords");

    const size_t np = (size_t)d->numPrims, nn = (size_t)d->bvhSize, nl = (size_t)(d->numLights > 0 ? d->numLights : 0);
    const size_t nlp = nl - (hasEnv ? 1 : 0);          // light primitives; the environment map is the last sampler entry
    rs_scene* s = new rs_scene();
    s->ctx = rs_ctx();
    rs_ctx_scope scope(s->ctx);                       // (selects the context's device for the uploads below)
    { static std::atomic<unsigned long long> counter{0}; s->id = ++counter; }
    s->numPrims = d->numPrims; s->bvhSize = d->bvhSize; s->numLights = (int)nl;
    s->sumLightPower = d->sumLightPower;
    s->hVertices.assign(d->vertices, d->vertices + np * 9);
    s->hNormals.assign(d->normals, d->normals + np * 9);
    if (d->texcoords) s->hTexcoords.assign(d->texcoords, d->texcoords + np * 6); else s->hTexcoords.assign(np * 6, 0.f);
    s->hMaterialIds.assign(d->materialIds, d->materialIds + np);
    s->hMaterials.assign(d->materials, d->materials + d->numMaterials);
    s->hBoxes.assign(d->boundingBoxes, d->boundingBoxes + nn * 6);
    for (int k = 0; k < 6; k++) s->hNodes[k].assign(d->bvhNodes[k], d->bvhNodes[k] + nn * 3);
    if (nl) {
        if (nlp) {
            s->hLightPrimIds.assign(d->lightPrimIds, d->lightPrimIds + nlp);
            s->hLightRadiance.assign(d->lightUnitRadiance, d->lightUnitRadiance + nlp * 3);
        }
        s->hLightProb.assign(d->lightProb, d->lightProb + nl);
        s->hLightFailId.assign(d->lightFailId, d->lightFailId + nl);
    }
    s->envMapTexId = d->envMapTexId;
    s->textured = anyMap || hasEnv;
    s->hTexData.resize((size_t)d->numTextures);
    s->hTextures.resize((size_t)d->numTextures);
    for (int i = 0; i < d->numTextures; i++) {
        const size_t n = (size_t)d->textures[i].width * d->textures[i].height * 3;
        s->hTexData[i].assign(d->textures[i].data, d->textures[i].data + n);
        s->hTextures[i] = rs_texture{ d->textures[i].width, d->textures[i].height, s->hTexData[i].data() };
    }
    if (hasEnv) {
        const size_t n = (size_t)d->textures[d->envMapTexId].width * d->textures[d->envMapTexId].height;
        s->hEnvProb.assign(d->envMapProb, d->envMapProb + n);
        s->hEnvFail.assign(d->envMapFailId, d->envMapFailId + n);
        for (size_t i = 0; i < n; i++)
            if (s->hEnvFail[i] < 0 || (size_t)s->hEnvFail[i] >= n) { delete s; return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: environment sampler index out of range"); }
    }

    // validate indices the kernels will chase (a bad link would walk off the arrays on the GPU)
    for (size_t i = 0; i < np; i++)
        if (s->hMaterialIds[i] < 0 || s->hMaterialIds[i] >= d->numMaterials) { delete s; return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: material id out of range"); }
    for (int k = 0; k < 6; k++)
        for (size_t i = 0; i < nn; i++) {
            const int* n = &s->hNodes[k][i * 3];
            if (n[0] < -1 || n[0] >= d->numPrims || n[1] < 0 || n[1] >= d->bvhSize || n[2] <= (int)i || n[2] > d->bvhSize) {
                delete s; return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: malformed MTBVH node (links must point forward)");
            }
        }
    // the reference tree's parent / leaf tables (shadow-ray verification, proper-hierarchy check): derived, and thereby
    // validated as a tree, before anything is uploaded -- a malformed caller-supplied table is refused on the host
    if (int e = rs_reference_chain_tables(s->bvhSize, s->hNodes[0].data(), s->hParent, s->hLeafOf, s->numPrims)) { delete s; return e; }
    for (size_t i = 0; i < nl; i++)
        if ((i < nlp && (s->hLightPrimIds[i] < 0 || s->hLightPrimIds[i] >= d->numPrims)) || s->hLightFailId[i] < 0 || s->hLightFailId[i] >= (int)nl) {
            delete s; return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_create: light table index out of range");
        }

    // fused node records: the six threaded orders back to back in one array, plus one padding record so
    // that the speculative successor fetch of the very last node stays inside the allocation
    {
        std::vector<BvhNode> rec(nn * 6 + 1);
        for (int k = 0; k < 6; k++) {
            for (size_t i = 0; i < nn; i++) {
                const int* n = &s->hNodes[k][i * 3];
                const float* b = &s->hBoxes[(size_t)n[1] * 6];
                BvhNode& r = rec[(size_t)k * nn + i];
                r.bminx = b[0]; r.bminy = b[1]; r.bminz = b[2]; r.primId = n[0];
                r.bmaxx = b[3]; r.bmaxy = b[4]; r.bmaxz = b[5]; r.next = n[2];
            }
        }
        BvhNode& pad = rec[nn * 6];
        pad.bminx = pad.bminy = pad.bminz = pad.bmaxx = pad.bmaxy = pad.bmaxz = 0.f; pad.primId = -1; pad.next = 0;
        if ((nn * 6 + 1) * sizeof(BvhNode) >= 0xffffffffull) { delete s; return rs_fail(RS_ERR_UNSUPPORTED, "rs_scene_create: BVH larger than 4 GiB of node records"); }
        if (int e = upload(&s->dNodesAll, rec)) { rs_scene_destroy(s); return e; }
    }
    // pre-differenced triangles
    {
        std::vector<TriRec> rec(np);
        for (size_t i = 0; i < np; i++) {
            const float* t = &s->hVertices[i * 9];
            f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
            f3 e1 = v1 - v0, e2 = v2 - v0;
            rec[i] = TriRec{ v0.x, v0.y, v0.z, 0.f, e1.x, e1.y, e1.z, 0.f, e2.x, e2.y, e2.z, 0.f };
        }
        if (int e = upload(&s->dTris, rec)) { rs_scene_destroy(s); return e; }
    }
    // light records with the per-light constants of sampleDirectLightNoVisibility (scene.h:411-424)
    {
        std::vector<LightRec> rec(nl);
        std::vector<AliasRec> al(nl);
        const float sumInv = 1.f / d->sumLightPower;                       // scene.cpp:489
        for (size_t i = 0; i < nl; i++) {
            al[i].prob = s->hLightProb[i];
            al[i].failId = s->hLightFailId[i];
            if (i >= nlp) { rec[i] = LightRec{}; continue; }          // the environment map's sampler entry has no triangle
            const float* t = &s->hVertices[(size_t)s->hLightPrimIds[i] * 9];
            f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
            f3 c = cross(v1 - v0, v2 - v0);
            f3 nrm = normalize(c);                                          // Math::triangleNormal
            float area = length(c) * .5f;                                   // Math::triangleArea
            f3 Le = ld3(&s->hLightRadiance[i * 3]);
            float power = luminance(Le) / (area * 2.f * kGlmPi);
            rec[i] = LightRec{ v0.x, v0.y, v0.z, nrm.x, v1.x, v1.y, v1.z, nrm.y, v2.x, v2.y, v2.z, nrm.z,
                               Le.x, Le.y, Le.z, power * sumInv };
        }
        if (int e = upload(&s->dLights, rec)) { rs_scene_destroy(s); return e; }
        if (int e = upload(&s->dAlias, al)) { rs_scene_destroy(s); return e; }
    }
    if (int e = upload(&s->dVertices, s->hVertices)) { rs_scene_destroy(s); return e; }
    if (int e = upload(&s->dNormals, s->hNormals)) { rs_scene_destroy(s); return e; }
    if (int e = upload(&s->dMaterialIds, s->hMaterialIds)) { rs_scene_destroy(s); return e; }
    if (int e = upload(&s->dMaterials, s->hMaterials)) { rs_scene_destroy(s); return e; }

    // textures (DevScene::create, src/scene.cpp:479-498)
    if (!s->hTextures.empty()) {
        std::vector<TexRec> recs(s->hTextures.size());
        s->dTexData.assign(s->hTextures.size(), nullptr);
        for (size_t i = 0; i < s->hTextures.size(); i++) {
            if (int e = upload(&s->dTexData[i], s->hTexData[i])) { rs_scene_destroy(s); return e; }
            recs[i] = TexRec{ s->dTexData[i], s->hTextures[i].width, s->hTextures[i].height };
        }
        if (int e = upload(&s->dTextures, recs)) { rs_scene_destroy(s); return e; }
    }
    if (hasEnv) {
        std::vector<AliasRec> al(s->hEnvProb.size());
        for (size_t i = 0; i < al.size(); i++) { al[i].prob = s->hEnvProb[i]; al[i].failId = s->hEnvFail[i]; }
        if (int e = upload(&s->dEnvAlias, al)) { rs_scene_destroy(s); return e; }
    }
    if (s->textured) { if (int e = upload(&s->dTexcoords, s->hTexcoords)) { rs_scene_destroy(s); return e; } }
    s->dev.texcoords = s->dTexcoords;
    s->dev.sumLightPowerInv = 1.f / d->sumLightPower;                      // scene.cpp:489
    s->dev.textures = s->dTextures;
    s->dev.envAlias = s->dEnvAlias;
    s->dev.envTex = s->envMapTexId;
    s->dev.envLen = hasEnv ? (int)s->hEnvProb.size() : 0;

    s->dev.nodesAll = s->dNodesAll;
    s->dev.tris = s->dTris;
    s->dev.vertices = s->dVertices;
    s->dev.normals = s->dNormals;
    s->dev.materialIds = s->dMaterialIds;
    s->dev.materials = s->dMaterials;
    s->dev.lights = s->dLights;
    s->dev.alias = s->dAlias;
    s->dev.bvhSize = s->bvhSize;
    s->dev.numPrims = s->numPrims;
    s->dev.numLights = s->numLights;
    s->dev.numMaterials = d->numMaterials;
    s->dev.occNodes = nullptr; s->dev.occChain = nullptr; s->dev.occTris = nullptr; s->dev.occCount = 0;
    s->dev.occBase = splat(0.f); s->dev.occScale = splat(0.f);
    s->dev.walkStats = nullptr; s->dev.occDepth = nullptr;
    s->dev.occNested = false; s->dev.occRootLo = splat(0.f); s->dev.occRootHi = splat(0.f);
    s->dev.ordNodes = nullptr; s->dev.ordTris = nullptr; s->dev.ordStride = 0;
    // Is the box table a proper hierarchy (finite, min <= max, every box inside its parent's, every leaf box
    // around its triangle)?  Tables from rs_build_bvh are; the two shortcuts that rely on it (skip_far_on_axis and
    // the leaf shortcut of the shadow tree) are switched off for any other caller-supplied table.
    {
        bool proper = true;
        for (size_t i = 0; i < nn * 6 && proper; i++) proper = std::isfinite(s->hBoxes[i]);
        for (size_t i = 0; i < nn && proper; i++) {
            const float* c = &s->hBoxes[i * 6];
            for (int k = 0; k < 3; k++) proper = proper && c[k] <= c[3 + k];
            if (s->hParent[i] >= 0) {
                const float* q = &s->hBoxes[(size_t)s->hParent[i] * 6];
                for (int k = 0; k < 3; k++) proper = proper && q[k] <= c[k] && q[3 + k] >= c[3 + k];
            }
        }
        for (size_t p = 0; p < np && proper; p++) {
            const float* c = &s->hBoxes[(size_t)s->hLeafOf[p] * 6];
            for (int v = 0; v < 3; v++)
                for (int k = 0; k < 3; k++) { const float x = s->hVertices[p * 9 + v * 3 + k]; proper = proper && c[k] <= x && x <= c[3 + k]; }
        }
        s->dev.axisCull = proper;
    }
    // Do the miss links of every threaded order nest (a node inside the span (a, link(a)) never links beyond link(a))?  They do
    // for a pre-order layout with link = end of the subtree, which is what BVHBuilder::buildMTBVH produces (src/bvh.cpp:160-202);
    // the packet walk's next-node shortcut (rs_scene.h packet_walk_order) relies on it and is off for any other table.
    {
        bool nested = true;
        std::vector<int> open;
        for (int k = 0; k < 6 && nested; k++) {
            open.clear();
            for (size_t i = 0; i < nn && nested; i++) {
                while (!open.empty() && open.back() <= (int)i) open.pop_back();
                const int link = s->hNodes[k][i * 3 + 2];
                if (!open.empty() && link > open.back()) nested = false;
                open.push_back(link);
            }
        }
        s->dev.linksNested = nested;
    }
#ifdef RS_WALK_STATS
    if (int e = rs_dev_alloc(&s->dWalkStats, 96)) { rs_scene_destroy(s); return e; }
    (void)hipMemset(s->dWalkStats, 0, 96 * sizeof(unsigned long long));
    s->dev.walkStats = s->dWalkStats;
#endif
    if (int e = build_occlusion_side(s)) { rs_scene_destroy(s); return e; }
    if (int e = build_ordered_side(s)) { rs_scene_destroy(s); return e; }
    if (int e = build_emissive_side(s)) { rs_scene_destroy(s); return e; }
    *out = s;
    return 0;
}

#ifdef RS_WALK_STATS
// measurement builds only (tools/walk_stats.py): wave-level counters of walk_occlusion_tree
extern "C" int rs_debug_walk_stats(rs_scene* s, unsigned long long* out64, int reset) {
    RS_SCOPE(s);
    RS_HIP(hipDeviceSynchronize());
    RS_HIP(hipMemcpy(out64, s->dWalkStats, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) RS_HIP(hipMemset(s->dWalkStats, 0, 64 * sizeof(unsigned long long)));
    return 0;
}
// slots 64..95: the closest-hit walk of the bounce rays (rs_scene.h walk_ordered_tree; tools/bench_closest_wave.py)
extern "C" int rs_debug_walk_stats_ordered(rs_scene* s, unsigned long long* out32, int reset) {
    RS_SCOPE(s);
    RS_HIP(hipDeviceSynchronize());
    RS_HIP(hipMemcpy(out32, s->dWalkStats + 64, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) RS_HIP(hipMemset(s->dWalkStats + 64, 0, 32 * sizeof(unsigned long long)));
    return 0;
}
#endif

extern "C" int rs_scene_build(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                              const int* materialIds, int numMaterials, const rs_material* materials, rs_scene** out) {
    return rs_scene_build_textured(numPrims, vertices, normals, texcoords, materialIds, numMaterials, materials, 0, nullptr, -1, out);
}

extern "C" int rs_scene_build_textured(int numPrims, const float* vertices, const float* normals, const float* texcoords,
                                       const int* materialIds, int numMaterials, const rs_material* materials,
                                       int numTextures, const rs_texture* textures, int envMapTexId, rs_scene** out) {
    if (!out) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_build: null output");
    if (numTextures < 0 || (numTextures > 0 && !textures) || envMapTexId < -1 || envMapTexId >= numTextures)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_build: inconsistent texture table");
    *out = nullptr;
    if (numPrims <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_build: no mesh data");   // scene.cpp:192-195
    const size_t np = (size_t)numPrims, nn = 2 * np - 1;

    int numLights = 0;
    std::vector<int> lightPrim(np);
    std::vector<float> lightRad(np * 3), lightPower(np);
    RS_TRY(rs_build_light_table(numPrims, vertices, materialIds, numMaterials, materials, &numLights,
                                lightPrim.data(), lightRad.data(), lightPower.data()));
    // Scene::createLightSampler (src/scene.cpp:136-157): the environment map, if any, is one more light
    std::vector<float> envProb;
    std::vector<int> envFail;
    if (envMapTexId >= 0) {
        const rs_texture& env = textures[envMapTexId];
        if (env.width <= 0 || env.height <= 0 || !env.data) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_build: empty environment map");
        const size_t n = (size_t)env.width * env.height;
        std::vector<float> pdf(n);
        RS_TRY(rs_build_envmap_pdf(env.width, env.height, env.data, pdf.data()));
        envProb.resize(n); envFail.resize(n);
        float envSum = 0.f;
        RS_TRY(rs_build_alias_table((int)n, pdf.data(), envProb.data(), envFail.data(), &envSum));
        lightPower.resize((size_t)numLights + 1);
        lightPower[(size_t)numLights] = envSum;
        numLights++;
    }
    std::vector<float> prob((size_t)numLights);
    std::vector<int> fail((size_t)numLights);
    float sumAll = 0.f;
    RS_TRY(rs_build_alias_table(numLights, lightPower.data(), prob.data(), fail.data(), &sumAll));

    std::vector<float> boxes(nn * 6);
    std::vector<int> nodes[6];
    int* nodePtr[6];
    for (int k = 0; k < 6; k++) { nodes[k].resize(nn * 3); nodePtr[k] = nodes[k].data(); }
    int bvhSize = 0;
    RS_TRY(rs_build_bvh(numPrims, vertices, boxes.data(), nodePtr, &bvhSize));

    rs_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.numPrims = numPrims; d.vertices = vertices; d.normals = normals; d.texcoords = texcoords;
    d.materialIds = materialIds; d.numMaterials = numMaterials; d.materials = materials;
    d.bvhSize = bvhSize; d.boundingBoxes = boxes.data();
    for (int k = 0; k < 6; k++) d.bvhNodes[k] = nodes[k].data();
    d.numLights = numLights; d.lightPrimIds = lightPrim.data(); d.lightUnitRadiance = lightRad.data();
    d.lightProb = prob.data(); d.lightFailId = fail.data(); d.sumLightPower = sumAll;
    d.numTextures = numTextures; d.textures = textures; d.envMapTexId = envMapTexId;
    d.envMapProb = envProb.empty() ? nullptr : envProb.data(); d.envMapFailId = envFail.empty() ? nullptr : envFail.data();
    return rs_scene_create(&d, out);
}

extern "C" int rs_scene_host_desc(const rs_scene* s, rs_scene_desc* d) {
    RS_SCOPE(s);
    if (!s || !d) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_host_desc: null argument");
    std::memset(d, 0, sizeof *d);
    d->numPrims = s->numPrims; d->vertices = s->hVertices.data(); d->normals = s->hNormals.data();
    d->texcoords = s->hTexcoords.data(); d->materialIds = s->hMaterialIds.data();
    d->numMaterials = (int)s->hMaterials.size(); d->materials = s->hMaterials.data();
    d->bvhSize = s->bvhSize; d->boundingBoxes = s->hBoxes.data();
    for (int k = 0; k < 6; k++) d->bvhNodes[k] = s->hNodes[k].data();
    d->numLights = s->numLights; d->lightPrimIds = s->hLightPrimIds.data();
    d->lightUnitRadiance = s->hLightRadiance.data(); d->lightProb = s->hLightProb.data();
    d->lightFailId = s->hLightFailId.data(); d->sumLightPower = s->sumLightPower;
    d->numTextures = (int)s->hTextures.size(); d->textures = s->hTextures.data(); d->envMapTexId = s->envMapTexId;
    d->envMapProb = s->hEnvProb.data(); d->envMapFailId = s->hEnvFail.data();
    return 0;
}

// ---- batched scene services (parity tests of DevScene::intersect / testOcclusion) -------------
__global__ void __launch_bounds__(256) k_trace_closest(DevScene s, int n, const float* __restrict__ rays,
                                                       int* __restrict__ primId, int* __restrict__ matId,
                                                       float* __restrict__ pos, float* __restrict__ norm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = ld3(rays + (size_t)i * 6); r.d = ld3(rays + (size_t)i * 6 + 3);
    Hit h = trace_closest(s, r);
    primId[i] = h.primId;
    matId[i] = h.primId != kNullPrim ? h.matId : -1;
    st3(pos + (size_t)i * 3, h.pos);
    st3(norm + (size_t)i * 3, h.norm);
}

__global__ void __launch_bounds__(256) k_trace_occlusion(DevScene s, int n, const float* __restrict__ seg, int* __restrict__ occ) {
    // same wave-level service the ReSTIR shadow pass uses: every lane of the wave takes part
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const size_t j = active ? (size_t)i : 0;
    const bool o = trace_occluded_wave(s, ld3(seg + j * 6), ld3(seg + j * 6 + 3), active);
    if (active) occ[i] = o ? 1 : 0;
}

// the wave-level service the multi-bounce kernels use for their bounce rays (gi.hip): every lane of the wave takes part
__global__ void __launch_bounds__(256) k_trace_closest_wave(DevScene s, int n, const float* __restrict__ rays,
                                                            int* __restrict__ primId, int* __restrict__ matId,
                                                            float* __restrict__ pos, float* __restrict__ norm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const size_t j = active ? (size_t)i : 0;
    Ray r; r.o = ld3(rays + j * 6); r.d = ld3(rays + j * 6 + 3);
    const Hit h = trace_closest_wave(s, r, active);
    if (!active) return;
    primId[i] = h.primId;
    matId[i] = h.primId != kNullPrim ? h.matId : -1;
    st3(pos + (size_t)i * 3, h.pos);
    st3(norm + (size_t)i * 3, h.norm);
}

extern "C" int rs_trace_closest_wave(const rs_scene* s, int n, const float* devRays, int* devPrimId, int* devMatId, float* devPos, float* devNorm) {
    RS_SCOPE(s);
    if (!s || n < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_trace_closest_wave: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_trace_closest_wave, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), s->dev, n, devRays, devPrimId, devMatId, devPos, devNorm);
    return rs_after_launch("rs_trace_closest_wave");
}

// 1: bounce rays take the closest-hit trees in the reference's visiting orders (the default when they could be built), 0: the
// reference's own tree (A/B and parity tests).  Returns through *was (may be null) whether they were in use.
extern "C" int rs_scene_set_ordered_tree(rs_scene* s, int on, int* was) {
    RS_SCOPE(s);
    if (!s) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_set_ordered_tree: null scene");
    if (was) *was = s->dev.ordNodes ? 1 : 0;
    (void)rs_gbuffer_release_scene(s);
    RS_TRY(rs_synchronize());                           // kernels in flight carry the previous DevScene by value; nothing to wait for but keep the order simple
    s->dev.ordNodes = (on && s->dOrdNodes) ? s->dOrdNodes : nullptr;
    return 0;
}

extern "C" int rs_trace_closest(const rs_scene* s, int n, const float* devRays, int* devPrimId, int* devMatId, float* devPos, float* devNorm) {
    RS_SCOPE(s);
    if (!s || n < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_trace_closest: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_trace_closest, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), s->dev, n, devRays, devPrimId, devMatId, devPos, devNorm);
    return rs_after_launch("rs_trace_closest");
}

extern "C" int rs_trace_occlusion(const rs_scene* s, int n, const float* devSegments, int* devOccluded) {
    RS_SCOPE(s);
    if (!s || n < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_trace_occlusion: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_trace_occlusion, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), s->dev, n, devSegments, devOccluded);
    return rs_after_launch("rs_trace_occlusion");
}
