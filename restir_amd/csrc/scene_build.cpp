// scene_build.cpp -- host side of the scene: what Scene::buildDevData does before the upload
// (src/scene.cpp:159-215).  The kernels' inputs are DEFINED by these arrays, so each builder
// reproduces the reference's result exactly (same float operations in the same order, same tie
// and degenerate-case behaviour) while being organised around index permutations and flat
// float arrays rather than the reference's arrays of structs.
//
//   rs_build_bvh          <- BVHBuilder::build + buildMTBVH   (src/bvh.cpp:10-202)
//   rs_build_alias_table  <- DiscreteSampler1D<float> ctor     (src/sampler.h:79-121)
//   rs_build_light_table  <- light loop of buildDevData        (src/scene.cpp:161-190)
//   rs_camera_update      <- Camera::update                    (src/sceneStructs.h:88-102)
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "rs_internal.h"

using namespace rs;

namespace {

struct Box { f3 lo, hi; };

inline Box empty_box() { Box b; b.lo = splat(FLT_MAX); b.hi = splat(-FLT_MAX); return b; }   // bvh.h:159-160
inline void grow(Box& b, const Box& o) { b.lo = vmin(b.lo, o.lo); b.hi = vmax(b.hi, o.hi); }    // bvh.h:30-32
inline void grow(Box& b, f3 p) { b.lo = vmin(b.lo, p); b.hi = vmax(b.hi, p); }                  // bvh.h:26-28
inline f3 centre(const Box& b) { return (b.lo + b.hi) * .5f; }                                   // bvh.h:47-49
inline float area(const Box& b) {                                                                // bvh.h:51-54
    f3 s = b.hi - b.lo;
    return 2.f * (s.x * s.y + s.y * s.z + s.z * s.x);
}
inline int widest_axis(const Box& b) {                                                           // bvh.h:59-67
    f3 s = b.hi - b.lo;
    if (s.x < s.y) return s.y > s.z ? 1 : 2;
    return s.x > s.z ? 0 : 2;
}
inline float axis(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// int(float) as the reference's HOST code gets it on x86-64 (cvttss2si): NaN and out-of-range
// values give INT_MIN.  bvh.cpp:83,118 rely on this when dimMax == dimMin (0/0 -> bucket 0).
inline int host_f2i(float f) {
    if (!(f == f) || f >= 2147483648.f || f < -2147483648.f) return INT_MIN;
    return (int)f;
}

constexpr int kBuckets = 16;   // bvh.cpp:34

}  // namespace

extern "C" int rs_build_bvh(int numPrims, const float* vertices, float* boundingBoxes, int* const bvhNodes[6], int* bvhSizeOut) {
    if (numPrims <= 0 || !vertices || !boundingBoxes || !bvhNodes) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_bvh: bad argument");
    const int total = 2 * numPrims - 1;

    // per-primitive bounds / centroids (SoA) and the permutation the partitioning acts on
    std::vector<Box> pbox((size_t)numPrims);
    std::vector<f3> pctr((size_t)numPrims);
    std::vector<int> order((size_t)numPrims), scratch((size_t)numPrims);
    for (int i = 0; i < numPrims; i++) {
        const float* t = vertices + (size_t)i * 9;
        f3 a = ld3(t), b = ld3(t + 3), c = ld3(t + 6);
        pbox[i].lo = vmin(vmin(a, b), c);
        pbox[i].hi = vmax(vmax(a, b), c);
        pctr[i] = centre(pbox[i]);
        order[i] = i;
    }

    // flattened pre-order tree: subtreeSize (1 = leaf) and leaf primitive per node
    std::vector<int> subtree((size_t)total), leafPrim((size_t)total, -1);
    std::vector<Box> nodeBox((size_t)total);

    struct Work { int node, first, last; };
    std::vector<Work> todo;
    todo.reserve(64);
    todo.push_back({ 0, 0, numPrims - 1 });

    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        const int count = w.last - w.first + 1;
        subtree[w.node] = 2 * count - 1;

        Box all = empty_box(), ctr = empty_box();
        for (int i = w.first; i <= w.last; i++) {
            grow(all, pbox[order[i]]);
            grow(ctr, pctr[order[i]]);
        }
        nodeBox[w.node] = all;
        if (count == 1) {
            leafPrim[w.node] = order[w.first];
            continue;
        }

        const int ax = widest_axis(ctr);
        const float lo = axis(ctr.lo, ax), hi = axis(ctr.hi, ax);
        auto bucket_of = [&](int prim) {
            return iclamp(host_f2i((axis(pctr[prim], ax) - lo) / (hi - lo) * (float)kBuckets), 0, kBuckets - 1);
        };

        Box bb[kBuckets];
        int bn[kBuckets];
        for (int k = 0; k < kBuckets; k++) { bb[k] = empty_box(); bn[k] = 0; }
        for (int i = w.first; i <= w.last; i++) {
            int k = bucket_of(order[i]);
            grow(bb[k], pbox[order[i]]);
            bn[k]++;
        }

        // Split cost.  As in the reference (bvh.cpp:92-100) the "left" box of split k is the single
        // bucket k-1 (bucket 0 for k = 0) and the "right" box is the single bucket k+2 (bucket 15
        // for k+1 = 15): the sweeps there union into a fresh empty box, not a running one.
        float best = FLT_MAX;
        int cut = 0, seen = 0;
        for (int k = 0; k < kBuckets - 1; k++) {
            seen += bn[k];
            const Box& L = bb[k == 0 ? 0 : k - 1];
            const Box& R = bb[k + 1 == kBuckets - 1 ? kBuckets - 1 : k + 2];
            float cost = mixf(area(L), area(R), (float)seen / (float)count);
            if (cost < best) { best = cost; cut = k; }
        }

        // stable-left / reversed-right partition (bvh.cpp:113-121)
        std::memcpy(scratch.data(), order.data() + w.first, sizeof(int) * (size_t)count);
        int l = w.first, r = w.last;
        for (int i = 0; i < count; i++) {
            int prim = scratch[i];
            if (bucket_of(prim) <= cut) order[l++] = prim; else order[r--] = prim;
        }
        const int mid = iclamp(l - 1, w.first, w.last - 1);
        const int leftNodes = 2 * (mid - w.first + 1) - 1;
        todo.push_back({ w.node + 1 + leftNodes, mid + 1, w.last });   // right, handled second
        todo.push_back({ w.node + 1, w.first, mid });                  // left, handled first
    }

    for (int i = 0; i < total; i++) {
        st3(boundingBoxes + (size_t)i * 6, nodeBox[i].lo);
        st3(boundingBoxes + (size_t)i * 6 + 3, nodeBox[i].hi);
    }

    // six threaded orders (bvh.cpp:156-193): pre-order walk visiting the child whose box centre is
    // greater (even order) / lesser (odd order) along axis order/2 first; miss link = end of subtree
    std::vector<int> walk;
    walk.reserve(128);
    for (int ord = 0; ord < 6; ord++) {
        int* out = bvhNodes[ord];
        const int ax = ord / 2;
        const bool lesser = (ord & 1) != 0;
        int emitted = 0;
        walk.clear();
        walk.push_back(0);
        while (!walk.empty()) {
            const int n = walk.back();
            walk.pop_back();
            const int size = subtree[n];
            out[(size_t)emitted * 3 + 0] = size == 1 ? leafPrim[n] : kNullPrim;
            out[(size_t)emitted * 3 + 1] = n;
            out[(size_t)emitted * 3 + 2] = emitted + size;
            emitted++;
            if (size == 1) continue;
            int a = n + 1, b = n + 1 + subtree[n + 1];
            if ((axis(centre(nodeBox[a]), ax) < axis(centre(nodeBox[b]), ax)) != lesser) { int t = a; a = b; b = t; }
            walk.push_back(b);
            walk.push_back(a);
        }
    }
    if (bvhSizeOut) *bvhSizeOut = total;
    return 0;
}

extern "C" int rs_build_alias_table(int n, const float* valuesIn, float* prob, int* failId, float* sumAllOut) {
    if (n < 0 || (n > 0 && (!valuesIn || !prob || !failId))) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_alias_table: bad argument");
    float total = 0.f;
    for (int i = 0; i < n; i++) total += valuesIn[i];
    const float norm = (float)n / total;

    // two LIFO work lists of (scaled probability, index); the pop order defines the table
    struct Entry { float p; int id; };
    std::vector<Entry> heavy, light;
    heavy.reserve((size_t)n); light.reserve((size_t)n * 2);
    for (int i = 0; i < n; i++) {
        float v = valuesIn[i] * norm;
        (v > 1.f ? heavy : light).push_back({ v, i });
    }
    while (!heavy.empty() && !light.empty()) {
        Entry big = heavy.back(); heavy.pop_back();
        Entry small = light.back(); light.pop_back();
        prob[small.id] = small.p;
        failId[small.id] = big.id;
        big.p -= (1.f - small.p);
        (big.p > 1.f ? heavy : light).push_back(big);
    }
    for (size_t i = heavy.size(); i-- > 0;) { prob[heavy[i].id] = heavy[i].p; failId[heavy[i].id] = heavy[i].id; }
    for (size_t i = light.size(); i-- > 0;) { prob[light[i].id] = light[i].p; failId[light[i].id] = light[i].id; }
    if (sumAllOut) *sumAllOut = total;
    return 0;
}

extern "C" int rs_build_light_table(int numPrims, const float* vertices, const int* materialIds,
                                    int numMaterials, const rs_material* materials, int* numLights,
                                    int* lightPrimIds, float* lightUnitRadiance, float* lightPower) {
    if (numPrims < 0 || !vertices || !materialIds || !materials || !numLights) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_light_table: bad argument");
    int n = 0;
    for (int p = 0; p < numPrims; p++) {
        int m = materialIds[p];
        if (m < 0 || m >= numMaterials) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_light_table: material id out of range");
        if (materials[m].type != 4) continue;                        // Material::Type::Light
        f3 Le = ld3(materials[m].baseColor);
        float perArea = luminance(Le) * 2.f * kGlmPi;                 // scene.cpp:164
        const float* t = vertices + (size_t)p * 9;
        f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
        float a = length(cross(v1 - v0, v2 - v0)) * .5f;              // Math::triangleArea
        lightPrimIds[n] = p;
        st3(lightUnitRadiance + (size_t)n * 3, Le);
        lightPower[n] = perArea * a;
        n++;
    }
    *numLights = n;
    return 0;
}

// pdf of Scene::createLightSampler's environment-map sampler (src/scene.cpp:139-146); host code in the reference too
extern "C" int rs_build_envmap_pdf(int width, int height, const float* data, float* pdf) {
    if (width <= 0 || height <= 0 || !data || !pdf) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_envmap_pdf: bad argument");
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int idx = i * width + j;
            pdf[idx] = luminance(ld3(data + (size_t)idx * 3)) * sinf((.5f + (float)i) / (float)height * kPi);
        }
    return 0;
}

extern "C" int rs_camera_update(rs_camera* c) {
    if (!c) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_camera_update: null camera");
    float yaw = radians(c->rotation[0]);
    float pitch = radians(c->rotation[1]);
    f3 view;
    view.x = cosf(yaw) * cosf(pitch);
    view.z = sinf(yaw) * cosf(pitch);
    view.y = sinf(pitch);
    view = normalize(view);
    f3 right = normalize(cross(view, mk3(0.f, 1.f, 0.f)));
    f3 up = normalize(cross(right, view));
    st3(c->view, view); st3(c->right, right); st3(c->up, up);

    // glm::inverse(mat3(right, up, view)) -- cofactor form of type_mat3x3.inl:37-56; m[col][row]
    const float m[3][3] = { { right.x, right.y, right.z }, { up.x, up.y, up.z }, { view.x, view.y, view.z } };
    const float det = + m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
                      - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
                      + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
    const float k = 1.f / det;
    float* o = c->rotationMatInv;         // o[col*3 + row]
    o[0] = + (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * k;
    o[3] = - (m[1][0] * m[2][2] - m[2][0] * m[1][2]) * k;
    o[6] = + (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * k;
    o[1] = - (m[0][1] * m[2][2] - m[2][1] * m[0][2]) * k;
    o[4] = + (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * k;
    o[7] = - (m[0][0] * m[2][1] - m[2][0] * m[0][1]) * k;
    o[2] = + (m[0][1] * m[1][2] - m[1][1] * m[0][2]) * k;
    o[5] = - (m[0][0] * m[1][2] - m[1][0] * m[0][2]) * k;
    o[8] = + (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * k;
    return 0;
}
