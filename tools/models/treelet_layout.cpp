// CPU model: how many 128-byte LINES a bounce ray's walk of the closest-hit trees touches, under the product's pre-order layout (eight
// consecutive nodes per line) and under a TREELET layout (a node, its children and grandchildren -- up to seven nodes -- per line).  The
// multi-bounce walks run at the measured gather rate of L2 and the Infinity Cache, so what can still make them faster
// is fewer lines per ray.  "Line changes" = steps whose node lies in another line than the previous step's (what misses a cache that
// holds one line per ray: with ~2 000 rays in flight per CU the 32 KB L1 holds little more).  A design study, not product code; derived
// from ordered_tree_closest_hit.cpp, whose description of the trees follows.
//
// CPU model of an EXACT closest hit (DevScene::intersect, src/scene.h:245-284) through a better tree that keeps the
// reference's visiting order.
//
// The reference's result depends on the order in which its threaded walk meets the triangles (ties, and boxes whose entry
// distance exceeds their triangle's hit distance by rounding), so a second tree can only be exact if it presents the
// triangles in the SAME order.  Every threaded order k of the MTBVH (src/bvh.cpp:132-193) is the pre-order leaf sequence of
// one tree; ANY binary tree whose leaves, read left to right, are that sequence visits the triangles in the reference's order.
// So: per order k, a tree over the sequence with every split chosen by the surface-area heuristic (a sweep over the
// sequence, cumulative -- the reference's own sweep is not, which is why its tree is poor), boxes = unions of the
// reference's leaf boxes (conservative by monotone rounding, occlusion_bvh.cpp).  The walk applies the reference's rule
// literally: enter a node iff tBox < closest; a triangle that is hit closer than `closest` is accepted iff the reference's
// own box test passes on its leaf box and every ancestor (the chain check of the shadow rays) with tLeaf < closest.
// A skipped node has tBox' >= closest with tBox' <= tLeaf of all its triangles, so the reference would not enter them either.
//
// Build (from restir_amd/csrc, after `make`):
//   hipcc -O2 -std=c++17 -ffp-contract=off -I. -x hip --offload-arch=gfx950 -c ../../tools/models/treelet_layout.cpp -o /tmp/ot.o
//   hipcc /tmp/ot.o scene_build.o occlusion_bvh.o api_common.o -o /tmp/ot && /tmp/ot vertices.bin [maxLeaf]
#include "rs_internal.h"
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
using namespace rs;

struct Bx {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    void add(const float* b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b[k]); hi[k] = std::max(hi[k], b[3 + k]); } }
    float area() const { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; }
};
struct ONode { float lo[3], hi[3]; int first, count, next; };     // count > 0: leaf over seq[first, first + count)

static int g_maxLeaf = 4;
static float g_costTri = 1.5f;
struct OBuilder {
    const float* pb; const int* seq; std::vector<ONode> nodes; std::vector<float> suf;
    void build(int first, int count) {
        Bx box; box.reset();
        for (int i = 0; i < count; i++) box.add(pb + (size_t)seq[first + i] * 6);
        const int me = (int)nodes.size(); nodes.emplace_back();
        int cut = -1;
        if (count > 1) {
            if ((int)suf.size() < count) suf.resize((size_t)count);
            Bx acc; acc.reset();
            for (int i = count - 1; i > 0; i--) { acc.add(pb + (size_t)seq[first + i] * 6); suf[(size_t)i] = acc.area(); }
            acc.reset(); float best = FLT_MAX;
            for (int i = 1; i < count; i++) {
                acc.add(pb + (size_t)seq[first + i - 1] * 6);
                const float c = acc.area() * (float)i + suf[(size_t)i] * (float)(count - i);
                if (c < best) { best = c; cut = i; }
            }
            if (count <= g_maxLeaf && !(best * g_costTri + box.area() * 2.f < box.area() * (float)count * g_costTri)) cut = -1;
        }
        ONode n; memcpy(n.lo, box.lo, 12); memcpy(n.hi, box.hi, 12);
        if (cut < 0) { n.first = first; n.count = count; n.next = me + 1; nodes[(size_t)me] = n; return; }
        build(first, cut); build(first + cut, count - cut);
        n.first = 0; n.count = 0; n.next = (int)nodes.size(); nodes[(size_t)me] = n;
    }
};

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); int np = ftell(f) / 36; fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)np * 9); if (fread(v.data(), 36, np, f) != (size_t)np) return 1; fclose(f);
    if (argc > 2) g_maxLeaf = atoi(argv[2]);
    size_t nn = 2 * (size_t)np - 1;
    std::vector<float> boxes(nn * 6); std::vector<int> nodes[6]; int* ptr[6];
    for (int k = 0; k < 6; k++) { nodes[k].resize(nn * 3); ptr[k] = nodes[k].data(); }
    int bvhSize = 0; rs_build_bvh(np, v.data(), boxes.data(), ptr, &bvhSize);
    std::vector<int> parent, leafOf; rs_reference_chain_tables(bvhSize, nodes[0].data(), parent, leafOf, np);
    std::vector<float> pb((size_t)np * 6);
    for (int p = 0; p < np; p++) memcpy(&pb[(size_t)p * 6], &boxes[(size_t)leafOf[p] * 6], 24);
    std::vector<BvhNode> on; std::vector<int> lp; rs_build_occlusion_bvh(np, pb.data(), on, lp);
    // ordered trees
    std::vector<int> seq[6]; std::vector<ONode> ot[6];
    for (int k = 0; k < 6; k++) {
        for (int i = 0; i < bvhSize; i++) if (nodes[k][(size_t)i * 3] >= 0) seq[k].push_back(nodes[k][(size_t)i * 3]);
        OBuilder b; b.pb = pb.data(); b.seq = seq[k].data(); b.nodes.reserve((size_t)np); b.build(0, np); ot[k].swap(b.nodes);
        int depthMax = 0; { std::vector<std::pair<int,int>> st; st.push_back({0, 1}); while (!st.empty()) { auto [i, d] = st.back(); st.pop_back(); depthMax = std::max(depthMax, d); if (ot[k][i].count == 0) { st.push_back({i + 1, d + 1}); st.push_back({ot[k][i + 1].next, d + 1}); } } }
        printf("order %d: %zu nodes (reference %d), depth %d\n", k, ot[k].size(), bvhSize, depthMax);
    }
    // treelet of every node: a treelet = its root, the root's children and grandchildren (leaves end a branch early); the children of the
    // inner nodes of its third level start treelets of their own
    std::vector<int> treelet[6]; size_t treelets[6];
    for (int k = 0; k < 6; k++) {
        const std::vector<ONode>& T = ot[k];
        treelet[k].assign(T.size(), -1);
        std::vector<int> roots; roots.push_back(0); int count = 0;
        while (!roots.empty()) {
            const int r = roots.back(); roots.pop_back();
            const int id = count++;
            std::vector<std::pair<int,int>> st; st.push_back({ r, 0 });
            while (!st.empty()) { auto [i, lvl] = st.back(); st.pop_back(); treelet[k][(size_t)i] = id;
                if (T[(size_t)i].count == 0) { const int l = i + 1, rr = T[(size_t)i + 1].next; if (lvl < 2) { st.push_back({ l, lvl + 1 }); st.push_back({ rr, lvl + 1 }); } else { roots.push_back(l); roots.push_back(rr); } } }
        }
        treelets[k] = (size_t)count;
        printf("order %d: %zu nodes in %d treelets of up to 7 (%.2f nodes per 128-byte line; pre-order: 8)\n", k, T.size(), count, (double)T.size() / count);
    }
    for (int k = 0; k < 6; k += 2) { bool mirror = true; for (int i = 0; i < np; i++) mirror &= seq[k][i] == seq[k + 1][np - 1 - i]; printf("orders %d / %d mirror each other: %s\n", k, k + 1, mirror ? "yes" : "NO"); }
    std::vector<TriRec> tr(np);
    for (int i = 0; i < np; i++) { const float* t = &v[(size_t)i * 9]; f3 v0 = ld3(t), e1 = ld3(t + 3) - v0, e2 = ld3(t + 6) - v0; tr[i] = TriRec{ v0.x, v0.y, v0.z, 0, e1.x, e1.y, e1.z, 0, e2.x, e2.y, e2.z, 0 }; }
    rs_camera cam; memset(&cam, 0, sizeof cam); int W = 480, H = 270;
    cam.resolution[0] = W; cam.resolution[1] = H; cam.position[0] = .5f; cam.position[1] = 2.2f; cam.position[2] = 17.f;
    cam.rotation[0] = -92.f; cam.rotation[1] = -2.f; cam.fov[1] = 30.f; cam.fov[0] = 30.f * W / H; cam.focalDist = 1.f;
    rs_camera_update(&cam); cam.tanFovY = tanf(radians(cam.fov[1]));
    CamParams cp = rs_make_cam_params(&cam);
    std::mt19937 g(1);
    auto uni = [&]() { return (float)(g() >> 8) * (1.f / 16777216.f); };
    for (int kind = 0; kind < 2; kind++) {       // 0: camera rays, 1: bounce rays from the camera rays' hit points
        long nrays = 0, stepsRef = 0, stepsOrd = 0, stepsOcc = 0, mism = 0, trisOrd = 0, cands = 0, chains = 0, linesPre = 0, linesTreelet = 0;
        for (int y = 0; y < H; y += 2) for (int x = 0; x < W; x += 2) {
            Ray ray = camera_sample(cp, x, y, uni(), uni());
            auto refWalk = [&](const Ray& ray, const RayBoxCtx& ctx, int k, int& refPrim, float& closest, long* steps) {
                refPrim = -1; closest = 3.402823466e+38f;
                const int* nd = nodes[k].data(); int cur = 0;
                while (cur != bvhSize) { const int* n = nd + (size_t)cur * 3; const float* b = &boxes[(size_t)n[1] * 6]; float tb; if (steps) (*steps)++;
                    if (box_hit(ctx, ld3(b), ld3(b + 3), tb) && tb < closest) { if (n[0] >= 0) { float bx, by, d; const TriRec& t = tr[n[0]];
                        if (tri_hit(ray.o, ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d) && d < closest) { closest = d; refPrim = n[0]; } } cur++; } else cur = n[2]; }
            };
            if (kind == 1) {
                RayBoxCtx c0 = make_box_ctx(ray); int p0; float d0; refWalk(ray, c0, mtbvh_order(-ray.d), p0, d0, nullptr);
                if (p0 < 0) continue;
                f3 pos = ray.o + ray.d * d0;
                const TriRec& t = tr[p0]; f3 n = normalize(cross(mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z)));
                if (dot(n, ray.d) > 0.f) n = -n;
                f3 dir; do { dir = mk3(uni() * 2.f - 1.f, uni() * 2.f - 1.f, uni() * 2.f - 1.f); } while (dot(dir, dir) > 1.f || dot(dir, dir) < 1e-3f);
                dir = normalize(dir); if (dot(dir, n) < 0.f) dir = -dir;
                ray.o = pos + dir * 1e-5f; ray.d = dir;
            }
            RayBoxCtx ctx = make_box_ctx(ray);
            if (ctx.mode || ctx.zx || ctx.zy || ctx.zz) continue;
            nrays++;
            const int k = mtbvh_order(-ray.d);
            int refPrim; float closest; refWalk(ray, ctx, k, refPrim, closest, &stepsRef);
            auto relaxed = [&](const float* lo, const float* hi, float& tMin) {
                float t1x = (lo[0] - ctx.o.x) * ctx.dinv.x, t1y = (lo[1] - ctx.o.y) * ctx.dinv.y, t1z = (lo[2] - ctx.o.z) * ctx.dinv.z, t2x = (hi[0] - ctx.o.x) * ctx.dinv.x, t2y = (hi[1] - ctx.o.y) * ctx.dinv.y, t2z = (hi[2] - ctx.o.z) * ctx.dinv.z;
                tMin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z)); float tMax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
                return tMax >= 0 && tMax >= tMin; };
            {   // ordered tree of order k: the reference's acceptance rule, literally
                int prim = -1; float c = 3.402823466e+38f; const std::vector<ONode>& T = ot[k]; size_t cur = 0;
                long lastPre = -1, lastTl = -1;
                while (cur != T.size()) { const ONode& n = T[cur]; stepsOrd++; float tMin;
                    if ((long)(cur / 8) != lastPre) { linesPre++; lastPre = (long)(cur / 8); }
                    if (treelet[k][cur] != lastTl) { linesTreelet++; lastTl = treelet[k][cur]; }
                    if (relaxed(n.lo, n.hi, tMin) && tMin < c) {
                        for (int j = 0; j < n.count; j++) { const int p = seq[k][(size_t)n.first + j]; trisOrd++; float bx, by, d; const TriRec& t = tr[p];
                            if (tri_hit(ray.o, ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d) && d < c) {
                                cands++; bool ok = true; float tbl = 0;
                                for (int a = leafOf[p]; a >= 0 && ok; a = parent[a]) { float tb; const float* b = &boxes[(size_t)a * 6]; ok = box_hit(ctx, ld3(b), ld3(b + 3), tb); chains++; if (a == leafOf[p]) tbl = tb; }
                                if (ok && tbl < c) { c = d; prim = p; } } }
                        cur++; } else cur = (size_t)n.next; }
                mism += prim != refPrim || (prim >= 0 && c != closest);
            }
            {   // for comparison: the shadow-ray tree with distance-only pruning (not exact)
                float c = 3.402823466e+38f; size_t cur = 0;
                while (cur != on.size()) { const BvhNode& n = on[cur]; stepsOcc++; float tMin; const float lo[3] = { n.bminx, n.bminy, n.bminz }, hi[3] = { n.bmaxx, n.bmaxy, n.bmaxz };
                    if (relaxed(lo, hi, tMin) && tMin < c) { if (n.primId >= 0) { int s0 = n.primId >> 3, cn = n.primId & 7; for (int j = 0; j < cn; j++) { int p = lp[s0 + j]; float bx, by, d; const TriRec& t = tr[p];
                            if (tri_hit(ray.o, ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d) && d < c) c = d; } } cur++; } else cur = n.next; }
            }
        }
        printf("%s: 128-byte line changes per ray: pre-order layout %.1f, treelet layout %.1f (of %.1f node steps)\n", kind ? "bounce rays" : "camera rays", (double)linesPre / nrays, (double)linesTreelet / nrays, (double)stepsOrd / nrays);
        printf("%s: rays %ld mismatches %ld | steps per ray: reference %.1f, ordered tree %.1f (%.2f triangle tests, %.2f candidates, %.2f chain steps), shadow tree distance-only %.1f\n",
               kind ? "bounce rays" : "camera rays", nrays, mism, (double)stepsRef / nrays, (double)stepsOrd / nrays, (double)trisOrd / nrays, (double)cands / nrays, (double)chains / nrays, (double)stepsOcc / nrays);
    }
    return 0;
}
