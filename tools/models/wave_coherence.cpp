// CPU model: how many distinct 128-byte LINES of the closest-hit trees' node arrays a WAVE of 64 bounce rays asks for per step when its lanes
// walk in lockstep, under different ways of putting rays into waves:
//   tile      the 64 first-bounce rays of an 8x8 pixel tile (what k_path does)
//   bucket    the rays of a 32x32 pixel block bucketed by threaded order, 64 at a time in pixel order (the wavefront form's queues, locally)
//   sorted    all rays sorted by (threaded order, cell of the origin on a 32^3 grid, direction octant), 64 at a time
//   random    64 rays drawn at random (no coherence at all)
// Lanes that ask for the same line in the same step are ONE request to the CU's L1 (tools/micro/gather_rate.hip: the L1-resident gather rate
// depends on the lanes that ask, and DI's tile-coherent shadow rays run three times faster in lockstep than desynchronised).  The model steps
// every lane once per iteration (node test, leaf triangles tested at once), counts the distinct node lines per iteration, and reports line
// requests per ray and iterations per wave.  A design study, not product code; derived from ordered_tree_closest_hit.cpp.
//
// Build (from restir_amd/csrc, after `make`):
//   hipcc -O2 -std=c++17 -ffp-contract=off -I. -x hip --offload-arch=gfx950 -c ../../tools/models/wave_coherence.cpp -o /tmp/wc.o
//   hipcc /tmp/wc.o scene_build.o occlusion_bvh.o api_common.o -o /tmp/wc && /tmp/wc vertices.bin
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
    auto refWalk = [&](const Ray& ray, const RayBoxCtx& ctx, int k, int& refPrim, float& closest) {
        refPrim = -1; closest = 3.402823466e+38f;
        const int* nd = nodes[k].data(); int cur = 0;
        while (cur != bvhSize) { const int* n = nd + (size_t)cur * 3; const float* b = &boxes[(size_t)n[1] * 6]; float tb;
            if (box_hit(ctx, ld3(b), ld3(b + 3), tb) && tb < closest) { if (n[0] >= 0) { float bx, by, d; const TriRec& t = tr[n[0]];
                if (tri_hit(ray.o, ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d) && d < closest) { closest = d; refPrim = n[0]; } } cur++; } else cur = n[2]; }
    };
    // first-bounce rays of a 256 x 128 pixel window (32 x 16 tiles of 8 x 8)
    struct BRay { Ray ray; int k; int px, py; bool valid; };
    const int WW = 256, WH = 128, X0 = 112, Y0 = 70;
    std::vector<BRay> rays((size_t)WW * WH);
    f3 lo = splat(3e38f), hi = splat(-3e38f);
    for (int y = 0; y < WH; y++) for (int x = 0; x < WW; x++) {
        BRay& br = rays[(size_t)y * WW + x]; br.px = x; br.py = y; br.valid = false;
        Ray ray = camera_sample(cp, X0 + x, Y0 + y, uni(), uni());
        RayBoxCtx c0 = make_box_ctx(ray); int p0; float d0; refWalk(ray, c0, mtbvh_order(-ray.d), p0, d0);
        if (p0 < 0) continue;
        f3 pos = ray.o + ray.d * d0;
        const TriRec& t = tr[p0]; f3 n = normalize(cross(mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z)));
        if (dot(n, ray.d) > 0.f) n = -n;
        f3 dir; do { dir = mk3(uni() * 2.f - 1.f, uni() * 2.f - 1.f, uni() * 2.f - 1.f); } while (dot(dir, dir) > 1.f || dot(dir, dir) < 1e-3f);
        dir = normalize(dir); if (dot(dir, n) < 0.f) dir = -dir;
        br.ray.o = pos + dir * 1e-5f; br.ray.d = dir;
        RayBoxCtx ctx = make_box_ctx(br.ray);
        if (ctx.mode || ctx.zx || ctx.zy || ctx.zz) continue;
        br.k = mtbvh_order(-dir); br.valid = true;
        lo = mk3(fminf(lo.x, pos.x), fminf(lo.y, pos.y), fminf(lo.z, pos.z)); hi = mk3(fmaxf(hi.x, pos.x), fmaxf(hi.y, pos.y), fmaxf(hi.z, pos.z));
    }
    // one wave of up to 64 rays in lockstep: distinct node lines per iteration
    auto wave = [&](const std::vector<int>& ids, long& lineReq, long& iters, long& laneSteps) {
        const size_t n = ids.size();
        std::vector<size_t> cur(n, 0); std::vector<float> closest(n, 3.402823466e+38f); std::vector<RayBoxCtx> ctx(n);
        for (size_t i = 0; i < n; i++) ctx[i] = make_box_ctx(rays[(size_t)ids[i]].ray);
        for (;;) {
            std::vector<long> lines;
            for (size_t i = 0; i < n; i++) { const BRay& br = rays[(size_t)ids[i]]; if (cur[i] != ot[br.k].size()) lines.push_back((long)br.k * (1l << 32) + (long)(cur[i] / 8)); }
            if (lines.empty()) break;
            laneSteps += (long)lines.size(); iters++;
            std::sort(lines.begin(), lines.end()); lineReq += (long)(std::unique(lines.begin(), lines.end()) - lines.begin());
            for (size_t i = 0; i < n; i++) {
                const BRay& br = rays[(size_t)ids[i]]; const std::vector<ONode>& T = ot[br.k];
                if (cur[i] == T.size()) continue;
                const ONode& nd = T[cur[i]];
                const RayBoxCtx& c = ctx[i];
                float t1x = (nd.lo[0] - c.o.x) * c.dinv.x, t1y = (nd.lo[1] - c.o.y) * c.dinv.y, t1z = (nd.lo[2] - c.o.z) * c.dinv.z, t2x = (nd.hi[0] - c.o.x) * c.dinv.x, t2y = (nd.hi[1] - c.o.y) * c.dinv.y, t2z = (nd.hi[2] - c.o.z) * c.dinv.z;
                const float tMin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z)), tMax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
                if (tMax >= 0 && tMax >= tMin && tMin < closest[i]) {
                    for (int j = 0; j < nd.count; j++) { const int p = seq[br.k][(size_t)nd.first + j]; float bx, by, d; const TriRec& t = tr[p];
                        if (tri_hit(br.ray.o, br.ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d) && d < closest[i]) closest[i] = d; }      // (acceptance without the chain check: a model of the fetches)
                    cur[i]++;
                } else cur[i] = (size_t)nd.next;
            }
        }
    };
    auto report = [&](const char* name, const std::vector<std::vector<int>>& waves) {
        long lineReq = 0, iters = 0, laneSteps = 0, nr = 0;
        for (const auto& w : waves) { nr += (long)w.size(); wave(w, lineReq, iters, laneSteps); }
        printf("%-7s %5zu waves, %ld rays: node steps per ray %.1f, iterations per wave %.1f (%.0f %% of the lanes stepping), distinct line requests per ray %.1f (%.2f per lane-step)\n",
               name, waves.size(), nr, (double)laneSteps / nr, (double)iters / waves.size(), 100.0 * laneSteps / (64.0 * iters), (double)lineReq / nr, (double)lineReq / laneSteps);
    };
    std::vector<std::vector<int>> waves;
    // tile
    for (int ty = 0; ty < WH / 8; ty++) for (int tx = 0; tx < WW / 8; tx++) { std::vector<int> w; for (int l = 0; l < 64; l++) { const int id = (ty * 8 + l / 8) * WW + tx * 8 + l % 8; if (rays[(size_t)id].valid) w.push_back(id); } if (!w.empty()) waves.push_back(w); }
    report("tile", waves);
    // bucket: 32 x 32 pixel blocks, bucketed by order, pixel order inside
    waves.clear();
    for (int by = 0; by < WH / 32; by++) for (int bx = 0; bx < WW / 32; bx++) for (int k = 0; k < 6; k++) {
        std::vector<int> w;
        for (int l = 0; l < 1024; l++) { const int id = (by * 32 + l / 32) * WW + bx * 32 + l % 32; if (rays[(size_t)id].valid && rays[(size_t)id].k == k) { w.push_back(id); if (w.size() == 64) { waves.push_back(w); w.clear(); } } }
        if (!w.empty()) waves.push_back(w);
    }
    report("bucket", waves);
    // sorted: order, origin cell (32^3), direction octant
    {
        std::vector<std::pair<unsigned long long, int>> keyed;
        for (size_t i = 0; i < rays.size(); i++) if (rays[i].valid) {
            const BRay& br = rays[i];
            auto cell = [&](float v, float a, float b) { int c = (int)((v - a) / (b - a + 1e-6f) * 32.f); return (unsigned long long)std::min(31, std::max(0, c)); };
            const unsigned long long cx = cell(br.ray.o.x, lo.x, hi.x), cy = cell(br.ray.o.y, lo.y, hi.y), cz = cell(br.ray.o.z, lo.z, hi.z);
            unsigned long long m = 0; for (int b = 0; b < 5; b++) m |= ((cx >> b & 1) << (3 * b)) | ((cy >> b & 1) << (3 * b + 1)) | ((cz >> b & 1) << (3 * b + 2));
            const unsigned long long oct = (br.ray.d.x < 0) | (br.ray.d.y < 0) << 1 | (br.ray.d.z < 0) << 2;
            keyed.push_back({ ((unsigned long long)br.k << 40) | (m << 3) | oct, (int)i });
        }
        std::sort(keyed.begin(), keyed.end());
        waves.clear(); std::vector<int> w;
        for (size_t i = 0; i < keyed.size(); i++) { if (!w.empty() && (keyed[i].first >> 40) != (keyed[i - 1].first >> 40)) { waves.push_back(w); w.clear(); } w.push_back(keyed[i].second); if (w.size() == 64) { waves.push_back(w); w.clear(); } }
        if (!w.empty()) waves.push_back(w);
        report("sorted", waves);
        // random
        std::vector<int> all; for (auto& kv : keyed) all.push_back(kv.second);
        std::shuffle(all.begin(), all.end(), g);
        waves.clear(); w.clear();
        for (int id : all) { w.push_back(id); if (w.size() == 64) { waves.push_back(w); w.clear(); } }
        if (!w.empty()) waves.push_back(w);
        report("random", waves);
    }
    return 0;
}
