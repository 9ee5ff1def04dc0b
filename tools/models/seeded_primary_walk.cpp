// CPU model: would the shading ray's closest-hit walk get shorter if it started with the G-buffer ray's answer?
// GBuffer::render walks the pixel-centre ray, ReSTIRDirect the jittered ray of the same pixel (restir.cu:127-133); the two almost always hit
// the same triangle T0.  Seeded walk: test T0 first; if the jittered ray hits it at t0, prune every node whose box distance is not below
// P = t0 * (1 + 2^-6) + 2^-10 (a margin far above the rounding slop between a box distance and a triangle distance), accept triangles by the
// reference's rule (dist < closest, closest from +inf).  Every triangle at distance <= t0 lies in nodes the bound keeps, triangles the bound
// removes are farther than T0 and can only have been temporary answers of the reference's walk, so the final (triangle, distance) is the
// reference's -- the model checks that on every ray.  What it measures: nodes visited per ray, and the UNION of visited nodes per 8x8 tile,
// which is what the wave-cooperative packet walk pays (rs_scene.h trace_closest_packet), for
//   G        pixel-centre rays alone (k_render_gbuffer)
//   P        jittered rays alone, unseeded (k_primary)
//   G+P      both rays of an 8x4 tile in one wave (k_gbuffer_primary, the product's fused launch)
//   Pseed    jittered rays seeded with their pixel's G-buffer hit
// Build (from restir_amd/csrc, after `make`):
//   hipcc -O2 -std=c++17 -ffp-contract=off -I. -x hip --offload-arch=gfx950 -c ../../tools/models/seeded_primary_walk.cpp -o /tmp/sp.o
//   hipcc /tmp/sp.o scene_build.o occlusion_bvh.o api_common.o -o /tmp/sp && /tmp/sp vertices.bin     (vertices.bin: the scene's float32 vertices)
#include "rs_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
using namespace rs;

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1; fseek(f, 0, SEEK_END); int np = ftell(f) / 36; fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)np * 9); if (fread(v.data(), 36, np, f) != (size_t)np) return 1; fclose(f);
    size_t nn = 2 * (size_t)np - 1;
    std::vector<float> boxes(nn * 6); std::vector<int> nodes[6]; int* ptr[6];
    for (int k = 0; k < 6; k++) { nodes[k].resize(nn * 3); ptr[k] = nodes[k].data(); }
    int bvhSize = 0; rs_build_bvh(np, v.data(), boxes.data(), ptr, &bvhSize);
    std::vector<TriRec> tr(np);
    for (int i = 0; i < np; i++) { const float* t = &v[(size_t)i * 9]; f3 v0 = ld3(t), e1 = ld3(t + 3) - v0, e2 = ld3(t + 6) - v0; tr[i] = TriRec{ v0.x, v0.y, v0.z, 0, e1.x, e1.y, e1.z, 0, e2.x, e2.y, e2.z, 0 }; }
    rs_camera cam; memset(&cam, 0, sizeof cam); const int W = 960, H = 540;
    cam.resolution[0] = W; cam.resolution[1] = H; cam.position[0] = .5f; cam.position[1] = 2.2f; cam.position[2] = 17.f;
    cam.rotation[0] = -92.f; cam.rotation[1] = -2.f; cam.fov[1] = 30.f; cam.fov[0] = 30.f * W / H; cam.focalDist = 1.f;
    rs_camera_update(&cam); cam.tanFovY = tanf(radians(cam.fov[1]));
    CamParams cp = rs_make_cam_params(&cam);
    std::mt19937 g(1);
    auto uni = [&]() { return (float)(g() >> 8) * (1.f / 16777216.f); };
    auto tri = [&](const Ray& ray, int p, float& d) { float bx, by; const TriRec& t = tr[p];
        return tri_hit(ray.o, ray.d, mk3(t.v0x, t.v0y, t.v0z), mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z), bx, by, d); };
    // the reference's walk with a pruning bound (3.4e38: none); visited: every node fetched and box-tested
    auto walk = [&](const Ray& ray, float bound, int& prim, float& closest, std::vector<int>* visited) {
        const RayBoxCtx ctx = make_box_ctx(ray);
        const int k = mtbvh_order(-ray.d);
        prim = -1; closest = 3.402823466e+38f;
        const int* nd = nodes[k].data(); int cur = 0, count = 0;
        while (cur != bvhSize) { const int* n = nd + (size_t)cur * 3; const float* b = &boxes[(size_t)n[1] * 6]; float tb;
            count++; if (visited) visited->push_back(k * bvhSize + cur);
            if (box_hit(ctx, ld3(b), ld3(b + 3), tb) && tb < fminf(closest, bound)) {
                if (n[0] >= 0) { float d; if (tri(ray, n[0], d) && d < closest) { closest = d; prim = n[0]; } }
                cur++; }
            else cur = n[2]; }
        return count;
    };
    const int X0 = 160, Y0 = 120, WW = 640, WH = 320;                 // a window of 80 x 40 tiles around the horizon (the heavy rows)
    long raysN = 0, nG = 0, nP = 0, nS = 0, seeded = 0, mismatch = 0, sameTri = 0;
    long uG = 0, uP = 0, uS = 0, uGP = 0, tiles = 0, halfTiles = 0;
    long maxG = 0, maxP = 0, maxS = 0, maxGP = 0;
    for (int ty = 0; ty < WH / 8; ty++) for (int tx = 0; tx < WW / 8; tx++) {
        std::set<int> sG, sP, sS, sGP[2];
        for (int l = 0; l < 64; l++) {
            const int x = X0 + tx * 8 + l % 8, y = Y0 + ty * 8 + l / 8;
            const Ray rg = camera_center_ray(cp, x, y), rp = camera_sample(cp, x, y, uni(), uni());
            std::vector<int> vg, vp, vs;
            int pg, pp, ps; float dg, dp, ds;
            nG += walk(rg, 3.402823466e+38f, pg, dg, &vg);
            nP += walk(rp, 3.402823466e+38f, pp, dp, &vp);
            float bound = 3.402823466e+38f, t0;
            if (pg >= 0 && tri(rp, pg, t0)) { bound = t0 * (1.f + 0.015625f) + 0.0009765625f; seeded++; }
            nS += walk(rp, bound, ps, ds, &vs);
            if (ps != pp || memcmp(&ds, &dp, 4) != 0) mismatch++;
            if (pg == pp) sameTri++;
            raysN++;
            sG.insert(vg.begin(), vg.end()); sP.insert(vp.begin(), vp.end()); sS.insert(vs.begin(), vs.end());
            sGP[(l / 8) / 4].insert(vg.begin(), vg.end()); sGP[(l / 8) / 4].insert(vp.begin(), vp.end());
        }
        uG += (long)sG.size(); uP += (long)sP.size(); uS += (long)sS.size(); tiles++;
        for (int h = 0; h < 2; h++) { uGP += (long)sGP[h].size(); halfTiles++; maxGP = std::max(maxGP, (long)sGP[h].size()); }
        maxG = std::max(maxG, (long)sG.size()); maxP = std::max(maxP, (long)sP.size()); maxS = std::max(maxS, (long)sS.size());
    }
    printf("%ld pixels: jittered ray hits the G-buffer ray's triangle in %.3f, seeded %.3f, results differing from the reference's walk: %ld\n", raysN, (double)sameTri / raysN, (double)seeded / raysN, mismatch);
    printf("nodes visited per RAY: G %.1f  P %.1f  P seeded %.1f (%.2f of P)\n", (double)nG / raysN, (double)nP / raysN, (double)nS / raysN, (double)nS / nP);
    printf("union of visited nodes per 8x8 tile (mean / max): G %.1f / %ld   P %.1f / %ld   P seeded %.1f / %ld;   per 8x4 tile with both rays (fused launch): %.1f / %ld\n",
           (double)uG / tiles, maxG, (double)uP / tiles, maxP, (double)uS / tiles, maxS, (double)uGP / halfTiles, maxGP);
    printf("union nodes per PIXEL: two launches G + P %.2f; fused launch (product) %.2f; G + seeded P %.2f\n",
           (double)(uG + uP) / (tiles * 64), (double)uGP / (halfTiles * 32), (double)(uG + uS) / (tiles * 64));
    return 0;
}
