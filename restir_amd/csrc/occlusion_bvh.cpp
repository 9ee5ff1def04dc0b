// occlusion_bvh.cpp -- host build of the second acceleration structure used ONLY by shadow rays.
//
// DevScene::testOcclusion (src/scene.h:286-316) answers "is there a triangle T such that (a) every
// node on the path from the root of the reference's MTBVH to T's leaf passes the reference's box test
// with tBox < range and (b) intersectTriangle(T) hits closer than range" -- the visiting order does
// not matter.  The reference's tree is poor for this (its SAH sweep is not cumulative, src/bvh.cpp:92-100:
// depth up to 52, 86 node visits per shadow ray on the Sponza-class scene), so shadow rays here look
// for candidates T satisfying (b) in a compact, well-built tree with CONSERVATIVE (slightly inflated)
// boxes and then check (a) for each candidate by walking T's ancestor chain in the reference's own
// tree with the reference's exact box test.  A conservative tree finds every T with (b), therefore
// OR over candidates of (a) is exactly the reference's answer (rs_scene.h trace_occluded_fast).
//
// "Conservative" needs no epsilon: the primitive bounds used here ARE the reference's leaf boxes and every
// box of this tree is an exact (min/max) union of them, so it contains the leaf box L of each of its
// triangles.  IEEE rounding is monotone, hence for a box B' containing L the slab distances computed
// with the reference's own expression (b - o) * (1/d) satisfy tNear' <= tNear and tFar' >= tFar on every
// axis, and  tMax >= 0 && tMax >= tMin && tMin < range  for L (part of the reference's test, bvh.h:124-156)
// implies the same for B'.  The walk of this tree uses exactly that relaxed test, so it reaches every
// triangle whose reference leaf box the ray passes; the rest of (a) -- the reference's extra overlap
// conditions at the leaf and all of its ancestors -- is what the chain check evaluates.  Rays that take one
// of the reference's special cases (axis-aligned / near-zero components / NaN) use the reference walk.
//
// Built here: binned-SAH BVH2, up to 4 triangles per leaf, pre-order with miss links (stackless,
// one order: any-hit does not care), and the reference-chain tables (parent of every reference node,
// leaf node of every primitive).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "rs_internal.h"

using namespace rs;

namespace {

struct Bx {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    void add(const Bx& o) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    void add(const float* p) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16, kMaxLeaf = 4;
constexpr float kCostTri = 1.5f, kCostBox = 1.f;     // a leaf test is three 16-byte loads + ~45 VALU ops, a box 16 B + ~30 (swept on the
                                                     // benchmark scene: cost 1.0-4.0 x leaf size 2-6 stay within 0.5 % of this, cost 1.0 is 3 % slower)

struct Builder {
    std::vector<Bx> pb;            // per-primitive bounds
    std::vector<float> pc;         // per-primitive centroids (3 each)
    std::vector<int> ids;          // permutation
    std::vector<BvhNode> nodes;    // output, pre-order
    std::vector<int> leafPrims;    // primitive ids in leaf order

    // returns index of the emitted node
    int build(int first, int count) {
        Bx box; box.reset();
        Bx cb; cb.reset();
        for (int i = first; i < first + count; i++) { box.add(pb[ids[i]]); cb.add(&pc[(size_t)ids[i] * 3]); }
        const int me = (int)nodes.size();
        nodes.emplace_back();
        auto make_leaf = [&]() {
            const int start = (int)leafPrims.size();
            for (int i = first; i < first + count; i++) leafPrims.push_back(ids[i]);
            set(me, box, start * 8 + count);
        };
        if (count <= 1) { make_leaf(); return me; }

        // binned SAH, all three axes
        int ax = 0;
        for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;      // fallback axis
        int mid = -1;
        {
            float best = FLT_MAX; int cut = -1, cutAx = -1;
            for (int a = 0; a < 3; a++) {
                const float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.f)) continue;
                Bx bb[kBins]; int bn[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); bn[b] = 0; }
                const float scale = (float)kBins / ext;
                for (int i = first; i < first + count; i++) {
                    const int b = std::min(kBins - 1, std::max(0, (int)((pc[(size_t)ids[i] * 3 + a] - cb.lo[a]) * scale)));
                    bb[b].add(pb[ids[i]]); bn[b]++;
                }
                float rightArea[kBins]; int rightCnt[kBins];
                Bx acc; acc.reset(); int c = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.add(bb[b]); c += bn[b]; rightArea[b] = c ? acc.half_area() : 0.f; rightCnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.add(bb[b]); c += bn[b];
                    if (c == 0 || rightCnt[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)c + rightArea[b + 1] * (float)rightCnt[b + 1];
                    if (cost < best) { best = cost; cut = b; cutAx = a; }
                }
            }
            const float leafCost = box.half_area() * (float)count * kCostTri;
            if (cut >= 0 && (count > kMaxLeaf || best * kCostTri + box.half_area() * 2.f * kCostBox < leafCost)) {
                const float lo = cb.lo[cutAx], scale = (float)kBins / (cb.hi[cutAx] - cb.lo[cutAx]);
                auto it = std::partition(ids.begin() + first, ids.begin() + first + count, [&](int prim) {
                    return std::min(kBins - 1, std::max(0, (int)((pc[(size_t)prim * 3 + cutAx] - lo) * scale))) <= cut; });
                mid = (int)(it - ids.begin());
            }
        }
        if (mid < 0) {
            if (count <= kMaxLeaf) { make_leaf(); return me; }
            mid = first + count / 2;                                   // degenerate centroids: median split
            std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count,
                             [&](int a, int b) { return pc[(size_t)a * 3 + ax] < pc[(size_t)b * 3 + ax]; });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        // the child with the larger box first: more likely to hold an occluder
        Bx lb, rb; lb.reset(); rb.reset();
        for (int i = first; i < mid; i++) lb.add(pb[ids[i]]);
        for (int i = mid; i < first + count; i++) rb.add(pb[ids[i]]);
        if (rb.half_area() > lb.half_area()) {
            std::rotate(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count);
            mid = first + (first + count - mid);
        }
        build(first, mid - first);
        build(mid, first + count - mid);
        set(me, box, -1);
        return me;
    }

    void set(int node, const Bx& b, int leaf) {
        BvhNode& n = nodes[(size_t)node];
        n.bminx = b.lo[0]; n.bminy = b.lo[1]; n.bminz = b.lo[2];
        n.bmaxx = b.hi[0]; n.bmaxy = b.hi[1]; n.bmaxz = b.hi[2];
        n.primId = leaf;                     // -1 inner, else firstTriangle * 8 + count
        n.next = (int)nodes.size();          // pre-order: everything emitted after `node` so far is its subtree
    }
};

}  // namespace

// primBoxes: 6 floats (min, max) per primitive = the reference's leaf boxes
int rs_build_occlusion_bvh(int numPrims, const float* primBoxes, std::vector<BvhNode>& nodes, std::vector<int>& leafPrims) {
    if (numPrims <= 0 || !primBoxes) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_occlusion_bvh: bad argument");
    Builder b;
    b.pb.resize((size_t)numPrims); b.pc.resize((size_t)numPrims * 3); b.ids.resize((size_t)numPrims);
    for (int i = 0; i < numPrims; i++) {
        const float* t = primBoxes + (size_t)i * 6;
        b.pb[i].reset(); b.pb[i].add(t); b.pb[i].add(t + 3);
        for (int k = 0; k < 3; k++) b.pc[(size_t)i * 3 + k] = 0.5f * (b.pb[i].lo[k] + b.pb[i].hi[k]);
        b.ids[i] = i;
    }
    b.nodes.reserve((size_t)numPrims);
    b.leafPrims.reserve((size_t)numPrims);
    b.build(0, numPrims);
    nodes.swap(b.nodes);
    leafPrims.swap(b.leafPrims);
    return 0;
}

// ---- 16-byte nodes -------------------------------------------------------------------------------
// Any box that CONTAINS a node's exact box keeps the walk conservative, so the device tree stores boxes
// on a 16-bit grid over the scene bounds: plane = base + q * scale.  The walk evaluates the slab distance
// of such a plane as fma(q, scale/d, (base - o)/d), which is not the reference's expression, so here the
// containment carries a margin instead of the exact monotonicity argument above: with D = the largest
// |coordinate difference| involved (<= 4 extents, enforced per ray on the device), both this value and the
// reference's fl(fl(p - o) * (1/d)) are within 2^-21 * D / |d| of the real-number distance of their
// plane, i.e. within 2^-21 * D in space; the grid planes are moved outward until they clear the exact
// box by 2^-17 * extent >= 4 * 2^-21 * D * 2 on every axis (checked here in double), a fraction of one
// grid step (2^-16 * extent).
int rs_quantize_occlusion_bvh(const std::vector<BvhNode>& nodes, float base[3], float scale[3], std::vector<unsigned>& out) {
    if (nodes.empty()) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_quantize_occlusion_bvh: empty tree");
    const BvhNode& root = nodes[0];
    const float lo[3] = { root.bminx, root.bminy, root.bminz }, hi[3] = { root.bmaxx, root.bmaxy, root.bmaxz };
    float ext = 0.f;
    for (int k = 0; k < 3; k++) ext = std::max(ext, hi[k] - lo[k]);
    if (!(ext > 0.f) || !std::isfinite(ext)) return rs_fail(RS_ERR_UNSUPPORTED, "rs_quantize_occlusion_bvh: degenerate scene bounds");
    const double margin = std::ldexp((double)ext, -17);
    for (int k = 0; k < 3; k++) {
        scale[k] = ext / 65000.f;                       // one cubic grid: same resolution on every axis
        base[k] = lo[k] - 200.f * scale[k];
    }
    out.resize(nodes.size() * 4);
    for (size_t i = 0; i < nodes.size(); i++) {
        const BvhNode& n = nodes[i];
        const float blo[3] = { n.bminx, n.bminy, n.bminz }, bhi[3] = { n.bmaxx, n.bmaxy, n.bmaxz };
        unsigned ql[3], qh[3];
        for (int k = 0; k < 3; k++) {
            long a = (long)std::floor(((double)blo[k] - margin - (double)base[k]) / (double)scale[k]);
            long b = (long)std::ceil(((double)bhi[k] + margin - (double)base[k]) / (double)scale[k]);
            while ((double)base[k] + (double)a * (double)scale[k] > (double)blo[k] - margin) a--;
            while ((double)base[k] + (double)b * (double)scale[k] < (double)bhi[k] + margin) b++;
            if (a < 0 || b > 65535) return rs_fail(RS_ERR_UNSUPPORTED, "rs_quantize_occlusion_bvh: box outside the grid");
            ql[k] = (unsigned)a; qh[k] = (unsigned)b;
        }
        unsigned* o = &out[i * 4];
        o[0] = ql[0] | (ql[1] << 16);
        o[1] = ql[2] | (qh[0] << 16);
        o[2] = qh[1] | (qh[2] << 16);
        o[3] = n.primId >= 0 ? ~(unsigned)n.primId : (unsigned)n.next * 16u;      // leaf: ~code (sign bit set), inner: miss link as a byte offset
    }
    return 0;
}

// parent of every reference node (indexed by the ORIGINAL pre-order id = MTBVHNode::boundingBoxId) and
// the leaf node of every primitive, derived from one threaded order (src/bvh.cpp:156-193: order 0).
int rs_reference_chain_tables(int bvhSize, const int* order0 /* 3 ints per node */, std::vector<int>& parent, std::vector<int>& leafOfPrim, int numPrims) {
    parent.assign((size_t)bvhSize, -1);
    leafOfPrim.assign((size_t)numPrims, -1);
    // order0[i] = {prim, origId, next}; subtree of i = [i, next); children of an inner node: i+1 and next(i+1)
    // (a caller-supplied table is only known to have forward links: every index is checked before it is used, and a table in
    // which some node is reached twice is not a tree)
    std::vector<int> stack;
    std::vector<char> seen((size_t)bvhSize, 0);
    stack.push_back(0);
    size_t visits = 0;
    while (!stack.empty()) {
        const int i = stack.back(); stack.pop_back();
        if (++visits > (size_t)bvhSize) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: the threaded order is not a tree");
        const int prim = order0[(size_t)i * 3], orig = order0[(size_t)i * 3 + 1];
        if (orig < 0 || orig >= bvhSize || seen[(size_t)orig]) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: boundingBoxId out of range or used twice");
        seen[(size_t)orig] = 1;
        if (prim >= 0) { if (prim < numPrims) leafOfPrim[(size_t)prim] = orig; continue; }
        const int a = i + 1;
        if (a >= bvhSize) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: inner node without children");
        const int b = order0[(size_t)a * 3 + 2];
        if (b <= a || b >= bvhSize) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: malformed tree");
        const int oa = order0[(size_t)a * 3 + 1], ob = order0[(size_t)b * 3 + 1];
        if (oa < 0 || oa >= bvhSize || ob < 0 || ob >= bvhSize) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: boundingBoxId out of range");
        parent[(size_t)oa] = orig;
        parent[(size_t)ob] = orig;
        stack.push_back(b); stack.push_back(a);
    }
    for (int p = 0; p < numPrims; p++) if (leafOfPrim[(size_t)p] < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_reference_chain_tables: primitive without a leaf");
    return 0;
}

// ---- closest-hit trees that keep the reference's visiting order ---------------------------------------------------------------
// DevScene::intersect (src/scene.h:245-284) depends on the ORDER in which its threaded walk meets the triangles: a triangle is
// accepted iff its leaf is entered (box test passes with tBox < closest AT THAT MOMENT) and it is hit closer than closest at
// that moment -- ties and boxes whose entry distance exceeds their triangle's hit distance by rounding are settled by who comes
// first.  A second tree can therefore only be exact if it presents the triangles in the same order.  Each threaded order k
// (src/bvh.cpp:156-193) is the pre-order leaf sequence of one tree, and ANY binary tree whose leaves, read in walking order, are
// that sequence visits the triangles in the reference's order.  Built here: over the leaf sequence of an order, every split
// chosen by a surface-area sweep along the SEQUENCE (cumulative, where the reference's bucket sweep is not, src/bvh.cpp:92-100;
// its tree costs 129 / 168 node visits per ray on the Sponza- / Bistro-class scene, this one 113 / 135 at half the bytes per
// node, tools/models/ordered_tree_closest_hit.cpp), leaves of up to 4 consecutive triangles, boxes = exact unions of the
// reference's leaf boxes (conservative for the relaxed slab test, see the top of this file).  The walk (rs_scene.h
// walk_ordered_tree) then applies the reference's rule literally: enter a node iff tBox' < closest -- a skipped node has
// tBox' >= closest and tBox' <= tLeaf of all its triangles, so the reference would not enter those leaves either -- and accept a
// triangle hit closer than closest iff the reference's own test passes on its leaf box with tLeaf < closest and on every ancestor.
// Orders 2a and 2a + 1 walk the same tree with the children swapped at EVERY inner node (src/bvh.cpp:186-190: the comparison is
// xor-ed with `lesser`), so their leaf sequences are mirror images: one tree per axis, emitted in forward and in mirrored pre-order.
// seq: the numPrims primitive ids in the order the even threaded order meets them.  Leaves: code = start * 8 + count with
// `start` the index INTO seq of the triangle met first; the mirrored order meets a leaf's triangles at start, start - 1, ...
int rs_build_ordered_bvh(int numPrims, const float* primBoxes, const int* seq, std::vector<BvhNode>& forward, std::vector<BvhNode>& mirrored) {
    if (numPrims <= 0 || !primBoxes || !seq) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_ordered_bvh: bad argument");
    struct Tmp { Bx box; int left, right, first, count; };
    std::vector<Tmp> t;
    t.reserve((size_t)numPrims);
    struct Job { int node, first, count, depth; };
    std::vector<Job> jobs;
    std::vector<float> suffix((size_t)numPrims);
    t.emplace_back(); jobs.push_back(Job{ 0, 0, numPrims, 0 });
    while (!jobs.empty()) {
        const Job j = jobs.back(); jobs.pop_back();
        Bx box; box.reset();
        for (int i = 0; i < j.count; i++) box.add(primBoxes + (size_t)seq[j.first + i] * 6), box.add(primBoxes + (size_t)seq[j.first + i] * 6 + 3);
        int cut = -1;
        if (j.count > 1) {
            if (j.depth >= 64) cut = j.count / 2;        // any split of the sequence is a valid tree: a sequence that drives the sweep into a chain gets halves from here on
            else {
                Bx acc; acc.reset();
                for (int i = j.count - 1; i > 0; i--) { const float* b = primBoxes + (size_t)seq[j.first + i] * 6; acc.add(b); acc.add(b + 3); suffix[(size_t)i] = acc.half_area(); }
                acc.reset();
                float best = FLT_MAX;
                for (int i = 1; i < j.count; i++) {
                    const float* b = primBoxes + (size_t)seq[j.first + i - 1] * 6; acc.add(b); acc.add(b + 3);
                    const float c = acc.half_area() * (float)i + suffix[(size_t)i] * (float)(j.count - i);
                    if (c < best) { best = c; cut = i; }
                }
                if (j.count <= kMaxLeaf && !(best * kCostTri + box.half_area() * 2.f * kCostBox < box.half_area() * (float)j.count * kCostTri)) cut = -1;
                if (cut < 0 && j.count > kMaxLeaf) cut = j.count / 2;      // (areas that are not finite: never for boxes that passed build_occlusion_side)
            }
        }
        Tmp& me = t[(size_t)j.node];
        me.box = box; me.first = j.first; me.count = cut < 0 ? j.count : 0; me.left = me.right = -1;
        if (cut >= 0) {
            const int l = (int)t.size(), r = l + 1;
            t[(size_t)j.node].left = l; t[(size_t)j.node].right = r;
            t.emplace_back(); t.emplace_back();
            jobs.push_back(Job{ r, j.first + cut, j.count - cut, j.depth + 1 });
            jobs.push_back(Job{ l, j.first, cut, j.depth + 1 });
        }
    }
    // the two pre-orders; next = the record after the subtree
    std::vector<int> size(t.size(), 1);
    for (size_t i = t.size(); i-- > 0;) if (t[i].left >= 0) size[i] = 1 + size[(size_t)t[i].left] + size[(size_t)t[i].right];     // children are created after their parent
    for (int dir = 0; dir < 2; dir++) {
        std::vector<BvhNode>& out = dir ? mirrored : forward;
        out.clear(); out.reserve(t.size());
        std::vector<int> stack; stack.push_back(0);
        while (!stack.empty()) {
            const int i = stack.back(); stack.pop_back();
            const Tmp& n = t[(size_t)i];
            BvhNode o;
            o.bminx = n.box.lo[0]; o.bminy = n.box.lo[1]; o.bminz = n.box.lo[2]; o.bmaxx = n.box.hi[0]; o.bmaxy = n.box.hi[1]; o.bmaxz = n.box.hi[2];
            o.primId = n.count ? (dir ? (n.first + n.count - 1) * 8 + n.count : n.first * 8 + n.count) : -1;
            o.next = (int)out.size() + size[(size_t)i];
            out.push_back(o);
            if (n.left >= 0) { if (dir) { stack.push_back(n.left); stack.push_back(n.right); } else { stack.push_back(n.right); stack.push_back(n.left); } }
        }
    }
    return 0;
}

// Host-only check of the above for the CPU test suite: from a reference table (boxes by original node id, six threaded orders)
// derives the leaf sequences, requires the odd orders to mirror the even ones, builds the three trees and verifies what the walk
// relies on -- read in walking order the leaves of the forward / mirrored tree list the triangles exactly as order 2a / 2a + 1
// meets them, every box contains the reference leaf boxes below it, every miss link is the end of its subtree.
// nodeCounts[3], maxDepth[3]: per axis.  Returns 0, RS_ERR_UNSUPPORTED when the orders do not mirror, or RS_ERR_INTERNAL on a broken tree.
extern "C" int rs_ordered_bvh_host_check(int numPrims, int bvhSize, const float* boundingBoxes, const int* const bvhNodes[6], int* nodeCounts, int* maxDepth) {
    if (numPrims <= 0 || bvhSize != 2 * numPrims - 1 || !boundingBoxes || !bvhNodes) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_ordered_bvh_host_check: bad argument");
    std::vector<int> parent, leafOf;
    if (int e = rs_reference_chain_tables(bvhSize, bvhNodes[0], parent, leafOf, numPrims)) return e;
    const size_t np = (size_t)numPrims;
    std::vector<float> pb(np * 6);
    for (size_t p = 0; p < np; p++) std::memcpy(&pb[p * 6], boundingBoxes + (size_t)leafOf[p] * 6, 24);
    for (int a = 0; a < 3; a++) {
        std::vector<int> seq, rev;
        for (int i = 0; i < bvhSize; i++) { if (bvhNodes[2 * a][(size_t)i * 3] >= 0) seq.push_back(bvhNodes[2 * a][(size_t)i * 3]); if (bvhNodes[2 * a + 1][(size_t)i * 3] >= 0) rev.push_back(bvhNodes[2 * a + 1][(size_t)i * 3]); }
        if (seq.size() != np || rev.size() != np) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_ordered_bvh_host_check: an order does not list every primitive once");
        for (size_t i = 0; i < np; i++) if (seq[i] != rev[np - 1 - i]) return rs_fail(RS_ERR_UNSUPPORTED, "rs_ordered_bvh_host_check: the odd order is not the mirror image of the even one");
        std::vector<BvhNode> tree[2];
        if (int e = rs_build_ordered_bvh(numPrims, pb.data(), seq.data(), tree[0], tree[1])) return e;
        if (tree[0].size() != tree[1].size()) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: forward and mirrored trees differ in size");
        for (int m = 0; m < 2; m++) {
            const std::vector<BvhNode>& t = tree[m];
            size_t met = 0;                         // triangles met so far, in walking order
            std::vector<int> ends;                  // open subtrees (their end indices)
            int depth = 0;
            for (size_t i = 0; i < t.size(); i++) {
                while (!ends.empty() && ends.back() <= (int)i) ends.pop_back();
                if (t[i].next <= (int)i || t[i].next > (int)t.size() || (!ends.empty() && t[i].next > ends.back())) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: miss link is not the end of a nested subtree");
                if (t[i].primId >= 0) {
                    if (t[i].next != (int)i + 1) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: leaf with a subtree");
                    const int start = t[i].primId >> 3, count = t[i].primId & 7;
                    if (count < 1 || count > kMaxLeaf) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: leaf size");
                    for (int j = 0; j < count; j++) {
                        const int at = m ? start - j : start + j;           // index into the even order's sequence
                        const size_t expect = m ? np - 1 - met : met;
                        if (at < 0 || (size_t)at != expect) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: the leaves do not list the triangles in the reference's order");
                        const float* b = &pb[(size_t)seq[(size_t)at] * 6];
                        // the leaf's box and every open ancestor's contain the reference leaf box
                        const BvhNode& n = t[i];
                        if (!(n.bminx <= b[0] && n.bminy <= b[1] && n.bminz <= b[2] && n.bmaxx >= b[3] && n.bmaxy >= b[4] && n.bmaxz >= b[5])) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: a leaf box does not contain its triangle's reference box");
                        met++;
                    }
                }
                else {
                    if ((size_t)i + 1 >= t.size()) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: inner node without children");
                    const BvhNode& n = t[i];
                    const BvhNode& c1 = t[i + 1];
                    const BvhNode& c2 = t[(size_t)c1.next < t.size() ? (size_t)c1.next : i + 1];
                    for (const BvhNode* c : { &c1, &c2 })
                        if (!(n.bminx <= c->bminx && n.bminy <= c->bminy && n.bminz <= c->bminz && n.bmaxx >= c->bmaxx && n.bmaxy >= c->bmaxy && n.bmaxz >= c->bmaxz)) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: a box does not contain its child's");
                    if (c2.next != n.next) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: the second child does not end where its parent ends");
                }
                ends.push_back(t[i].next);
                depth = std::max(depth, (int)ends.size());
            }
            if (met != np) return rs_fail(RS_ERR_INTERNAL, "rs_ordered_bvh_host_check: not every triangle is in a leaf");
            if (maxDepth && m == 0) maxDepth[a] = depth;
        }
        if (nodeCounts) nodeCounts[a] = (int)tree[0].size();
    }
    return 0;
}
