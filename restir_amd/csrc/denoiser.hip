// denoiser.hip -- src/denoiser.cu: the EAW a-trous path and SVGF (SpatioTemporalFilter, second half of this file):
//   waveletFilter (colour)         src/denoiser.cu:64-134    (one level)
//   EAWaveletFilter::filter        src/denoiser.cu:427-437
//   LeveledEAWFilter               src/denoiser.cu:453-477   (5 levels, sigma 64 / .2 / 1, ping-pong)
//   modulate / add                 src/denoiser.cu:218-248,405-425
//
// The reference reconstructs the world position of every tap with cam.getPosition(qx,qy,depth) --
// a normalize per tap, 25 taps, 5 levels.  getPosition is a pure function of the pixel, so it is
// evaluated once per pixel into a position plane at the start of a filter call (same expression,
// same bits) and the levels gather from it.
#include <cmath>

#include "rs_internal.h"

using namespace rs;

namespace {

__constant__ float kGaussian5x5[5][5] = {        // src/denoiser.cu:18-24
    { .0030f, .0133f, .0219f, .0133f, .0030f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0219f, .0983f, .1621f, .0983f, .0219f },
    { .0133f, .0596f, .0983f, .0596f, .0133f },
    { .0030f, .0133f, .0219f, .0133f, .0030f }
};

__global__ void __launch_bounds__(256) k_positions(CamParams cam, const float* __restrict__ depth, const int* __restrict__ primId,
                                                   float* __restrict__ pos, int first, int last) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    const int x = i % cam.width, y = i / cam.width;
    f3 p = splat(0.f);
    if (primId[i] > kNullPrim) p = camera_get_position(cam, x, y, depth[i]);
    st3(pos + (size_t)i * 3, p);
}

// ---- the weight of one tap, shared by every form of the colour filter (so that all forms give the same bits) ----------------------
// x / sigma.  POW2: sigma is a power of two, x * (1 / sigma) is the quotient exactly.  Otherwise Markstein's form of the division
// with the correctly rounded reciprocal r = RN(1 / sigma): q0 = x * r is within an ulp of the quotient, the residual
// rem = x - q0 * sigma is exact in one fused operation, and q0 + rem * r rounds to the correctly rounded quotient (Markstein 1990,
// theorem 8.3 in Muller et al.) wherever no intermediate leaves the normal range -- any x in [2^-100, 2^100] for sigmas between
// 2^-20 and 2^20; beyond that range it is still within an ulp.  3 instructions for the 10 of the scaled IEEE expansion
// (v_div_scale x 2, v_rcp, 4 fused steps, v_div_fmas, v_div_fixup), once per tap.  tests: test_eaw_division_by_sigma_is_exact.
template <bool POW2>
__device__ __forceinline__ float div_sigma(float x, float sigma, float rsigma) {
    if (POW2) return x * rsigma;
    const float q0 = x * rsigma;
    const float rem = __builtin_fmaf(-q0, sigma, x);
    return __builtin_fmaf(rem, rsigma, q0);
}
// exp(-e) for e >= 0 (or NaN, which passes through).  The reference's three factors min(1, exp(-a)) * min(1, exp(-b)) * min(1, exp(-c))
// share one exponential (the min() never acts on arguments <= 0): 1e-7-relative rounding differences in w.  The exponential itself is
// the hardware's 2^t on t = e * -log2(e): t carries a relative rounding error of 2^-24, i.e. an absolute one of |t| * 2^-24, so the
// weight is off by < 7e-7 relative down to weights of 1.5e-5 and by < 5e-6 relative for every weight above 1e-38 -- a taps' weight
// enters the result relative to the centre tap's .1621.  Stated tolerance of the filter against the oracle (glibc expf): rtol 1e-5.
// 2 instructions for the 12 of the library expf (range reduction, v_exp_f32, ldexp, two range tests).
__device__ __forceinline__ float exp_neg(float e) { return __builtin_amdgcn_exp2f(e * -1.44269504088896340736f); }

// MUL == kFusedTaps: the fused form of the tap (rs_eaw_set_fused).  The three sigma arguments then hold -log2(e) / sigma (rounded once,
// from double), the exponent is one chain of fused multiply-adds over the three squared distances, themselves fused dot products, and
// the caller accumulates colour * w with fused operations as well: 36 vector instructions per tap for 50.  Every term is a sum of
// non-negative products, so there is no cancellation: the exponent t differs from the separately rounded form by a few 2^-24 * |t|, the
// weight by < 2e-6 relative down to weights of 1.5e-5 -- inside the filter's stated rtol 1e-5 against the oracle, like the exponential.
constexpr int kFusedTaps = 8;
__device__ __forceinline__ float dot_fused(f3 a) { return __builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x)); }
template <int MUL>
__device__ __forceinline__ float tap_weight(f3 dc, f3 dn, f3 dp, float sigLumin, float rLumin, float sigNormal, float rNormal, float sigDepth, float rDepth) {
    if (MUL == kFusedTaps)
        return __builtin_amdgcn_exp2f(__builtin_fmaf(dot_fused(dp), sigDepth, __builtin_fmaf(dot_fused(dn), sigNormal, dot_fused(dc) * sigLumin)));
    const float eC = div_sigma<(MUL & 1) != 0>(dot(dc, dc), sigLumin, rLumin);
    const float eN = div_sigma<(MUL & 2) != 0>(dot(dn, dn), sigNormal, rNormal);
    const float eP = div_sigma<(MUL & 4) != 0>(dot(dp, dp), sigDepth, rDepth);
    return exp_neg(eC + eN + eP);
}

// MUL: bit 0 / 1 / 2 set = sigLumin / sigNormal / sigDepth is a power of two (the reference's defaults are 64, 0.2 and 1).
template <int MUL>
__global__ void __launch_bounds__(256) k_wavelet(float* __restrict__ colorOut, const float* __restrict__ colorIn,
                                                 const int* __restrict__ primId, const float* __restrict__ normal,
                                                 const float* __restrict__ pos, int W, int H,
                                                 float sigDepth, float sigNormal, float sigLumin, int level, int y0, int y1) {
    const float rLumin = 1.f / sigLumin, rNormal = 1.f / sigNormal, rDepth = 1.f / sigDepth;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= y1) return;
    const int step = 1 << level;
    const int idxP = y * W + x;
    const int idP = primId[idxP];
    const f3 colorP = ld3(colorIn + (size_t)idxP * 3);
    if (idP <= kNullPrim) { st3(colorOut + (size_t)idxP * 3, colorP); return; }
    const f3 normP = ld3(normal + (size_t)idxP * 3);
    const f3 posP = ld3(pos + (size_t)idxP * 3);

    f3 sum = splat(0.f);
    float sumW = 0.f;
#pragma unroll
    for (int i = -2; i <= 2; i++) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const int qx = x + j * step, qy = y + i * step;
            if (qx >= W || qy >= H || qx < 0 || qy < 0) continue;
            const int idxQ = qy * W + qx;
            if (primId[idxQ] != idP) continue;
            const f3 normQ = ld3(normal + (size_t)idxQ * 3);
            const f3 colorQ = ld3(colorIn + (size_t)idxQ * 3);
            const f3 posQ = ld3(pos + (size_t)idxQ * 3);
            const f3 dc = colorP - colorQ, dn = normP - normQ, dp = posP - posQ;
            const float w = tap_weight<MUL>(dc, dn, dp, sigLumin, rLumin, sigNormal, rNormal, sigDepth, rDepth) * kGaussian5x5[i + 2][j + 2];
            if (MUL == kFusedTaps) sum = mk3(__builtin_fmaf(colorQ.x, w, sum.x), __builtin_fmaf(colorQ.y, w, sum.y), __builtin_fmaf(colorQ.z, w, sum.z));
            else sum = sum + colorQ * w;
            sumW += w;
        }
    }
    st3(colorOut + (size_t)idxP * 3, sumW == 0.f ? colorP : sum / sumW);
}

// The same level from an LDS tile, for every step up to 16.  A level of step s is s independent filters, one per row phase
// (y mod s): the rows y, y + s, y + 2s ... form an image in which the taps are the vertical neighbours +-1, +-2.  A block takes
// kTileW x kTileH pixels of ONE phase -- 64 consecutive pixels of 8 rows that lie s apart -- and stages them with 2 rows of that
// phase above and below and 2s pixels to the left and right: (64 + 4s) x 12 records of 40 B (id, normal, colour, position as two
// 16-byte and one 8-byte record), read from global memory as row segments, i.e. coalesced at every step.  (The round-2 tile held
// (32 + 4s) x (8 + 4s) records, 4.5 per pixel at step 4 and too many for steps 8 and 16, whose 25 taps x 10 gathers per pixel
// through L1 kept the texture addresser 87 % busy and fetched 11.5 x the algorithmic bytes over the fabric: the two levels were
// half of the filter's time, profiles/r03_eaw_before.txt.)  Staged records per pixel: 1.6 (s = 1) ... 3 (s = 16).
// POS0 (level 0 of a full-frame filter): the positions are computed while staging -- cam.getPosition of the staged pixel, the same
// expression as k_positions -- and the block writes the position plane for its own pixels; the later levels read the plane.
// Same arithmetic in the same order as k_wavelet: same bits.
constexpr int kTileW = 64, kTileH = 8, kTileThreads = kTileW * kTileH;
template <int STEP, int MUL, bool POS0>
__global__ void __launch_bounds__(kTileThreads) k_wavelet_tiled(float* __restrict__ colorOut, const float* __restrict__ colorIn,
                                                               const int* __restrict__ primId, const float* __restrict__ normal,
                                                               float* __restrict__ pos, const float* __restrict__ depth, CamParams cam, int W, int H,
                                                               float sigDepth, float sigNormal, float sigLumin, int y0, int y1) {
    constexpr int kHaloX = 2 * STEP, kRW = kTileW + 2 * kHaloX, kRH = kTileH + 4, kRN = kRW * kRH;
    RS_SETPRIO(RS_PRIO_EAW);
    __shared__ float4 sColId[kRN];          // colour xyz, id bits
    __shared__ float4 sNormPx[kRN];         // normal xyz, position x
    __shared__ float2 sPyz[kRN];            // position y, z
    const int phase = blockIdx.y % STEP, group = blockIdx.y / STEP;
    const int rowBase = y0 + group * (kTileH * STEP) + phase;            // the row of tile row 0; tile row r is row rowBase + r * STEP
    const int ox = blockIdx.x * kTileW - kHaloX;
    for (int e = threadIdx.x; e < kRN; e += kTileThreads) {
        const int lx = e % kRW, lr = e / kRW;
        const int gx = ox + lx, gy = rowBase + (lr - 2) * STEP;
        float4 a = make_float4(0.f, 0.f, 0.f, __int_as_float(-3)), b = make_float4(0.f, 0.f, 0.f, 0.f);      // id -3 matches no pixel
        float2 c = make_float2(0.f, 0.f);
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const size_t q = (size_t)gy * W + gx;
            const int id = primId[q];
            const f3 col = ld3(colorIn + q * 3), n = ld3(normal + q * 3);
            f3 p;
            if (POS0) {
                p = splat(0.f);
                if (id > kNullPrim) p = camera_get_position(cam, gx, gy, depth[q]);
                if (lx >= kHaloX && lx < kHaloX + kTileW && lr >= 2 && lr < 2 + kTileH && gy < y1) st3(pos + q * 3, p);
            }
            else p = ld3(pos + q * 3);
            a = make_float4(col.x, col.y, col.z, __int_as_float(id));
            b = make_float4(n.x, n.y, n.z, p.x);
            c = make_float2(p.y, p.z);
        }
        sColId[e] = a; sNormPx[e] = b; sPyz[e] = c;
    }
    __syncthreads();
    const float rLumin = 1.f / sigLumin, rNormal = 1.f / sigNormal, rDepth = 1.f / sigDepth;
    const int tx = threadIdx.x % kTileW, ty = threadIdx.x / kTileW;
    const int x = blockIdx.x * kTileW + tx, y = rowBase + ty * STEP;
    if (x >= W || y >= y1) return;
    const int idxP = y * W + x, lp = (ty + 2) * kRW + tx + kHaloX;
    const float4 pa = sColId[lp], pb = sNormPx[lp];
    const float2 pc = sPyz[lp];
    const int idP = __float_as_int(pa.w);
    const f3 colorP = mk3(pa.x, pa.y, pa.z);
    if (idP <= kNullPrim) { st3(colorOut + (size_t)idxP * 3, colorP); return; }
    const f3 normP = mk3(pb.x, pb.y, pb.z), posP = mk3(pb.w, pc.x, pc.y);
    f3 sum = splat(0.f);
    float sumW = 0.f;
#pragma unroll
    for (int i = -2; i <= 2; i++) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const int lq = lp + i * kRW + j * STEP;
            const float4 qa = sColId[lq];
            // (adding +0 for a rejected tap instead of skipping it -- same bits, no exec-mask regions -- measured slower, 337 -> 378 us
            // per call: the filter is bound by VALU issue, and whole waves skip foreign-id and out-of-image taps)
            if (__float_as_int(qa.w) != idP) continue;         // also a tap outside the image (id -3)
            const float4 qb = sNormPx[lq];
            const float2 qc = sPyz[lq];
            const f3 colorQ = mk3(qa.x, qa.y, qa.z);
            const f3 dc = colorP - colorQ, dn = normP - mk3(qb.x, qb.y, qb.z), dp = posP - mk3(qb.w, qc.x, qc.y);
            const float w = tap_weight<MUL>(dc, dn, dp, sigLumin, rLumin, sigNormal, rNormal, sigDepth, rDepth) * kGaussian5x5[i + 2][j + 2];
            if (MUL == kFusedTaps) sum = mk3(__builtin_fmaf(colorQ.x, w, sum.x), __builtin_fmaf(colorQ.y, w, sum.y), __builtin_fmaf(colorQ.z, w, sum.z));
            else sum = sum + colorQ * w;
            sumW += w;
        }
    }
    st3(colorOut + (size_t)idxP * 3, sumW == 0.f ? colorP : sum / sumW);
}

// test hook: div_sigma<false> against the IEEE division on every float in [2^-100, 2^100]
__global__ void k_div_sigma_check(float sigma, unsigned long long* mismatches) {
    const float rs = 1.f / sigma;
    const unsigned lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k <= (unsigned long long)(hi - lo); k += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo + (unsigned)k);
        if (__float_as_int(div_sigma<false>(x, sigma, rs)) != __float_as_int(x / sigma)) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

__global__ void __launch_bounds__(256) k_modulate(float* __restrict__ image, const float* __restrict__ albedo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 c = ld3(image + (size_t)i * 3);
    c = c / 1.f;                                                   // Math::LDRToHDR (mathUtil.h:40-43)
    c = c / ((splat(1.f) - c) + 1e-4f);
    st3(image + (size_t)i * 3, c * vmax(ld3(albedo + (size_t)i * 3), splat(0.f)));
}
__global__ void __launch_bounds__(256) k_add(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---- SVGF: SpatioTemporalFilter (src/denoiser.cu:136-216,250-371,479-568) ---------------------------------
__constant__ float kGaussian3x3[3][3] = {        // src/denoiser.cu:11-15
    { .075f, .124f, .075f },
    { .124f, .204f, .124f },
    { .075f, .124f, .075f }
};

// temporalAccumulate (:250-305), Alpha = .2.  The reference reads the history at lastIdx before it looks at
// `diff` (also for lastIdx = -1); the values are only used when !diff, so they are only loaded then.
__global__ void __launch_bounds__(256) k_svgf_temporal(float* __restrict__ colorOut, const float* __restrict__ colorAccIn,
                                                       float* __restrict__ momentOut, const float* __restrict__ momentAccIn,
                                                       const float* __restrict__ colorIn, GBufView g, int first, int firstIdx, int lastIdx1) {
    const int idx = firstIdx + blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= lastIdx1) return;
    const int primId = g.primId[idx];
    const int lastIdx = g.motion[idx];
    bool diff = first != 0;
    if (lastIdx < 0) diff = true;
    else if (primId <= kNullPrim) diff = true;
    else if (g.lastPrimId[lastIdx] != primId) diff = true;
    else {
        const f3 norm = ld3(g.normal + (size_t)idx * 3), lastNorm = ld3(g.lastNormal + (size_t)lastIdx * 3);
        if (gabs(dot(norm, lastNorm)) < .1f) diff = true;
    }
    const f3 color = ld3(colorIn + (size_t)idx * 3);
    const float lum = luminance(color);
    f3 accColor, accMoment;
    if (diff) {
        accColor = color;
        accMoment = mk3(lum, lum * lum, 0.f);
    }
    else {
        const f3 lastColor = ld3(colorAccIn + (size_t)lastIdx * 3), lastMoment = ld3(momentAccIn + (size_t)lastIdx * 3);
        accColor = mix(lastColor, color, .2f);
        accMoment = mk3(mixf(lastMoment.x, lum, .2f), mixf(lastMoment.y, lum * lum, .2f), lastMoment.z + 1.f);
    }
    st3(colorOut + (size_t)idx * 3, accColor);
    st3(momentOut + (size_t)idx * 3, accMoment);
}

// estimateVariance (:307-343): temporal variance after 4 accumulated frames, else the 3x3 spatial estimate
__global__ void __launch_bounds__(256) k_svgf_variance(float* __restrict__ variance, const float* __restrict__ moment, int W, int H, int y0, int y1) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= y1) return;
    const int idx = y * W + x;
    const f3 m = ld3(moment + (size_t)idx * 3);
    if (m.z > 3.5f) { variance[idx] = m.y - m.x * m.x; return; }
    float sx = 0.f, sy = 0.f;
    int n = 0;
    for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++) {
            const int qx = x + j, qy = y + i;
            if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
            const float* q = moment + (size_t)(qy * W + qx) * 3;
            sx += q[0]; sy += q[1];
            n++;
        }
    sx /= (float)n; sy /= (float)n;
    variance[idx] = sy - sx * sx;
}

// filterVariance (:345-371); the reference walks qx with the OUTER loop variable (:358-359), kept for the summation order
__global__ void __launch_bounds__(256) k_svgf_filter_variance(float* __restrict__ out, const float* __restrict__ in, int W, int H, int y0, int y1) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= y1) return;
    float sum = 0.f, sumW = 0.f;
#pragma unroll
    for (int i = -1; i <= 1; i++) {
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            const int qx = x + i, qy = y + j;
            if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
            const float w = kGaussian3x3[i + 1][j + 1];
            sum += in[qy * W + qx] * w;
            sumW += w;
        }
    }
    out[y * W + x] = sum / sumW;
}

// waveletFilter, SVGF form (:139-216): colour and variance filtered together, luminance weight scaled by the
// 3x3-filtered variance.  Positions come from the per-call plane (see the header comment).
// NSQ >= 0: sigNormal is 2^NSQ (the reference's default is 128 = 2^7), and pow(c, sigNormal) is NSQ squarings instead of powf's
// ~100 VALU instructions -- a third of the tap.  Every squaring rounds once, so the power is within 2^(NSQ-1) ulp of the exact
// one (3.8e-6 relative for 128; CUDA's powf is specified to 4 ulp), well inside the filter's stated tolerance (tests: rtol 3e-5).
// NSQ < 0: powf.  DMUL: sigDepth is a power of two, x / sigDepth == x * (1 / sigDepth) exactly (default 1).
// The weight of one SVGF tap (src/denoiser.cu:176-189), shared by both forms of the kernel: wColor * wNorm * wPos with
//   wPos   = exp(-|dp|^2 / sigDepth) + 1e-4,  wNorm = max(nP . nQ, 0)^sigNormal + 1e-4,
//   wColor = exp(-|lumP - lumQ| / (sigLumin * sqrt(max(varFiltered[Q], 0)) + 1e-4)) + 1e-4.
// The denominator of wColor depends on the tap's pixel only: `denomQ` and its correctly rounded reciprocal are evaluated once per
// pixel where the kernel stages its tile (the same expressions, so the same values as evaluating them per tap), and the division
// takes Markstein's form (div_sigma).  Exponentials as in the EAW filter (exp_neg).  Stated tolerance against the oracle: rtol 3e-5.
// FUSED (rs_svgf_set_fused, the reference's default sigmas only): as for the EAW taps (kFusedTaps) -- |dp|^2 as a fused dot product, each
// exponent one multiplication by a coefficient that holds -log2(e): `rDepth` is -log2(e) / sigDepth and `rdenomQ` is -log2(e) / denomQ, the
// latter staged per pixel like the luminances; the caller accumulates with fused operations.  What stays separately rounded, because
// the weight amplifies its last bit: the luminances (their DIFFERENCE is divided by a denominator that can be 1e-4: fusing them put the
// result 3e-5 from the oracle, measured) and the normals' dot product (raised to the 128th power).
__device__ __forceinline__ float lum_fused(f3 c) { return luminance(c); }
__device__ __forceinline__ float svgf_color_coefficient(float denom) { return -1.44269504088896340736f / denom; }
template <int NSQ, bool DMUL, bool FUSED = false>
__device__ __forceinline__ float svgf_tap_weight(f3 dp, f3 normP, f3 normQ, float lumP, float lumQ, float denomQ, float rdenomQ,
                                                 float sigDepth, float rDepth, float sigNormal) {
    if (FUSED) {
        const float wPos = __builtin_amdgcn_exp2f(dot_fused(dp) * rDepth) + 1e-4f;
        float pn = sat_dot(normP, normQ);
#pragma unroll
        for (int k = 0; k < NSQ; k++) pn = pn * pn;
        const float wColor = __builtin_amdgcn_exp2f(gabs(lumP - lumQ) * rdenomQ) + 1e-4f;
        return wColor * (pn + 1e-4f) * wPos;
    }
    const float wPos = exp_neg(div_sigma<DMUL>(dot(dp, dp), sigDepth, rDepth)) + 1e-4f;
    float pn = sat_dot(normP, normQ);
    if (NSQ >= 0) {
#pragma unroll
        for (int k = 0; k < NSQ; k++) pn = pn * pn;
    }
    else pn = powf(pn, sigNormal);
    const float wNorm = pn + 1e-4f;
    const float wColor = exp_neg(div_sigma<false>(gabs(lumP - lumQ), denomQ, rdenomQ)) + 1e-4f;
    return wColor * wNorm * wPos;
}
__device__ __forceinline__ float svgf_color_denominator(float varFiltered, float sigLumin) { return sigLumin * sqrtf(gmax(varFiltered, 0.f)) + 1e-4f; }

template <int NSQ, bool DMUL, bool FUSED = false>
__global__ void __launch_bounds__(256) k_svgf_wavelet(float* __restrict__ colorOut, const float* __restrict__ colorIn,
                                                      float* __restrict__ varOut, const float* __restrict__ varIn,
                                                      const float* __restrict__ varFiltered,
                                                      const int* __restrict__ primId, const float* __restrict__ normal,
                                                      const float* __restrict__ pos, int W, int H,
                                                      float sigDepth, float sigNormal, float sigLumin, int level, int y0, int y1) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= y1) return;
    const float rDepth = FUSED ? (float)(-1.44269504088896340736 / (double)sigDepth) : 1.f / sigDepth;
    const int step = 1 << level;
    const int idxP = y * W + x;
    const int idP = primId[idxP];
    const f3 colorP = ld3(colorIn + (size_t)idxP * 3);
    if (idP <= kNullPrim) { st3(colorOut + (size_t)idxP * 3, colorP); varOut[idxP] = varIn[idxP]; return; }
    const f3 normP = ld3(normal + (size_t)idxP * 3);
    const f3 posP = ld3(pos + (size_t)idxP * 3);
    const float lumP = FUSED ? lum_fused(colorP) : luminance(colorP);

    f3 sumColor = splat(0.f);
    float sumVar = 0.f, sumW = 0.f, sumW2 = 0.f;
#pragma unroll
    for (int i = -2; i <= 2; i++) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const int qx = x + j * step, qy = y + i * step;
            if (qx >= W || qy >= H || qx < 0 || qy < 0) continue;
            const int idxQ = qy * W + qx;
            if (primId[idxQ] != idP) continue;
            const f3 normQ = ld3(normal + (size_t)idxQ * 3);
            const f3 colorQ = ld3(colorIn + (size_t)idxQ * 3);
            const f3 dp = posP - ld3(pos + (size_t)idxQ * 3);
            const float denomQ = svgf_color_denominator(varFiltered[idxQ], sigLumin);
            const float w = svgf_tap_weight<NSQ, DMUL, FUSED>(dp, normP, normQ, lumP, FUSED ? lum_fused(colorQ) : luminance(colorQ), denomQ,
                                                              FUSED ? svgf_color_coefficient(denomQ) : 1.f / denomQ, sigDepth, rDepth, sigNormal) * kGaussian5x5[i + 2][j + 2];
            const float w2 = w * w;
            if (FUSED) {
                sumColor = mk3(__builtin_fmaf(colorQ.x, w, sumColor.x), __builtin_fmaf(colorQ.y, w, sumColor.y), __builtin_fmaf(colorQ.z, w, sumColor.z));
                sumVar = __builtin_fmaf(varIn[idxQ], w2, sumVar);
                sumW += w; sumW2 += w2;
                continue;
            }
            sumColor = sumColor + colorQ * w;
            sumVar += varIn[idxQ] * w2;
            sumW += w;
            sumW2 += w2;
        }
    }
    st3(colorOut + (size_t)idxP * 3, sumW < 1.1920928955078125e-7f ? colorP : sumColor / sumW);
    varOut[idxP] = sumW2 < 1.1920928955078125e-7f ? varIn[idxP] : sumVar / sumW2;
}

// The same level from the EAW filter's row-phase LDS tile (k_wavelet_tiled above): 64 x 8 pixels of one row phase, staged with
// 2 rows of that phase above and below and 2 * STEP pixels to the left and right, 52 B per record -- colour + id, normal + position x,
// {position y, z, variance, the colour weight's denominator} and its reciprocal.  Same arithmetic in the same order as
// k_svgf_wavelet: same bits (test_svgf_tiled_levels_equal_plain_gathers).
template <int STEP, int NSQ, bool DMUL, bool FUSED = false>
__global__ void __launch_bounds__(kTileThreads) k_svgf_wavelet_tiled(float* __restrict__ colorOut, const float* __restrict__ colorIn,
                                                                    float* __restrict__ varOut, const float* __restrict__ varIn,
                                                                    const float* __restrict__ varFiltered,
                                                                    const int* __restrict__ primId, const float* __restrict__ normal,
                                                                    const float* __restrict__ pos, int W, int H,
                                                                    float sigDepth, float sigNormal, float sigLumin, int y0, int y1) {
    constexpr int kHaloX = 2 * STEP, kRW = kTileW + 2 * kHaloX, kRH = kTileH + 4, kRN = kRW * kRH;
    __shared__ float4 sColId[kRN];          // colour xyz, id bits
    __shared__ float4 sNormPx[kRN];         // normal xyz, position x
    __shared__ float4 sMisc[kRN];           // position y, z, variance, denominator of the colour weight
    __shared__ float sRden[kRN];            // 1 / denominator
    const int phase = blockIdx.y % STEP, group = blockIdx.y / STEP;
    const int rowBase = y0 + group * (kTileH * STEP) + phase;
    const int ox = blockIdx.x * kTileW - kHaloX;
    for (int e = threadIdx.x; e < kRN; e += kTileThreads) {
        const int lx = e % kRW, lr = e / kRW;
        const int gx = ox + lx, gy = rowBase + (lr - 2) * STEP;
        float4 a = make_float4(0.f, 0.f, 0.f, __int_as_float(-3)), b = make_float4(0.f, 0.f, 0.f, 0.f), c = b;      // id -3 matches no pixel
        float r = 0.f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const size_t q = (size_t)gy * W + gx;
            const f3 col = ld3(colorIn + q * 3), n = ld3(normal + q * 3), p = ld3(pos + q * 3);
            const float denom = svgf_color_denominator(varFiltered[q], sigLumin);
            a = make_float4(col.x, col.y, col.z, __int_as_float(primId[q]));
            b = make_float4(n.x, n.y, n.z, p.x);
            c = make_float4(p.y, p.z, varIn[q], FUSED ? lum_fused(col) : denom);          // FUSED: the slot carries the pixel's luminance
            r = FUSED ? svgf_color_coefficient(denom) : 1.f / denom;
        }
        sColId[e] = a; sNormPx[e] = b; sMisc[e] = c; sRden[e] = r;
    }
    __syncthreads();
    const float rDepth = FUSED ? (float)(-1.44269504088896340736 / (double)sigDepth) : 1.f / sigDepth;
    const int tx = threadIdx.x % kTileW, ty = threadIdx.x / kTileW;
    const int x = blockIdx.x * kTileW + tx, y = rowBase + ty * STEP;
    if (x >= W || y >= y1) return;
    const int idxP = y * W + x, lp = (ty + 2) * kRW + tx + kHaloX;
    const float4 pa = sColId[lp], pb = sNormPx[lp], pc = sMisc[lp];
    const int idP = __float_as_int(pa.w);
    const f3 colorP = mk3(pa.x, pa.y, pa.z);
    if (idP <= kNullPrim) { st3(colorOut + (size_t)idxP * 3, colorP); varOut[idxP] = pc.z; return; }
    const f3 normP = mk3(pb.x, pb.y, pb.z), posP = mk3(pb.w, pc.x, pc.y);
    const float lumP = FUSED ? pc.w : luminance(colorP);
    f3 sumColor = splat(0.f);
    float sumVar = 0.f, sumW = 0.f, sumW2 = 0.f;
#pragma unroll
    for (int i = -2; i <= 2; i++) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const int lq = lp + i * kRW + j * STEP;
            const float4 qa = sColId[lq];
            if (__float_as_int(qa.w) != idP) continue;         // also a tap outside the image (id -3)
            const float4 qb = sNormPx[lq], qc = sMisc[lq];
            const f3 colorQ = mk3(qa.x, qa.y, qa.z);
            const f3 dp = posP - mk3(qb.w, qc.x, qc.y);
            const float w = svgf_tap_weight<NSQ, DMUL, FUSED>(dp, normP, mk3(qb.x, qb.y, qb.z), lumP, FUSED ? qc.w : luminance(colorQ), qc.w, sRden[lq], sigDepth, rDepth, sigNormal) * kGaussian5x5[i + 2][j + 2];
            const float w2 = w * w;
            if (FUSED) {
                sumColor = mk3(__builtin_fmaf(colorQ.x, w, sumColor.x), __builtin_fmaf(colorQ.y, w, sumColor.y), __builtin_fmaf(colorQ.z, w, sumColor.z));
                sumVar = __builtin_fmaf(qc.z, w2, sumVar);
                sumW += w; sumW2 += w2;
                continue;
            }
            sumColor = sumColor + colorQ * w;
            sumVar += qc.z * w2;
            sumW += w;
            sumW2 += w2;
        }
    }
    st3(colorOut + (size_t)idxP * 3, sumW < 1.1920928955078125e-7f ? colorP : sumColor / sumW);
    varOut[idxP] = sumW2 < 1.1920928955078125e-7f ? pc.z : sumVar / sumW2;
}

// pos0: level 0 computes the positions itself (and writes the plane): only for a full-frame call, whose level 0 visits every pixel
int wavelet_level(const rs_eaw* f, float* out, const float* in, const rs_gbuffer* g, int level, int y0, int y1, const rs_camera* posCam = nullptr) {
    const auto pow2 = [](float v) { int e; return v > 0.f && std::isfinite(v) && std::frexp(v, &e) == 0.5f && 1.f / v > 0.f && std::isfinite(1.f / v) && std::isnormal(1.f / v); };
    const bool fused = f->fused && f->sigLumin > 0.f && f->sigNormal > 0.f && f->sigDepth > 0.f;
    const int mul = fused ? kFusedTaps : (pow2(f->sigLumin) ? 1 : 0) | (pow2(f->sigNormal) ? 2 : 0) | (pow2(f->sigDepth) ? 4 : 0);
    const auto coef = [](float sigma) { return (float)(-1.44269504088896340736 / (double)sigma); };
    const float aDepth = fused ? coef(f->sigDepth) : f->sigDepth, aNormal = fused ? coef(f->sigNormal) : f->sigNormal, aLumin = fused ? coef(f->sigLumin) : f->sigLumin;
    const bool tiled = f->tiled && level <= 4;
    const int c = g->cur();
    if (tiled) {
        const int step = 1 << level;
        const dim3 grid((f->width + kTileW - 1) / kTileW, ((y1 - y0 + kTileH * step - 1) / (kTileH * step)) * step);
        const bool pos0 = posCam != nullptr && level == 0;
        const CamParams cp = pos0 ? rs_make_cam_params(posCam) : CamParams{};
#define RS_TILED_ARGS out, in, g->primId[c], g->normal[c], f->devPos, g->depth[c], cp, f->width, f->height, aDepth, aNormal, aLumin, y0, y1
#define RS_TILED(M) do { \
        if (pos0) hipLaunchKernelGGL((k_wavelet_tiled<1, M, true>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); \
        else if (level == 0) hipLaunchKernelGGL((k_wavelet_tiled<1, M, false>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); \
        else if (level == 1) hipLaunchKernelGGL((k_wavelet_tiled<2, M, false>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); \
        else if (level == 2) hipLaunchKernelGGL((k_wavelet_tiled<4, M, false>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); \
        else if (level == 3) hipLaunchKernelGGL((k_wavelet_tiled<8, M, false>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); \
        else hipLaunchKernelGGL((k_wavelet_tiled<16, M, false>), grid, dim3(kTileThreads), 0, rs_stream(), RS_TILED_ARGS); } while (0)
        switch (mul) {
            case 0: RS_TILED(0); break; case 1: RS_TILED(1); break; case 2: RS_TILED(2); break; case 3: RS_TILED(3); break;
            case 4: RS_TILED(4); break; case 5: RS_TILED(5); break; case 6: RS_TILED(6); break; case 7: RS_TILED(7); break;
            default: RS_TILED(kFusedTaps); break;
        }
#undef RS_TILED
#undef RS_TILED_ARGS
        return rs_after_launch("EAW Filter");
    }
    const dim3 grid((f->width + 31) / 32, (y1 - y0 + 7) / 8);
#define RS_WAVELET(M) hipLaunchKernelGGL(k_wavelet<M>, grid, dim3(256), 0, rs_stream(), out, in, g->primId[c], g->normal[c], f->devPos, f->width, f->height, \
                                         aDepth, aNormal, aLumin, level, y0, y1)
    switch (mul) {
        case 0: RS_WAVELET(0); break; case 1: RS_WAVELET(1); break; case 2: RS_WAVELET(2); break; case 3: RS_WAVELET(3); break;
        case 4: RS_WAVELET(4); break; case 5: RS_WAVELET(5); break; case 6: RS_WAVELET(6); break; case 7: RS_WAVELET(7); break;
        default: RS_WAVELET(kFusedTaps); break;
    }
#undef RS_WAVELET
    return rs_after_launch("EAW Filter");
}

}  // namespace

extern "C" {

// test hook: the filter's division by a sigma that is not a power of two (div_sigma) against the IEEE division, on every float in
// [2^-100, 2^100]; returns the number of differing quotients
int rs_debug_div_sigma_mismatches(float sigma, unsigned long long* mismatches) {
    rs_ctx_scope scope(nullptr);
    if (!mismatches || !(sigma > 0.f)) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_div_sigma_mismatches: bad argument");
    unsigned long long* d = nullptr;
    RS_TRY(rs_dev_alloc(&d, 1));
    RS_HIP(hipMemsetAsync(d, 0, 8, rs_stream()));
    hipLaunchKernelGGL(k_div_sigma_check, dim3(8192), dim3(256), 0, rs_stream(), sigma, d);
    RS_HIP(hipStreamSynchronize(rs_stream()));
    RS_HIP(hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost));
    rs_dev_free(d);
    return 0;
}

int rs_eaw_destroy(rs_eaw* f) {
    RS_SCOPE(f);
    if (!f) return 0;
    rs_dev_free(f->devTempImg); rs_dev_free(f->devPos);
    delete f;
    return 0;
}

int rs_eaw_create(int width, int height, int level, rs_eaw** out) {
    if (!out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_create: bad size");
    *out = nullptr;
    rs_eaw* f = new rs_eaw();
    f->ctx = rs_ctx();
    rs_ctx_scope scope(f->ctx);
    f->width = width; f->height = height; f->level = level;
    int e = rs_dev_alloc(&f->devTempImg, (size_t)width * height * 3);
    if (!e) e = rs_dev_alloc(&f->devPos, (size_t)width * height * 3);
    if (e) { rs_eaw_destroy(f); return e; }
    *out = f;
    return 0;
}

int rs_eaw_set_params(rs_eaw* f, float sigLumin, float sigNormal, float sigDepth, int level) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_set_params: null filter");
    f->sigLumin = sigLumin; f->sigNormal = sigNormal; f->sigDepth = sigDepth; f->level = level;
    return 0;
}
// which form the levels of step 1, 2 and 4 take: 1 (default) the LDS tile, 0 the plain gathers -- same bits, for measurements and tests
int rs_eaw_set_tiled(rs_eaw* f, int tiled) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_set_tiled: null filter");
    f->tiled = tiled != 0;
    return 0;
}
// the arithmetic of a tap: 0 every operation rounded separately, in the reference's order; 1 (DEFAULT) fused multiply-adds for the squared
// distances, the exponent and the accumulation (kFusedTaps above): same taps, same weights to < 2e-6 relative, 0.7 x the instructions
int rs_eaw_set_fused(rs_eaw* f, int fused) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_set_fused: null filter");
    f->fused = fused != 0;
    return 0;
}
int rs_eaw_get_params(const rs_eaw* f, float* sigLumin, float* sigNormal, float* sigDepth, int* level) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_eaw_get_params: null filter");
    if (sigLumin) *sigLumin = f->sigLumin;
    if (sigNormal) *sigNormal = f->sigNormal;
    if (sigDepth) *sigDepth = f->sigDepth;
    if (level) *level = f->level;
    return 0;
}

int rs_eaw_filter(rs_eaw* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam) {
    RS_SCOPE(f);
    RS_TRY(rs_gbuffer_join(g));                         // the render may still be on the auxiliary stream
    if (!f || !devColorOut || !*devColorOut || !devColorIn || !g || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: null argument");
    if (g->width != f->width || g->height != f->height || cam->resolution[0] != f->width || cam->resolution[1] != f->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: size mismatch");
    const int n = f->width * f->height;
    const size_t imageBytes = (size_t)n * 3 * sizeof(float);
    // rs_set_denoise_stream(1), asynchronous launches: the levels go to the denoise stream, ordered after everything enqueued on the
    // library stream so far (phase B of this frame, the render's join above); the library stream goes on with the next frame.  What
    // the levels read and write there is noted with the events that cover it (rs_denoise_mark): the input image up to level 0 -- the
    // next frame's phase B writes it --, the two buffers the levels alternate between and the G-buffer set up to the last level.
    rs_denoise_scope onDenoiseStream(true);
    RS_TRY(onDenoiseStream.err);
    // (no denoise stream now, but there may have been one: the library stream after that stream's last use of these buffers; inside the scope these are no-ops)
    RS_TRY(rs_denoise_order(devColorIn, false)); RS_TRY(rs_denoise_order(*devColorOut)); RS_TRY(rs_denoise_order(f->devTempImg));
    const bool fusedPositions = f->tiled;           // the tiled level 0 computes the positions while it stages its tile
    if (!fusedPositions) {
        hipLaunchKernelGGL(k_positions, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), rs_make_cam_params(cam),
                           g->depth[g->cur()], g->primId[g->cur()], f->devPos, 0, n);
        RS_TRY(rs_after_launch("EAW positions"));
    }
    // LeveledEAWFilter::filter (denoiser.cu:463-477): level 0 into out, then four ping-pongs with the
    // internal buffer; the caller's pointer and the internal one are swapped after each
    RS_TRY(wavelet_level(f, *devColorOut, devColorIn, g, 0, 0, f->height, fusedPositions ? cam : nullptr));
    if (onDenoiseStream.active) RS_TRY(rs_denoise_mark(devColorIn, imageBytes, true));
    for (int level = 1; level <= 4; level++) {
        RS_TRY(wavelet_level(f, f->devTempImg, *devColorOut, g, level, 0, f->height));
        float* t = *devColorOut; *devColorOut = f->devTempImg; f->devTempImg = t;
    }
    if (onDenoiseStream.active) {
        RS_TRY(rs_denoise_mark(*devColorOut, imageBytes, false));
        RS_TRY(rs_denoise_mark(f->devTempImg, imageBytes, false));
        RS_TRY(rs_gbuffer_denoise_mark(g));
    }
    return 0;
}

// Row-strip form of the filter for framebuffer tiling (the levels of rs_eaw_filter one by one, on rows [y0, y1)): a level reads
// its input, the G-buffer ids / normals and the positions up to 2 << level rows outside the strip, so between the levels the
// caller exchanges those rows of the colour buffer with the neighbouring strips (restir_amd/tiling.py).
int rs_eaw_positions_rows(rs_eaw* f, const rs_gbuffer* g, const rs_camera* cam, int y0, int y1) {
    RS_SCOPE(f);
    RS_TRY(rs_gbuffer_join(g));
    if (!f || !g || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW positions: null argument");
    if (g->width != f->width || g->height != f->height || cam->resolution[0] != f->width || cam->resolution[1] != f->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: size mismatch");
    if (y0 < 0) y0 = 0;
    if (y1 > f->height) y1 = f->height;
    if (y1 <= y0) return 0;
    const int first = y0 * f->width, last = y1 * f->width;
    hipLaunchKernelGGL(k_positions, dim3((last - first + 255) / 256), dim3(256), 0, rs_stream(), rs_make_cam_params(cam),
                       g->depth[g->cur()], g->primId[g->cur()], f->devPos, first, last);
    return rs_after_launch("EAW positions");
}

int rs_eaw_level_rows(rs_eaw* f, float* devColorOut, const float* devColorIn, const rs_gbuffer* g, int level, int y0, int y1) {
    RS_SCOPE(f);
    RS_TRY(rs_gbuffer_join(g));
    if (!f || !devColorOut || !devColorIn || !g || level < 0 || level > 30) return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW level: bad argument");
    if (g->width != f->width || g->height != f->height) return rs_fail(RS_ERR_INVALID_ARGUMENT, "EAW filter: size mismatch");
    if (y0 < 0) y0 = 0;
    if (y1 > f->height) y1 = f->height;
    if (y1 <= y0) return 0;
    RS_TRY(rs_denoise_order(devColorIn, false)); RS_TRY(rs_denoise_order(devColorOut));
    return wavelet_level(f, devColorOut, devColorIn, g, level, y0, y1);
}


// ---- SpatioTemporalFilter (src/denoiser.cu:479-568) -----------------------------------------------------------
int rs_svgf_destroy(rs_svgf* f) {
    RS_SCOPE(f);
    if (!f) return 0;
    for (int i = 0; i < 2; i++) { rs_dev_free(f->devAccumColor[i]); rs_dev_free(f->devAccumMoment[i]); }
    rs_dev_free(f->devVariance); rs_dev_free(f->devTempVariance); rs_dev_free(f->devFilteredVariance);
    rs_dev_free(f->devTempColor); rs_dev_free(f->devPos);
    delete f;
    return 0;
}

int rs_svgf_set_params(rs_svgf* f, float sigLumin, float sigNormal, float sigDepth, int level) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_set_params: null filter");
    f->sigLumin = sigLumin; f->sigNormal = sigNormal; f->sigDepth = sigDepth; f->level = level;
    return 0;
}
// which form the a-trous levels take: 1 (default) the LDS tile, 0 the plain gathers -- same bits, for measurements and tests
int rs_svgf_set_tiled(rs_svgf* f, int tiled) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_set_tiled: null filter");
    f->tiled = tiled != 0;
    return 0;
}
// the arithmetic of a tap, as rs_eaw_set_fused; acts with the reference's default sigNormal 128 and a power-of-two sigDepth
int rs_svgf_set_fused(rs_svgf* f, int fused) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_set_fused: null filter");
    f->fused = fused != 0;
    return 0;
}
int rs_svgf_get_params(const rs_svgf* f, float* sigLumin, float* sigNormal, float* sigDepth, int* level) {
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_get_params: null filter");
    if (sigLumin) *sigLumin = f->sigLumin;
    if (sigNormal) *sigNormal = f->sigNormal;
    if (sigDepth) *sigDepth = f->sigDepth;
    if (level) *level = f->level;
    return 0;
}

int rs_svgf_create(int width, int height, int level, rs_svgf** out) {
    if (!out || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_create: bad size");
    *out = nullptr;
    rs_svgf* f = new rs_svgf();
    f->ctx = rs_ctx();
    rs_ctx_scope scope(f->ctx);
    f->width = width; f->height = height; f->level = level;
    const size_t n = (size_t)width * height;
    int e = 0;
    for (int i = 0; i < 2 && !e; i++) {
        e = rs_dev_alloc(&f->devAccumColor[i], n * 3);
        if (!e) e = rs_dev_alloc(&f->devAccumMoment[i], n * 3);
    }
    if (!e) e = rs_dev_alloc(&f->devVariance, n);
    if (!e) e = rs_dev_alloc(&f->devTempVariance, n);
    if (!e) e = rs_dev_alloc(&f->devFilteredVariance, n);
    if (!e) e = rs_dev_alloc(&f->devTempColor, n * 3);
    if (!e) e = rs_dev_alloc(&f->devPos, n * 3);
    if (e) { rs_svgf_destroy(f); return e; }
    *out = f;
    return 0;
}

int rs_svgf_next_frame(rs_svgf* f) {                          // SpatioTemporalFilter::nextFrame (:566-568)
    RS_SCOPE(f);
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_next_frame: null filter");
    f->frameIdx ^= 1;
    return 0;
}

int rs_svgf_get_view(const rs_svgf* f, rs_svgf_view* v) {
    RS_SCOPE(f);
    if (!f || !v) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_svgf_get_view: null argument");
    for (int i = 0; i < 2; i++) { v->devAccumColor[i] = f->devAccumColor[i]; v->devAccumMoment[i] = f->devAccumMoment[i]; }
    v->devVariance = f->devVariance; v->frameIdx = f->frameIdx; v->width = f->width; v->height = f->height;
    return 0;
}

// SpatioTemporalFilter::filter (:532-564).  *devColorOut is the reference's `glm::vec3*& devColorOut`: after level 0
// it is swapped with devAccumColor[frameIdx], so the caller's buffer becomes the filter's history and the caller
// continues with one of the filter's buffers; as in the reference the caller must keep using the pointer it gets back.
// Rows [y0, y1): the whole frame for rs_svgf_filter, a strip for the strip driver (strips.hip rs_strips_svgf_filter), which
// passes `hooks` to exchange border rows with its neighbours at the three places a strip reads beyond its rows:
//   after the temporal accumulation   1 row of devAccumMoment[frameIdx] (the 3x3 variance estimate)
//   before level lv                   2 * step + 1 rows of the level's input colour and of devVariance (the 3x3 variance
//                                     pre-filter is then evaluated on rows [y0 - 2 step, y1 + 2 step) locally)
// (the G-buffer rows and positions up to 32 rows beyond the strip are the caller's business, before the call).
}  // extern "C"
int rs_svgf_filter_rows(rs_svgf* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam, int y0, int y1,
                        const rs_svgf_row_hooks* hooks) {
    const int W = f->width, H = f->height;
    const int fi = f->frameIdx;
    const GBufView gv = gbuf_view(g);
    const auto grid_rows = [&](int a, int b) { return dim3((W + 31) / 32, (b - a + 7) / 8); };
    const auto clampRow = [&](int y) { return y < 0 ? 0 : (y > H ? H : y); };
    const bool strip = hooks != nullptr;
    const int reach = 2 << 4;
    {   // positions of every row a tap can look at
        const int a = clampRow(y0 - (strip ? reach : 0)), b = clampRow(y1 + (strip ? reach : 0));
        hipLaunchKernelGGL(k_positions, dim3(((b - a) * W + 255) / 256), dim3(256), 0, rs_stream(), rs_make_cam_params(cam),
                           g->depth[g->cur()], g->primId[g->cur()], f->devPos, a * W, b * W);
    }
    // temporalAccumulate (:506-519), estimateVariance (:521-527)
    hipLaunchKernelGGL(k_svgf_temporal, dim3(((y1 - y0) * W + 255) / 256), dim3(256), 0, rs_stream(), f->devAccumColor[fi], f->devAccumColor[fi ^ 1],
                       f->devAccumMoment[fi], f->devAccumMoment[fi ^ 1], devColorIn, gv, f->firstTime ? 1 : 0, y0 * W, y1 * W);
    f->firstTime = false;
    if (strip) RS_TRY(hooks->exchange(hooks->ctx, f->devAccumMoment[fi], 3, nullptr, 0, 1));
    hipLaunchKernelGGL(k_svgf_variance, grid_rows(y0, y1), dim3(256), 0, rs_stream(), f->devVariance, f->devAccumMoment[fi], W, H, y0, y1);
    RS_TRY(rs_after_launch("SpatioTemporalFilter::temporalAccumulate"));

    int de = 0;
    const bool depthPow2 = f->sigDepth > 0.f && std::isfinite(f->sigDepth) && std::frexp(f->sigDepth, &de) == 0.5f && std::isnormal(1.f / f->sigDepth);
    int err = 0;
    auto level = [&](float* out, float* in, int lv) {
        const int step = 1 << lv;
        if (strip && !err) err = hooks->exchange(hooks->ctx, in, 3, f->devVariance, 1, 2 * step + 1);
        const int va = clampRow(y0 - (strip ? 2 * step : 0)), vb = clampRow(y1 + (strip ? 2 * step : 0));
        hipLaunchKernelGGL(k_svgf_filter_variance, grid_rows(va, vb), dim3(256), 0, rs_stream(), f->devFilteredVariance, f->devVariance, W, H, va, vb);
#define RS_SVGF_WAVELET(N, D) hipLaunchKernelGGL((k_svgf_wavelet<N, D>), grid_rows(y0, y1), dim3(256), 0, rs_stream(), out, in, f->devTempVariance, f->devVariance, \
                                                f->devFilteredVariance, gv.primId, gv.normal, f->devPos, W, H, f->sigDepth, f->sigNormal, f->sigLumin, lv, y0, y1)
        // the reference's defaults (sigNormal 128, sigDepth 1) from the LDS tile; edited sigmas keep the plain gathers
        if (f->tiled && f->sigNormal == 128.f && depthPow2 && lv <= 4) {
            const dim3 gridT((W + kTileW - 1) / kTileW, ((y1 - y0 + kTileH * step - 1) / (kTileH * step)) * step);
#define RS_SVGF_TILED_ARGS gridT, dim3(kTileThreads), 0, rs_stream(), out, in, f->devTempVariance, f->devVariance, \
                           f->devFilteredVariance, gv.primId, gv.normal, f->devPos, W, H, f->sigDepth, f->sigNormal, f->sigLumin, y0, y1
#define RS_SVGF_TILED(S) do { if (f->fused) hipLaunchKernelGGL((k_svgf_wavelet_tiled<S, 7, true, true>), RS_SVGF_TILED_ARGS); \
                              else hipLaunchKernelGGL((k_svgf_wavelet_tiled<S, 7, true>), RS_SVGF_TILED_ARGS); } while (0)
            if (lv == 0) RS_SVGF_TILED(1); else if (lv == 1) RS_SVGF_TILED(2); else if (lv == 2) RS_SVGF_TILED(4); else if (lv == 3) RS_SVGF_TILED(8); else RS_SVGF_TILED(16);
#undef RS_SVGF_TILED
#undef RS_SVGF_TILED_ARGS
        }
        else if (f->sigNormal == 128.f && depthPow2 && f->fused)
            hipLaunchKernelGGL((k_svgf_wavelet<7, true, true>), grid_rows(y0, y1), dim3(256), 0, rs_stream(), out, in, f->devTempVariance, f->devVariance,
                               f->devFilteredVariance, gv.primId, gv.normal, f->devPos, W, H, f->sigDepth, f->sigNormal, f->sigLumin, lv, y0, y1);
        else if (f->sigNormal == 128.f) { if (depthPow2) RS_SVGF_WAVELET(7, true); else RS_SVGF_WAVELET(7, false); }
        else if (f->sigNormal == 64.f) { if (depthPow2) RS_SVGF_WAVELET(6, true); else RS_SVGF_WAVELET(6, false); }
        else if (f->sigNormal == 32.f) { if (depthPow2) RS_SVGF_WAVELET(5, true); else RS_SVGF_WAVELET(5, false); }
        else { if (depthPow2) RS_SVGF_WAVELET(-1, true); else RS_SVGF_WAVELET(-1, false); }
#undef RS_SVGF_WAVELET
        float* t = f->devTempVariance; f->devTempVariance = f->devVariance; f->devVariance = t;      // std::swap(devTempVariance, devVariance)
    };
    level(*devColorOut, f->devAccumColor[fi], 0);
    { float* t = *devColorOut; *devColorOut = f->devAccumColor[fi]; f->devAccumColor[fi] = t; }       // the filtered colour is the new history
    level(*devColorOut, f->devAccumColor[fi], 1);
    for (int lv = 2; lv <= 4; lv++) {
        level(f->devTempColor, *devColorOut, lv);
        float* t = f->devTempColor; f->devTempColor = *devColorOut; *devColorOut = t;
    }
    if (err) return err;
    return rs_after_launch("SpatioTemporalFilter::filter");
}
extern "C" {

int rs_svgf_filter(rs_svgf* f, float** devColorOut, const float* devColorIn, const rs_gbuffer* g, const rs_camera* cam) {
    RS_SCOPE(f);
    RS_TRY(rs_gbuffer_join(g));                         // the render may still be on the auxiliary stream
    if (!f || !devColorOut || !*devColorOut || !devColorIn || !g || !cam) return rs_fail(RS_ERR_INVALID_ARGUMENT, "SVGF filter: null argument");
    if (g->width != f->width || g->height != f->height || cam->resolution[0] != f->width || cam->resolution[1] != f->height)
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "SVGF filter: size mismatch");
    RS_TRY(rs_denoise_order(devColorIn, false)); RS_TRY(rs_denoise_order(*devColorOut));
    return rs_svgf_filter_rows(f, devColorOut, devColorIn, g, cam, 0, f->height, nullptr);
}

int rs_modulate_albedo(float* devImage, const rs_gbuffer* g) {
    RS_SCOPE(g);
    RS_TRY(rs_gbuffer_join(g));                         // the render may still be on the auxiliary stream
    if (!devImage || !g) return rs_fail(RS_ERR_INVALID_ARGUMENT, "modulateAlbedo: null argument");
    const int n = g->width * g->height;
    RS_TRY(rs_denoise_order(devImage));
    hipLaunchKernelGGL(k_modulate, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devImage, g->albedo[g->latest()], n);
    return rs_after_launch("modulate");
}

int rs_add_image(float* devImage, const float* devIn, int width, int height) {
    rs_ctx_scope scope(nullptr);
    if (!devImage || !devIn || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "addImage: bad argument");
    const int n = width * height * 3;
    RS_TRY(rs_denoise_order(devImage)); RS_TRY(rs_denoise_order(devIn, false));
    hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devImage, devImage, devIn, n);
    return rs_after_launch("addImage");
}

int rs_add_image3(float* devOut, const float* devIn1, const float* devIn2, int width, int height) {
    rs_ctx_scope scope(nullptr);
    if (!devOut || !devIn1 || !devIn2 || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "addImage: bad argument");
    const int n = width * height * 3;
    RS_TRY(rs_denoise_order(devOut)); RS_TRY(rs_denoise_order(devIn1, false)); RS_TRY(rs_denoise_order(devIn2, false));
    hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, rs_stream(), devOut, devIn1, devIn2, n);
    return rs_after_launch("addImage");
}

}  // extern "C"
