// rs_exact.h -- correctly rounded FP32 division, reciprocal and square root without the compiler's scaled expansions, for operands in
// a guarded range; outside it the plain operator runs, so every function here returns the IEEE result for EVERY input -- the parity
// contract of rs_math.h is untouched, only the instruction count changes.
//
// The compiler's x / d is v_div_scale x 2, v_rcp, five fused steps, v_div_fmas, v_div_fixup (10 instructions, 30 for the three
// quotients of a vector by one scalar: the scaling depends on numerator AND denominator, nothing is shared); its sqrtf is 16.
// With all operands in [2^-60, 2^60) no intermediate of the forms below leaves the normal range, and every operation in them (product,
// fused multiply-add, the hardware's estimates as functions of the significand) commutes with scaling by powers of two, so a statement
// about all significands at one exponent is a statement about the whole range.  The statements are established by EXHAUSTIVE comparison
// with the compiler's operators on gfx950 (rs_debug_exact_ops_mismatches, exact_checks.hip), not by an error analysis of v_rcp_f32 /
// v_sqrt_f32, whose documented accuracy is 1 ulp:
//   * reciprocal: y0 = v_rcp_f32(d), e = fma(-d, y0, 1), y = fma(e, y0, y0) is RN(1 / d) -- Markstein's refinement (1990; Muller et al.,
//     Handbook of Floating-Point Arithmetic, 5.3), whose one theoretical exception (a significand of all ones) does not occur with this
//     hardware's estimate: all 1 006 632 960 floats of the range agree (tests/test_gpu_exact_ops.py);
//   * quotient: with that y: q0 = x * y, rem = fma(-q0, d, x) (exact), q = fma(rem, y, q0) is RN(x / d) -- ALL 2^46 pairs of significands
//     agree (tools/verify_exact_division.py, half a minute of one MI355X; profiles/r03_exact_division_all_pairs.log), the test suite
//     runs a slice of it plus pairs at the edges of the exponent range;
//   * square root: s = v_sqrt_f32(x), then one ulp down or up by the sign of the two exact residuals -- the compiler's own refinement
//     without its range scaling and class test; every float of the range agrees.
// The guards are wave-uniform (__all): one scalar branch, no exec-mask region; a wave with one operand outside the range takes the
// compiler's operator for all its lanes.  Zero, negative numbers, infinity and NaN are outside the range by construction of the test.
#ifndef RS_MATH_H_BODY
#include "rs_math.h"        // which includes this file once its vector type and operators are defined (normalize / length use the forms below)
#elif !defined(RS_EXACT_H_BODY)
#define RS_EXACT_H_BODY

namespace rs {

// -DRS_EXACT_PLAIN: every function below is the compiler's operator (A/B measurements: tools/build_variant.sh plain "-DRS_EXACT_PLAIN").
// The short forms are correct by exhaustive test of gfx950's v_rcp_f32 / v_sqrt_f32, not by analysis: a device pass for any other target
// takes the compiler's operators until tests/test_gpu_exact_ops.py has passed there (the host pass of a .hip file never runs them).
#if defined(RS_EXACT_PLAIN) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__))
#define RS_EXACT_GUARD(cond) ((void)(cond), false)
#else
#define RS_EXACT_GUARD(cond) __all(cond)
#endif

constexpr unsigned kExactLo = 0x21800000u;       // 2^-60
constexpr unsigned kExactHi = 0x5D800000u;       // 2^60

// positive, normal, inside [2^-60, 2^60): one unsigned comparison on the bit pattern (negative numbers, NaN, infinity and zero fail it)
__device__ __forceinline__ bool exact_range(unsigned bits) { return bits - kExactLo < kExactHi - kExactLo; }

__device__ __forceinline__ float rcp_refined(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_by_rcp(float x, float d, float y) {
    const float q0 = x * y;
    const float rem = __builtin_fmaf(-q0, d, x);
    return __builtin_fmaf(rem, y, q0);
}

// the same range for |x|: the forms are sign-symmetric (round-to-nearest is, and so is the hardware's estimate: the reciprocal check runs
// over the negative floats of the range as well)
__device__ __forceinline__ bool exact_range_abs(unsigned bits) { return (bits & 0x7FFFFFFFu) - kExactLo < kExactHi - kExactLo; }

// 1.f / d
__device__ __forceinline__ float rcp_exact(float d) {
    const unsigned b = __float_as_uint(d);
    if (RS_EXACT_GUARD(exact_range(b))) return rcp_refined(d);          // wave-uniform: one scalar branch, no exec mask
    return 1.f / d;
}

// x / d
__device__ __forceinline__ float div_exact(float x, float d, bool unused = false) {
    const unsigned bx = __float_as_uint(x), bd = __float_as_uint(d);
    if (RS_EXACT_GUARD(unused || (exact_range(bx) && exact_range(bd)))) return div_by_rcp(x, d, rcp_refined(d));
    return x / d;
}

// g / d, three quotients by one denominator (f3 operator/(f3, float)).  unused: the caller discards this lane's quotients whatever they
// are (a candidate whose pdf is <= 0), so the lane does not keep its wave from the short form.
__device__ __forceinline__ f3 div3_exact(f3 g, float d, bool unused = false) {
    const unsigned bx = __float_as_uint(g.x), by = __float_as_uint(g.y), bz = __float_as_uint(g.z), bd = __float_as_uint(d);
    const unsigned lo = min(min(bx, by), min(bz, bd)), hi = max(max(bx, by), max(bz, bd));
    if (RS_EXACT_GUARD(unused || (lo >= kExactLo && hi < kExactHi))) {
        const float y = rcp_refined(d);
        return mk3(div_by_rcp(g.x, d, y), div_by_rcp(g.y, d, y), div_by_rcp(g.z, d, y));
    }
    return g / d;
}

// the same for operands of either sign: 1 / d per component of a direction, g / d of a difference vector
__device__ __forceinline__ f3 rcp3_exact_signed(f3 d) {
    if (RS_EXACT_GUARD(exact_range_abs(__float_as_uint(d.x)) && exact_range_abs(__float_as_uint(d.y)) && exact_range_abs(__float_as_uint(d.z))))
        return mk3(rcp_refined(d.x), rcp_refined(d.y), rcp_refined(d.z));
    return mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
}
__device__ __forceinline__ f3 div3_exact_signed(f3 g, float d) {
    if (RS_EXACT_GUARD(exact_range_abs(__float_as_uint(g.x)) && exact_range_abs(__float_as_uint(g.y)) && exact_range_abs(__float_as_uint(g.z)) &&
                       exact_range_abs(__float_as_uint(d)))) {
        const float y = rcp_refined(d);
        return mk3(div_by_rcp(g.x, d, y), div_by_rcp(g.y, d, y), div_by_rcp(g.z, d, y));
    }
    return g / d;
}

__device__ __forceinline__ float sqrt_refined(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rDown = __builtin_fmaf(-down, s, x), rUp = __builtin_fmaf(-up, s, x);
    float root = rDown <= 0.f ? down : s;
    root = rUp > 0.f ? up : root;
    return root;
}
// sqrtf(x)
__device__ __forceinline__ float sqrt_exact(float x) {
    if (RS_EXACT_GUARD(exact_range(__float_as_uint(x)))) return sqrt_refined(x);
    return sqrtf(x);
}

}  // namespace rs

#endif  // RS_MATH_H_BODY / RS_EXACT_H_BODY
