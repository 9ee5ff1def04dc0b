// exact_checks.hip -- test hooks for rs_exact.h: the short forms of reciprocal, quotient and square root against the compiler's
// correctly rounded operators (-fhip-fp32-correctly-rounded-divide-sqrt), on every operand the guards of rs_exact.h admit.
#include "rs_internal.h"
#include "rs_exact.h"

using namespace rs;

namespace {

// out[0] results that differ, out[1] operands (pairs) compared, out[2] reserved (0)
__device__ __forceinline__ void tally(unsigned long long* out, unsigned long long bad, unsigned long long n, unsigned long long badExcluded) {
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off); n += __shfl_down(n, off); badExcluded += __shfl_down(badExcluded, off); }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(out, bad);
        atomicAdd(out + 1, n);
        if (badExcluded) atomicAdd(out + 2, badExcluded);
    }
}

// op 0: 1 / d for every float d with |d| in [2^-60, 2^60)
// op 1: sqrt(x) for every float x in [2^-60, 2^60)
__global__ void __launch_bounds__(256) k_check_unary(int op, unsigned long long* out) {
    unsigned long long bad = 0, n = 0, badExcluded = 0;
    const unsigned long long span = kExactHi - kExactLo;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < span; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = kExactLo + (unsigned)k;
        const float v = __uint_as_float(bits);
        if (op == 0) {
            n += 2;
            bad += __float_as_uint(rcp_refined(v)) == __float_as_uint(1.f / v) ? 0 : 1;
            bad += __float_as_uint(rcp_refined(-v)) == __float_as_uint(1.f / -v) ? 0 : 1;
        }
        else { n++; bad += __float_as_uint(sqrt_refined(v)) == __float_as_uint(sqrtf(v)) ? 0 : 1; }
    }
    tally(out, bad, n, badExcluded);
}

// op 2: x / d for d = 1.sd * 2^expD over the significands sd in [firstSig, firstSig + gridDim.x) -- one per block -- and x = 1.sx * 2^expX
// over ALL 2^23 significands sx
// signs: bit 0 = negative numerators, bit 1 = negative denominator
__global__ void __launch_bounds__(256) k_check_division(unsigned firstSig, int expX, int expD, int signs, unsigned long long* out) {
    const unsigned sd = firstSig + blockIdx.x;
    const unsigned bd = ((unsigned)(127 + expD) << 23) | sd | ((signs & 2) ? 0x80000000u : 0u);
    const float d = __uint_as_float(bd);
    const float y = rcp_refined(d);
    const unsigned ex = ((unsigned)(127 + expX) << 23) | ((signs & 1) ? 0x80000000u : 0u);
    unsigned long long bad = 0, n = 0;
    for (unsigned sx = threadIdx.x; sx < (1u << 23); sx += 256) {
        const float x = __uint_as_float(ex | sx);
        bad += __float_as_uint(div_by_rcp(x, d, y)) == __float_as_uint(x / d) ? 0 : 1;
        n++;
    }
    tally(out, bad, n, 0);
}

}  // namespace

extern "C" {

int rs_debug_exact_ops_mismatches(int op, unsigned firstSig, unsigned countSig, int expX, int expD, unsigned long long* out3) {
    rs_ctx_scope scope(nullptr);
    if (!out3 || op < 0 || op > 5) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_exact_ops_mismatches: bad argument");
    if (op >= 2 && (countSig == 0 || countSig > (1u << 16) || firstSig >= (1u << 23) || firstSig + countSig > (1u << 23) ||
                    expX < -60 || expX > 59 || expD < -60 || expD > 59))
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_debug_exact_ops_mismatches: significands beyond 2^23, more than 65536 per call, or an exponent outside [-60, 59]");
    unsigned long long* d = nullptr;
    RS_TRY(rs_dev_alloc(&d, 3));
    RS_HIP(hipMemsetAsync(d, 0, 24, rs_stream()));
    if (op >= 2) hipLaunchKernelGGL(k_check_division, dim3(countSig), dim3(256), 0, rs_stream(), firstSig, expX, expD, op - 2, d);
    else hipLaunchKernelGGL(k_check_unary, dim3(8192), dim3(256), 0, rs_stream(), op, d);
    int err = rs_after_launch("rs_debug_exact_ops_mismatches");
    if (!err) err = rs_check_hip(hipStreamSynchronize(rs_stream()), "rs_debug_exact_ops_mismatches: synchronize");
    if (!err) err = rs_check_hip(hipMemcpy(out3, d, 24, hipMemcpyDeviceToHost), "rs_debug_exact_ops_mismatches: copy");
    rs_dev_free(d);
    return err;
}

}
