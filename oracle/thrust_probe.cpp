/*
 * thrust_probe.cpp -- the third-party RNG the reference samples with (src/sampler.h:38-49:
 * thrust::default_random_engine + thrust::uniform_real_distribution<float>(0,1)), run from the
 * dependency itself.  The reference pins no Thrust version (CMakeLists.txt:26 asks for CUDA 10);
 * this image ships rocThrust (THRUST_VERSION 200805), whose minstd_rand / uniform_real code is the
 * same header-only algorithm.  Built host-only by oracle/Makefile into oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY.
 */
#include <thrust/random.h>
#include <cstdint>

extern "C" void thr_rng_stream_raw(int n, const int* seeds, int m, float* out) {
    for (int i = 0; i < n; i++) {
        thrust::default_random_engine rng(seeds[i]);
        for (int k = 0; k < m; k++)
            out[(size_t)i * m + k] = thrust::uniform_real_distribution<float>(0.f, 1.f)(rng);
    }
}
extern "C" int thr_version(void) { return THRUST_VERSION; }
