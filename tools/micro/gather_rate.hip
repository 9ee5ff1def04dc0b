// tools/micro/gather_rate.hip -- the rate at which a CU serves DEPENDENT per-lane 16-byte gathers: the roofline of an incoherent tree walk.
//
// A per-lane walk of a 16-byte-node tree (rs_scene.h walk_occlusion_tree / walk_ordered_tree) is, to the memory system, a pointer chase
// per lane: load 16 bytes, a few dozen ALU instructions, the next address depends on what was loaded.  Neither the HBM roofline (the trees
// sit in L2 / the Infinity Cache) nor the VALU issue rate describes it.  This measures what the hardware can do with that pattern:
//
//   every lane chases its own chain through a table of 16-byte records (record.x = index of the next record, a random permutation cycle),
//   ALU instructions per step: 0 or ~32 (the shadow walk has 29 per step),   lanes active per wave: 64 or 32 (the walks run at ~50 %),
//   table sizes 16 KB .. 128 MB (L1 of a CU: 32 KB; L2 of one XCD: 4 MB; Infinity Cache: 256 MB),   waves per SIMD 4 / 8.
//
// Output: gathers per second chip-wide and per clock and CU, per configuration.
//     hipcc --offload-arch=gfx950 -O2 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

template <int ALU, bool COHERENT>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ table, unsigned mask, int steps, int activeLanes, unsigned* out) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    // COHERENT: the lanes of a wave start 16 bytes apart (neighbours share 128-byte lines at first, then diverge with the chains)
    unsigned idx = COHERENT ? ((gid * 2654435761u) & ~63u) + lane : gid * 2654435761u;
    idx &= mask;
    float acc = (float)lane;
    if (lane < (unsigned)activeLanes) {
        for (int i = 0; i < steps; i++) {
            const uint4 n = table[idx];
            if (ALU) {
                float t = __uint_as_float((n.y & 0x007fffffu) | 0x3f800000u);
#pragma unroll
                for (int k = 0; k < ALU; k++) t = __builtin_fmaf(t, 1.0000001f, acc * 1e-9f);
                acc += t;
            }
            idx = n.x;
        }
    }
    out[gid] = idx + (unsigned)acc;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { std::fprintf(stderr, "no GPU\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    std::printf("%s: %d CUs at %.2f GHz\n", prop.name, cus, ghz);
    std::printf("%-9s %-6s %-5s %-6s %-9s | %9s %9s %8s\n", "table", "waves", "alu", "lanes", "start", "Ggather/s", "per clk,CU", "us");
    unsigned* out = nullptr;
    const int maxThreads = cus * 4 * 8 * 64;
    if (hipMalloc((void**)&out, sizeof(unsigned) * (size_t)maxThreads) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t kb : { 16, 64, 256, 1024, 4096, 16384, 32768, 65536, 131072 }) {
        const size_t n = kb * 1024 / 16;                            // records, a power of two
        std::vector<unsigned> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1234u + (unsigned)kb);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint4> h(n);
        for (size_t i = 0; i < n; i++) h[perm[i]] = uint4{ perm[(i + 1) % n], (unsigned)rng(), 0u, 0u };      // one cycle through all records
        uint4* table = nullptr;
        if (hipMalloc((void**)&table, n * 16) != hipSuccess) return 1;
        (void)hipMemcpy(table, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int wavesPerSimd : { 4, 8 }) {
            for (int alu : { 0, 32 }) {
                for (int lanes : { 64, 32 }) {
                    for (int coherent = 0; coherent < 2; coherent++) {
                        if (coherent && (alu == 0 || lanes == 32)) continue;
                        const int blocks = cus * wavesPerSimd, steps = 512;          // 4 waves per block: blocks per CU = waves per SIMD
                        auto launch = [&] {
                            if (alu == 0) hipLaunchKernelGGL((k_chase<0, false>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)n - 1u, steps, lanes, out);
                            else if (!coherent) hipLaunchKernelGGL((k_chase<32, false>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)n - 1u, steps, lanes, out);
                            else hipLaunchKernelGGL((k_chase<32, true>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)n - 1u, steps, lanes, out);
                        };
                        launch();
                        (void)hipDeviceSynchronize();
                        float best = 1e30f;
                        for (int rep = 0; rep < 3; rep++) {
                            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
                            best = std::min(best, ms);
                        }
                        const double gathers = (double)blocks * 4.0 * lanes * steps;
                        const double rate = gathers / (best * 1e-3);
                        std::printf("%6zu KB %-6d %-5d %-6d %-9s | %9.1f %9.3f %8.1f\n", kb, wavesPerSimd, alu, lanes, coherent ? "adjacent" : "random",
                                    rate * 1e-9, rate / (cus * ghz * 1e9), best * 1e3);
                    }
                }
            }
        }
        (void)hipFree(table);
    }
    return 0;
}
