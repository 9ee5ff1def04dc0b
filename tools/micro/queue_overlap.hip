// tools/micro/queue_overlap.hip -- which streams of a process run their kernels next to each other, and which take turns?
// Two spinning kernels of SPIN_US each (one wave each: they occupy nothing), one per stream, launched back to back; concurrent streams
// finish both after ~SPIN_US, streams whose hardware queues share a pipe of the command processor after ~2 x SPIN_US.
// Prints the matrix for: the legacy default stream (D), PRE high-priority streams made first (argv[1], default 0; they stay idle),
// one ordinary stream (L, "the caller's"), and four streams of each priority level (H0-3, N0-3, W0-3 = high, normal, low).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/queue_overlap.hip -o tools/micro/queue_overlap && tools/micro/queue_overlap [PRE]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
    const int pre = argc > 1 ? std::atoi(argv[1]) : 0;
    const double spinUs = 100.0;
    const long long ticks = (long long)(spinUs * 100.0);            // wall_clock64: 100 MHz
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    std::vector<hipStream_t> preStreams(pre);
    for (auto& s : preStreams) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1);
    std::vector<std::string> names{ "D" };
    std::vector<hipStream_t> st{ nullptr };
    hipStream_t l; (void)hipStreamCreateWithFlags(&l, hipStreamNonBlocking); names.push_back("L"); st.push_back(l);
    const char tag[3] = { 'H', 'N', 'W' };
    for (int p = -1; p <= 1; p++) for (int i = 0; i < 4; i++) {
        hipStream_t s; (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p);
        names.push_back(std::string(1, tag[p + 1]) + std::to_string(i)); st.push_back(s);
    }
    hipEvent_t e0, e1, e2;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    for (auto s : st) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100); }
    (void)hipDeviceSynchronize();
    const int n = (int)st.size();
    std::printf("# %d idle high-priority streams made first; ratio = time until both of two %g-us spins have finished / %g (1 = side by side, 2 = one after the other)\n      ", pre, spinUs, spinUs);
    for (int j = 0; j < n; j++) std::printf("%5s", names[j].c_str());
    std::printf("\n");
    for (int i = 0; i < n; i++) {
        std::printf("%5s ", names[i].c_str());
        for (int j = 0; j < n; j++) {
            if (j == i) { std::printf("    -"); continue; }
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, st[i]);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[j], ticks);
                (void)hipEventRecord(e1, st[i]); (void)hipEventRecord(e2, st[j]);
                (void)hipDeviceSynchronize();
                float a = 0, b = 0;
                (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e0, e2);
                const double t = (a > b ? a : b) * 1e3 / spinUs;
                if (t < best) best = t;
            }
            std::printf(" %4.1f", best);
        }
        std::printf("\n");
    }
    return 0;
}
