// tools/micro/queue_quads.hip [PRE] [rccl] -- does a set of FOUR streams run four kernels side by side?  The library keeps four streams
// busy in overlapped mode (the caller's and three of its own); this asks, for the caller's stream being the legacy default stream (D),
// an ordinary stream (L) or a high-priority one (P), and for every triple out of four streams of each priority level, how long four
// simultaneous SPIN_US spins take (1 = side by side; 2 = two of them took turns), and the same with a chain of 8 short dependent spins
// per stream with an event hand-over to the next stream in between (closer to what a frame enqueues).
// PRE idle high-priority streams are made first; `rccl`: a one-rank ncclComm is created first (what bench.py --gpus N has in its process).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/queue_quads.hip -o tools/micro/queue_quads -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static double quad(const hipStream_t s[4], long long ticks, hipEvent_t e0, hipEvent_t* e) {
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, s[0]);
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[k], ticks);
        for (int k = 0; k < 4; k++) (void)hipEventRecord(e[k], s[k]);
        (void)hipDeviceSynchronize();
        float worst = 0;
        for (int k = 0; k < 4; k++) { float t = 0; (void)hipEventElapsedTime(&t, e0, e[k]); if (t > worst) worst = t; }
        if (worst < best) best = worst;
    }
    return best * 1e3;
}
// 8 rounds: every stream runs a short spin, records an event, and waits for its neighbour's event of the round before the next spin
static double chained(const hipStream_t s[4], long long ticks, hipEvent_t e0, hipEvent_t* e, hipEvent_t (*h)[4]) {
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, s[0]);
        for (int r = 0; r < 8; r++) {
            for (int k = 0; k < 4; k++) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[k], ticks); (void)hipEventRecord(h[r][k], s[k]); }
            for (int k = 0; k < 4; k++) (void)hipStreamWaitEvent(s[k], h[r][(k + 1) % 4], 0);
        }
        for (int k = 0; k < 4; k++) (void)hipEventRecord(e[k], s[k]);
        (void)hipDeviceSynchronize();
        float worst = 0;
        for (int k = 0; k < 4; k++) { float t = 0; (void)hipEventElapsedTime(&t, e0, e[k]); if (t > worst) worst = t; }
        if (worst < best) best = worst;
    }
    return best * 1e3;
}

int main(int argc, char** argv) {
    const int pre = argc > 1 ? std::atoi(argv[1]) : 0;
    const bool rccl = argc > 2 && std::strcmp(argv[2], "rccl") == 0;
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    if (rccl) {
        ncclUniqueId id; ncclComm_t comm;
        if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "rccl init failed\n"); return 1; }
    }
    std::vector<hipStream_t> preStreams(pre);
    for (auto& s : preStreams) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1);
    hipStream_t L, P;
    (void)hipStreamCreateWithFlags(&L, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&P, hipStreamNonBlocking, -1);
    hipStream_t lv[3][4];
    for (int p = 0; p < 3; p++) for (int i = 0; i < 4; i++) (void)hipStreamCreateWithPriority(&lv[p][i], hipStreamNonBlocking, p - 1);
    hipEvent_t e0, e[4], h[8][4];
    (void)hipEventCreate(&e0);
    for (auto& x : e) (void)hipEventCreate(&x);
    for (auto& r : h) for (auto& x : r) (void)hipEventCreateWithFlags(&x, hipEventDisableTiming);
    const double spinUs = 100.0;
    std::printf("# %d idle high-priority streams first, RCCL communicator %s; four streams at once: [caller's stream + three of level X without stream i]\n", pre, rccl ? "yes" : "no");
    std::printf("# columns: 4 x %g us side by side (us) | 8 rounds of 4 x 20 us with event hand-overs (us; ideal 160)\n", spinUs);
    const char* callers[3] = { "default stream ", "ordinary stream", "high-prio stream" };
    const hipStream_t cs[3] = { nullptr, L, P };
    const char* levels[3] = { "high  ", "normal", "low   " };
    for (int c = 0; c < 3; c++)
        for (int p = 0; p < 3; p++) {
            std::printf("%s + %s:", callers[c], levels[p]);
            for (int skip = 0; skip < 4; skip++) {
                hipStream_t s[4] = { cs[c] };
                int n = 1;
                for (int i = 0; i < 4; i++) if (i != skip) s[n++] = lv[p][i];
                std::printf("  %5.0f|%5.0f", quad(s, (long long)(spinUs * 100), e0, e), chained(s, 2000, e0, e, h));
            }
            std::printf("\n");
        }
    return 0;
}
