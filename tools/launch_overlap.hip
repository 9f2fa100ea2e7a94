// What does the boundary between two dependent launches cost, and can the successor's dispatch be hidden behind the
// predecessor?  (round 6; the headline step is three dependent launches of 9-15 us with ~2 us between them.)
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_overlap tools/launch_overlap.hip && /tmp/launch_overlap
//
// A chain of N "steps" of three kernels each.  Every kernel: G blocks of 256 threads; each block runs a dependent chain of
// global loads of `rounds` rounds (a stand-in for the engine's latency-bound kernels), then bumps a done-counter.  Forms:
//   inorder    plain launches on one stream (the barrier bit between them): what the engine does now
//   anyorder   launches 2 and 3 of a step with hipExtAnyOrderLaunch; a block's first thread waits (bounded) until the
//              predecessor's done-counter has reached G before it starts its chain
//   streams    the three kernels of a step on three streams, same device-side wait
// Reported: wall us per step over the chain, and whether a wait ever ran into its bound (= the launches did not overlap and
// the form is useless / would deadlock).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

struct Sync {
    unsigned long long *done;  // [nKernels] blocks finished, per launch in the chain
    int *timedOut;
};

template <bool WAIT>
__global__ __launch_bounds__(256) void k_chain(const int *next, int *out, int rounds, Sync s, int self, int nBlocksPrev) {
    if (WAIT && self > 0) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            // (100 MHz wall clock: 2 000 000 ticks = 20 ms)
            while (__hip_atomic_load(&s.done[self - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long) nBlocksPrev) {
                if (wall_clock64() - t0 > 2000000ULL) {
                    atomicAdd(s.timedOut, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int v = i;
    for (int r = 0; r < rounds; ++r) v = next[v];
    out[i] = v + out[i];
    if (WAIT) {
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&s.done[self], 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 288, rounds = argc > 2 ? atoi(argv[2]) : 6, steps = argc > 3 ? atoi(argv[3]) : 200;
    const int n = G * 256, nK = steps * 3;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int) (((long long) i * 7919 + 12345) % n);
    int *next, *out, *timedOut;
    unsigned long long *done;
    CHECK(hipMalloc(&next, n * sizeof(int)));
    CHECK(hipMalloc(&out, n * sizeof(int)));
    CHECK(hipMalloc(&done, nK * sizeof(unsigned long long)));
    CHECK(hipMalloc(&timedOut, sizeof(int)));
    CHECK(hipMemcpy(next, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemset(out, 0, n * sizeof(int)));
    hipStream_t st[3];
    for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Sync sy{done, timedOut};
    auto reset = [&]() {
        CHECK(hipMemset(done, 0, nK * sizeof(unsigned long long)));
        CHECK(hipMemset(timedOut, 0, sizeof(int)));
        CHECK(hipDeviceSynchronize());
    };
    auto report = [&](const char *name, double us) {
        int t = 0;
        CHECK(hipMemcpy(&t, timedOut, sizeof t, hipMemcpyDeviceToHost));
        printf("%-28s %7.2f us per step of 3 launches (G = %d blocks, %d rounds)%s\n", name, us / steps, G, rounds,
               t ? "   ** waits ran into their bound: no overlap **" : "");
        fflush(stdout);
    };
    using clk = std::chrono::steady_clock;
    for (int rep = 0; rep < 2; ++rep) {
        // ---- in order, no device-side wait
        reset();
        auto t0 = clk::now();
        for (int k = 0; k < nK; ++k) hipLaunchKernelGGL(k_chain<false>, dim3(G), dim3(256), 0, st[0], next, out, rounds, sy, k, G);
        CHECK(hipStreamSynchronize(st[0]));
        report("inorder", std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        // ---- in order WITH the counters (what the bookkeeping alone costs)
        reset();
        t0 = clk::now();
        for (int k = 0; k < nK; ++k) hipLaunchKernelGGL(k_chain<true>, dim3(G), dim3(256), 0, st[0], next, out, rounds, sy, k, G);
        CHECK(hipStreamSynchronize(st[0]));
        report("inorder + counters", std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        // ---- launches 2 and 3 of every step in any order
        reset();
        t0 = clk::now();
        for (int k = 0; k < nK; ++k)
            hipExtLaunchKernelGGL(k_chain<true>, dim3(G), dim3(256), 0, st[0], nullptr, nullptr, (k % 3) ? hipExtAnyOrderLaunch : 0, next, out,
                                  rounds, sy, k, G);
        CHECK(hipStreamSynchronize(st[0]));
        report("anyorder (2 of 3)", std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        // ---- three streams
        reset();
        t0 = clk::now();
        for (int k = 0; k < nK; ++k) hipLaunchKernelGGL(k_chain<true>, dim3(G), dim3(256), 0, st[k % 3], next, out, rounds, sy, k, G);
        for (auto &s : st) CHECK(hipStreamSynchronize(s));
        report("three streams", std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        // ---- two streams
        reset();
        t0 = clk::now();
        for (int k = 0; k < nK; ++k) hipLaunchKernelGGL(k_chain<true>, dim3(G), dim3(256), 0, st[k % 2], next, out, rounds, sy, k, G);
        for (auto &s : st) CHECK(hipStreamSynchronize(s));
        report("two streams", std::chrono::duration<double, std::micro>(clk::now() - t0).count());
    }
    return 0;
}
