// Developer probe (not product): what does COLD CODE cost a launch?  The same chain of N dependent integer operations as
// straight-line code (N x 8 bytes of instructions, each fetched once) and as a loop of 16 (one cache line of code, run N / 16
// times), one wavefront per SIMD and many: if the straight-line form takes much longer than the loop, the time is instruction
// fetch, not execution.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_cost.hip -o gpurun_exp/icache_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __global__ __launch_bounds__(256) void straight(unsigned x, unsigned *out) {
    unsigned v = x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) v = (v ^ (0x9E3779B9u + 2654435761u * (unsigned) i)) * 3u + (unsigned) i;
    if (v == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
template <int N> __global__ __launch_bounds__(256) void looped(unsigned x, unsigned *out) {
    unsigned v = x + threadIdx.x;
#pragma unroll 1
    for (int j = 0; j < N / 16; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v = (v ^ (0x9E3779B9u + 2654435761u * (unsigned) i)) * 3u + (unsigned) j;
    }
    if (v == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
template <typename K> float timeIt(K kern, int grid, unsigned *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, 1u, d);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, 1u, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / 100.0f;
}
int main() {
    unsigned *d = nullptr;
    (void) hipMalloc(&d, 1 << 26);
    printf("us per launch, 100 back-to-back launches (a wavefront executes one of these operations in about 3 x 4 cycles)\n");
    printf("%-40s %10s %10s %10s\n", "kernel", "256 blk", "1024 blk", "4096 blk");
    const int grids[3] = {256, 1024, 4096};
#define ROW(K, name)                                                \
    {                                                               \
        printf("%-40s", name);                                      \
        for (int g : grids) printf(" %10.2f", timeIt(K, g, d));     \
        printf("\n");                                               \
    }
    ROW(straight<64>, "straight-line, 64 steps")
    ROW(looped<64>, "loop, 64 steps")
    ROW(straight<512>, "straight-line, 512 steps")
    ROW(looped<512>, "loop, 512 steps")
    ROW(straight<2048>, "straight-line, 2048 steps")
    ROW(looped<2048>, "loop, 2048 steps")
    ROW(straight<8192>, "straight-line, 8192 steps")
    ROW(looped<8192>, "loop, 8192 steps")
    return 0;
}
