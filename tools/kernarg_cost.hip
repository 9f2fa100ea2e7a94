// Developer probe (not product): what does a by-value kernel argument block of S bytes cost a launch of G blocks?
// Every block's wavefronts read the arguments with scalar loads; where the runtime keeps the argument block (host memory
// behind PCIe, or device memory with HIP_FORCE_DEV_KERNARG=1) decides what a cold scalar-cache miss costs.
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_cost.hip -o gpurun_exp/kernarg_cost
//   gpurun_exp/kernarg_cost; HIP_FORCE_DEV_KERNARG=1 gpurun_exp/kernarg_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N> struct Big { int v[N]; };
template <int N, bool ALL> __global__ __launch_bounds__(256) void k(Big<N> b, int *out) {
    int sum = b.v[0] + b.v[N - 1];
    if (ALL) {
#pragma unroll
        for (int i = 16; i < N; i += 16) sum += b.v[i];  // one word of every 64-byte line of the block
    }
    if (sum == 123456789) out[blockIdx.x * blockDim.x + threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void kptr(const int *p, int n, int *out) {  // the same words from device memory
    int sum = 0;
    for (int i = 0; i < n; i += 16) sum += p[i];
    if (sum == 123456789) out[blockIdx.x * blockDim.x + threadIdx.x] = 1;
}
template <int N, bool ALL> float run(int grid, int *d) {
    Big<N> b;
    for (int i = 0; i < N; ++i) b.v[i] = i;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<N, ALL>), dim3(grid), dim3(256), 0, 0, b, d);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<N, ALL>), dim3(grid), dim3(256), 0, 0, b, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / 200.0f;
}
float runPtr(int n, int grid, int *d, const int *p) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kptr, dim3(grid), dim3(256), 0, 0, p, n, d);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kptr, dim3(grid), dim3(256), 0, 0, p, n, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / 200.0f;
}
int main() {
    int *d = nullptr, *p = nullptr;
    hipMalloc(&d, 1 << 26);
    hipMalloc(&p, 1 << 16);
    hipMemset(p, 0, 1 << 16);
    const char *env = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s   us per launch (200 back-to-back launches on the null stream)\n", env ? env : "(unset)");
    printf("%-34s %10s %10s %10s %10s\n", "arguments", "256 blk", "1024 blk", "2048 blk", "8192 blk");
    const int grids[4] = {256, 1024, 2048, 8192};
#define ROW(N, ALL, name)                                                     \
    {                                                                         \
        printf("%-34s", name);                                                \
        for (int g : grids) printf(" %10.2f", run<N, ALL>(g, d));             \
        printf("\n");                                                         \
    }
    ROW(4, false, "16 B")
    ROW(64, true, "256 B, all lines read")
    ROW(256, false, "1 KB, first + last word read")
    ROW(256, true, "1 KB, all lines read")
    ROW(512, true, "2 KB, all lines read")
    ROW(1000, true, "4 KB, all lines read")
    for (int n : {256, 1000}) {
        printf("%-34s", n == 256 ? "pointer to 1 KB in device memory" : "pointer to 4 KB in device memory");
        for (int g : grids) printf(" %10.2f", runPtr(n, g, d, p));
        printf("\n");
    }
    return 0;
}
