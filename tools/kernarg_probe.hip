// Developer probe (not product): does the runtime take kernel arguments beyond 4 KB, and what does a dynamically indexed
// by-value struct argument cost?  hipcc --offload-arch=gfx950 tools/kernarg_probe.hip -o gpurun_exp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int n; int v[N]; };
template <int N> __global__ void k(Big<N> b, int *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.n) out[i] = b.v[i] * 2 + 1;
}
template <int N> int run() {
    Big<N> b;
    b.n = N;
    for (int i = 0; i < N; ++i) b.v[i] = i;
    int *d = nullptr;
    if (hipMalloc(&d, N * sizeof(int)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k<N>, dim3((N + 255) / 256), dim3(256), 0, 0, b, d);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("N=%d (%zu B): launch failed: %s\n", N, sizeof(b), hipGetErrorString(e)); return 1; }
    int *h = new int[N];
    hipMemcpy(h, d, N * sizeof(int), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) bad += h[i] != i * 2 + 1;
    printf("N=%d (%zu B of arguments): %s\n", N, sizeof(b), bad ? "WRONG" : "ok");
    return bad;
}
int main() { return run<512>() | run<1000>() | run<2040>() | run<4000>() | run<8000>(); }
