// What happens to a store stream (the label fill's shape) when a second HIP stream of the same process is busy or waiting:
//   0 alone   1 other stream holds a pending wait on an event recorded behind the fill   2 other stream runs small
//   latency-bound kernels meanwhile   3 other stream runs a second fill   4 as 2, the small kernels limited to few blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill(int *out, int W, int H)
{
    const size_t row = ((size_t)blockIdx.z * H + blockIdx.y) * W;
    const int x = blockIdx.x * 1024 + threadIdx.x * 4;
    i32x4 v = {0, 0, 0, 0};
    if (x < W) __builtin_nontemporal_store(v, (i32x4 *)(out + row + x));
}
__global__ __launch_bounds__(256) void k_chase(const int *next, int *out, int n, int steps)
{
    int i = (blockIdx.x * 256 + threadIdx.x) % n;
    for (int s = 0; s < steps; s++) i = next[i];
    if (i == -1) out[0] = 1;
}
int main()
{
    const int W = 2560, H = 2560, F = 200;
    int *lab[2];
    for (int k = 0; k < 2; k++) CK(hipMalloc(&lab[k], (size_t)W * H * F * 4));
    const int n = 1 << 22;
    int *next, *out;
    CK(hipMalloc(&next, n * 4)); CK(hipMalloc(&out, 4));
    {
        int *h = (int *)malloc(n * 4);
        for (int i = 0; i < n; i++) h[i] = (int)(((long long)i * 2654435761LL + 12345) % (i < (1 << 18) ? (1 << 18) : n));
        CK(hipMemcpy(next, h, n * 4, hipMemcpyHostToDevice));
        free(h);
    }
    hipStream_t a, b;
    int plo = 0, phi = 0;
    CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    printf("stream priorities: least %d greatest %d\n", plo, phi);
    if (getenv("FILL_HIGH")) { CK(hipStreamCreateWithPriority(&a, hipStreamDefault, phi)); CK(hipStreamCreateWithPriority(&b, hipStreamDefault, plo)); printf("fill on the high-priority stream\n"); }
    else { CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b)); }
    hipEvent_t e0, e1, dep;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    const dim3 grid((W + 1023) / 1024, H, F);
    for (int mode = 0; mode < 8; mode++)
        for (int rep = 0; rep < 3; rep++) {
            CK(hipDeviceSynchronize());
            if (mode == 2) for (int k = 0; k < 30; k++) hipLaunchKernelGGL(k_chase, dim3(4000), dim3(256), 0, b, next, out, n, 20);
            if (mode == 4) for (int k = 0; k < 30; k++) hipLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, b, next, out, n, 40);
            // 5: small-table chase (L2 hits), 1200 blocks x 30 launches; 6: the same, 150 blocks; 7: the same, 1200 blocks, 60 launches of 3 steps
            if (mode == 5) for (int k = 0; k < 30; k++) hipLaunchKernelGGL(k_chase, dim3(1200), dim3(256), 0, b, next, out, 1 << 18, 10);
            if (mode == 6) for (int k = 0; k < 30; k++) hipLaunchKernelGGL(k_chase, dim3(150), dim3(256), 0, b, next, out, 1 << 18, 10);
            if (mode == 7) for (int k = 0; k < 60; k++) hipLaunchKernelGGL(k_chase, dim3(1200), dim3(256), 0, b, next, out, 1 << 18, 3);
            if (mode == 3) hipLaunchKernelGGL(k_fill, grid, dim3(256), 0, b, lab[1], W, H);
            CK(hipEventRecord(e0, a));
            hipLaunchKernelGGL(k_fill, grid, dim3(256), 0, a, lab[0], W, H);
            CK(hipEventRecord(e1, a));
            if (mode == 1) {
                CK(hipEventRecord(dep, a));
                CK(hipStreamWaitEvent(b, dep, 0));
                hipLaunchKernelGGL(k_chase, dim3(64), dim3(256), 0, b, next, out, n, 4);
            }
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d rep %d: fill %.3f ms (%.2f TB/s)\n", mode, rep, ms, (double)W * H * F * 4 / ms * 1e-9);
        }
    return 0;
}
