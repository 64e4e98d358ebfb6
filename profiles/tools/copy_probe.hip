// Micro-benchmark: what a read+write stream can reach on this box, in the shapes the background kernel could use.
//   A: flat 16-B-per-thread copy, one element per thread
//   B: flat copy, grid-stride with 4 x 16 B in flight per thread
//   C: frame-streaming shape: block = strip of 256*16 B, loops over the 200 frames (read frame k, write frame k)
//   D: as C with 8 B per thread (2 KiB strips)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void kA(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ __launch_bounds__(256) void kB(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n) v[k] = a[i + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n) b[i + 256 * k] = v[k];
}
template <typename T, int DEPTH>
__global__ __launch_bounds__(256) void kC(const T *__restrict__ a, T *__restrict__ b, size_t per_frame, int nframes)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_frame) return;
    for (int f = 0; f < nframes; f += DEPTH) {
        T v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) v[k] = a[(size_t)(f + k) * per_frame + i];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) b[(size_t)(f + k) * per_frame + i] = v[k];
    }
}
// E: as C, but a block only walks `chunk` frames (grid = strips x chunks, strip fastest)
template <typename T>
__global__ __launch_bounds__(256) void kE(const T *__restrict__ a, T *__restrict__ b, size_t per_frame, int chunk, int nstrips)
{
    const int strip = blockIdx.x % nstrips, c = blockIdx.x / nstrips;
    size_t i = (size_t)strip * 256 + threadIdx.x;
    if (i >= per_frame) return;
    for (int f = c * chunk; f < (c + 1) * chunk; f++) b[(size_t)f * per_frame + i] = a[(size_t)f * per_frame + i];
}
// F: one-shot threads like A, but blocks ordered frame-fastest (block -> strip = id / nframes, frame = id % nframes)
__global__ __launch_bounds__(256) void kF(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t per_frame, int nframes)
{
    const size_t strip = blockIdx.x / nframes, f = blockIdx.x % nframes;
    size_t i = strip * 256 + threadIdx.x;
    if (i < per_frame) b[f * per_frame + i] = a[f * per_frame + i];
}
// G: as C (1 in flight) with non-temporal loads and stores
__global__ __launch_bounds__(256) void kG(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t per_frame, int nframes)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_frame) return;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *pa = (const u4 *)a; u4 *pb = (u4 *)b;
    for (int f = 0; f < nframes; f++) { u4 v = __builtin_nontemporal_load(pa + (size_t)f * per_frame + i); __builtin_nontemporal_store(v, pb + (size_t)f * per_frame + i); }
}
// H: as C with 1024-thread blocks (16 KiB strips)
__global__ __launch_bounds__(1024) void kH(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t per_frame, int nframes)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i >= per_frame) return;
    for (int f = 0; f < nframes; f++) b[(size_t)f * per_frame + i] = a[(size_t)f * per_frame + i];
}
// K: as C, blocks kept loosely in step: every G frames a block counts itself in and waits (bounded spin) until every
// block has finished the group before the previous one
template <int G>
__global__ __launch_bounds__(256) void kK(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t per_frame, int nframes, unsigned *cnt, unsigned nblk)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int f = 0; f < nframes; f++) {
        if (f % G == 0 && f >= G) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&cnt[f / G - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f >= 2 * G) {
                    int spins = 0;
                    while (__hip_atomic_load(&cnt[f / G - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk && ++spins < 20000) __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
        if (i < per_frame) b[(size_t)f * per_frame + i] = a[(size_t)f * per_frame + i];
    }
}
// W: the window shape without a frame loop: a thread makes G consecutive output frames of one 16-byte (or 8-byte) strip
// from G + 11 input frames (11 of them read before by neighbours in time: cache hits), blocks ordered tile-major, then
// frame group, then strip inside the tile.  Output = max over the window (stand-in arithmetic).
template <typename T, int G>
__global__ __launch_bounds__(256) void kW(const T *__restrict__ a, T *__restrict__ b, size_t per_frame, int nframes, int strips_per_tile)
{
    const int ngroups = (nframes - 11) / G;                   // the tail frames are left out of this probe
    const unsigned bpt = strips_per_tile / 256;               // blocks per tile and frame group
    const unsigned tile = blockIdx.x / (bpt * ngroups), r = blockIdx.x % (bpt * ngroups);
    const unsigned g = r / bpt, sb = r % bpt;
    const size_t i = (size_t)tile * strips_per_tile + sb * 256 + threadIdx.x;
    if (i >= per_frame) return;
    T v[G + 11];
#pragma unroll
    for (int k = 0; k < G + 11; k++) v[k] = a[(size_t)(g * G + k) * per_frame + i];
#pragma unroll
    for (int j = 0; j < G; j++) {
        T m = v[j];
#pragma unroll
        for (int k = 1; k < 12; k++) { m.x = m.x > v[j + k].x ? m.x : v[j + k].x; m.y = m.y > v[j + k].y ? m.y : v[j + k].y; }
        __builtin_nontemporal_store(m.x, &b[(size_t)(g * G + j) * per_frame + i].x);
        __builtin_nontemporal_store(m.y, &b[(size_t)(g * G + j) * per_frame + i].y);
    }
}
int main()
{
    const size_t P = 2560 * 2560, n = 200, bytes = P * n;
    uint8_t *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; r++) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %.3f ms  %.2f TB/s (read+write)\n", name, ms, 2.0 * bytes / ms * 1e-9);
        return 0;
    };
    const size_t n16 = bytes / 16, pf16 = P / 16, pf8 = P / 8;
    run("A flat 16B/thread", [&] { hipLaunchKernelGGL(kA, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n16); });
    run("B flat 4x16B/thread", [&] { hipLaunchKernelGGL(kB, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n16); });
    run("C strip 4KiB, 1 frame in flight", [&] { hipLaunchKernelGGL((kC<uint4, 1>), dim3((pf16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    run("C strip 4KiB, 4 frames in flight", [&] { hipLaunchKernelGGL((kC<uint4, 4>), dim3((pf16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    run("C strip 4KiB, 8 frames in flight", [&] { hipLaunchKernelGGL((kC<uint4, 8>), dim3((pf16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    run("D strip 2KiB, 1 frame in flight", [&] { hipLaunchKernelGGL((kC<uint2, 1>), dim3((pf8 + 255) / 256), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, pf8, (int)n); });
    run("D strip 2KiB, 4 frames in flight", [&] { hipLaunchKernelGGL((kC<uint2, 4>), dim3((pf8 + 255) / 256), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, pf8, (int)n); });
    run("D strip 2KiB, 8 frames in flight", [&] { hipLaunchKernelGGL((kC<uint2, 8>), dim3((pf8 + 255) / 256), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, pf8, (int)n); });
    for (int chunk : {100, 50, 25, 10, 5})
        run(chunk == 100 ? "E chunk 100" : chunk == 50 ? "E chunk 50" : chunk == 25 ? "E chunk 25" : chunk == 10 ? "E chunk 10" : "E chunk 5",
            [&] { hipLaunchKernelGGL((kE<uint4>), dim3((unsigned)(1600 * (n / chunk))), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, chunk, 1600); });
    run("F one-shot, frame-fastest order", [&] { hipLaunchKernelGGL(kF, dim3((unsigned)(1600 * n)), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    run("G strip 4KiB, non-temporal", [&] { hipLaunchKernelGGL(kG, dim3(1600), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    run("H strip 16KiB (1024 threads)", [&] { hipLaunchKernelGGL(kH, dim3(400), dim3(1024), 0, 0, (const uint4 *)a, (uint4 *)b, pf16, (int)n); });
    {
        auto w = [&](const char *name, auto kern, size_t pf, int G, int tiles) {
            const int spt = (int)(pf / tiles), ngroups = (int)((n - 11) / G);
            run(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * ngroups * (spt / 256))), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, pf, (int)n, spt); });
        };
        w("W 8B G=4 tiles=1", kW<uint2, 4>, pf8, 4, 1);
        w("W 8B G=4 tiles=4", kW<uint2, 4>, pf8, 4, 4);
        w("W 8B G=4 tiles=8", kW<uint2, 4>, pf8, 4, 8);
        w("W 8B G=4 tiles=16", kW<uint2, 4>, pf8, 4, 16);
        w("W 8B G=8 tiles=4", kW<uint2, 8>, pf8, 8, 4);
        w("W 8B G=8 tiles=8", kW<uint2, 8>, pf8, 8, 8);
        w("W 8B G=8 tiles=16", kW<uint2, 8>, pf8, 8, 16);
        w("W 8B G=2 tiles=8", kW<uint2, 2>, pf8, 2, 8);
    }
    unsigned *cnt; CK(hipMalloc(&cnt, 4096));
    CK(hipDeviceSynchronize());
    hipFree(a); hipFree(b);
    return 0;
}
