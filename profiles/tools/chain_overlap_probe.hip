// Probe for the "one-launch labelling chain beside the label fill" lead: what do a pure store stream of the
// label fill's shape and a PERSISTENT latency-bound kernel (one block per CU, dependent loads, device-wide barriers between its
// phases) cost each other when they run on two streams of one device?
//   F: grid (W / 1024, H, frames) x 256 threads, one 16-byte non-temporal store per lane (k_label_fill without its lookups)
//   C: `blocks` blocks of 256 threads; `phases` phases of `steps` dependent loads per thread through a random cycle in a table
//      of `table_mb` MiB, a device-wide barrier (one atomic counter, bounded spin) after every phase
// Prints the duration of each alone and of both when started together (HIP events on their own streams).
// build: hipcc -O3 --offload-arch=gfx950 -o chain_overlap_probe chain_overlap_probe.hip ; run: ./chain_overlap_probe [blocks] [phases] [steps] [table_mb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void kF(int *__restrict__ labels, int W, int H)
{
    const int lane = threadIdx.x & 63, x = blockIdx.x * 1024 + (threadIdx.x >> 6) * 256 + 4 * lane;
    if (x >= W) return;
    const size_t row = ((size_t)blockIdx.z * H + blockIdx.y) * W;
    const i32x4_t z = {0, 0, 0, 0};
    __builtin_nontemporal_store(z, (i32x4_t *)(labels + row + x));
}

__global__ __launch_bounds__(256) void kC(const unsigned *__restrict__ next, unsigned n, int phases, int steps, unsigned *barrier, unsigned *sink,
                                          unsigned *timeout)
{
    unsigned p = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n;
    for (int ph = 0; ph < phases; ph++) {
        for (int s = 0; s < steps; s++) p = next[p];
        __syncthreads();
        if (threadIdx.x == 0) {                               // device-wide barrier: every block arrives, then waits for all (bounded)
            __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(ph + 1) * gridDim.x;
            int spins = 0;
            while (__hip_atomic_load(barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(4);
            if (spins >= (1 << 20)) *timeout = 1;
        }
        __syncthreads();
    }
    if (p == 0xffffffffu) *sink = p;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, phases = argc > 2 ? atoi(argv[2]) : 8, steps = argc > 3 ? atoi(argv[3]) : 24;
    const int table_mb = argc > 4 ? atoi(argv[4]) : 64;
    const int W = 2560, H = 2560, frames = 200;
    int *labels; CHK(hipMalloc(&labels, (size_t)frames * H * W * 4));
    const unsigned n = (unsigned)table_mb * (1u << 20) / 4;
    std::vector<unsigned> perm(n), nxt(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (unsigned i = 0; i < n; i++) nxt[perm[i]] = perm[(i + 1) % n];       // one random cycle
    unsigned *d_next, *d_misc; CHK(hipMalloc(&d_next, (size_t)n * 4)); CHK(hipMalloc(&d_misc, 64));
    CHK(hipMemcpy(d_next, nxt.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipStream_t sf, sc; CHK(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t f0, f1, c0, c1; CHK(hipEventCreate(&f0)); CHK(hipEventCreate(&f1)); CHK(hipEventCreate(&c0)); CHK(hipEventCreate(&c1));
    auto run = [&](bool withF, bool withC, float &msF, float &msC) {
        CHK(hipMemset(d_misc, 0, 64));
        CHK(hipDeviceSynchronize());
        if (withC) { CHK(hipEventRecord(c0, sc)); hipLaunchKernelGGL(kC, dim3(blocks), dim3(256), 0, sc, d_next, n, phases, steps, d_misc, d_misc + 1, d_misc + 2); CHK(hipEventRecord(c1, sc)); }
        if (withF) { CHK(hipEventRecord(f0, sf)); hipLaunchKernelGGL(kF, dim3((W + 1023) / 1024, H, frames), dim3(256), 0, sf, labels, W, H); CHK(hipEventRecord(f1, sf)); }
        CHK(hipDeviceSynchronize());
        msF = msC = 0;
        if (withF) CHK(hipEventElapsedTime(&msF, f0, f1));
        if (withC) CHK(hipEventElapsedTime(&msC, c0, c1));
        unsigned t; CHK(hipMemcpy(&t, d_misc + 2, 4, hipMemcpyDeviceToHost));
        if (t) printf("  (barrier timed out)\n");
    };
    float a, b;
    for (int rep = 0; rep < 3; rep++) {
        run(true, false, a, b); printf("fill alone            %.3f ms\n", a);
        run(false, true, a, b); printf("chain alone                       %.3f ms   (%d blocks, %d phases x %d loads, %d MiB table)\n", b, blocks, phases, steps, table_mb);
        run(true, true, a, b);  printf("together: fill        %.3f ms, chain %.3f ms\n", a, b);
    }
    return 0;
}
