// Dependent-load latency seen by ONE resident workgroup (the k-means++ chain's regime): pointer chase over buffers of
// several sizes, alone on the chip and beside a streaming kernel on the other CUs.  hipcc --offload-arch=gfx950 chase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const int *next, int hops, int start, unsigned long long *out) {
    int p = start + 977 * (threadIdx.x >> 6);            // every wave walks its own chain (lanes of a wave: the same address)
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) p = next[p];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)p; }
}
__global__ void stream(const float4 *a, float4 *b, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    for (size_t kb : {256, 16 * 1024, 512 * 1024}) {
        size_t n = kb * 1024 / 4, mb = kb / 1024;
        std::vector<int> perm(n), nxt(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 g(1);
        std::shuffle(perm.begin(), perm.end(), g);
        for (size_t i = 0; i < n; ++i) nxt[perm[i]] = perm[(i + 1) % n];
        int *d; unsigned long long *o; hipMalloc(&d, n * 4); hipMalloc(&o, 32);
        hipMemcpy(d, nxt.data(), n * 4, hipMemcpyHostToDevice);
        float4 *sa, *sb; size_t sn = 64 * 1024 * 1024; hipMalloc(&sa, sn * 16); hipMalloc(&sb, sn * 16);
        hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
        for (int busy = 0; busy < 2; ++busy) {
            unsigned long long h[3];
            for (int waves : {1, 16}) {
                const int hops = busy ? 4000 : 40000;          // alone: a launch of tens of ms, like the resident chain
                for (int rep = 0; rep < 2; ++rep) {
                    if (busy) stream<<<240 * 4, 256, 0, s2>>>(sa, sb, sn, 4);
                    chase<<<1, 64 * waves, 0, s1>>>(d, hops, 0, o);
                    hipStreamSynchronize(s1);
                    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
                    hipDeviceSynchronize();
                }
                printf("buffer %4zu MB %2d waves %s: %.0f cycles = %.0f ns per dependent load (clock %.2f GHz)\n", mb, waves,
                       busy ? "beside a streaming kernel" : "alone on the chip        ", (double)h[0] / hops, (double)h[1] * 10.0 / hops,
                       (double)h[0] / ((double)h[1] * 10.0));
            }
        }
        hipFree(d); hipFree(o); hipFree(sa); hipFree(sb);
    }
    return 0;
}
