// Cost of global atomics on MI355X as the near-far push solve uses them.  hipcc --offload-arch=gfx950 atom.hip -o atom
//   (a) latency of a dependent chain of RETURNING 64-bit atomicMin on random 128-byte lines, one wave alone;
//   (b) throughput: G blocks x 256 threads, each 16-lane slot doing R rounds of {16 plain 8-byte gathers of random lines}
//       with variants: loads only / + non-returning atomicMin on 1 of 16 lines / + returning atomicMin / + returning
//       atomicExch on a separate int array / + returning atomicAdd on ONE counter per 1024 slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void chain(unsigned long long *d, int n_lines, int hops, unsigned long long *out) {
    unsigned x = 12345u + threadIdx.x / 64 * 977u;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long acc = 0;
    for (int i = 0; i < hops; ++i) {
        x = x * 1664525u + 1013904223u + (unsigned)acc;
        const unsigned line = x % (unsigned)n_lines;
        acc = atomicMin(&d[(size_t)line * 16 + (threadIdx.x & 15)], 0x4000000000000000ull + i);     // returning
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}
template <int MODE>
__global__ __launch_bounds__(256) void tput(unsigned long long *d, int *stamps, int *counters, int n_lines, int rounds, unsigned long long *sink) {
    const unsigned s = threadIdx.x & 15;
    const unsigned slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    unsigned x = slot * 2654435761u + 17u;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned long long v[16];
        unsigned line[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { x = x * 1664525u + 1013904223u; line[j] = x % (unsigned)n_lines; v[j] = d[(size_t)line[j] * 16 + s]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j];
        if (MODE == 1) { if ((s & 7) == 0) atomicMin(&d[(size_t)line[3] * 16 + s], 0x3ff0000000000000ull + r); }          // result unused -> no return
        if (MODE >= 2) { unsigned long long o = 0; if ((s & 7) == 0) o = atomicMin(&d[(size_t)line[3] * 16 + s], 0x3ff0000000000000ull + r); acc += o; }
        if (MODE >= 3) { int o = 0; if (s < 2) o = atomicExch(&stamps[line[5 + s]], r + 1); acc += o; }
        if (MODE >= 4) { int o = 0; if (s == 0) o = atomicAdd(&counters[(slot >> 10) * 32], 1); acc += o; }
        x += (unsigned)acc;                               // rounds depend on each other, like the solver's sweeps do not -- worst case
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}
int main() {
    const int n_lines = 60000 * 32;                       // 32 batches x 60 000 rows of 128 bytes = 246 MB
    unsigned long long *d, *o; int *st, *ct;
    hipMalloc(&d, (size_t)n_lines * 128); hipMalloc(&o, 64); hipMalloc(&st, (size_t)n_lines * 4); hipMalloc(&ct, 1 << 20);
    hipMemset(d, 0x7f, (size_t)n_lines * 128); hipMemset(st, 0, (size_t)n_lines * 4); hipMemset(ct, 0, 1 << 20);
    unsigned long long h[2];
    for (int waves : {1, 4}) {
        for (int rep = 0; rep < 2; ++rep) { chain<<<1, 64 * waves>>>(d, n_lines, 2000, o); hipDeviceSynchronize(); }
        hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        printf("returning atomicMin chain, %d wave(s): %.0f ns per dependent atomic\n", waves, (double)h[0] * 10.0 / 2000);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"16 gathers", "+ non-returning atomicMin", "+ returning atomicMin", "+ returning atomicExch x2", "+ hot counter atomicAdd"};
    for (int blocks : {64, 2048}) for (int mode = 0; mode < 5; ++mode) {
        const int rounds = 64;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) { case 0: tput<0><<<blocks, 256>>>(d, st, ct, n_lines, rounds, o); break; case 1: tput<1><<<blocks, 256>>>(d, st, ct, n_lines, rounds, o); break;
                            case 2: tput<2><<<blocks, 256>>>(d, st, ct, n_lines, rounds, o); break; case 3: tput<3><<<blocks, 256>>>(d, st, ct, n_lines, rounds, o); break;
                            default: tput<4><<<blocks, 256>>>(d, st, ct, n_lines, rounds, o); }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%5d blocks, %-28s: %.3f ms for %d rounds = %.2f us per round, %.1f G line-gathers/s\n", blocks, names[mode], ms, rounds,
               ms * 1e3 / rounds, (double)blocks * 16 * 16 * rounds / (ms * 1e-3) / 1e9);
    }
    return 0;
}
