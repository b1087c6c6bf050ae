// sssp_device.h -- device code shared by sssp.hip (host-driven solves) and kpp.hip (device-resident chain).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace geo {

// One in-place relaxation sweep of a single-source solve: 16 lanes share one node's adjacency row
// (fp64 path sums, see sssp.hip for why chaotic relaxation reproduces Dijkstra bit for bit).
// Returns true for the threads that stored an improved distance.
template <bool WEIGHTED>
__device__ __forceinline__ bool sweep_single_body(const int32_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices,
                                                  const float *__restrict__ weights, int32_t n, double *d) {
    const int sub = threadIdx.x & 15;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int ngrp = (gridDim.x * blockDim.x) >> 4;
    bool any = false;
    for (int32_t v = grp; v < n; v += ngrp) {
        const int32_t e0 = indptr[v], e1 = indptr[v + 1];
        const double curv = d[v];
        double best = curv;
        for (int32_t e = e0 + sub; e < e1; e += 16) {
            const double w = WEIGHTED ? (double)weights[e] : 1.0;
            best = fmin(best, d[indices[e]] + w);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) best = fmin(best, __shfl_xor(best, off, 16));
        if (sub == 0 && best < curv) {
            d[v] = best;
            any = true;
        }
    }
    return any;
}

}  // namespace geo
