// train_fold.h -- the second stage of the training kernels' two-stage reductions: out[i] = scale * sum over g of
// part[g * stride + i], added in float64 in a FIXED order (a run repeats bit for bit), itself in two launches so that no
// thread adds more than groups / 32 numbers one after the other (one block adding 2 048 partials per column took 80-130 us,
// a dozen times per training step): 32 segments of the groups in parallel, then the 32 segment sums.
// scratch: 32 * n doubles, carved from the end of the caller's partials buffer (the *_partials() sizes include it).
#pragma once
#include "common.h"

#define TF_SEG 32

__global__ __launch_bounds__(256) static void k_tf_stage1(const float *__restrict__ part, int groups, long stride, int n,
                                                          double *__restrict__ scratch)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y, S = gridDim.y;
    if (i >= n) return;
    double acc = 0.0;
    int g = s;
#ifndef TF_NO_BATCH16
    // sixteen independent loads in flight per thread, added in the same fixed order (with four, a thread that folds 128 block partials
    // -- every convolution's sums, eighteen times per training step -- waited for 32 L2 round trips one after the other: 12.9 us)
    for (; g + 15 * S < groups; g += 16 * S) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(long)(g + u * S) * stride + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (double)v[u];
    }
#endif
    for (; g + 3 * S < groups; g += 4 * S) {
        const float a = part[(long)g * stride + i], b = part[(long)(g + S) * stride + i];
        const float c = part[(long)(g + 2 * S) * stride + i], d = part[(long)(g + 3 * S) * stride + i];
        acc += (double)a; acc += (double)b; acc += (double)c; acc += (double)d;
    }
    for (; g < groups; g += S) acc += (double)part[(long)g * stride + i];
    scratch[(long)s * n + i] = acc;
}

template <typename OUT>
__global__ __launch_bounds__(256) static void k_tf_stage2(const double *__restrict__ scratch, int S, int n, double scale,
                                                          OUT *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
#ifndef TF_NO_BATCH16
    if (S == TF_SEG) {                     // (the usual case: all 32 segment sums requested at once, added in order)
        double v[TF_SEG];
#pragma unroll
        for (int s = 0; s < TF_SEG; ++s) v[s] = scratch[(long)s * n + i];
#pragma unroll
        for (int s = 0; s < TF_SEG; ++s) acc += v[s];
    } else
#endif
    for (int s = 0; s < S; ++s) acc += scratch[(long)s * n + i];
    out[i] = (OUT)(acc * scale);
}

template <typename OUT>
static inline void tf_fold(const float *part, int groups, long stride, int n, double scale, OUT *out, double *scratch, hipStream_t st)
{
    const int S = groups < TF_SEG ? groups : TF_SEG;
    k_tf_stage1<<<dim3((n + 255) / 256, S), 256, 0, st>>>(part, groups, stride, n, scratch);
    k_tf_stage2<OUT><<<(n + 255) / 256, 256, 0, st>>>(scratch, S, n, scale, out);
}

#define TF_SCRATCH_FLOATS(n) (2 * TF_SEG * (n))          // floats of scratch behind the partials for an n-column fold
