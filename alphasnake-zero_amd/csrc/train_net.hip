// train_net.hip -- what the training step (AlphaNNet.train = model.fit, alpha_nnet.py:58-59)
// needs besides the tower's three convolution passes (conv_split.hip, train_wgrad.hip) and the element-wise batch-norm
// passes (train.hip):
//
//   batch-norm bookkeeping   k_bn_finalize / k_bn_grad_finalize: per-channel sums -> mean, 1/sigma, scale, shift, moving
//                            averages (Keras: momentum 0.99, epsilon 1e-3, unbiased variance into the average) and the three
//                            backward coefficients, in float64 (one launch instead of ~15 tiny tensor expressions per layer)
//   stem (alpha_nnet.py:21)  weight gradient of the 3 -> 128 convolution on v_mfma_f32_32x32x2_f32: GEMM rows = the 27 patch
//                            elements, columns = 128 outputs, reduction over pixels, two pixels per MFMA
//   head (alpha_nnet.py:49-54) 1x1 convolution 128 -> 1 with its batch-norm sums; Flatten -> Dense(128) -> Dense(3) -> tanh
//                            with the squared-error sum, forward and backward (16 states per block, the 226 KB Dense kernel
//                            read once per 16 states); the Dense kernel's gradient; the 1-channel batch-norm backward fused
//                            with the rank-1 expansion dz x w1x1 into the last tower activation's gradient
//   optimizer                Keras' Adam (epsilon added to sqrt(v) uncorrected) + the l2(1e-5) kernel regularizer's gradient
//                            over ONE flat parameter / gradient / moment buffer; sum of squares of the regularized kernels
//
// Reductions are two-stage and deterministic (block partials, then one block adding them in float64 in a fixed order).
#include "common.h"
#include "train_fold.h"

#define TN_WAVE_SYNC()                                                                          \
    do {                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                  \
        __builtin_amdgcn_wave_barrier();                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                  \
    } while (0)

#define TN_C 128
#define TN_THREADS 256

typedef float tn_f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------
// batch norm: sums of (y - center) and (y - center)^2 per channel (center = the layer's moving mean: the variance is then
// a difference of SMALL numbers even for channels whose mean is many standard deviations from zero), float64 results
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TN_THREADS) void k_bn_sums_c(const float *__restrict__ y, long rows, const float *__restrict__ center,
                                                         float *__restrict__ part)
{
    __shared__ float4 sh[2][8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float4 cen = make_float4(0.f, 0.f, 0.f, 0.f);
    if (center) cen = *(const float4 *)(center + 4 * cq);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    for (long r = (long)blockIdx.x * 8 + rl; r < rows; r += (long)gridDim.x * 8) {
        float4 v = *(const float4 *)(y + r * TN_C + 4 * cq);
        v.x -= cen.x; v.y -= cen.y; v.z -= cen.z; v.w -= cen.w;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
    sh[0][rl][cq] = s;
    sh[1][rl][cq] = q;
    __syncthreads();
    if (rl < 2) {
        float4 t = sh[rl][0][cq];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sh[rl][r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *(float4 *)(part + (size_t)blockIdx.x * (2 * TN_C) + rl * TN_C + 4 * cq) = t;
    }
}

// sums[0..C-1] = sum (y - center), sums[C..2C-1] = sum (y - center)^2 over `count` values per channel (all ranks)
// `center` and `mov_mean` are NOT restrict-qualified: train_step.py passes the moving mean for both (the sums are taken about it
// and it is updated in place); every thread reads its centre before it writes its moving mean
__global__ void k_bn_finalize(const double *__restrict__ sums, double count, const float *center,
                              const float *__restrict__ gamma, const float *__restrict__ beta, float *mov_mean,
                              float *mov_var, double momentum, double eps, float *__restrict__ mean_out,
                              float *__restrict__ inv_out, float *__restrict__ scale_out, float *__restrict__ shift_out, int C,
                              const float *__restrict__ amax = nullptr, float *__restrict__ range_tail = nullptr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float bound[128];
    if (range_tail) {                                                    // (one block of 128 threads: the launcher checks C <= 128)
        bound[threadIdx.x] = 0.f;
        __syncthreads();
    }
    if (c < C && range_tail) {
        // what relu(y * scale + shift) can reach when |y - center| <= amax: the larger end of the interval's image, a hair above
        // it for the float32 roundings of k_bn_apply's expression (the bound only has to hold; it picks a power of two)
        const double cen = center ? (double)center[c] : 0.0;
        const double m0 = sums[c] / count;
        double var = sums[C + c] / count - m0 * m0;
        if (var < 0.0) var = 0.0;
        const double mean = cen + m0, inv = 1.0 / sqrt(var + eps);
        const double sc = (double)(float)((double)gamma[c] * inv), sh = (double)(float)((double)beta[c] - mean * ((double)gamma[c] * inv));
        const double hi = sc * (cen + (double)amax[c]) + sh, lo = sc * (cen - (double)amax[c]) + sh;
        bound[threadIdx.x] = (float)(fmax(fmax(hi, lo), 0.0) * (1.0 + 1e-5));
    }
    if (range_tail) {
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = 0.f;
            for (int i = 0; i < 128; ++i) mx = fmaxf(mx, bound[i]);
            int k = 0;
            if (mx > 0.f && mx < 3.0e38f) k = 11 - ilogbf(mx);           // 2^11 <= mx * 2^k < 2^12 (k_amax_scale, csrc/train.hip)
            k = max(-100, min(100, k));
            range_tail[2] = ldexpf(1.0f, k);
            range_tail[3] = ldexpf(1.0f, -k);
        }
    }
    if (c >= C) return;
    const double cen = center ? (double)center[c] : 0.0;
    const double m0 = sums[c] / count;                                   // mean of (y - center)
    double var = sums[C + c] / count - m0 * m0;                          // biased batch variance
    if (var < 0.0) var = 0.0;
    const double mean = cen + m0, inv = 1.0 / sqrt(var + eps);
    const double sc = (double)gamma[c] * inv;
    mean_out[c] = (float)mean;
    inv_out[c] = (float)inv;
    scale_out[c] = (float)sc;
    shift_out[c] = (float)((double)beta[c] - mean * sc);
    if (mov_mean) {                                                      // after `center` was read: it may BE mov_mean
        const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
        mov_mean[c] = (float)((double)mov_mean[c] * momentum + mean * (1.0 - momentum));
        mov_var[c] = (float)((double)mov_var[c] * momentum + unbiased * (1.0 - momentum));
    }
}

// backward coefficients from {sum g, sum g xhat}: a = gamma inv, b = sum g / count, c = sum g xhat / count (global sums);
// the parameter gradients are this rank's LOCAL sums (the gradient bucket is all-reduced later): dbeta = sum g, dgamma = sum g xhat
__global__ void k_bn_grad_finalize(const double *__restrict__ sums_global, const double *__restrict__ sums_local, double count,
                                   const float *__restrict__ gamma, const float *__restrict__ inv, float *__restrict__ a,
                                   float *__restrict__ b, float *__restrict__ c_out, float *__restrict__ dgamma,
                                   float *__restrict__ dbeta, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    a[c] = gamma[c] * inv[c];
    b[c] = (float)(sums_global[c] / count);
    c_out[c] = (float)(sums_global[C + c] / count);
    dbeta[c] = (float)sums_local[c];
    dgamma[c] = (float)sums_local[C + c];
}

#define TN_BN_PART (2048 * 2 * TN_C)                       // floats of block partials in a snk_bn_train_partials() buffer; fold scratch behind
static int tn_grid(long rows)
{
    const long want = (rows + 7) / 8;
    return (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);
}

extern "C" int snk_bn_train_sums_f64(const float *d_y, long rows, const float *d_center, float *d_partials, double *d_sums,
                                     void *stream)
{
    SNK_REQUIRE(d_y && d_partials && d_sums && rows > 0, "snk_bn_train_sums_f64: bad argument");
    const int grid = tn_grid(rows);
    k_bn_sums_c<<<grid, TN_THREADS, 0, (hipStream_t)stream>>>(d_y, rows, d_center, d_partials);
    tf_fold<double>(d_partials, grid, 2 * TN_C, 2 * TN_C, 1.0, d_sums, (double *)(d_partials + TN_BN_PART), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_bn_train_finalize(const double *d_sums, double count, const float *d_center, const float *d_gamma,
                                     const float *d_beta, float *d_moving_mean, float *d_moving_var, double momentum, double eps,
                                     float *d_mean, float *d_inv, float *d_scale, float *d_shift, int channels, void *stream)
{
    SNK_REQUIRE(d_sums && d_gamma && d_beta && d_mean && d_inv && d_scale && d_shift && channels > 0 && count > 0 &&
                (!d_moving_mean == !d_moving_var), "snk_bn_train_finalize: bad argument");
    k_bn_finalize<<<(channels + 127) / 128, 128, 0, (hipStream_t)stream>>>(d_sums, count, d_center, d_gamma, d_beta, d_moving_mean,
                                                                          d_moving_var, momentum, eps, d_mean, d_inv, d_scale,
                                                                          d_shift, channels);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_bn_train_finalize for a layer whose batch norm + ReLU output is never written (deferred into the next convolution's staging):
// d_amax = the largest |y - center| per channel (snk_conv3x3_f16s_stats_deferred), d_out_scale_tail = { ., ., scale, 1 / scale } as
// snk_bn_train_apply writes it for the output it measured -- here from the bound max_c relu(scale_c (center_c +- amax_c) + shift_c),
// which the output cannot exceed (a looser bound than the measured maximum costs nothing but exponent headroom of the lo parts)
extern "C" int snk_bn_train_finalize_range(const double *d_sums, double count, const float *d_center, const float *d_gamma,
                                           const float *d_beta, float *d_moving_mean, float *d_moving_var, double momentum, double eps,
                                           float *d_mean, float *d_inv, float *d_scale, float *d_shift, const float *d_amax,
                                           float *d_out_scale_tail, int channels, void *stream)
{
    SNK_REQUIRE(d_sums && d_gamma && d_beta && d_mean && d_inv && d_scale && d_shift && d_amax && d_out_scale_tail && channels > 0 &&
                channels <= 128 && count > 0 && (!d_moving_mean == !d_moving_var), "snk_bn_train_finalize_range: bad argument");
    k_bn_finalize<<<1, 128, 0, (hipStream_t)stream>>>(d_sums, count, d_center, d_gamma, d_beta, d_moving_mean, d_moving_var, momentum, eps,
                                                      d_mean, d_inv, d_scale, d_shift, channels, d_amax, d_out_scale_tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_bn_train_grad_finalize(const double *d_sums_global, const double *d_sums_local, double count,
                                          const float *d_gamma, const float *d_inv, float *d_a, float *d_b, float *d_c,
                                          float *d_dgamma, float *d_dbeta, int channels, void *stream)
{
    SNK_REQUIRE(d_sums_global && d_sums_local && d_gamma && d_inv && d_a && d_b && d_c && d_dgamma && d_dbeta && channels > 0 &&
                count > 0, "snk_bn_train_grad_finalize: bad argument");
    k_bn_grad_finalize<<<(channels + 127) / 128, 128, 0, (hipStream_t)stream>>>(d_sums_global, d_sums_local, count, d_gamma, d_inv,
                                                                               d_a, d_b, d_c, d_dgamma, d_dbeta, channels);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// stem weight gradient: dW[tap][ci][co] = sum over images and pixels of x[n][y + dy - 1][x + dx - 1][ci] * dY[n][y][x][co]
// v_mfma_f32_32x32x2_f32: A[m][k] (lane = 32 k + m) = patch element m = 3 tap + ci of pixel p + k, B[k][j] (lane = 32 k + j)
// = dY of pixel p + k; one float4 load per lane gives the B values of FOUR N tiles (tile t holds outputs 4 j + t).  A wave
// works through whole images (its own zero-bordered copy of the 3-channel image in LDS); float32 products, exact inputs.
// ---------------------------------------------------------------------------------------------------------------------
struct StemWgArgs {
    const float *x;      // [n][H][W][3]
    const float *dy;     // [n][H][W][128]
    float *part;         // [gridDim.x][32][128]
    int n, H, W;
};

__global__ __launch_bounds__(256) void k_stem_wgrad(StemWgArgs p)
{
    extern __shared__ __align__(16) float sw_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 31, kk = lane >> 5;
    const int H = p.H, W = p.W, HW = H * W, P3 = (W + 2) * 3, n_lds = (H + 2) * P3;
    float *img = sw_lds + wv * n_lds;
    for (int j = lane; j < n_lds; j += 64) img[j] = 0.f;               // the border stays zero
    TN_WAVE_SYNC();
    const int tap = m / 3, ci = m - 3 * tap;
    const int aoff = m < 27 ? ((tap / 3) * P3 + (tap % 3) * 3 + ci) : 0;
    const float amask = m < 27 ? 1.f : 0.f;
    tn_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int n = blockIdx.x * 4 + wv; n < p.n; n += gridDim.x * 4) {
        const float *src = p.x + (long)n * HW * 3;
        for (int j = lane; j < HW * 3; j += 64) {                        // wave-private image: LDS instructions of a wave run in order
            const int yy = j / (W * 3), rr = j - yy * (W * 3);
            img[(yy + 1) * P3 + 3 + rr] = src[j];
        }
        TN_WAVE_SYNC();
        const float *dyn = p.dy + (long)n * HW * TN_C + 4 * m;
        // four pixel pairs per trip, their dY rows requested together and no branch around a load: with `if (ok) b = ...` per pair every
        // pair waited for its own HBM round trip before its four MFMAs (0.20 ms for 0.07 ms of matrix work); a pair past the image
        // multiplies a clamped row by a = 0
        const float invW = 1.0f / (float)W;
        for (int p0 = 0; p0 < HW; p0 += 8) {
            float av[4];
            float4 bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = p0 + 2 * u + kk;                           // this lane's pixel of the pair
                const int qc = min(q, HW - 1);
                const int y = (int)(((float)qc + 0.5f) * invW), x = qc - y * W;
                av[u] = q < HW ? img[y * P3 + x * 3 + aoff] * amask : 0.f;
                bv[u] = *(const float4 *)(dyn + (long)qc * TN_C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u].w, acc[3], 0, 0, 0);
            }
        }
        TN_WAVE_SYNC();                                                  // the next image overwrites what this one's reads used
    }
    // the four waves' accumulators are added through LDS; C layout: register r = row (r & 3) + 8 (r >> 2) + 4 kk, column m
    __syncthreads();
    float *red = sw_lds;                                                 // [4][32][128]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * TN_C + 4 * m + t] = acc[t][r];
    __syncthreads();
    for (int i = tid; i < 32 * TN_C; i += 256)
        p.part[(long)blockIdx.x * (32 * TN_C) + i] = ((red[i] + red[32 * TN_C + i]) + red[2 * 32 * TN_C + i]) + red[3 * 32 * TN_C + i];
}

extern "C" long snk_stem_wgrad_partials(int n_images, int height, int width)
{
    const long lds = (long)4 * (height + 2) * (width + 2) * 3 * sizeof(float);
    if (height < 1 || width < 1 || lds > 150 * 1024) return -1;
    const int grid = n_images < 4 ? 1 : (n_images / 4 < 512 ? n_images / 4 : 512);
    return (long)grid * 32 * TN_C + TF_SCRATCH_FLOATS(27 * TN_C);
}

extern "C" int snk_stem_wgrad_f32(const float *d_x, const float *d_dy, float *d_partials, float *d_dw, int n_images, int height,
                                  int width, void *stream)
{
    SNK_REQUIRE(d_x && d_dy && d_partials && d_dw && n_images > 0, "snk_stem_wgrad_f32: bad argument");
    const long need = snk_stem_wgrad_partials(n_images, height, width);
    SNK_REQUIRE(need > 0, "snk_stem_wgrad_f32: %d x %d image does not fit the LDS", height, width);
    const int grid = (int)((need - TF_SCRATCH_FLOATS(27 * TN_C)) / (32 * TN_C));
    size_t lds = (size_t)4 * (height + 2) * (width + 2) * 3 * sizeof(float);
    if (lds < (size_t)4 * 32 * TN_C * sizeof(float)) lds = (size_t)4 * 32 * TN_C * sizeof(float);
    StemWgArgs a = {d_x, d_dy, d_partials, n_images, height, width};
    k_stem_wgrad<<<grid, 256, lds, (hipStream_t)stream>>>(a);
    // rows 0..26 of the 32-row tile are the Keras kernel (kh, kw, cin, cout) flattened: 27 x 128
    tf_fold<float>(d_partials, grid, 32 * TN_C, 27 * TN_C, 1.0, d_dw, (double *)(d_partials + (long)grid * 32 * TN_C), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// head, forward.  k_head1x1: z[r] = dot(a[r][:], w1x1) for r = (state, pixel) + block partials of sum (z - center) and
// sum (z - center)^2 (the single channel's batch-norm statistics)
// ---------------------------------------------------------------------------------------------------------------------
#define tn_dpp(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
__device__ static inline float tn_sum32(float d)          // sum over the 32 lanes that share (lane >> 5): result in every lane
{
    d += tn_dpp(d, 0xB1); d += tn_dpp(d, 0x4E); d += tn_dpp(d, 0x141); d += tn_dpp(d, 0x140);
    d += __shfl_xor(d, 16, 64);
    return d;
}

__device__ static inline void tn_block_sum2(float s, float q, float *__restrict__ dst)      // dst[0] = sum s, dst[1] = sum q over the block
{
    __shared__ float sh2[2][TN_THREADS];
    sh2[0][threadIdx.x] = s;
    sh2[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = TN_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh2[0][threadIdx.x] += sh2[0][threadIdx.x + o];
            sh2[1][threadIdx.x] += sh2[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dst[0] = sh2[0][0]; dst[1] = sh2[1][0]; }
}

__global__ __launch_bounds__(TN_THREADS) void k_head1x1(const float *__restrict__ a, const float *__restrict__ w1x1, long rows,
                                                       const float *__restrict__ center, float *__restrict__ z,
                                                       float *__restrict__ part)
{
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 w = *(const float4 *)(w1x1 + 4 * cq);
    const float cen = center ? center[0] : 0.f;
    float s = 0.f, q = 0.f;
    for (long r = (long)blockIdx.x * 8 + rl; r < rows; r += (long)gridDim.x * 8) {
        const float4 v = *(const float4 *)(a + r * TN_C + 4 * cq);
        const float d = tn_sum32((v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w));
        if (cq == 0) {
            z[r] = d;
            const float e = d - cen;
            s += e; q += e * e;
        }
    }
    tn_block_sum2(s, q, part + 2 * blockIdx.x);
}

extern "C" int snk_head_conv1x1_sums(const float *d_a, const float *d_w1x1, long rows, const float *d_center, float *d_z,
                                     float *d_partials, double *d_sums, void *stream)
{
    SNK_REQUIRE(d_a && d_w1x1 && d_z && d_partials && d_sums && rows > 0, "snk_head_conv1x1_sums: bad argument");
    const int grid = tn_grid(rows);
    k_head1x1<<<grid, TN_THREADS, 0, (hipStream_t)stream>>>(d_a, d_w1x1, rows, d_center, d_z, d_partials);
    tf_fold<double>(d_partials, grid, 2, 2, 1.0, d_sums, (double *)(d_partials + TN_BN_PART), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// Flatten -> Dense(128) + ReLU -> Dense(3) + tanh for 16 states per block, keeping what the backward pass needs:
// h = relu(z * s1 + b1) [n][HW], d1 [n][128], q [n][3]; block partial of sum (q - target)^2
#define TN_S 16
struct HeadTrainArgs {
    const float *z;          // [n][HW] 1x1 convolution output
    const float *sb;         // {scale, shift} of the single channel's batch norm (device memory)
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    const float *target;     // [n][3] or NULL
    float *h, *d1, *q;       // outputs (h, d1 optional)
    float *part;             // [blocks] squared-error partials (optional)
    int n, HW;
};

__global__ __launch_bounds__(256) void k_head_dense_train_fwd(HeadTrainArgs p)
{
    extern __shared__ float sm[];
    float *h1 = sm;                        // [TN_S][HW]
    float *h2 = sm + TN_S * p.HW;          // [TN_S][128]
    __shared__ float se[TN_S * 3];
    const int tid = threadIdx.x, j = tid & 127, half = tid >> 7;
    const int s0 = blockIdx.x * TN_S, ns = min(TN_S, p.n - s0);
    const float s1 = p.sb[0], b1 = p.sb[1];
    for (int i = tid; i < TN_S * p.HW; i += 256) {
        float v = 0.f;
        if (i < ns * p.HW) {
            v = fmaxf(p.z[(long)s0 * p.HW + i] * s1 + b1, 0.f);
            if (p.h) p.h[(long)s0 * p.HW + i] = v;
        }
        h1[i] = v;
    }
    __syncthreads();
    float acc[TN_S / 2];
#pragma unroll
    for (int k = 0; k < TN_S / 2; ++k) acc[k] = 0.f;
    const float *hh = h1 + half * (TN_S / 2) * p.HW;
    int i = 0;
    for (; i + 8 <= p.HW; i += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = p.fc1_w[(long)(i + u) * 128 + j];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < TN_S / 2; ++k) acc[k] = fmaf(hh[k * p.HW + i + u], w[u], acc[k]);
    }
    for (; i < p.HW; ++i) {
        const float w = p.fc1_w[(long)i * 128 + j];
#pragma unroll
        for (int k = 0; k < TN_S / 2; ++k) acc[k] = fmaf(hh[k * p.HW + i], w, acc[k]);
    }
    const float b = p.fc1_b[j];
#pragma unroll
    for (int k = 0; k < TN_S / 2; ++k) {
        const int sl = half * (TN_S / 2) + k;
        const float v = fmaxf(acc[k] + b, 0.f);
        h2[sl * 128 + j] = v;
        if (p.d1 && sl < ns) p.d1[(long)(s0 + sl) * 128 + j] = v;
    }
    __syncthreads();
    if (tid < TN_S * 3) {
        const int sl = tid / 3, o = tid - sl * 3;
        float e = 0.f;
        if (sl < ns) {
            float v = 0.f;
            for (int k = 0; k < 128; ++k) v = fmaf(h2[sl * 128 + k], p.fc2_w[k * 3 + o], v);
            const float qv = tanhf(v + p.fc2_b[o]);
            p.q[(long)(s0 + sl) * 3 + o] = qv;
            if (p.target) { e = qv - p.target[(long)(s0 + sl) * 3 + o]; e *= e; }
        }
        se[tid] = e;
    }
    __syncthreads();
    if (tid == 0 && p.part) {
        float t = 0.f;
        for (int k = 0; k < TN_S * 3; ++k) t += se[k];
        p.part[blockIdx.x] = t;
    }
}

extern "C" int snk_head_dense_train_fwd(const float *d_z, const float *d_scale_shift, const float *d_fc1_w, const float *d_fc1_b,
                                        const float *d_fc2_w, const float *d_fc2_b, const float *d_target, float *d_h, float *d_d1,
                                        float *d_q, float *d_partials, float *d_sq_err, double err_scale, int n_images, int height,
                                        int width, void *stream)
{
    SNK_REQUIRE(d_z && d_scale_shift && d_fc1_w && d_fc1_b && d_fc2_w && d_fc2_b && d_q && n_images > 0 &&
                (!d_sq_err || (d_partials && d_target)), "snk_head_dense_train_fwd: bad argument");
    const size_t lds = (size_t)TN_S * (height * width + 128) * sizeof(float);
    SNK_REQUIRE(lds <= 150 * 1024, "snk_head_dense_train_fwd: %d x %d observation too large", height, width);
    const int grid = (n_images + TN_S - 1) / TN_S;
    HeadTrainArgs a = {d_z, d_scale_shift, d_fc1_w, d_fc1_b, d_fc2_w, d_fc2_b, d_target, d_h, d_d1, d_q,
                       d_sq_err ? d_partials : nullptr, n_images, height * width};
    k_head_dense_train_fwd<<<grid, 256, lds, (hipStream_t)stream>>>(a);
    if (d_sq_err) tf_fold<float>(d_partials, grid, 1, 1, err_scale, d_sq_err, (double *)(d_partials + TN_BN_PART), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// head, backward.  k_head_dense_train_bwd (16 states per block): dq = 2 (q - target) * norm, through tanh and the two
// Dense layers down to g = dh masked by h > 0; writes dpre1 [n][128] (for the Dense(128) kernel gradient) and g [n][HW];
// block partials: dW2 [128][3], db2 [3], db1 [128], sum g, sum g zhat (zhat = (z - mean) inv of the 1-channel batch norm)
// ---------------------------------------------------------------------------------------------------------------------
#define TN_BWD_STRIDE 520
struct HeadBwdArgs {
    const float *q, *target, *d1, *d1_mask, *h_mask, *z;
    const float *mean_inv;   // {mean, inv} of the 1-channel batch norm
    const float *fc1_w, *fc2_w;
    float *dpre1, *g, *part;
    float norm;              // 1 / (3 * rows of the GLOBAL batch)
    int n, HW;
};

__global__ __launch_bounds__(256) void k_head_dense_train_bwd(HeadBwdArgs p)
{
    extern __shared__ float sm[];
    float *dp1 = sm;                       // [TN_S][128] dpre1
    float *d1s = dp1 + TN_S * 128;         // [TN_S][128] d1
    float *dp2 = d1s + TN_S * 128;         // [TN_S][4] dpre2
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * TN_S, ns = min(TN_S, p.n - s0);
    if (tid < TN_S * 3) {
        const int sl = tid / 3, o = tid - sl * 3;
        float v = 0.f;
        if (sl < ns) {
            const float qv = p.q[(long)(s0 + sl) * 3 + o];
            v = 2.f * (qv - p.target[(long)(s0 + sl) * 3 + o]) * p.norm * (1.f - qv * qv);
        }
        dp2[sl * 4 + o] = v;
    }
    for (int i = tid; i < TN_S * 128; i += 256) d1s[i] = i < ns * 128 ? p.d1[(long)s0 * 128 + i] : 0.f;
    __syncthreads();
    float *part = p.part + (long)blockIdx.x * TN_BWD_STRIDE;
    // dW2[k][o] = sum_s d1[s][k] dpre2[s][o]   (384 values), db2[o] (3 values)
    for (int i = tid; i < 384 + 3; i += 256) {
        float t = 0.f;
        if (i < 384) { const int k = i / 3, o = i - 3 * k; for (int s = 0; s < TN_S; ++s) t += d1s[s * 128 + k] * dp2[s * 4 + o]; }
        else { const int o = i - 384; for (int s = 0; s < TN_S; ++s) t += dp2[s * 4 + o]; }
        part[i] = t;
    }
    // dpre1[s][k] = (sum_o dpre2[s][o] W2[k][o]) masked by d1 > 0
    for (int i = tid; i < TN_S * 128; i += 256) {
        const int sl = i >> 7, k = i & 127;
        float v = 0.f;
        if (sl < ns) {
            v = dp2[sl * 4] * p.fc2_w[k * 3] + dp2[sl * 4 + 1] * p.fc2_w[k * 3 + 1] + dp2[sl * 4 + 2] * p.fc2_w[k * 3 + 2];
            if (!(p.d1_mask[(long)(s0 + sl) * 128 + k] > 0.f)) v = 0.f;
            p.dpre1[(long)(s0 + sl) * 128 + k] = v;
        }
        dp1[i] = v;
    }
    __syncthreads();
    if (tid < 128) {                                                     // db1[k] = sum_s dpre1[s][k]
        float t = 0.f;
        for (int s = 0; s < TN_S; ++s) t += dp1[s * 128 + tid];
        part[387 + tid] = t;
    }
    // dh[s][px] = sum_k dpre1[s][k] W1[px][k]; g = dh masked by h > 0
    const float mean = p.mean_inv[0], inv = p.mean_inv[1];
    float sg = 0.f, sgz = 0.f;
    for (int px = tid; px < p.HW; px += 256) {
        float acc[TN_S];
#pragma unroll
        for (int s = 0; s < TN_S; ++s) acc[s] = 0.f;
        const float4 *wr = (const float4 *)(p.fc1_w + (long)px * 128);
        for (int k4 = 0; k4 < 32; ++k4) {
            const float4 w = wr[k4];
#pragma unroll
            for (int s = 0; s < TN_S; ++s) {
                const float4 d = *(const float4 *)(dp1 + s * 128 + 4 * k4);
                acc[s] = fmaf(d.x, w.x, fmaf(d.y, w.y, fmaf(d.z, w.z, fmaf(d.w, w.w, acc[s]))));
            }
        }
#pragma unroll
        for (int s = 0; s < TN_S; ++s) {
            if (s < ns) {
                const long o = (long)(s0 + s) * p.HW + px;
                const float gv = p.h_mask[o] > 0.f ? acc[s] : 0.f;
                p.g[o] = gv;
                sg += gv;
                sgz += gv * ((p.z[o] - mean) * inv);
            }
        }
    }
    tn_block_sum2(sg, sgz, part + 515);
}

// Dense(128) kernel gradient dW1[px][j] = sum_n h[n][px] dpre1[n][j]: block = two flattened pixels x 128 outputs, thread
// (j, half) runs over half of the states
__global__ __launch_bounds__(256) void k_head_dw1(const float *__restrict__ h, const float *__restrict__ dpre1, float *__restrict__ dw1,
                                                  int n, int HW)
{
    __shared__ float sh[2][256];
    const int tid = threadIdx.x, j = tid & 127, half = tid >> 7;
    const int px0 = 2 * blockIdx.x, px1 = min(px0 + 1, HW - 1);
    const int lo = half ? n / 2 : 0, hi = half ? n : n / 2;
    float a0 = 0.f, a1 = 0.f;
    int s = lo;
    for (; s + 4 <= hi; s += 4) {
        float d[4], u[4], v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] = dpre1[(long)(s + e) * 128 + j]; u[e] = h[(long)(s + e) * HW + px0]; v[e] = h[(long)(s + e) * HW + px1]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0 = fmaf(u[e], d[e], a0); a1 = fmaf(v[e], d[e], a1); }
    }
    for (; s < hi; ++s) {
        const float d = dpre1[(long)s * 128 + j];
        a0 = fmaf(h[(long)s * HW + px0], d, a0);
        a1 = fmaf(h[(long)s * HW + px1], d, a1);
    }
    sh[0][tid] = a0;
    sh[1][tid] = a1;
    __syncthreads();
    if (tid < 128) {
        dw1[(long)px0 * 128 + tid] = sh[0][tid] + sh[0][tid + 128];
        if (px0 + 1 < HW) dw1[(long)(px0 + 1) * 128 + tid] = sh[1][tid] + sh[1][tid + 128];
    }
}

extern "C" int snk_head_dense_train_bwd_partials(int n_images)
{
    return ((n_images + TN_S - 1) / TN_S) * TN_BWD_STRIDE + TF_SCRATCH_FLOATS(TN_BWD_STRIDE);
}

// outputs: d_dpre1 [n][128] (scratch), d_g [n][HW], d_dw1 [HW][128], d_small[517] = dW2 [128][3], db2 [3], db1 [128],
// then as float64 d_gsums[2] = {sum g, sum g zhat}
extern "C" int snk_head_dense_train_bwd(const float *d_q, const float *d_target, const float *d_h, const float *d_d1,
                                        const float *d_h_mask, const float *d_d1_mask, const float *d_z, const float *d_mean_inv,
                                        const float *d_fc1_w, const float *d_fc2_w, double norm, float *d_dpre1, float *d_g,
                                        float *d_dw1, float *d_small, double *d_gsums, float *d_partials, int n_images, int height,
                                        int width, void *stream)
{
    SNK_REQUIRE(d_q && d_target && d_h && d_d1 && d_z && d_mean_inv && d_fc1_w && d_fc2_w && d_dpre1 && d_g && d_dw1 && d_small &&
                d_gsums && d_partials && n_images > 0, "snk_head_dense_train_bwd: bad argument");
    const int HW = height * width, grid = (n_images + TN_S - 1) / TN_S;
    HeadBwdArgs a = {d_q, d_target, d_d1, d_d1_mask ? d_d1_mask : d_d1, d_h_mask ? d_h_mask : d_h, d_z, d_mean_inv, d_fc1_w, d_fc2_w,
                     d_dpre1, d_g, d_partials, (float)norm, n_images, HW};
    const size_t lds = (size_t)(2 * TN_S * 128 + TN_S * 4) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    k_head_dense_train_bwd<<<grid, 256, lds, st>>>(a);
    double *scratch = (double *)(d_partials + (long)grid * TN_BWD_STRIDE);
    tf_fold<float>(d_partials, grid, TN_BWD_STRIDE, 515, 1.0, d_small, scratch, st);
    tf_fold<double>(d_partials + 515, grid, TN_BWD_STRIDE, 2, 1.0, d_gsums, scratch, st);
    k_head_dw1<<<(HW + 1) / 2, 256, 0, st>>>(d_h, d_dpre1, d_dw1, n_images, HW);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// 1-channel batch-norm backward fused with the rank-1 expansion into the last tower activation's gradient:
// dz[r] = a (g[r] - b - zhat[r] c);  da[r][c] = dz[r] w1x1[c];  dw1x1[c] = sum_r a_last[r][c] dz[r]  (block partials)
// STATS: the gradient it writes is the one the LAST tower layer's batch-norm backward starts from -- its two sums, sum(g') and
//   sum(g' xhat) with g' = da where that layer's ReLU bit is set (k_bn_grad_sums, csrc/train.hip), are taken from the values on their
//   way out (the layer's pre-batch-norm output and mask bytes are read here instead; the pass over da + y + mask is not needed)
template <bool STATS>
__global__ __launch_bounds__(TN_THREADS) void k_head_expand(const float *__restrict__ g, const float *__restrict__ z,
                                                           const float *__restrict__ mean_inv, const float *__restrict__ abc,
                                                           const float *__restrict__ a_last, const float *__restrict__ w1x1,
                                                           float *__restrict__ da, float *__restrict__ part, long rows,
                                                           const float *__restrict__ y_last = nullptr, const uint8_t *__restrict__ mask_last = nullptr,
                                                           const float *__restrict__ mean_last = nullptr, const float *__restrict__ inv_last = nullptr,
                                                           float *__restrict__ stat_part = nullptr)
{
    __shared__ float4 sh[STATS ? 24 : 8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 w = *(const float4 *)(w1x1 + 4 * cq);
    const float mean = mean_inv[0], inv = mean_inv[1], A = abc[0], B = abc[1], Cc = abc[2];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ts = s, tq = s, mu = s, iv = s;
    if (STATS) { mu = *(const float4 *)(mean_last + 4 * cq); iv = *(const float4 *)(inv_last + 4 * cq); }
    for (long r = (long)blockIdx.x * 8 + rl; r < rows; r += (long)gridDim.x * 8) {
        const float dz = A * (g[r] - B - ((z[r] - mean) * inv) * Cc);
        const float4 v = *(const float4 *)(a_last + r * TN_C + 4 * cq);
        s.x += v.x * dz; s.y += v.y * dz; s.z += v.z * dz; s.w += v.w * dz;
        const float4 d = make_float4(dz * w.x, dz * w.y, dz * w.z, dz * w.w);
        *(float4 *)(da + r * TN_C + 4 * cq) = d;
        if (STATS) {
            const unsigned m_ = mask_last[r * 32 + cq];
            const float4 yv = *(const float4 *)(y_last + r * TN_C + 4 * cq);
            const float gx = (m_ & 1u) ? d.x : 0.f, gy = (m_ & 2u) ? d.y : 0.f, gz = (m_ & 4u) ? d.z : 0.f, gw = (m_ & 8u) ? d.w : 0.f;
            ts.x += gx; ts.y += gy; ts.z += gz; ts.w += gw;
            tq.x += gx * ((yv.x - mu.x) * iv.x); tq.y += gy * ((yv.y - mu.y) * iv.y);
            tq.z += gz * ((yv.z - mu.z) * iv.z); tq.w += gw * ((yv.w - mu.w) * iv.w);
        }
    }
    sh[rl][cq] = s;
    if (STATS) { sh[8 + rl][cq] = ts; sh[16 + rl][cq] = tq; }
    __syncthreads();
    if (STATS && rl >= 1 && rl <= 2) {
        float4 t = sh[8 * rl][cq];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sh[8 * rl + r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *(float4 *)(stat_part + (size_t)blockIdx.x * (2 * TN_C) + (rl - 1) * TN_C + 4 * cq) = t;
    }
    if (rl == 0) {
        float4 t = sh[0][cq];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sh[r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *(float4 *)(part + (size_t)blockIdx.x * TN_C + 4 * cq) = t;
    }
}

extern "C" int snk_head_conv1x1_bwd(const float *d_g, const float *d_z, const float *d_mean_inv, const float *d_abc,
                                    const float *d_a_last, const float *d_w1x1, float *d_da, float *d_dw1x1, float *d_partials,
                                    long rows, void *stream)
{
    SNK_REQUIRE(d_g && d_z && d_mean_inv && d_abc && d_a_last && d_w1x1 && d_da && d_dw1x1 && d_partials && rows > 0,
                "snk_head_conv1x1_bwd: bad argument");
    const int grid = tn_grid(rows);
    k_head_expand<false><<<grid, TN_THREADS, 0, (hipStream_t)stream>>>(d_g, d_z, d_mean_inv, d_abc, d_a_last, d_w1x1, d_da, d_partials, rows);
    tf_fold<float>(d_partials, grid, TN_C, TN_C, 1.0, d_dw1x1, (double *)(d_partials + TN_BN_PART), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_head_conv1x1_bwd that also leaves the two sums the last tower layer's batch-norm backward starts from: d_sums[0..127] = sum(g'),
// d_sums[128..255] = sum(g' (y - mean) inv), g' = d_da where d_mask_last's bit is set (snk_bn_train_grad_sums_f64(d_da, ..) without its
// pass).  d_stat_partials: a second buffer of snk_bn_train_partials() floats.
extern "C" int snk_head_conv1x1_bwd_stats(const float *d_g, const float *d_z, const float *d_mean_inv, const float *d_abc,
                                          const float *d_a_last, const float *d_w1x1, float *d_da, float *d_dw1x1, float *d_partials,
                                          const float *d_y_last, const uint8_t *d_mask_last, const float *d_mean_last,
                                          const float *d_inv_last, float *d_stat_partials, double *d_sums, long rows, void *stream)
{
    SNK_REQUIRE(d_g && d_z && d_mean_inv && d_abc && d_a_last && d_w1x1 && d_da && d_dw1x1 && d_partials && d_y_last && d_mask_last &&
                d_mean_last && d_inv_last && d_stat_partials && d_sums && rows > 0, "snk_head_conv1x1_bwd_stats: bad argument");
    const int grid = tn_grid(rows);
    k_head_expand<true><<<grid, TN_THREADS, 0, (hipStream_t)stream>>>(d_g, d_z, d_mean_inv, d_abc, d_a_last, d_w1x1, d_da, d_partials, rows,
                                                                       d_y_last, d_mask_last, d_mean_last, d_inv_last, d_stat_partials);
    tf_fold<float>(d_partials, grid, TN_C, TN_C, 1.0, d_dw1x1, (double *)(d_partials + TN_BN_PART), (hipStream_t)stream);
    tf_fold<double>(d_stat_partials, grid, 2 * TN_C, 2 * TN_C, 1.0, d_sums, (double *)(d_stat_partials + TN_BN_PART), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// optimizer: tf.keras Adam on one flat buffer.  decay[i] != 0 marks the elements of Conv2D / Dense kernels: their gradient
// gets the l2(c) regularizer's 2 c w (alpha_nnet.py:15, 21 ...).  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) comes from the host.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TN_THREADS) void k_adam(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, const uint8_t *__restrict__ decay, long n, float lr_t,
                                                    float b1, float b2, float omb1, float omb2, float eps, float l2c2)
{
    const long i = (long)blockIdx.x * TN_THREADS + threadIdx.x;
    if (i >= n) return;
    const float wi = w[i];
    const float gi = g[i] + (decay[i] ? l2c2 * wi : 0.f);
    const float mi = m[i] * b1 + gi * omb1;                 // omb = 1 - beta, rounded once from float64 (1.f - 0.999f is 1.3e-5 off)
    const float vi = v[i] * b2 + (gi * gi) * omb2;
    m[i] = mi;
    v[i] = vi;
    w[i] = wi - lr_t * (mi / (sqrtf(vi) + eps));
}

__global__ __launch_bounds__(TN_THREADS) void k_sumsq(const float *__restrict__ w, const uint8_t *__restrict__ decay, long n,
                                                     float *__restrict__ part)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * TN_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * TN_THREADS)
        if (decay[i]) s += w[i] * w[i];
    tn_block_sum2(s, 0.f, part + 2 * blockIdx.x);
}

extern "C" int snk_adam_l2_step(float *d_w, const float *d_g, float *d_m, float *d_v, const uint8_t *d_decay, long n, double lr_t,
                                double beta1, double beta2, double epsilon, double l2, void *stream)
{
    SNK_REQUIRE(d_w && d_g && d_m && d_v && d_decay && n > 0, "snk_adam_l2_step: bad argument");
    k_adam<<<(int)((n + TN_THREADS - 1) / TN_THREADS), TN_THREADS, 0, (hipStream_t)stream>>>(d_w, d_g, d_m, d_v, d_decay, n, (float)lr_t,
                                                                                            (float)beta1, (float)beta2, (float)(1.0 - beta1),
                                                                                            (float)(1.0 - beta2), (float)epsilon, (float)(2.0 * l2));
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// d_out[0] = scale * sum of w[i]^2 over the elements with decay[i] != 0 (the regularization loss: scale = the l2 constant)
extern "C" int snk_l2_sum(const float *d_w, const uint8_t *d_decay, long n, double scale, float *d_partials, float *d_out, void *stream)
{
    SNK_REQUIRE(d_w && d_decay && d_partials && d_out && n > 0, "snk_l2_sum: bad argument");
    const int grid = 256;
    k_sumsq<<<grid, TN_THREADS, 0, (hipStream_t)stream>>>(d_w, d_decay, n, d_partials);
    tf_fold<float>(d_partials, grid, 2, 1, scale, d_out, (double *)(d_partials + 512), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
