// train.hip -- the memory-bound half of the training step (AlphaNNet.train, alpha_nnet.py:58-59):
// training-mode batch normalisation of a 128-channel activation (the BatchNormalization layers of alpha_nnet.py:23-46 under
// model.fit), fused with the ReLU and the residual add around it, forward and backward.  Activations are channels-last
// float32 [rows = n * h * w][128]; a thread owns four channels (one float4 column) of every eighth row of its block's share.
//
//   forward   k_bn_sums_c         per-channel sums about the moving mean (train_net.hip)           (1 read)
//             k_bn_apply          out = act(y * scale + shift (+ residual)), + 1 bit per element: out > 0   (1-2 reads, 1 write)
//   backward  k_bn_grad_sums      per-channel sum(g), sum(g * xhat), g = dout masked by the ReLU   (2 reads + the bits)
//             k_bn_grad_apply     dx = a (g - b - xhat c)  (+ g itself for the residual branch)    (2 reads + the bits, 1-2 writes)
// The ReLU mask travels as one BYTE per thread-quad of channels (4 bits used: 29 MB per layer at 2 048 x 21 x 21 instead of
// re-reading the 462 MB activation in both backward kernels); a caller may still hand the activation itself (or any tensor
// whose sign is the mask) instead.
//
// The reductions are two-stage and deterministic: every block writes its 256 partial sums, one block adds them in
// float64 in a fixed order (a run repeats bit for bit; ranks all-reduce the 256 numbers between the two kernels of each
// direction, snake_engine/train_step.py).  Everything here is HBM-bound: the PyTorch expressions these kernels replace made ten to
// fifteen passes over the activation per layer and direction.
#include "common.h"
#include "train_fold.h"

#define TR_C 128
#define TR_THREADS 256
#define TR_ROWLANES (TR_THREADS / 32)

__device__ static inline void tr_block_reduce_store(float4 a, float4 b, float *__restrict__ part)
{
    __shared__ float4 sh[2][TR_ROWLANES][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    sh[0][rl][cq] = a;
    sh[1][rl][cq] = b;
    __syncthreads();
    if (rl < 2) {                                   // row lane 0 finishes the first quantity, row lane 1 the second
        float4 s = sh[rl][0][cq];
#pragma unroll
        for (int r = 1; r < TR_ROWLANES; ++r) {
            const float4 v = sh[rl][r][cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *(float4 *)(part + (size_t)blockIdx.x * (2 * TR_C) + rl * TR_C + 4 * cq) = s;
    }
}

// largest magnitude of a float32 array -> the power-of-two input scale of the split-f16 convolution, written into the
// tail of its weight image { 2^-k, 2^k, x_scale, 1 / x_scale, flag } (conv_split.hip): 2^11 <= max * x_scale < 2^12
__global__ __launch_bounds__(TR_THREADS) void k_amax_part(const float *__restrict__ x, long n4, float *__restrict__ part)
{
    float m = 0.f;
    for (long i = (long)blockIdx.x * TR_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * TR_THREADS) {
        const float4 v = ((const float4 *)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    __shared__ float sh[TR_THREADS];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = TR_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(1024) void k_amax_scale(const float *__restrict__ part, int n_blocks, float *__restrict__ tail)
{
    __shared__ float sh[1024];
    float m = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 1024) m = fmaxf(m, part[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = sh[0];
        int k = 0;
        if (mx > 0.f && mx < 3.0e38f) k = 11 - ilogbf(mx);          // 2^11 <= mx * 2^k < 2^12
        k = max(-100, min(100, k));
        tail[2] = ldexpf(1.0f, k);
        tail[3] = ldexpf(1.0f, -k);
    }
}

// optional by-product of the two element-wise kernels: the largest magnitude they WRITE (the next convolution's input), one
// value per block into amax_part -- k_amax_scale turns them into that convolution's power-of-two input scale, so the tensor
// is not read once more just for its maximum
__device__ static inline void tr_block_amax(float m, float *__restrict__ amax_part)
{
    __shared__ float shm[TR_THREADS];
    shm[threadIdx.x] = m;
    __syncthreads();
    for (int s_ = TR_THREADS / 2; s_ > 0; s_ >>= 1) {
        if ((int)threadIdx.x < s_) shm[threadIdx.x] = fmaxf(shm[threadIdx.x], shm[threadIdx.x + s_]);
        __syncthreads();
    }
    if (threadIdx.x == 0) amax_part[blockIdx.x] = shm[0];
}

// HEAD (the tower's last layer, alpha_nnet.py:46-50): the head's 1x1 convolution rides on the values being written -- z[r] = dot(out[r][:],
// w1x1), summed over the 32 lanes that hold a row, and the block partials of sum (z - center), sum (z - center)^2 (the single
// channel's batch-norm statistics): snk_head_conv1x1_sums's pass over the 462 MB activation is not needed
#define tr_dpp(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
template <bool HEAD>
__global__ __launch_bounds__(TR_THREADS) void k_bn_apply(const float *__restrict__ y, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, const float *__restrict__ res,
                                                        float *__restrict__ out, long rows, int relu,
                                                        float *__restrict__ amax_part, uint8_t *__restrict__ mask_out,
                                                        const float *__restrict__ w1x1 = nullptr, const float *__restrict__ center1 = nullptr,
                                                        float *__restrict__ z = nullptr, float *__restrict__ part1 = nullptr,
                                                        const float *__restrict__ rsc = nullptr, const float *__restrict__ rsh = nullptr)
{
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 sc = *(const float4 *)(scale + 4 * cq), sh = *(const float4 *)(shift + 4 * cq);
    float4 rs4 = sc, rh4 = sc;            // rsc: the shortcut is itself a deferred batch norm + ReLU output: res holds its PRE-batch-norm values
    if (rsc) { rs4 = *(const float4 *)(rsc + 4 * cq); rh4 = *(const float4 *)(rsh + 4 * cq); }
    float4 w1 = sc;
    float cen1 = 0.f, hs = 0.f, hq = 0.f;
    if (HEAD) { w1 = *(const float4 *)(w1x1 + 4 * cq); if (center1) cen1 = center1[0]; }
    float am = 0.f;
    for (long r = (long)blockIdx.x * TR_ROWLANES + rl; r < rows; r += (long)gridDim.x * TR_ROWLANES) {
        const long o = r * TR_C + 4 * cq;
        const float4 v = *(const float4 *)(y + o);
        float4 t = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
        if (res) {
            float4 a = *(const float4 *)(res + o);
            if (rsc) a = make_float4(fmaxf(a.x * rs4.x + rh4.x, 0.f), fmaxf(a.y * rs4.y + rh4.y, 0.f), fmaxf(a.z * rs4.z + rh4.z, 0.f), fmaxf(a.w * rs4.w + rh4.w, 0.f));
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        }
        if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        *(float4 *)(out + o) = t;
        if (mask_out) mask_out[r * 32 + cq] = (uint8_t)((t.x > 0.f ? 1 : 0) | (t.y > 0.f ? 2 : 0) | (t.z > 0.f ? 4 : 0) | (t.w > 0.f ? 8 : 0));
        am = fmaxf(fmaxf(am, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
        if (HEAD) {                 // (k_head1x1's arithmetic, csrc/train_net.hip: the same sums in the same order)
            float d = (t.x * w1.x + t.y * w1.y) + (t.z * w1.z + t.w * w1.w);
            d += tr_dpp(d, 0xB1); d += tr_dpp(d, 0x4E); d += tr_dpp(d, 0x141); d += tr_dpp(d, 0x140);
            d += __shfl_xor(d, 16, 64);
            if (cq == 0) {
                z[r] = d;
                const float e = d - cen1;
                hs += e; hq += e * e;
            }
        }
    }
    if (amax_part) tr_block_amax(am, amax_part);
    if (HEAD) {
        __shared__ float sh2[2][TR_THREADS];
        sh2[0][threadIdx.x] = hs;
        sh2[1][threadIdx.x] = hq;
        __syncthreads();
        for (int o = TR_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { sh2[0][threadIdx.x] += sh2[0][threadIdx.x + o]; sh2[1][threadIdx.x] += sh2[1][threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { part1[2 * blockIdx.x] = sh2[0][0]; part1[2 * blockIdx.x + 1] = sh2[1][0]; }
    }
}

// g = dout where the layer's output is positive (ReLU) or everywhere (no ReLU); xhat = (y - mean) * inv
#define TR_G(dv, ov) (relu ? make_float4(ov.x > 0.f ? dv.x : 0.f, ov.y > 0.f ? dv.y : 0.f, ov.z > 0.f ? dv.z : 0.f, ov.w > 0.f ? dv.w : 0.f) : dv)
// the ReLU decision of a quad of channels: from the mask byte k_bn_apply wrote, from the sign of a tensor, or (msc: the layer's
// output was never written, its batch norm + ReLU being deferred into the next convolution's staging) recomputed from the
// pre-batch-norm value yv with k_bn_apply's own expression
#define TR_MASKED(dv, o, yv)                                                                    \
    ({                                                                                          \
        float4 g_ = dv;                                                                         \
        if (relu) {                                                                             \
            if (msc) {                                                                          \
                g_ = make_float4(yv.x * ms4.x + mh4.x > 0.f ? dv.x : 0.f, yv.y * ms4.y + mh4.y > 0.f ? dv.y : 0.f, \
                                 yv.z * ms4.z + mh4.z > 0.f ? dv.z : 0.f, yv.w * ms4.w + mh4.w > 0.f ? dv.w : 0.f); \
            } else if (mask) {                                                                         \
                const unsigned m_ = mask[(o) >> 2];                                             \
                g_ = make_float4((m_ & 1u) ? dv.x : 0.f, (m_ & 2u) ? dv.y : 0.f, (m_ & 4u) ? dv.z : 0.f, (m_ & 8u) ? dv.w : 0.f); \
            } else {                                                                            \
                const float4 ov_ = *(const float4 *)(out + (o));                                \
                g_ = TR_G(dv, ov_);                                                             \
            }                                                                                   \
        }                                                                                       \
        g_;                                                                                     \
    })

__global__ __launch_bounds__(TR_THREADS) void k_bn_grad_sums(const float *__restrict__ dout, const float *__restrict__ out,
                                                            const uint8_t *__restrict__ mask,
                                                            const float *__restrict__ y, const float *__restrict__ mean,
                                                            const float *__restrict__ inv, long rows, int relu,
                                                            float *__restrict__ part, const float *__restrict__ msc = nullptr,
                                                            const float *__restrict__ msh = nullptr)
{
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float4 mu = *(const float4 *)(mean + 4 * cq), iv = *(const float4 *)(inv + 4 * cq);
    float4 ms4 = mu, mh4 = mu;
    if (msc) { ms4 = *(const float4 *)(msc + 4 * cq); mh4 = *(const float4 *)(msh + 4 * cq); }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    for (long r = (long)blockIdx.x * TR_ROWLANES + rl; r < rows; r += (long)gridDim.x * TR_ROWLANES) {
        const long o = r * TR_C + 4 * cq;
        const float4 dv = *(const float4 *)(dout + o);
        const float4 v = *(const float4 *)(y + o);
        const float4 g = TR_MASKED(dv, o, v);
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        q.x += g.x * ((v.x - mu.x) * iv.x); q.y += g.y * ((v.y - mu.y) * iv.y);
        q.z += g.z * ((v.z - mu.z) * iv.z); q.w += g.w * ((v.w - mu.w) * iv.w);
    }
    tr_block_reduce_store(s, q, part);
}

// dx = a (g - b - xhat c) with a = gamma * inv, b = sum(g) / count, c = sum(g * xhat) / count; g_out (optional) = g
__global__ __launch_bounds__(TR_THREADS) void k_bn_grad_apply(const float *__restrict__ dout, const float *__restrict__ out,
                                                             const uint8_t *__restrict__ mask,
                                                             const float *__restrict__ y, const float *__restrict__ mean,
                                                             const float *__restrict__ inv, const float *__restrict__ a,
                                                             const float *__restrict__ b, const float *__restrict__ c,
                                                             float *__restrict__ dx, float *__restrict__ g_out, long rows,
                                                             int relu, float *__restrict__ amax_part,
                                                             const float *__restrict__ msc = nullptr, const float *__restrict__ msh = nullptr)
{
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float am = 0.f;
    const float4 mu = *(const float4 *)(mean + 4 * cq), iv = *(const float4 *)(inv + 4 * cq);
    float4 ms4 = mu, mh4 = mu;
    if (msc) { ms4 = *(const float4 *)(msc + 4 * cq); mh4 = *(const float4 *)(msh + 4 * cq); }
    const float4 a4 = *(const float4 *)(a + 4 * cq), b4 = *(const float4 *)(b + 4 * cq), c4 = *(const float4 *)(c + 4 * cq);
    for (long r = (long)blockIdx.x * TR_ROWLANES + rl; r < rows; r += (long)gridDim.x * TR_ROWLANES) {
        const long o = r * TR_C + 4 * cq;
        const float4 dv = *(const float4 *)(dout + o);
        const float4 v = *(const float4 *)(y + o);
        const float4 g = TR_MASKED(dv, o, v);
        float4 d;
        d.x = a4.x * (g.x - b4.x - ((v.x - mu.x) * iv.x) * c4.x);
        d.y = a4.y * (g.y - b4.y - ((v.y - mu.y) * iv.y) * c4.y);
        d.z = a4.z * (g.z - b4.z - ((v.z - mu.z) * iv.z) * c4.z);
        d.w = a4.w * (g.w - b4.w - ((v.w - mu.w) * iv.w) * c4.w);
        *(float4 *)(dx + o) = d;
        if (g_out) *(float4 *)(g_out + o) = g;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
    }
    if (amax_part) tr_block_amax(am, amax_part);
}

static int tr_grid(long rows)
{
    const long want = (rows + TR_ROWLANES - 1) / TR_ROWLANES;
    return (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);      // eight blocks per CU; every thread strides over its rows
}

extern "C" int snk_bn_train_apply(const float *d_y, const float *d_scale, const float *d_shift, const float *d_residual,
                                  float *d_out, long rows, int relu, float *d_partials, float *d_out_scale_tail,
                                  uint8_t *d_relu_mask, void *stream)
{
    SNK_REQUIRE(d_y && d_scale && d_shift && d_out && rows > 0 && (!d_out_scale_tail || d_partials), "snk_bn_train_apply: bad argument");
    const int grid = tr_grid(rows);
    k_bn_apply<false><<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_y, d_scale, d_shift, d_residual, d_out, rows, relu,
                                                                     d_out_scale_tail ? d_partials : nullptr, d_relu_mask);
    if (d_out_scale_tail) k_amax_scale<<<1, 1024, 0, (hipStream_t)stream>>>(d_partials, grid, d_out_scale_tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_bn_train_apply (relu = 1) whose shortcut is a deferred activation: d_res_y holds PRE-batch-norm values and every one is taken as
// relu(y * res_scale[c] + res_shift[c]) -- the residual block right above a deferred stem reads the stem's output this way
extern "C" int snk_bn_train_apply_res_deferred(const float *d_y, const float *d_scale, const float *d_shift, const float *d_res_y,
                                               const float *d_res_scale, const float *d_res_shift, float *d_out, long rows,
                                               float *d_partials, float *d_out_scale_tail, uint8_t *d_relu_mask, void *stream)
{
    SNK_REQUIRE(d_y && d_scale && d_shift && d_res_y && d_res_scale && d_res_shift && d_out && rows > 0 && (!d_out_scale_tail || d_partials),
                "snk_bn_train_apply_res_deferred: bad argument");
    const int grid = tr_grid(rows);
    k_bn_apply<false><<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_y, d_scale, d_shift, d_res_y, d_out, rows, 1,
                                                                     d_out_scale_tail ? d_partials : nullptr, d_relu_mask, nullptr, nullptr,
                                                                     nullptr, nullptr, d_res_scale, d_res_shift);
    if (d_out_scale_tail) k_amax_scale<<<1, 1024, 0, (hipStream_t)stream>>>(d_partials, grid, d_out_scale_tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_bn_train_apply (relu = 1) for the tower's LAST layer, with the head's 1x1 convolution and its batch-norm sums taken from the
// values on their way out: d_z[rows] = dot(out[row][:], d_w1x1), d_hsums = { sum (z - center1), sum (z - center1)^2 } (float64;
// d_center1: one float or NULL) -- what snk_head_conv1x1_sums computes from d_out in a pass of its own.
extern "C" int snk_bn_train_apply_head(const float *d_y, const float *d_scale, const float *d_shift, const float *d_residual,
                                       float *d_out, long rows, float *d_partials, uint8_t *d_relu_mask, const float *d_w1x1,
                                       const float *d_center1, float *d_z, double *d_hsums, void *stream)
{
    SNK_REQUIRE(d_y && d_scale && d_shift && d_out && rows > 0 && d_partials && d_w1x1 && d_z && d_hsums, "snk_bn_train_apply_head: bad argument");
    const int grid = tr_grid(rows);
    float *part1 = d_partials + 2048;          // (behind the 2 048 maxima a plain launch would leave)
    k_bn_apply<true><<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_y, d_scale, d_shift, d_residual, d_out, rows, 1, nullptr, d_relu_mask,
                                                                    d_w1x1, d_center1, d_z, part1);
    tf_fold<double>(part1, grid, 2, 2, 1.0, d_hsums, (double *)(d_partials + 2048 * 2 * TR_C), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the same sums kept in float64 (the manual training step, snake_engine/train_step.py: the ranks all-reduce float64 sums)
extern "C" int snk_bn_train_grad_sums_f64(const float *d_dout, const float *d_out, const uint8_t *d_relu_mask, const float *d_y,
                                          const float *d_mean, const float *d_inv, long rows, int relu, float *d_partials,
                                          double *d_sums, void *stream)
{
    SNK_REQUIRE(d_dout && d_y && d_mean && d_inv && d_partials && d_sums && rows > 0 && (!relu || d_out || d_relu_mask),
                "snk_bn_train_grad_sums_f64: bad argument");
    const int grid = tr_grid(rows);
    k_bn_grad_sums<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_dout, d_out, d_relu_mask, d_y, d_mean, d_inv, rows, relu, d_partials);
    tf_fold<double>(d_partials, grid, 2 * TR_C, 2 * TR_C, 1.0, d_sums, (double *)(d_partials + 2048 * 2 * TR_C), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_bn_train_grad_apply(const float *d_dout, const float *d_out, const uint8_t *d_relu_mask, const float *d_y,
                                       const float *d_mean, const float *d_inv, const float *d_a, const float *d_b, const float *d_c,
                                       float *d_dx, float *d_g, long rows, int relu, float *d_partials, float *d_dx_scale_tail,
                                       void *stream)
{
    SNK_REQUIRE(d_dout && d_y && d_mean && d_inv && d_a && d_b && d_c && d_dx && rows > 0 && (!relu || d_out || d_relu_mask) &&
                (!d_dx_scale_tail || d_partials), "snk_bn_train_grad_apply: bad argument");
    const int grid = tr_grid(rows);
    k_bn_grad_apply<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_dout, d_out, d_relu_mask, d_y, d_mean, d_inv, d_a, d_b, d_c, d_dx, d_g, rows,
                                                                   relu, d_dx_scale_tail ? d_partials : nullptr);
    if (d_dx_scale_tail) k_amax_scale<<<1, 1024, 0, (hipStream_t)stream>>>(d_partials, grid, d_dx_scale_tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the two backward kernels for a layer whose batch norm + ReLU output was never written (deferred): the ReLU decision is
// d_y * d_scale[c] + d_shift[c] > 0, the sign of what snk_bn_train_apply(relu = 1, no residual) would have written
extern "C" int snk_bn_train_grad_sums_f64_deferred(const float *d_dout, const float *d_y, const float *d_scale, const float *d_shift,
                                                   const float *d_mean, const float *d_inv, long rows, float *d_partials, double *d_sums,
                                                   void *stream)
{
    SNK_REQUIRE(d_dout && d_y && d_scale && d_shift && d_mean && d_inv && d_partials && d_sums && rows > 0,
                "snk_bn_train_grad_sums_f64_deferred: bad argument");
    const int grid = tr_grid(rows);
    k_bn_grad_sums<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_dout, nullptr, nullptr, d_y, d_mean, d_inv, rows, 1, d_partials, d_scale, d_shift);
    tf_fold<double>(d_partials, grid, 2 * TR_C, 2 * TR_C, 1.0, d_sums, (double *)(d_partials + 2048 * 2 * TR_C), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_bn_train_grad_apply_deferred(const float *d_dout, const float *d_y, const float *d_scale, const float *d_shift,
                                                const float *d_mean, const float *d_inv, const float *d_a, const float *d_b,
                                                const float *d_c, float *d_dx, float *d_g, long rows, float *d_partials,
                                                float *d_dx_scale_tail, void *stream)
{
    SNK_REQUIRE(d_dout && d_y && d_scale && d_shift && d_mean && d_inv && d_a && d_b && d_c && d_dx && rows > 0 &&
                (!d_dx_scale_tail || d_partials), "snk_bn_train_grad_apply_deferred: bad argument");
    const int grid = tr_grid(rows);
    k_bn_grad_apply<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_dout, nullptr, nullptr, d_y, d_mean, d_inv, d_a, d_b, d_c, d_dx, d_g, rows, 1,
                                                                   d_dx_scale_tail ? d_partials : nullptr, d_scale, d_shift);
    if (d_dx_scale_tail) k_amax_scale<<<1, 1024, 0, (hipStream_t)stream>>>(d_partials, grid, d_dx_scale_tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_f16s_input_scale(const float *d_x, long n_floats, void *d_wS, float *d_partials, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_partials && n_floats > 0 && n_floats % 4 == 0, "snk_conv3x3_f16s_input_scale: bad argument");
    const long n4 = n_floats / 4;
    const long want = (n4 + TR_THREADS - 1) / TR_THREADS;
    const int grid = (int)(want < 2048 ? want : 2048);
    float *tail = (float *)((char *)d_wS + SNK_CONV_F16S_TAIL_OFFSET);
    k_amax_part<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(d_x, n4, d_partials);
    k_amax_scale<<<1, 1024, 0, (hipStream_t)stream>>>(d_partials, grid, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// floats the caller provides for d_partials: 2 048 blocks' partials and, behind them, the scratch of the float64 folds
extern "C" int snk_bn_train_partials(void) { return 2048 * 2 * TR_C + TF_SCRATCH_FLOATS(2 * TR_C); }
