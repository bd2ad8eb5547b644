// train_wgrad.hip -- weight gradient of the tower's 3x3 convolution (AlphaNNet.train,
// alpha_nnet.py:58-59) at float32 accuracy on the f16 matrix pipe, the third of the training step's three convolution passes
// (forward and input gradient run on k_conv3x3_f16s, csrc/conv_split.hip):
//
//     dW[dy][dx][ci][co] = sum over images n and pixels (y, x) of  X[n][y + dy][x + dx][ci] * dY[n][y][x][co]      (X zero-padded)
//
// A GEMM whose reduction runs over PIXELS -- the slow axis of the channels-last tensors, while an MFMA operand wants eight
// consecutive k values per lane.  A block owns a 32 x 32 slice (ci, co) of all nine taps and a group of images.  Per image it
// lays its slices of X and dY into LDS in the order they have in HBM, [padded pixel][32 channels] as f16 hi / lo images with
// 64-byte rows (pixel (y, x) in row (y + 1) * P + x, P = W + 1 rounded up to 8: the zero rows behind an image row are the
// next row's left border, image rows 0 and H + 1 are zero): a float4 load and one 8-byte LDS store per image and item.  The MFMA
// fragments are taken with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 channels and
// every lane receives ITS channel's four rows), so a tap (dy, dx) is simply the X image read dy * P + dx ROWS further on; the
// three dx taps of a dy share three reads of twelve consecutive rows and one 16-bit funnel shift (v_alignbit).  Every wave
// runs a quarter of the 16-pixel k-steps for all nine taps (nine 32 x 32 accumulators); products are hi*hi + hi*lo + lo*hi as in
// the forward kernel; the operands' power-of-two scales are the ones the forward / input-gradient calls derived from the data;
// the next image's values travel from HBM into registers while the current image's MFMAs run.  The image groups' partial
// results are summed in a fixed order by k_wgrad_fold (float64): a run repeats bit for bit.
//
// Measured (MI355X, 2 048 images of 21 x 21, tools/wgrad_time.py): 1.27 ms = 210 TFLOP/s algorithmic, error 3e-7 against float64;
// the library's float32 kernel (igemm_wrw, f32 matrix pipe) takes 2.35 ms.  The MFMA phase alone is 0.64 ms (the pipe's rate for
// its 3 x 1.25 -- split, padded positions -- executed flops); staging does not overlap it (one wave per SIMD: the block fills
// the LDS).  The first form of this kernel kept [channel][padded pixel] planes (taps = element offsets, dx = +-1 fragments
// funnel-shifted from two aligned reads): 1.41 ms whichever way its staging was written -- float4 loads + sixteen 2-byte LDS
// stores per item (bank conflicts) or 4-byte loads + 16-byte stores (128 load instructions per thread and image).
#include "common.h"

typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

#define WG_GROUPS 16                 // image groups: 16 slices x 16 groups = 256 blocks, one per CU (the block fills its LDS)

__device__ static inline wg_f16x8 wg_frag(uint4 v)
{
    union { uint4 u; wg_f16x8 f; } c;
    c.u = v;
    return c.f;
}

typedef short wg_s4 __attribute__((ext_vector_type(4)));
#define WG_TR(byte_ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4 *)(byte_ptr))

struct WgArgs {
    const float *x, *dy;
    float *part;
    const float *x_tail, *dy_tail;
    int n_images, H, W, P, nk, rows_x, per_group, gx;     // gx: rows in front of the X image (the taps reach P + 1 rows back)
};

__device__ static inline wg_f16x8 wg_join(wg_s4 a, wg_s4 b)
{
    union { struct { wg_s4 a, b; } s; wg_f16x8 f; } c;
    c.s.a = a; c.s.b = b;
    return c.f;
}

__global__ __launch_bounds__(256) void k_wgrad_f16s(WgArgs p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *XH = smem, *XL = XH + p.rows_x * 64, *YH = XL + p.rows_x * 64, *YL = YH + p.nk * 16 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    static_assert(WG_GROUPS == 16, "256 workgroups = 8 XCDs x 2 image groups x 16 slices");
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int grp = 2 * xcd + (local >> 4), slice = local & 15, cs = slice >> 2, os = slice & 3;
    const float sx = p.x_tail[2], sy = p.dy_tail[2];
    const int lds_bytes = 2 * 64 * (p.rows_x + p.nk * 16);
    for (int o = tid * 16; o < lds_bytes; o += 256 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // borders stay zero
    wg_f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();

    // a transposing read of rows R .. R + 3: lane 4 q + p of a 16-lane group gives the address of row R + q, channels
    // 16 (group & 1) + 4 p .. + 3; the group's lane i receives channel 16 (group & 1) + i = lane & 31, row q in element q
    const int i16 = lane & 15, g4 = lane >> 4;
    const int lane_off = (i16 >> 2) * 64 + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;
    const int HW = p.H * p.W, P = p.P, items = HW * 8;
    const float invW = 1.0f / (float)p.W;
    const int n0 = grp * p.per_group, n1 = min(n0 + p.per_group, p.n_images);
    constexpr int MAXIT = 14;                          // (pixel, four channels) items per thread: h * w * 8 / 256 <= 14 (21 x 21)
    float4 xv[MAXIT], yv[MAXIT];
    auto fetch = [&](int n) {
        const float *xn = p.x + (long)n * HW * 128 + 32 * cs, *yn = p.dy + (long)n * HW * 128 + 32 * os;
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int it = min(tid + 256 * j, items - 1);       // clamped: every load is in range, unused ones are not stored
            xv[j] = *(const float4 *)(xn + (long)(it >> 3) * 128 + 4 * (it & 7));
            yv[j] = *(const float4 *)(yn + (long)(it >> 3) * 128 + 4 * (it & 7));
        }
    };
    auto split4 = [](float4 v, float s, uint2 &hi, uint2 &lo) {
        const float a[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        union { _Float16 f[4]; uint2 u; } H_, L_;
#pragma unroll
        for (int e = 0; e < 4; ++e) { H_.f[e] = (_Float16)a[e]; L_.f[e] = (_Float16)(a[e] - (float)H_.f[e]); }
        hi = H_.u; lo = L_.u;
    };
    if (n0 < n1) fetch(n0);
    for (int n = n0; n < n1; ++n) {
        int tid_v = tid;
        asm volatile("" : "+v"(tid_v));                // opaque per image: the unrolled loop's LDS addresses are not hoisted
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int it = tid_v + 256 * j;
            if (it < items) {
                const int pix = it >> 3, c4 = it & 7;
                const int y = (int)(((float)pix + 0.5f) * invW), x = pix - y * p.W;
                const int q = (y + 1) * P + x;
                uint2 hi, lo;
                split4(xv[j], sx, hi, lo);
                *(uint2 *)(XH + (q + p.gx) * 64 + 8 * c4) = hi; *(uint2 *)(XL + (q + p.gx) * 64 + 8 * c4) = lo;
                split4(yv[j], sy, hi, lo);
                *(uint2 *)(YH + q * 64 + 8 * c4) = hi; *(uint2 *)(YL + q * 64 + 8 * c4) = lo;
            }
        }
        __syncthreads();
        if (n + 1 < n1) fetch(n + 1);
        for (int ks = wv; ks < p.nk; ks += 4) {
            const int R = 16 * ks + 8 * h;
            const wg_f16x8 bh = wg_join(WG_TR(YH + R * 64 + lane_off), WG_TR(YH + (R + 4) * 64 + lane_off));
            const wg_f16x8 bl = wg_join(WG_TR(YL + R * 64 + lane_off), WG_TR(YL + (R + 4) * 64 + lane_off));
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const int Rx = R + p.gx + (dyi - 1) * P - 1;           // rows Rx .. Rx + 11 hold the three dx taps' elements
                union { struct { wg_s4 a, b, c; } s; unsigned d[6]; } uh, ul;
                uh.s.a = WG_TR(XH + Rx * 64 + lane_off); uh.s.b = WG_TR(XH + (Rx + 4) * 64 + lane_off); uh.s.c = WG_TR(XH + (Rx + 8) * 64 + lane_off);
                ul.s.a = WG_TR(XL + Rx * 64 + lane_off); ul.s.b = WG_TR(XL + (Rx + 4) * 64 + lane_off); ul.s.c = WG_TR(XL + (Rx + 8) * 64 + lane_off);
#define WG_AB(hi_, lo_) __builtin_amdgcn_alignbit((hi_), (lo_), 16)
                const wg_f16x8 ah[3] = {wg_frag(make_uint4(uh.d[0], uh.d[1], uh.d[2], uh.d[3])),                                              // dx = -1
                                        wg_frag(make_uint4(WG_AB(uh.d[1], uh.d[0]), WG_AB(uh.d[2], uh.d[1]), WG_AB(uh.d[3], uh.d[2]), WG_AB(uh.d[4], uh.d[3]))),
                                        wg_frag(make_uint4(uh.d[1], uh.d[2], uh.d[3], uh.d[4]))};                                              // dx = +1
                const wg_f16x8 al[3] = {wg_frag(make_uint4(ul.d[0], ul.d[1], ul.d[2], ul.d[3])),
                                        wg_frag(make_uint4(WG_AB(ul.d[1], ul.d[0]), WG_AB(ul.d[2], ul.d[1]), WG_AB(ul.d[3], ul.d[2]), WG_AB(ul.d[4], ul.d[3]))),
                                        wg_frag(make_uint4(ul.d[1], ul.d[2], ul.d[3], ul.d[4]))};
#undef WG_AB
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi) {
                    const int t = 3 * dyi + dxi;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dxi], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dxi], bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[dxi], bh, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *red = (float *)smem;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[((wv * 9 + t) * 32 + ((r & 3) + 8 * (r >> 2) + 4 * h)) * 32 + l31] = acc[t][r];
    __syncthreads();
    const float inv = p.x_tail[3] * p.dy_tail[3];
    for (int i = tid; i < 9 * 1024; i += 256) {
        const float s = ((red[i] + red[9216 + i]) + red[2 * 9216 + i]) + red[3 * 9216 + i];
        const int t = i >> 10, row = (i >> 5) & 31, col = i & 31;
        p.part[(((long)grp * 9 + t) * 128 + 32 * cs + row) * 128 + 32 * os + col] = s * inv;
    }
}

__global__ __launch_bounds__(256) void k_wgrad_fold(const float *__restrict__ part, int groups, float *__restrict__ dw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;      // 9 * 128 * 128 outputs
    double acc = 0.0;
    for (int g = 0; g < groups; ++g) acc += (double)part[(long)g * (9 * 128 * 128) + i];
    dw[i] = (float)acc;
}

struct WgShape { int P, nk, rows_x, gx, lds; };
static WgShape wg_shape(int h, int w)
{
    WgShape s;
    s.P = (w + 1 + 7) / 8 * 8;
    s.nk = ((h + 2) * s.P + 15) / 16;
    s.gx = s.P + 8;
    s.rows_x = 16 * s.nk + 2 * s.P + 16;
    s.lds = 2 * 64 * (s.rows_x + 16 * s.nk);
    return s;
}

extern "C" long snk_conv3x3_wgrad_partials(int height, int width)
{
    const WgShape s = wg_shape(height, width);
    if (height != width || width < 3 || s.lds > 160 * 1024 || height * width * 8 > 14 * 256) return -1;   // the images of one
    return (long)WG_GROUPS * 9 * 128 * 128;                                  // slice must fit the LDS, its items the registers
}

extern "C" int snk_conv3x3_wgrad_f16s(const float *d_x, const float *d_dy, const float *d_x_tail, const float *d_dy_tail,
                                      float *d_partials, float *d_dw, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_dy && d_x_tail && d_dy_tail && d_partials && d_dw && n_images > 0, "snk_conv3x3_wgrad_f16s: bad argument");
    SNK_REQUIRE(snk_conv3x3_wgrad_partials(height, width) > 0, "snk_conv3x3_wgrad_f16s: %d x %d does not fit the LDS images", height, width);
    const WgShape s = wg_shape(height, width);
    WgArgs a = {d_x, d_dy, d_partials, d_x_tail, d_dy_tail, n_images, height, width, s.P, s.nk, s.rows_x,
                (n_images + WG_GROUPS - 1) / WG_GROUPS, s.gx};
    const int lds = s.lds > 4 * 9 * 1024 * 4 ? s.lds : 4 * 9 * 1024 * 4;      // the images, or the four waves' accumulators at the end
    k_wgrad_f16s<<<16 * WG_GROUPS, 256, lds, (hipStream_t)stream>>>(a);
    k_wgrad_fold<<<9 * 128 * 128 / 256, 256, 0, (hipStream_t)stream>>>(d_partials, WG_GROUPS, d_dw);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
