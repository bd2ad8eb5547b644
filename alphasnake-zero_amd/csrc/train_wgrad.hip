// train_wgrad.hip -- weight gradient of the tower's 3x3 convolution (AlphaNNet.train,
// alpha_nnet.py:58-59) at float32 accuracy on the f16 matrix pipe, the third of the training step's three convolution passes
// (forward and input gradient run on k_conv3x3_f16s, csrc/conv_split.hip):
//
//     dW[dy][dx][ci][co] = sum over images n and pixels (y, x) of  X[n][y + dy][x + dx][ci] * dY[n][y][x][co]      (X zero-padded)
//
// A GEMM whose reduction runs over PIXELS -- the slow axis of the channels-last tensors -- so both operands are transposed on
// their way into LDS: a block owns a 32 x 32 slice (ci, co) of all nine taps and a group of images; per image it lays the
// slice of X and of dY into LDS as [channel][padded pixel] f16 hi / lo planes (pixel (y, x) at (y + 1) * P + x, row pitch
// P = W + 1 rounded up to 8: the zero columns behind a row are the next row's left border, rows 0 and H + 1 are zero, so
// that the tap (dy, dx) is the SAME plane read dy * P + dx elements further on), then every wave runs the
// MFMAs of a quarter of the 16-pixel k-steps for all nine taps (nine 32 x 32 accumulators).  A tap with dx = +-1 starts one
// element off the 16-byte grid: its fragment is the aligned one and a neighbouring dword funnel-shifted by 16 bits
// (v_alignbit), four VALU instructions per fragment.  Products are hi*hi + hi*lo + lo*hi as in the forward kernel; the
// operands' power-of-two scales are the ones the forward / input-gradient calls already derived from the data.
// The image groups' partial results are summed in a fixed order by k_wgrad_fold (float64): a run repeats bit for bit.
//
// Measured (MI355X, 2 048 images of 21 x 21, tools/wgrad_time.py): 1.41 ms = 188 TFLOP/s algorithmic, error 3e-7 against
// float64; the library's float32 kernel (igemm_wrw, f32 matrix pipe) takes 2.35 ms.  With the MFMA loop compiled out the kernel
// takes 0.77 ms, with the next-image fetch compiled out 0.99: the MFMA phase itself (0.64 ms) is at the matrix pipe's rate
// for its 3 x 1.25 (split, padded positions) executed flops, the rest is staging that does not overlap it (one wave per SIMD:
// the block fills the LDS).  What did NOT change the time: float4 loads + 2-byte LDS stores against 4-byte loads + 16-byte
// LDS stores (the first trades LDS bank conflicts for the second's 128 load instructions per thread and image), the
// permlane swap in place of half the 4-byte LDS reads, the XCD-aware block order.  Next: an LDS image [pixel][channel] filled
// with float4 loads / 8-byte stores and read through ds_read_b64_tr_b16 (gfx950's transposing LDS read): the taps become
// row offsets, no funnel shifts, 4 LDS stores per item instead of 16.
#include "common.h"

typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

#define WG_GROUPS 16                 // image groups: 16 slices x 16 groups = 256 blocks, one per CU (the block fills its LDS)

struct WgArgs {
    const float *x, *dy;
    float *part;
    const float *x_tail, *dy_tail;   // { ., ., scale, 1 / scale } of the two operands (weight-image tails, conv_split.hip)
    int n_images, H, W, P, nk, pitch_x, pitch_y, per_group, guard;   // guard: elements in front of the X planes (the taps reach P + 1 back)
};

__device__ static inline wg_f16x8 wg_frag(uint4 v)
{
    union { uint4 u; wg_f16x8 f; } c;
    c.u = v;
    return c.f;
}

__global__ __launch_bounds__(256) void k_wgrad_f16s(WgArgs p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    _Float16 *TXh = (_Float16 *)smem, *TXl = TXh + 32 * p.pitch_x;
    _Float16 *TYh = TXl + 32 * p.pitch_x, *TYl = TYh + 32 * p.pitch_y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    // consecutive workgroups go to different XCDs (own L2 each): the sixteen slices of an image group -- which read the same
    // images, each X slice four times, each dY slice four times -- are put on ONE XCD, two groups per XCD
    static_assert(WG_GROUPS == 16, "256 workgroups = 8 XCDs x 2 image groups x 16 slices");
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int grp = 2 * xcd + (local >> 4), slice = local & 15, cs = slice >> 2, os = slice & 3;
    const float sx = p.x_tail[2], sy = p.dy_tail[2];
    const int lds_bytes = 2 * 32 * (p.pitch_x + p.pitch_y) * 2;
    for (int o = tid * 16; o < lds_bytes; o += 256 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // borders stay zero
    wg_f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();

    const int HW = p.H * p.W, P = p.P;
    const int n0 = grp * p.per_group, n1 = min(n0 + p.per_group, p.n_images);
    // Staging: a thread owns ONE channel of eight consecutive pixels of a row (item = channel + 32 * (row * groups + group)):
    // eight 4-byte loads that are contiguous over the 32 channel lanes, one 16-byte LDS store per plane.  (The first form --
    // a float4 of four channels per thread, sixteen 2-byte LDS stores per item -- spent 16k cycles per image in LDS bank
    // conflicts, twice the MFMA time.)  The next image's values travel from HBM into registers while the current image's
    // MFMAs run: the loads are issued just before the MFMA loop, which reads LDS only.
    constexpr int MAXIT = 8;                           // items per thread: 32 * h * ceil(w / 8) / 256 <= 8
    float xv[MAXIT][8], yv[MAXIT][8];
    const int G3 = (p.W + 7) >> 3, items = 32 * p.H * G3;
    const float invG = 1.0f / (float)G3;
    auto fetch = [&](int n) {
        const float *xn = p.x + (long)n * HW * 128 + 32 * cs, *yn = p.dy + (long)n * HW * 128 + 32 * os;
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int it = tid + 256 * j, c = it & 31, gi = it >> 5;
            const int y = (int)(((float)gi + 0.5f) * invG), x0 = 8 * (gi - y * G3);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = it < items && x0 + e < p.W;
                const long o = ok ? ((long)(y * p.W + x0 + e) * 128 + c) : 0;
                const float a = xn[o], b = yn[o];
                xv[j][e] = ok ? a : 0.f;
                yv[j][e] = ok ? b : 0.f;
            }
        }
    };
    if (n0 < n1) fetch(n0);
    for (int n = n0; n < n1; ++n) {
        int tid_v = tid;
        asm volatile("" : "+v"(tid_v));                // opaque per image: keeps the LDS addresses of the unrolled loop from being
                                                       // hoisted out of the image loop (hoisted, they spilled the register file)
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int it = tid_v + 256 * j, c = it & 31, gi = it >> 5;
            if (it < items) {
                const int y = (int)(((float)gi + 0.5f) * invG), x0 = 8 * (gi - y * G3);
                const int q0 = (y + 1) * P + x0;       // a multiple of 8 elements: one 16-byte store per plane
                wg_f16x8 xh, xl, yh, yl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = xv[j][e] * sx, b = yv[j][e] * sy;
                    xh[e] = (_Float16)a; xl[e] = (_Float16)(a - (float)xh[e]);
                    yh[e] = (_Float16)b; yl[e] = (_Float16)(b - (float)yh[e]);
                }
                *(wg_f16x8 *)&TXh[c * p.pitch_x + q0 + p.guard] = xh; *(wg_f16x8 *)&TXl[c * p.pitch_x + q0 + p.guard] = xl;
                *(wg_f16x8 *)&TYh[c * p.pitch_y + q0] = yh; *(wg_f16x8 *)&TYl[c * p.pitch_y + q0] = yl;
            }
        }
        __syncthreads();
        if (n + 1 < n1) fetch(n + 1);
        // ---- MFMA: this wave's k-steps, nine taps each
        for (int ks = wv; ks < p.nk; ks += 4) {
            const int k0 = 16 * ks + 8 * h;
            const wg_f16x8 bh = *(const wg_f16x8 *)&TYh[l31 * p.pitch_y + k0], bl = *(const wg_f16x8 *)&TYl[l31 * p.pitch_y + k0];
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const int base = l31 * p.pitch_x + k0 + p.guard + (dyi - 1) * P;      // a multiple of 8 elements
                const uint4 ch = *(const uint4 *)&TXh[base], cl = *(const uint4 *)&TXl[base];
                // the dword in front of / behind the 16-byte block: for the upper k-half (h = 1) the one in front is the lower
                // half's last dword, for the lower half the one behind is the upper half's first -- one v_permlane32_swap
                // delivers both; the other one is a 4-byte LDS read (bank-conflicted: half as many of them this way)
                const unsigned eh = *(const unsigned *)&TXh[h ? base + 8 : base - 2], el = *(const unsigned *)&TXl[h ? base + 8 : base - 2];
                const auto sh_ = __builtin_amdgcn_permlane32_swap(ch.x, ch.w, false, false);
                const auto sl_ = __builtin_amdgcn_permlane32_swap(cl.x, cl.w, false, false);
                const unsigned ph = h ? sh_[0] : eh, nh = h ? eh : sh_[1];
                const unsigned pl = h ? sl_[0] : el, nl = h ? el : sl_[1];
#define WG_AB(hi_, lo_) __builtin_amdgcn_alignbit((hi_), (lo_), 16)
                const wg_f16x8 ah[3] = {wg_frag(make_uint4(WG_AB(ch.x, ph), WG_AB(ch.y, ch.x), WG_AB(ch.z, ch.y), WG_AB(ch.w, ch.z))),   // dx = -1
                                        wg_frag(ch),                                                                              // dx = 0
                                        wg_frag(make_uint4(WG_AB(ch.y, ch.x), WG_AB(ch.z, ch.y), WG_AB(ch.w, ch.z), WG_AB(nh, ch.w)))};  // dx = +1
                const wg_f16x8 al[3] = {wg_frag(make_uint4(WG_AB(cl.x, pl), WG_AB(cl.y, cl.x), WG_AB(cl.z, cl.y), WG_AB(cl.w, cl.z))),
                                        wg_frag(cl),
                                        wg_frag(make_uint4(WG_AB(cl.y, cl.x), WG_AB(cl.z, cl.y), WG_AB(cl.w, cl.z), WG_AB(nl, cl.w)))};
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
        __syncthreads();                               // the planes are rewritten for the next image
    }
    // ---- the four waves' partial sums (disjoint k-steps) through LDS, then this block's slice of its group's partial dW
    float *red = (float *)smem;                        // [wave][tap][row ci][col co] = 4 x 9 x 1024 floats
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

struct WgShape { int P, nk, pitch_x, pitch_y, guard, lds; };
static WgShape wg_shape(int h, int w)
{
    WgShape s;
    s.P = (w + 1 + 7) / 8 * 8;
    s.nk = ((h + 2) * s.P + 15) / 16;
    auto pitch = [](int need) { int v = (need + 63) / 64 * 64 + 8; return v; };      // dword pitch = 4 mod 32: conflict-free 16-byte rows
    s.guard = s.P + 8;
    s.pitch_x = pitch(16 * s.nk + s.guard + s.P + 16);
    s.pitch_y = pitch(16 * s.nk);
    s.lds = 2 * 32 * (s.pitch_x + s.pitch_y) * 2;
    return s;
}

extern "C" long snk_conv3x3_wgrad_partials(int height, int width)
{
    const WgShape s = wg_shape(height, width);
    if (height != width || width < 3 || s.lds > 160 * 1024 || 32 * height * ((width + 7) / 8) > 8 * 256) return -1;     // the planes of one image slice must fit the LDS, its items the registers
    return (long)WG_GROUPS * 9 * 128 * 128;
}

extern "C" int snk_conv3x3_wgrad_f16s(const float *d_x, const float *d_dy, const float *d_x_tail, const float *d_dy_tail,
                                      float *d_partials, float *d_dw, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_dy && d_x_tail && d_dy_tail && d_partials && d_dw && n_images > 0, "snk_conv3x3_wgrad_f16s: bad argument");
    SNK_REQUIRE(snk_conv3x3_wgrad_partials(height, width) > 0, "snk_conv3x3_wgrad_f16s: %d x %d does not fit the LDS planes", height, width);
    const WgShape s = wg_shape(height, width);
    WgArgs a = {d_x, d_dy, d_partials, d_x_tail, d_dy_tail, n_images, height, width, s.P, s.nk, s.pitch_x, s.pitch_y,
                (n_images + WG_GROUPS - 1) / WG_GROUPS, s.guard};
    const int lds = s.lds > 4 * 9 * 1024 * 4 ? s.lds : 4 * 9 * 1024 * 4;      // the planes, or the four waves' accumulators at the end
    k_wgrad_f16s<<<16 * WG_GROUPS, 256, lds, (hipStream_t)stream>>>(a);
    k_wgrad_fold<<<9 * 128 * 128 / 256, 256, 0, (hipStream_t)stream>>>(d_partials, WG_GROUPS, d_dw);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
