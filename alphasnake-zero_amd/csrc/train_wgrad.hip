// train_wgrad.hip -- weight gradient of the tower's 3x3 convolution (AlphaNNet.train,
// alpha_nnet.py:58-59) at float32 accuracy on the f16 matrix pipe, the third of the training step's three convolution passes
// (forward and input gradient run on k_conv3x3_f16s, csrc/conv_split.hip):
//
//     dW[dy][dx][ci][co] = sum over images n and pixels (y, x) of  X[n][y + dy][x + dx][ci] * dY[n][y][x][co]      (X zero-padded)
//
// A GEMM whose reduction runs over PIXELS -- the slow axis of the channels-last tensors, while an MFMA operand wants eight
// consecutive k values per lane.  A block owns a 32 x 32 slice (ci, co) of all nine taps and a group of images, and works
// through them in SLABS of RB image rows.  A slab's slices of dY (RB rows) and X (RB + 2 rows: one above, one below, zero
// outside the image) lie in LDS in the order they have in HBM, [pixel][32 channels] as f16 hi / lo images with 64-byte rows
// and pitch P = W + 1 (the zero cell behind an image row is the next row's left border).  The MFMA fragments are taken with
// gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 channels and every lane receives ITS
// channel's four rows), so a tap (dy, dx) is simply the X image read (dy + 1) * P + dx ROWS further on (a transposing read may
// start at any row: no funnel shifts), and the reads run one (k-step, dy) group ahead of the MFMAs that use them.  Products are hi*hi + hi*lo +
// lo*hi as in the forward kernel; the operands' power-of-two scales are the ones the element-wise kernels measured.
//
// Round 3: the block is 8 wavefronts with two ROLES and two LDS buffers.  Wavefronts 0-3 run the MFMAs of slab t from buffer
// t & 1 (a quarter of the 16-pixel k-steps each, nine 32 x 32 accumulators); wavefronts 4-7 meanwhile split slab t + 1 into
// the other buffer and fetch slab t + 2 from HBM into registers; one block barrier per slab.  Every SIMD holds one MFMA
// wavefront and one staging wavefront, whose vector instructions issue in the gaps of the other's MFMAs.  (Round 2's form --
// 4 wavefronts, a whole image per barrier pair, staging and MFMA phases in sequence -- took 1.27 ms per layer at 2 048
// images of 21 x 21 with an MFMA phase of 0.64 ms; it also kept whole images in LDS and so refused 37 x 37.)  Slabs make the
// GEMM's k range follow the real pixels more closely (32 k-steps per 21 x 21 image instead of 35).
// The image groups' partial results are summed in a fixed order by k_wgrad_fold (float64): a run repeats bit for bit.
#include "common.h"
#include <stdlib.h>
#include <string.h>

typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

#define WG_GROUPS 16                 // image groups: 16 slices x 16 groups = 256 blocks, one per CU (the block fills its LDS)
#define WG_GX 8                      // rows in front of the X image (the dx = -1 tap of the first pixel reads one row back)
#define WG_MAXX 9                    // (pixel, four channels) items per staging thread and slab: X, then dY
#define WG_MAXY 8

typedef short wg_s4 __attribute__((ext_vector_type(4)));
#define WG_TR(byte_ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4 *)(byte_ptr))

struct WgArgs {
    const float *x, *dy;
    float *part;
    const float *x_tail, *dy_tail;
    int n_images, H, W, P, RB, n_slabs, nk, rows_x, per_group, nx_items, ny_items;
};

__device__ static inline wg_f16x8 wg_join(wg_s4 a, wg_s4 b)
{
    union { struct { wg_s4 a, b; } s; wg_f16x8 f; } c;
    c.s.a = a; c.s.b = b;
    return c.f;
}

// hi = f16(v s), lo = f16(v s - hi): s is a power of two, so both are single fused operations on the unscaled value
// (v_fma_mixlo / mixhi_f16: float32 operands and an f16 addend in, f16 out) -- two instructions per element
__device__ static inline void wg_split4(float4 v, float s, uint2 &hi, uint2 &lo)
{
    const float a[4] = {v.x, v.y, v.z, v.w};
    union { _Float16 f[4]; uint2 u; } H_, L_;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        H_.f[e] = (_Float16)__builtin_fmaf(a[e], s, 0.0f);
        L_.f[e] = (_Float16)__builtin_fmaf(a[e], s, -(float)H_.f[e]);
    }
    hi = H_.u; lo = L_.u;
}

__global__ __launch_bounds__(512) void k_wgrad_f16s(WgArgs p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    static_assert(WG_GROUPS == 16, "256 workgroups = 8 XCDs x 2 image groups x 16 slices");
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int grp = 2 * xcd + (local >> 4), slice = local & 15, cs = slice >> 2, os = slice & 3;
    const int P = p.P, W = p.W, H = p.H;
    const int y_rows = p.nk * 16;
    const int buf_bytes = 2 * 64 * (p.rows_x + y_rows);               // XH | XL | YH | YL of one buffer
    for (int o = tid * 16; o < 2 * buf_bytes; o += 512 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // borders stay zero
    const int n0 = grp * p.per_group, n1 = min(n0 + p.per_group, p.n_images);
    const int T = max(n1 - n0, 0) * p.n_slabs;                        // slabs this block works through
    float *red = (float *)smem;

    if (wv >= 4) {
        // ---------------- staging role: 256 threads lay slab after slab into the buffers ----------------
        const int st = tid - 256;
        const float sx = p.x_tail[2], sy = p.dy_tail[2];
        const float invW = 1.0f / (float)W;
        int xl[WG_MAXX], xg[WG_MAXX];                                 // LDS byte offset, byte offset inside the image (slab at row 0)
        int yl[WG_MAXY], yg[WG_MAXY];
        // items past the slab's last one repeat it (same value to the same LDS address): no predication, no branches
#pragma unroll
        for (int j = 0; j < WG_MAXX; ++j) {
            const int it = min(st + 256 * j, p.nx_items - 1);
            const int pix = it >> 3, c4 = it & 7;
            const int yy = (int)(((float)pix + 0.5f) * invW), xx = pix - yy * W;      // yy = 0 .. RB + 1: image row r0 - 1 + yy
            xl[j] = (WG_GX + yy * P + xx) * 64 + 8 * c4;
            xg[j] = (((yy - 1) * W + xx) * 128 + 4 * c4) * 4;
        }
#pragma unroll
        for (int j = 0; j < WG_MAXY; ++j) {
            const int it = min(st + 256 * j, p.ny_items - 1);
            const int pix = it >> 3, c4 = it & 7;
            const int yy = (int)(((float)pix + 0.5f) * invW), xx = pix - yy * W;      // yy = 0 .. RB - 1: image row r0 + yy
            yl[j] = (yy * P + xx) * 64 + 8 * c4;
            yg[j] = ((yy * W + xx) * 128 + 4 * c4) * 4;
        }
        float4 xv[WG_MAXX], yv[WG_MAXY];
        // One buffer descriptor per image and tensor: rows above the image have negative offsets, rows below it offsets past
        // the image's bytes -- the hardware's range check returns zeros for both, which is exactly the padding the taps need
        const int img_bytes = H * W * 128 * 4;
        auto fetch = [&](int t) {                                     // slab t of this block -> registers
            const int n = __builtin_amdgcn_readfirstlane(n0 + t / p.n_slabs);
            const int slab = __builtin_amdgcn_readfirstlane((t % p.n_slabs) * p.RB * W * 128 * 4);
            const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (long)n * H * W * 128 + 32 * cs), 0, img_bytes - 128 * cs, 0x00020000);
            const auto ry = __builtin_amdgcn_make_buffer_rsrc((void *)(p.dy + (long)n * H * W * 128 + 32 * os), 0, img_bytes - 128 * os, 0x00020000);
#pragma unroll
            for (int j = 0; j < WG_MAXX; ++j) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rx, xg[j] + slab, 0, 0);
                xv[j] = __builtin_bit_cast(float4, v);
            }
#pragma unroll
            for (int j = 0; j < WG_MAXY; ++j) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(ry, yg[j] + slab, 0, 0);
                yv[j] = __builtin_bit_cast(float4, v);
            }
        };
        auto store = [&](int b) {                                     // registers -> buffer b, split into f16 hi / lo
            unsigned char *XH = smem + b * buf_bytes, *XL = XH + p.rows_x * 64, *YH = XL + p.rows_x * 64, *YL = YH + y_rows * 64;
#pragma unroll
            for (int j = 0; j < WG_MAXX; ++j) {
                uint2 hi, lo;
                wg_split4(xv[j], sx, hi, lo);
                *(uint2 *)(XH + xl[j]) = hi; *(uint2 *)(XL + xl[j]) = lo;
            }
#pragma unroll
            for (int j = 0; j < WG_MAXY; ++j) {
                uint2 hi, lo;
                wg_split4(yv[j], sy, hi, lo);
                *(uint2 *)(YH + yl[j]) = hi; *(uint2 *)(YL + yl[j]) = lo;
            }
        };
        if (T > 0) fetch(0);
        __syncthreads();                                              // the zero fill is complete
        if (T > 0) store(0);
        if (T > 1) fetch(1);
        __syncthreads();                                              // slab 0 is in buffer 0
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) store((t + 1) & 1);
            if (t + 2 < T) fetch(t + 2);
            __syncthreads();
        }
    } else {
        // ---------------- MFMA role: wavefront wv runs k-steps wv, wv + 4, ... of every slab ----------------
        // a transposing read of rows R .. R + 3: lane 4 q + p of a 16-lane group gives the address of row R + q, channels
        // 16 (group & 1) + 4 p .. + 3; the group's lane i receives channel 16 (group & 1) + i = lane & 31, row q in element q
        const int i16 = lane & 15, g4 = lane >> 4;
        const int lane_off = (i16 >> 2) * 64 + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;
        const int xoff = WG_GX + P;                                   // X row of the pixel at dY position 0
        wg_f32x16 acc[9];                                             // live in this role only: the staging role's registers hold slabs
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        __syncthreads();
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const unsigned char *XH = smem + (t & 1) * buf_bytes, *XL = XH + p.rows_x * 64, *YH = XL + p.rows_x * 64, *YL = YH + y_rows * 64;
            // Software pipeline over (k-step, dy) groups: the twelve transposing reads of group g + 1 (plus the four dY reads of
            // the next k-step) are issued before the nine MFMAs of group g, whose operands were read one group earlier -- this
            // is the only MFMA wavefront of its SIMD, nothing else covers an LDS wait.  Two fragment sets P / Q alternate:
            // dy = 0, 1, 2 of one k-step use P, Q, P and the next k-step Q, P, Q.  The fences pin that order.
#define WG_SB() __builtin_amdgcn_sched_barrier(0)
#define WG_LOADB1(B_, Y_, R_) B_ = wg_join(WG_TR(Y_ + (R_) * 64 + lane_off), WG_TR(Y_ + ((R_) + 4) * 64 + lane_off));
#define WG_LOADA(AH, AL, R_, dyi_)                                                              \
            _Pragma("unroll") for (int dxi = 0; dxi < 3; ++dxi) {                               \
                const int Rx_ = (R_) + xoff + ((dyi_) - 1) * P + (dxi - 1);                     \
                AH[dxi] = wg_join(WG_TR(XH + Rx_ * 64 + lane_off), WG_TR(XH + (Rx_ + 4) * 64 + lane_off)); \
                AL[dxi] = wg_join(WG_TR(XL + Rx_ * 64 + lane_off), WG_TR(XL + (Rx_ + 4) * 64 + lane_off)); \
            }
#define WG_MFMA3(AH, AL, dyi_)                                                                  \
            _Pragma("unroll") for (int dxi = 0; dxi < 3; ++dxi) {                               \
                const int tp = 3 * (dyi_) + dxi;                                                \
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[dxi], bh, acc[tp], 0, 0, 0); \
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[dxi], bl, acc[tp], 0, 0, 0); \
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[dxi], bh, acc[tp], 0, 0, 0); \
            }
// at most 14 reads are in flight behind the ones an MFMA group waits for (the LDS counter has four bits); the last k-step of
// a slab "prefetches" itself again instead of branching
#define WG_KSTEP(PH, PL, QH, QL)                                                                \
            {                                                                                   \
                const int R = 16 * ks + 8 * h;                                                  \
                const int Rn = ks + 4 < p.nk ? R + 64 : R;                                      \
                WG_LOADA(QH, QL, R, 1) WG_LOADB1(nbh, YH, Rn) WG_SB(); WG_MFMA3(PH, PL, 0) WG_SB(); \
                WG_LOADA(PH, PL, R, 2) WG_LOADB1(nbl, YL, Rn) WG_SB(); WG_MFMA3(QH, QL, 1) WG_SB(); \
                WG_LOADA(QH, QL, Rn, 0) WG_SB(); WG_MFMA3(PH, PL, 2) WG_SB();                    \
                bh = nbh; bl = nbl;                                                             \
            }
            int ks = wv;
            if (ks < p.nk) {
                wg_f16x8 bh, bl, nbh, nbl, a0h[3], a0l[3], a1h[3], a1l[3];
                WG_LOADB1(bh, YH, 16 * ks + 8 * h)
                WG_LOADB1(bl, YL, 16 * ks + 8 * h)
                WG_LOADA(a0h, a0l, 16 * ks + 8 * h, 0)
                for (;;) {
                    WG_KSTEP(a0h, a0l, a1h, a1l)
                    ks += 4;
                    if (ks >= p.nk) break;
                    WG_KSTEP(a1h, a1l, a0h, a0l)
                    ks += 4;
                    if (ks >= p.nk) break;
                }
            }
#undef WG_KSTEP
#undef WG_MFMA3
#undef WG_LOADA
#undef WG_LOADB1
#undef WG_SB
            __syncthreads();
        }
        // the four MFMA wavefronts' accumulators are added through LDS (all slabs are done: the buffers are free)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[((wv * 9 + t) * 32 + ((r & 3) + 8 * (r >> 2) + 4 * h)) * 32 + l31] = acc[t][r];
    }
    __syncthreads();
    const float inv = p.x_tail[3] * p.dy_tail[3];
    for (int i = tid; i < 9 * 1024; i += 512) {
        const float s = ((red[i] + red[9216 + i]) + red[2 * 9216 + i]) + red[3 * 9216 + i];
        const int t = i >> 10, row = (i >> 5) & 31, col = i & 31;
        p.part[(((long)grp * 9 + t) * 128 + 32 * cs + row) * 128 + 32 * os + col] = s * inv;
    }
}

// the groups' partial results, summed in a fixed order (float64): 64 outputs per block, four wavefronts take a quarter of the groups
// each (the window form writes 128 partials: one thread per output walked them in 36 us, a twentieth of the kernel it follows)
__global__ __launch_bounds__(256) void k_wgrad_fold(const float *__restrict__ part, int groups, float *__restrict__ dw)
{
    __shared__ double red[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;                 // 9 * 128 * 128 outputs
    const int per = (groups + 3) / 4, g0 = q * per, g1 = min(g0 + per, groups);
    double acc = 0.0;
#pragma unroll 8
    for (int g = g0; g < g1; ++g) acc += (double)part[(long)g * (9 * 128 * 128) + i];
    red[q][o] = acc;
    __syncthreads();
    if (q == 0) dw[i] = (float)(((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the form the training step runs (k_wgrad_f16s above stays selectable, SNK_WGRAD=slabs, as the A/B partner).
//
// What the slab form lost (SQ counters, profiles/r4_wgrad_sq_counters.json: matrix pipes busy 67 % of the SIMD cycles): its four
// MFMA wavefronts split a slab's k-steps four ways, so a slab costs ceil(nk / 4) k-step times (16 for 14), each of them is the only
// MFMA wavefront of its SIMD (nothing covers an LDS wait), and per-image slabs round 441 pixels up to 480 slots.  Here
//   * every wavefront owns its OWN output tiles and runs ALL k-steps: a block = 64 input channels x all 128 output channels x nine
//     taps, wavefront wv = (32 ci: wv & 1) x (32 co: wv >> 1), nine 32 x 32 accumulators -- nothing is split along k, nothing is
//     summed across wavefronts at the end;
//   * all eight wavefronts stage AND multiply (two MFMA wavefronts per SIMD: one's LDS waits are the other's MFMA time); a
//     wavefront's share of the next window's split-and-store work is spread over its (k-step, dy) groups;
//   * the reduction runs over ONE slot stream per image group: image i, row y, column x sits at slot (i (H + 1) + y) (W + 1) + x,
//     with a zero column behind every row and a zero row behind every image (they are the taps' zero padding: the dx = +-1 / dy = +-1
//     neighbour of an edge pixel is such a slot), and the block works through the stream in windows of NK k-steps whatever the image
//     boundaries: 441 pixels cost 484 slots, nothing is rounded per image or per slab.  A slot's place in HBM comes from two
//     divisions per staged item; slots that are padding get an offset beyond the group's buffer descriptor, which returns zeros.
// LDS per buffer: X as two 32-channel planes x (hi, lo) of 16 NK + 2 P + 16 rows, dY as four planes x (hi, lo) of 16 NK rows, 64-byte
// rows read with ds_read_b64_tr_b16 exactly as above; two buffers; one barrier per window.  256 blocks = 2 channel halves x 128 image
// groups (the two halves of a group read the same dY: neighbours on one XCD); partial results are folded in a fixed order.
#define WG2_GROUPS 128

#ifdef WG_STAMPS        // development build only: s_memtime / s_memrealtime at the block's start, first window and end
__device__ unsigned long long wg_stamp_buf[512 * 8];
#define WG_STAMP(k, v) if (tid == 0) wg_stamp_buf[blockIdx.x * 8 + (k)] = (v);
extern "C" int snk_dbg_wgrad_stamps(unsigned long long *h_out)
{
    SNK_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(wg_stamp_buf), sizeof(unsigned long long) * 512 * 8));
    return 0;
}
#else
#define WG_STAMP(k, v)
#endif

// two transposing reads -> one MFMA operand, as a vector concatenation (the union of wg_join costs the window form a v_mov per half)
typedef short wg_s8 __attribute__((ext_vector_type(8)));
__device__ static inline wg_f16x8 wg_cat(wg_s4 a, wg_s4 b)
{
    return __builtin_bit_cast(wg_f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct Wg2Args {
    const float *x, *dy;
    float *part;
    const float *x_tail, *dy_tail;
    int n_images, H, W, per_group, rows_x, nxs;      // nxs: X slots a window stages = 16 NK + 2 P + 2
    const float *x_scale, *x_shift;                  // AFF: x is a PRE-batch-norm tensor, every value is taken as relu(x * scale[c] + shift[c])
};

// WT: the image width as a compile-time constant (0: read from the arguments): with it every tap's LDS offset is an immediate
// AFF: the deferred batch norm (snake_engine/train_step.py): X is the pre-batch-norm output of the layer below, its batch norm + ReLU
//   -- k_bn_apply's expression, csrc/train.hip -- is applied when an item is laid into LDS; the stream's padding slots, which the
//   buffer descriptor answers with zeros, must stay zeros (they are the padding of the ACTIVATION): one bit per item remembers it
template <int NK, int MAXX, int WT, bool AFF = false>
__global__ __launch_bounds__(512) void k_wgrad2_f16s(Wg2Args p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wa = wv & 1, wb = wv >> 1;                                  // this wavefront's 32 input / 32 output channels of the block's 64 x 128
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int half = local & 1, grp = (local >> 1) * 8 + xcd;             // the two halves of a group are 8 blocks apart: one XCD, one L2
    const int W = WT ? WT : p.W, H = WT ? WT : p.H, P = W + 1, SR = H + 1;      // (square images when WT is given)
    const int n0 = grp * p.per_group, n1 = min(n0 + p.per_group, p.n_images), nimg = max(n1 - n0, 0);
    const int S = nimg * SR * P;                                          // slots of this group's stream
    const int T = (S + 16 * NK - 1) / (16 * NK);                          // windows
    const int plane_x = (WT ? 16 * NK + WG_GX + 2 * (WT + 1) + 8 : p.rows_x) * 64, plane_y = 16 * NK * 64;
    const int buf_bytes = 4 * plane_x + 8 * plane_y;                      // XH0 XH1 XL0 XL1 | YH0..3 YL0..3
    for (int o = tid * 16; o < 2 * buf_bytes; o += 512 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);

    // ---- staging: item j of this thread = (slot, four channels) of the window; X first, then dY
    const float sx = p.x_tail[2], sy = p.dy_tail[2];
    const int nxs = WT ? 16 * NK + 2 * (WT + 1) + 2 : p.nxs;
    const float invP = 1.0f / (float)P, invSR = 1.0f / (float)SR;
    // X item j: slot row (tid >> 4) + 32 j of the window's nxs (rows past the last repeat it: same value, same address), channels
    // 4 (tid & 15) ..; dY item j: slot row (tid >> 5) + 16 j of 16 NK (exactly NK items per thread), channels 4 (tid & 31) ..
    const int xr0 = tid >> 4, xc4 = tid & 15, yr0 = tid >> 5, yc4 = tid & 31;
    const int xl0 = (xc4 >> 3) * plane_x + (WG_GX - 1) * 64 + 8 * (xc4 & 7), xcb = (64 * half + 4 * xc4) * 4;   // LDS (hi image) / channel byte offsets
    const int yl0 = (yc4 >> 3) * plane_y + 8 * (yc4 & 7), ycb = 16 * yc4;
    const long grp_elems = (long)n0 * H * W * 128;
    const unsigned grp_bytes = (unsigned)((long)nimg * H * W * 128 * 4);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + grp_elems), 0, grp_bytes, 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc((void *)(p.dy + grp_elems), 0, grp_bytes, 0x00020000);
    // byte offset of stream slot s in the group's tensor, or an offset no descriptor covers (zero column, zero row, outside the stream)
    auto slot_off = [&](int s) -> unsigned {
        // (24-bit multiplies: full rate, and every operand here is far below 2^24 -- the launcher checks the group's size)
        const int r = (int)(((float)s + 0.5f) * invP), x = s - __mul24(r, P);     // stream row, column
        const int i = (int)(((float)r + 0.5f) * invSR), y = r - __mul24(i, SR);   // image, row
        const bool ok = s >= 0 && s < S && x < W && y < H;
        return ((unsigned)(__mul24(__mul24(i, H) + y, W) + x) << 9) | (ok ? 0u : 0x80000000u);      // (no branch: a select of two constants)
    };
    float4 xv[MAXX], yv[NK];
    // AFF: scale and shift of the block's 64 input channels sit in LDS behind the two buffers (512 bytes; the kernel has no eight
    // registers to keep a thread's own for its life: they were spilled into the window loop)
    const unsigned char *aff_tab = smem + 2 * buf_bytes + 16 * xc4;
    if (AFF && tid < 32) *(float4 *)(smem + 2 * buf_bytes + 16 * tid) = *(const float4 *)((tid < 16 ? p.x_scale : p.x_shift - 64) + 64 * half + 4 * tid);
    unsigned okbits = 0u;                                                 // AFF: bit j = item j in the registers is a pixel (not padding)
    auto fetch_x = [&](int j, int t) {                                    // item j of window t -> registers
        const unsigned o = slot_off(t * 16 * NK - (P + 1) + min(xr0 + 32 * j, nxs - 1));
        if (AFF) okbits = (okbits & ~(1u << j)) | ((~o >> 31) << j);
        xv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, o + xcb, 0, 0));
    };
    auto fetch_y = [&](int j, int t) {
        yv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, slot_off(t * 16 * NK + yr0 + 16 * j) + ycb, 0, 0));
    };
    auto fetch = [&](int t) {                                             // a whole window (the prologue's two)
#pragma unroll
        for (int j = 0; j < MAXX; ++j) {
            fetch_x(j, t);
            __builtin_amdgcn_sched_barrier(0);                            // one item's address arithmetic at a time: its temporaries die with the load
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            fetch_y(j, t);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto store_x = [&](int j, unsigned char *B) {
        uint2 hi, lo;
        if (AFF) {
            const bool ok = (okbits >> j) & 1u;
            const float4 xsc = *(const float4 *)aff_tab, xsh = *(const float4 *)(aff_tab + 256);
            float4 v = xv[j];
            v.x = fmaxf(v.x * xsc.x + xsh.x, 0.f); v.y = fmaxf(v.y * xsc.y + xsh.y, 0.f);
            v.z = fmaxf(v.z * xsc.z + xsh.z, 0.f); v.w = fmaxf(v.w * xsc.w + xsh.w, 0.f);
            xv[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
        wg_split4(xv[j], sx, hi, lo);
        unsigned char *d = B + xl0 + min(xr0 + 32 * j, nxs - 1) * 64;
        *(uint2 *)d = hi; *(uint2 *)(d + 2 * plane_x) = lo;
    };
    auto store_y = [&](int j, unsigned char *B) {
        uint2 hi, lo;
        wg_split4(yv[j], sy, hi, lo);
        unsigned char *d = B + 4 * plane_x + yl0 + (yr0 + 16 * j) * 64;
        *(uint2 *)d = hi; *(uint2 *)(d + 4 * plane_y) = lo;
    };

    // ---- MFMA side (addressing as in k_wgrad_f16s): a transposing read of rows R .. R + 3
    const int i16 = lane & 15, g4 = lane >> 4;
    const int lane_off = (i16 >> 2) * 64 + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;
    const int xoff = WG_GX + P;                                           // X row of the pixel at dY row 0 of the window
    wg_f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    WG_STAMP(0, __builtin_amdgcn_s_memtime())
    WG_STAMP(4, __builtin_amdgcn_s_memrealtime())
    if (T > 0) fetch(0);
    __syncthreads();                                                      // the zero fill is complete
    if (T > 0) {
#pragma unroll
        for (int j = 0; j < MAXX; ++j) store_x(j, smem);
#pragma unroll
        for (int j = 0; j < NK; ++j) store_y(j, smem);
    }
    if (T > 1) fetch(1);
    __syncthreads();                                                      // window 0 is in buffer 0
    WG_STAMP(1, __builtin_amdgcn_s_memtime())
    for (int t = 0; t < T; ++t) {
#ifdef WG_SKEW
        // the two wavefronts of a SIMD (w and w + 4) leave the window's barrier together and run the same code: the second one waits a
        // little, so that its staging bursts fall into the first one's MFMA runs and the other way round
        if (wv >> 2) __builtin_amdgcn_s_sleep(WG_SKEW);
#endif
        unsigned char *B = smem + (t & 1) * buf_bytes, *Bn = smem + ((t + 1) & 1) * buf_bytes;
        const unsigned char *XH = B + wa * plane_x, *XL = XH + 2 * plane_x;
        const unsigned char *YH = B + 4 * plane_x + wb * plane_y, *YL = YH + 4 * plane_y;
        // the registers hold window t + 1: group g lays item g into the other buffer and requests the same item of window t + 2 into the
        // registers that just became free (a whole window ahead of its use; its address arithmetic sits in the shadow of the MFMAs)

#define WG_SB() __builtin_amdgcn_sched_barrier(0)
        // Software pipeline inside the wavefront, without a second fragment set: the three MFMAs of tap (dy, dx) are followed at once
        // by the reads of tap (next dy, dx) into the registers they have just consumed, so a fragment is requested two taps (192+
        // matrix cycles) before its use and the wavefront never sits between a read burst and its data.  (A group of twelve reads
        // in front of nine MFMAs, the first form of this kernel, left each wavefront ~300 cycles per group without an MFMA to
        // issue -- more than its partner's 288-cycle burst covers: 12.1 k cycles per window against 8.6 k of MFMAs.)
        wg_f16x8 ah[3], al[3], bh, bl;
#define WG2_LOADA(dxi, ks_, dyi_)                                                               \
        {                                                                                       \
            const int Rx = 16 * (ks_) + 8 * h + xoff + ((dyi_) - 1) * P + ((dxi) - 1);          \
            ah[dxi] = wg_cat(WG_TR(XH + Rx * 64 + lane_off), WG_TR(XH + (Rx + 4) * 64 + lane_off)); \
            al[dxi] = wg_cat(WG_TR(XL + Rx * 64 + lane_off), WG_TR(XL + (Rx + 4) * 64 + lane_off)); \
        }
#define WG2_LOADB(ks_)                                                                          \
        {                                                                                       \
            const int R = 16 * (ks_) + 8 * h;                                                   \
            bh = wg_cat(WG_TR(YH + R * 64 + lane_off), WG_TR(YH + (R + 4) * 64 + lane_off));    \
            bl = wg_cat(WG_TR(YL + R * 64 + lane_off), WG_TR(YL + (R + 4) * 64 + lane_off));    \
        }
        WG2_LOADB(0)
        WG2_LOADA(0, 0, 0) WG2_LOADA(1, 0, 0) WG2_LOADA(2, 0, 0)
        WG_SB();
#pragma unroll
        for (int g = 0; g < 3 * NK; ++g) {
            const int ks = g / 3, dyi = g - 3 * ks;
            const int gn = g + 1, ksn = gn / 3, dyn = gn - 3 * ksn;      // the next group of this window
            // this group's share of the staging work: item g of window t + 1 into the other buffer, the same item of window t + 2 requested
            // into the registers that have just become free (its address arithmetic runs in the shadow of the MFMAs)
            // (unconditionally: past the stream's end the store writes a buffer nobody reads and the request is answered with zeros --
            // a branch here would make the compiler wait for EVERY load in flight before each store, vmcnt(0), i.e. for the request
            // of the group before: one HBM latency per group)
            static_assert(3 * NK >= MAXX + NK, "a window has a (k-step, dy) group per staged item");
            if (g < MAXX) { store_x(g, Bn); fetch_x(g, t + 2); }
            else if (g < MAXX + NK) { store_y(g - MAXX, Bn); fetch_y(g - MAXX, t + 2); }
            WG_SB();
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int tp = 3 * dyi + dxi;
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dxi], bh, acc[tp], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dxi], bl, acc[tp], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[dxi], bh, acc[tp], 0, 0, 0);
                if (gn < 3 * NK) {
                    if (dxi == 2 && dyn == 0) WG2_LOADB(ksn)             // the k-step's last MFMA has been issued: its dY fragments may go
                    WG2_LOADA(dxi, ksn, dyn)
                }
                WG_SB();
            }
        }
#undef WG2_LOADA
#undef WG2_LOADB
#undef WG_SB
        __syncthreads();
    }
    WG_STAMP(2, __builtin_amdgcn_s_memtime())
    WG_STAMP(5, __builtin_amdgcn_s_memrealtime())
    WG_STAMP(6, (unsigned long long)T)
    // every wavefront owns its tiles: straight to the group's partial result, 128-byte rows
    const float inv = p.x_tail[3] * p.dy_tail[3];
    float *out = p.part + ((long)grp * 9 * 128 + 64 * half + 32 * wa) * 128 + 32 * wb + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[((long)t * 128 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128] = acc[t][r] * inv;
}

struct Wg2Shape { int nk, rows_x, nxs, lds, maxx; };
static bool wg2_shape(int h, int w, Wg2Shape &s)
{
    if (h < 1 || w < 3) return false;
    const int P = w + 1;
    static const int nk_max = getenv("SNK_WGRAD_NK") ? atoi(getenv("SNK_WGRAD_NK")) : 5;      // development: SNK_WGRAD_NK=4
    for (s.nk = nk_max >= 5 ? 5 : 4; s.nk >= 4; --s.nk) {
        s.nxs = 16 * s.nk + 2 * P + 2;
        s.rows_x = 16 * s.nk + WG_GX + 2 * P + 8;
        s.lds = 2 * (4 * s.rows_x * 64 + 8 * 16 * s.nk * 64);
        s.maxx = (s.nxs * 16 + 511) / 512;
        if (s.maxx <= 5 && s.lds <= 158 * 1024) return true;       // (+ 512 bytes in the deferred form: 160 KB per block is the limit)
    }
    return false;
}

struct WgShape { int P, RB, n_slabs, nk, rows_x, lds, nx_items, ny_items; };
static bool wg_shape(int h, int w, WgShape &s)
{
    if (h < 1 || w < 3) return false;
    s.P = w + 1;
    // slab height: the staging threads hold a slab's items in registers, both buffers share the LDS with nothing else
    for (s.n_slabs = 1; s.n_slabs <= h; ++s.n_slabs) {
        s.RB = (h + s.n_slabs - 1) / s.n_slabs;
        s.nx_items = (s.RB + 2) * w * 8;
        s.ny_items = s.RB * w * 8;
        s.nk = (s.RB * s.P + 15) / 16;
        s.rows_x = 16 * s.nk + WG_GX + 2 * s.P + 8;
        s.lds = 2 * 2 * 64 * (s.rows_x + 16 * s.nk);
        if (s.nx_items <= WG_MAXX * 256 && s.ny_items <= WG_MAXY * 256 && s.lds <= 150 * 1024) {
            if (s.lds < 4 * 9 * 1024 * 4) s.lds = 4 * 9 * 1024 * 4;          // or the four wavefronts' accumulators at the end
            return true;
        }
    }
    return false;
}

// which form runs: the window form (k_wgrad2_f16s) wherever the shape fits it, unless SNK_WGRAD=slabs asks for the slab form
static bool wg_use_windows(int height, int width)
{
    static const bool slabs = getenv("SNK_WGRAD") && !strcmp(getenv("SNK_WGRAD"), "slabs");
    Wg2Shape s2;
    return !slabs && wg2_shape(height, width, s2);
}

extern "C" long snk_conv3x3_wgrad_partials(int height, int width)
{
    WgShape s;
    Wg2Shape s2;
    const bool a = wg_shape(height, width, s), b = wg2_shape(height, width, s2);
    if (!a && !b) return -1;
    return (long)(b ? WG2_GROUPS : WG_GROUPS) * 9 * 128 * 128;             // the larger of the two forms' needs
}

static int wgrad_launch(const float *d_x, const float *d_dy, const float *d_x_tail, const float *d_dy_tail, float *d_partials,
                        float *d_dw, int n_images, int height, int width, void *stream, const float *d_x_scale,
                        const float *d_x_shift)
{
    SNK_REQUIRE((long)n_images * height * width * 128 < (1l << 40), "snk_conv3x3_wgrad_f16s: batch too large");
    if (wg_use_windows(height, width)) {
        Wg2Shape s2;
        wg2_shape(height, width, s2);
        const int per_group = (n_images + WG2_GROUPS - 1) / WG2_GROUPS;
        SNK_REQUIRE((long)per_group * height * width * 512 < (1l << 31) && (long)per_group * (height + 1) * (width + 1) < (1l << 22),
                    "snk_conv3x3_wgrad_f16s: %d images per group do not fit a buffer descriptor", per_group);
        Wg2Args a2 = {d_x, d_dy, d_partials, d_x_tail, d_dy_tail, n_images, height, width, per_group, s2.rows_x, s2.nxs, d_x_scale, d_x_shift};
        const dim3 g2(2 * WG2_GROUPS);
        const int wt = height == width && (width == 21 || width == 37) ? width : 0;      // the two canvases of the BASELINE configs
#define WG2_LAUNCH(NK_, MX_, WT_)                                                               \
        {                                                                                       \
            static bool attr_ = false;       /* more than 64 KB of dynamic LDS needs the attribute */ \
            if (!attr_) {                                                                       \
                SNK_CHECK_HIP(hipFuncSetAttribute((const void *)k_wgrad2_f16s<NK_, MX_, WT_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                SNK_CHECK_HIP(hipFuncSetAttribute((const void *)k_wgrad2_f16s<NK_, MX_, WT_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                attr_ = true;                                                                   \
            }                                                                                   \
            if (d_x_scale) k_wgrad2_f16s<NK_, MX_, WT_, true><<<g2, 512, s2.lds + 512, (hipStream_t)stream>>>(a2); \
            else k_wgrad2_f16s<NK_, MX_, WT_, false><<<g2, 512, s2.lds, (hipStream_t)stream>>>(a2); \
        }
        if (wt == 21 && s2.nk == 5 && s2.maxx <= 4) WG2_LAUNCH(5, 4, 21)
        else if (wt == 21 && s2.nk == 4 && s2.maxx <= 4) WG2_LAUNCH(4, 4, 21)
        else if (wt == 37 && s2.nk == 4 && s2.maxx <= 5) WG2_LAUNCH(4, 5, 37)
        else if (s2.nk == 5 && s2.maxx <= 4) WG2_LAUNCH(5, 4, 0)
        else if (s2.nk == 5) WG2_LAUNCH(5, 5, 0)
        else if (s2.maxx <= 4) WG2_LAUNCH(4, 4, 0)
        else WG2_LAUNCH(4, 5, 0)
#undef WG2_LAUNCH
        k_wgrad_fold<<<9 * 128 * 128 / 64, 256, 0, (hipStream_t)stream>>>(d_partials, WG2_GROUPS, d_dw);
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    WgShape s;
    SNK_REQUIRE(!d_x_scale, "snk_conv3x3_wgrad_f16s_deferred: %d x %d images run the slab form, which has no deferred input", height, width);
    SNK_REQUIRE(wg_shape(height, width, s), "snk_conv3x3_wgrad_f16s: %d x %d images are not supported (width 3 .. 96)", height, width);
    WgArgs a = {d_x, d_dy, d_partials, d_x_tail, d_dy_tail, n_images, height, width, s.P, s.RB, s.n_slabs, s.nk, s.rows_x,
                (n_images + WG_GROUPS - 1) / WG_GROUPS, s.nx_items, s.ny_items};
    k_wgrad_f16s<<<16 * WG_GROUPS, 512, s.lds, (hipStream_t)stream>>>(a);
    k_wgrad_fold<<<9 * 128 * 128 / 64, 256, 0, (hipStream_t)stream>>>(d_partials, WG_GROUPS, d_dw);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_wgrad_f16s(const float *d_x, const float *d_dy, const float *d_x_tail, const float *d_dy_tail,
                                      float *d_partials, float *d_dw, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_dy && d_x_tail && d_dy_tail && d_partials && d_dw && n_images > 0, "snk_conv3x3_wgrad_f16s: bad argument");
    return wgrad_launch(d_x, d_dy, d_x_tail, d_dy_tail, d_partials, d_dw, n_images, height, width, stream, nullptr, nullptr);
}

// The weight gradient of a residual block's second layer when the activation between the block's two convolutions was never
// written (deferred batch norm, snake_engine/train_step.py): d_y_below is the PRE-batch-norm output of the layer below and every
// value is taken as relu(y * scale[c] + shift[c]) -- bit for bit what snk_bn_train_apply would have written -- when it is laid
// into LDS.  d_x_tail: the range of that activation (snk_bn_train_finalize_range).  Window form only
// (snk_train_deferred_bn_supported).
extern "C" int snk_conv3x3_wgrad_f16s_deferred(const float *d_y_below, const float *d_scale, const float *d_shift, const float *d_dy,
                                               const float *d_x_tail, const float *d_dy_tail, float *d_partials, float *d_dw,
                                               int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_y_below && d_scale && d_shift && d_dy && d_x_tail && d_dy_tail && d_partials && d_dw && n_images > 0,
                "snk_conv3x3_wgrad_f16s_deferred: bad argument");
    return wgrad_launch(d_y_below, d_dy, d_x_tail, d_dy_tail, d_partials, d_dw, n_images, height, width, stream, d_scale, d_shift);
}

// 1 when the training step may defer the batch norm + ReLU of a residual block's first layer into the kernels that read it
// (the weight gradient's window form carries the input transform; the slab form does not)
extern "C" int snk_train_deferred_bn_supported(int height, int width) { return wg_use_windows(height, width) ? 1 : 0; }
