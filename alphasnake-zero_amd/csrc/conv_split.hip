// conv_split.hip -- the Q-net's 3x3 residual-tower layer (alpha_nnet.py:25-47) at float32 accuracy on the
// f16 matrix pipe of gfx950.
//
// Every float32 operand v is carried as two f16 numbers, hi = f16(v) and lo = f16(v - hi) (22 significand bits
// together); a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32 accumulation (the lo*lo
// term is below float32 resolution).  Three 32-cycle MFMAs of k = 16 replace eight 64-cycle f32 MFMAs of k = 2:
// 5.3x the f32 MFMA rate, 2.6x the Winograd kernel's arithmetic rate, at the same |dQ| (about 3e-7 for the whole
// net, tests/test_net_gpu.py).  Weights are pre-scaled by a power of two so that their lo parts are normal f16
// numbers; the inverse goes into the batch-norm scale (exact).  Activations are multiplied by a caller-chosen power
// of two on the way into LDS (so that their lo parts are normal numbers too) and clamped to +-65504 after it; a clamp
// sets the layer's range flag (a word behind the scales in the weight image) so that the host can refuse the result.
//
// GEMM rows are the image's pixels in row-major order, 32 per M tile (14 tiles at 21x21, 7 junk rows).  One block =
// up to 8 consecutive M tiles of one image (7 + 7 at 21x21) x all 128 outputs, 4 wavefronts, at most 128 accumulator
// registers per wave, so TWO blocks share a CU (2 waves per SIMD, 68 KB of LDS each): while one is in its prologue or
// epilogue the other one's MFMAs keep the matrix pipe busy.
// The image rows a block needs (those of its pixels, one above, one below) sit in LDS with a zero border, rows pitched
// W + 1 so that the right border of a row is the left border of the next: the input of output pixel (y, x) for tap
// (dy, dx) is at LDS position of (y, x) + dy (W+1) + dx, so all nine taps of an MFMA A-fragment are the SAME LDS image
// at nine constant offsets from the lane's pixel address -- the rows are staged (and split into hi/lo) once per
// 16-channel chunk, not once per tap.  Wave wn owns outputs 32 wn .. 32 wn + 31 for all M tiles; per (chunk, tap) and M
// tile it reads two ds_read_b128 (hi, lo) for three MFMAs, two tiles ahead of their use.  B fragments (the pre-split weights, 576 KB,
// L2 resident, stored in fragment order so that a wave's load is 1 KB contiguous) stream straight into a 3-deep
// register ring two taps ahead; the next chunk's strip is loaded at the chunk's first tap, split and written to the
// other LDS buffer in the chunk's last MFMA regions; one block barrier per chunk.  Epilogue: accumulators -> LDS
// (64 rows at a time, double-buffered), read back as float4 rows, * scale + shift + residual, ReLU, whole 512-byte
// pixel rows written; the residual loads of the next 64 rows are in flight meanwhile.  In the tower's last layer the
// epilogue also reduces the head's 1x1 convolution per pixel and skips the layer output.  Workgroups map to (image,
// part) XCD-aware: the parts of an image, which share halo rows, run on the same XCD (same L2) back to back.
// The block body (hs_block) has two callers: k_conv3x3_f16s (whole images) and k_conv3x3_f16s_rect (the part of an image
// that depends on the state: the first tower layers of the self-play path, see the comment above that kernel).
#include "common.h"
#include "train_fold.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// the 16-bit activation type of the reduced-precision towers is f16 or (BF) bf16; both travel through the kernel as 16-bit
// patterns typed _Float16 (staging copies them untouched), only the MFMA and these two conversions know the difference
template <bool BF> __device__ __forceinline__ float hs_from16(_Float16 v)
{
    if (BF) return (float)__builtin_bit_cast(__bf16, v);
    return (float)v;
}
template <bool BF> __device__ __forceinline__ _Float16 hs_to16(float v)
{
    if (BF) return __builtin_bit_cast(_Float16, (__bf16)v);            // v_cvt_pk_bf16_f32: round to nearest even, float32's range
    return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);    // saturating: an f16 infinity would turn the next layer into NaNs
}

#define HS_C 128
#define HS_KC 16                                         // input channels per chunk = one MFMA k step
#define HS_LDP 80                                        // bytes per LDS pixel: [hi k0-15 | lo k0-15] + 16 (16 x odd: conflict-free b128)
#define HS_NPB 352                                       // pixels per LDS buffer
#define HS_MLD 132                                       // epilogue row (floats): 528 bytes = 16 x 33
#define HS_SMEM_STAGE (2 * HS_BUF16)                     // 65 536 bytes (split form: 2 x 352 x 80 = 56 320)
#define HS_SMEM_EPI (2 * 64 * HS_MLD * 4)                // 67 584 bytes
#ifdef HS_ONE_PER_CU      // development build only: a block alone on its compute unit (what does the co-resident block cost it?)
#define HS_SMEM (96 * 1024)
#else
#define HS_SMEM (HS_SMEM_EPI > HS_SMEM_STAGE ? HS_SMEM_EPI : HS_SMEM_STAGE)
#endif
#define HS_WS_ELEMS (9 * HS_C * HS_C * 2)                // f16 numbers in the split weight image; four floats follow it
#define HS_RING 3
#define HS_AHEAD 2
#ifndef HS_XLOAD_TAP
#define HS_XLOAD_TAP 0                                   // the tap at whose start the next chunk's pixel loads are issued (A/B: make variant EXTRA=-DHS_XLOAD_TAP=1)
#endif
#define HS_NST 5                                         // staging items per thread and chunk: 64 pixels x 4 float4 each
// the frame of the towers with 16-bit activations (IO16 bit 0; BASELINE configs[4]): a chunk is 32 input channels = two MFMA k
// steps, so the 80-byte LDS pixel holds [channels 0-15 | 16-31] where the split form keeps [hi | lo]: half the chunks, barriers and
// A-pipeline restarts per block, and a chunk's MFMAs (2 x 9 x NI) outlast the HBM latency of the next chunk's pixel loads even
// when the block has the SIMDs to itself (with 16-channel chunks a block's chunk loop was bound by that latency: 3.7 k cycles per
// chunk for 1.4 k of MFMA issue, profiles/r5_a16_stamps.log).  No lo parts: the same LDS bytes hold more pixels (8-tile blocks on a
// 37-pixel-wide canvas) and a staging item is one 16-byte load and one ds_write_b128.
#define HS_NST16 6                                       // 16-bit frame: staging items per thread and chunk (64 pixels x 4 pieces of 16 bytes each)
#define HS_BUF16 32768                                   // 16-bit frame: bytes per LDS buffer.  Its rows are pitched in BYTES:
//   80 W + 256 (whole images: the 256-byte gap holds the zero border on both sides) or 80 (w + 2) + 96 (sub-rectangles: real halo
//   columns), so that stepping from a row's last pixel to the next row's first moves the address by 336 bytes = 21 bank quads = 5
//   mod 16, exactly like a step inside a row: the sixteen lanes a ds_read_b128 serves together never share a bank quad.  (With
//   rows pitched W + 1 pixels a lane group that straddles a row end spans 17 slots and two lanes collide: 37 % of this frame's
//   LDS cycles were conflict cycles, profiles/r5_a16_sq_counters_before.json, and the LDS, which serves a fragment read per MFMA
//   here, is what the co-resident block's epilogue and prologue wait for.)
#define HS_GAP16_FULL 256
#define HS_GAP16_RECT 96
#ifndef HS_PRIO
#define HS_PRIO 3
#endif
#ifndef HS_RING16
#define HS_RING16 3                                      // 16-bit frame: B-fragment ring (A/B: make variant EXTRA="-DHS_RING16=4 -DHS_AHEAD16=3")
#define HS_AHEAD16 2
#endif

struct ConvHsArgs {
    const float *x;            // [n][Hd][Wd][128]
    const f16x8 *wS;           // [chunk 8][tap 9][wn 4][hi/lo][lane 64] x 8 f16
    const float *wscale_inv;   // tail of the weight image: {2^-k (weights), 2^k, activation scale s (a power of two), 1 / s,
                               //   int32 range flag (set by the kernel when an input was clamped), 3 words of padding}
    const float *scale, *shift;
    const float *res;          // or NULL
    float *out;                // or NULL when only the fused 1x1 head output is wanted
    const float *w1x1;         // optional fused head (alpha_nnet.py:49-50): h1[pixel] = relu(dot(out[pixel][:], w1x1) * s1 + b1)
    float *h1;                 // [n][Hd * Wd], written when w1x1 is given
    float s1, b1;
    int Hd, Wd, n_blk, tiles_base, tiles_rem, relu;      // block b of an image has tiles_base + (b < tiles_rem) M tiles
    int n_img_grouped;         // images (a multiple of 8) that use the XCD-aware block order
    const float *center;       // MODE 4 (training forward): per-channel centre of the batch-norm sums (or NULL = 0)
    float *stat_part;          // MODE 4: [gridDim.x][2][128] sums of (out - center) and (out - center)^2 over the block's pixels
    // sub-rectangle form (k_conv3x3_f16s_rect): one descriptor per block, written on the device by k_rect_plan
    const uint4 *desc;         // { image, y0 | x0 << 8 | h << 16 | w << 24, tile0 | ntile << 8 | part << 16 | parts << 24, bounding box }
    const int *n_desc;         // number of descriptors (blocks past it leave at once)
    const float *bg_out;       // or NULL; [Hd][Wd][128]: this layer's output on an all-background image, copied into every pixel
                               //   of the canvas outside the rectangle (the layer's readers are full layers)
    const float *bg_in;        // or NULL; the input's background image: x is valid on the bounding box grown by grow_in (cut to the
    const float *bg_res;       //   canvas) and stale outside, where the producing layer's constant is read instead; the same for res
    int grow_in, grow_res;
    // MODE 5 (training step, input gradient): the batch-norm backward sums of the layer whose output gradient this launch produces
    const float *g_y;          // that layer's pre-batch-norm output [n][Hd][Wd][128]
    const unsigned char *g_mask;   // its ReLU bits, one byte per quad of channels (csrc/train.hip)
    const float *g_inv;        // its 1 / sigma per channel (center = its batch mean)
    // MODE 6 / 7 (training step, a block's first layer whose batch norm + ReLU is never written out, see MODE 6 below): the
    // producer's batch-norm scale and shift per channel -- MODE 6 applies relu(x * aff_scale + aff_shift) to every staged input,
    // MODE 7 takes the ReLU decision of g_y from the same expression instead of from mask bytes
    const float *aff_scale, *aff_shift;
    float *amax_part;          // MODE 4 (optional): [gridDim.x][128] largest |out - center| per channel over the block's pixels
    const unsigned char *res_mask;     // MODE 8: ReLU bits (one byte per quad of channels) the shortcut rows are taken through
};

#ifdef HS_STAMPS       // development build only (tools/conv_stamps.py): s_memtime at the phase boundaries of every block
__device__ unsigned long long hs_stamp_buf[16384 * 8];
#define HS_STAMP(k) if (tid == 0 && blockIdx.x < 16384) hs_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();
#define HS_STAMP_REAL(k) if (tid == 0 && blockIdx.x < 16384) hs_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
// where the block runs: HW_ID (hwreg 4: wave slot, SIMD, CU, SH, SE, workgroup slot) | XCC_ID (hwreg 20) << 32
#define HS_STAMP_HWID(k) if (tid == 0 && blockIdx.x < 16384) hs_stamp_buf[blockIdx.x * 8 + (k)] = \
        (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
// per-tap stamps of wave 0 of the first 2 048 blocks: [block][chunk 0..7][tap 0..8 + chunk end]
__device__ unsigned long long hs_tap_buf[2048 * 8 * 10];
#define HS_TSTAMP(c_, s_) if (tid == 0 && blockIdx.x < 2048 && (c_) < 8) hs_tap_buf[(blockIdx.x * 8 + (c_)) * 10 + (s_)] = __builtin_amdgcn_s_memtime();
extern "C" int snk_dbg_conv_tap_stamps(unsigned long long *h_out)
{
    SNK_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(hs_tap_buf), sizeof(unsigned long long) * 2048 * 8 * 10));
    return 0;
}
extern "C" int snk_dbg_conv_stamps(unsigned long long *h_out, int n_blocks)
{
    SNK_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(hs_stamp_buf), (size_t)n_blocks * 8 * sizeof(unsigned long long)));
    return 0;
}
#else
#define HS_STAMP(k)
#define HS_STAMP_REAL(k)
#define HS_STAMP_HWID(k)
#define HS_TSTAMP(c_, s_)
#endif

// NI: M tiles (of 32 GEMM rows) per block, <= 8.  MODE fixes the epilogue's options at compile time (no branches per row):
//   0 all read from the arguments; 1 ReLU, no residual; 2 ReLU + residual; 3 ReLU + residual + fused head, no layer output;
//   4 the training step's forward pass: the bare convolution (no scale / shift / ReLU / residual) plus the per-channel sums its
//     batch norm needs, taken from the values on their way out (one pass over the activation less per layer and step)
//   5 the training step's input gradient: the bare convolution (+ the shortcut's gradient as residual) = the gradient at the previous
//     layer's output, plus the two per-channel sums that layer's batch-norm backward needs -- sum(g), sum(g xhat), g = the gradient
//     where the layer's ReLU let the value through -- taken from the values on their way out (snk_bn_train_grad_sums_f64's pass saved)
//   6 MODE 4 whose INPUT is the previous layer's pre-batch-norm output: relu(x * aff_scale + aff_shift) -- that layer's batch norm
//     and ReLU, the very expression of k_bn_apply (csrc/train.hip) -- is applied to every value on its way into LDS, so the
//     activation between the two convolutions of a residual block is never written to HBM nor read back (a pass over two
//     462 MB tensors less per block and step); the zero border stays zero (it is the padding of the ACTIVATION)
//   7 MODE 5 for such a layer below: its ReLU decision is y * aff_scale + aff_shift > 0, recomputed from the g_y values the sums
//     read anyway (no mask bytes exist for it)
//   8 MODE 5 whose shortcut gradient is res WHERE res_mask's bit is set: res is then the gradient at the block's OUTPUT as the layer
//     above received it, res_mask that output's ReLU bits -- the batch-norm backward of the block's second layer no longer writes
//     the masked copy (462 MB per block) just for this epilogue to read it back.  out may be res itself (a thread reads the rows
//     it writes, and reads them first)
//   9 MODE 6 that also takes MODE 4's per-channel maxima (its own batch norm is deferred too: the layer above a deferred stem)
//  10 MODE 8 whose ReLU decision for the sums is recomputed like MODE 7's (the layer below is the deferred stem: no mask bytes)
#define hs_dpp(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
// SPLIT = false: the reduced-precision form for BASELINE configs[4] ("bf16 MFMA conv"): the same kernel with the hi parts
//   only, one MFMA per product instead of three (f16 operands: 11 significand bits against bf16's 8, float32 accumulate).
// IO16 (reduced-precision form only): bit 0 = the input and the residual are f16 arrays [n][H][W][128], bit 1 = the output
//   is written as f16.  The activations of BASELINE configs[4]'s tower then cost 2 bytes in HBM instead of 4 (the f16 form
//   with float32 activations is HBM-bound), staging is a plain copy (the values are f16 already: no scale, no clamp, no
//   conversion) and the range guard has nothing to watch.
// RECT: the block's "image" is a sub-rectangle (ry0, rx0, rh, rw) of the Hd x Wd canvas (see k_rect_plan below): GEMM rows are
//   the rectangle's pixels in row-major order, the LDS rows are pitched rw + 2 with the halo columns taken from the canvas
//   (zero outside it), global rows stay pitched Wd.
// BF (with IO16 bit 0, reduced-precision form only): the 16-bit type is bf16 -- BASELINE configs[4]'s "bf16 MFMA conv" as it is
//   worded: bf16 activations in HBM, bf16 weights, v_mfma_f32_32x32x16_bf16, float32 accumulation and epilogue.  bf16 has
//   float32's exponent range, so there is no scale to choose and nothing to clamp.
template <int NI, int MODE, bool SPLIT, int IO16, bool RECT, bool BF = false>
__device__ __forceinline__ void hs_block(const ConvHsArgs &p, unsigned char *smem, const int img, const int tile0, const int ntile,
                                         const int ry0, const int rx0, const int rh, const int rw, const unsigned bbox,
                                         const int part, const int parts)
{
    static_assert(IO16 == 0 || !SPLIT, "f16 activations only exist in the reduced-precision form");
    static_assert(!BF || (IO16 & 1), "the bf16 form reads bf16 activations");
    constexpr bool IN16 = (IO16 & 1) != 0, OUT16 = (IO16 & 2) != 0;
    constexpr bool STATS = MODE == 4 || MODE == 6 || MODE == 9, GSTATS = MODE == 5 || MODE == 7 || MODE == 8 || MODE == 10, BARE = STATS || GSTATS;
    constexpr bool RESMASK = MODE == 8 || MODE == 10, AMAX = MODE == 4 || MODE == 9;
    constexpr bool AFF = MODE == 6 || MODE == 9, GAFF = MODE == 7 || MODE == 10, NORES = MODE == 7;
    static_assert(!(AFF || GAFF || RESMASK) || (SPLIT && IO16 == 0 && !RECT), "the deferred batch norm exists in the training step's float32 form only");
    constexpr bool K32 = IN16;
    constexpr int RING = K32 ? HS_RING16 : HS_RING, AHEAD = K32 ? HS_AHEAD16 : HS_AHEAD;      // B-fragment register ring: slots (a divisor of 9), taps ahead
    constexpr bool TWO = SPLIT || K32;                     // two A fragments / two B fragments per (tap, M tile)
    constexpr int KC = K32 ? 32 : HS_KC, NCHUNK = HS_C / KC;
    constexpr int NST = K32 ? HS_NST16 : HS_NST, BUFB = K32 ? HS_BUF16 : HS_NPB * HS_LDP;
    constexpr int PIECE = K32 ? 8 : 4;                     // channels of a staging item
    constexpr bool P8 = IN16 && OUT16;                     // the epilogue handles 8 channels (16 bytes in and out) per thread and row
    // (MODE 7 never takes a shortcut's gradient: it produces the gradient at a block's FIRST layer's output -- its epilogue has the
    // registers the shortcut rows would take for the g_y rows instead, requested a pass ahead)
    const bool has_res = NORES ? false : RESMASK ? true : (MODE == 0 || GSTATS) ? p.res != nullptr : (MODE == 2 || MODE == 3);
    const bool has_head = MODE == 0 ? p.w1x1 != nullptr : MODE == 3;
    const bool has_out = MODE == 0 ? p.out != nullptr : MODE != 3;
    const float relu_floor = ((MODE != 0 && !BARE) || (MODE == 0 && p.relu)) ? 0.f : -__builtin_inff();
    (void)has_head; (void)has_out; (void)relu_floor;       // (the input-gradient epilogue uses none of them)
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;        // wave wn owns outputs 32 wn .. 32 wn + 31
    const int h = lane >> 5, l31 = lane & 31;
    const int Wr = RECT ? rw : p.Wd;                       // width of the GEMM's image
    const int cy0 = RECT ? ry0 : 0, cx0 = RECT ? rx0 : 0;  // its origin on the canvas
    const int HW = RECT ? rh * rw : p.Hd * p.Wd;           // its pixels = GEMM rows
    const int HWc = p.Hd * p.Wd;                           // the canvas: what the tensors in HBM are pitched by
    const int m0 = 32 * tile0, m1 = min(32 * (tile0 + ntile), HW);      // the block's GEMM rows = pixels m0 .. m1 - 1 of the image
    const int P = RECT ? Wr + 2 : Wr + 1;                  // padded pitch (pixels)
    const int PB = K32 ? (RECT ? (Wr + 2) * HS_LDP + HS_GAP16_RECT : Wr * HS_LDP + HS_GAP16_FULL) : P * HS_LDP;   // LDS row pitch (bytes)
    const float invW = 1.0f / (float)Wr;
    const int y_first = (int)(((float)m0 + 0.5f) * invW), y_last = (int)(((float)(m1 - 1) + 0.5f) * invW);
    // LDS row r holds image row y_first - 1 + r (zero when outside the canvas), LDS column x + 1 image column x

    // ---- staging role: item k of this thread = pixel (tid / 4 + 64 k) of the strip (the rows and columns of the canvas the
    //      block's taps reach), float4 (tid % 4) of the chunk
    const int ya = max(cy0 + y_first - 1, 0), yb = min(cy0 + y_last + 1, p.Hd - 1);          // canvas rows of the strip
    const int xa = RECT ? max(cx0 - 1, 0) : 0, xb = RECT ? min(cx0 + Wr, p.Wd - 1) : p.Wd - 1;   // its canvas columns
    const int ry_lo = ya - (cy0 + y_first - 1), cx_lo = RECT ? xa - (cx0 - 1) : 1;             // LDS row / column of (ya, xa)
    const int ws = xb - xa + 1;
    const int npx = (yb - ya + 1) * ws;
    const float invWs = RECT ? 1.0f / (float)ws : invW;
    const float *xrow = p.x + (((long)img * p.Hd + ya) * p.Wd + xa) * HS_C + PIECE * (tid & 3);
    const _Float16 *xrow16 = (const _Float16 *)p.x + (((long)img * p.Hd + ya) * p.Wd + xa) * HS_C + PIECE * (tid & 3);
    const int pix0 = tid >> 2;
    const float xs = p.wscale_inv[2];                      // activations are multiplied by this power of two before the split
    unsigned ldo[NST];                                     // LDS byte offset (inside a buffer) of the thread's items
    unsigned gof[NST];                                     // element offset of the items in x (chunk 0)
    const float *gp[NST];                                  // RECT: the items' addresses (chunk 0): in x where the producing layer wrote
    const _Float16 *gp16[NST];                             //   (bounding box grown by grow_in), else in that layer's background image
    const int by0 = bbox & 255, bx0 = (bbox >> 8) & 255, by1 = (bbox >> 16) & 255, bx1 = bbox >> 24;
    const bool sel_in = RECT && p.bg_in != nullptr;
    // items past the strip's last pixel repeat it (same value to the same LDS address): no predication, no branches
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int pix_ = min(pix0 + 64 * k, npx - 1);
        const int r_ = (int)(((float)pix_ + 0.5f) * invWs), x_ = pix_ - r_ * ws;
        ldo[k] = (ry_lo + r_) * PB + (cx_lo + x_) * HS_LDP + (tid & 3) * (K32 ? 16 : 8);
        gof[k] = (r_ * p.Wd + x_) * HS_C;
        if (RECT) {
            const int Y = ya + r_, X = xa + x_;
            const bool stale = sel_in && (Y < by0 - p.grow_in || Y > by1 + p.grow_in || X < bx0 - p.grow_in || X > bx1 + p.grow_in);
            const long o_ = (long)(Y * p.Wd + X) * HS_C + PIECE * (tid & 3);
            gp[k] = (stale ? p.bg_in : p.x + (long)img * HWc * HS_C) + o_;
            gp16[k] = (stale ? (const _Float16 *)p.bg_in : (const _Float16 *)p.x + (long)img * HWc * HS_C) + o_;
        }
    }
    float4 st[K32 ? 1 : NST];
    f16x4 st16[K32 ? 1 : NST];
    f16x8 st8[K32 ? NST : 1];                              // the 16-bit frame's items: eight channels, staged as they are
    f16x4 hi_t;
    float4 d_t;
    float4 asc = make_float4(1.f, 1.f, 1.f, 1.f), ash = make_float4(0.f, 0.f, 0.f, 0.f);    // MODE 6: the producer's scale / shift of the
    //   four channels this thread stages in the chunk being split; all 128 of each sit in LDS behind the two staging buffers
#define HS_AFF_READ(c) { asc = *(const float4 *)(smem + 2 * BUFB + (16 * (c) + 4 * (tid & 3)) * 4);      \
                         ash = *(const float4 *)(smem + 2 * BUFB + 512 + (16 * (c) + 4 * (tid & 3)) * 4); }
    float amax = 0.f;                                      // largest |scaled input| this thread staged: 65504 = something was clamped
#define HS_LOAD(c) _Pragma("unroll") for (int k_ = 0; k_ < NST; ++k_) {                         \
        if (K32) st8[k_] = *(const f16x8 *)((RECT ? gp16[k_] : xrow16 + gof[k_]) + KC * (c));   \
        else if (IN16) st16[k_] = *(const f16x4 *)((RECT ? gp16[k_] : xrow16 + gof[k_]) + KC * (c)); \
        else st[k_] = *(const float4 *)((RECT ? gp[k_] : xrow + gof[k_]) + KC * (c)); }
// split of one staged float4 in two halves that sit in different MFMA regions (a region hides about 15 VALU instructions):
//   A: clamp to the f16 range, hi = f16(v), d = v - hi;   B: lo = f16(d), both written to LDS
#define HS_SPLIT_A(kk)                                                                          \
    if (K32) { } else if (IN16) { hi_t = st16[kk]; } else                                       \
    {                                                                                           \
        float4 v_ = st[kk];                                                                     \
        if (AFF) {          /* the producer's batch norm + ReLU, as k_bn_apply computes it (multiply, then add: no fma) */ \
            v_.x = fmaxf(v_.x * asc.x + ash.x, 0.f); v_.y = fmaxf(v_.y * asc.y + ash.y, 0.f);   \
            v_.z = fmaxf(v_.z * asc.z + ash.z, 0.f); v_.w = fmaxf(v_.w * asc.w + ash.w, 0.f);   \
        }                                                                                       \
        v_.x *= xs; v_.y *= xs; v_.z *= xs; v_.w *= xs;                                         \
        v_.x = __builtin_amdgcn_fmed3f(v_.x, -65504.f, 65504.f); v_.y = __builtin_amdgcn_fmed3f(v_.y, -65504.f, 65504.f); \
        v_.z = __builtin_amdgcn_fmed3f(v_.z, -65504.f, 65504.f); v_.w = __builtin_amdgcn_fmed3f(v_.w, -65504.f, 65504.f); \
        amax = fmaxf(fmaxf(amax, fabsf(v_.x)), fabsf(v_.y)); amax = fmaxf(fmaxf(amax, fabsf(v_.z)), fabsf(v_.w)); /* 2 v_max3 */ \
        hi_t[0] = (_Float16)v_.x; hi_t[1] = (_Float16)v_.y; hi_t[2] = (_Float16)v_.z; hi_t[3] = (_Float16)v_.w; \
        if (SPLIT) {                                                                            \
            d_t.x = v_.x - (float)hi_t[0]; d_t.y = v_.y - (float)hi_t[1];                       \
            d_t.z = v_.z - (float)hi_t[2]; d_t.w = v_.w - (float)hi_t[3];                       \
        }                                                                                       \
    }
#define HS_SPLIT_B(k_, bufoff)                                                                  \
    if (K32) { *(f16x8 *)(smem + (bufoff) + ldo[k_]) = st8[k_]; } else                          \
    {                                                                                           \
        unsigned char *d_ = smem + (bufoff) + ldo[k_];                                          \
        *(f16x4 *)d_ = hi_t;                                                                    \
        if (SPLIT) {                                                                            \
            f16x4 lo_;                                                                          \
            lo_[0] = (_Float16)d_t.x; lo_[1] = (_Float16)d_t.y; lo_[2] = (_Float16)d_t.z; lo_[3] = (_Float16)d_t.w; \
            *(f16x4 *)(d_ + 32) = lo_;                                                          \
        }                                                                                       \
    }

    HS_STAMP(0)
    HS_STAMP_HWID(7)
    const f16x8 *wl = p.wS + wn * 128 + lane;              // this wave's fragments of global step g = 9 chunk + tap: wl[g * 512 + {0 hi, 64 lo}]
    f16x8 Bq[RING][2];
#define HS_LOADB(slot, g)                                                                       \
    {                                                                                           \
        const f16x8 *w_ = wl + (long)(g) * 512;                                                 \
        Bq[slot][0] = w_[0];                                                                    \
        if (TWO) Bq[slot][1] = w_[64];                                                          \
    }
    HS_LOAD(0)
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) HS_LOADB(s, s);
    if (K32) {
        // the 16-bit frame zeroes only what is read and never staged (both buffers): per LDS row the left border / halo slot and the
        // 176 bytes from slot Wr + 1 to the end of the row (the right border; a halo column inside the canvas is staged over it
        // after the barrier), and the rows above / below the canvas -- 4-6 stores per thread instead of 16
        const int R = y_last - y_first + 3;
        for (int i = tid; i < 2 * 16 * R; i += 256) {
            const int b = i >= 16 * R, j = i - b * 16 * R, r = j >> 4, q = j & 15;
            *(uint4 *)(smem + b * BUFB + r * PB + (q < 5 ? q * 16 : (Wr + 1) * HS_LDP + (q - 5) * 16)) = make_uint4(0u, 0u, 0u, 0u);
        }
        const bool top = ry_lo > 0, bot = cy0 + y_last + 1 > p.Hd - 1;
        if (top || bot)
            for (int i = tid; i < 2 * (PB >> 4); i += 256) {
                const int b = i >= (PB >> 4), q = i - b * (PB >> 4);
                if (top) *(uint4 *)(smem + b * BUFB + q * 16) = make_uint4(0u, 0u, 0u, 0u);
                if (bot) *(uint4 *)(smem + b * BUFB + (R - 1) * PB + q * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
    } else
    for (int o = tid * 16; o < 2 * BUFB; o += 256 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // borders stay zero
    if (AFF && tid < 64) *(float4 *)(smem + 2 * BUFB + 16 * tid) = *(const float4 *)((tid < 32 ? p.aff_scale : p.aff_shift - HS_C) + 4 * tid);
    f32x16 acc[NI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    __syncthreads();
    if (AFF) HS_AFF_READ(0)
#pragma unroll
    for (int k = 0; k < NST; ++k) { HS_SPLIT_A(k) HS_SPLIT_B(k, 0) }
    __syncthreads();

    unsigned la[NI];                                       // LDS byte address (buffer 0, centre tap) of the lane's pixel in M tile i
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m_ = min(m0 + 32 * i + l31, HW - 1);     // rows past the image repeat its last pixel (computed, never stored)
        const int y_ = (int)(((float)m_ + 0.5f) * invW), x_ = m_ - y_ * Wr;
        la[i] = (unsigned)((y_ - (y_first - 1)) * PB + (x_ + 1) * HS_LDP) + 16 * h;
    }
#define HS_LDS(off) (*(const f16x8 *)(smem + (off)))
#define HS_MFMA(a, b, c) c = BF ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0) \
                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// One tap of a chunk = NI fenced regions, each = { the ds_reads of the tile two ahead; a share of the staging work;
// this tile's three MFMAs }.  The fences keep the A fragments two tiles ahead of their use (the compiler otherwise
// sinks the reads to the MFMAs that consume them and the wave eats the LDS latency once per tile).  The tile sequence
// runs across taps: tile t of the chunk is (tap t / NI, M tile t % NI).  Staging of the next chunk: HS_NST global
// loads at tap 0, their split halves (A, B) in the chunk's last 2 HS_NST regions.
#define HS_AOFF(t) (la[(t) % NI] + rb + (unsigned)((((t) / NI) / 3 - 1) * PB + (((t) / NI) % 3 - 1) * HS_LDP))
#define HS_TAP(s, MORE)                                                                         \
        {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                    \
                const int t2_ = (s) * NI + i + 2;                                               \
                f16x8 nh = a1h, nl = a1l;                                                       \
                if (t2_ < 9 * NI) { nh = HS_LDS(HS_AOFF(t2_)); if (TWO) nl = HS_LDS(HS_AOFF(t2_) + 32); } \
                __builtin_amdgcn_sched_barrier(0);                                              \
                if (i == 0) {                                                                   \
                    HS_TSTAMP(c, (s))                                                           \
                    if (MORE || (s) + AHEAD < 9) { HS_LOADB(((s) + AHEAD) % RING, gnext + (s)); } \
                    if ((s) == HS_XLOAD_TAP && MORE) { HS_LOAD(c + 1) }                         \
                }                                                                               \
                {                                                                               \
                    const int tl_ = (s) * NI + i - (9 * NI - 2 * NST);                          \
                    if (AFF && MORE && 9 * NI >= 2 * NST && tl_ == -1) HS_AFF_READ(c + 1)       \
                    if (MORE && 9 * NI >= 2 * NST && tl_ >= 0) {                                \
                        if ((tl_ & 1) == 0) { HS_SPLIT_A((tl_ < 0 ? 0 : tl_ >> 1)) }            \
                        else { HS_SPLIT_B((tl_ < 0 ? 0 : tl_ >> 1), wb) }                       \
                    }                                                                           \
                }                                                                               \
                HS_MFMA(a0h, Bq[(s) % RING][0], acc[i]);                                     \
                if (SPLIT) {                                                                    \
                    HS_MFMA(a0h, Bq[(s) % RING][1], acc[i]);                                 \
                    HS_MFMA(a0l, Bq[(s) % RING][0], acc[i]);                                 \
                } else if (K32) {                        /* the chunk's second k step: channels 16-31 */ \
                    HS_MFMA(a0l, Bq[(s) % RING][1], acc[i]);                                 \
                }                                                                               \
                a0h = a1h; a0l = a1l; a1h = nh; a1l = nl;                                       \
                __builtin_amdgcn_sched_barrier(0);                                              \
            }                                                                                   \
        }
#define HS_CHUNK(MORE)                                                                          \
    {                                                                                           \
        const unsigned rb = (unsigned)(c & 1) * BUFB;                                           \
        const unsigned wb = (unsigned)((c & 1) ^ 1) * BUFB;                                     \
        const int gnext = c * 9 + AHEAD;             /* global step the first prefetch of this chunk fetches */ \
        f16x8 a0h = HS_LDS(HS_AOFF(0)), a1h = HS_LDS(HS_AOFF(1)), a0l = a0h, a1l = a1h;         \
        if (TWO) { a0l = HS_LDS(HS_AOFF(0) + 32); a1l = HS_LDS(HS_AOFF(1) + 32); }              \
        HS_TAP(0, MORE) HS_TAP(1, MORE) HS_TAP(2, MORE) HS_TAP(3, MORE) HS_TAP(4, MORE)         \
        HS_TAP(5, MORE) HS_TAP(6, MORE) HS_TAP(7, MORE) HS_TAP(8, MORE)                         \
        if (MORE && 9 * NI < 2 * NST) {                 /* too few regions to spread the split over: do it here */ \
            if (AFF) HS_AFF_READ(c + 1)                                                         \
            _Pragma("unroll") for (int k = 0; k < NST; ++k) { HS_SPLIT_A(k) HS_SPLIT_B(k, wb) } \
        }                                                                                       \
        __syncthreads();                                                                        \
    }
    int c = 0;
    HS_STAMP(1)
    // the chunk loop's waves go before the co-resident block's prologue / epilogue waves (16-bit frame: a third of the split form's
    // MFMAs per block, so a block's frame is 30 % of its life and what it costs the partner's matrix stream shows: +2.5-3 %, A/B
    // in one call, profiles/r5_a16_ab.log; the split form measured no gain in rounds 3-4)
    if (K32) __builtin_amdgcn_s_setprio(HS_PRIO);
    HS_STAMP_REAL(5)
#pragma unroll 1
    for (; c < NCHUNK - 1; ++c) HS_CHUNK(true)
    HS_STAMP(2)
    // MODE 7: the g_y rows of the epilogue's first pass are requested here, into the registers the staging items no longer need: their
    // HBM latency passes under the last chunk's MFMAs (the epilogue below requests every later pass's rows one pass ahead)
    float4 gyv[NORES ? 2 : 1][NORES ? 8 : 1];
    if (NORES) {
#pragma unroll
        for (int j = 0; j < (NI >= 2 ? 8 : 4); ++j) {
            const int m_ = min(m0 + (tid >> 5) + 8 * j, m1 - 1);
            gyv[0][j] = *(const float4 *)(p.g_y + (long)img * HWc * HS_C + 4 * (tid & 31) + (long)m_ * HS_C);
        }
    }
    HS_CHUNK(false)                    // the last chunk stages nothing
    HS_STAMP(3)
    if (K32) __builtin_amdgcn_s_setprio(0);
    // range guard: an input beyond the f16 range after scaling was clamped, the layer's result is then NOT float32-accurate.
    // The flag word sits behind the scales in the weight image (one per layer); the host reads it (QNet.check_range).
    // The same event is also stored to the device word tail[6..7] points to, when the caller registered one (ONE word for all
    // layers of a net, snk_conv3x3_f16s_set_guard_word): the kernels of a rollout tick that follow the evaluation are gated on it
    // (csrc/mcts.hip TICK_GATE), and the host learns from one word whether any launch of a forward clamped.
    if (amax >= 65504.f) {
        int *t = (int *)const_cast<float *>(p.wscale_inv);
        atomicOr(t + 4, 1);
        int *shared = *(int *const *)(t + 6);
        if (shared) __hip_atomic_store(shared, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    HS_STAMP_REAL(6)
#undef HS_TAP
#undef HS_CHUNK
#undef HS_AOFF
#undef HS_LOAD
#undef HS_SPLIT_A
#undef HS_SPLIT_B
#undef HS_LOADB
#undef HS_AFF_READ
#undef HS_LDS
#undef HS_MFMA

    // ---- epilogue: 64 GEMM rows (2 M tiles) at a time through a double-buffered LDS exchange, so that every thread
    //      handles float4 pieces of whole pixel rows: thread (rr0 = tid / 32, cq = tid % 32) takes rows rr0 + 8 j of the
    //      pass, outputs 4 cq .. 4 cq + 3.  The residual loads of pass k + 1 are issued before pass k is exchanged and
    //      stored; one block barrier per pass (the other block of the CU computes meanwhile).
    if constexpr (P8) {
        // ---- the 16-bit frame's epilogue: the same double-buffered exchange, but a thread takes EIGHT channels of a row (16 bytes of
        //      shortcut in, 16 bytes out; 16 lanes per pixel row, rows rr0 + 16 j of the pass): half the memory instructions
        const int cq = tid & 15, rr0 = tid >> 4;
        const float winv = p.wscale_inv[0] * p.wscale_inv[3];
        float4 scA = *(const float4 *)(p.scale + 8 * cq), scB = *(const float4 *)(p.scale + 8 * cq + 4);
        const float4 shA = *(const float4 *)(p.shift + 8 * cq), shB = *(const float4 *)(p.shift + 8 * cq + 4);
        scA.x *= winv; scA.y *= winv; scA.z *= winv; scA.w *= winv; scB.x *= winv; scB.y *= winv; scB.z *= winv; scB.w *= winv;
        float *Ms = (float *)smem;                                      // [2][64 rows][HS_MLD]
        constexpr int NPASS = (NI + 1) / 2;
        f16x8 rv[NPASS][4];                 // ALL shortcut rows of the block are requested before the first pass: one HBM latency per
        unsigned off[NPASS][4];             //   block instead of one per pass (the registers of the main loop's fragments are free by now)
        // the output goes through a buffer descriptor of the image: rows past the block's pixels carry an offset beyond the image and
        // the hardware's range check drops their stores.  No branch around a store: with branches the compiler cannot count the
        // stores in flight and puts s_waitcnt vmcnt(0) before every use of a shortcut row -- each row then waits for the previous
        // row's store to be acknowledged by HBM (16 round trips per block: 11 k of this epilogue's 16 k cycles, profiles/r5_a16_stamps.log)
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((_Float16 *)p.out + (long)img * HWc * HS_C), 0, HWc * HS_C * 2, 0x00020000);
        float omax = 0.f;                   // f16 tower: the largest |output| this thread wrote (65504 = it was saturated)
        const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((_Float16 *)(has_res ? p.res : p.out) + (long)img * HWc * HS_C), 0, HWc * HS_C * 2, 0x00020000);
        const bool sel_res = RECT && p.bg_res != nullptr;
#define HS_ROWS(pass) (((pass) + 1 < NPASS ? 2 : NI - 2 * (NPASS - 1)) * 32)
#define HS_EPI_PREP(pass)                 /* element offsets of this thread's rows (-1: past the block's pixels) + shortcut loads */ \
        {                                                                                       \
            _Pragma("unroll") for (int j = 0; j < HS_ROWS(pass) / 16; ++j) {                    \
                const int m_ = m0 + 64 * (pass) + rr0 + 16 * j;                                 \
                bool stale_ = false;                                                            \
                if (RECT) {                                                                     \
                    const int y_ = (int)(((float)m_ + 0.5f) * invW), x_ = m_ - y_ * Wr;         \
                    off[pass][j] = m_ < m1 ? (unsigned)(((cy0 + y_) * p.Wd + cx0 + x_) * HS_C + 8 * cq) : 0x40000000u; \
                    stale_ = sel_res && (cy0 + y_ < by0 - p.grow_res || cy0 + y_ > by1 + p.grow_res || \
                                         cx0 + x_ < bx0 - p.grow_res || cx0 + x_ > bx1 + p.grow_res); \
                } else off[pass][j] = m_ < m1 ? (unsigned)(m_ * HS_C + 8 * cq) : 0x40000000u;  \
                if (has_res && !RECT) {       /* whole images: the shortcut through the image's descriptor too (one offset register per row) */ \
                    rv[pass][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, 2u * off[pass][j], 0, 0)); \
                } else if (has_res) {                                                           \
                    const _Float16 *rb_ = stale_ ? (const _Float16 *)p.bg_res : (const _Float16 *)p.res + (long)img * HWc * HS_C; \
                    rv[pass][j] = *(const f16x8 *)(rb_ + (off[pass][j] < 0x40000000u ? off[pass][j] : 8u * cq)); \
                }                                                                               \
            }                                                                                   \
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) HS_EPI_PREP(pass)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            float *Mb = Ms + (pass & 1) * (64 * HS_MLD);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                if (2 * pass + ii < NI) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Mb[(32 * ii + (r & 3) + 8 * (r >> 2) + 4 * h) * HS_MLD + 32 * wn + l31] = acc[2 * pass + ii][r];
                }
            __syncthreads();
            float4 mA[4], mB[4];
#pragma unroll
            for (int j = 0; j < HS_ROWS(pass) / 16; ++j) {
                mA[j] = *(const float4 *)&Mb[(rr0 + 16 * j) * HS_MLD + 8 * cq];
                mB[j] = *(const float4 *)&Mb[(rr0 + 16 * j) * HS_MLD + 8 * cq + 4];
            }
#pragma unroll
            for (int j = 0; j < HS_ROWS(pass) / 16; ++j) {
                float v[8] = {__builtin_fmaf(mA[j].x, scA.x, shA.x), __builtin_fmaf(mA[j].y, scA.y, shA.y),
                              __builtin_fmaf(mA[j].z, scA.z, shA.z), __builtin_fmaf(mA[j].w, scA.w, shA.w),
                              __builtin_fmaf(mB[j].x, scB.x, shB.x), __builtin_fmaf(mB[j].y, scB.y, shB.y),
                              __builtin_fmaf(mB[j].z, scB.z, shB.z), __builtin_fmaf(mB[j].w, scB.w, shB.w)};
                f16x8 o_;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t_ = v[e];
                    if (has_res) t_ += hs_from16<BF>(rv[pass][j][e]);
                    t_ = fmaxf(t_, relu_floor);
                    if (!BF && off[pass][j] < 0x40000000u) omax = fmaxf(omax, fabsf(t_));      // f16 activations saturate at 65504: seen, not silent
                    o_[e] = hs_to16<BF>(t_);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o_), o_rsrc, 2u * off[pass][j], 0, 0);
            }
        }
#undef HS_ROWS
#undef HS_EPI_PREP
        if (!BF && omax >= 65504.f) {        // the f16 tower's range guard: the same flag word and guard word as the split form's clamp
            int *t = (int *)const_cast<float *>(p.wscale_inv);
            atomicOr(t + 4, 1);
            int *shared = *(int *const *)(t + 6);
            if (shared) __hip_atomic_store(shared, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (RECT && p.bg_out) {              // see the other frame's copy loop below: 16 lanes per pixel here
            const float invF = 1.0f / (float)p.Wd;
            for (int q = part * 16 + rr0; q < HWc; q += 16 * parts) {
                const int y_ = (int)(((float)q + 0.5f) * invF), x_ = q - y_ * p.Wd;
                if (y_ >= cy0 && y_ < cy0 + rh && x_ >= cx0 && x_ < cx0 + rw) continue;
                const int o_ = (y_ * p.Wd + x_) * HS_C + 8 * cq;
                *(f16x8 *)((_Float16 *)p.out + (long)img * HWc * HS_C + o_) = *(const f16x8 *)((const _Float16 *)p.bg_out + o_);
            }
        }
    } else if constexpr (GSTATS) {
    // ---- the training step's input-gradient epilogue (MODE 5 / 7 / 8): the same 64-row exchange; everything a pass needs from HBM --
    //      the shortcut rows, the g_y rows, the mask bytes -- is requested in ONE batch before the pass is exchanged, at clamped
    //      addresses, and NO branch surrounds a load or a store: rows past the block's pixels are computed, kept out of the sums by
    //      their mask, and stored to an offset the image's buffer descriptor drops.  (Round 5's first form kept the g_y row and
    //      the mask byte inside `if (row is the block's)`: the compiler then waited vmcnt(0) in every row -- for the row's own two
    //      loads AND for the previous row's store -- 28 HBM round trips one after the other per thread and block, which is what
    //      made this launch 0.11 ms slower than the bare convolution.)  MODE 7 has no shortcut rows and holds the g_y rows of TWO
    //      passes instead: requested a pass ahead, the first pass's under the last chunk above.
    const int cq = tid & 31, rr0 = tid >> 5;
    const float winv = p.wscale_inv[0] * p.wscale_inv[3];
    const float4 cen4 = *(const float4 *)(p.center + 4 * cq), iv4 = *(const float4 *)(p.g_inv + 4 * cq);
    float4 gsc = cen4, gsh = cen4;
    if (GAFF) { gsc = *(const float4 *)(p.aff_scale + 4 * cq); gsh = *(const float4 *)(p.aff_shift + 4 * cq); }
    float4 st_s = make_float4(0.f, 0.f, 0.f, 0.f), st_q = st_s;
    float *Ms = (float *)smem;                                      // [2][64 rows][HS_MLD]
    // every tensor through a buffer descriptor of the IMAGE: one offset register serves all rows of all passes (a row is a constant
    // further on), and what lies past the image reads as zero (mask byte 0: out of the sums)
    const long ielem = (long)img * HWc * HS_C;
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + ielem), 0, HWc * HS_C * 4, 0x00020000);
    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.g_y + ielem), 0, HWc * HS_C * 4, 0x00020000);
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((has_res ? p.res : p.g_y) + ielem), 0, HWc * HS_C * 4, 0x00020000);
    const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((GAFF ? (const unsigned char *)p.g_y : p.g_mask) + (ielem >> 2)), 0, HWc * (HS_C / 4), 0x00020000);
    const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((RESMASK ? p.res_mask : (const unsigned char *)p.g_y) + (ielem >> 2)), 0, HWc * (HS_C / 4), 0x00020000);
    const unsigned vo4 = (unsigned)((m0 + rr0) * HS_C + 4 * cq) * 4u, vo1 = (unsigned)((m0 + rr0) * (HS_C / 4) + cq);
    constexpr int NPASS = (NI + 1) / 2;
    float4 rv[8], yv[NORES ? 1 : 8];
    unsigned mbv[GAFF ? 1 : 8], rmv[RESMASK ? 8 : 1];
#define HS_ROWS(pass) (((pass) + 1 < NPASS ? 2 : NI - 2 * (NPASS - 1)) * 32)
#define HS_LD4(rsrc, pass, j) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo4, (64 * (pass) + 8 * (j)) * HS_C * 4, 0))
#define HS_LD1(rsrc, pass, j) (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsrc, vo1, (64 * (pass) + 8 * (j)) * (HS_C / 4), 0)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < HS_ROWS(pass) / 8; ++j) {
            if (!NORES) rv[j] = HS_LD4(r_rsrc, pass, j);     // (MODE 5 without a shortcut: the row is read -- through g_y's descriptor -- and not used)
            if (RESMASK) rmv[j] = HS_LD1(q_rsrc, pass, j);
            if (!NORES) yv[j] = HS_LD4(y_rsrc, pass, j);
            if (!GAFF) mbv[j] = HS_LD1(m_rsrc, pass, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        float *Mb = Ms + (pass & 1) * (64 * HS_MLD);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
            if (2 * pass + ii < NI) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Mb[(32 * ii + (r & 3) + 8 * (r >> 2) + 4 * h) * HS_MLD + 32 * wn + l31] = acc[2 * pass + ii][r];
            }
        __syncthreads();
        float4 m[8];
#pragma unroll
        for (int j = 0; j < HS_ROWS(pass) / 8; ++j) m[j] = *(const float4 *)&Mb[(rr0 + 8 * j) * HS_MLD + 4 * cq];
        if (NORES && pass + 1 < NPASS) {
#pragma unroll
            for (int j = 0; j < HS_ROWS(pass + 1) / 8; ++j) gyv[(pass + 1) & 1][j] = HS_LD4(y_rsrc, pass + 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < HS_ROWS(pass) / 8; ++j) {
            const bool mine = m0 + 64 * pass + rr0 + 8 * j < m1;
            float4 v = m[j];
            v.x = __builtin_fmaf(v.x, winv, 0.f); v.y = __builtin_fmaf(v.y, winv, 0.f);
            v.z = __builtin_fmaf(v.z, winv, 0.f); v.w = __builtin_fmaf(v.w, winv, 0.f);
            if (has_res) {
                float4 r_ = rv[j];
                if (RESMASK) r_ = make_float4((rmv[j] & 1u) ? r_.x : 0.f, (rmv[j] & 2u) ? r_.y : 0.f, (rmv[j] & 4u) ? r_.z : 0.f, (rmv[j] & 8u) ? r_.w : 0.f);
                v.x += r_.x; v.y += r_.y; v.z += r_.z; v.w += r_.w;
            }
            const float4 y_ = NORES ? gyv[pass & 1][j] : yv[j];
            bool bx, by, bz, bw;
            if (GAFF) {                      // the layer's output was never written: its sign is that of k_bn_apply's expression
                bx = y_.x * gsc.x + gsh.x > 0.f; by = y_.y * gsc.y + gsh.y > 0.f; bz = y_.z * gsc.z + gsh.z > 0.f; bw = y_.w * gsc.w + gsh.w > 0.f;
            } else { bx = mbv[j] & 1u; by = mbv[j] & 2u; bz = mbv[j] & 4u; bw = mbv[j] & 8u; }
            const float gx = (mine && bx) ? v.x : 0.f, gy = (mine && by) ? v.y : 0.f, gz = (mine && bz) ? v.z : 0.f, gw = (mine && bw) ? v.w : 0.f;
            st_s.x += gx; st_s.y += gy; st_s.z += gz; st_s.w += gw;
            st_q.x += gx * ((y_.x - cen4.x) * iv4.x); st_q.y += gy * ((y_.y - cen4.y) * iv4.y);
            st_q.z += gz * ((y_.z - cen4.z) * iv4.z); st_q.w += gw * ((y_.w - cen4.w) * iv4.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v), o_rsrc, mine ? vo4 : 0x80000000u,
                                                   (64 * pass + 8 * j) * HS_C * 4, 0);
            // gfx950 reads a 16-byte store's data registers over several cycles; the compiler guards a VALU write to them that follows at
            // once only when the store has NO scalar offset (measured here: rows whose data register pair was rewritten by the
            // next row's v_pk_fma_f32 in the cycle after the store reached HBM with the NEXT row's values in the last four lanes of
            // every sixteen, tools/dbg/igrad_dbg.py) -- so: wait states by hand, and nothing scheduled across them.  The rule as
            // measured (tools/micro/store_hazard.hip, profiles/r6_store_hazard_micro.json): with an SGPR soffset ONE wait state is
            // enough, without one it takes two (those the compiler inserts).  tests/test_store_hazard_cpu.py checks every wide
            // store of the shipped code objects against it; HS_NO_STORE_NOP is that test's proof that it bites.
#ifndef HS_NO_STORE_NOP
            asm volatile("s_nop 2");
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef HS_ROWS
#undef HS_LD4
#undef HS_LD1
    __syncthreads();                       // the eight row lanes' sums of a channel quad are added through LDS, in a fixed order
    float4 *R = (float4 *)smem;            // [2][8][32]
    R[rr0 * 32 + cq] = st_s;
    R[(8 + rr0) * 32 + cq] = st_q;
    __syncthreads();
    if (rr0 < 2) {
        float4 t = R[(8 * rr0) * 32 + cq];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 u = R[(8 * rr0 + r) * 32 + cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        *(float4 *)(p.stat_part + (size_t)blockIdx.x * 256 + rr0 * 128 + 4 * cq) = t;
    }
    } else {
    const int cq = tid & 31, rr0 = tid >> 5;
    const float winv = p.wscale_inv[0] * p.wscale_inv[3];
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!BARE) { sc4 = *(const float4 *)(p.scale + 4 * cq); sh4 = *(const float4 *)(p.shift + 4 * cq); }
    sc4.x *= winv; sc4.y *= winv; sc4.z *= winv; sc4.w *= winv;
    float4 cen4 = make_float4(0.f, 0.f, 0.f, 0.f), st_s = cen4, st_q = cen4;
    if (BARE && p.center) cen4 = *(const float4 *)(p.center + 4 * cq);
    float4 st_m = cen4;
    float *Ms = (float *)smem;                                      // [2][64 rows][HS_MLD]
    const long obase = (long)img * HWc * HS_C + 4 * cq;
    float4 wh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_head) wh = *(const float4 *)(p.w1x1 + 4 * cq);
    constexpr int NPASS = (NI + 1) / 2;
    float4 rv[2][8];
    int off[2][8];
    bool res_stale[8];
    const bool sel_res = RECT && p.bg_res != nullptr;
#define HS_ROWS(pass) (((pass) + 1 < NPASS ? 2 : NI - 2 * (NPASS - 1)) * 32)
#define HS_EPI_PREP(pass)                 /* element offsets of this thread's rows (-1: past the block's pixels) + residual loads */ \
    {                                                                                           \
        _Pragma("unroll") for (int j = 0; j < HS_ROWS(pass) / 8; ++j) {                         \
            const int m_ = m0 + 64 * (pass) + rr0 + 8 * j;                                      \
            if (RECT) {                                                                         \
                const int y_ = (int)(((float)m_ + 0.5f) * invW), x_ = m_ - y_ * Wr;             \
                off[(pass) & 1][j] = m_ < m1 ? ((cy0 + y_) * p.Wd + cx0 + x_) * HS_C : -1;      \
                res_stale[j] = sel_res && (cy0 + y_ < by0 - p.grow_res || cy0 + y_ > by1 + p.grow_res || \
                                           cx0 + x_ < bx0 - p.grow_res || cx0 + x_ > bx1 + p.grow_res); \
            } else off[(pass) & 1][j] = m_ < m1 ? m_ * HS_C : -1;                               \
        }                                                                                       \
        if (has_res) {                                                                          \
            _Pragma("unroll") for (int j = 0; j < HS_ROWS(pass) / 8; ++j) {                     \
                if (IN16) {                                                                     \
                    const _Float16 *rb_ = (RECT && res_stale[j]) ? (const _Float16 *)p.bg_res + 4 * cq : (const _Float16 *)p.res + obase; \
                    const f16x4 r16_ = *(const f16x4 *)(rb_ + max(off[(pass) & 1][j], 0));      \
                    rv[(pass) & 1][j] = make_float4(hs_from16<BF>(r16_[0]), hs_from16<BF>(r16_[1]), hs_from16<BF>(r16_[2]), hs_from16<BF>(r16_[3])); \
                } else {                                                                        \
                    const float *rb_ = (RECT && res_stale[j]) ? p.bg_res + 4 * cq : p.res + obase; \
                    rv[(pass) & 1][j] = *(const float4 *)(rb_ + max(off[(pass) & 1][j], 0));    \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
    }
    HS_EPI_PREP(0)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass + 1 < NPASS) HS_EPI_PREP(pass + 1)
        float *Mb = Ms + (pass & 1) * (64 * HS_MLD);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
            if (2 * pass + ii < NI) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Mb[(32 * ii + (r & 3) + 8 * (r >> 2) + 4 * h) * HS_MLD + 32 * wn + l31] = acc[2 * pass + ii][r];
            }
        __syncthreads();
        float4 m[8];
#pragma unroll
        for (int j = 0; j < HS_ROWS(pass) / 8; ++j) m[j] = *(const float4 *)&Mb[(rr0 + 8 * j) * HS_MLD + 4 * cq];
#pragma unroll
        for (int j = 0; j < HS_ROWS(pass) / 8; ++j) {
            float4 v = m[j];
            v.x = __builtin_fmaf(v.x, sc4.x, sh4.x); v.y = __builtin_fmaf(v.y, sc4.y, sh4.y);
            v.z = __builtin_fmaf(v.z, sc4.z, sh4.z); v.w = __builtin_fmaf(v.w, sc4.w, sh4.w);
            if (has_res) {
                const float4 r_ = rv[pass & 1][j];
                v.x += r_.x; v.y += r_.y; v.z += r_.z; v.w += r_.w;
            }
            v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
            if (STATS && off[pass & 1][j] >= 0) {
                const float ex = v.x - cen4.x, ey = v.y - cen4.y, ez = v.z - cen4.z, ew = v.w - cen4.w;
                st_s.x += ex; st_s.y += ey; st_s.z += ez; st_s.w += ew;
                st_q.x += ex * ex; st_q.y += ey * ey; st_q.z += ez * ez; st_q.w += ew * ew;
                if (AMAX) {                  // the range of what this layer's batch norm can make of it (snk_bn_train_finalize_range)
                    st_m.x = fmaxf(st_m.x, fabsf(ex)); st_m.y = fmaxf(st_m.y, fabsf(ey));
                    st_m.z = fmaxf(st_m.z, fabsf(ez)); st_m.w = fmaxf(st_m.w, fabsf(ew));
                }
            }
            if (has_out && off[pass & 1][j] >= 0) {
                if (OUT16) {
                    f16x4 o16_;
                    o16_[0] = hs_to16<BF>(v.x); o16_[1] = hs_to16<BF>(v.y); o16_[2] = hs_to16<BF>(v.z); o16_[3] = hs_to16<BF>(v.w);
                    *(f16x4 *)((_Float16 *)p.out + obase + off[pass & 1][j]) = o16_;
                } else *(float4 *)(p.out + obase + off[pass & 1][j]) = v;
            }
            if (has_head) {                 // the 32 lanes of this row hold its 128 outputs: reduce their dot product with w1x1
                float d_ = (v.x * wh.x + v.y * wh.y) + (v.z * wh.z + v.w * wh.w);
                // butterfly inside each 16-lane DPP row (quad swaps, half-row mirror, row mirror), then the other row of the 32
                d_ += hs_dpp(d_, 0xB1); d_ += hs_dpp(d_, 0x4E); d_ += hs_dpp(d_, 0x141); d_ += hs_dpp(d_, 0x140);
                d_ += __shfl_xor(d_, 16, 64);
                if (cq == 0 && off[pass & 1][j] >= 0)
                    p.h1[(long)img * HWc + off[pass & 1][j] / HS_C] = fmaxf(__builtin_fmaf(d_, p.s1, p.b1), 0.f);
            }
        }
    }
#undef HS_ROWS
#undef HS_EPI_PREP
    if (BARE) {                            // the eight row lanes' sums of a channel quad are added through LDS, in a fixed order
        __syncthreads();
        float4 *R = (float4 *)smem;        // [2 (3)][8][32]
        R[rr0 * 32 + cq] = st_s;
        R[(8 + rr0) * 32 + cq] = st_q;
        if (AMAX) R[(16 + rr0) * 32 + cq] = st_m;
        __syncthreads();
        if (AMAX && rr0 == 2 && p.amax_part) {
            float4 t = R[16 * 32 + cq];
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                const float4 u = R[(16 + r) * 32 + cq];
                t.x = fmaxf(t.x, u.x); t.y = fmaxf(t.y, u.y); t.z = fmaxf(t.z, u.z); t.w = fmaxf(t.w, u.w);
            }
            *(float4 *)(p.amax_part + (size_t)blockIdx.x * HS_C + 4 * cq) = t;
        }
        if (rr0 < 2) {
            float4 t = R[(8 * rr0) * 32 + cq];
#pragma unroll
            for (int r = 1; r < 8; ++r) { const float4 u = R[(8 * rr0 + r) * 32 + cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *(float4 *)(p.stat_part + (size_t)blockIdx.x * 256 + rr0 * 128 + 4 * cq) = t;
        }
    }
    if (RECT && p.bg_out) {
        // the layers that read this output are full layers: every pixel of the canvas outside the rectangle takes the layer's
        // state-independent background value; the image's parts share the pixels, 32 lanes per pixel
        const float invF = 1.0f / (float)p.Wd;
        for (int q = part * 8 + rr0; q < HWc; q += 8 * parts) {
            const int y_ = (int)(((float)q + 0.5f) * invF), x_ = q - y_ * p.Wd;
            if (y_ >= cy0 && y_ < cy0 + rh && x_ >= cx0 && x_ < cx0 + rw) continue;
            const int o_ = (y_ * p.Wd + x_) * HS_C + 4 * cq;
            if (OUT16) *(f16x4 *)((_Float16 *)p.out + (long)img * HWc * HS_C + o_) = *(const f16x4 *)((const _Float16 *)p.bg_out + o_);
            else *(float4 *)(p.out + (long)img * HWc * HS_C + o_) = *(const float4 *)(p.bg_out + o_);
        }
    }
    }      // the epilogue of the other frames
    HS_STAMP(4)
}

template <int NI, int MODE, bool SPLIT = true, int IO16 = 0, bool BF = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_f16s(ConvHsArgs p)
{
    __shared__ __align__(16) unsigned char smem[HS_SMEM];
    // XCD-aware block -> (image, part) map: workgroups go round-robin to the 8 XCDs, each with its own L2, and the parts of
    // an image share halo rows; parts of one image are therefore 8 workgroups apart (same XCD, dispatched back to back)
    int img, blk;
    {
        const int b = blockIdx.x, per = 8 * p.n_blk;
        if (b < p.n_img_grouped * p.n_blk) { const int r = b % per; img = (b / per) * 8 + (r & 7); blk = r >> 3; }
        else { img = b / p.n_blk; blk = b - img * p.n_blk; }
    }
    const int tile0 = blk * p.tiles_base + min(blk, p.tiles_rem), ntile = p.tiles_base + (blk < p.tiles_rem ? 1 : 0);
    // an image whose tile count is not a multiple of its block count has blocks of NI - 1 tiles: the 16-bit towers' kernel carries
    // that body too (37 x 37: 43 tiles = one block of 8 + five of 7, which all ran the 8-tile body before: a ninth of the full
    // layers' MFMAs computed rows nobody stored); the choice is wave-uniform
    if constexpr (IO16 != 0 && NI > 1) {
        switch (ntile) {
        case NI - 1: hs_block<NI - 1, MODE, SPLIT, IO16, false, BF>(p, smem, img, tile0, ntile, 0, 0, 0, 0, 0u, 0, 1); break;
        default: hs_block<NI, MODE, SPLIT, IO16, false, BF>(p, smem, img, tile0, ntile, 0, 0, 0, 0, 0u, 0, 1); break;
        }
    } else hs_block<NI, MODE, SPLIT, IO16, false, BF>(p, smem, img, tile0, ntile, 0, 0, 0, 0, 0u, 0, 1);
}

// The sub-rectangle form.  The reference's observation (game.py:215-257) is one background pixel (0, WALL, 0) everywhere
// outside the board window, so the output of tower layer k (0-based) is the same for every state outside the window grown
// by k + 2 pixels (the stem: 1): only that rectangle is computed and written per image; what a reader needs beyond its
// producer's rectangle it takes from the producer's constant background image (QNet.backgrounds; p.bg_in / p.bg_res, chosen
// per staging item and per shortcut row when the block sets up its addresses).  Blocks take (image, rectangle, tile range,
// bounding box) from descriptors written by k_rect_plan; the M-tile count of a block selects the body at run time
// (wave-uniform), so one launch covers every rectangle shape of the batch.
template <int MODE, bool SPLIT = true, int IO16 = 0, bool BF = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_f16s_rect(ConvHsArgs p)
{
    __shared__ __align__(16) unsigned char smem[HS_SMEM];
    const int nd = *p.n_desc;
    const uint4 d = p.desc[blockIdx.x];                    // both loads in flight together: the grid never exceeds the array
    if ((int)blockIdx.x >= nd) return;
    const int img = (int)d.x, ry0 = d.y & 255, rx0 = (d.y >> 8) & 255, rh = (d.y >> 16) & 255, rw = d.y >> 24;
    const int tile0 = d.z & 255, ntile = (d.z >> 8) & 255, part = (d.z >> 16) & 255, parts = d.z >> 24;
#define HS_RECT_CASE(NI_) case NI_: hs_block<NI_, MODE, SPLIT, IO16, true, BF>(p, smem, img, tile0, ntile, ry0, rx0, rh, rw, d.w, part, parts); break;
    switch (ntile) {
        HS_RECT_CASE(1) HS_RECT_CASE(2) HS_RECT_CASE(3) HS_RECT_CASE(4) HS_RECT_CASE(5) HS_RECT_CASE(6) HS_RECT_CASE(7)
    default: hs_block<8, MODE, SPLIT, IO16, true, BF>(p, smem, img, tile0, ntile, ry0, rx0, rh, rw, d.w, part, parts); break;
    }
#undef HS_RECT_CASE
}

// max |w| of the layer -> k with 256 <= max * 2^k < 512; writes {2^-k, 2^k, x_scale, 1 / x_scale, range flag = 0} behind the fragment image
__global__ __launch_bounds__(1024) void k_f16s_wscale(const float *__restrict__ w, float *__restrict__ tail, float x_scale,
                                                      const float *__restrict__ in_tail = nullptr)
{
    __shared__ float red[1024];
    float m = 0.f;
    const float4 *w4 = (const float4 *)w;                      // 9 * 128 * 128 / 4 = 36 float4 per thread, all in flight
#pragma unroll 12
    for (int i = threadIdx.x; i < 9 * HS_C * HS_C / 4; i += 1024) {
        const float4 v = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        int k = 0;
        if (mx > 0.f && mx < 3.0e38f) k = 8 - ilogbf(mx);
        k = max(-100, min(100, k));
        tail[0] = ldexpf(1.0f, -k);
        tail[1] = ldexpf(1.0f, k);
        tail[2] = in_tail ? in_tail[2] : x_scale;             // the producer of the input measured its range (train.hip)
        tail[3] = in_tail ? in_tail[3] : 1.0f / x_scale;
        tail[4] = tail[5] = tail[6] = tail[7] = 0.f;          // range flag (an int32, bit pattern 0) + padding
    }
}

// the tail of an image of the SAME kernel values (their power-of-two scale is known) with another input's range
__global__ void k_f16s_tail_from(const float *__restrict__ w_tail, const float *__restrict__ in_tail, float *__restrict__ tail)
{
    if (threadIdx.x == 0) {
        tail[0] = w_tail[0]; tail[1] = w_tail[1];
        tail[2] = in_tail[2]; tail[3] = in_tail[3];
        tail[4] = tail[5] = tail[6] = tail[7] = 0.f;
    }
}

// Keras kernel (kh, kw, cin, cout) float32 -> split f16 fragments in the order the conv kernel's waves load them
// FLIP: the kernel of the INPUT GRADIENT of the same layer -- taps mirrored, channel axes swapped: w'[tap][ci][co] = w[8 - tap][co][ci]
template <bool FLIP>
__global__ void k_f16s_weights(const float *__restrict__ w, _Float16 *__restrict__ wS, const float *__restrict__ tail)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;          // one f16x8 fragment piece
    if (v >= HS_WS_ELEMS / 8) return;
    const float mul = tail[1];
    const int g = v / 512, rem = v - g * 512;
    const int c = g / 9, tap = g - 9 * c;                       // 16-channel chunk, tap
    const int wn = rem >> 7, hl = (rem >> 6) & 1, lane = rem & 63;
    const int h = lane >> 5, l31 = lane & 31;
    const int cout = 32 * wn + l31;
    for (int j = 0; j < 8; ++j) {
        const int cin = HS_KC * c + 8 * h + j;
        const float val = (FLIP ? w[(long)((8 - tap) * HS_C + cout) * HS_C + cin] : w[(long)(tap * HS_C + cin) * HS_C + cout]) * mul;
        const _Float16 hi = (_Float16)val;
        wS[(long)v * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
    }
}

// the 16-bit frame's weight image: [chunk of 32 input channels 4][tap 9][wn 4][k step 2][lane 64] x 8 values (f16, or bf16
// patterns): the two fragments a wave loads per (chunk, tap) are the chunk's two MFMA k steps, where the split image keeps hi and lo
template <bool BF>
__global__ void k_a16_weights(const float *__restrict__ w, _Float16 *__restrict__ wS, const float *__restrict__ tail)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;          // one fragment piece of eight values
    if (v >= 9 * HS_C * HS_C / 8) return;
    const float mul = tail[1];
    const int g = v / 512, rem = v - g * 512;
    const int c = g / 9, tap = g - 9 * c;
    const int wn = rem >> 7, ks = (rem >> 6) & 1, lane = rem & 63;
    const int h = lane >> 5, l31 = lane & 31;
    const int cout = 32 * wn + l31;
    for (int j = 0; j < 8; ++j) {
        const int cin = 32 * c + 16 * ks + 8 * h + j;
        const float val = w[(long)(tap * HS_C + cin) * HS_C + cout] * mul;
        wS[(long)v * 8 + j] = BF ? __builtin_bit_cast(_Float16, (__bf16)val) : (_Float16)val;
    }
}

// ---- the guard word: one device int32 that every layer of a net reports clamps to ----------------------------------------------
__global__ void k_f16s_set_guard_word(int *tail, int *word) { *(int **)(tail + 6) = word; }

extern "C" int snk_conv3x3_f16s_set_guard_word(void *d_wS, int32_t *d_word, void *stream)
{
    SNK_REQUIRE(d_wS, "snk_conv3x3_f16s_set_guard_word: NULL weight image");
    k_f16s_set_guard_word<<<1, 1, 0, (hipStream_t)stream>>>((int *)((_Float16 *)d_wS + HS_WS_ELEMS), (int *)d_word);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_prepare_weights_f16s(const float *d_w_hwio, void *d_wS, float x_scale, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS, "snk_conv3x3_prepare_weights_f16s: NULL argument");
    int e_ = 0;
    SNK_REQUIRE(x_scale > 0.f && frexpf(x_scale, &e_) == 0.5f, "snk_conv3x3_prepare_weights_f16s: x_scale %g is not a power of two", x_scale);
    float *tail = (float *)((_Float16 *)d_wS + HS_WS_ELEMS);
    k_f16s_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail, x_scale);
    k_f16s_weights<false><<<(HS_WS_ELEMS / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the weight image of the bf16 tower (snk_conv3x3_bn_bf16_act16): the same layout with bf16 parts, activation scale 1
extern "C" int snk_conv3x3_prepare_weights_bf16(const float *d_w_hwio, void *d_wS, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS, "snk_conv3x3_prepare_weights_bf16: NULL argument");
    float *tail = (float *)((_Float16 *)d_wS + HS_WS_ELEMS);
    k_f16s_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail, 1.0f);
    k_a16_weights<true><<<(9 * HS_C * HS_C / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the weight image of the f16-activation tower (snk_conv3x3_bn_f16_act16): the same layout with f16 values (pre-scaled by the power of
// two that brings max |w| to [256, 512); undone in the epilogue), activation scale 1
extern "C" int snk_conv3x3_prepare_weights_f16_act16(const float *d_w_hwio, void *d_wS, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS, "snk_conv3x3_prepare_weights_f16_act16: NULL argument");
    float *tail = (float *)((_Float16 *)d_wS + HS_WS_ELEMS);
    k_f16s_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail, 1.0f);
    k_a16_weights<false><<<(9 * HS_C * HS_C / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the training step's form: the input scale is read on the device from d_in_tail = { ., ., scale, 1 / scale } (what
// snk_bn_train_apply / snk_bn_train_grad_apply measured while writing the tensor this layer reads); input_gradient != 0
// lays out the kernel of the layer's input gradient (the same convolution with mirrored taps and swapped channel axes);
// d_wS_same_kernel (optional): an image made from the same d_w_hwio values, whose weight scale is reused instead of
// scanning the kernel for its largest entry again
extern "C" int snk_conv3x3_prepare_weights_f16s_train(const float *d_w_hwio, void *d_wS, const float *d_in_tail, int input_gradient,
                                                      const void *d_wS_same_kernel, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS && d_in_tail && d_wS != d_wS_same_kernel, "snk_conv3x3_prepare_weights_f16s_train: bad argument");
    float *tail = (float *)((_Float16 *)d_wS + HS_WS_ELEMS);
    if (d_wS_same_kernel)
        k_f16s_tail_from<<<1, 64, 0, (hipStream_t)stream>>>((const float *)((const _Float16 *)d_wS_same_kernel + HS_WS_ELEMS), d_in_tail, tail);
    else k_f16s_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail, 1.0f, d_in_tail);
    if (input_gradient) k_f16s_weights<true><<<(HS_WS_ELEMS / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    else k_f16s_weights<false><<<(HS_WS_ELEMS / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// All tower layers of a training step at once (snake_engine/train_step.py): for every layer i the power-of-two weight scale (once
// per kernel, shared by its two images), the forward image and -- where h_wS_bwd[i] is given -- the input gradient's image, in TWO
// launches instead of four per layer and direction.  The images' input scales (tail[2..3]) are NOT set here: the training step has
// the element-wise kernel that writes a convolution's input leave them there directly (d_out_scale_tail / d_dx_scale_tail =
// the image's tail), so a weight image is made once per optimizer step and stays valid for as long as the weights do (every
// forward-only step at learning rate 0 reuses it).
#define HS_BATCH_MAX 40
struct F16sBatch { const float *w[HS_BATCH_MAX]; _Float16 *fwd[HS_BATCH_MAX]; _Float16 *bwd[HS_BATCH_MAX]; };

__global__ __launch_bounds__(1024) void k_f16s_wscale_batch(F16sBatch b)
{
    __shared__ float red[1024];
    const float4 *w4 = (const float4 *)b.w[blockIdx.x];
    float m = 0.f;
#pragma unroll 12
    for (int i = threadIdx.x; i < 9 * HS_C * HS_C / 4; i += 1024) {
        const float4 v = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x < 2) {
        _Float16 *img = threadIdx.x == 0 ? b.fwd[blockIdx.x] : b.bwd[blockIdx.x];
        if (!img) return;
        float *tail = (float *)(img + HS_WS_ELEMS);
        const float mx = red[0];
        int k = 0;
        if (mx > 0.f && mx < 3.0e38f) k = 8 - ilogbf(mx);          // as k_f16s_wscale
        k = max(-100, min(100, k));
        tail[0] = ldexpf(1.0f, -k);
        tail[1] = ldexpf(1.0f, k);
        tail[4] = tail[5] = tail[6] = tail[7] = 0.f;              // range flag, no guard word; tail[2..3] belong to the input's producer
    }
}

// grid (pieces / 256, layers, 2 directions); the tails' weight scales were written by k_f16s_wscale_batch (the launch before)
__global__ __launch_bounds__(256) void k_f16s_weights_batch(F16sBatch b)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;          // one f16x8 fragment piece
    const bool flip = blockIdx.z != 0;
    _Float16 *wS = flip ? b.bwd[blockIdx.y] : b.fwd[blockIdx.y];
    if (v >= HS_WS_ELEMS / 8 || !wS) return;
    const float *w = b.w[blockIdx.y];
    const float mul = ((const float *)(wS + HS_WS_ELEMS))[1];
    const int g = v / 512, rem = v - g * 512;
    const int c = g / 9, tap = g - 9 * c;                       // (the fragment order of k_f16s_weights)
    const int wn = rem >> 7, hl = (rem >> 6) & 1, lane = rem & 63;
    const int h = lane >> 5, l31 = lane & 31;
    const int cout = 32 * wn + l31;
    for (int j = 0; j < 8; ++j) {
        const int cin = HS_KC * c + 8 * h + j;
        const float val = (flip ? w[(long)((8 - tap) * HS_C + cout) * HS_C + cin] : w[(long)(tap * HS_C + cin) * HS_C + cout]) * mul;
        const _Float16 hi = (_Float16)val;
        wS[(long)v * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
    }
}

extern "C" int snk_conv3x3_prepare_weights_f16s_train_batch(const float *const *h_w_hwio, void *const *h_wS_fwd, void *const *h_wS_bwd,
                                                            int n_layers, void *stream)
{
    SNK_REQUIRE(h_w_hwio && h_wS_fwd && n_layers > 0 && n_layers <= HS_BATCH_MAX,
                "snk_conv3x3_prepare_weights_f16s_train_batch: 1 .. %d layers per call", HS_BATCH_MAX);
    F16sBatch b = {};
    bool any_bwd = false;
    for (int i = 0; i < n_layers; ++i) {
        SNK_REQUIRE(h_w_hwio[i] && h_wS_fwd[i], "snk_conv3x3_prepare_weights_f16s_train_batch: NULL kernel or image (layer %d)", i);
        b.w[i] = h_w_hwio[i];
        b.fwd[i] = (_Float16 *)h_wS_fwd[i];
        b.bwd[i] = h_wS_bwd ? (_Float16 *)h_wS_bwd[i] : nullptr;
        any_bwd = any_bwd || b.bwd[i];
    }
    k_f16s_wscale_batch<<<n_layers, 1024, 0, (hipStream_t)stream>>>(b);
    k_f16s_weights_batch<<<dim3((HS_WS_ELEMS / 8 + 255) / 256, n_layers, any_bwd ? 2 : 1), 256, 0, (hipStream_t)stream>>>(b);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

static int conv_f16s_launch(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                            const float *d_residual, float *d_out, const float *d_w1x1, float s1, float b1, float *d_h1,
                            int n_images, int height, int width, int relu, bool split, void *stream, int io16 = 0,
                            const float *d_center = nullptr, float *d_stat_part = nullptr, int *grid_out = nullptr, bool bf = false,
                            const float *d_gy = nullptr, const unsigned char *d_gmask = nullptr, const float *d_ginv = nullptr,
                            const float *d_aff_scale = nullptr, const float *d_aff_shift = nullptr, float *d_amax_part = nullptr,
                            const unsigned char *d_res_mask = nullptr)
{
    SNK_REQUIRE(n_images >= 0 && height >= 1 && width >= 3, "snk_conv3x3_bn_f16s: bad shape %d x %d x %d", n_images, height, width);
    SNK_REQUIRE(d_out != d_x, "snk_conv3x3_bn_f16s: in-place convolution is not possible (blocks read their neighbours' input rows)");
    SNK_REQUIRE(d_out != d_residual || d_res_mask, "snk_conv3x3_bn_f16s: the output may overwrite the shortcut in the masked-shortcut form only");
    if (n_images == 0) return 0;
    const int P = width + 1, HW = height * width;
    const bool a16 = (io16 & 1) != 0;                      // the 16-bit frame: LDS rows pitched in bytes, six staging items
    const int nst_px = 64 * (a16 ? HS_NST16 : HS_NST);
    // GEMM rows = the image's pixels in row-major order, 32 per M tile; an image is cut into n_blk blocks of at most 8 M
    // tiles whose input rows (those of its pixels + one above and below) fit the LDS buffer and the staging items
    const int T = (HW + 31) / 32;
    int n_blk = (T + 7) / 8, tiles_max = 0;
    bool fits = false;
    for (;; ++n_blk) {
        tiles_max = (T + n_blk - 1) / n_blk;
        const int rows_out = min((tiles_max * 32 + width - 2) / width + 1, height);       // worst alignment of 32 tiles_max pixels
        fits = (a16 ? (rows_out + 2) * (width * HS_LDP + HS_GAP16_FULL) <= HS_BUF16 : (rows_out + 2) * P + 1 <= HS_NPB) &&
               min(rows_out + 2, height) * width <= nst_px;
        if (fits || tiles_max == 1) break;
    }
    SNK_REQUIRE(fits, "snk_conv3x3_bn_f16s: observation width %d not supported (max 80)", width);
    // very small batches (measured: up to ~40 images at 21x21): cut the images finer so that the grid covers more of the
    // chip's block slots -- a block's duration, not the throughput, is what such a launch costs (16 states: 31 -> 14 us);
    // from ~48 images on the 7-tile blocks with their compile-time epilogues are faster again
    static const int fine_max = getenv("SNK_CONV_FINE_MAX") ? atoi(getenv("SNK_CONV_FINE_MAX")) : 40;
    if (n_images <= fine_max && n_images * n_blk < 512) {
        n_blk = min(T, max(n_blk, (512 + n_images - 1) / n_images));
        tiles_max = (T + n_blk - 1) / n_blk;
    }
    const int tiles_base = T / n_blk, tiles_rem = T % n_blk;
    SNK_REQUIRE((long)n_images * n_blk < (1l << 31) && (long)HW * HS_C < (1l << 31), "snk_conv3x3_bn_f16s: batch too large");
    ConvHsArgs a = {d_x, (const f16x8 *)d_wS, (const float *)((const _Float16 *)d_wS + HS_WS_ELEMS), d_scale, d_shift,
                    d_residual, d_out, d_w1x1, d_h1, s1, b1, height, width, n_blk, tiles_base, tiles_rem, relu, (n_images / 8) * 8,
                    d_center, d_stat_part, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, d_gy, d_gmask, d_ginv,
                    d_aff_scale, d_aff_shift, d_amax_part, d_res_mask};
    const int n_mt = tiles_max, grid = n_images * n_blk;
    hipStream_t st = (hipStream_t)stream;
    if (grid_out) *grid_out = grid;
#define HS_LAUNCH_TRAIN(MODE_)                                                                   \
        switch (n_mt) {                                                                         \
        case 1: k_conv3x3_f16s<1, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 2: k_conv3x3_f16s<2, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 3: k_conv3x3_f16s<3, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 4: k_conv3x3_f16s<4, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 5: k_conv3x3_f16s<5, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 6: k_conv3x3_f16s<6, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        case 7: k_conv3x3_f16s<7, MODE_><<<grid, 256, 0, st>>>(a); break;                       \
        default: k_conv3x3_f16s<8, MODE_><<<grid, 256, 0, st>>>(a); break;                      \
        }                                                                                       \
        SNK_CHECK_HIP(hipGetLastError());                                                       \
        return 0;
    SNK_REQUIRE(!(d_stat_part && a.g_y && d_aff_scale && d_residual && !d_res_mask),
                "snk_conv3x3_f16s_igrad_stats_deferred: the layer below a block's second convolution has no shortcut gradient to add");
    if (d_stat_part && a.g_y && d_res_mask && d_aff_scale) { HS_LAUNCH_TRAIN(10) }   // ... and the ReLU decision of the sums recomputed (deferred stem below)
    if (d_stat_part && a.g_y && d_res_mask) { HS_LAUNCH_TRAIN(8) }     // input gradient + sums, the shortcut's gradient taken through ReLU bits
    if (d_stat_part && a.g_y && d_aff_scale) { HS_LAUNCH_TRAIN(7) }    // ... of a layer whose ReLU decision is recomputed from its scale / shift
    if (d_stat_part && d_aff_scale && d_amax_part) { HS_LAUNCH_TRAIN(9) }
    if (d_stat_part && d_aff_scale) { HS_LAUNCH_TRAIN(6) }             // forward pass, the producer's batch norm + ReLU applied on the way in
#undef HS_LAUNCH_TRAIN
    if (d_stat_part && a.g_y) {          // the training step's input gradient + the previous layer's batch-norm backward sums
        switch (n_mt) {
        case 1: k_conv3x3_f16s<1, 5><<<grid, 256, 0, st>>>(a); break;
        case 2: k_conv3x3_f16s<2, 5><<<grid, 256, 0, st>>>(a); break;
        case 3: k_conv3x3_f16s<3, 5><<<grid, 256, 0, st>>>(a); break;
        case 4: k_conv3x3_f16s<4, 5><<<grid, 256, 0, st>>>(a); break;
        case 5: k_conv3x3_f16s<5, 5><<<grid, 256, 0, st>>>(a); break;
        case 6: k_conv3x3_f16s<6, 5><<<grid, 256, 0, st>>>(a); break;
        case 7: k_conv3x3_f16s<7, 5><<<grid, 256, 0, st>>>(a); break;
        default: k_conv3x3_f16s<8, 5><<<grid, 256, 0, st>>>(a); break;
        }
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (d_stat_part) {                   // the training step's forward pass: bare convolution + batch-norm sums
        switch (n_mt) {
        case 1: k_conv3x3_f16s<1, 4><<<grid, 256, 0, st>>>(a); break;
        case 2: k_conv3x3_f16s<2, 4><<<grid, 256, 0, st>>>(a); break;
        case 3: k_conv3x3_f16s<3, 4><<<grid, 256, 0, st>>>(a); break;
        case 4: k_conv3x3_f16s<4, 4><<<grid, 256, 0, st>>>(a); break;
        case 5: k_conv3x3_f16s<5, 4><<<grid, 256, 0, st>>>(a); break;
        case 6: k_conv3x3_f16s<6, 4><<<grid, 256, 0, st>>>(a); break;
        case 7: k_conv3x3_f16s<7, 4><<<grid, 256, 0, st>>>(a); break;
        default: k_conv3x3_f16s<8, 4><<<grid, 256, 0, st>>>(a); break;
        }
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // the three epilogue shapes the net wrapper uses get compile-time versions at the 21x21 tile count; everything else
    // takes the generic version
    if (!split && io16) {        // f16 activations in HBM (io16: 1 = f16 in, f32 out; 3 = f16 in and out)
#define HS_LAUNCH_IO4(NI_, MODE_)                                                               \
        if (bf && io16 == 3) k_conv3x3_f16s<NI_, MODE_, false, 3, true><<<grid, 256, 0, st>>>(a); \
        else if (bf) k_conv3x3_f16s<NI_, MODE_, false, 1, true><<<grid, 256, 0, st>>>(a);       \
        else if (io16 == 3) k_conv3x3_f16s<NI_, MODE_, false, 3><<<grid, 256, 0, st>>>(a);      \
        else k_conv3x3_f16s<NI_, MODE_, false, 1><<<grid, 256, 0, st>>>(a);
// the tower's two shapes (ReLU without / with shortcut) have compile-time epilogues; anything else takes the generic one
#define HS_LAUNCH_IO(NI_)                                                                       \
    case NI_:                                                                                   \
        if (relu && !d_residual) { HS_LAUNCH_IO4(NI_, 1) }                                      \
        else if (relu) { HS_LAUNCH_IO4(NI_, 2) }                                                \
        else { HS_LAUNCH_IO4(NI_, 0) }                                                          \
        break;
        if (d_w1x1) {            // the tower's last layer with the head's 1x1 stage in its epilogue: no layer output
            SNK_REQUIRE(io16 == 1 && relu && d_residual && !d_out && d_h1, "snk_conv3x3_bn_*_act16_head: the fused head closes a residual block");
#define HS_LAUNCH_HEAD(NI_) case NI_: if (bf) k_conv3x3_f16s<NI_, 3, false, 1, true><<<grid, 256, 0, st>>>(a); \
                                      else k_conv3x3_f16s<NI_, 3, false, 1><<<grid, 256, 0, st>>>(a); break;
            switch (n_mt) {
                HS_LAUNCH_HEAD(1) HS_LAUNCH_HEAD(2) HS_LAUNCH_HEAD(3) HS_LAUNCH_HEAD(4) HS_LAUNCH_HEAD(5) HS_LAUNCH_HEAD(6) HS_LAUNCH_HEAD(7)
            default: if (bf) k_conv3x3_f16s<8, 3, false, 1, true><<<grid, 256, 0, st>>>(a);
                     else k_conv3x3_f16s<8, 3, false, 1><<<grid, 256, 0, st>>>(a); break;
            }
#undef HS_LAUNCH_HEAD
            SNK_CHECK_HIP(hipGetLastError());
            return 0;
        }
        switch (n_mt) {
            HS_LAUNCH_IO(1) HS_LAUNCH_IO(2) HS_LAUNCH_IO(3) HS_LAUNCH_IO(4) HS_LAUNCH_IO(5) HS_LAUNCH_IO(6) HS_LAUNCH_IO(7)
        default:
            if (relu && !d_residual) { HS_LAUNCH_IO4(8, 1) }
            else if (relu) { HS_LAUNCH_IO4(8, 2) }
            else { HS_LAUNCH_IO4(8, 0) }
            break;
        }
#undef HS_LAUNCH_IO4
#undef HS_LAUNCH_IO
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (!split) {
        switch (n_mt) {
        case 1: k_conv3x3_f16s<1, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 2: k_conv3x3_f16s<2, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 3: k_conv3x3_f16s<3, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 4: k_conv3x3_f16s<4, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 5: k_conv3x3_f16s<5, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 6: k_conv3x3_f16s<6, 0, false><<<grid, 256, 0, st>>>(a); break;
        case 7: k_conv3x3_f16s<7, 0, false><<<grid, 256, 0, st>>>(a); break;
        default: k_conv3x3_f16s<8, 0, false><<<grid, 256, 0, st>>>(a); break;
        }
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int mode = !relu ? 0 : (d_w1x1 ? (d_residual && !d_out ? 3 : 0) : (d_residual ? 2 : 1));
#define HS_LAUNCH_MODES(NI_)                                                                    \
    case NI_:                                                                                   \
        if (mode == 1) k_conv3x3_f16s<NI_, 1><<<grid, 256, 0, st>>>(a);                         \
        else if (mode == 2) k_conv3x3_f16s<NI_, 2><<<grid, 256, 0, st>>>(a);                    \
        else if (mode == 3) k_conv3x3_f16s<NI_, 3><<<grid, 256, 0, st>>>(a);                    \
        else k_conv3x3_f16s<NI_, 0><<<grid, 256, 0, st>>>(a);                                   \
        break;
    switch (n_mt) {
        HS_LAUNCH_MODES(1) HS_LAUNCH_MODES(2) HS_LAUNCH_MODES(3) HS_LAUNCH_MODES(4)
        HS_LAUNCH_MODES(5) HS_LAUNCH_MODES(6) HS_LAUNCH_MODES(7)
    default:
        if (mode == 1) k_conv3x3_f16s<8, 1><<<grid, 256, 0, st>>>(a);
        else if (mode == 2) k_conv3x3_f16s<8, 2><<<grid, 256, 0, st>>>(a);
        else if (mode == 3) k_conv3x3_f16s<8, 3><<<grid, 256, 0, st>>>(a);
        else k_conv3x3_f16s<8, 0><<<grid, 256, 0, st>>>(a);
        break;
    }
#undef HS_LAUNCH_MODES
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- planning of the sub-rectangle form ------------------------------------------------------------------------------
// parts (blocks) a rectangle of hr x wr pixels is cut into and the M tiles of its largest part: the fewest parts of at most
// 8 tiles whose strip (their rows, one above, one below, halo columns) fits the LDS buffer and the staging items
// (a16: the 16-bit towers' frame -- rows pitched in bytes with a gap, six staging items)
__host__ __device__ static inline int hs_rect_parts(int hr, int wr, int Hd, int Wd, int *tiles_max, bool a16 = false)
{
    const int nst_px = 64 * (a16 ? HS_NST16 : HS_NST);
    const int T = (hr * wr + 31) / 32;
    for (int parts = (T + 7) / 8;; ++parts) {
        const int tm = (T + parts - 1) / parts;
        int rows_out = (tm * 32 + wr - 2) / wr + 1;                       // worst alignment of 32 tm pixels
        if (rows_out > hr) rows_out = hr;
        const int rows_in = rows_out + 2 < Hd ? rows_out + 2 : Hd, cols_in = wr + 2 < Wd ? wr + 2 : Wd;
        const bool fits = (a16 ? (rows_out + 2) * ((wr + 2) * HS_LDP + HS_GAP16_RECT) <= HS_BUF16 : (rows_out + 2) * (wr + 2) <= HS_NPB) &&
                          rows_in * cols_in <= nst_px;
        if (fits || tm == 1) { *tiles_max = tm; return fits ? parts : -1; }
    }
}

// bounding box of the pixels of an observation [H][W][3] that differ from the background pixel (b0, b1, b2): one wavefront
// per observation; an observation without such a pixel gets the centre pixel
__global__ __launch_bounds__(256) void k_obs_bbox(const float *__restrict__ planes, int n, int Hd, int Wd, float b0, float b1, float b2,
                                                  unsigned *__restrict__ bbox)
{
    const int img = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (img >= n) return;
    const float *x = planes + (long)img * Hd * Wd * 3;
    const float invW = 1.0f / (float)Wd;
    int y0 = 255, x0 = 255, y1 = -1, x1 = -1;
    for (int q = lane; q < Hd * Wd; q += 64) {
        const float v0 = x[3 * q], v1 = x[3 * q + 1], v2 = x[3 * q + 2];
        if (!(v0 == b0 && v1 == b1 && v2 == b2)) {
            const int y_ = (int)(((float)q + 0.5f) * invW), x_ = q - y_ * Wd;
            y0 = min(y0, y_); y1 = max(y1, y_); x0 = min(x0, x_); x1 = max(x1, x_);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        y0 = min(y0, __shfl_xor(y0, s, 64)); x0 = min(x0, __shfl_xor(x0, s, 64));
        y1 = max(y1, __shfl_xor(y1, s, 64)); x1 = max(x1, __shfl_xor(x1, s, 64));
    }
    if (lane == 0) {
        if (y1 < 0) { y0 = y1 = Hd / 2; x0 = x1 = Wd / 2; }
        bbox[img] = (unsigned)y0 | (unsigned)x0 << 8 | (unsigned)y1 << 16 | (unsigned)x1 << 24;
    }
}

#define HS_RECT_MAX_LAYERS 24
struct RectPlanArgs {
    int n_layers;
    int grow[HS_RECT_MAX_LAYERS];          // the layer's rectangle = bounding box grown by this many pixels, cut to the canvas
    uint4 *desc;                           // [n_layers][max_blocks]
    long max_blocks;
    int *counts;                           // [n_layers][2]: descriptors written, M tiles they cover
    bool a16;                              // the descriptors are cut for the 16-bit towers' block frame
};

// One workgroup per layer: every image's rectangle -> its parts -> block descriptors, largest blocks first (bins by the
// image's largest part; the order inside a bin is whatever the atomics give -- no result depends on it)
__global__ __launch_bounds__(1024) void k_rect_plan(const unsigned *__restrict__ bbox, int n, int Hd, int Wd, RectPlanArgs a)
{
    __shared__ int cnt[9], cur[9], tiles_sum;
    const int L = blockIdx.x, g = a.grow[L], tid = threadIdx.x;
    if (tid < 9) cnt[tid] = 0;
    if (tid == 0) tiles_sum = 0;
    __syncthreads();
    int tiles = 0;
    for (int img = tid; img < n; img += 1024) {
        const unsigned b = bbox[img];
        const int y0 = max((int)(b & 255) - g, 0), x0 = max((int)((b >> 8) & 255) - g, 0);
        const int y1 = min((int)((b >> 16) & 255) + g, Hd - 1), x1 = min((int)(b >> 24) + g, Wd - 1);
        int tm;
        const int parts = hs_rect_parts(y1 - y0 + 1, x1 - x0 + 1, Hd, Wd, &tm, a.a16);
        atomicAdd(&cnt[tm], parts);
        tiles += ((y1 - y0 + 1) * (x1 - x0 + 1) + 31) / 32;
    }
    atomicAdd(&tiles_sum, tiles);
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int tm = 8; tm >= 1; --tm) { cur[tm] = o; o += cnt[tm]; }
        a.counts[2 * L] = o;
        a.counts[2 * L + 1] = tiles_sum;
    }
    __syncthreads();
    uint4 *desc = a.desc + (long)L * a.max_blocks;
    for (int img = tid; img < n; img += 1024) {
        const unsigned b = bbox[img];
        const int y0 = max((int)(b & 255) - g, 0), x0 = max((int)((b >> 8) & 255) - g, 0);
        const int y1 = min((int)((b >> 16) & 255) + g, Hd - 1), x1 = min((int)(b >> 24) + g, Wd - 1);
        const int hr = y1 - y0 + 1, wr = x1 - x0 + 1;
        int tm;
        const int parts = hs_rect_parts(hr, wr, Hd, Wd, &tm, a.a16);
        const int T = (hr * wr + 31) / 32, base = T / parts, rem = T % parts;
        const unsigned rect = (unsigned)y0 | (unsigned)x0 << 8 | (unsigned)hr << 16 | (unsigned)wr << 24;
        const int slot = atomicAdd(&cur[tm], parts);
        for (int k = 0; k < parts; ++k) {
            const int tile0 = k * base + min(k, rem), ntile = base + (k < rem ? 1 : 0);
            desc[slot + k] = make_uint4((unsigned)img, rect, (unsigned)tile0 | (unsigned)ntile << 8 | (unsigned)k << 16 | (unsigned)parts << 24, b);
        }
    }
}

static int rect_max_parts(int height, int width, bool act16)
{
    static int cache_h[2] = {0, 0}, cache_w[2] = {0, 0}, cache_v[2] = {0, 0};
    const int f = act16 ? 1 : 0;
    if (cache_h[f] == height && cache_w[f] == width) return cache_v[f];
    int worst = 1;
    for (int hr = 1; hr <= height; ++hr)
        for (int wr = 1; wr <= width; ++wr) {
            int tm;
            const int parts = hs_rect_parts(hr, wr, height, width, &tm, act16);
            if (parts < 0) return -1;
            if (parts > worst) worst = parts;
        }
    cache_h[f] = height; cache_w[f] = width; cache_v[f] = worst;
    return worst;
}

static long rect_max_blocks(int n_images, int height, int width, bool act16)
{
    if (n_images < 0 || height < 3 || width < 3 || height > 80 || width > 80) return -1;
    const int mp = rect_max_parts(height, width, act16);
    return mp < 0 ? -1 : (long)n_images * mp;
}

extern "C" long snk_conv_rect_max_blocks(int n_images, int height, int width) { return rect_max_blocks(n_images, height, width, false); }
// the towers with 16-bit activations cut rectangles for their own block frame (more pixels per LDS buffer and per staging pass)
extern "C" long snk_conv_rect_max_blocks_act16(int n_images, int height, int width) { return rect_max_blocks(n_images, height, width, true); }

static int rect_plan(const float *d_planes, float b0, float b1, float b2, int n_images, int height, int width,
                     int n_layers, const int *grow, void *d_bbox, void *d_desc, int *d_counts, void *stream, bool act16)
{
    SNK_REQUIRE(d_planes && grow && d_bbox && d_desc && d_counts, "snk_conv_rect_plan: NULL argument");
    SNK_REQUIRE(n_layers >= 1 && n_layers <= HS_RECT_MAX_LAYERS, "snk_conv_rect_plan: %d layers (at most %d)", n_layers, HS_RECT_MAX_LAYERS);
    const long mb = rect_max_blocks(n_images, height, width, act16);
    SNK_REQUIRE(mb >= 0 && mb < (1l << 31), "snk_conv_rect_plan: bad shape %d x %d x %d", n_images, height, width);
    if (n_images == 0) return 0;
    RectPlanArgs a;
    a.n_layers = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        SNK_REQUIRE(grow[i] >= 0 && grow[i] < 128, "snk_conv_rect_plan: grow[%d] = %d", i, grow[i]);
        a.grow[i] = grow[i];
    }
    a.desc = (uint4 *)d_desc; a.max_blocks = mb; a.counts = d_counts;
    a.a16 = act16;
    hipStream_t st = (hipStream_t)stream;
    k_obs_bbox<<<(n_images + 3) / 4, 256, 0, st>>>(d_planes, n_images, height, width, b0, b1, b2, (unsigned *)d_bbox);
    k_rect_plan<<<n_layers, 1024, 0, st>>>((const unsigned *)d_bbox, n_images, height, width, a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv_rect_plan(const float *d_planes, float b0, float b1, float b2, int n_images, int height, int width,
                                  int n_layers, const int *grow, void *d_bbox, void *d_desc, int *d_counts, void *stream)
{
    return rect_plan(d_planes, b0, b1, b2, n_images, height, width, n_layers, grow, d_bbox, d_desc, d_counts, stream, false);
}

// the plan of the towers with 16-bit activations (descriptor array: snk_conv_rect_max_blocks_act16 per layer)
extern "C" int snk_conv_rect_plan_act16(const float *d_planes, float b0, float b1, float b2, int n_images, int height, int width,
                                        int n_layers, const int *grow, void *d_bbox, void *d_desc, int *d_counts, void *stream)
{
    return rect_plan(d_planes, b0, b1, b2, n_images, height, width, n_layers, grow, d_bbox, d_desc, d_counts, stream, true);
}

static int conv_f16s_rect_launch(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                 const float *d_residual, float *d_out, const void *d_desc, const int *d_count,
                                 const float *d_bg_in, int grow_in, const float *d_bg_res, int grow_res, const float *d_bg_out,
                                 int n_images, int height, int width, bool act16, void *stream, bool bf = false)
{
    SNK_REQUIRE(grow_in >= 0 && grow_in < 128 && grow_res >= 0 && grow_res < 128, "snk_conv3x3_bn_f16s_rect: grow_in %d, grow_res %d", grow_in, grow_res);
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_out && d_desc && d_count, "snk_conv3x3_bn_f16s_rect: NULL argument");
    SNK_REQUIRE(d_out != d_x, "snk_conv3x3_bn_f16s_rect: in-place convolution is not possible");
    const long mb = rect_max_blocks(n_images, height, width, act16);
    SNK_REQUIRE(mb >= 0 && mb < (1l << 31) && (long)height * width * HS_C < (1l << 31), "snk_conv3x3_bn_f16s_rect: bad shape %d x %d x %d",
                n_images, height, width);
    if (n_images == 0) return 0;
    ConvHsArgs a = {d_x, (const f16x8 *)d_wS, (const float *)((const _Float16 *)d_wS + HS_WS_ELEMS), d_scale, d_shift,
                    d_residual, d_out, nullptr, nullptr, 0.f, 0.f, height, width, 1, 0, 0, 1, 0, nullptr, nullptr,
                    (const uint4 *)d_desc, d_count, d_bg_out, d_bg_in, d_residual ? d_bg_res : nullptr, grow_in, grow_res, nullptr, nullptr, nullptr};
    if (act16 && bf) {
        if (d_residual) k_conv3x3_f16s_rect<2, false, 3, true><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
        else k_conv3x3_f16s_rect<1, false, 3, true><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
    } else if (act16) {
        if (d_residual) k_conv3x3_f16s_rect<2, false, 3><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
        else k_conv3x3_f16s_rect<1, false, 3><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
    } else if (d_residual) k_conv3x3_f16s_rect<2><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
    else k_conv3x3_f16s_rect<1><<<(int)mb, 256, 0, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_bn_f16s_rect(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                        const float *d_residual, float *d_out, const void *d_desc, const int *d_count,
                                        const float *d_bg_in, int grow_in, const float *d_bg_res, int grow_res,
                                        const float *d_bg_out, int n_images, int height, int width, void *stream)
{
    return conv_f16s_rect_launch(d_x, d_wS, d_scale, d_shift, d_residual, d_out, d_desc, d_count, d_bg_in, grow_in, d_bg_res, grow_res,
                                 d_bg_out, n_images, height, width, false, stream);
}

// the sub-rectangle form of snk_conv3x3_bn_f16_act16 (f16 activations in and out, relu = 1; the background images are f16
// [height][width][128])
extern "C" int snk_conv3x3_bn_f16_act16_rect(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                             const void *d_residual16, void *d_out16, const void *d_desc, const int *d_count,
                                             const void *d_bg_in16, int grow_in, const void *d_bg_res16, int grow_res,
                                             const void *d_bg_out16, int n_images, int height, int width, void *stream)
{
    return conv_f16s_rect_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, (float *)d_out16, d_desc,
                                 d_count, (const float *)d_bg_in16, grow_in, (const float *)d_bg_res16, grow_res,
                                 (const float *)d_bg_out16, n_images, height, width, true, stream);
}

// the same for the bf16 tower (bf16 activations in and out, bf16 background images, weights = snk_conv3x3_prepare_weights_bf16)
extern "C" int snk_conv3x3_bn_bf16_act16_rect(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                              const void *d_residual16, void *d_out16, const void *d_desc, const int *d_count,
                                              const void *d_bg_in16, int grow_in, const void *d_bg_res16, int grow_res,
                                              const void *d_bg_out16, int n_images, int height, int width, void *stream)
{
    return conv_f16s_rect_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, (float *)d_out16, d_desc,
                                 d_count, (const float *)d_bg_in16, grow_in, (const float *)d_bg_res16, grow_res,
                                 (const float *)d_bg_out16, n_images, height, width, true, stream, true);
}

extern "C" int snk_conv3x3_bn_f16s(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                   const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                                   void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_f16s: NULL argument");
    return conv_f16s_launch(d_x, d_wS, d_scale, d_shift, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width,
                            relu, true, stream);
}

extern "C" int snk_conv3x3_bn_f16(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                  const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                                  void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_f16: NULL argument");
    return conv_f16s_launch(d_x, d_wS, d_scale, d_shift, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width,
                            relu, false, stream);
}

extern "C" int snk_conv3x3_bn_f16s_head(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                        const float *d_residual, float *d_out, const float *d_w1x1, float bn_scale,
                                        float bn_shift, float *d_h1, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_w1x1 && d_h1, "snk_conv3x3_bn_f16s_head: NULL argument");
    return conv_f16s_launch(d_x, d_wS, d_scale, d_shift, d_residual, d_out, d_w1x1, bn_scale, bn_shift, d_h1, n_images, height,
                            width, 1, true, stream);
}

// The training step's forward convolution: d_out = conv3x3_same(d_x, w) (no batch norm, no ReLU) and, from the same values on
// their way out, d_sums[0..127] = sum over all pixels of (out - center), d_sums[128..255] = sum of (out - center)^2 (float64;
// d_center: 128 floats or NULL) -- what snk_bn_train_sums_f64 would compute from d_out in a pass of its own.
// d_partials: snk_conv3x3_stats_partials(n_images, height, width) floats.
#define HS_AMAX_STAGE 64                                 // rows the first of the two maximum folds leaves
extern "C" long snk_conv3x3_stats_partials(int n_images, int height, int width)
{
    if (n_images <= 0 || height < 1 || width < 3) return -1;
    const long T = ((long)height * width + 31) / 32;
    // at most one block per M tile: 2 x 128 sums + 128 maxima each; the two folds' scratch
    return (long)n_images * T * 384 + TF_SCRATCH_FLOATS(256) + HS_AMAX_STAGE * HS_C;
}

// per-channel maximum over the blocks' maxima: [n_rows][128] -> [gridDim.x][128] (whole 512-byte rows per eight-lane group, block b
// takes rows b, b + gridDim.x, ..); run twice: the blocks' maxima -> 64 rows -> 1
__global__ __launch_bounds__(256) void k_amax_fold128(const float *__restrict__ part, int n_rows, float *__restrict__ out)
{
    __shared__ float4 sh[8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = blockIdx.x * 8 + rl; r < n_rows; r += gridDim.x * 8) {
        const float4 v = *(const float4 *)(part + (size_t)r * HS_C + 4 * cq);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    sh[rl][cq] = m;
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sh[r][cq]; m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w); }
        *(float4 *)(out + (size_t)blockIdx.x * HS_C + 4 * cq) = m;
    }
}

extern "C" int snk_conv3x3_f16s_stats(const float *d_x, const void *d_wS, float *d_out, const float *d_center, float *d_partials,
                                      double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_out && d_partials && d_sums && n_images > 0, "snk_conv3x3_f16s_stats: bad argument");
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, nullptr, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_center, d_partials, &grid);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// The same with the two options of the deferred batch norm (snake_engine/train_step.py: the activation between the two convolutions
// of a residual block is never written out):
//   d_in_scale / d_in_shift (both or neither): d_x is the PRE-batch-norm output of the layer below and every value is taken as
//     relu(x * in_scale[c] + in_shift[c]) on its way into the kernel -- bit for bit what snk_bn_train_apply(relu = 1, no residual)
//     would have written; the input scale in d_wS's tail must cover that range (snk_bn_train_finalize_range);
//   d_amax (or NULL; with either input form): 128 floats, the largest |out - center| per channel -- what snk_bn_train_finalize_range turns into the
//     range of this layer's own batch norm + ReLU output without a pass over it.
extern "C" int snk_conv3x3_f16s_stats_deferred(const float *d_x, const void *d_wS, float *d_out, const float *d_center,
                                               const float *d_in_scale, const float *d_in_shift, float *d_amax, float *d_partials,
                                               double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_out && d_partials && d_sums && n_images > 0 && !d_in_scale == !d_in_shift,
                "snk_conv3x3_f16s_stats_deferred: bad argument");
    // the launch's grid is known only inside: the maxima sit behind the largest sums + fold scratch the buffer is sized for
    const long T = ((long)height * width + 31) / 32;
    float *amax_part = d_amax ? d_partials + (long)n_images * T * 256 + TF_SCRATCH_FLOATS(256) : nullptr;
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, nullptr, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_center, d_partials, &grid, false, nullptr, nullptr, nullptr, d_in_scale, d_in_shift,
                                    amax_part);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    if (d_amax) {
        float *stage = d_partials + (long)n_images * T * 384 + TF_SCRATCH_FLOATS(256);
        k_amax_fold128<<<HS_AMAX_STAGE, 256, 0, (hipStream_t)stream>>>(amax_part, grid, stage);
        k_amax_fold128<<<1, 256, 0, (hipStream_t)stream>>>(stage, HS_AMAX_STAGE, d_amax);
    }
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// reduced precision with f16 activations in HBM (BASELINE configs[4]): d_x16 / d_residual16 are f16 [n][H][W][128]; the output
// is f16 (out_f16 != 0: every tower layer but the last) or float32 (the layer the head reads).  Weights: the image
// snk_conv3x3_prepare_weights_f16s makes with x_scale = 1.
extern "C" int snk_conv3x3_bn_f16_act16(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                        const void *d_residual16, void *d_out, int out_f16, int n_images, int height, int width,
                                        int relu, void *stream)
{
    SNK_REQUIRE(d_x16 && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_f16_act16: NULL argument");
    return conv_f16s_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, (float *)d_out, nullptr, 0.f,
                            0.f, nullptr, n_images, height, width, relu, false, stream, out_f16 ? 3 : 1);
}

// BASELINE configs[4] as it is worded, "bf16 MFMA conv": bf16 activations in HBM (d_x16 / d_residual16: bf16 [n][H][W][128]; the
// output is bf16, or float32 for the layer the head reads), bf16 weights (snk_conv3x3_prepare_weights_bf16), products on
// v_mfma_f32_32x32x16_bf16, float32 accumulation, batch norm / shortcut / ReLU in float32.  The block body is hs_block: the same
// staging, LDS image, fragment pipeline and epilogue as the f16-activation tower.
extern "C" int snk_conv3x3_bn_bf16_act16(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                         const void *d_residual16, void *d_out, int out_bf16, int n_images, int height, int width,
                                         int relu, void *stream)
{
    SNK_REQUIRE(d_x16 && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_bf16_act16: NULL argument");
    return conv_f16s_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, (float *)d_out, nullptr, 0.f,
                            0.f, nullptr, n_images, height, width, relu, false, stream, out_bf16 ? 3 : 1, nullptr, nullptr, nullptr, true);
}

// The last layer of the towers with 16-bit activations with the head's 1x1 stage in its epilogue (alpha_nnet.py:46-50), as
// snk_conv3x3_bn_f16s_head does for the float32 tower: d_h1[n][height * width] = relu(dot(relu(bn(conv) + shortcut)[pixel][:],
// w1x1) * bn_scale + bn_shift); the layer's own output never goes to HBM (in float32 it would be twice the bytes of every other
// activation of these towers).  snk_head_dense_f32 finishes AlphaNNet.v from d_h1.
extern "C" int snk_conv3x3_bn_f16_act16_head(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                             const void *d_residual16, const float *d_w1x1, float bn_scale, float bn_shift,
                                             float *d_h1, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x16 && d_wS && d_scale && d_shift && d_residual16 && d_w1x1 && d_h1, "snk_conv3x3_bn_f16_act16_head: NULL argument");
    return conv_f16s_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, nullptr, d_w1x1, bn_scale,
                            bn_shift, d_h1, n_images, height, width, 1, false, stream, 1);
}

extern "C" int snk_conv3x3_bn_bf16_act16_head(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                              const void *d_residual16, const float *d_w1x1, float bn_scale, float bn_shift,
                                              float *d_h1, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x16 && d_wS && d_scale && d_shift && d_residual16 && d_w1x1 && d_h1, "snk_conv3x3_bn_bf16_act16_head: NULL argument");
    return conv_f16s_launch((const float *)d_x16, d_wS, d_scale, d_shift, (const float *)d_residual16, nullptr, d_w1x1, bn_scale,
                            bn_shift, d_h1, n_images, height, width, 1, false, stream, 1, nullptr, nullptr, nullptr, true);
}

// The training step's INPUT-GRADIENT convolution with the next batch-norm backward's sums in its epilogue: d_out = conv3x3_same(d_x,
// mirrored kernel) (+ d_residual, the shortcut's gradient) is the gradient at the output of the layer below; from the same values
// on their way out d_sums[0..127] = sum(g), d_sums[128..255] = sum(g * (y - mean) * inv) with g = d_out where that layer's ReLU bit
// is set (d_mask: one byte per quad of channels, as snk_bn_train_apply writes them; d_y: its pre-batch-norm output) -- what
// snk_bn_train_grad_sums_f64 computes from d_out in a pass of its own.  d_partials: snk_conv3x3_stats_partials floats.
extern "C" int snk_conv3x3_f16s_igrad_stats(const float *d_x, const void *d_wS, const float *d_residual, float *d_out, const float *d_y,
                                            const uint8_t *d_mask, const float *d_mean, const float *d_inv, float *d_partials,
                                            double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_out && d_y && d_mask && d_mean && d_inv && d_partials && d_sums && n_images > 0,
                "snk_conv3x3_f16s_igrad_stats: bad argument");
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_mean, d_partials, &grid, false, d_y, d_mask, d_inv);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// The same for a layer below whose batch norm + ReLU output was never written (deferred): its ReLU decision is
// d_y * d_scale[c] + d_shift[c] > 0 -- the sign of what snk_bn_train_apply would have written -- instead of mask bytes.
extern "C" int snk_conv3x3_f16s_igrad_stats_deferred(const float *d_x, const void *d_wS, const float *d_residual, float *d_out,
                                                     const float *d_y, const float *d_scale, const float *d_shift, const float *d_mean,
                                                     const float *d_inv, float *d_partials, double *d_sums, int n_images, int height,
                                                     int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_out && d_y && d_scale && d_shift && d_mean && d_inv && d_partials && d_sums && n_images > 0,
                "snk_conv3x3_f16s_igrad_stats_deferred: bad argument");
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_mean, d_partials, &grid, false, d_y, nullptr, d_inv, d_scale, d_shift);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_conv3x3_f16s_igrad_stats whose shortcut gradient is d_residual WHERE d_residual_mask's bit is set: d_residual = the gradient
// at the residual block's output (before that output's ReLU), d_residual_mask = that output's ReLU bits (snk_bn_train_apply's
// bytes) -- the masked copy snk_bn_train_grad_apply(d_g) writes is not needed.  d_out may be d_residual (in place).
extern "C" int snk_conv3x3_f16s_igrad_stats_masked_res(const float *d_x, const void *d_wS, const float *d_residual,
                                                       const uint8_t *d_residual_mask, float *d_out, const float *d_y,
                                                       const uint8_t *d_mask, const float *d_mean, const float *d_inv, float *d_partials,
                                                       double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_residual && d_residual_mask && d_out && d_y && d_mask && d_mean && d_inv && d_partials && d_sums &&
                n_images > 0, "snk_conv3x3_f16s_igrad_stats_masked_res: bad argument");
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_mean, d_partials, &grid, false, d_y, d_mask, d_inv, nullptr, nullptr, nullptr,
                                    d_residual_mask);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// snk_conv3x3_f16s_igrad_stats_masked_res for a layer below whose batch norm + ReLU output was never written (the deferred stem):
// the ReLU decision of the SUMS is d_y * d_scale[c] + d_shift[c] > 0 instead of mask bytes
extern "C" int snk_conv3x3_f16s_igrad_stats_masked_res_deferred(const float *d_x, const void *d_wS, const float *d_residual,
                                                                const uint8_t *d_residual_mask, float *d_out, const float *d_y,
                                                                const float *d_scale, const float *d_shift, const float *d_mean,
                                                                const float *d_inv, float *d_partials, double *d_sums, int n_images,
                                                                int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_residual && d_residual_mask && d_out && d_y && d_scale && d_shift && d_mean && d_inv && d_partials &&
                d_sums && n_images > 0, "snk_conv3x3_f16s_igrad_stats_masked_res_deferred: bad argument");
    int grid = 0;
    const int rc = conv_f16s_launch(d_x, d_wS, nullptr, nullptr, d_residual, d_out, nullptr, 0.f, 0.f, nullptr, n_images, height, width, 0,
                                    true, stream, 0, d_mean, d_partials, &grid, false, d_y, nullptr, d_inv, d_scale, d_shift, nullptr,
                                    d_residual_mask);
    if (rc) return rc;
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + (long)grid * 256), (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
