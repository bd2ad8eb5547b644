// EXPERIMENT OF ROUND 4, NOT PART OF THE LIBRARY: measured slower than csrc/conv_split.hip ("What comes next").
// Built and run by tools/experiments/wino1d/run.py; the sub-rectangle launch and its plan live on the git branch wino-experiment.
// conv_wino.hip -- the Q-net's 3x3 residual-tower layer (alpha_nnet.py:25-47) at float32 accuracy on the f16 matrix pipe with
// a third fewer MFMAs than csrc/conv_split.hip: the split-f16 scheme of that file (every float32 operand = hi + lo f16, a product
// = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, float32 accumulation) applied to the ONE-DIMENSIONAL Winograd form
// F(2, 3) along the image rows.
//
// Two horizontally adjacent outputs (y, 2q), (y, 2q + 1) form a PAIR.  With d0..d3 = the inputs at columns 2q - 1 .. 2q + 2 of a
// row and g0..g2 = the three taps of a kernel row:
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3               (input transform, float32, then scaled and split)
//     U0 = g0,  U1 = (g0 + g1 + g2) / 2,  U2 = (g0 - g1 + g2) / 2,  U3 = g2   (kernel transform, float64, once per weight set)
//     M_p[y][q][co] = sum over dy, ci of V_p[y + dy - 1][q][ci] * U_{dy,p}[ci][co]                       (four GEMMs, K = 3 x 128)
//     out(y, 2q) = M0 + M1 + M2,   out(y, 2q + 1) = M1 - M2 - M3
// 4 x 384 multiply-adds per pair and output instead of 2 x 1152: 1.5x fewer MFMAs.  (The two-dimensional F(2x2, 3x3) would save
// 2.25x but needs 16 accumulator positions per tile and streams 16 transformed kernels per block: DESIGN.md section 4 prices it
// out.  F(2, 3) in one dimension keeps 4 positions and 12 kernels.)
//
// GEMM rows are the image's pairs in row-major order, 32 per M tile.  One block = 8 wavefronts = up to 3 consecutive M tiles (96
// pairs = 192 pixels) of one image x all 128 outputs x all 4 positions; wave (wn, ph) owns outputs 32 wn .. 32 wn + 31 and
// positions 2 ph, 2 ph + 1 for all M tiles: 8 accumulator tiles = 128 registers, ONE block per CU (2 waves per SIMD).  A block
// streams every transformed kernel once (786 KB per 256 pixels, 21 B/clk/CU from L2 -- twice the direct kernel's rate; the bare
// loop in these proportions sustains 1 390 TFLOP/s against 1 465 for the direct proportions, tools/micro/wino1d_loop.hip).
// The transformed inputs sit in LDS as four planes [pair position][hi 16 | lo 16] (80-byte slots: conflict-free ds_read_b128),
// pair rows pitched by the pairs per row, one zero row above and below where the canvas ends: the three vertical taps of an
// A fragment are the SAME plane at three constant offsets, the four positions four planes.  Staging, per 16-channel chunk: a
// thread takes (pair position, 4 channels), loads the four pixels d0..d3, transforms, scales by the layer's power of two, clamps
// to the f16 range (range flag as in conv_split.hip), splits and writes 4 x (hi, lo).  B fragments (pre-split transformed kernels
// in fragment order) stream into a 3-deep register ring two steps ahead; A fragments are read two tiles ahead of their MFMAs.
// Epilogue, one M tile at a time: the 4 x 32 x 128 position sums go through LDS, a thread combines the four positions of a
// pair into its two pixels, applies batch norm (+ shortcut) + ReLU and writes two 16-byte pieces of two 512-byte pixel rows.
//
// The sub-rectangle form (see conv_split.hip) is the same body on a rectangle whose left edge is EVEN, so that its pairs are the
// full form's pairs: every computed pixel is then the same chain of operations on the same operands as in the full form, and the
// two forms stay bit-identical (tests/test_rect_conv_gpu.py).
#include "../../../alphasnake-zero_amd/csrc/common.h"
#include "conv_plan.h"         // HW_NPB, HW_NST, HW_NMAX, hw_rect_parts
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define HW_C 128
#define HW_KC 16
#define HW_LDP 80                                        // bytes per LDS pair slot: [hi k0-15 | lo k0-15] + 16
#define HW_PLANE ((HW_NPB + 1) * HW_LDP)                 // 14 160 bytes: the positions + one spare slot
#define HW_BUF (4 * HW_PLANE)                            // 56 640 bytes: the four positions of one chunk
#define HW_SMEM_STAGE (2 * HW_BUF)                       // 113 280 bytes
#define HW_MLD 132                                       // epilogue row (floats)
#define HW_SMEM_EPI (4 * 32 * HW_MLD * 4)                // 67 584 bytes: four positions of one M tile
#define HW_SMEM (HW_SMEM_STAGE > HW_SMEM_EPI ? HW_SMEM_STAGE : HW_SMEM_EPI)
#define HW_WS_ELEMS (12 * HW_C * HW_C * 2)               // f16 numbers in the weight image (786 432 bytes); the 32-byte tail follows

struct ConvWArgs {
    const float *x;            // [n][Hd][Wd][128]
    const f16x8 *wS;           // [chunk 8][ph 2][step 6 = dy * 2 + (p & 1)][wn 4][hi/lo][lane 64] x 8 f16
    const float *tail;         // {2^-k, 2^k, x scale s, 1 / s, int32 range flag, pad, 8-byte guard-word address}: conv_split.hip's tail
    const float *scale, *shift;
    const float *res;          // or NULL
    float *out;
    int Hd, Wd, n_blk, tiles_base, tiles_rem, relu;
    int n_img_grouped;
    const uint4 *desc;         // sub-rectangle form: the descriptors of k_rect_plan (pair mode)
    const int *n_desc;
    const float *bg_out, *bg_in, *bg_res;
    int grow_in, grow_res;
};

#ifdef HW_STAMPS       // development build only (tools/wino_stamps.py): s_memtime at the phase boundaries of every block
__device__ unsigned long long hw_stamp_buf[16384 * 8];
#define HW_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 16384) hw_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();
#define HW_STAMP_REAL(k) if (threadIdx.x == 0 && blockIdx.x < 16384) hw_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
extern "C" int snk_dbg_wino_stamps(unsigned long long *h_out, int n_blocks)
{
    SNK_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(hw_stamp_buf), (size_t)n_blocks * 8 * sizeof(unsigned long long)));
    return 0;
}
#else
#define HW_STAMP(k)
#define HW_STAMP_REAL(k)
#endif

// MODE: 0 = options from the arguments, 1 = ReLU, 2 = ReLU + residual (compile-time epilogues, as in conv_split.hip)
template <int NM, int MODE, bool RECT>
__device__ __forceinline__ void hw_block(const ConvWArgs &p, unsigned char *smem, const int img, const int tile0, const int ntile,
                                         const int ry0, const int rx0, const int rh, const int rw, const unsigned bbox,
                                         const int part, const int parts)
{
    const bool has_res = MODE == 0 ? p.res != nullptr : MODE == 2;
    const float relu_floor = (MODE != 0 || p.relu) ? 0.f : -__builtin_inff();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wn = wv & 3, ph = wv >> 2;
    const int h = lane >> 5, l31 = lane & 31;
    const int Wr = RECT ? rw : p.Wd, Hr = RECT ? rh : p.Hd;        // the GEMM's image in pixels
    const int Wp = (Wr + 1) >> 1;                                  // pairs per row
    const int cy0 = RECT ? ry0 : 0, cx0 = RECT ? rx0 : 0;          // its origin on the canvas (cx0 even)
    const int NP = Hr * Wp;                                        // GEMM rows = pairs
    const int HWc = p.Hd * p.Wd;
    const int m0 = 32 * tile0, m1 = min(32 * (tile0 + ntile), NP);
    const float invWp = 1.0f / (float)Wp;
    const int y_first = (int)(((float)m0 + 0.5f) * invWp), y_last = (int)(((float)(m1 - 1) + 0.5f) * invWp);
    // LDS row r of a plane holds pair row y_first - 1 + r of the image (zeros when that row is outside the canvas)
    const int ya = max(cy0 + y_first - 1, 0), yb = min(cy0 + y_last + 1, p.Hd - 1);
    const int ry_lo = ya - (cy0 + y_first - 1);
    const float xs = p.tail[2];
    const int by0 = bbox & 255, bx0 = (bbox >> 8) & 255, by1 = (bbox >> 16) & 255, bx1 = bbox >> 24;
    const bool sel_in = RECT && p.bg_in != nullptr;
    const float *ximg = p.x + (long)img * HWc * HW_C;

    // ---- staging role: ONE item per thread and chunk = two horizontally adjacent pairs ("double pair" tid / 4 of the strip, rows
    //      of (Wp + 1) / 2 double pairs) x float4 (tid % 4) of the chunk: six pixels, columns X0 - 1 .. X0 + 4 of canvas row Y,
    //      serve both pairs (a pair alone would need four).  A pixel outside the canvas counts as zero (its load is redirected to
    //      pixel X0 and multiplied by 0: what lies there is finite data); (RECT) a pixel the producing layer did not compute is
    //      read from that layer's background image.  Threads past the strip repeat its last item; the second pair of a row's
    //      last double pair, when the row has an odd number of pairs, goes to a spare slot behind the planes' positions.
    const int Wp2 = (Wp + 1) >> 1;
    const float invWp2 = 1.0f / (float)Wp2;
    unsigned ldo[2];                                       // LDS byte offsets (inside a plane) of the item's two pairs
    int poff;                                              // element offset of pixel X0 in the image (the thread's float4 adds 4 (tid % 4))
    unsigned pflg = 0;                                     // bits 0-5: pixel j is on the canvas; bits 8-13: it comes from the background image
    {
        const int pos = min(tid >> 2, (yb - ya + 1) * Wp2 - 1);
        const int r_ = (int)(((float)pos + 0.5f) * invWp2), q2_ = pos - r_ * Wp2;
        const int Y = ya + r_, X0 = cx0 + 4 * q2_;
        ldo[0] = (unsigned)((ry_lo + r_) * Wp + 2 * q2_) * HW_LDP + (tid & 3) * 8;
        ldo[1] = 2 * q2_ + 1 < Wp ? ldo[0] + HW_LDP : (unsigned)HW_NPB * HW_LDP + (tid & 3) * 8;
        poff = (Y * p.Wd + X0) * HW_C;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int X = X0 - 1 + j;
            if (X >= 0 && X < p.Wd) pflg |= 1u << j;
            if (sel_in && (Y < by0 - p.grow_in || Y > by1 + p.grow_in || X < bx0 - p.grow_in || X > bx1 + p.grow_in)) pflg |= 256u << j;
        }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 st[6][2];                                        // the six pixels' float4s as two packed halves
    float amax = 0.f;                                      // largest |V| (unscaled) this thread staged
// a pixel off the canvas is loaded from a clamped address inside the image (whatever lies there is multiplied by 0 in HW_PREP):
// no per-pixel offsets to keep in registers
#define HW_LOAD(c) _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) {                                                    \
        const int o_ = min(max(poff + (j_ - 1) * HW_C, 0), (HWc - 1) * HW_C) + 4 * (tid & 3) + HW_KC * (c);              \
        const float *b_ = (RECT && ((pflg >> (8 + j_)) & 1u)) ? p.bg_in : ximg;                                          \
        const float4 v_ = *(const float4 *)(b_ + o_);                                                                   \
        st[j_][0] = (f32x2){v_.x, v_.y}; st[j_][1] = (f32x2){v_.z, v_.w}; }
// unit 0 of the staging work: pixels outside the canvas become zero
#define HW_PREP() _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) {                                                     \
        const float m_ = ((pflg >> j_) & 1u) ? 1.0f : 0.0f;                                                             \
        st[j_][0] *= (f32x2){m_, m_}; st[j_][1] *= (f32x2){m_, m_}; }
// units 1-8: position P of pair e of the item: V_P (packed float32), hi = f16(V s), lo = f16(V s - hi) as single fused operations
// on the unscaled value (s is a power of two), two 8-byte LDS stores.  No clamp: a |V s| beyond the f16 range becomes an infinity,
// the launch's result is then garbage AND reported (amax -> the range flag), which every caller checks (QNet.forward_guarded).
#define HW_POS(e, P, bufoff)                                                                    \
    {                                                                                           \
        const f32x2 v0_ = (P) == 0 ? st[2 * (e)][0] - st[2 * (e) + 2][0] : (P) == 1 ? st[2 * (e) + 1][0] + st[2 * (e) + 2][0] : \
                          (P) == 2 ? st[2 * (e) + 2][0] - st[2 * (e) + 1][0] : st[2 * (e) + 1][0] - st[2 * (e) + 3][0];         \
        const f32x2 v1_ = (P) == 0 ? st[2 * (e)][1] - st[2 * (e) + 2][1] : (P) == 1 ? st[2 * (e) + 1][1] + st[2 * (e) + 2][1] : \
                          (P) == 2 ? st[2 * (e) + 2][1] - st[2 * (e) + 1][1] : st[2 * (e) + 1][1] - st[2 * (e) + 3][1];         \
        amax = fmaxf(fmaxf(amax, fabsf(v0_[0])), fabsf(v0_[1])); amax = fmaxf(fmaxf(amax, fabsf(v1_[0])), fabsf(v1_[1])); \
        f16x4 hi_, lo_;                                                                         \
        hi_[0] = (_Float16)__builtin_fmaf(v0_[0], xs, 0.0f); hi_[1] = (_Float16)__builtin_fmaf(v0_[1], xs, 0.0f); \
        hi_[2] = (_Float16)__builtin_fmaf(v1_[0], xs, 0.0f); hi_[3] = (_Float16)__builtin_fmaf(v1_[1], xs, 0.0f); \
        lo_[0] = (_Float16)__builtin_fmaf(v0_[0], xs, -(float)hi_[0]); lo_[1] = (_Float16)__builtin_fmaf(v0_[1], xs, -(float)hi_[1]); \
        lo_[2] = (_Float16)__builtin_fmaf(v1_[0], xs, -(float)hi_[2]); lo_[3] = (_Float16)__builtin_fmaf(v1_[1], xs, -(float)hi_[3]); \
        unsigned char *d_ = smem + (bufoff) + (P) * HW_PLANE + ldo[e];                           \
        *(f16x4 *)d_ = hi_;                                                                     \
        *(f16x4 *)(d_ + 32) = lo_;                                                              \
    }
#define HW_UNIT(u, bufoff)                                                                      \
    {                                                                                           \
        if ((u) == 0) { HW_PREP() }                                                             \
        else if ((u) == 1) { HW_POS(0, 0, bufoff) } else if ((u) == 2) { HW_POS(0, 1, bufoff) } \
        else if ((u) == 3) { HW_POS(0, 2, bufoff) } else if ((u) == 4) { HW_POS(0, 3, bufoff) } \
        else if ((u) == 5) { HW_POS(1, 0, bufoff) } else if ((u) == 6) { HW_POS(1, 1, bufoff) } \
        else if ((u) == 7) { HW_POS(1, 2, bufoff) } else { HW_POS(1, 3, bufoff) }               \
    }

    // this wave's B fragments: global step g = 6 chunk + s, s = 2 dy + (p & 1)
    const f16x8 *wl = p.wS + ((long)ph * 6 * 4 + wn) * 128 + lane;
    f16x8 Bq[3][2];
#define HW_LOADB(slot, g)                                                                       \
    {                                                                                           \
        const int g_ = (g);                                                                     \
        const f16x8 *w_ = wl + (long)((g_ / 6) * 12 + (g_ % 6)) * 512;                          \
        Bq[slot][0] = w_[0];                                                                    \
        Bq[slot][1] = w_[64];                                                                   \
    }
    HW_STAMP(0)
    HW_LOAD(0)
    HW_LOADB(0, 0) HW_LOADB(1, 1)
    for (int o = tid * 16; o < HW_SMEM_STAGE; o += 512 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // rows above / below the canvas stay zero
    f32x16 acc[2 * NM];
#pragma unroll
    for (int a = 0; a < 2 * NM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 9; ++u) HW_UNIT(u, 0)
    __syncthreads();

    unsigned la[NM];                                       // LDS byte address (plane 0, buffer 0, centre tap) of the lane's pair in M tile i
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int m_ = min(m0 + 32 * i + l31, NP - 1);     // rows past the image repeat its last pair (computed, never stored)
        const int y_ = (int)(((float)m_ + 0.5f) * invWp), q_ = m_ - y_ * Wp;
        la[i] = (unsigned)((y_ - (y_first - 1)) * Wp + q_) * HW_LDP + 16 * h;
    }
    const unsigned rowb = (unsigned)Wp * HW_LDP;           // one pair row further down
    const unsigned pl0 = (unsigned)(2 * ph) * HW_PLANE;    // this wave's first position
#define HW_LDS(off) (*(const f16x8 *)(smem + (off)))
#define HW_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// tile sequence of a chunk: t = s * NM + i, s = 2 dy + pp (vertical tap dy, position 2 ph + pp), i = M tile; A fragments are read
// one tile ahead of their MFMAs (fenced: the compiler otherwise sinks each read to its consumer; two ahead costs 8 registers
// this kernel does not have).  The two wave groups issue the next chunk's pixel loads at different steps (ph 0: step 0, ph 1:
// step 1, before the first staging unit of the smallest spread form): loads retire in order, so the B fragments a wave requests after its pixel loads wait for those (HBM latency) -- the
// two waves of a SIMD must not sit in that wait at the same time
#define HW_AOFF(t) (la[(t) % NM] + rb + pl0 + (unsigned)((((t) / NM) & 1) * HW_PLANE) + (unsigned)(((t) / NM) >> 1) * rowb - rowb)
#define HW_STEP(s, MORE)                                                                        \
        {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                    \
                const int t1_ = (s) * NM + i + 1;                                               \
                f16x8 nh = a0h, nl = a0l;                                                       \
                if (t1_ < 6 * NM) { nh = HW_LDS(HW_AOFF(t1_)); nl = HW_LDS(HW_AOFF(t1_) + 32); } \
                __builtin_amdgcn_sched_barrier(0);                                              \
                if (i == 0) {                                                                   \
                    if (MORE || (s) + 2 < 6) { HW_LOADB(((s) + 2) % 3, c * 6 + (s) + 2); }      \
                    if ((s) == (ph ? 1 : 0) && MORE) { HW_LOAD(c + 1) }                         \
                }                                                                               \
                {                                                                               \
                    const int tl_ = (s) * NM + i - (6 * NM - 9);                                \
                    if (MORE && 6 * NM >= 9 && tl_ >= 0) { HW_UNIT(tl_ < 0 ? 0 : tl_, wb) }     \
                }                                                                               \
                HW_MFMA(a0h, Bq[(s) % 3][0], acc[((s) & 1) * NM + i]);                          \
                HW_MFMA(a0h, Bq[(s) % 3][1], acc[((s) & 1) * NM + i]);                          \
                HW_MFMA(a0l, Bq[(s) % 3][0], acc[((s) & 1) * NM + i]);                          \
                a0h = nh; a0l = nl;                                                             \
                __builtin_amdgcn_sched_barrier(0);                                              \
            }                                                                                   \
        }
#define HW_CHUNK(MORE)                                                                          \
    {                                                                                           \
        const unsigned rb = (unsigned)(c & 1) * HW_BUF;                                         \
        const unsigned wb = (unsigned)((c & 1) ^ 1) * HW_BUF;                                   \
        f16x8 a0h = HW_LDS(HW_AOFF(0)), a0l = HW_LDS(HW_AOFF(0) + 32);                          \
        HW_STEP(0, MORE) HW_STEP(1, MORE) HW_STEP(2, MORE) HW_STEP(3, MORE) HW_STEP(4, MORE) HW_STEP(5, MORE) \
        if (MORE && 6 * NM < 9) {                       /* too few regions to spread the staging units over: do them here */ \
            _Pragma("unroll") for (int u = 0; u < 9; ++u) HW_UNIT(u, wb)                        \
        }                                                                                       \
        __syncthreads();                                                                        \
    }
    int c = 0;
    HW_STAMP(1)
    HW_STAMP_REAL(5)
#pragma unroll 1
    for (; c < HW_C / HW_KC - 1; ++c) HW_CHUNK(true)
    HW_STAMP(2)
    HW_CHUNK(false)
    HW_STAMP(3)
    HW_STAMP_REAL(6)
    if (amax * xs >= 65504.f) {            // range guard, as in conv_split.hip: the layer's flag + the net's host-mapped guard word
        int *t = (int *)const_cast<float *>(p.tail);
        atomicOr(t + 4, 1);
        int *shared = *(int *const *)(t + 6);
        if (shared) __hip_atomic_store(shared, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef HW_STEP
#undef HW_CHUNK
#undef HW_AOFF
#undef HW_LOAD
#undef HW_UNIT
#undef HW_POS
#undef HW_PREP
#undef HW_LOADB
#undef HW_LDS
#undef HW_MFMA

    // ---- epilogue, one M tile (32 pairs) per pass: the four positions' sums meet in LDS [position][pair][output]; thread
    //      (rr = tid / 32, cq = tid % 32) takes pairs rr and rr + 16 of the tile, outputs 4 cq .. 4 cq + 3, both pixels
    const int cq = tid & 31, rr = tid >> 5;
    const float winv = p.tail[0] * p.tail[3];
    float4 sc4 = *(const float4 *)(p.scale + 4 * cq), sh4 = *(const float4 *)(p.shift + 4 * cq);
    sc4.x *= winv; sc4.y *= winv; sc4.z *= winv; sc4.w *= winv;
    float *Ms = (float *)smem;                                      // [4][32][HW_MLD]
    const long obase = (long)img * HWc * HW_C + 4 * cq;
    const bool sel_res = RECT && p.bg_res != nullptr;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        int off[2][2];                                              // element offsets of the two pixels of this thread's two pairs (-1: not stored)
        float4 rv[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m_ = m0 + 32 * i + rr + 16 * j;
            const int y_ = (int)(((float)m_ + 0.5f) * invWp), q_ = m_ - y_ * Wp;
            const int Y = cy0 + y_, X0 = cx0 + 2 * q_;
            off[j][0] = m_ < m1 ? (Y * p.Wd + X0) * HW_C : -1;
            off[j][1] = (m_ < m1 && 2 * q_ + 1 < Wr) ? (Y * p.Wd + X0 + 1) * HW_C : -1;
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int X = X0 + e;
                    const bool stale = sel_res && (Y < by0 - p.grow_res || Y > by1 + p.grow_res || X < bx0 - p.grow_res || X > bx1 + p.grow_res);
                    const float *rb_ = (RECT && stale) ? p.bg_res + 4 * cq : p.res + obase;
                    rv[j][e] = *(const float4 *)(rb_ + max(off[j][e], 0));
                }
            }
        }
        if (i > 0) __syncthreads();                                 // the previous pass's readers are done with Ms
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ms[((2 * ph + pp) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * HW_MLD + 32 * wn + l31] = acc[pp * NM + i][r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = rr + 16 * j;
            const float4 q0 = *(const float4 *)&Ms[(0 * 32 + row) * HW_MLD + 4 * cq], q1 = *(const float4 *)&Ms[(1 * 32 + row) * HW_MLD + 4 * cq];
            const float4 q2 = *(const float4 *)&Ms[(2 * 32 + row) * HW_MLD + 4 * cq], q3 = *(const float4 *)&Ms[(3 * 32 + row) * HW_MLD + 4 * cq];
            float4 v[2];
            v[0] = make_float4((q0.x + q1.x) + q2.x, (q0.y + q1.y) + q2.y, (q0.z + q1.z) + q2.z, (q0.w + q1.w) + q2.w);
            v[1] = make_float4((q1.x - q2.x) - q3.x, (q1.y - q2.y) - q3.y, (q1.z - q2.z) - q3.z, (q1.w - q2.w) - q3.w);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float4 u = v[e];
                u.x = __builtin_fmaf(u.x, sc4.x, sh4.x); u.y = __builtin_fmaf(u.y, sc4.y, sh4.y);
                u.z = __builtin_fmaf(u.z, sc4.z, sh4.z); u.w = __builtin_fmaf(u.w, sc4.w, sh4.w);
                if (has_res) { const float4 r_ = rv[j][e]; u.x += r_.x; u.y += r_.y; u.z += r_.z; u.w += r_.w; }
                u.x = fmaxf(u.x, relu_floor); u.y = fmaxf(u.y, relu_floor); u.z = fmaxf(u.z, relu_floor); u.w = fmaxf(u.w, relu_floor);
                if (off[j][e] >= 0) *(float4 *)(p.out + obase + off[j][e]) = u;
            }
        }
    }
    if (RECT && p.bg_out) {
        // the layers that read this output are full layers: every pixel of the canvas outside the rectangle takes the layer's
        // state-independent background value; the image's parts share the pixels, 32 lanes per pixel
        const float invF = 1.0f / (float)p.Wd;
        for (int q = part * 16 + rr; q < HWc; q += 16 * parts) {
            const int y_ = (int)(((float)q + 0.5f) * invF), x_ = q - y_ * p.Wd;
            if (y_ >= cy0 && y_ < cy0 + rh && x_ >= cx0 && x_ < cx0 + rw) continue;
            const int o_ = (y_ * p.Wd + x_) * HW_C + 4 * cq;
            *(float4 *)(p.out + (long)img * HWc * HW_C + o_) = *(const float4 *)(p.bg_out + o_);
        }
    }
    HW_STAMP(4)
}

template <int NM, int MODE>
__global__ __launch_bounds__(512, 2) void k_conv3x3_f16sw(ConvWArgs p)
{
    extern __shared__ __align__(16) unsigned char hw_smem[];
    int img, blk;
    {       // XCD-aware block -> (image, part) map, as in conv_split.hip: the parts of an image run on the same XCD back to back
        const int b = blockIdx.x, per = 8 * p.n_blk;
        if (b < p.n_img_grouped * p.n_blk) { const int r = b % per; img = (b / per) * 8 + (r & 7); blk = r >> 3; }
        else { img = b / p.n_blk; blk = b - img * p.n_blk; }
    }
    const int tile0 = blk * p.tiles_base + min(blk, p.tiles_rem), ntile = p.tiles_base + (blk < p.tiles_rem ? 1 : 0);
    hw_block<NM, MODE, false>(p, hw_smem, img, tile0, ntile, 0, 0, 0, 0, 0u, 0, 1);
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_conv3x3_f16sw_rect(ConvWArgs p)
{
    extern __shared__ __align__(16) unsigned char hw_smem[];
    const int nd = *p.n_desc;
    const uint4 d = p.desc[blockIdx.x];
    if ((int)blockIdx.x >= nd) return;
    const int img = (int)d.x, ry0 = d.y & 255, rx0 = (d.y >> 8) & 255, rh = (d.y >> 16) & 255, rw = d.y >> 24;
    const int tile0 = d.z & 255, ntile = (d.z >> 8) & 255, part = (d.z >> 16) & 255, parts = d.z >> 24;
    switch (ntile) {
    case 1: hw_block<1, MODE, true>(p, hw_smem, img, tile0, ntile, ry0, rx0, rh, rw, d.w, part, parts); break;
    case 2: hw_block<2, MODE, true>(p, hw_smem, img, tile0, ntile, ry0, rx0, rh, rw, d.w, part, parts); break;
    default: hw_block<3, MODE, true>(p, hw_smem, img, tile0, ntile, ry0, rx0, rh, rw, d.w, part, parts); break;
    }
}

// ---- transformed, split weights ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double hw_u(const float *__restrict__ w, int dy, int p, int cin, int cout)
{
    const double g0 = w[(long)((dy * 3 + 0) * HW_C + cin) * HW_C + cout], g1 = w[(long)((dy * 3 + 1) * HW_C + cin) * HW_C + cout];
    const double g2 = w[(long)((dy * 3 + 2) * HW_C + cin) * HW_C + cout];
    return p == 0 ? g0 : p == 1 ? 0.5 * (g0 + g1 + g2) : p == 2 ? 0.5 * (g0 - g1 + g2) : g2;
}

// max |U| of the layer -> k with 256 <= max * 2^k < 512; the tail of conv_split.hip's weight image
__global__ __launch_bounds__(1024) void k_f16sw_wscale(const float *__restrict__ w, float *__restrict__ tail, float x_scale)
{
    __shared__ float red[1024];
    float m = 0.f;
    for (int i = threadIdx.x; i < 3 * HW_C * HW_C; i += 1024) {
        const int dy = i / (HW_C * HW_C), r = i - dy * HW_C * HW_C, cin = r / HW_C, cout = r - cin * HW_C;
#pragma unroll
        for (int p = 0; p < 4; ++p) m = fmaxf(m, (float)fabs(hw_u(w, dy, p, cin, cout)));
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
        tail[2] = x_scale;
        tail[3] = 1.0f / x_scale;
        tail[4] = tail[5] = tail[6] = tail[7] = 0.f;
    }
}

__global__ void k_f16sw_weights(const float *__restrict__ w, _Float16 *__restrict__ wS, const float *__restrict__ tail)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;          // one f16x8 fragment piece
    if (v >= HW_WS_ELEMS / 8) return;
    const double mul = tail[1];
    const int lane = v & 63, hl = (v >> 6) & 1, wn = (v >> 7) & 3, g = v >> 9;     // g = (chunk * 2 + ph) * 6 + s
    const int s = g % 6, ph = (g / 6) & 1, c = g / 12;
    const int dy = s >> 1, p = 2 * ph + (s & 1);
    const int h = lane >> 5, l31 = lane & 31, cout = 32 * wn + l31;
    for (int j = 0; j < 8; ++j) {
        const int cin = HW_KC * c + 8 * h + j;
        const float val = (float)(hw_u(w, dy, p, cin, cout) * mul);
        const _Float16 hi = (_Float16)val;
        wS[(long)v * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
    }
}

extern "C" int snk_conv3x3_prepare_weights_f16sw(const float *d_w_hwio, void *d_wS, float x_scale, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS, "snk_conv3x3_prepare_weights_f16sw: NULL argument");
    int e_ = 0;
    SNK_REQUIRE(x_scale > 0.f && frexpf(x_scale, &e_) == 0.5f, "snk_conv3x3_prepare_weights_f16sw: x_scale %g is not a power of two", x_scale);
    float *tail = (float *)((_Float16 *)d_wS + HW_WS_ELEMS);
    k_f16sw_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail, x_scale);
    k_f16sw_weights<<<(HW_WS_ELEMS / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- launches ------------------------------------------------------------------------------------------------------------------------
template <typename K>
static int hw_smem_attr(K kernel)
{
    SNK_CHECK_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HW_SMEM));
    return 0;
}

static int hw_prepare_kernels()
{
    static int done = 0;
    if (done) return 0;
#define HW_ATTR(NM_) { int rc_; if ((rc_ = hw_smem_attr(k_conv3x3_f16sw<NM_, 0>))) return rc_; if ((rc_ = hw_smem_attr(k_conv3x3_f16sw<NM_, 1>))) return rc_; \
                       if ((rc_ = hw_smem_attr(k_conv3x3_f16sw<NM_, 2>))) return rc_; }
    HW_ATTR(1) HW_ATTR(2) HW_ATTR(3)
#undef HW_ATTR
    done = 1;
    return 0;
}

extern "C" int snk_conv3x3_bn_f16sw(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                    const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                                    void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_f16sw: NULL argument");
    SNK_REQUIRE(n_images >= 0 && height >= 1 && width >= 3, "snk_conv3x3_bn_f16sw: bad shape %d x %d x %d", n_images, height, width);
    SNK_REQUIRE(d_out != d_x, "snk_conv3x3_bn_f16sw: in-place convolution is not possible (blocks read their neighbours' input rows)");
    if (n_images == 0) return 0;
    int tiles_max = 0;
    const int n_blk = hw_rect_parts(height, width, &tiles_max);
    SNK_REQUIRE(n_blk > 0, "snk_conv3x3_bn_f16sw: observation width %d not supported", width);
    const int wp = (width + 1) / 2, T = (height * wp + 31) / 32;
    const int tiles_base = T / n_blk, tiles_rem = T % n_blk;
    SNK_REQUIRE((long)n_images * n_blk < (1l << 31) && (long)height * width * HW_C < (1l << 31), "snk_conv3x3_bn_f16sw: batch too large");
    int rc = hw_prepare_kernels();
    if (rc) return rc;
    ConvWArgs a = {d_x, (const f16x8 *)d_wS, (const float *)((const _Float16 *)d_wS + HW_WS_ELEMS), d_scale, d_shift, d_residual, d_out,
                   height, width, n_blk, tiles_base, tiles_rem, relu, (n_images / 8) * 8, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    const int grid = n_images * n_blk, mode = !relu ? 0 : (d_residual ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
#define HW_LAUNCH(NM_)                                                                          \
    case NM_:                                                                                   \
        if (mode == 1) k_conv3x3_f16sw<NM_, 1><<<grid, 512, HW_SMEM, st>>>(a);                  \
        else if (mode == 2) k_conv3x3_f16sw<NM_, 2><<<grid, 512, HW_SMEM, st>>>(a);             \
        else k_conv3x3_f16sw<NM_, 0><<<grid, 512, HW_SMEM, st>>>(a);                            \
        break;
    switch (tiles_max) {
        HW_LAUNCH(1) HW_LAUNCH(2)
    default:
        if (mode == 1) k_conv3x3_f16sw<3, 1><<<grid, 512, HW_SMEM, st>>>(a);
        else if (mode == 2) k_conv3x3_f16sw<3, 2><<<grid, 512, HW_SMEM, st>>>(a);
        else k_conv3x3_f16sw<3, 0><<<grid, 512, HW_SMEM, st>>>(a);
        break;
    }
#undef HW_LAUNCH
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

