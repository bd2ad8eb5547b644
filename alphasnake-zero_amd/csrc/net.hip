// net.hip -- the Q-net's inference path on gfx950 (reference: AlphaNNet.__init__ / .v,
// alpha_nnet.py:19-56, 61-76).  Activations are NHWC float32 (the reference is channels-last),
// batch-norm is folded into a per-channel scale/shift, and every layer is a hand-written kernel:
//
//   k_stem_conv      3x3 conv 3 -> C, + BN + ReLU            (alpha_nnet.py:21-22)      VALU
//   k_conv3x3_f32    3x3 conv C -> C as an implicit GEMM on v_mfma_f32_32x32x2_f32,
//                    + BN (+ residual) + ReLU                (alpha_nnet.py:25-47)      MFMA
//   k_head           1x1 conv C -> 1 + BN + ReLU, Flatten, Dense(128) + ReLU,
//                    Dense(3) + tanh, obstacle mask          (alpha_nnet.py:49-54, 63-73)
//
// Implicit GEMM of the 3x3 layer: M = pixels (batch * HW), N = C_out = 128, K = 9 taps * 128.
// Block tile 128 (pixels) x 128 (C_out), 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator VGPRs).  K advances in chunks of 32 input channels of one tap; both
// operand tiles sit in LDS K-contiguous ([row][k], rows padded to 36 floats so that 16-lane
// ds_read_b128 groups are bank-conflict free), so one ds_read_b128 per lane delivers the lane's
// operand for 4 consecutive MFMAs: lanes 0-31 take k = 8c+q, lanes 32-63 k = 8c+4+q, q = 0..3 --
// the f32 MFMA's two k-slots only have to agree between A and B.  Global -> LDS staging is
// register double-buffered: the loads of chunk c+1 are issued before the 64 MFMAs of chunk c.
#include "common.h"
#include "train_fold.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CV_BM 128
#define CV_BN 128
#define CV_C 128

struct ConvArgs {
    const float *x;      // [M][128] NHWC activations
    const float *wT;     // [9][128 cout][128 cin]
    const float *scale;  // [128]
    const float *shift;  // [128]
    const float *res;    // [M][128] or NULL
    float *out;          // [M][128]
    int M;               // batch * HW
    int Hd, Wd;          // image height / width (21 x 21 or 37 x 37)
    int relu;
};

// BK = input channels per K chunk.  BK 16: 40 KB of LDS per block -> 3 blocks (3 waves per SIMD) per CU, so the
// matrix pipe always finds a wave that is not parked at the chunk barrier; BK 32: 72 KB -> 2 blocks per CU.
template <int BK, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_conv3x3_f32(ConvArgs p)
{
    constexpr int LD = BK + 4;                 // padded row: 16-lane ds_read_b128 groups hit distinct banks
    constexpr int F4 = BK / 4;                 // float4 per tile row
    constexpr int NLD = CV_BM * F4 / 256;      // float4 loads per thread per operand per chunk (2 or 4)
    constexpr int RSTEP = 256 / F4;            // tile rows covered by one pass of the 256 threads
    constexpr int NCC = CV_C / BK;             // channel chunks
    __shared__ __align__(16) float As[2][CV_BM][LD];
    __shared__ __align__(16) float Bs[2][CV_BN][LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv & 1, wn = wv >> 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * CV_BM;
    const int HW = p.Hd * p.Wd;

    // staging assignment: float4 number f = tid + 256 i -> tile row tid / F4 + RSTEP i, 16-byte column tid % F4
    const int c4 = tid % F4, srow = tid / F4;
    int ay[NLD], ax[NLD];
    const float *abase[NLD];
    bool avalid[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int m = m0 + srow + RSTEP * i;
        avalid[i] = m < p.M;
        const int mm = avalid[i] ? m : 0;
        const int rem = mm % HW;
        ay[i] = rem / p.Wd;
        ax[i] = rem - ay[i] * p.Wd;
        abase[i] = p.x + (long)mm * CV_C + c4 * 4;       // always a readable address (the pixel itself)
    }
    const float *bbase = p.wT + (long)srow * CV_C + c4 * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    bool ok0 = false, ok1 = false, ok2 = false, ok3 = false;
    ra2 = ra3 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);

    // issue the global loads of chunk c (tap c % 9 of input channels BK (c / 9) ...; the 9 taps of one channel
    // chunk re-read the same quarter rows, which stay in L1/L2): branch-free -- a tap that falls outside the
    // image reads the pixel itself and is zeroed when the registers go to LDS
#define CV_LOAD1(i, RA, RB, OK)                                                                 \
    if (i < NLD) {                                                                              \
        const int yy_ = ay[i < NLD ? i : 0] + dy_, xx_ = ax[i < NLD ? i : 0] + dx_;             \
        OK = avalid[i < NLD ? i : 0] && yy_ >= 0 && yy_ < p.Hd && xx_ >= 0 && xx_ < p.Wd;       \
        RA = *(const float4 *)(abase[i < NLD ? i : 0] + (OK ? aoff_ : 0l) + cin0_);             \
        RB = *(const float4 *)(bbase + ((long)(tap_ * CV_C + RSTEP * i) * CV_C + cin0_));       \
    }
#define CV_LOAD(c)                                                                              \
    {                                                                                           \
        const int tap_ = (c) % 9, cin0_ = ((c) / 9) * BK;                                       \
        const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                          \
        const long aoff_ = (long)(dy_ * p.Wd + dx_) * CV_C;                                     \
        CV_LOAD1(0, ra0, rb0, ok0) CV_LOAD1(1, ra1, rb1, ok1)                                   \
        CV_LOAD1(2, ra2, rb2, ok2) CV_LOAD1(3, ra3, rb3, ok3)                                   \
    }
#define CV_STORE1(buf, i, RA, RB, OK)                                                           \
    if (i < NLD) {                                                                              \
        float4 v_ = RA;                                                                         \
        if (!OK) v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                          \
        *(float4 *)&As[buf][srow + RSTEP * i][c4 * 4] = v_;                                     \
        *(float4 *)&Bs[buf][srow + RSTEP * i][c4 * 4] = RB;                                     \
    }
#define CV_STORE(buf)                                                                           \
    {                                                                                           \
        CV_STORE1(buf, 0, ra0, rb0, ok0) CV_STORE1(buf, 1, ra1, rb1, ok1)                       \
        CV_STORE1(buf, 2, ra2, rb2, ok2) CV_STORE1(buf, 3, ra3, rb3, ok3)                       \
    }

    CV_LOAD(0);
    CV_STORE(0);
    __syncthreads();

    constexpr int NCHUNK = 9 * NCC;
    for (int c = 0; c < NCHUNK; ++c) {
        const int buf = c & 1;
        if (c + 1 < NCHUNK) CV_LOAD(c + 1);
        __builtin_amdgcn_sched_barrier(0);               // keep the loads ahead of the MFMAs that hide them
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            float4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *(const float4 *)&As[buf][wm * 64 + t * 32 + l31][k8 * 8 + 4 * h];
                b[t] = *(const float4 *)&Bs[buf][wn * 64 + t * 32 + l31][k8 * 8 + 4 * h];
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt].x, b[ct].x, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt].y, b[ct].y, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt].z, b[ct].z, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt].w, b[ct].w, acc[rt][ct], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCHUNK) CV_STORE(buf ^ 1);
        __syncthreads();
    }
#undef CV_LOAD
#undef CV_STORE
#undef CV_LOAD1
#undef CV_STORE1

    // epilogue: BN scale/shift (+ residual) + ReLU.  C/D map of the 32x32 MFMA:
    // col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = wn * 64 + ct * 32 + l31;
        const float sc = p.scale[col], sh = p.shift[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + row;
                if (m < p.M) {
                    float v = acc[rt][ct][r] * sc + sh;
                    if (p.res) v += p.res[(long)m * CV_C + col];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.out[(long)m * CV_C + col] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) form of the same layer (fp32 throughout): Y = A^T [ (G g G^T) . (B^T d B) ] A.
// 16 transformed positions p = (xi, nu), each a GEMM [tiles x 128 cin] x [128 cin x 128 cout] on
// v_mfma_f32_32x32x2_f32: 16 * 121 * 128 * 128 MACs per 21x21 image instead of 441 * 9 * 128 * 128
// (2.05x fewer).  One block = 32 consecutive 2x2-output tiles x all 16 positions x all 128 outputs,
// 8 wavefronts, wave w owns positions 2w and 2w+1 (2 x 4 MFMA tiles = 128 accumulator VGPRs).
//   per 16-channel chunk: every thread transforms one (tile, channel) 4x4 patch (B^T d B: 32 adds) and
//   scatters the 16 results to LDS V[p][tile][channel]; A fragments come from LDS, B fragments (U, the
//   pre-transformed weights, 1 MB, L2 resident) stream straight into registers one k8-step ahead;
//   epilogue: accumulators -> LDS M[p][tile][cout] one 32-output slice at a time, A^T M A per (tile, cout),
//   BN scale/shift (+ residual) + ReLU, 2x2 pixels written.
// ------------------------------------------------------------------------------------------
#define WG_TB 32          // tiles per block
#define WG_KC 16          // input channels per chunk
#define WG_LD 20          // padded V row (floats)

struct WinoArgs {
    const float *x;      // [n][Hd][Wd][128]
    const float *U;      // [16 p][32 cin/4][128 cout][4]
    const float *scale, *shift;
    const float *res;    // or NULL
    float *out;
    int n_tiles;         // n * TY * TX
    int Hd, Wd, TY, TX;
    int relu;
};

// NW wavefronts per block, each owning PPW = 16 / NW positions:
//   NW = 8: 512 threads, 2 positions per wave (128 accumulator VGPRs), 1 block per CU (2 waves per SIMD share its barriers);
//   NW = 4: 256 threads, 4 positions per wave (256 accumulator VGPRs), 80 KB of LDS -> 2 independent blocks per CU, one
//           wave per SIMD each, so one block's transform / barrier / epilogue hides behind the other block's MFMAs.
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void k_conv3x3_wino_f32(WinoArgs p)
{
    constexpr int PPW = 16 / NW;                   // positions per wave
    constexpr int NT = 64 * NW;                    // threads
    constexpr int ITEMS = (WG_TB * WG_KC) / NT;    // (tile, channel) patches each thread transforms per chunk
    constexpr int EPC = (NW == 8) ? 64 : 32;       // outputs per epilogue pass
    constexpr int MLD = EPC + 4;                   // M row (floats), 16-byte aligned
    constexpr int SMEM_V = 2 * 16 * WG_TB * WG_LD, SMEM_M = 16 * WG_TB * MLD;
    __shared__ __align__(16) float smem[SMEM_V > SMEM_M ? SMEM_V : SMEM_M];    // V double buffer, reused as M
    float(*Vs)[16][WG_TB][WG_LD] = (float(*)[16][WG_TB][WG_LD])smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int t0 = blockIdx.x * WG_TB;
    const int tiles_per_img = p.TY * p.TX;

    // ---- input-transform role: item it of this thread = (tile tl[it], channel cl of the chunk)
    const int cl = tid & 15;
    unsigned vmask[ITEMS];                       // bit (4 i + j): patch pixel (i, j) lies inside the image
    unsigned rowoff[ITEMS][4], coloff[ITEMS][4]; // element offsets of the (clamped, always readable) patch rows / columns
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int tl = (tid + it * NT) >> 4;
        const int t = t0 + tl;
        vmask[it] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) rowoff[it][i] = coloff[it][i] = 0;
        if (t < p.n_tiles) {
            const int img = t / tiles_per_img, r = t - img * tiles_per_img;
            const int ty = r / p.TX, tx = r - ty * p.TX;
            const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yc = min(max(y0 + i, 0), p.Hd - 1), xc = min(max(x0 + i, 0), p.Wd - 1);
                rowoff[it][i] = (unsigned)((img * p.Hd + yc) * p.Wd) * CV_C;
                coloff[it][i] = (unsigned)xc * CV_C;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (y0 + i >= 0 && y0 + i < p.Hd && x0 + j >= 0 && x0 + j < p.Wd) vmask[it] |= 1u << (4 * i + j);
            }
        }
    }
    float raw[ITEMS][16];
    const float *xcl = p.x + cl;
#define WG_LOAD_RAW(c)                                                                          \
    _Pragma("unroll") for (int it = 0; it < ITEMS; ++it)                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                raw[it][4 * i + j] = xcl[(size_t)(rowoff[it][i] + coloff[it][j]) + (c) * WG_KC];
#define WG_TRANSFORM_STORE(buf)                                                                 \
    _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                      \
        /* B^T d B row by row: output row i needs only r_i[j] = (B^T d)[i][j] */               \
        const int tl_ = (tid + it * NT) >> 4;                                                   \
        float *w_ = raw[it];                                                                    \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) w_[q] = ((vmask[it] >> q) & 1u) ? w_[q] : 0.f; \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
            float r0_, r1_, r2_, r3_;                                                           \
            if (i == 0)      { r0_ = w_[0] - w_[8];  r1_ = w_[1] - w_[9];  r2_ = w_[2] - w_[10];  r3_ = w_[3] - w_[11]; }  \
            else if (i == 1) { r0_ = w_[4] + w_[8];  r1_ = w_[5] + w_[9];  r2_ = w_[6] + w_[10];  r3_ = w_[7] + w_[11]; }  \
            else if (i == 2) { r0_ = w_[8] - w_[4];  r1_ = w_[9] - w_[5];  r2_ = w_[10] - w_[6];  r3_ = w_[11] - w_[7]; }  \
            else             { r0_ = w_[4] - w_[12]; r1_ = w_[5] - w_[13]; r2_ = w_[6] - w_[14];  r3_ = w_[7] - w_[15]; } \
            Vs[buf][4 * i + 0][tl_][cl] = r0_ - r2_;                                            \
            Vs[buf][4 * i + 1][tl_][cl] = r1_ + r2_;                                            \
            Vs[buf][4 * i + 2][tl_][cl] = r2_ - r1_;                                            \
            Vs[buf][4 * i + 3][tl_][cl] = r1_ - r3_;                                            \
        }                                                                                       \
    }

    // ---- GEMM role: wave wv owns positions PPW wv .. PPW wv + PPW - 1
    f32x16 acc[PPW][4];
#pragma unroll
    for (int a = 0; a < PPW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // U fragment of (position pi, n-tile nt) for k8-step (c, s): float4 at U[p][(16 c + 8 s + 4 h) / 4][32 nt + l31][0..3]
    const float4 *Ub = (const float4 *)p.U + ((long)(PPW * wv) * 32 + h) * 128 + l31;
#define WG_LOAD_B(dst, c, s)                                                                    \
    _Pragma("unroll") for (int pi = 0; pi < PPW; ++pi)                                          \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                        \
            dst[pi][nt] = Ub[((long)pi * 32 + (c) * 4 + (s) * 2) * 128 + nt * 32];
#define WG_MFMA(buf, s, B)                                                                      \
    {                                                                                           \
        float4 a_[PPW];                                                                         \
        _Pragma("unroll") for (int pi = 0; pi < PPW; ++pi)                                      \
            a_[pi] = *(const float4 *)&Vs[buf][PPW * wv + pi][l31][(s) * 8 + 4 * h];            \
        _Pragma("unroll") for (int pi = 0; pi < PPW; ++pi)                                      \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                  \
                acc[pi][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[pi].x, B[pi][nt].x, acc[pi][nt], 0, 0, 0); \
                acc[pi][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[pi].y, B[pi][nt].y, acc[pi][nt], 0, 0, 0); \
                acc[pi][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[pi].z, B[pi][nt].z, acc[pi][nt], 0, 0, 0); \
                acc[pi][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[pi].w, B[pi][nt].w, acc[pi][nt], 0, 0, 0); \
            }                                                                                   \
    }

    constexpr int NCH = CV_C / WG_KC;
    float4 B0[PPW][4], B1[PPW][4];
    WG_LOAD_RAW(0);
    WG_LOAD_B(B0, 0, 0);
    WG_TRANSFORM_STORE(0);
    __syncthreads();

    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        WG_LOAD_B(B1, c, 1);                              // only the 8 B loads right after the barrier (all waves issue here at once)
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(buf, 0, B0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) { WG_LOAD_B(B0, c + 1, 0); WG_LOAD_RAW(c + 1); }   // B before raw: vmcnt retires in order
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(buf, 1, B1);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) { WG_TRANSFORM_STORE(buf ^ 1); }
        __syncthreads();
    }
#undef WG_LOAD_RAW
#undef WG_TRANSFORM_STORE
#undef WG_LOAD_B
#undef WG_MFMA

    // ---- epilogue: M -> LDS one EPC-output slice at a time, inverse transform A^T M A on 4 outputs per thread,
    //      BN (+ residual) + ReLU, 2x2 pixels x float4 written
    float(*Ms)[WG_TB][MLD] = (float(*)[WG_TB][MLD])smem;          // [16 p][32 tiles][EPC + 4]
    constexpr int QPT = EPC / 4;                                    // output quads per tile and pass
    const int tq = tid / QPT, quad = tid % QPT;                     // output role: tile tq, outputs 4 quad .. 4 quad + 3 of the slice
    const int t = t0 + tq;
    const bool tvalid = t < p.n_tiles;
    int oy = 0, ox = 0;
    long obase = 0;
    if (tvalid) {
        const int img = t / tiles_per_img, r = t - img * tiles_per_img;
        const int ty = r / p.TX, tx = r - ty * p.TX;
        oy = 2 * ty; ox = 2 * tx;
        obase = ((long)(img * p.Hd + oy) * p.Wd + ox) * CV_C;
    }
#pragma unroll
    for (int pass = 0; pass < CV_C / EPC; ++pass) {
        const int co = pass * EPC + quad * 4;
        float4 rres[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool ok = tvalid && p.res && oy + a < p.Hd && ox + b < p.Wd;
                rres[a][b] = ok ? *(const float4 *)(p.res + obase + (long)(a * p.Wd + b) * CV_C + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
            for (int nn = 0; nn < EPC / 32; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ms[PPW * wv + pi][(r & 3) + 8 * (r >> 2) + 4 * h][nn * 32 + l31] = acc[pi][pass * (EPC / 32) + nn][r];
        __syncthreads();
        if (tvalid) {
            const float4 sc = *(const float4 *)(p.scale + co), sh = *(const float4 *)(p.shift + co);
            float4 m[16];
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) m[pp] = *(const float4 *)&Ms[pp][tq][quad * 4];
            float4 y[2][2];
#define WG_INV(F)                                                                               \
            {                                                                                   \
                float s0[4], s1[4];                                                             \
                _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) {                              \
                    s0[nu] = (m[0 + nu].F + m[4 + nu].F) + m[8 + nu].F;                         \
                    s1[nu] = (m[4 + nu].F - m[8 + nu].F) - m[12 + nu].F;                        \
                }                                                                               \
                y[0][0].F = (s0[0] + s0[1]) + s0[2]; y[0][1].F = (s0[1] - s0[2]) - s0[3];       \
                y[1][0].F = (s1[0] + s1[1]) + s1[2]; y[1][1].F = (s1[1] - s1[2]) - s1[3];       \
            }
            WG_INV(x) WG_INV(y) WG_INV(z) WG_INV(w)
#undef WG_INV
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (oy + a < p.Hd && ox + b < p.Wd) {
                        float4 v;
                        v.x = y[a][b].x * sc.x + sh.x + rres[a][b].x;
                        v.y = y[a][b].y * sc.y + sh.y + rres[a][b].y;
                        v.z = y[a][b].z * sc.z + sh.z + rres[a][b].z;
                        v.w = y[a][b].w * sc.w + sh.w + rres[a][b].w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *(float4 *)(p.out + obase + (long)(a * p.Wd + b) * CV_C + co) = v;
                    }
        }
        __syncthreads();
    }
}

// U[p = 4 xi + nu][cin / 4][cout][cin % 4] = sum_ij G[xi][i] g[i][j][cin][cout] G[nu][j], evaluated in float64
__global__ void k_wino_weights(const float *__restrict__ w, float *__restrict__ U)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CV_C * CV_C) return;
    const int ci = i / CV_C, co = i - ci * CV_C;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double g[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = (double)w[((long)(a * 3 + b) * CV_C + ci) * CV_C + co];
    for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu) {
            double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) s += G[xi][a] * g[a][b] * G[nu][b];
            U[(((long)(4 * xi + nu) * 32 + ci / 4) * CV_C + co) * 4 + (ci & 3)] = (float)s;
        }
}

// ------------------------------------------------------------------------------------------
// stem: 3x3 conv over the 3 observation planes (NHWC), BN, ReLU.  K = 27: VALU work, weights in LDS.
// One block = 64 pixels x 128 output channels; thread (tid & 127) owns a channel, (tid >> 7) a pixel parity.
// ------------------------------------------------------------------------------------------
struct StemArgs {
    const float *x;      // [M][3]
    const float *w;      // [3][3][3][128] (kh, kw, cin, cout) = Keras kernel layout
    const float *scale, *shift;
    float *out;          // [M][128]
    int M, Hd, Wd;
    int raw;             // != 0 (k_stem_conv_mfma only): the bare convolution -- no scale / shift (both may be NULL), no ReLU
    const unsigned *bbox; // k_stem_conv_mfma only, or NULL: per image y0 | x0 << 8 | y1 << 16 | x1 << 24 (k_obs_bbox, conv_split.hip);
    int grow;            //   only the pixels of that box grown by `grow` (cut to the canvas) are computed and written
    int group;           // k_stem_conv_mfma: images per iteration of a block (their padded copies sit side by side in LDS); 0 = 1
    const float *center; // k_stem_conv_mfma<0, true> (training step): per-channel centre of the batch-norm sums (or NULL = 0)
    float *stat_part;    //   [gridDim.x][2][128] sums of (out - center) and (out - center)^2 over the block's images
    float *amax_part;    //   or NULL; [gridDim.x][128] largest |out - center| per channel (the range of a deferred batch norm's output)
};

// 256 pixels per block; thread (pg = tid / 32, cq = tid % 32) computes outputs 4 cq .. 4 cq + 3 of pixels pg + 8 k with its
// 27 x 4 weights in registers: per pixel 7 broadcast ds_read_b128 of the gathered 27-value patch feed 108 FMAs, and the
// 32 lanes of a pixel write its whole 512-byte row.
#define ST_PX 256
__global__ __launch_bounds__(256) void k_stem_conv(StemArgs p)
{
    __shared__ __align__(16) float patch[ST_PX][28];
    const int tid = threadIdx.x, cq = tid & 31, pg = tid >> 5;
    const int HW = p.Hd * p.Wd;
    const int m0 = blockIdx.x * ST_PX;
    // gather the 27-value input patch (tap-major, channel-minor = the Keras kernel's row order) of each pixel
    for (int i = tid; i < ST_PX * 27; i += 256) {
        const int px = i / 27, q = i - px * 27;
        const int tap = q / 3, ch = q - tap * 3;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int m = m0 + px;
        float v = 0.f;
        if (m < p.M) {
            const int rem = m % HW;
            const int y = rem / p.Wd, x = rem - y * p.Wd;
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < p.Hd && xx >= 0 && xx < p.Wd) v = p.x[((long)m + dy * p.Wd + dx) * 3 + ch];
        }
        patch[px][q] = v;
    }
    if (tid < ST_PX) patch[tid][27] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wlo[27], whi[27];                    // outputs (4 cq, 4 cq + 1) and (4 cq + 2, 4 cq + 3): v_pk_fma_f32 operands
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        const float4 w_ = *(const float4 *)(p.w + q * CV_C + 4 * cq);
        wlo[q] = (f32x2){w_.x, w_.y}; whi[q] = (f32x2){w_.z, w_.w};
    }
    const float4 sc = *(const float4 *)(p.scale + 4 * cq), sh = *(const float4 *)(p.shift + 4 * cq);
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < ST_PX / 8; ++k) {
        const int px = pg + 8 * k;
        const int m = m0 + px;
        if (m >= p.M) break;
        f32x2 alo = (f32x2){0.f, 0.f}, ahi = (f32x2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float4 pv = *(const float4 *)&patch[px][4 * j];
#define ST_FMA(PV, Q) alo = __builtin_elementwise_fma((f32x2){PV, PV}, wlo[Q], alo); ahi = __builtin_elementwise_fma((f32x2){PV, PV}, whi[Q], ahi);
            ST_FMA(pv.x, 4 * j + 0) ST_FMA(pv.y, 4 * j + 1) ST_FMA(pv.z, 4 * j + 2)
            if (j < 6) { ST_FMA(pv.w, 4 * j + 3) }
#undef ST_FMA
        }
        const float4 a = make_float4(alo.x, alo.y, ahi.x, ahi.y);
        float4 v;
        v.x = fmaxf(a.x * sc.x + sh.x, 0.f); v.y = fmaxf(a.y * sc.y + sh.y, 0.f);
        v.z = fmaxf(a.z * sc.z + sh.z, 0.f); v.w = fmaxf(a.w * sc.w + sh.w, 0.f);
        *(float4 *)(p.out + (long)m * CV_C + 4 * cq) = v;
    }
}

// The same layer with one block per image (the form the launcher picks whenever the padded image fits 64 KB of LDS):
// the image is read once, coalesced, into LDS with a zero border ([(H+2)][(W+2)][3] floats), so a pixel's 27-value
// patch is three runs of 9 consecutive floats at constant offsets from one address -- no per-element index arithmetic
// (the gather of k_stem_conv costs as many instructions as its FMAs: 2.3 TB/s of output, round 1).  Thread
// (pg = tid / 32, cq = tid % 32) computes outputs 4 cq .. 4 cq + 3 of pixels pg, pg + 8, ...; its 27 x 4 weights stay in
// registers; the 32 lanes of a pixel write its 512-byte row.  Same tap order as k_stem_conv: identical sums.
template <int OUT16>           // OUT16: the output activations are written as f16 (1) or bf16 (2): the reduced-precision towers with 16-bit activations
__global__ __launch_bounds__(256) void k_stem_conv_img(StemArgs p)
{
    extern __shared__ __align__(16) float st_img[];
    const int tid = threadIdx.x, cq = tid & 31, pg = tid >> 5;
    const int Hd = p.Hd, Wd = p.Wd, HW = Hd * Wd, row3 = Wd * 3, P3 = (Wd + 2) * 3;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wlo[27], whi[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        const float4 w_ = *(const float4 *)(p.w + q * CV_C + 4 * cq);
        wlo[q] = (f32x2){w_.x, w_.y}; whi[q] = (f32x2){w_.z, w_.w};
    }
    const float4 sc = *(const float4 *)(p.scale + 4 * cq), sh = *(const float4 *)(p.shift + 4 * cq);
    const int n_img = p.M / HW;
    // persistent blocks: the weights are loaded once per block, the images of the batch are taken round-robin
    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
    const float *src = p.x + (long)img * HW * 3;
    __syncthreads();                                           // the previous image's readers are done with the LDS image
    for (int j = tid; j < (Hd + 2) * P3; j += 256) {
        const int yy = j / P3, rr = j - yy * P3;
        const bool inside = yy >= 1 && yy <= Hd && rr >= 3 && rr < 3 + row3;
        st_img[j] = inside ? src[(yy - 1) * row3 + rr - 3] : 0.f;
    }
    float *out = p.out + (long)img * HW * CV_C + 4 * cq;
    _Float16 *out16 = (_Float16 *)p.out + (long)img * HW * CV_C + 4 * cq;
    __syncthreads();
    int y = pg / Wd, x = pg - y * Wd;
    for (int px = pg; px < HW; px += 8) {
        const float *b = st_img + y * P3 + x * 3;              // padded (y, x) = top-left tap of pixel (y, x)
        f32x2 alo = (f32x2){0.f, 0.f}, ahi = (f32x2){0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float v = b[r * P3 + c];
                alo = __builtin_elementwise_fma((f32x2){v, v}, wlo[9 * r + c], alo);
                ahi = __builtin_elementwise_fma((f32x2){v, v}, whi[9 * r + c], ahi);
            }
        float4 v;
        v.x = fmaxf(alo.x * sc.x + sh.x, 0.f); v.y = fmaxf(alo.y * sc.y + sh.y, 0.f);
        v.z = fmaxf(ahi.x * sc.z + sh.z, 0.f); v.w = fmaxf(ahi.y * sc.w + sh.w, 0.f);
        if (OUT16 == 2) {
            typedef __bf16 st_bf16x4 __attribute__((ext_vector_type(4)));
            st_bf16x4 o_;
            o_[0] = (__bf16)v.x; o_[1] = (__bf16)v.y; o_[2] = (__bf16)v.z; o_[3] = (__bf16)v.w;    // v_cvt_pk_bf16_f32: round to nearest even
            *(st_bf16x4 *)(out16 + (long)px * CV_C) = o_;
        } else if (OUT16) {
            typedef _Float16 st_f16x4 __attribute__((ext_vector_type(4)));
            st_f16x4 o_;
            o_[0] = (_Float16)fminf(v.x, 65504.f); o_[1] = (_Float16)fminf(v.y, 65504.f);      // post-ReLU: saturate upwards only
            o_[2] = (_Float16)fminf(v.z, 65504.f); o_[3] = (_Float16)fminf(v.w, 65504.f);
            *(st_f16x4 *)(out16 + (long)px * CV_C) = o_;
        } else *(float4 *)(out + (long)px * CV_C) = v;
        x += 8;
        while (x >= Wd) { x -= Wd; ++y; }
    }
    }
}

// The stem on the matrix pipe.  k_stem_conv_img is VALU-bound at the clock the chip holds under packed-FMA load (measured:
// 480-550 us for 7 483-8 192 images, 3.2-3.9 TB/s of output, against 244 us for a plain fill of the same bytes); the same
// 3 MFLOP per image cost the matrix pipe next to nothing, so the kernel becomes a store stream.  GEMM: rows = the image's
// pixels (32 per M tile), K = 27 patch values padded to 32 (two k steps of v_mfma_f32_32x32x16_f16), N = 128 outputs (4 N
// tiles).  Float32 accuracy the same way as the tower (conv_split.hip): operands carried as f16 hi + lo after power-of-two
// scales (inputs x 2^10: observation values are at most ~2.5; weights by the power of two that brings max |w| to [256, 512)),
// hi*hi + hi*lo + lo*hi with float32 accumulation, the scales undone in the batch-norm scale.  One block = persistent loop
// over images; the padded image sits in LDS as float32 ([(H+2)][(W+2)][3]); wave w takes M tiles w, w + 4, ...: per tile
// each lane gathers its 16 patch values (k = 16 s + 8 h + j -> tap row k / 9, column k % 9: constant LDS offsets from the
// pixel's address), splits them, and runs 24 MFMAs against the split weights it keeps in 64 VGPRs.  Output straight from
// the accumulators: for each accumulator register the two 32-lane halves write 128 contiguous bytes of two pixel rows.
typedef _Float16 sm_f16x8 __attribute__((ext_vector_type(8)));
typedef float sm_f32x16 __attribute__((ext_vector_type(16)));
// STATS (training step, with raw): the per-channel sums the stem's batch norm starts from are taken from the values on their way
//   out -- a lane owns four channels (32 nt + l31) of the rows it stores -- and leave as one partial row per block: the pass
//   snk_bn_train_sums_f64 makes over the 462 MB output (0.11 ms per step, in every step) is not needed
template <int OUT16, bool STATS = false>
__global__ __launch_bounds__(256) void k_stem_conv_mfma(StemArgs p)
{
    extern __shared__ __align__(16) float st_img[];
    __shared__ float s_wmax[4];
    __shared__ float s_stat[STATS ? 4 : 1][STATS ? 3 : 1][STATS ? CV_C : 1];
    float cen[4] = {0.f, 0.f, 0.f, 0.f}, ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f}, smax[4] = {0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int Hd = p.Hd, Wd = p.Wd, HW = Hd * Wd, row3 = Wd * 3, P3 = (Wd + 2) * 3;
    const int n_img = p.M / HW;
    // weight scale: max |w| over the 27 x 128 kernel -> 2^k with 256 <= max * 2^k < 512
    float wm = 0.f;
    for (int i = tid; i < 27 * CV_C; i += 256) wm = fmaxf(wm, fabsf(p.w[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
    if (lane == 0) s_wmax[wv] = wm;
    __syncthreads();
    wm = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
    int wk = 0;
    if (wm > 0.f && wm < 3.0e38f) wk = 8 - ilogbf(wm);
    wk = max(-100, min(100, wk));
    const float wmul = ldexpf(1.0f, wk);
    constexpr int AX = 10;                                     // input scale 2^10
    const float amul = 1024.0f, undo = ldexpf(1.0f, -wk - AX);
    // B fragments: lane (column 32 nt + l31, k half h) holds k = 16 s + 8 h + j, j = 0..7, as hi and lo
    sm_f16x8 Bh[4][2], Bl[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s_ + 8 * h + j;
                const float v = k < 27 ? p.w[k * CV_C + 32 * nt + l31] * wmul : 0.f;
                const _Float16 hi_ = (_Float16)v;
                Bh[nt][s_][j] = hi_;
                Bl[nt][s_][j] = (_Float16)(v - (float)hi_);
            }
    // LDS offsets (in floats, relative to the pixel's top-left tap) of this lane's 16 patch values; k >= 27 reads offset 0, masked
    int koff[2][8];
    unsigned kvalid = 0;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s_ + 8 * h + j;
            koff[s_][j] = k < 27 ? (k / 9) * P3 + (k % 9) : 0;
            if (k < 27) kvalid |= 1u << (8 * s_ + j);
        }
    float scl[4], shf[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        scl[nt] = (p.raw ? 1.f : p.scale[32 * nt + l31]) * undo;
        shf[nt] = p.raw ? 0.f : p.shift[32 * nt + l31];
    }
    const float floor_ = p.raw ? -__builtin_inff() : 0.f;
    if (STATS && p.center) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) cen[nt] = p.center[32 * nt + l31];
    }

    const int n_lds = (Hd + 2) * P3;
    // A block takes G images per iteration: one load / barrier / compute / store round per image left the wavefronts unevenly
    // loaded (6 tiles of a 13 x 13 rectangle over 4 wavefronts) and the chain's latency exposed.  The tiles of the G images
    // form one list; their rectangles (s_par: ry0, rx0, wr, HWr, first tile) are worked out by one thread during the load.
    __shared__ int s_par[4][5];
    __shared__ int s_tiles;
    const int G = max(1, min(4, p.group));
    for (int base = blockIdx.x * G; base < n_img; base += gridDim.x * G) {
        const int ng = min(G, n_img - base);
        __syncthreads();                                       // the previous images' readers are done with the LDS images
        for (int j = tid; j < ng * n_lds; j += 256) {
            const int g_ = j / n_lds, jj = j - g_ * n_lds;
            const float *src = p.x + (long)(base + g_) * HW * 3;
            const int yy = jj / P3, rr = jj - yy * P3;
            const bool inside = yy >= 1 && yy <= Hd && rr >= 3 && rr < 3 + row3;
            // observation values are bounded by construction (game.py:229-248: at most (H W + 2.5) * 0.04 = 14.6 on 19x19), so
            // 2^10 leaves 4x headroom to the f16 range; anything a caller passes beyond +-63.97 saturates instead of becoming inf
            st_img[j] = inside ? __builtin_amdgcn_fmed3f(src[(yy - 1) * row3 + rr - 3] * amul, -65504.f, 65504.f) : 0.f;
        }
        if (tid == 0) {
            // the GEMM rows of an image are the pixels of a rectangle of it (all of it without a bounding box)
            int first = 0;
            for (int g_ = 0; g_ < ng; ++g_) {
                int ry0 = 0, rx0 = 0, wr = Wd, HWr = HW;
                if (p.bbox) {
                    const unsigned bb = p.bbox[base + g_];
                    ry0 = max((int)(bb & 255) - p.grow, 0); rx0 = max((int)((bb >> 8) & 255) - p.grow, 0);
                    wr = min((int)(bb >> 24) + p.grow, Wd - 1) - rx0 + 1;
                    HWr = (min((int)((bb >> 16) & 255) + p.grow, Hd - 1) - ry0 + 1) * wr;
                }
                s_par[g_][0] = ry0; s_par[g_][1] = rx0; s_par[g_][2] = wr; s_par[g_][3] = HWr; s_par[g_][4] = first;
                first += (HWr + 31) / 32;
            }
            s_tiles = first;
        }
        __syncthreads();                                       // (prefetching the next image through registers into a second
        //                                                        buffer was measured: no change, the loads are not what it waits for)
        const int n_tiles = s_tiles;
        for (int u = wv; u < n_tiles; u += 4) {
            int g_ = 0;
            for (int k = 1; k < ng; ++k) if (u >= s_par[k][4]) g_ = k;
            const int ry0 = s_par[g_][0], rx0 = s_par[g_][1], wr = s_par[g_][2], HWr = s_par[g_][3], t = u - s_par[g_][4];
            const float inv_wr = 1.0f / (float)wr;
            float *out = p.out + (long)(base + g_) * HW * CV_C;
            _Float16 *out16 = (_Float16 *)p.out + (long)(base + g_) * HW * CV_C;
            const int pix = min(32 * t + l31, HWr - 1);        // rows past the image repeat its last pixel (computed, never stored)
            const int yr = (int)(((float)pix + 0.5f) * inv_wr);
            const int y = ry0 + yr, x = rx0 + pix - yr * wr;
            const float *b = st_img + g_ * n_lds + y * P3 + x * 3;
            sm_f16x8 Ah[2], Al[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = b[koff[s_][j]];
                    if (!((kvalid >> (8 * s_ + j)) & 1u)) v = 0.f;
                    const _Float16 hi_ = (_Float16)v;
                    Ah[s_][j] = hi_;
                    Al[s_][j] = (_Float16)(v - (float)hi_);
                }
            sm_f32x16 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s_], Bh[nt][s_], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s_], Bl[nt][s_], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s_], Bh[nt][s_], acc[nt], 0, 0, 0);
                }
            }
            // output straight from the accumulators: register r holds rows (r & 3) + 8 (r >> 2) + 4 h of the tile (C layout of
            // the 32x32 MFMA), so one store instruction writes 128 contiguous bytes of two pixel rows.  Measured alternatives:
            // float4 stores through an LDS patch, one N tile at a time (same time) or whole 512-byte rows (slower: 476 vs 372 us)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (rr < HWr) {
                    const int yr = (int)(((float)rr + 0.5f) * inv_wr);
                    const int row = (ry0 + yr) * Wd + rx0 + rr - yr * wr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float v = fmaxf(__builtin_fmaf(acc[nt][r], scl[nt], shf[nt]), floor_);
                        if (STATS) { const float e_ = v - cen[nt]; ssum[nt] += e_; ssq[nt] += e_ * e_; smax[nt] = fmaxf(smax[nt], fabsf(e_)); }
                        if (OUT16 == 2) ((__bf16 *)out16)[(long)row * CV_C + 32 * nt + l31] = (__bf16)v;
                        else if (OUT16) out16[(long)row * CV_C + 32 * nt + l31] = (_Float16)fminf(v, 65504.f);
                        else out[(long)row * CV_C + 32 * nt + l31] = v;
                    }
                }
            }
        }
    }
    if (STATS) {                                // the two row halves of a lane pair, then the four wavefronts, in a fixed order
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            ssum[nt] += __shfl_xor(ssum[nt], 32, 64);
            ssq[nt] += __shfl_xor(ssq[nt], 32, 64);
            smax[nt] = fmaxf(smax[nt], __shfl_xor(smax[nt], 32, 64));
            if (h == 0) { s_stat[wv][0][32 * nt + l31] = ssum[nt]; s_stat[wv][1][32 * nt + l31] = ssq[nt]; s_stat[wv][2][32 * nt + l31] = smax[nt]; }
        }
        __syncthreads();
        const int q = tid >> 7, c = tid & 127;
        p.stat_part[(size_t)blockIdx.x * 256 + tid] = ((s_stat[0][q][c] + s_stat[1][q][c]) + s_stat[2][q][c]) + s_stat[3][q][c];
        if (p.amax_part && tid < CV_C)
            p.amax_part[(size_t)blockIdx.x * CV_C + tid] = fmaxf(fmaxf(s_stat[0][2][tid], s_stat[1][2][tid]), fmaxf(s_stat[2][2][tid], s_stat[3][2][tid]));
    }
}

// ------------------------------------------------------------------------------------------
// head: conv1x1 (C -> 1) + BN + ReLU -> Flatten (HW) -> Dense(128) + ReLU -> Dense(3) + tanh,
// then AlphaNNet.v's obstacle overwrite (alpha_nnet.py:67-72).  One block per state.
// ------------------------------------------------------------------------------------------
struct HeadArgs {
    const float *x;        // [n][HW][128]
    const float *w1x1;     // [128]
    float s1, b1;          // folded BN of the single channel
    const float *fc1_w;    // [HW][128]  (Keras Dense kernel (in, out))
    const float *fc1_b;    // [128]
    const float *fc2_w;    // [128][3]
    const float *fc2_b;    // [3]
    const uint8_t *mask;   // [n][3] or NULL
    float *q;              // [n][3]
    int n, HW;
};

__global__ __launch_bounds__(256) void k_head(HeadArgs p)
{
    extern __shared__ float sm[];
    float *h1 = sm;               // [HW]
    float *h2 = sm + p.HW;        // [128]
    float *part = h2 + 128;       // [256]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.x;
    const float *xs = p.x + (long)s * p.HW * CV_C;
    const float w_a = p.w1x1[lane], w_b = p.w1x1[lane + 64];
    for (int px = wv; px < p.HW; px += 4) {
        float v = xs[(long)px * CV_C + lane] * w_a + xs[(long)px * CV_C + lane + 64] * w_b;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) h1[px] = fmaxf(v * p.s1 + p.b1, 0.f);
    }
    __syncthreads();
    // Dense(HW -> 128): two half-sums per output, combined through LDS
    {
        const int j = tid & 127, half = tid >> 7;
        const int mid = (p.HW + 1) / 2;
        const int lo = half ? mid : 0, hi = half ? p.HW : mid;
        float acc = 0.f;
        for (int i = lo; i < hi; ++i) acc = fmaf(h1[i], p.fc1_w[(long)i * 128 + j], acc);
        part[tid] = acc;
    }
    __syncthreads();
    if (tid < 128) h2[tid] = fmaxf(part[tid] + part[tid + 128] + p.fc1_b[tid], 0.f);
    __syncthreads();
    if (wv < 3) {
        float v = h2[lane] * p.fc2_w[lane * 3 + wv] + h2[lane + 64] * p.fc2_w[(lane + 64) * 3 + wv];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) {
            float qv = tanhf(v + p.fc2_b[wv]);
            if (p.mask && p.mask[(long)s * 3 + wv]) qv = -1.0f;
            p.q[(long)s * 3 + wv] = qv;
        }
    }
}

// The head behind a fused 1x1 stage (snk_conv3x3_bn_f16s_head wrote h1 = relu(bn(conv1x1)) [n][HW]):
// Flatten -> Dense(128) + ReLU -> Dense(3) + tanh -> obstacle overwrite.  16 states per block, so the 226 KB Dense
// kernel is read once per 16 states instead of once per state; thread (j = tid % 128, half = tid / 128) accumulates
// output j of 8 states.  (HD_S = 8 or 4 states per block where sixteen states' h1 do not fit 64 KB of LDS: 37 x 37.)
template <int HD_S>
__global__ __launch_bounds__(256) void k_head_dense(HeadArgs p, const float *__restrict__ h1g)
{
    extern __shared__ float sm[];
    float *h1 = sm;                        // [HD_S][HW]
    float *h2 = sm + HD_S * p.HW;          // [HD_S][128]
    const int tid = threadIdx.x, j = tid & 127, half = tid >> 7;
    const int s0 = blockIdx.x * HD_S, ns = min(HD_S, p.n - s0);
    for (int i = tid; i < HD_S * p.HW; i += 256) h1[i] = i < ns * p.HW ? h1g[(long)s0 * p.HW + i] : 0.f;
    __syncthreads();
    float acc[HD_S / 2];
#pragma unroll
    for (int k = 0; k < HD_S / 2; ++k) acc[k] = 0.f;
    const float *hh = h1 + half * (HD_S / 2) * p.HW;
    int i = 0;
    for (; i + 8 <= p.HW; i += 8) {             // eight Dense rows in flight per thread: the loop is load-latency bound otherwise
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = p.fc1_w[(long)(i + u) * 128 + j];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < HD_S / 2; ++k) acc[k] = fmaf(hh[k * p.HW + i + u], w[u], acc[k]);
    }
    for (; i < p.HW; ++i) {
        const float w = p.fc1_w[(long)i * 128 + j];
#pragma unroll
        for (int k = 0; k < HD_S / 2; ++k) acc[k] = fmaf(hh[k * p.HW + i], w, acc[k]);
    }
    const float b = p.fc1_b[j];
#pragma unroll
    for (int k = 0; k < HD_S / 2; ++k) h2[(half * (HD_S / 2) + k) * 128 + j] = fmaxf(acc[k] + b, 0.f);
    __syncthreads();
    if (tid < HD_S * 3) {
        const int sl = tid / 3, o = tid - sl * 3;
        if (sl < ns) {
            float v = 0.f;
            for (int i = 0; i < 128; ++i) v = fmaf(h2[sl * 128 + i], p.fc2_w[i * 3 + o], v);
            float qv = tanhf(v + p.fc2_b[o]);
            if (p.mask && p.mask[(long)(s0 + sl) * 3 + o]) qv = -1.0f;
            p.q[(long)(s0 + sl) * 3 + o] = qv;
        }
    }
}

// weights (kh, kw, cin, cout) -> (tap, cout, cin) so that both GEMM operands are K-contiguous
__global__ void k_transpose_w(const float *__restrict__ w, float *__restrict__ wT)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * CV_C * CV_C) return;
    const int tap = i / (CV_C * CV_C), r = i - tap * CV_C * CV_C;
    const int co = r / CV_C, ci = r - co * CV_C;
    wT[i] = w[(long)(tap * CV_C + ci) * CV_C + co];
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int snk_conv3x3_prepare_weights(const float *d_w_hwio, float *d_wT, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wT, "snk_conv3x3_prepare_weights: NULL argument");
    k_transpose_w<<<(9 * CV_C * CV_C + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, d_wT);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_bn_f32(const float *d_x, const float *d_wT, const float *d_scale, const float *d_shift,
                                  const float *d_residual, float *d_out, int n_images, int height, int width,
                                  int relu, void *stream)
{
    SNK_REQUIRE(d_x && d_wT && d_scale && d_shift && d_out, "snk_conv3x3_bn_f32: NULL argument");
    SNK_REQUIRE(d_out != d_x, "snk_conv3x3_bn_f32: in-place convolution is not possible");
    if (n_images <= 0) return 0;
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "snk_conv3x3_bn_f32: batch of %d images too large for one call", n_images);
    ConvArgs a = {d_x, d_wT, d_scale, d_shift, d_residual, d_out, (int)M, height, width, relu};
    static int variant = -1;
    if (variant < 0) { const char *v = getenv("SNK_CONV_BK"); variant = v ? atoi(v) : 16; }
    if (variant == 32) k_conv3x3_f32<32, 2><<<(int)((M + CV_BM - 1) / CV_BM), 256, 0, (hipStream_t)stream>>>(a);
    else k_conv3x3_f32<16, 3><<<(int)((M + CV_BM - 1) / CV_BM), 256, 0, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}



extern "C" int snk_conv3x3_prepare_weights_winograd(const float *d_w_hwio, float *d_U, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_U, "snk_conv3x3_prepare_weights_winograd: NULL argument");
    k_wino_weights<<<(CV_C * CV_C + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, d_U);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_bn_f32_winograd(const float *d_x, const float *d_U, const float *d_scale, const float *d_shift,
                                           const float *d_residual, float *d_out, int n_images, int height, int width,
                                           int relu, void *stream)
{
    SNK_REQUIRE(d_x && d_U && d_scale && d_shift && d_out, "snk_conv3x3_bn_f32_winograd: NULL argument");
    SNK_REQUIRE(d_out != d_x, "snk_conv3x3_bn_f32_winograd: in-place convolution is not possible");
    if (n_images <= 0) return 0;
    const int TY = (height + 1) / 2, TX = (width + 1) / 2;
    const long tiles = (long)n_images * TY * TX;
    SNK_REQUIRE((long)n_images * height * width * CV_C < (1l << 32),      // 32-bit element offsets inside the kernel
                "snk_conv3x3_bn_f32_winograd: batch of %d images too large for one call (chunk it)", n_images);
    WinoArgs a = {d_x, d_U, d_scale, d_shift, d_residual, d_out, (int)tiles, height, width, TY, TX, relu};
    static int nw = -1;
    if (nw < 0) { const char *v = getenv("SNK_WINO_WAVES"); nw = v ? atoi(v) : 8; }
    if (nw == 8) k_conv3x3_wino_f32<8><<<(int)((tiles + WG_TB - 1) / WG_TB), 512, 0, (hipStream_t)stream>>>(a);
    else k_conv3x3_wino_f32<4><<<(int)((tiles + WG_TB - 1) / WG_TB), 256, 0, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// images a block of k_stem_conv_mfma takes per iteration (their padded float32 copies share the 64 KB of dynamic LDS), and the
// launch that goes with it; SNK_STEM_GROUP=1 restores one image per iteration (A/B runs)
template <int OUT16, bool STATS = false>
static int stem_mfma_launch(StemArgs a, int n_images, size_t lds_one, int max_grid, hipStream_t st)
{
    static const int group_max = getenv("SNK_STEM_GROUP") ? max(1, min(4, atoi(getenv("SNK_STEM_GROUP")))) : 4;
    int g = (int)min((size_t)group_max, (size_t)((STATS ? 57 : 64) * 1024) / lds_one);     // (STATS: 6 KB of static LDS for the sums and maxima)
    g = max(1, min(g, n_images));
    a.group = g;
    const int grid = min((n_images + g - 1) / g, max_grid);
    k_stem_conv_mfma<OUT16, STATS><<<grid, 256, g * lds_one, st>>>(a);
    return grid;
}

extern "C" int snk_stem_conv_bn_relu_f32(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                         float *d_out, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_scale && d_shift && d_out, "snk_stem_conv_bn_relu_f32: NULL argument");
    if (n_images <= 0) return 0;
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "snk_stem_conv_bn_relu_f32: batch too large");
    StemArgs a = {d_x, d_w, d_scale, d_shift, d_out, (int)M, height, width};
    const size_t lds = (size_t)(height + 2) * (width + 2) * 3 * sizeof(float);
    // default: the MFMA form on 512 persistent blocks (2 resident per CU); SNK_STEM=valu selects the packed-FMA form
    static const int stem_grid = getenv("SNK_STEM_GRID") ? atoi(getenv("SNK_STEM_GRID")) : 512;
    static const int stem_valu = getenv("SNK_STEM") ? !strcmp(getenv("SNK_STEM"), "valu") : 0;
    if (lds <= 64 * 1024 && !stem_valu) stem_mfma_launch<0>(a, n_images, lds, stem_grid, (hipStream_t)stream);
    else if (lds <= 64 * 1024) k_stem_conv_img<0><<<min(n_images, 768), 256, lds, (hipStream_t)stream>>>(a);
    else k_stem_conv<<<(int)((M + ST_PX - 1) / ST_PX), 256, 0, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the stem on the bounding box of every observation's non-background pixels grown by `grow` pixels (d_bbox: what
// snk_conv_rect_plan wrote); pixels of d_out outside that rectangle are left untouched
extern "C" int snk_stem_conv_bn_relu_f32_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                              float *d_out, const void *d_bbox, int grow, int n_images, int height, int width,
                                              void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_scale && d_shift && d_out && d_bbox && grow >= 1 && grow < 128, "snk_stem_conv_bn_relu_f32_rect: bad argument");
    if (n_images <= 0) return 0;
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "snk_stem_conv_bn_relu_f32_rect: batch too large");
    const size_t lds = (size_t)(height + 2) * (width + 2) * 3 * sizeof(float);
    SNK_REQUIRE(lds <= 64 * 1024 && height <= 255 && width <= 255, "snk_stem_conv_bn_relu_f32_rect: observation %d x %d too large", height, width);
    StemArgs a = {d_x, d_w, d_scale, d_shift, d_out, (int)M, height, width, 0, (const unsigned *)d_bbox, grow};
    stem_mfma_launch<0>(a, n_images, lds, 512, (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the same with the output written as f16 / bf16 (the towers with 16-bit activations: snk_conv3x3_bn_f16_act16_rect, _bf16_act16_rect)
static int stem_out16_launch(const char *who, int bf, const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                             void *d_out16, const void *d_bbox, int grow, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_scale && d_shift && d_out16, "%s: NULL argument", who);
    SNK_REQUIRE(!d_bbox || (grow >= 1 && grow < 128 && height <= 255 && width <= 255), "%s: bad rectangle arguments", who);
    if (n_images <= 0) return 0;
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "%s: batch too large", who);
    const size_t lds = (size_t)(height + 2) * (width + 2) * 3 * sizeof(float);
    SNK_REQUIRE(lds <= 64 * 1024, "%s: observation %d x %d too large", who, height, width);
    StemArgs a = {d_x, d_w, d_scale, d_shift, (float *)d_out16, (int)M, height, width, 0, (const unsigned *)d_bbox, d_bbox ? grow : 0};
    if (bf) stem_mfma_launch<2>(a, n_images, lds, 512, (hipStream_t)stream);
    else stem_mfma_launch<1>(a, n_images, lds, 512, (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_stem_conv_bn_relu_f16out_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                                 void *d_out16, const void *d_bbox, int grow, int n_images, int height, int width,
                                                 void *stream)
{
    SNK_REQUIRE(d_bbox, "snk_stem_conv_bn_relu_f16out_rect: NULL bounding boxes");
    return stem_out16_launch("snk_stem_conv_bn_relu_f16out_rect", 0, d_x, d_w, d_scale, d_shift, d_out16, d_bbox, grow, n_images, height, width, stream);
}

extern "C" int snk_stem_conv_bn_relu_bf16out_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                                  void *d_out16, const void *d_bbox, int grow, int n_images, int height, int width,
                                                  void *stream)
{
    SNK_REQUIRE(d_bbox, "snk_stem_conv_bn_relu_bf16out_rect: NULL bounding boxes");
    return stem_out16_launch("snk_stem_conv_bn_relu_bf16out_rect", 1, d_x, d_w, d_scale, d_shift, d_out16, d_bbox, grow, n_images, height, width, stream);
}

// the bare stem convolution (training step: the batch statistics of its output come first, train.hip)
extern "C" int snk_stem_conv_f32(const float *d_x, const float *d_w, float *d_out, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_out, "snk_stem_conv_f32: NULL argument");
    if (n_images <= 0) return 0;
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "snk_stem_conv_f32: batch too large");
    const size_t lds = (size_t)(height + 2) * (width + 2) * 3 * sizeof(float);
    SNK_REQUIRE(lds <= 64 * 1024, "snk_stem_conv_f32: observation %d x %d too large", height, width);
    StemArgs a = {d_x, d_w, nullptr, nullptr, d_out, (int)M, height, width, 1};
    stem_mfma_launch<0>(a, n_images, lds, 512, (hipStream_t)stream);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// per-channel maximum over at most 512 rows of 128: [n_rows][128] -> [128]
// (eight row lanes x 32 float4 columns: whole 512-byte rows per load, 64 loads in flight per lane group instead of 512 one after the other)
__global__ __launch_bounds__(256) void k_stem_amax_fold(const float *__restrict__ part, int n_rows, float *__restrict__ amax)
{
    __shared__ float4 sh[8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int r = rl; r < n_rows; r += 8) {
        const float4 v = *(const float4 *)(part + (size_t)r * CV_C + 4 * cq);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    sh[rl][cq] = m;
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = sh[r][cq]; m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w); }
        *(float4 *)(amax + 4 * cq) = m;
    }
}

static int stem_stats_launch(const char *who, const float *d_x, const float *d_w, float *d_out, const float *d_center, float *d_amax,
                             float *d_partials, double *d_sums, int n_images, int height, int width, void *stream)
{
    const long M = (long)n_images * height * width;
    SNK_REQUIRE(M < (1l << 31), "%s: batch too large", who);
    const size_t lds = (size_t)(height + 2) * (width + 2) * 3 * sizeof(float);
    SNK_REQUIRE(lds <= 57 * 1024, "%s: observation %d x %d too large", who, height, width);
    float *amax_part = d_amax ? d_partials + 512 * 256 : nullptr;          // (behind the at most 512 blocks' sums)
    StemArgs a = {d_x, d_w, nullptr, nullptr, d_out, (int)M, height, width, 1, nullptr, 0, 0, d_center, d_partials, amax_part};
    const int grid = stem_mfma_launch<0, true>(a, n_images, lds, 512, (hipStream_t)stream);
    tf_fold<double>(d_partials, grid, 256, 256, 1.0, d_sums, (double *)(d_partials + 2048 * 256), (hipStream_t)stream);
    if (d_amax) k_stem_amax_fold<<<1, 256, 0, (hipStream_t)stream>>>(amax_part, grid, d_amax);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// the same with the sums its batch norm starts from taken in the kernel's epilogue: d_sums[0..127] = sum over all pixels of (out -
// center), d_sums[128..255] = sum of (out - center)^2 (float64; d_center: 128 floats or NULL) -- snk_bn_train_sums_f64(d_out) without
// its pass over the output.  d_partials: snk_bn_train_partials() floats.
extern "C" int snk_stem_conv_f32_stats(const float *d_x, const float *d_w, float *d_out, const float *d_center, float *d_partials,
                                       double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_out && d_partials && d_sums && n_images > 0, "snk_stem_conv_f32_stats: bad argument");
    return stem_stats_launch("snk_stem_conv_f32_stats", d_x, d_w, d_out, d_center, nullptr, d_partials, d_sums, n_images, height, width, stream);
}

// ... and, for a stem whose batch norm + ReLU output is never written (deferred, snake_engine/train_step.py), d_amax[128] = the largest
// |out - center| per channel, what snk_bn_train_finalize_range turns into the range of that output
extern "C" int snk_stem_conv_f32_stats_deferred(const float *d_x, const float *d_w, float *d_out, const float *d_center, float *d_amax,
                                                float *d_partials, double *d_sums, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w && d_out && d_amax && d_partials && d_sums && n_images > 0, "snk_stem_conv_f32_stats_deferred: bad argument");
    return stem_stats_launch("snk_stem_conv_f32_stats_deferred", d_x, d_w, d_out, d_center, d_amax, d_partials, d_sums, n_images, height, width, stream);
}

// the whole-canvas stem with the output written as f16 / bf16 [n][H][W][128] (input of snk_conv3x3_bn_f16_act16 / _bf16_act16)
extern "C" int snk_stem_conv_bn_relu_f16out(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                            void *d_out16, int n_images, int height, int width, void *stream)
{
    return stem_out16_launch("snk_stem_conv_bn_relu_f16out", 0, d_x, d_w, d_scale, d_shift, d_out16, nullptr, 0, n_images, height, width, stream);
}

extern "C" int snk_stem_conv_bn_relu_bf16out(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                             void *d_out16, int n_images, int height, int width, void *stream)
{
    return stem_out16_launch("snk_stem_conv_bn_relu_bf16out", 1, d_x, d_w, d_scale, d_shift, d_out16, nullptr, 0, n_images, height, width, stream);
}

extern "C" int snk_head_f32(const float *d_x, const float *d_w1x1, float bn_scale, float bn_shift, const float *d_fc1_w,
                            const float *d_fc1_b, const float *d_fc2_w, const float *d_fc2_b, const uint8_t *d_mask,
                            float *d_q, int n_images, int height, int width, void *stream)
{
    SNK_REQUIRE(d_x && d_w1x1 && d_fc1_w && d_fc1_b && d_fc2_w && d_fc2_b && d_q, "snk_head_f32: NULL argument");
    if (n_images <= 0) return 0;
    HeadArgs a = {d_x, d_w1x1, bn_scale, bn_shift, d_fc1_w, d_fc1_b, d_fc2_w, d_fc2_b, d_mask, d_q, n_images, height * width};
    const size_t lds = (size_t)(height * width + 128 + 256) * sizeof(float);
    k_head<<<n_images, 256, lds, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_head_dense_f32(const float *d_h1, const float *d_fc1_w, const float *d_fc1_b, const float *d_fc2_w,
                                  const float *d_fc2_b, const uint8_t *d_mask, float *d_q, int n_images, int height, int width,
                                  void *stream)
{
    SNK_REQUIRE(d_h1 && d_fc1_w && d_fc1_b && d_fc2_w && d_fc2_b && d_q, "snk_head_dense_f32: NULL argument");
    if (n_images <= 0) return 0;
    const size_t per_state = (size_t)(height * width + 128) * sizeof(float);
    SNK_REQUIRE(4 * per_state <= 64 * 1024, "snk_head_dense_f32: %d x %d observation too large", height, width);
    HeadArgs a = {nullptr, nullptr, 0.f, 0.f, d_fc1_w, d_fc1_b, d_fc2_w, d_fc2_b, d_mask, d_q, n_images, height * width};
    if (16 * per_state <= 64 * 1024) k_head_dense<16><<<(n_images + 15) / 16, 256, 16 * per_state, (hipStream_t)stream>>>(a, d_h1);
    else if (8 * per_state <= 64 * 1024) k_head_dense<8><<<(n_images + 7) / 8, 256, 8 * per_state, (hipStream_t)stream>>>(a, d_h1);
    else k_head_dense<4><<<(n_images + 3) / 4, 256, 4 * per_state, (hipStream_t)stream>>>(a, d_h1);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
