// conv_split.hip -- the Q-net's 3x3 residual-tower layer (alpha_nnet.py:25-47) at float32 accuracy on the
// f16 matrix pipe of gfx950.
//
// Every float32 operand v is carried as two f16 numbers, hi = f16(v) and lo = f16(v - hi) (22 significand bits
// together); a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32 accumulation (the lo*lo
// term is below float32 resolution).  Three 32-cycle MFMAs of k = 16 replace eight 64-cycle f32 MFMAs of k = 2:
// 5.3x the f32 MFMA rate, 2.6x the Winograd kernel's arithmetic rate, at the same |dQ| (about 3e-7 for the whole
// net, tests/test_net_gpu.py).  Weights are pre-scaled by a power of two so that their lo parts are normal f16
// numbers; the inverse goes into the batch-norm scale (exact).  Activations outside +-65504 are clamped.
//
// One block = one row strip of one image (the whole image at 21x21) x all 128 outputs, 4 wavefronts (one per SIMD,
// 512 registers each).  The strip sits in LDS with a zero border, rows pitched W + 1 so that the right border of a row
// is the left border of the next: the output of padded position q needs input position q + dy (W+1) + dx, so all
// nine taps of an MFMA A-fragment are the SAME LDS image at nine constant offsets -- the strip is staged (and split)
// once per 32-channel chunk, not once per tap.  GEMM rows are consecutive padded positions (border positions are
// computed and dropped: 441 of 480 rows useful at 21x21).  Wave (wm, wn) owns M tiles {2i + wm} x N tiles {2wn, 2wn+1}
// = 256 accumulator registers; per (k16 step, tap) it reads 2 ds_read_b128 per M tile for 6 MFMAs.  B fragments
// (the pre-split weights, 576 KB, L2 resident, stored in fragment order so that a wave's load is 1 KB contiguous)
// stream straight into a 6-deep register ring 5 steps ahead; the next chunk's strip is loaded, split and written to
// the other LDS buffer under the MFMAs; one block barrier per chunk.  Epilogue: accumulators * scale + shift -> LDS
// (256 rows at a time), read back as float4 rows, + residual, ReLU, 512-byte pixel rows written.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define HS_C 128
#define HS_LDP 144                                       // bytes per LDS pixel: [hi k0-15 | lo k0-15 | hi k16-31 | lo k16-31] + 16 (16 x odd)
#define HS_NPB 520                                       // pixels per LDS buffer
#define HS_SLACK 72                                      // pixels behind the second buffer that junk GEMM rows may read
#define HS_SMEM ((2 * HS_NPB + HS_SLACK) * HS_LDP)       // 160 128 bytes of the CU's 163 840
#define HS_WS_ELEMS (9 * HS_C * HS_C * 2)                // f16 numbers in the split weight image; a float (2^-k) follows it
#define HS_MLD 132                                       // epilogue row (floats): 528 bytes = 16 x 33
#define HS_STEPS 18                                      // (k16 step, tap) pairs per 32-channel chunk
#define HS_RING 3
#define HS_AHEAD 2

struct ConvHsArgs {
    const float *x;            // [n][Hd][Wd][128]
    const f16x8 *wS;           // [chunk 4][k16 2][tap 9][wn 2][nt 2][hi/lo][lane 64] x 8 f16
    const float *wscale_inv;   // 2^-k
    const float *scale, *shift;
    const float *res;          // or NULL
    float *out;
    int Hd, Wd, R, n_strips, relu;
};

template <int NI>      // M tiles per wave: 8 (up to 512 GEMM rows per block) or 4 (up to 256)
__global__ __launch_bounds__(256, 1) void k_conv3x3_f16s(ConvHsArgs p)
{
    __shared__ __align__(16) unsigned char smem[HS_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv & 1, wn = wv >> 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int img = blockIdx.x / p.n_strips, strip = blockIdx.x - img * p.n_strips;
    const int y0 = strip * p.R, rows = min(p.R, p.Hd - y0);
    const int P = p.Wd + 1, q0 = P + 1;                    // padded pitch; position of the strip's first pixel

    // ---- staging role: item i of this thread = pixel (tid / 8 + 32 i) of the rows the strip needs, float4 (tid % 8) of the chunk
    const int ry_lo = y0 == 0 ? 1 : 0, ry_hi = min(rows + 1, p.Hd - y0);     // padded rows that exist in the image
    const int npx = (ry_hi - ry_lo + 1) * p.Wd;
    const float *xrow = p.x + ((long)img * p.Hd + (y0 + ry_lo - 1)) * p.Wd * HS_C + 4 * (tid & 7);
    const int pix0 = tid >> 3;
    const unsigned lds_c4 = ((tid & 7) >> 2) * 64 + (tid & 3) * 8;
    const float invW = 1.0f / (float)p.Wd;
    float4 st[9];
// items past the strip's last pixel repeat it (same value to the same LDS address): no predication, no branches
#define HS_LOAD(half, c)                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) {                                          \
        const int pix_ = min(pz + 32 * ((half) * 9 + j_), npx - 1);                             \
        st[j_] = *(const float4 *)(xrow + (unsigned)(pix_ * HS_C + 32 * (c)));                  \
    }
#define HS_STORE1(half, j_, bufoff)                                                             \
    {                                                                                           \
        const int pix_ = min(pz + 32 * ((half) * 9 + (j_)), npx - 1);                           \
        const int r_ = (int)(((float)pix_ + 0.5f) * invW), x_ = pix_ - r_ * p.Wd;               \
        unsigned char *d_ = smem + (bufoff) + ((ry_lo + r_) * P + x_ + 1) * HS_LDP + lds_c4;    \
        float4 v_ = st[j_];                                                                     \
        v_.x = fminf(fmaxf(v_.x, -65504.f), 65504.f); v_.y = fminf(fmaxf(v_.y, -65504.f), 65504.f); \
        v_.z = fminf(fmaxf(v_.z, -65504.f), 65504.f); v_.w = fminf(fmaxf(v_.w, -65504.f), 65504.f); \
        f16x4 hi_, lo_;                                                                         \
        hi_[0] = (_Float16)v_.x; hi_[1] = (_Float16)v_.y; hi_[2] = (_Float16)v_.z; hi_[3] = (_Float16)v_.w; \
        lo_[0] = (_Float16)(v_.x - (float)hi_[0]); lo_[1] = (_Float16)(v_.y - (float)hi_[1]);   \
        lo_[2] = (_Float16)(v_.z - (float)hi_[2]); lo_[3] = (_Float16)(v_.w - (float)hi_[3]);   \
        *(f16x4 *)d_ = hi_;                                                                     \
        *(f16x4 *)(d_ + 32) = lo_;                                                              \
    }
#define HS_STORE(half, bufoff)                                                                  \
    {                                                                                           \
        HS_STORE1(half, 0, bufoff) HS_STORE1(half, 1, bufoff) HS_STORE1(half, 2, bufoff)        \
        HS_STORE1(half, 3, bufoff) HS_STORE1(half, 4, bufoff) HS_STORE1(half, 5, bufoff)        \
        HS_STORE1(half, 6, bufoff) HS_STORE1(half, 7, bufoff) HS_STORE1(half, 8, bufoff)        \
    }

    int pz = pix0;
    HS_LOAD(0, 0);
    for (int o = tid * 16; o < HS_SMEM; o += 256 * 16) *(uint4 *)(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // borders stay zero

    // ---- GEMM role
    f32x16 acc[NI][2];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const f16x8 *wl = p.wS + wn * 256 + lane;              // this wave's fragments of global step g: wl[g * 512 + {0, 64, 128, 192}]
    f16x8 Bq[HS_RING][4];                                  // [slot][nt0 hi, nt0 lo, nt1 hi, nt1 lo]
#define HS_LOADB(slot, g)                                                                       \
    {                                                                                           \
        const f16x8 *w_ = wl + (long)(g) * 512;                                                 \
        Bq[slot][0] = w_[0]; Bq[slot][1] = w_[64]; Bq[slot][2] = w_[128]; Bq[slot][3] = w_[192]; \
    }
#pragma unroll
    for (int s = 0; s < HS_AHEAD; ++s) HS_LOADB(s, s);
    __syncthreads();
    HS_STORE(0, 0);
    HS_LOAD(1, 0);
    HS_STORE(1, 0);
    __syncthreads();

    const unsigned lane_base = (unsigned)(q0 + 32 * wm + l31) * HS_LDP + 16 * h;
#define HS_LDS(off) (*(const f16x8 *)(smem + (off)))
#define HS_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// One (k16 step, tap): NI fenced regions, each = { the NEXT tile's two ds_reads; a share of the staging work; this tile's
// six MFMAs }.  The fences keep the reads one tile ahead of their use (the compiler otherwise sinks them to the MFMAs
// that consume them and the wave eats the LDS latency once per tile).  Staging of the next chunk: 9 global loads at
// steps 0 and 9, their split + LDS writes spread over the tiles of steps 7-8 and 16-17.
#define HS_SSTEPS ((9 + NI - 1) / NI)      /* steps over whose tiles one half's 9 staging items are spread */
#define HS_STEP(s, MORE)                                                                        \
        {                                                                                       \
            constexpr int ks = (s) / 9, tap = (s) - 9 * ks;                                     \
            constexpr int tap1 = ((s) + 1) % 9, ks1 = (((s) + 1) / 9) & 1;                      \
            const unsigned ab = rb + (unsigned)((tap / 3 - 1) * P + (tap % 3 - 1)) * HS_LDP + ks * 64;      \
            const unsigned ab1 = rb + (unsigned)((tap1 / 3 - 1) * P + (tap1 % 3 - 1)) * HS_LDP + ks1 * 64;  \
            _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                    \
                f16x8 nh = ah, nl = al;                                                         \
                if (i + 1 < NI) {                                                               \
                    nh = HS_LDS(ab + (i + 1) * (64 * HS_LDP)); nl = HS_LDS(ab + (i + 1) * (64 * HS_LDP) + 32); \
                } else if ((s) + 1 < HS_STEPS) {                                                \
                    nh = HS_LDS(ab1); nl = HS_LDS(ab1 + 32);                                    \
                }                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                              \
                if (i == 0) {                                                                   \
                    if (MORE || (s) + HS_AHEAD < HS_STEPS) { HS_LOADB(((s) + HS_AHEAD) % HS_RING, gnext + (s)); } \
                    if ((s) == 0 && MORE) { HS_LOAD(0, c + 1); }                                \
                    if ((s) == 9 && MORE) { HS_LOAD(1, c + 1); }                                \
                }                                                                               \
                if (MORE && ((s) % 9) >= 9 - HS_SSTEPS) {       /* item (slot - pad) of half s / 9, the last slot = item 8 */ \
                    const int j_ = (((s) % 9) - (9 - HS_SSTEPS)) * NI + i - (HS_SSTEPS * NI - 9);     \
                    if (j_ >= 0) { HS_STORE1((s) / 9, (j_ < 0 ? 0 : j_), wb); }                 \
                }                                                                               \
                HS_MFMA(ah, Bq[(s) % HS_RING][0], acc[i][0]);                                   \
                HS_MFMA(ah, Bq[(s) % HS_RING][2], acc[i][1]);                                   \
                HS_MFMA(ah, Bq[(s) % HS_RING][1], acc[i][0]);                                   \
                HS_MFMA(ah, Bq[(s) % HS_RING][3], acc[i][1]);                                   \
                HS_MFMA(al, Bq[(s) % HS_RING][0], acc[i][0]);                                   \
                HS_MFMA(al, Bq[(s) % HS_RING][2], acc[i][1]);                                   \
                ah = nh; al = nl;                                                               \
                __builtin_amdgcn_sched_barrier(0);                                              \
            }                                                                                   \
        }
#define HS_CHUNK(MORE)                                                                          \
    {                                                                                           \
        const unsigned rb = lane_base + (unsigned)(c & 1) * (HS_NPB * HS_LDP);                  \
        const unsigned wb = (unsigned)((c & 1) ^ 1) * (HS_NPB * HS_LDP);                        \
        asm volatile("" : "+v"(pz));     /* keeps the 18 staging addresses out of loop-carried registers: recomputed per chunk */ \
        const int gnext = c * HS_STEPS + HS_AHEAD;      /* global step the first prefetch of this chunk fetches */ \
        f16x8 ah = HS_LDS(rb + (unsigned)(-P - 1) * HS_LDP), al = HS_LDS(rb + (unsigned)(-P - 1) * HS_LDP + 32); \
        HS_STEP(0, MORE) HS_STEP(1, MORE) HS_STEP(2, MORE) HS_STEP(3, MORE) HS_STEP(4, MORE) HS_STEP(5, MORE)       \
        HS_STEP(6, MORE) HS_STEP(7, MORE) HS_STEP(8, MORE) HS_STEP(9, MORE) HS_STEP(10, MORE) HS_STEP(11, MORE)     \
        HS_STEP(12, MORE) HS_STEP(13, MORE) HS_STEP(14, MORE) HS_STEP(15, MORE) HS_STEP(16, MORE) HS_STEP(17, MORE) \
        __syncthreads();                                                                        \
    }
    int c = 0;
#pragma unroll 1
    for (; c < HS_C / 32 - 1; ++c) HS_CHUNK(true)
    HS_CHUNK(false)                    // the last chunk stages nothing
#undef HS_STEP
#undef HS_SSTEPS
#undef HS_CHUNK
#undef HS_LOAD
#undef HS_STORE
#undef HS_STORE1
#undef HS_LOADB
#undef HS_LDS
#undef HS_MFMA

    // ---- epilogue: 256 GEMM rows at a time through LDS, so that every thread handles float4 pieces of whole pixel rows
    const float winv = *p.wscale_inv;
    float sc[2], sh[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        sc[nt] = p.scale[64 * wn + 32 * nt + l31] * winv;
        sh[nt] = p.shift[64 * wn + 32 * nt + l31];
    }
    float *Ms = (float *)smem;
    const float invP = 1.0f / (float)P;
    const int cq = tid & 31, rr0 = tid >> 5;
    const long obase = ((long)img * p.Hd + y0) * p.Wd * HS_C + 4 * cq;
#pragma unroll
    for (int pass = 0; pass < NI / 4; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ms[(32 * (2 * ii + wm) + (r & 3) + 8 * (r >> 2) + 4 * h) * HS_MLD + 64 * wn + 32 * nt + l31] =
                        acc[4 * pass + ii][nt][r] * sc[nt] + sh[nt];
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const int lrow = rr0 + 8 * j;
            const int q = q0 + 256 * pass + lrow;
            const int py = (int)(((float)q + 0.5f) * invP), px = q - py * P;
            if (px >= 1 && py <= rows) {
                float4 v = *(const float4 *)&Ms[lrow * HS_MLD + 4 * cq];
                const long off = obase + ((long)(py - 1) * p.Wd + (px - 1)) * HS_C;
                if (p.res) {
                    const float4 rv = *(const float4 *)(p.res + off);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *(float4 *)(p.out + off) = v;
            }
        }
        if (pass + 1 < NI / 4) __syncthreads();
    }
}

// max |w| of the layer -> k with 256 <= max * 2^k < 512; writes 2^k and 2^-k behind the fragment image
__global__ __launch_bounds__(1024) void k_f16s_wscale(const float *__restrict__ w, float *__restrict__ tail)
{
    __shared__ float red[1024];
    float m = 0.f;
    for (int i = threadIdx.x; i < 9 * HS_C * HS_C; i += 1024) m = fmaxf(m, fabsf(w[i]));
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
    }
}

// Keras kernel (kh, kw, cin, cout) float32 -> split f16 fragments in the order the conv kernel's waves load them
__global__ void k_f16s_weights(const float *__restrict__ w, _Float16 *__restrict__ wS, const float *__restrict__ tail)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;          // one f16x8 fragment piece
    if (v >= HS_WS_ELEMS / 8) return;
    const float mul = tail[1];
    const int g = v / 512, rem = v - g * 512;
    const int c = g / HS_STEPS, s = g - c * HS_STEPS, ks = s / 9, tap = s - 9 * ks;
    const int wn = rem >> 8, nt = (rem >> 7) & 1, hl = (rem >> 6) & 1, lane = rem & 63;
    const int h = lane >> 5, l31 = lane & 31;
    const int cout = 64 * wn + 32 * nt + l31;
    for (int j = 0; j < 8; ++j) {
        const int cin = 32 * c + 16 * ks + 8 * h + j;
        const float val = w[(long)(tap * HS_C + cin) * HS_C + cout] * mul;
        const _Float16 hi = (_Float16)val;
        wS[(long)v * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
    }
}

extern "C" int snk_conv3x3_prepare_weights_f16s(const float *d_w_hwio, void *d_wS, void *stream)
{
    SNK_REQUIRE(d_w_hwio && d_wS, "snk_conv3x3_prepare_weights_f16s: NULL argument");
    float *tail = (float *)((_Float16 *)d_wS + HS_WS_ELEMS);
    k_f16s_wscale<<<1, 1024, 0, (hipStream_t)stream>>>(d_w_hwio, tail);
    k_f16s_weights<<<(HS_WS_ELEMS / 8 + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_w_hwio, (_Float16 *)d_wS, tail);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_conv3x3_bn_f16s(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                                   const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                                   void *stream)
{
    SNK_REQUIRE(d_x && d_wS && d_scale && d_shift && d_out, "snk_conv3x3_bn_f16s: NULL argument");
    SNK_REQUIRE(n_images >= 0 && height >= 1 && width >= 1, "snk_conv3x3_bn_f16s: bad shape %d x %d x %d", n_images, height, width);
    if (n_images == 0) return 0;
    const int P = width + 1;
    // rows per strip: (R - 1) P + W GEMM rows <= 512, (R + 2) P + 1 staged pixels <= HS_NPB, junk rows stay inside the slack
    int Rmax = min((512 - width) / P + 1, (HS_NPB - 1) / P - 2);
    SNK_REQUIRE(Rmax >= 1 && 2 * P + 513 <= HS_NPB + HS_SLACK - 1,
                "snk_conv3x3_bn_f16s: observation width %d not supported (max 38)", width);
    const int n_strips = (height + Rmax - 1) / Rmax;
    const int R = (height + n_strips - 1) / n_strips;
    const int gemm_rows = (R - 1) * P + width;
    SNK_REQUIRE((long)n_images * n_strips < (1l << 31), "snk_conv3x3_bn_f16s: batch too large");
    ConvHsArgs a = {d_x, (const f16x8 *)d_wS, (const float *)((const _Float16 *)d_wS + HS_WS_ELEMS), d_scale, d_shift,
                    d_residual, d_out, height, width, R, n_strips, relu};
    if (gemm_rows > 256) k_conv3x3_f16s<8><<<n_images * n_strips, 256, 0, (hipStream_t)stream>>>(a);
    else k_conv3x3_f16s<4><<<n_images * n_strips, 256, 0, (hipStream_t)stream>>>(a);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
