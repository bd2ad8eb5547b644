// Microbenchmark (round 4): can the chip keep the split-f16 MFMA rate when the weight fragments stream from L2 at the ratio a
// 1-D Winograd F(2,3) form of the tower layer would need?  Two loop skeletons, the same LDS-read : MFMA ratio (2 : 3):
//   D  the direct kernel's proportions: 256 threads, 2 blocks per CU, per 16-channel chunk 9 B-fragment pairs (hi, lo) from
//      global memory (a 590 KB L2-resident image, fragment order), each used for 7 M tiles x 3 MFMAs
//   W  the Winograd proportions: 512 threads, 1 block per CU, per chunk 6 B pairs (of a 786 KB image), each used for 4 M tiles
//      x 3 MFMAs (twice the B loads per MFMA), 8 accumulator tiles per wave
//   hipcc -O3 --offload-arch=gfx950 -o wino1d_loop wino1d_loop.hip && ./wino1d_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int NT, int NB, int THREADS, int BPC, int RING = 3>      // M tiles per B fragment, B fragments per chunk, block size, blocks per CU, B-fragment ring (prefetch RING - 1 steps ahead)
__global__ __launch_bounds__(THREADS, BPC) void k(const f16x8 *src, const f16x8 *wS, float *out, int passes)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 3584; i += THREADS) ((f16x8 *)smem)[i] = src[i & 4095];
    __syncthreads();
    const unsigned base = (lane & 31) * 80 + (lane >> 5) * 16;
    const int n_w = THREADS / 64;
    constexpr int NA = NB == 9 ? NT : 8;                  // the Winograd forms keep 8 accumulator tiles (positions x M tiles) per wave
    f32x16 acc[NA];
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const f16x8 *wl = wS + w * 128 + lane;                 // [step][wave][hi/lo][lane]
    const long step_stride = (long)n_w * 128;
    for (int pass = 0; pass < passes; ++pass) {
        f16x8 B[RING][2];
#pragma unroll
        for (int q = 0; q < RING - 1; ++q) { B[q][0] = wl[q * step_stride]; B[q][1] = wl[q * step_stride + 64]; }
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            // the real kernel's pipeline: A fragments are read two tiles ahead of their MFMAs, fenced so that the compiler keeps them there
#define AOFF(tt) (base + (((tt) % NT) * 2560 + ((((tt) / NT) * 5 + c) & 7) * 160))
            f16x8 a0h = *(const f16x8 *)(smem + AOFF(0)), a0l = *(const f16x8 *)(smem + AOFF(0) + 32);
            f16x8 a1h = *(const f16x8 *)(smem + AOFF(1)), a1l = *(const f16x8 *)(smem + AOFF(1) + 32);
#pragma unroll
            for (int s = 0; s < NB; ++s) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int t2 = s * NT + t + 2;
                    f16x8 nh = a1h, nl = a1l;
                    if (t2 < NB * NT) { nh = *(const f16x8 *)(smem + AOFF(t2)); nl = *(const f16x8 *)(smem + AOFF(t2) + 32); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t == 0) {
                        const f16x8 *wn_ = wl + (long)((c * NB + s + RING - 1) % (8 * NB)) * step_stride;
                        B[(s + RING - 1) % RING][0] = wn_[0]; B[(s + RING - 1) % RING][1] = wn_[64];
                    }
                    const int ai = NB == 9 ? t : (s % (NA / NT)) * NT + t;
                    MFMA(a0h, B[s % RING][0], acc[ai]);
                    MFMA(a0h, B[s % RING][1], acc[ai]);
                    MFMA(a0l, B[s % RING][0], acc[ai]);
                    a0h = a1h; a0l = a1l; a1h = nh; a1l = nl;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float sum = 0.f;
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * THREADS + tid] = sum;
}

int main()
{
    std::vector<_Float16> h(4096 * 8), hw(786432 / 2 + 4096);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    for (auto &v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    f16x8 *d, *dw; float *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&o, 1024 * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 40;
    for (int rep = 0; rep < 2; ++rep)
        for (int var = 0; var < 5; ++var) {
            auto launch = [&]() {
                if (var == 0) k<7, 9, 256, 2><<<2048, 256, 3584 * 16>>>(d, dw, o, passes);
                else if (var == 1) k<4, 6, 512, 1><<<1024, 512, 3584 * 16>>>(d, dw, o, passes);
                else if (var == 2) k<2, 12, 256, 2><<<2048, 256, 3584 * 16>>>(d, dw, o, passes);      // 4-wave Winograd block: every wave all 4 positions x 2 M tiles
                else if (var == 3) k<2, 12, 256, 2, 4><<<2048, 256, 3584 * 16>>>(d, dw, o, passes);
                else k<2, 12, 256, 2, 6><<<2048, 256, 3584 * 16>>>(d, dw, o, passes);
            };
            for (int i = 0; i < 10; ++i) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = var == 0 ? 2048.0 * 4 * passes * 8 * 9 * 7 * 3 : var == 1 ? 1024.0 * 8 * passes * 8 * 6 * 4 * 3 : 2048.0 * 4 * passes * 8 * 12 * 2 * 3;
            const char *names[5] = {"D direct proportions (4 waves x 2 blocks, B x7)", "W winograd (8 waves x 1 block, B x4)", "W4 winograd (4 waves x 2 blocks, B x2), ring 3",
                                    "W4 ring 4", "W4 ring 6"};
            printf("%s: %.3f ms per launch, %.0f TFLOP/s executed\n", names[var], ms / 10, 10 * mf * 32768.0 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
