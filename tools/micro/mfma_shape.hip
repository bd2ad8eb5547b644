// Microbenchmark: f16 MFMA throughput under the chip's clock give-back, 32x32x16 vs 16x16x32, operands re-read from
// LDS with ds_read_b128 in the ratio the split-f16 conv kernel uses (2 reads per 3 MFMAs / 3 reads per 6 MFMAs).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const f16x8 *src, float *out, int iters)
{
    __shared__ __align__(16) unsigned char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) ((f16x8 *)smem)[i] = src[i];
    f16x8 b0 = src[tid], b1 = src[tid + 256], b2 = src[tid + 512], b3 = src[tid + 768];
    __syncthreads();
    const unsigned base = (lane & 31) * 80 + (lane >> 5) * 16;
    if (SHAPE == 32) {
        f32x16 acc[7];
        for (int a = 0; a < 7; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const f16x8 ah = *(const f16x8 *)(smem + base + t * 2560 + (it & 7) * 160);
                const f16x8 al = *(const f16x8 *)(smem + base + t * 2560 + (it & 7) * 160 + 32);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0, acc[t], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int a = 0; a < 7; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 acc[14][2];
        for (int a = 0; a < 14; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        const unsigned base16 = (lane & 15) * 80 + (lane >> 4) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 14; ++t) {      // per 16-row tile and tap pair: 3 reads, 6 MFMAs (2 N tiles of 16)
                const f16x8 a0 = *(const f16x8 *)(smem + base16 + t * 1280 + (it & 7) * 160);
                const f16x8 a1 = *(const f16x8 *)(smem + base16 + t * 1280 + (it & 7) * 160 + 80);
                const f16x8 a2 = *(const f16x8 *)(smem + base16 + t * 1280 + (it & 7) * 160 + 160);
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[t][1], 0, 0, 0);
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b3, acc[t][1], 0, 0, 0);
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, acc[t][1], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int a = 0; a < 14; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
        out[blockIdx.x * 256 + tid] = s;
    }
}

int main()
{
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    f16x8 *d; float *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 512 * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 30; ++w) {
                if (shape == 32) k<32><<<512, 256>>>(d, o, iters); else k<16><<<512, 256>>>(d, o, iters);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 20; ++w) {
                if (shape == 32) k<32><<<512, 256>>>(d, o, iters); else k<16><<<512, 256>>>(d, o, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 20.0 * 512 * 4 * iters * (shape == 32 ? 7 * 3 * 32768.0 : 14 * 6 * 16384.0);
            printf("shape %s: %.3f ms per launch, %.0f TFLOP/s executed\n", shape == 32 ? "32x32x16" : "16x16x32", ms / 20, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
