// Micro-experiment: the gfx950 "wide VMEM store, then a VALU write of its data registers" hazard, measured.
//   hipcc -O2 --offload-arch=gfx950 -o store_hazard store_hazard.hip && ./store_hazard > store_hazard.json
//
// What LLVM's hazard recognizer believes (GCNHazardRecognizer::createsVALUHazard): a MUBUF store of more than 64 data bits needs
// ONE wait state before a VALU instruction rewrites its data VGPRs -- and only when the store has NO SGPR soffset.  Round 5 saw
// wrong rows in HBM from exactly the form the compiler leaves unguarded (conv_split.hip's input-gradient epilogue).  This
// program states the rule from measurements: every kernel below is ONE hand-placed instruction sequence
//     <data registers := OLD> ; STORE ; GAP ; OVERWRITE (data register(s) := NEW)
// written as a single asm block with fixed registers (the compiler schedules nothing inside it), eight rows per thread
// alternating between two register sets, and a checker kernel classes every stored dword as OLD (right), NEW (the hazard) or
// OTHER.  Dimensions:
//     STORE      buffer x4 with SGPR soffset | buffer x4 soffset 0 | buffer x3 SGPR | buffer x2 SGPR | global x4 saddr | global x4 vaddr
//     GAP        nothing | 1..4 independent VALU | s_nop 0 / 1 / 2
//     OVERWRITE  v_mov_b32 of dword 0 / 1 / 2 / 3 | v_pk_mul_f32 of dwords 0:1 / 2:3
// Also: whether a raw buffer's range check sees the SGPR soffset (LLVM documents it as excluded).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct P { float *out; unsigned bytes; unsigned rowb; };

__device__ inline i32x4 make_rsrc(const void *ptr, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// ---- the sequence's three parts, as string pieces over a register set (R0..R3 are decimal register numbers) -----------------
#define S_(x) #x
#define V1(r) "v" S_(r)
#define VR(a, b) "v[" S_(a) ":" S_(b) "]"

#define ST_bsg4(R0, R1, R2, R3) "buffer_store_dwordx4 " VR(R0, R3) ", %[vo], %[rs], %[so] offen\n"
#define ST_bim4(R0, R1, R2, R3) "buffer_store_dwordx4 " VR(R0, R3) ", %[vo2], %[rs], 0 offen\n"
#define ST_bsg3(R0, R1, R2, R3) "buffer_store_dwordx3 " VR(R0, R2) ", %[vo], %[rs], %[so] offen\n"
#define ST_bsg2(R0, R1, R2, R3) "buffer_store_dwordx2 " VR(R0, R1) ", %[vo], %[rs], %[so] offen\n"
#define ST_gsa4(R0, R1, R2, R3) "global_store_dwordx4 %[vo2], " VR(R0, R3) ", %[sa]\n"
#define ST_gva4(R0, R1, R2, R3) "global_store_dwordx4 %[va], " VR(R0, R3) ", off\n"
// LDS stores (round 6, for completeness: no hazard is documented for them; a ds_write moves its data registers to the LDS over
// several cycles too).  The row goes to LDS and is copied out to the checker's array by ordinary code afterwards.
#define ST_ldw4(R0, R1, R2, R3) "ds_write_b128 %[la], " VR(R0, R3) "\n"
#define ST_ldw3(R0, R1, R2, R3) "ds_write_b96 %[la], " VR(R0, R2) "\n"
#define ST_ldw2(R0, R1, R2, R3) "ds_write_b64 %[la], " VR(R0, R1) "\n"
#define IS_LDS_bsg4 0
#define IS_LDS_bim4 0
#define IS_LDS_bsg3 0
#define IS_LDS_bsg2 0
#define IS_LDS_gsa4 0
#define IS_LDS_gva4 0
#define IS_LDS_ldw4 1
#define IS_LDS_ldw3 1
#define IS_LDS_ldw2 1

#define GAP_d0 ""
#define GAP_v1 "v_mov_b32 v60, v61\n"
#define GAP_v2 GAP_v1 "v_mov_b32 v61, v62\n"
#define GAP_v3 GAP_v2 "v_mov_b32 v62, v63\n"
#define GAP_v4 GAP_v3 "v_mov_b32 v63, v60\n"
#define GAP_n0 "s_nop 0\n"
#define GAP_n1 "s_nop 1\n"
#define GAP_n2 "s_nop 2\n"

#define OV_m0(R0, R1, R2, R3) "v_mov_b32 " V1(R0) ", %[n0]\n"
#define OV_m1(R0, R1, R2, R3) "v_mov_b32 " V1(R1) ", %[n1]\n"
#define OV_m2(R0, R1, R2, R3) "v_mov_b32 " V1(R2) ", %[n2]\n"
#define OV_m3(R0, R1, R2, R3) "v_mov_b32 " V1(R3) ", %[n3]\n"
#define OV_p01(R0, R1, R2, R3) "v_pk_mul_f32 " VR(R0, R1) ", %[np01], %[one2]\n"
#define OV_p23(R0, R1, R2, R3) "v_pk_mul_f32 " VR(R2, R3) ", %[np23], %[one2]\n"

#define ROW_ASM(ST, GAP, OV, R0, R1, R2, R3)                                                                                     \
    asm volatile("v_mov_b32 " V1(R0) ", %[o0]\n v_mov_b32 " V1(R1) ", %[o1]\n v_mov_b32 " V1(R2) ", %[o2]\n v_mov_b32 " V1(R3)   \
                 ", %[o3]\n s_nop 1\n" ST_##ST(R0, R1, R2, R3) GAP_##GAP OV_##OV(R0, R1, R2, R3)                                   \
                 :                                                                                                                \
                 : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [n3] "v"(n3), \
                   [np01] "v"(np01), [np23] "v"(np23), [one2] "v"(one2), [vo] "v"(vo), [vo2] "v"(vo2), [rs] "s"(rs),               \
                   [so] "s"(so), [sa] "s"(sa), [va] "v"(va), [la] "v"(la)                                                          \
                 : "memory", "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63")

#define ROWS 8
#define OLDV(t, r, k) ((float)((((t) * ROWS + (r)) * 4 + (k)) & 0xffffff))
#define NEWV(t, r, k) (-(OLDV(t, r, k) + 1.0f))

#define DEFK(ST, GAP, OV)                                                                                                        \
    __global__ __launch_bounds__(256) void k_##ST##_##GAP##_##OV(P p)                                                           \
    {                                                                                                                             \
        __shared__ float4 lds_rows[IS_LDS_##ST ? 256 * ROWS : 1];          /* the block's only LDS: offset 0 */                    \
        const unsigned t = blockIdx.x * 256 + threadIdx.x;                                                                        \
        const i32x4 rs = make_rsrc(p.out, p.bytes);                                                                               \
        const unsigned long long sa = (unsigned long long)p.out;                                                                  \
        const unsigned vo = t * 16u;                                                                                              \
        const f32x2 one2 = {1.0f, 1.0f};                                                                                          \
        _Pragma("unroll") for (int r = 0; r < ROWS; ++r)                                                                          \
        {                                                                                                                         \
            const unsigned so = (unsigned)r * p.rowb;                                                                             \
            const unsigned vo2 = vo + so;                                                                                         \
            const unsigned la = ((unsigned)r * 256u + threadIdx.x) * 16u;                                                         \
            const unsigned long long va = sa + vo2;                                                                               \
            const float o0 = OLDV(t, r, 0), o1 = OLDV(t, r, 1), o2 = OLDV(t, r, 2), o3 = OLDV(t, r, 3);                           \
            const float n0 = NEWV(t, r, 0), n1 = NEWV(t, r, 1), n2 = NEWV(t, r, 2), n3 = NEWV(t, r, 3);                           \
            const f32x2 np01 = {n0, n1}, np23 = {n2, n3};                                                                         \
            if (r & 1) ROW_ASM(ST, GAP, OV, 48, 49, 50, 51);                                                                      \
            else ROW_ASM(ST, GAP, OV, 40, 41, 42, 43);                                                                            \
        }                                                                                                                         \
        if (IS_LDS_##ST) {                                                                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        /* the compiler does not count the asm blocks' ds_writes */   \
            __syncthreads();                                                                                                      \
            for (int r = 0; r < ROWS; ++r)                                                                                        \
                *(float4 *)((char *)p.out + (size_t)r * p.rowb + (size_t)t * 16) = lds_rows[r * 256 + threadIdx.x];               \
        }                                                                                                                         \
    }

#define FOR_OV(ST, GAP) DEFK(ST, GAP, m0) DEFK(ST, GAP, m1) DEFK(ST, GAP, m2) DEFK(ST, GAP, m3) DEFK(ST, GAP, p01) DEFK(ST, GAP, p23)
#define FOR_GAP(ST) FOR_OV(ST, d0) FOR_OV(ST, v1) FOR_OV(ST, v2) FOR_OV(ST, v3) FOR_OV(ST, v4) FOR_OV(ST, n0) FOR_OV(ST, n1) FOR_OV(ST, n2)
FOR_GAP(bsg4) FOR_GAP(bim4) FOR_GAP(bsg3) FOR_GAP(bsg2) FOR_GAP(gsa4) FOR_GAP(gva4) FOR_GAP(ldw4) FOR_GAP(ldw3) FOR_GAP(ldw2)

// ---- checker: class every dword; hist[k][lane & 15][class]  (class 0 = OLD, 1 = NEW, 2 = other) --------------------------------
__global__ void k_check(const float *out, unsigned nthreads, unsigned rowb, int width, unsigned *hist)
{
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    for (int r = 0; r < ROWS; ++r)
        for (int k = 0; k < width; ++k) {
            const float v = out[((size_t)r * rowb + (size_t)t * 16) / 4 + k];
            const int c = v == OLDV(t, r, k) ? 0 : v == NEWV(t, r, k) ? 1 : 2;
            if (c) atomicAdd(&hist[(k * 16 + (t & 15)) * 3 + c], 1u);
        }
}

// ---- range check and the SGPR soffset ----------------------------------------------------------------------------------------
// buffer of `bytes` records; the memory behind it holds a sentinel.  Each lane reads / writes dword `lane` THROUGH soffset = bytes
// (so the address lies wholly behind the descriptor's range while voffset alone is in range), and the same address through
// voffset alone.
__global__ void k_range(float *buf, unsigned bytes, unsigned *res)
{
    const i32x4 rs = make_rsrc(buf, bytes);
    const unsigned vo = threadIdx.x * 4u, so = bytes, vo_far = vo + bytes;
    unsigned via_s, via_v;
    asm volatile("buffer_load_dword %0, %2, %3, %4 offen\n buffer_load_dword %1, %5, %3, 0 offen\n s_waitcnt vmcnt(0)\n"
                 : "=&v"(via_s), "=&v"(via_v) : "v"(vo), "s"(rs), "s"(so), "v"(vo_far) : "memory");
    res[threadIdx.x] = via_s;
    res[64 + threadIdx.x] = via_v;
    const unsigned mark = 0x77770000u + threadIdx.x;
    const unsigned so2 = bytes + 256u, vo_far2 = vo + bytes + 512u;
    asm volatile("buffer_store_dword %0, %1, %2, %3 offen\n buffer_store_dword %0, %4, %2, 0 offen\n s_waitcnt vmcnt(0)\n"
                 : : "v"(mark), "v"(vo), "s"(rs), "s"(so2), "v"(vo_far2) : "memory");
}

// ---- a neighbour that keeps a compute unit's other pipes busy while the sequences run (second pass of main: "stress") ------------
// one workgroup per compute unit on a stream of its own: MFMAs back to back, LDS reads and writes, global loads and 16-byte global
// stores, for `iters` rounds -- the sequences above then issue their stores into busy vector-memory and LDS paths
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_stress(float4 *scratch, int iters)
{
    __shared__ float4 lds[1024];
    const int t = threadIdx.x;
    f32x16 acc = {0};
    f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    float4 v = make_float4(t, 1.f, 2.f, 3.f);
    float4 *mine = scratch + (size_t)blockIdx.x * 4096 + t;
    for (int i = 0; i < iters; ++i) {
        lds[(t * 5 + i) & 1023] = v;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
        const float4 u = lds[(t * 7 + 3 * i) & 1023];
        const float4 g = mine[(i & 7) * 256];
        v.x += u.y + g.z; v.y += acc[i & 15];
        mine[((i + 3) & 7) * 256] = v;
    }
    if (v.x == 123.456f) scratch[0] = v;
}

struct Variant { const char *st, *gap, *ov; void (*fn)(P); int width; };
#define VAR(ST, GAP, OV, W) {#ST, #GAP, #OV, k_##ST##_##GAP##_##OV, W},
#define VAR_OV(ST, GAP, W) VAR(ST, GAP, m0, W) VAR(ST, GAP, m1, W) VAR(ST, GAP, m2, W) VAR(ST, GAP, m3, W) VAR(ST, GAP, p01, W) VAR(ST, GAP, p23, W)
#define VAR_GAP(ST, W) VAR_OV(ST, d0, W) VAR_OV(ST, v1, W) VAR_OV(ST, v2, W) VAR_OV(ST, v3, W) VAR_OV(ST, v4, W) VAR_OV(ST, n0, W) VAR_OV(ST, n1, W) VAR_OV(ST, n2, W)
static const Variant variants[] = {VAR_GAP(bsg4, 4) VAR_GAP(bim4, 4) VAR_GAP(bsg3, 3) VAR_GAP(bsg2, 2) VAR_GAP(gsa4, 4) VAR_GAP(gva4, 4)
                                   VAR_GAP(ldw4, 4) VAR_GAP(ldw3, 3) VAR_GAP(ldw2, 2)};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const int stress = argc > 2 ? atoi(argv[2]) : 0;             // 1: every launch runs beside k_stress on another stream
    const unsigned grids[2] = {8, 2048};                         // a nearly idle chip, and every CU's store path busy
    hipStream_t s_other;
    float4 *scratch;
    CK(hipStreamCreateWithFlags(&s_other, hipStreamNonBlocking));
    CK(hipMalloc(&scratch, (size_t)256 * 4096 * sizeof(float4)));
    CK(hipMemset(scratch, 0, (size_t)256 * 4096 * sizeof(float4)));
    float *out;
    unsigned *hist;
    const size_t max_bytes = (size_t)2048 * 256 * 16 * ROWS;
    CK(hipMalloc(&out, max_bytes));
    CK(hipMalloc(&hist, 4 * 16 * 3 * sizeof(unsigned)));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"reps\": %d, \"rows_per_thread\": %d, \"stress\": %d,\n \"variants\": [\n", prop.gcnArchName, reps, ROWS, stress);
    bool first = true;
    for (const Variant &v : variants)
        for (unsigned grid : grids) {
            const unsigned nthreads = grid * 256, rowb = nthreads * 16;
            unsigned h[4 * 16 * 3];
            CK(hipMemset(hist, 0, sizeof(h)));
            for (int rep = 0; rep < reps; ++rep) {
                CK(hipMemset(out, 0xff, (size_t)rowb * ROWS));
                P p{out, rowb * ROWS, rowb};
                if (stress) hipLaunchKernelGGL(k_stress, dim3(256), dim3(256), 0, s_other, scratch, 600);      // ~100 us beside it
                hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, p);
                hipLaunchKernelGGL(k_check, dim3(grid), dim3(256), 0, 0, out, nthreads, rowb, v.width, hist);
            }
            CK(hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost));
            unsigned long long n_new[4] = {0, 0, 0, 0}, n_other[4] = {0, 0, 0, 0};
            unsigned lanes_new = 0;
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 16; ++l) {
                    n_new[k] += h[(k * 16 + l) * 3 + 1];
                    n_other[k] += h[(k * 16 + l) * 3 + 2];
                    if (h[(k * 16 + l) * 3 + 1]) lanes_new |= 1u << l;
                }
            printf("%s  {\"store\": \"%s\", \"gap\": \"%s\", \"overwrite\": \"%s\", \"grid\": %u, \"stored_per_dword\": %llu, "
                   "\"new\": [%llu, %llu, %llu, %llu], \"other\": [%llu, %llu, %llu, %llu], \"lanes_mod16_with_new\": \"0x%04x\"}",
                   first ? "" : ",\n", v.st, v.gap, v.ov, grid, (unsigned long long)nthreads * ROWS * reps, n_new[0], n_new[1], n_new[2],
                   n_new[3], n_other[0], n_other[1], n_other[2], n_other[3], lanes_new);
            first = false;
        }
    printf("\n ],\n");
    // range check
    {
        const unsigned bytes = 4096;
        float *buf;
        unsigned *res;
        CK(hipMalloc(&buf, bytes + 4096));
        CK(hipMalloc(&res, 128 * sizeof(unsigned)));
        std::vector<unsigned> host((bytes + 4096) / 4, 0x5a5a5a5au);
        CK(hipMemcpy(buf, host.data(), bytes + 4096, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_range, dim3(1), dim3(64), 0, 0, buf, bytes, res);
        unsigned r[128];
        CK(hipMemcpy(r, res, sizeof(r), hipMemcpyDeviceToHost));
        CK(hipMemcpy(host.data(), buf, bytes + 4096, hipMemcpyDeviceToHost));
        int ld_s_zero = 0, ld_s_sent = 0, ld_v_zero = 0, ld_v_sent = 0, st_s_landed = 0, st_v_landed = 0;
        for (int l = 0; l < 64; ++l) {
            ld_s_zero += r[l] == 0; ld_s_sent += r[l] == 0x5a5a5a5au;
            ld_v_zero += r[64 + l] == 0; ld_v_sent += r[64 + l] == 0x5a5a5a5au;
            st_s_landed += host[(bytes + 256) / 4 + l] == 0x77770000u + l;
            st_v_landed += host[(bytes + 512) / 4 + l] == 0x77770000u + l;
        }
        printf(" \"range_check\": {\"num_records\": %u, \"load_past_range_via_sgpr_soffset\": {\"lanes_zero\": %d, \"lanes_memory\": %d}, "
               "\"load_past_range_via_voffset\": {\"lanes_zero\": %d, \"lanes_memory\": %d}, "
               "\"store_past_range_via_sgpr_soffset_lanes_landed\": %d, \"store_past_range_via_voffset_lanes_landed\": %d}\n}\n",
               bytes, ld_s_zero, ld_s_sent, ld_v_zero, ld_v_sent, st_s_landed, st_v_landed);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
