// common.h -- shared declarations for the gfx950 self-play engine (libsnake_engine.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "snake_engine.h"

// ------------------------------------------------------------------------------------------
// HBM layout of one game ("slot").  Everything a tick needs is one contiguous, 16-byte aligned
// record, so one wavefront streams it in with 16-byte loads (37 lanes x 16 B for 11x11/4):
//
//   ring[S][CAP]   cell_t   snake bodies as ring buffers of board-cell indices; entry
//                           (tail + k) & (CAP-1), k = 0..len-1, walks tail -> head
//   meta[S]        8 B      { u16 tail, u16 len, i16 health, u8 dir, u8 alive }
//   food[FW]       u64      food bit-plane, bit c = cell c
//   counters[6]    u32      wall, body, head, starvation, food_eaten, game_length
//   rewards[S]     i8       0 none / +1 / -1  (padded to 4 bytes)
//   uid            u32      identity of the game for the counter-based RNG
//
//   11x11/4 : CAP 128 (u8)  -> 512 + 32 + 16 + 24 + 4 + 4 = 592 B
//   7x7/2   : CAP  64 (u8)  -> 128 + 16 +  8 + 24 + 4 + 4 = 184 -> 192 B
//   19x19/8 : CAP 512 (u16) -> 8192 + 64 + 48 + 24 + 8 + 4 = 8340 -> 8352 B
// ------------------------------------------------------------------------------------------
struct SnakeMeta {
    uint16_t tail;
    uint16_t len;
    int16_t health;
    uint8_t dir;
    uint8_t alive;
};
static_assert(sizeof(SnakeMeta) == 8, "meta is 8 bytes");

struct Layout {
    int H, W, S, NC;
    int cap, cap_mask, cell_bytes, ring_bytes;
    int meta_off, food_off, cnt_off, rew_off, uid_off, stride;
    int FW;        // 64-bit words of the food plane
    int nc_pad;    // NC rounded up to 16
};

static inline int next_pow2_int(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

static inline Layout make_layout(int H, int W, int S)
{
    Layout L;
    L.H = H; L.W = W; L.S = S; L.NC = H * W;
    L.cap = next_pow2_int(L.NC + 2);
    L.cap_mask = L.cap - 1;
    L.cell_bytes = (L.NC <= 255) ? 1 : 2;
    L.ring_bytes = L.cap * L.cell_bytes;
    L.FW = (L.NC + 63) / 64;
    L.meta_off = S * L.ring_bytes;
    L.food_off = L.meta_off + S * 8;
    L.cnt_off = L.food_off + L.FW * 8;
    L.rew_off = L.cnt_off + 24;
    L.uid_off = L.rew_off + align_up(S, 4);
    L.stride = align_up(L.uid_off + 4, 16);
    L.nc_pad = align_up(L.NC, 16);
    return L;
}

struct snk_engine {
    Layout L;
    int n_slots;
    int health_dec;
    double food_chance;
    uint64_t seed;
    uint32_t next_uid;
    int device;
    uint8_t *d_state;
    unsigned long long *d_scratch64;   // 8 x u64 scratch (counter sums)
};

void snk_set_error(const char *fmt, ...);

#define SNK_CHECK_HIP(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            snk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                      \
        }                                                                                   \
    } while (0)

#define SNK_REQUIRE(cond, ...)                                                              \
    do {                                                                                    \
        if (!(cond)) { snk_set_error(__VA_ARGS__); return -1; }                             \
    } while (0)

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based; no per-game RNG state in HBM)
// ------------------------------------------------------------------------------------------
__host__ __device__ static inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                   uint32_t k0, uint32_t k1, uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ static inline uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// wave-wide (64 lanes) sum of a 64-bit value, result in every lane
__device__ static inline uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t lo = __shfl_xor((int)(uint32_t)v, off, 64);
        uint32_t hi = __shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// sum of a 64-bit value over an aligned group of GL lanes (16 or 64), result in every lane of the group
template <int GL>
__device__ static inline uint64_t group_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = GL / 2; off >= 1; off >>= 1) {
        uint32_t lo = __shfl_xor((int)(uint32_t)v, off, 64);
        uint32_t hi = __shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
