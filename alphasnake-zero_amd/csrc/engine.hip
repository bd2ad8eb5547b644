// engine.hip -- gfx950 kernels + C ABI for the batched Battlesnake engine.
//
// A lane group owns one game (or one observation) for the duration of a kernel: the game's HBM record (common.h) is
// streamed into LDS with 16-byte loads, the rules are evaluated with one lane per snake, body occupancy is rebuilt in LDS
// from the rings, and the record is streamed back.  The tick: k_step_quad -- a quad per game, sixteen games per wavefront,
// cross-snake questions as DPP quad_perm moves, occupancy as bit planes -- for up to 4 snakes on up to 255 cells; k_step --
// 16 or 64 lanes per game, group-confined shuffles / ballots, byte planes -- for everything else.  All lanes of a group sit
// in one wavefront, so the phases of a kernel are ordered by GAME_SYNC (no workgroup barrier).
// Reference semantics: Game.tic game.py:87-205, Game.make_state game.py:215-257,
// Game.__init__ game.py:13-61, Game.subgame game.py:266-276 (see the per-kernel comments).
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <vector>

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void snk_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *snk_last_error(void) { return g_err; }
extern "C" int snk_version(void) { return SNK_ABI_VERSION; }

#define WAVES_PER_BLOCK 4
#define BLOCK_THREADS (WAVES_PER_BLOCK * 64)

// Kernels are templated on the board side: 11x11, 7x7 and 19x19 (the BASELINE configs) get compile-time geometry; any
// other square board runs the same code with the geometry read from the Layout: <0, 0> for boards of at most 255 cells
// (8-bit ring entries), <-1, -1> for larger ones (16-bit entries; CellT<(-1) * (-1)> is the primary template).
template <int NC> struct CellT { using type = uint16_t; };
template <> struct CellT<121> { using type = uint8_t; };
template <> struct CellT<49> { using type = uint8_t; };
template <> struct CellT<0> { using type = uint8_t; };
#define BOARD_DIMS(L)                                                                                   \
    const int HH = H > 0 ? H : (L).H, WW = W > 0 ? W : (L).W;                                           \
    const int NC = HH * WW;                                                                             \
    constexpr int MAXFW = H > 0 ? (H * W + 63) / 64 : (SNK_MAX_CELLS + 63) / 64;                        \
    (void)HH; (void)WW; (void)NC; (void)MAXFW;

__device__ static inline bool food_bit(const uint64_t *food, int c) { return (food[c >> 6] >> (c & 63)) & 1ull; }

// DPP row_share:o (gfx90a+): every lane of a 16-lane row reads lane o of its row
__device__ static inline int row_share16(int v, int o)
{
    switch (o & 15) {
    case 0: return __builtin_amdgcn_update_dpp(0, v, 0x150, 0xF, 0xF, true);
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x151, 0xF, 0xF, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x152, 0xF, 0xF, true);
    case 3: return __builtin_amdgcn_update_dpp(0, v, 0x153, 0xF, 0xF, true);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x154, 0xF, 0xF, true);
    case 5: return __builtin_amdgcn_update_dpp(0, v, 0x155, 0xF, 0xF, true);
    case 6: return __builtin_amdgcn_update_dpp(0, v, 0x156, 0xF, 0xF, true);
    case 7: return __builtin_amdgcn_update_dpp(0, v, 0x157, 0xF, 0xF, true);
    default: return v;
    }
}

// A game's lanes all sit in ONE wavefront and its LDS region is touched by no other wave, so the phases of a kernel need
// no workgroup barrier: LDS instructions of one wave execute in issue order; what is needed is that the compiler keeps
// them in program order across the phase boundary (a wavefront-scope fence + the wave barrier, which emit no s_barrier).
#define GAME_SYNC()                                                                                     \
    do {                                                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                          \
        __builtin_amdgcn_wave_barrier();                                                                \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                          \
    } while (0)

// per-wave LDS carve-up used by step / observe
__host__ __device__ static inline int lds_per_wave(const Layout &L) { return L.stride + 4 * L.nc_pad + 64; }
// k_observe: the record + two float planes (channel 0 and channel 1 values per board cell).  Boards of more than 255 cells (16-bit
// ring entries: 19x19 / 8 snakes = 8 KB of rings per record) keep the rings in HBM / L2 and bring only the record's tail (meta,
// food, counters) and the LIVE ring segments -- at most one node per cell plus the stacked tails -- to LDS (round 6: with the
// whole record an observation took 19.8 KB of LDS, 8 wavefronts per CU; now 11.2 KB, 12): obs_rec_bytes
__host__ __device__ static inline int obs_seg_cap(const Layout &L) { return L.nc_pad + 32; }             // live nodes of all snakes
__host__ __device__ static inline int obs_rec_bytes(const Layout &L)
{
    if (L.cell_bytes == 1) return L.stride;
    return (L.stride - L.meta_off) + 32 + (2 * obs_seg_cap(L) + 15) / 16 * 16;       // record tail, 8 segment offsets, the segments
}
__host__ __device__ static inline int lds_per_obs(const Layout &L) { return obs_rec_bytes(L) + 8 * L.nc_pad; }
// k_observe, NHWC planes: + a canvas of the H window rows, 3 (2W-1) H floats (+ two pieces of slack at the ends)
__host__ __device__ static inline int lds_per_wave_win(const Layout &L)
{
    return lds_per_obs(L) + ((3 * (2 * L.W - 1) * L.H + 8) * 4 + 15) / 16 * 16;
}
// k_observe, channel-major planes: + an output canvas of (2H-1)(2W-1)3 floats (+ alignment slack)
__host__ __device__ static inline int lds_per_wave_obs(const Layout &L)
{
    return lds_per_obs(L) + (((2 * L.H - 1) * (2 * L.W - 1) * 3 + 8) * 4 + 15) / 16 * 16;
}

// ------------------------------------------------------------------------------------------
// Game.tic (game.py:87-205)
// GL lanes own one game: 16 (four games per wavefront; 11x11 and 7x7, whose per-game LDS is ~1.2 KB) or 64.  With 4 snakes
// and 121 cells most of a 64-lane wave idles in the per-snake and per-cell steps, and the kernel is instruction-issue
// bound, not HBM bound: four games per wave cut the wave-instructions per game about three times.
// Sub-lane sl = lane % GL plays the role the lane index had; shuffles and ballots are confined to the game's lane group.
// ------------------------------------------------------------------------------------------
template <int H, int W, int GL, int SS = 0>      // SS: compile-time snake count (0 = read it from the layout): the per-snake loops unroll
__global__ __launch_bounds__(BLOCK_THREADS) void k_step(uint8_t *__restrict__ state, Layout L,
                                                       const int32_t *__restrict__ slots, int n,
                                                       const uint8_t *__restrict__ moves,
                                                       const int16_t *__restrict__ spawn_tape,
                                                       uint8_t *__restrict__ done_out,
                                                       int16_t *__restrict__ spawned_out,
                                                       uint64_t *__restrict__ empty_out, int health_dec,
                                                       double chance, uint32_t seed_lo, uint32_t seed_hi,
                                                       const uint8_t *__restrict__ active, const int *__restrict__ skip)
{
    using cell_t = typename CellT<H * W>::type;
    BOARD_DIMS(L)
    constexpr int GPW = 64 / GL;                       // games per wavefront
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sl = lane % GL, gq = lane / GL, gbase = lane - sl;
    const int gi = (blockIdx.x * WAVES_PER_BLOCK + wv) * GPW + gq;
    // games whose `active` flag is 0 are frozen (a rollout sub-game that reached its depth cap, mp_game_runner.py:108):
    // neither loaded nor stored; the flag load is independent of the record load (no slot indirection, one round trip)
    // (skip: the rollout tick's gate, csrc/mcts.hip TICK_GATE -- a non-zero word freezes every game of the launch)
    const bool frozen = gi < n && ((active && !active[gi]) || (skip && *(volatile const int *)skip));
    const bool valid = gi < n && !frozen;
    const int S = SS > 0 ? SS : L.S, mask = L.cap_mask;
// value of sub-lane o of the game's lane group: with 16-lane groups (= one DPP row) a row_share move, one VALU instruction,
// instead of a ds_bpermute through the LDS crossbar (o is a constant after the per-snake loops unroll)
#define GSHFL(v, o) (GL == 16 ? row_share16((v), (o)) : __shfl((v), gbase + (o), 64))
#define GBALLOT(pr) (GL == 64 ? __ballot(pr) : ((__ballot(pr) >> gbase) & ((1ull << (GL & 63)) - 1ull)))
    uint8_t *g = smem + (wv * GPW + gq) * lds_per_wave(L);
    uint8_t *occ = g + L.stride;       // 1 = some snake's non-head node sits here (Game.bodies)
    uint8_t *hd = occ + L.nc_pad;      // 1 = some snake's head sits here (Game.heads, on-board ones)
    const int slot = valid ? (slots ? slots[gi] : gi) : 0;
    uint8_t *gsrc = state + (size_t)slot * L.stride;

    // the per-game inputs do not depend on the record: request them together with it (one HBM round trip, not two)
    const int mv_in = (valid && sl < S) ? moves[(size_t)gi * S + sl] : 1;
    const int tape_in = (valid && spawn_tape) ? spawn_tape[gi] : -1;
    if (valid)
        for (int i = sl; i < L.stride / 16; i += GL) ((uint4 *)g)[i] = ((const uint4 *)gsrc)[i];
    for (int i = sl * 16; i < 2 * L.nc_pad; i += GL * 16) *(uint4 *)(occ + i) = make_uint4(0u, 0u, 0u, 0u);   // both byte planes
    GAME_SYNC();

    SnakeMeta *meta = (SnakeMeta *)(g + L.meta_off);
    uint64_t *food = (uint64_t *)(g + L.food_off);
    uint32_t *cnt = (uint32_t *)(g + L.cnt_off);
    int8_t *rew = (int8_t *)(g + L.rew_off);

    const bool act = valid && sl < S;
    SnakeMeta m = {0, 0, 0, 0, 0};
    if (act) m = meta[sl];
    const bool alive0 = act && m.alive;
    const int n_alive0 = __popcll(GBALLOT(alive0));
    const bool ended = !valid || n_alive0 <= 1;     // tic already returned the rewards list earlier
    const bool go = alive0 && !ended;
    cell_t *ring = (cell_t *)(g + (sl < S ? sl : 0) * L.ring_bytes);

    // ---- execute moves (game.py:90-114; Snake.move game.py:329-358) + health (117-118)
    int head_cell = -1;
    bool oob = false;
    if (go) {
        const int d = (mv_in + m.dir + 3) & 3;            // (move + last - 1) % 4
        m.dir = (uint8_t)d;
        const int hi = (m.tail + m.len - 1) & mask;
        const int hc = ring[hi];
        int y = hc / WW, x = hc - y * WW;
        y += (d == 2) - (d == 0);
        x += (d == 1) - (d == 3);
        oob = (y < 0) | (y >= HH) | (x < 0) | (x >= WW);
        head_cell = oob ? -1 : y * WW + x;
        m.tail = (uint16_t)((m.tail + 1) & mask);      // pop the tail node, push the new head
        if (!oob) ring[(hi + 1) & mask] = (cell_t)head_cell;
        m.health = (int16_t)(m.health - health_dec);
    }
    // ---- food (game.py:121-127): list order, the first snake on a cell eats; Snake.grow (360-365)
    const bool hasfood = go && !oob && food_bit(food, head_cell);
    bool eats = hasfood;
#pragma unroll
    for (int o = 0; o < (SS > 0 ? SS : SNK_MAX_SNAKES); ++o) {
        if (o >= S) break;
        const int ho = GSHFL(head_cell, o);
        const int fo = GSHFL((int)hasfood, o);
        if (o < sl && fo && ho == head_cell) eats = false;
    }
    if (eats) {
        m.health = 100;
        const int t = (m.tail - 1) & mask;
        ring[t] = ring[m.tail];                        // duplicate the tail node
        m.tail = (uint16_t)t;
        m.len = (uint16_t)(m.len + 1);
    }
    const int n_eat = __popcll(GBALLOT(eats));
#pragma unroll
    for (int o = 0; o < (SS > 0 ? SS : SNK_MAX_SNAKES); ++o) {
        if (o >= S) break;
        const int eo = GSHFL((int)eats, o);
        const int co = GSHFL(head_cell, o);
        if (eo && sl == 0) food[co >> 6] &= ~(1ull << (co & 63));
    }
    GAME_SYNC();

    // ---- Game.bodies / Game.heads as LDS byte planes, rebuilt from the rings by all lanes of the game
#pragma unroll
    for (int s = 0; s < (SS > 0 ? SS : SNK_MAX_SNAKES); ++s) {
        if (s >= S) break;
        const int len_s = GSHFL((int)m.len, s);
        const int tail_s = GSHFL((int)m.tail, s);
        const int go_s = GSHFL((int)go, s);
        if (go_s) {
            const cell_t *r = (const cell_t *)(g + s * L.ring_bytes);
            for (int k = sl; k < len_s - 1; k += GL) occ[r[(tail_s + k) & mask]] = 1;
        }
    }
    if (go && !oob) hd[head_cell] = 1;
    GAME_SYNC();

    // ---- spawn food (game.py:130-138).  The game's empty-cell mask is assembled 64 cells per word from GL-cell ballots.
    int spawn = -1;
    if (chance > 0.0 && !ended) {
        int n_food = 0;
        for (int w = 0; w < L.FW; ++w) n_food += __popcll(food[w]);
        uint64_t emk[MAXFW];
        int n_empty = 0;
#pragma unroll
        for (int w = 0; w < MAXFW; ++w) {
            if (w >= L.FW) { emk[w] = 0ull; continue; }
            uint64_t mk = 0ull;
#pragma unroll
            for (int q = 0; q < GPW; ++q) {
                const int c = w * 64 + q * GL + sl;
                const bool e = c < NC && !occ[c] && !hd[c] && !food_bit(food, c);
                mk |= GBALLOT(e) << (q * GL);
            }
            emk[w] = mk;
            if (empty_out && sl == 0) empty_out[(size_t)gi * L.FW + w] = mk;
            n_empty += __popcll(mk);
        }
        if (spawn_tape) {
            spawn = tape_in;
        } else {
            const uint32_t uid = *(const uint32_t *)(g + L.uid_off);
            uint32_t r[4];
            philox4x32(uid, cnt[5], 0x5350574Eu /* 'SPWN' */, 0u, seed_lo, seed_hi, r);
            const double u1 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0);
            if ((n_food == 0 || u1 <= chance) && n_empty > 0) {
                int k = (int)(((uint64_t)r[1] * (uint64_t)n_empty) >> 32);   // uniform in [0, n_empty)
#pragma unroll
                for (int w = 0; w < MAXFW; ++w) {
                    const uint64_t mk = emk[w];
                    const int pc = __popcll(mk);
                    if (spawn < 0) {
                        if (k < pc) {                  // position of the k-th set bit of mk by halving
                            uint64_t t = mk;
                            int pos = 0;
#pragma unroll
                            for (int sh = 32; sh >= 1; sh >>= 1) {
                                const int cl = __popcll(t & ((1ull << sh) - 1ull));
                                if (k >= cl) { k -= cl; t >>= sh; pos += sh; }
                            }
                            spawn = w * 64 + pos;
                        } else {
                            k -= pc;
                        }
                    }
                }
            }
        }
    } else if (empty_out && valid && sl == 0) {
        for (int w = 0; w < L.FW; ++w) empty_out[(size_t)gi * L.FW + w] = 0ull;
    }
    if (spawned_out && valid && sl == 0) spawned_out[gi] = (int16_t)spawn;

    // ---- deaths (game.py:144-165, an if/elif chain) and removal (167-192)
    const bool body_hit = go && !oob && occ[head_cell];
    bool shared = false, lose = false;
#pragma unroll
    for (int o = 0; o < (SS > 0 ? SS : SNK_MAX_SNAKES); ++o) {
        if (o >= S) break;
        const int ho = GSHFL(head_cell, o);
        const int lo = GSHFL((int)m.len, o);
        if (o != sl && ho >= 0 && ho == head_cell) {
            shared = true;
            if ((int)m.len <= lo) lose = true;
        }
    }
    int cause = -1;
    if (go) {
        if (oob) cause = 0;
        else if (body_hit) cause = 1;
        else if (shared) { if (lose) cause = 2; }      // a head-on survivor skips the starvation test
        else if (m.health <= 0) cause = 3;
    }
    const bool dead = cause >= 0;
    const int c0 = __popcll(GBALLOT(cause == 0)), c1 = __popcll(GBALLOT(cause == 1));
    const int c2 = __popcll(GBALLOT(cause == 2)), c3 = __popcll(GBALLOT(cause == 3));
    const int n_alive = __popcll(GBALLOT(go && !dead));
    if (!ended) {
        if (spawn >= 0 && sl == 0) food[spawn >> 6] |= 1ull << (spawn & 63);
        if (dead) {
            m.alive = 0; m.len = 0; m.health = 0; m.dir = 0; m.tail = 0;
            rew[sl] = -1;
        } else if (go && n_alive == 1) {
            rew[sl] = 1;                               // game.py:199-202
        }
        if (act) meta[sl] = m;
        if (sl == 0) {
            cnt[0] += c0; cnt[1] += c1; cnt[2] += c2; cnt[3] += c3;
            cnt[4] += n_eat; cnt[5] += 1;              // game_length (game.py:197)
        }
    }
    GAME_SYNC();
    if (!ended)
        for (int i = sl; i < L.stride / 16; i += GL) ((uint4 *)gsrc)[i] = ((const uint4 *)g)[i];
    if (done_out && valid && sl == 0) done_out[gi] = (uint8_t)(ended || n_alive <= 1);
    if (done_out && frozen && sl == 0) done_out[gi] = 0;
#undef GSHFL
#undef GBALLOT
}

// ------------------------------------------------------------------------------------------
// Game.tic again, for games of at most 4 snakes on boards of at most 255 cells: ONE LANE PER SNAKE, a quad per game,
// sixteen games per wavefront.  k_step above is VALU-issue bound (SQ counters at 262 144 games: 455 VALU instructions
// per wave of four games, three quarters of its lanes idle in the per-snake steps); here every lane of the per-snake steps
// works, cross-snake questions are DPP quad_perm moves (one VALU instruction, no LDS crossbar), the body and head
// occupancy are LDS BIT planes filled with ds_or_b32 by each snake's own lane (its ring walk is sequential), the
// empty-cell mask of the food spawn is three 64-bit logic operations per word, and the death / eat counts of a game
// travel as one packed quad sum.  Same record, same arguments, same results as k_step.
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int q_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
__host__ __device__ constexpr int q_stride(int nc, int s)          // make_layout's stride for 8-bit rings
{
    return (s * q_pow2(nc + 2) + s * 8 + (nc + 63) / 64 * 8 + 24 + (s + 3) / 4 * 4 + 4 + 15) / 16 * 16;
}
__device__ static inline int quad_bcast(int v, int o)              // lane o of the quad
{
    switch (o & 3) {
    case 0: return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, true);
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x55, 0xF, 0xF, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xF, 0xF, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, true);
    }
}
__device__ static inline int quad_sum(int v)                       // all four lanes get the sum
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    return v;
}
// LDS per game of the quad form: the record + two bit planes (bodies, heads) of QFW 64-bit words each
__host__ __device__ static inline int lds_per_game_quad(const Layout &L) { return L.stride + 16 * ((L.FW + 1) / 2) * 2; }

template <int H, int W, int SS>
__global__ __launch_bounds__(BLOCK_THREADS) void k_step_quad(uint8_t *__restrict__ state, Layout L,
                                                            const int32_t *__restrict__ slots, int n,
                                                            const uint8_t *__restrict__ moves,
                                                            const int16_t *__restrict__ spawn_tape,
                                                            uint8_t *__restrict__ done_out,
                                                            int16_t *__restrict__ spawned_out,
                                                            uint64_t *__restrict__ empty_out, int health_dec,
                                                            double chance, uint32_t seed_lo, uint32_t seed_hi,
                                                            const uint8_t *__restrict__ active, const int *__restrict__ skip)
{
    using cell_t = uint8_t;
    const int HH = H > 0 ? H : L.H, WW = W > 0 ? W : L.W;
    const int NC = HH * WW;
    constexpr int QFW = H > 0 ? (H * W + 63) / 64 : 4;             // 64-bit words of a bit plane (255 cells at most)
    constexpr int CHUNKS = (H > 0 && SS > 0) ? q_stride(H * W, SS) / 16 : 0;
    constexpr int GPWQ = 16;
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sl = lane & 3, gq = lane >> 2;
    const int gi = (blockIdx.x * WAVES_PER_BLOCK + wv) * GPWQ + gq;
    // (skip: the rollout tick's gate, csrc/mcts.hip TICK_GATE -- a non-zero word freezes every game of the launch)
    const bool frozen = gi < n && ((active && !active[gi]) || (skip && *(volatile const int *)skip));
    const bool valid = gi < n && !frozen;
    const int S = SS > 0 ? SS : L.S, mask = L.cap_mask;
    const int FW = H > 0 ? QFW : L.FW;
    const int plane_bytes = 8 * ((L.FW + 1) / 2) * 2;
    uint8_t *g = smem + (size_t)(wv * GPWQ + gq) * L.stride;                       // records first: 16-byte chunks stay aligned
    uint32_t *bodies = (uint32_t *)(smem + (size_t)WAVES_PER_BLOCK * GPWQ * L.stride
                                    + (size_t)(wv * GPWQ + gq) * 2 * plane_bytes);
    uint32_t *heads = bodies + plane_bytes / 4;
    const int slot = valid ? (slots ? slots[gi] : gi) : 0;
    uint8_t *gsrc = state + (size_t)slot * L.stride;

    const int mv_in = (valid && sl < S) ? moves[(size_t)gi * S + sl] : 1;
    const int tape_in = (valid && spawn_tape) ? spawn_tape[gi] : -1;
    if (CHUNKS > 0) {                                              // all loads of the record in flight at once
        uint4 t[(CHUNKS + 3) / 4 > 0 ? (CHUNKS + 3) / 4 : 1];         // unconditional loads (an idle quad re-reads slot 0's
#pragma unroll                                                       // first chunk), so that the array stays in registers
        for (int j = 0; j < (CHUNKS + 3) / 4; ++j)
            t[j] = ((const uint4 *)gsrc)[valid ? min(sl + 4 * j, CHUNKS - 1) : 0];
#pragma unroll
        for (int j = 0; j < (CHUNKS + 3) / 4; ++j)
            if (valid && sl + 4 * j < CHUNKS) ((uint4 *)g)[sl + 4 * j] = t[j];
    } else if (valid) {
        for (int i = sl; i < L.stride / 16; i += 4) ((uint4 *)g)[i] = ((const uint4 *)gsrc)[i];
    }
    for (int i = sl; i < plane_bytes / 2; i += 4) bodies[i] = 0u;                 // both planes
    GAME_SYNC();

    SnakeMeta *meta = (SnakeMeta *)(g + L.meta_off);
    uint64_t *food = (uint64_t *)(g + L.food_off);
    uint32_t *food32 = (uint32_t *)(g + L.food_off);
    uint32_t *cnt = (uint32_t *)(g + L.cnt_off);
    int8_t *rew = (int8_t *)(g + L.rew_off);

    const bool act = valid && sl < S;
    SnakeMeta m = {0, 0, 0, 0, 0};
    if (act) m = meta[sl];
    const bool alive0 = act && m.alive;
    const int n_alive0 = quad_sum((int)alive0);
    const bool ended = !valid || n_alive0 <= 1;
    const bool go = alive0 && !ended;
    cell_t *ring = (cell_t *)(g + (sl < S ? sl : 0) * L.ring_bytes);

    // ---- execute moves (game.py:90-114) + health (117-118)
    int head_cell = -1;
    bool oob = false;
    if (go) {
        const int d = (mv_in + m.dir + 3) & 3;
        m.dir = (uint8_t)d;
        const int hi = (m.tail + m.len - 1) & mask;
        const int hc = ring[hi];
        int y = hc / WW, x = hc - y * WW;
        y += (d == 2) - (d == 0);
        x += (d == 1) - (d == 3);
        oob = (y < 0) | (y >= HH) | (x < 0) | (x >= WW);
        head_cell = oob ? -1 : y * WW + x;
        m.tail = (uint16_t)((m.tail + 1) & mask);
        if (!oob) ring[(hi + 1) & mask] = (cell_t)head_cell;
        m.health = (int16_t)(m.health - health_dec);
    }
    // ---- food (game.py:121-127): the first snake in list order on a cell eats
    const bool hasfood = go && !oob && ((food32[head_cell >> 5] >> (head_cell & 31)) & 1u);
    bool eats = hasfood;
    const int fcell = hasfood ? head_cell : -1;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int fo = quad_bcast(fcell, o);
        if (o < sl && fo >= 0 && fo == head_cell) eats = false;
    }
    if (eats) {
        m.health = 100;
        const int t = (m.tail - 1) & mask;
        ring[t] = ring[m.tail];                        // duplicate the tail node (Snake.grow game.py:360-365)
        m.tail = (uint16_t)t;
        m.len = (uint16_t)(m.len + 1);
        atomicAnd(&food32[head_cell >> 5], ~(1u << (head_cell & 31)));
    }
    // ---- Game.bodies / Game.heads as bit planes: every snake's own lane walks its ring
    if (go) {
        for (int k = 0; k < (int)m.len - 1; ++k) {
            const int c = ring[(m.tail + k) & mask];
            atomicOr(&bodies[c >> 5], 1u << (c & 31));
        }
        if (!oob) atomicOr(&heads[head_cell >> 5], 1u << (head_cell & 31));
    }
    GAME_SYNC();

    // ---- spawn food (game.py:130-138): empty = not body, not head, not food
    int spawn = -1;
    if (chance > 0.0 && !ended) {
        int n_food = 0, n_empty = 0;
        uint64_t emk[QFW];
#pragma unroll
        for (int w = 0; w < QFW; ++w) {
            emk[w] = 0ull;
            if (w < FW) {
                const uint64_t fw = food[w];
                const uint64_t used = fw | ((const uint64_t *)bodies)[w] | ((const uint64_t *)heads)[w];
                const int left = NC - 64 * w;                                       // cells of this word
                const uint64_t cells = left >= 64 ? ~0ull : ((1ull << (left > 0 ? left : 0)) - 1ull);
                emk[w] = ~used & cells;
                n_food += __popcll(fw);
                n_empty += __popcll(emk[w]);
                if (empty_out && sl == 0) empty_out[(size_t)gi * L.FW + w] = emk[w];
            }
        }
        if (spawn_tape) {
            spawn = tape_in;
        } else {
            const uint32_t uid = *(const uint32_t *)(g + L.uid_off);
            uint32_t r[4];
            philox4x32(uid, cnt[5], 0x5350574Eu /* 'SPWN' */, 0u, seed_lo, seed_hi, r);
            const double u1 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0);
            if ((n_food == 0 || u1 <= chance) && n_empty > 0) {
                int k = (int)(((uint64_t)r[1] * (uint64_t)n_empty) >> 32);
#pragma unroll
                for (int w = 0; w < QFW; ++w) {
                    const uint64_t mk = emk[w];
                    const int pc = __popcll(mk);
                    if (spawn < 0) {
                        if (k < pc) {
                            uint64_t t = mk;
                            int pos = 0;
#pragma unroll
                            for (int sh = 32; sh >= 1; sh >>= 1) {
                                const int cl = __popcll(t & ((1ull << sh) - 1ull));
                                if (k >= cl) { k -= cl; t >>= sh; pos += sh; }
                            }
                            spawn = w * 64 + pos;
                        } else {
                            k -= pc;
                        }
                    }
                }
            }
        }
    } else if (empty_out && valid && sl == 0) {
        for (int w = 0; w < L.FW; ++w) empty_out[(size_t)gi * L.FW + w] = 0ull;
    }
    if (spawned_out && valid && sl == 0) spawned_out[gi] = (int16_t)spawn;

    // ---- deaths (game.py:144-165) and removal (167-192)
    const bool body_hit = go && !oob && ((bodies[head_cell >> 5] >> (head_cell & 31)) & 1u);
    bool shared = false, lose = false;
    const int hl = (head_cell + 1) | ((int)m.len << 9);            // head_cell + 1 <= 255, length in the bits above
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int q = quad_bcast(hl, o);
        const int ho = (q & 511) - 1, lo = q >> 9;
        if (o != sl && ho >= 0 && ho == head_cell) {
            shared = true;
            if ((int)m.len <= lo) lose = true;
        }
    }
    int cause = -1;
    if (go) {
        if (oob) cause = 0;
        else if (body_hit) cause = 1;
        else if (shared) { if (lose) cause = 2; }
        else if (m.health <= 0) cause = 3;
    }
    const bool dead = cause >= 0;
    const int tally = quad_sum((cause >= 0 ? 1 << (3 * cause) : 0) | ((go && !dead) ? 1 << 12 : 0) | (eats ? 1 << 15 : 0));
    const int n_alive = (tally >> 12) & 7;
    if (!ended) {
        if (dead) {
            m.alive = 0; m.len = 0; m.health = 0; m.dir = 0; m.tail = 0;
            rew[sl] = -1;
        } else if (go && n_alive == 1) {
            rew[sl] = 1;
        }
        if (act) meta[sl] = m;
        if (sl == 0) {
            if (spawn >= 0) food[spawn >> 6] |= 1ull << (spawn & 63);
            cnt[0] += tally & 7; cnt[1] += (tally >> 3) & 7; cnt[2] += (tally >> 6) & 7; cnt[3] += (tally >> 9) & 7;
            cnt[4] += (tally >> 15) & 7; cnt[5] += 1;
        }
    }
    GAME_SYNC();
    if (!ended) {
        if (CHUNKS > 0) {
#pragma unroll
            for (int j = 0; j < (CHUNKS + 3) / 4; ++j)
                if (sl + 4 * j < CHUNKS) ((uint4 *)gsrc)[sl + 4 * j] = ((const uint4 *)g)[sl + 4 * j];
        } else {
            for (int i = sl; i < L.stride / 16; i += 4) ((uint4 *)gsrc)[i] = ((const uint4 *)g)[i];
        }
    }
    if (done_out && valid && sl == 0) done_out[gi] = (uint8_t)(ended || n_alive <= 1);
    if (done_out && frozen && sl == 0) done_out[gi] = 0;
}

// ------------------------------------------------------------------------------------------
// Game.__init__ (game.py:13-61)
// ------------------------------------------------------------------------------------------
template <int H, int W>
__global__ __launch_bounds__(BLOCK_THREADS) void k_reset(uint8_t *__restrict__ state, Layout L,
                                                        const int32_t *__restrict__ slots, int n,
                                                        const uint8_t *__restrict__ tape, uint32_t uid_base,
                                                        uint32_t seed_lo, uint32_t seed_hi)
{
    using cell_t = typename CellT<H * W>::type;
    BOARD_DIMS(L)
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gi = blockIdx.x * WAVES_PER_BLOCK + wv;
    const bool valid = gi < n;
    const int S = L.S;
    uint8_t *g = smem + wv * lds_per_wave(L);
    for (int i = lane; i < L.stride / 16; i += 64) ((uint4 *)g)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (valid && lane == 0) {
        const int sy[8] = {1, HH - 2, HH - 2, 1, 1, HH / 2, HH - 2, HH / 2};
        const int sx[8] = {1, WW - 2, 1, WW - 2, WW / 2, WW - 2, WW / 2, 1};
        uint8_t pos[8], dirs[8], fc[8];
        const uint32_t uid = uid_base + (uint32_t)gi;
        if (tape) {
            for (int s = 0; s < S; ++s) {
                pos[s] = tape[((size_t)gi * 3 + 0) * S + s];
                dirs[s] = tape[((size_t)gi * 3 + 1) * S + s];
                fc[s] = tape[((size_t)gi * 3 + 2) * S + s];
            }
        } else {
            uint32_t r[24];
            for (int q = 0; q < 6; ++q) philox4x32(uid, (uint32_t)q, 0x494E4954u /* 'INIT' */, 0u, seed_lo, seed_hi, r + 4 * q);
            uint8_t perm[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            for (int s = 0; s < S; ++s) {          // random.sample(8 start cells, S): ordered S-subset
                const int j = s + (int)(((uint64_t)r[s] * (uint64_t)(8 - s)) >> 32);
                const uint8_t t = perm[s]; perm[s] = perm[j]; perm[j] = t;
                pos[s] = perm[s];
                dirs[s] = (uint8_t)(r[8 + s] >> 30);
                fc[s] = (uint8_t)(r[16 + s] >> 30);
            }
        }
        SnakeMeta *meta = (SnakeMeta *)(g + L.meta_off);
        uint64_t *food = (uint64_t *)(g + L.food_off);
        const int center = (HH / 2) * WW + WW / 2;
        food[center >> 6] |= 1ull << (center & 63);
        for (int s = 0; s < S; ++s) {
            const int y = sy[pos[s]], x = sx[pos[s]];
            cell_t *ring = (cell_t *)(g + s * L.ring_bytes);
            ring[0] = ring[1] = ring[2] = (cell_t)(y * WW + x);     // 3 stacked nodes (game.py:37)
            meta[s].tail = 0; meta[s].len = 3; meta[s].health = 100; meta[s].dir = dirs[s]; meta[s].alive = 1;
            const int fy = y + ((fc[s] & 2) ? 1 : -1), fx = x + ((fc[s] & 1) ? 1 : -1);
            const int c = fy * WW + fx;
            food[c >> 6] |= 1ull << (c & 63);
        }
        *(uint32_t *)(g + L.uid_off) = uid;
    }
    __syncthreads();
    if (valid) {
        const int slot = slots ? slots[gi] : gi;
        uint8_t *gdst = state + (size_t)slot * L.stride;
        for (int i = lane; i < L.stride / 16; i += 64) ((uint4 *)gdst)[i] = ((const uint4 *)g)[i];
    }
}

// ------------------------------------------------------------------------------------------
// Game.subgame (game.py:266-276): one wavefront reads a parent once and writes `fanout` copies
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS) void k_clone(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                        Layout L, const int32_t *__restrict__ src_slots, int n,
                                                        const int32_t *__restrict__ dst_slots, int fanout)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gi = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (gi >= n) return;
    const int ss = src_slots ? src_slots[gi] : gi;
    const uint4 *p = (const uint4 *)(src + (size_t)ss * L.stride);
    const int nq = L.stride / 16;
    // the counters (6 x u32 at cnt_off, 8-byte aligned) are zeroed in flight
    const int c_lo = L.cnt_off, c_hi = L.cnt_off + 24;
    for (int i = lane; i < nq; i += 64) {
        uint4 v = p[i];
        uint32_t *w = (uint32_t *)&v;
        for (int q = 0; q < 4; ++q) {
            const int off = i * 16 + q * 4;
            if (off >= c_lo && off < c_hi) w[q] = 0u;
        }
        for (int j = 0; j < fanout; ++j) {
            const int ds = dst_slots ? dst_slots[(size_t)gi * fanout + j] : gi * fanout + j;
            ((uint4 *)(dst + (size_t)ds * L.stride))[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// alive flags (Game.get_ids, game.py:76-77)
// ------------------------------------------------------------------------------------------
__global__ void k_alive(const uint8_t *__restrict__ state, Layout L, const int32_t *__restrict__ slots, int n,
                        uint8_t *__restrict__ alive_out, int32_t *__restrict__ n_alive_out)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n) return;
    const int slot = slots ? slots[gi] : gi;
    const SnakeMeta *meta = (const SnakeMeta *)(state + (size_t)slot * L.stride + L.meta_off);
    int c = 0;
    for (int s = 0; s < L.S; ++s) {
        const uint8_t a = meta[s].alive;
        alive_out[(size_t)gi * L.S + s] = a;
        c += a;
    }
    if (n_alive_out) n_alive_out[gi] = c;
}

__global__ void k_sum_counters(const uint8_t *__restrict__ state, Layout L, const int32_t *__restrict__ slots, int n,
                               unsigned long long *__restrict__ out6)
{
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int gi = blockIdx.x * blockDim.x + threadIdx.x; gi < n; gi += gridDim.x * blockDim.x) {
        const int slot = slots ? slots[gi] : gi;
        const uint32_t *c = (const uint32_t *)(state + (size_t)slot * L.stride + L.cnt_off);
        for (int q = 0; q < 6; ++q) acc[q] += c[q];
    }
    for (int q = 0; q < 6; ++q) {
        const unsigned long long s = wave_sum_u64(acc[q]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&out6[q], s);
    }
}

// ------------------------------------------------------------------------------------------
// Game.make_state (game.py:215-257) + obstacle test (alpha_nnet.py:63-76) + observation key
// GL lanes per (slot, snake) pair: 16 (four observations per wavefront; 11x11 and 7x7) or 64 (19x19).  Everything before
// the first plane byte (record load, tail-distance plane, 2 x splitmix64 per board cell for the key) is per-observation
// work that left most of a 64-lane wave idle; with four observations per wave it costs a quarter of the instructions.
// The host uses GL = 16 for requests without planes and GL = 64 for requests with planes (see snk_engine_observe).
// ------------------------------------------------------------------------------------------
template <int H, int W, int GL, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_observe(const uint8_t *__restrict__ state, Layout L,
                                                          const int32_t *__restrict__ pairs, int m, int layout,
                                                          float *__restrict__ planes, uint8_t *__restrict__ mask_out,
                                                          uint64_t *__restrict__ key_out, int legacy_mask, int reps,
                                                          const int32_t *__restrict__ index, const uint8_t *__restrict__ sub_active,
                                                          uint8_t *__restrict__ row_active)
{
    // index (optional): output row i observes pairs[index[i]] (the rollout tick's rows to evaluate: no gathered copy of the pairs);
    // row_active (optional, with sub_active): row_active[i] = the observing snake is alive AND sub_active[its slot] -- the tick's
    // "which rows are live" (mp_game_runner.py:99-103) taken where the record is in hand instead of by two launches of its own
    using cell_t = typename CellT<H * W>::type;
    BOARD_DIMS(L)
    const int N = 2 * HH - 1, NPIX = N * N, NEL = NPIX * 3;
    static_assert(H == W, "only square boards batch (rot90 transposes odd-k shapes)");
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int GPW = 64 / GL;                       // observations per wavefront
    // wave-uniform values are made scalar (readfirstlane): with a whole wavefront per observation everything derived from
    // the pair, the observing snake's meta word and its head then runs on the scalar unit instead of costing VALU issue
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sl = lane % GL, gq = lane / GL;
    // A lane group takes `reps` consecutive observations, one after the other, the next one's record already travelling
    // while the current one is worked on: a wavefront's life is two dependent loads (pair, record), the work, and the wait
    // for its stores to be acknowledged (s_endpgm waits for them); with one observation per wavefront that chain, not the
    // instruction count or the HBM rate, set the time (measured: 8 192 wave slots x 5.3 KB per ~10 us lifetime).
    const int pi0 = ((blockIdx.x * WPB + wv) * GPW + gq) * reps;
    const int S = L.S, mask = L.cap_mask;
    constexpr bool RG = sizeof(cell_t) == 2;              // rings stay in global memory (lds_per_obs above)
    const int roff = RG ? L.meta_off : 0;                 // first byte of the record that is staged in LDS
    const int nch = (L.stride - roff) / 16;
    uint4 ahead = make_uint4(0u, 0u, 0u, 0u);             // chunk `sl` of the (staged part of the) record of the next observation
    int ahead_you = 0, ahead_slot = 0;
    const uint8_t *ahead_src = state;
    if (pi0 < m) {
        const int pr = index ? index[pi0] : pi0;
        ahead_slot = pairs[2 * pr];
        ahead_src = state + (size_t)ahead_slot * L.stride;
        ahead_you = pairs[2 * pr + 1];
        if (sl < nch) ahead = ((const uint4 *)(ahead_src + roff))[sl];
    }
    for (int rep = 0; rep < reps; ++rep) {
    const int pi = pi0 + rep;
    const bool valid = pi < m;
    uint8_t *g = smem + (wv * GPW + gq) * (!planes ? lds_per_obs(L) : layout == SNK_NHWC_F32 ? lds_per_wave_win(L) : lds_per_wave_obs(L));
    float *v1 = (float *)(g + obs_rec_bytes(L));            // channel 1 per board cell: 0.02 x the largest tail-distance of a node on it
    float *v0 = v1 + L.nc_pad;                              // channel 0 per board cell: the head value of the snake whose head is here
    int *seg_off = (int *)(g + (L.stride - roff));          // RG: where each snake's live segment starts in `seg`
    cell_t *seg = (cell_t *)(g + (L.stride - roff) + 32);   // RG: the snakes' nodes tail -> head, one snake after the other

    const int you = ahead_you, slot = ahead_slot;
    const uint8_t *gsrc = ahead_src;
    if (valid && sl < nch) ((uint4 *)g)[sl] = ahead;
    if (valid)
        for (int i = sl + GL; i < nch; i += GL) ((uint4 *)g)[i] = ((const uint4 *)(gsrc + roff))[i];
    if (rep + 1 < reps && pi + 1 < m) {                     // request the next record now
        const int pr = index ? index[pi + 1] : pi + 1;
        ahead_slot = pairs[2 * pr];
        ahead_src = state + (size_t)ahead_slot * L.stride;
        ahead_you = pairs[2 * pr + 1];
        if (sl < nch) ahead = ((const uint4 *)(ahead_src + roff))[sl];
    }
    for (int i = sl * 16; i < 8 * L.nc_pad; i += GL * 16) *(uint4 *)((uint8_t *)v1 + i) = make_uint4(0u, 0u, 0u, 0u);   // both planes
    GAME_SYNC();

    const SnakeMeta *meta = (const SnakeMeta *)(g + L.meta_off - roff);
    const uint64_t *food = (const uint64_t *)(g + L.food_off - roff);
    if constexpr (RG) {
        // the live ring segments, straight from the record in global memory (its rings are L2-resident: the snakes of a game share
        // them) into `seg`, all snakes in ONE flattened pass so that the loads of different snakes travel together
        int tl_[SNK_MAX_SNAKES], of_[SNK_MAX_SNAKES], total = 0;
#pragma unroll
        for (int s = 0; s < SNK_MAX_SNAKES; ++s) {
            SnakeMeta ms = meta[s < S ? s : 0];
            const int len = (valid && s < S && ms.alive) ? (int)ms.len : 0;
            tl_[s] = ms.tail; of_[s] = total; total += len;
        }
        total = min(total, obs_seg_cap(L));                 // (a legal board holds at most one node per cell plus the stacked tails)
        if (sl < SNK_MAX_SNAKES) {
            int o = 0;
#pragma unroll
            for (int s = 0; s < SNK_MAX_SNAKES; ++s) o = sl == s ? of_[s] : o;
            seg_off[sl] = o;
        }
        for (int idx = sl; idx < total; idx += GL) {
            int s = 0, tl = tl_[0], of = 0;
#pragma unroll
            for (int q = 1; q < SNK_MAX_SNAKES; ++q)
                if (idx >= of_[q]) { s = q; tl = tl_[q]; of = of_[q]; }     // the LAST snake that starts at or before idx (dead ones are empty)
            seg[idx] = ((const cell_t *)(gsrc + s * L.ring_bytes))[(tl + (idx - of)) & mask];
        }
        GAME_SYNC();
    }
    SnakeMeta me = meta[you];
    if (GL == 64) {
        uint2 w = *(const uint2 *)&me;
        w.x = __builtin_amdgcn_readfirstlane(w.x); w.y = __builtin_amdgcn_readfirstlane(w.y);
        me = *(const SnakeMeta *)&w;
    }
    const bool live = valid && me.alive;

    // body plane: walking tail -> head with dist 1,2,... the last write wins (game.py:236-241);
    // only the node nearest the head of a run of stacked nodes writes, so there is no race.
    // With a whole wavefront per observation and at most four snakes, sixteen lanes walk each snake at once (one pass for
    // bodies of up to 16 nodes) instead of all lanes walking the snakes one after the other.  (Alive snakes never share a
    // cell in a state a tick produced, so the order between snakes does not matter.)
    constexpr int LPS = GL == 64 ? 16 : GL;              // lanes per snake
    const bool by_snake = GL == 64 && S <= 4;
    for (int s0 = 0; s0 < S; ++s0) {
        if (!valid) break;                               // an idle wave holds no record: touch nothing
        const int s = by_snake ? sl / LPS : s0;
        if (by_snake && (s0 > 0 || s >= S)) break;
        const SnakeMeta ms = meta[s];
        if (!ms.alive) continue;
        // node k of snake s, tail -> head: from the ring in LDS, or (RG) from the snake's segment
        const cell_t *r = RG ? seg + seg_off[s] : (const cell_t *)(g + s * L.ring_bytes);
        const int rt = RG ? 0 : ms.tail, rm = RG ? 0xFFFF : mask;
        const int len_s = RG ? min((int)ms.len, obs_seg_cap(L) - seg_off[s]) : (int)ms.len;
        for (int k = by_snake ? sl % LPS : sl; k < len_s; k += LPS) {
            const int c = r[(rt + k) & rm];
            const bool last = (k == ms.len - 1) || (r[(rt + k + 1) & rm] != c);
            if (last) v1[c] = (float)((double)(k + 1) * 0.02);          // float64 product, then float32 (game.py:236-241, 257)
            if (k == ms.len - 1)   // (snake.length - (you.length - 0.5)) * 0.04 in float64, then float32 (game.py:229-232,257)
                v0[c] = (float)(((double)ms.len - ((double)me.len - 0.5)) * 0.04);
        }
    }
    GAME_SYNC();

    int my_head = RG ? (int)seg[max(0, min(seg_off[you] + me.len - 1, obs_seg_cap(L) - 1))]
                     : (int)((const cell_t *)(g + you * L.ring_bytes))[(me.tail + me.len - 1) & mask];
    if (GL == 64) my_head = __builtin_amdgcn_readfirstlane(my_head);
    const int hy = my_head / WW, hx = my_head - hy * WW;
    const int k = me.dir & 3;
    const float fval = (float)((double)(101 - (int)me.health) * 0.01);   // game.py:243-244

    // value of board cell c, channel ch, exactly as make_state writes it
    auto cell_val = [&](int c, int ch) -> float {
        if (c == my_head) return -1.0f;                                   // game.py:248
        if (ch == 0) return v0[c];
        if (ch == 1) return v1[c];
        return food_bit(food, c) ? fval : 0.0f;
    };

    if (planes && layout == SNK_NHWC_F32) {
        // The reference's layout, the one the net reads: every output byte is written once, in aligned 16-byte pieces.
        // The observation is the wall pattern (0, 1, 0 per pixel) except inside the H canvas rows the board window occupies
        // -- one contiguous element range.  Pieces outside that range go out straight from registers (the pattern of a piece
        // depends on the channel of its first element only, which advances by one from a lane's piece to its next: three
        // float4 registers used in turn); pieces inside it are first laid into a small LDS canvas (window rows only), the
        // board cells are scattered over them there, and the canvas is streamed out.  Measured on the way: the full-canvas
        // form with per-element pattern arithmetic was VALU-issue bound (SQ counters: 715 VALU instructions per observation,
        // 104 us of VALU issue against 67 us of HBM time at 78 229 observations); pattern everywhere + cells stored over it
        // as 12-byte pixels cost 50 us for the cells alone (partial-line writes of lines that had left the L2).
        float *out = planes + (size_t)pi * NEL;
        const int lead = (int)(((size_t)pi * NEL) & 3);               // out + e is 16-byte aligned where (e + lead) % 4 == 0
        const int nvec = (NEL + lead + 3) / 4;                          // piece q = elements 4 q - lead .. 4 q - lead + 3
        float *cvs = (float *)(g + lds_per_obs(L));                    // cvs[4 (q - qa) + t] <-> element 4 q - lead + t
        int qa = 0, qb = 0;                                            // pieces of the window rows
        if (live) {
            const int i0 = k == 0 ? HH - 1 - hy : k == 1 ? hx : k == 2 ? hy : WW - 1 - hx;   // first canvas row of the window
            qa = (3 * N * i0 + lead) >> 2;
            qb = (3 * N * (i0 + HH) + lead + 3) >> 2;
        }
        const int q_lo = lead ? 1 : 0, q_hi = (NEL + lead) >> 2;       // pieces q_lo <= q < q_hi lie wholly inside the observation
        const float one = live ? 1.0f : 0.0f;                          // the observation of a dead snake is all zeros
        if (valid) {
            static_assert((4 * GL) % 3 == 1, "the channel of a lane's next piece advances by one");
            const int c0 = (4 * sl - lead + 3) % 3;                    // channel of the first element of piece q = sl
            const float a = c0 == 0 ? one : 0.0f, b = c0 == 1 ? one : 0.0f, c = c0 == 2 ? one : 0.0f;
            float4 p0 = make_float4(b, a, c, b), p1 = make_float4(a, c, b, a), p2 = make_float4(c, b, a, c);
            for (int q = sl; q < nvec; q += GL) {
                if ((unsigned)(q - qa) < (unsigned)(qb - qa)) *(float4 *)(cvs + 4 * (q - qa)) = p0;
                else if ((unsigned)(q - q_lo) < (unsigned)(q_hi - q_lo)) *(float4 *)(out + (4 * q - lead)) = p0;
                const float4 t4 = p0; p0 = p1; p1 = p2; p2 = t4;
            }
        }
        GAME_SYNC();
        if (live) {
            const int base = lead - 4 * qa;
            for (int c = sl; c < NC; c += GL) {
                const int y = c / WW, x = c - y * WW;
                const int si = y - hy + (HH - 1), sj = x - hx + (WW - 1);
                int i, j;                              // numpy.rot90(grid, k): out[i][j] = grid[si][sj], inverted
                if (k == 0) { i = si; j = sj; }
                else if (k == 1) { i = N - 1 - sj; j = si; }
                else if (k == 2) { i = N - 1 - si; j = N - 1 - sj; }
                else { i = sj; j = N - 1 - si; }
                float *px = cvs + (3 * (i * N + j) + base);
                px[0] = v0[c]; px[1] = v1[c]; px[2] = food_bit(food, c) ? fval : 0.0f;
            }
            // the observer's own head (game.py:248) is the centre pixel whatever the rotation: -1 in all three channels,
            // written after the scatter (LDS instructions of a wave execute in order)
            if (sl < 3) cvs[3 * ((HH - 1) * N + (WW - 1)) + base + sl] = -1.0f;
        }
        GAME_SYNC();
        if (live)
            for (int q = max(qa, q_lo) + sl; q < min(qb, q_hi); q += GL)
                *(float4 *)(out + (4 * q - lead)) = *(const float4 *)(cvs + 4 * (q - qa));
        // the at most 3 + 3 elements of the partial pieces at both ends: lanes 0..3 the head, lanes 4..7 the tail
        if (valid && sl < 8) {
            const int e = sl < 4 ? sl : 4 * q_hi - lead + (sl - 4);
            const bool mine = sl < 4 ? (sl < 4 * q_lo - lead) : (e < NEL);
            if (mine) {
                const int q = (e + lead) >> 2;
                float v = (e % 3 == 1) ? one : 0.0f;
                if ((unsigned)(q - qa) < (unsigned)(qb - qa)) v = cvs[e + lead - 4 * qa];
                out[e] = v;
            }
        }
    } else if (planes) {
        // The channel-major layouts.  The observation is the wall pattern almost everywhere (441 canvas pixels, 121 board cells): fill an LDS
        // canvas with the pattern, scatter the board cells into their rotated positions, stream the canvas out
        // with 16-byte stores.  The canvas is shifted by `lead` floats so that LDS and HBM addresses are congruent
        // mod 16 bytes whatever the row's position in the output array.
        float *out = planes + (size_t)pi * NEL;
        float *canvas = (float *)(g + lds_per_obs(L));
        const int lead = (layout == SNK_NCHW_BF16) ? 0 : (int)(((size_t)pi * NEL) & 3);
        float *cv = canvas + lead;                     // cv[e] <-> out[e]
        const int nvec = (NEL + lead + 3) / 4;
        if (valid)
        {
            // channel (element % 3) of the float4's first element: one division here, then it advances by 4 GL mod 3 per
            // iteration (the per-element `% 3` cost four quarter-rate multiplies per float4)
            int ch0 = (4 * sl - lead + 3) % 3;
            constexpr int ADV = (4 * GL) % 3;
            for (int q = sl; q < nvec; q += GL) {
                float4 v;
                float *pv = (float *)&v;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int e = 4 * q - lead + t;
                    int ch = ch0 + t;                  // 0 .. 5
                    ch -= ch >= 3 ? 3 : 0;
                    float d = 0.0f;
                    if (live) d = (layout == SNK_NHWC_F32) ? ((ch == 1) ? 1.0f : 0.0f)
                                                           : ((e >= NPIX && e < 2 * NPIX) ? 1.0f : 0.0f);   // WALL (game.py:4,219)
                    pv[t] = d;
                }
                *(float4 *)(canvas + 4 * q) = v;
                ch0 += ADV;
                ch0 -= ch0 >= 3 ? 3 : 0;
            }
        }
        GAME_SYNC();
        if (live)
            for (int c = sl; c < NC; c += GL) {
                const int y = c / WW, x = c - y * WW;
                const int si = y - hy + (HH - 1), sj = x - hx + (WW - 1);
                int i, j;                              // numpy.rot90(grid, k): out[i][j] = grid[si][sj], inverted
                if (k == 0) { i = si; j = sj; }
                else if (k == 1) { i = N - 1 - sj; j = si; }
                else if (k == 2) { i = N - 1 - si; j = N - 1 - sj; }
                else { i = sj; j = N - 1 - si; }
                const int pp = i * N + j;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    cv[(layout == SNK_NHWC_F32) ? 3 * pp + ch : ch * NPIX + pp] = cell_val(c, ch);
            }
        GAME_SYNC();
        if (valid && layout == SNK_NCHW_BF16) {            // same values, channel-major, rounded to bf16 (nearest even)
            unsigned short *o16 = (unsigned short *)planes + (size_t)pi * NEL;
            for (int e = sl; e < NEL; e += GL) {
                const uint32_t u = __float_as_uint(canvas[e]);
                o16[e] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);      // observation values are finite
            }
        } else if (valid)
            for (int q = sl; q < nvec; q += GL) {
                const int e0 = 4 * q - lead;
                const float4 v = *(const float4 *)(canvas + 4 * q);
                if (e0 >= 0 && e0 + 3 < NEL) {
                    *(float4 *)(out + e0) = v;
                } else {
                    const float *pv = (const float *)&v;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (e0 + t >= 0 && e0 + t < NEL) out[e0 + t] = pv[t];
                }
            }
    }
    if (row_active && valid && sl == 0) row_active[pi] = (live && sub_active[slot]) ? 1 : 0;
    if (mask_out && valid && sl < 3) {
        uint8_t b = 1;
        if (live) {
            const int ad = (k + 3 + sl) & 3;           // left / straight / right of the heading
            const int y = hy + (ad == 2) - (ad == 0), x = hx + (ad == 1) - (ad == 3);
            float v = 1.0f;
            if (y >= 0 && y < HH && x >= 0 && x < WW) v = cell_val(y * WW + x, 1);
            b = legacy_mask ? ((double)v >= 0.04) : (v >= 0.04f);          // alpha_nnet.py:75-76
        }
        mask_out[(size_t)pi * 3 + sl] = b;
    }
    if (key_out) {
        uint64_t lo = 0, hi = 0;
        if (live) {
            for (int c = sl; c < NC; c += GL) {
                const uint32_t b0 = __float_as_uint(cell_val(c, 0)), b1 = __float_as_uint(cell_val(c, 1)),
                               b2 = __float_as_uint(cell_val(c, 2));
                if (b0 == 0u && b1 == 0x3F800000u && b2 == 0u) continue;   // indistinguishable from a wall
                const int y = c / WW, x = c - y * WW;
                const int si = y - hy + (HH - 1), sj = x - hx + (WW - 1);
                int i, j;                                // inverse of the rot90 map above
                if (k == 0) { i = si; j = sj; }
                else if (k == 1) { i = N - 1 - sj; j = si; }
                else if (k == 2) { i = N - 1 - si; j = N - 1 - sj; }
                else { i = sj; j = N - 1 - si; }
                const uint64_t p = (uint64_t)(i * N + j);
                uint64_t xk = sm64((p << 32) | b0);
                xk = sm64(xk ^ (((uint64_t)b1 << 32) | b2));
                lo += xk;
                hi += sm64(xk ^ 0xD6E8FEB86659FD93ull);
            }
        }
        lo = group_sum_u64<GL>(lo);
        hi = group_sum_u64<GL>(hi);
        if (valid && sl == 0) { key_out[2 * (size_t)pi] = lo; key_out[2 * (size_t)pi + 1] = hi; }
    }
    GAME_SYNC();                                           // the LDS region is reused by the next observation
    }
}

// ------------------------------------------------------------------------------------------
// deterministic stream compaction: indices of non-zero flags, ascending
// ------------------------------------------------------------------------------------------
#define CMP_THREADS 256
#define CMP_ITEMS 8
#define CMP_TILE (CMP_THREADS * CMP_ITEMS)

__device__ static inline int block_exclusive_scan_256(int v, int *total, int *sh /* >= 8 ints */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sh[w];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(CMP_THREADS) void k_cmp_count(const uint8_t *__restrict__ flags, int n, int32_t *__restrict__ tile_sums)
{
    __shared__ int sh[8];
    const int base = blockIdx.x * CMP_TILE + threadIdx.x * CMP_ITEMS;
    int c = 0;
    for (int q = 0; q < CMP_ITEMS; ++q) { const int i = base + q; if (i < n && flags[i]) ++c; }
    int total;
    block_exclusive_scan_256(c, &total, sh);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of tile sums (any count, processed 256 at a time)
__global__ __launch_bounds__(CMP_THREADS) void k_cmp_scan(int32_t *__restrict__ tile_sums, int n_tiles, int32_t *__restrict__ count)
{
    __shared__ int sh[8];
    int carry = 0;
    for (int b = 0; b < n_tiles; b += CMP_THREADS) {
        const int i = b + threadIdx.x;
        const int v = i < n_tiles ? tile_sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan_256(v, &total, sh);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(CMP_THREADS) void k_cmp_scatter(const uint8_t *__restrict__ flags, int n, const int32_t *__restrict__ tile_offs,
                                                             int32_t *__restrict__ out)
{
    __shared__ int sh[8];
    const int base = blockIdx.x * CMP_TILE + threadIdx.x * CMP_ITEMS;
    int c = 0;
    uint8_t f[CMP_ITEMS];
    for (int q = 0; q < CMP_ITEMS; ++q) { const int i = base + q; f[q] = (i < n) ? flags[i] : 0; if (f[q]) ++c; }
    int total;
    int pos = tile_offs[blockIdx.x] + block_exclusive_scan_256(c, &total, sh);
    for (int q = 0; q < CMP_ITEMS; ++q) if (f[q]) out[pos++] = base + q;
}

// the same compaction in ONE launch for lists a single block walks quickly (the rollout ticks of small and medium batches are
// bound by their launch count): every thread takes a run of consecutive flags; same output order as the three-kernel form
#define CMP_ONE_MAX (CMP_THREADS * 64)
__global__ __launch_bounds__(CMP_THREADS) void k_cmp_one(const uint8_t *__restrict__ flags, int n, int32_t *__restrict__ out,
                                                         int32_t *__restrict__ count)
{
    __shared__ int sh[8];
    const int per = (n + CMP_THREADS - 1) / CMP_THREADS, base = threadIdx.x * per;
    int c = 0;
    for (int q = 0; q < per; ++q) { const int i = base + q; if (i < n && flags[i]) ++c; }
    int total;
    int pos = block_exclusive_scan_256(c, &total, sh);
    for (int q = 0; q < per; ++q) { const int i = base + q; if (i < n && flags[i]) out[pos++] = i; }
    if (threadIdx.x == 0) *count = total;
}

// ------------------------------------------------------------------------------------------ C ABI
// square boards of 5x5 (the eight standard start cells are distinct from there on, game.py:25-29) to 19x19 (SNK_MAX_CELLS);
// the reference's observation is a rot90 of a (2H-1)x(2W-1) canvas, so only square boards batch (game.py:257)
static bool supported_board(int H, int W) { return H == W && H >= 5 && H * W <= SNK_MAX_CELLS; }

#define DISPATCH_BOARD(L, CALL)                                                 \
    do {                                                                        \
        if ((L).H == 11) { constexpr int BH = 11, BW = 11; CALL; }              \
        else if ((L).H == 7) { constexpr int BH = 7, BW = 7; CALL; }            \
        else if ((L).H == 19) { constexpr int BH = 19, BW = 19; CALL; }         \
        else if ((L).NC <= 255) { constexpr int BH = 0, BW = 0; CALL; }         \
        else { constexpr int BH = -1, BW = -1; CALL; }                          \
    } while (0)

extern "C" int snk_engine_create(snk_engine **out, int n_slots, int H, int W, int S, int health_dec,
                                 double food_spawn_chance, uint64_t seed, int device)
{
    SNK_REQUIRE(out != nullptr, "snk_engine_create: out is NULL");
    SNK_REQUIRE(supported_board(H, W), "snk_engine_create: unsupported board %dx%d (square boards from 5x5 to 19x19)", H, W);
    SNK_REQUIRE(S >= 2 && S <= SNK_MAX_SNAKES, "snk_engine_create: snake count %d outside 2..8", S);
    SNK_REQUIRE(n_slots > 0, "snk_engine_create: n_slots must be positive");
    SNK_CHECK_HIP(hipSetDevice(device));
    snk_engine *e = new snk_engine();
    e->L = make_layout(H, W, S);
    e->n_slots = n_slots;
    e->health_dec = health_dec;
    e->food_chance = food_spawn_chance;
    e->seed = seed;
    e->next_uid = 1;
    e->device = device;
    e->d_state = nullptr;
    e->d_scratch64 = nullptr;
    hipError_t err = hipMalloc((void **)&e->d_state, (size_t)n_slots * e->L.stride);
    if (err == hipSuccess) err = hipMalloc((void **)&e->d_scratch64, 8 * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMemset(e->d_state, 0, (size_t)n_slots * e->L.stride);
    if (err != hipSuccess) {
        snk_set_error("snk_engine_create: device allocation of %zu bytes failed: %s", (size_t)n_slots * e->L.stride, hipGetErrorString(err));
        if (e->d_state) (void)hipFree(e->d_state);
        if (e->d_scratch64) (void)hipFree(e->d_scratch64);
        delete e;
        return -2;
    }
    *out = e;
    return 0;
}

extern "C" int snk_engine_destroy(snk_engine *e)
{
    if (!e) return 0;
    (void)hipFree(e->d_state);
    (void)hipFree(e->d_scratch64);
    delete e;
    return 0;
}

extern "C" int snk_engine_info(const snk_engine *e, int *n_slots, int *H, int *W, int *S, int *slot_bytes)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_info: engine is NULL");
    if (n_slots) *n_slots = e->n_slots;
    if (H) *H = e->L.H;
    if (W) *W = e->L.W;
    if (S) *S = e->L.S;
    if (slot_bytes) *slot_bytes = e->L.stride;
    return 0;
}

extern "C" int snk_engine_raw(const snk_engine *e, void **d_base, int *slot_bytes)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_raw: engine is NULL");
    if (d_base) *d_base = e->d_state;
    if (slot_bytes) *slot_bytes = e->L.stride;
    return 0;
}

extern "C" int snk_engine_set_params(snk_engine *e, int health_dec, double food_spawn_chance)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_set_params: engine is NULL");
    e->health_dec = health_dec;
    e->food_chance = food_spawn_chance;
    return 0;
}

static inline int wave_grid(int n) { return (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; }

extern "C" int snk_engine_reset(snk_engine *e, const int32_t *d_slots, int n, const uint8_t *d_init_tape, void *stream)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_reset: engine is NULL");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_reset: n=%d exceeds %d slots", n, e->n_slots);
    if (n == 0) return 0;
    const Layout L = e->L;
    const size_t lds = (size_t)WAVES_PER_BLOCK * lds_per_wave(L);
    const uint32_t uid_base = e->next_uid;
    e->next_uid += (uint32_t)n;
    DISPATCH_BOARD(L, (k_reset<BH, BW><<<wave_grid(n), BLOCK_THREADS, lds, (hipStream_t)stream>>>(
        e->d_state, L, d_slots, n, d_init_tape, uid_base, (uint32_t)e->seed, (uint32_t)(e->seed >> 32))));
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_engine_clone(const snk_engine *src, const int32_t *d_src_slots, int n, snk_engine *dst,
                                const int32_t *d_dst_slots, int fanout, void *stream)
{
    SNK_REQUIRE(src && dst, "snk_engine_clone: engine is NULL");
    SNK_REQUIRE(src->L.H == dst->L.H && src->L.W == dst->L.W && src->L.S == dst->L.S, "snk_engine_clone: geometry mismatch");
    SNK_REQUIRE(fanout >= 1, "snk_engine_clone: fanout must be >= 1");
    SNK_REQUIRE(n >= 0 && (d_src_slots || n <= src->n_slots), "snk_engine_clone: n=%d exceeds the source's %d slots", n, src->n_slots);
    SNK_REQUIRE(d_dst_slots || (long long)n * fanout <= dst->n_slots, "snk_engine_clone: %lld copies exceed the destination's %d slots",
                (long long)n * fanout, dst->n_slots);
    if (n == 0) return 0;
    k_clone<<<wave_grid(n), BLOCK_THREADS, 0, (hipStream_t)stream>>>(src->d_state, dst->d_state, src->L, d_src_slots, n, d_dst_slots, fanout);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

static int step_launch(snk_engine *e, const int32_t *d_slots, int n, const uint8_t *d_moves, const int16_t *d_spawn_tape,
                       uint8_t *d_done, int16_t *d_spawned, uint64_t *d_empty, const uint8_t *d_active, void *stream,
                       const int32_t *d_skip = nullptr)
{
    const Layout L = e->L;
    // four games per wavefront where the per-game LDS is small (11x11, 7x7), one per wavefront on 19x19
#define STEP_ARGS e->d_state, L, d_slots, n, d_moves, d_spawn_tape, d_done, d_spawned, d_empty, e->health_dec, e->food_chance, \
                  (uint32_t)e->seed, (uint32_t)(e->seed >> 32), d_active, d_skip
    static const bool wide = getenv("SNK_STEP_FORM") && !strcmp(getenv("SNK_STEP_FORM"), "wide");   // A/B: the lane-group kernel
    if (L.S <= 4 && L.NC <= 255 && !wide) {               // a quad per game, sixteen games per wavefront
        const size_t lds = (size_t)WAVES_PER_BLOCK * 16 * lds_per_game_quad(L);
        const int grid = (n + WAVES_PER_BLOCK * 16 - 1) / (WAVES_PER_BLOCK * 16);
        if (L.H == 11 && L.W == 11 && L.S == 4) k_step_quad<11, 11, 4><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else if (L.H == 7 && L.W == 7 && L.S == 2) k_step_quad<7, 7, 2><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else k_step_quad<0, 0, 0><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
    } else if (L.H <= 11) {
        const size_t lds = (size_t)WAVES_PER_BLOCK * 4 * lds_per_wave(L);
        const int grid = (n + WAVES_PER_BLOCK * 4 - 1) / (WAVES_PER_BLOCK * 4);
        if (L.H == 11 && L.S == 4) k_step<11, 11, 16, 4><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else if (L.H == 11) k_step<11, 11, 16><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else if (L.H == 7) k_step<7, 7, 16><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else k_step<0, 0, 16><<<grid, BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
    } else {
        const size_t lds = (size_t)WAVES_PER_BLOCK * lds_per_wave(L);
        if (L.H == 19) k_step<19, 19, 64><<<wave_grid(n), BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else if (L.NC <= 255) k_step<0, 0, 64><<<wave_grid(n), BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
        else k_step<-1, -1, 64><<<wave_grid(n), BLOCK_THREADS, lds, (hipStream_t)stream>>>(STEP_ARGS);
    }
#undef STEP_ARGS
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_engine_step(snk_engine *e, const int32_t *d_slots, int n, const uint8_t *d_moves,
                               const int16_t *d_spawn_tape, uint8_t *d_done, int16_t *d_spawned,
                               uint64_t *d_empty, void *stream)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_step: engine is NULL");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_step: n=%d exceeds %d slots", n, e->n_slots);
    if (n == 0) return 0;
    SNK_REQUIRE(d_moves != nullptr, "snk_engine_step: d_moves is NULL");
    return step_launch(e, d_slots, n, d_moves, d_spawn_tape, d_done, d_spawned, d_empty, nullptr, stream);
}

extern "C" int snk_engine_step_active(snk_engine *e, const uint8_t *d_active, int n, const uint8_t *d_moves, uint8_t *d_done,
                                      const int32_t *d_skip, void *stream)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_step_active: engine is NULL");
    SNK_REQUIRE(n >= 0 && n <= e->n_slots, "snk_engine_step_active: n=%d exceeds %d slots", n, e->n_slots);
    if (n == 0) return 0;
    SNK_REQUIRE(d_moves != nullptr && d_active != nullptr, "snk_engine_step_active: NULL argument");
    return step_launch(e, nullptr, n, d_moves, nullptr, d_done, nullptr, nullptr, d_active, stream, d_skip);
}

extern "C" int snk_engine_alive(const snk_engine *e, const int32_t *d_slots, int n, uint8_t *d_alive,
                                int32_t *d_n_alive, void *stream)
{
    SNK_REQUIRE(e != nullptr && d_alive != nullptr, "snk_engine_alive: NULL argument");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_alive: n=%d exceeds %d slots", n, e->n_slots);
    if (n == 0) return 0;
    k_alive<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(e->d_state, e->L, d_slots, n, d_alive, d_n_alive);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

static int engine_observe(const snk_engine *e, const int32_t *d_pairs, const int32_t *d_index, int m, int layout, float *d_planes,
                          uint8_t *d_mask, uint64_t *d_key, int legacy_mask, const uint8_t *d_sub_active, uint8_t *d_row_active,
                          void *stream);
extern "C" int snk_engine_observe(const snk_engine *e, const int32_t *d_pairs, int m, int layout,
                                  float *d_planes, uint8_t *d_mask, uint64_t *d_key, int legacy_mask, void *stream)
{
    return engine_observe(e, d_pairs, nullptr, m, layout, d_planes, d_mask, d_key, legacy_mask, nullptr, nullptr, stream);
}
extern "C" int snk_engine_observe_rows(const snk_engine *e, const int32_t *d_pairs, const int32_t *d_index, int m, int layout,
                                       float *d_planes, uint8_t *d_mask, uint64_t *d_key, int legacy_mask,
                                       const uint8_t *d_sub_active, uint8_t *d_row_active, void *stream)
{
    SNK_REQUIRE((d_row_active == nullptr) == (d_sub_active == nullptr), "snk_engine_observe_rows: d_row_active needs d_sub_active (and the other way round)");
    return engine_observe(e, d_pairs, d_index, m, layout, d_planes, d_mask, d_key, legacy_mask, d_sub_active, d_row_active, stream);
}
static int engine_observe(const snk_engine *e, const int32_t *d_pairs, const int32_t *d_index, int m, int layout, float *d_planes,
                          uint8_t *d_mask, uint64_t *d_key, int legacy_mask, const uint8_t *d_sub_active, uint8_t *d_row_active,
                          void *stream)
{
    SNK_REQUIRE(e != nullptr, "snk_engine_observe: engine is NULL");
    SNK_REQUIRE(layout == SNK_NHWC_F32 || layout == SNK_NCHW_F32 || layout == SNK_NCHW_BF16, "snk_engine_observe: unknown layout %d", layout);
    if (m <= 0) return 0;                               // an empty request is a no-op (its buffers may be NULL)
    SNK_REQUIRE(d_pairs != nullptr, "snk_engine_observe: d_pairs is NULL");
    const Layout L = e->L;
    // Without planes (mask and key only: every rollout state of the MCTS) the work is the per-observation preamble and four
    // observations share a wavefront (11x11, 7x7: their LDS is 1.2 KB each); with planes the kernel is a 5.3 KB store
    // stream per observation and one wavefront per observation keeps more stores in flight (measured: 61 us against 105).
    if (L.H <= 11 && !d_planes) {
        constexpr int WPB = 4;
        const size_t lds = (size_t)WPB * 4 * lds_per_obs(L);
        const int grid = (m + WPB * 4 - 1) / (WPB * 4);
        if (L.H == 11) k_observe<11, 11, 16, WPB><<<grid, WPB * 64, lds, (hipStream_t)stream>>>(e->d_state, L, d_pairs, m, layout, d_planes, d_mask, d_key, legacy_mask, 1, d_index, d_sub_active, d_row_active);
        else if (L.H == 7) k_observe<7, 7, 16, WPB><<<grid, WPB * 64, lds, (hipStream_t)stream>>>(e->d_state, L, d_pairs, m, layout, d_planes, d_mask, d_key, legacy_mask, 1, d_index, d_sub_active, d_row_active);
        else k_observe<0, 0, 16, WPB><<<grid, WPB * 64, lds, (hipStream_t)stream>>>(e->d_state, L, d_pairs, m, layout, d_planes, d_mask, d_key, legacy_mask, 1, d_index, d_sub_active, d_row_active);
    } else {
        const size_t lds = (size_t)WAVES_PER_BLOCK * (!d_planes ? lds_per_obs(L) : layout == SNK_NHWC_F32 ? lds_per_wave_win(L) : lds_per_wave_obs(L));
        // several observations per wavefront once the request fills the chip's wave slots (256 CUs x 32) a few times over
        static const int reps_env = getenv("SNK_OBS_REPS") ? atoi(getenv("SNK_OBS_REPS")) : 0;
        const int reps = reps_env > 0 ? reps_env : (d_planes && m >= 4 * 8192) ? 2 : 1;
        DISPATCH_BOARD(L, (k_observe<BH, BW, 64, WAVES_PER_BLOCK><<<wave_grid((m + reps - 1) / reps), BLOCK_THREADS, lds, (hipStream_t)stream>>>(
            e->d_state, L, d_pairs, m, layout, d_planes, d_mask, d_key, legacy_mask, reps, d_index, d_sub_active, d_row_active)));
    }
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- host views --------------------------------------------------------------------------
static void raw_to_canonical(const Layout &L, const uint8_t *raw, snk_game_state *o)
{
    memset(o, 0, sizeof(*o));
    o->H = L.H; o->W = L.W; o->S = L.S;
    o->uid = *(const uint32_t *)(raw + L.uid_off);
    const SnakeMeta *meta = (const SnakeMeta *)(raw + L.meta_off);
    const uint64_t *food = (const uint64_t *)(raw + L.food_off);
    const uint32_t *cnt = (const uint32_t *)(raw + L.cnt_off);
    const int8_t *rew = (const int8_t *)(raw + L.rew_off);
    for (int s = 0; s < SNK_MAX_SNAKES; ++s)
        for (int k = 0; k < SNK_MAX_NODES; ++k) o->nodes[s][k] = -1;
    for (int s = 0; s < L.S; ++s) {
        o->alive[s] = meta[s].alive;
        o->rewards[s] = rew[s];
        if (!meta[s].alive) continue;
        o->health[s] = meta[s].health;
        o->length[s] = (int16_t)meta[s].len;
        o->dir[s] = meta[s].dir;
        for (int k = 0; k < meta[s].len && k < SNK_MAX_NODES; ++k) {
            const int idx = (meta[s].tail + meta[s].len - 1 - k) & L.cap_mask;      // head first
            const uint8_t *r = raw + s * L.ring_bytes;
            o->nodes[s][k] = (L.cell_bytes == 1) ? (int16_t)r[idx] : (int16_t)((const uint16_t *)r)[idx];
        }
    }
    for (int c = 0; c < L.NC; ++c) o->food[c] = (uint8_t)((food[c >> 6] >> (c & 63)) & 1ull);
    for (int q = 0; q < 6; ++q) o->counters[q] = (int32_t)cnt[q];
}

static int canonical_to_raw(const Layout &L, const snk_game_state *in, uint8_t *raw, int ring_start = 0)
{
    memset(raw, 0, (size_t)L.stride);
    if (in->H != L.H || in->W != L.W || in->S != L.S) return -1;
    SnakeMeta *meta = (SnakeMeta *)(raw + L.meta_off);
    uint64_t *food = (uint64_t *)(raw + L.food_off);
    uint32_t *cnt = (uint32_t *)(raw + L.cnt_off);
    int8_t *rew = (int8_t *)(raw + L.rew_off);
    *(uint32_t *)(raw + L.uid_off) = in->uid;
    for (int s = 0; s < L.S; ++s) {
        rew[s] = in->rewards[s];
        if (!in->alive[s]) continue;
        const int len = in->length[s];
        if (len < 1 || len > L.cap || len > SNK_MAX_NODES) return -2;
        meta[s].alive = 1; meta[s].tail = (uint16_t)(ring_start & L.cap_mask); meta[s].len = (uint16_t)len;
        meta[s].health = in->health[s]; meta[s].dir = in->dir[s];
        uint8_t *r = raw + s * L.ring_bytes;
        for (int k = 0; k < len; ++k) {
            const int cell = in->nodes[s][len - 1 - k];                          // ring index tail + k counts from the tail
            if (cell < 0 || cell >= L.NC) return -3;
            const int at = (ring_start + k) & L.cap_mask;
            if (L.cell_bytes == 1) r[at] = (uint8_t)cell; else ((uint16_t *)r)[at] = (uint16_t)cell;
        }
    }
    for (int c = 0; c < L.NC; ++c) if (in->food[c]) food[c >> 6] |= 1ull << (c & 63);
    for (int q = 0; q < 6; ++q) cnt[q] = (uint32_t)in->counters[q];
    return 0;
}

extern "C" int snk_engine_export_sync(const snk_engine *e, const int32_t *h_slots, int n, snk_game_state *h_out)
{
    SNK_REQUIRE(e != nullptr && h_out != nullptr, "snk_engine_export_sync: NULL argument");
    SNK_REQUIRE(n >= 0 && (h_slots || n <= e->n_slots), "snk_engine_export_sync: n=%d exceeds %d slots", n, e->n_slots);
    const Layout &L = e->L;
    SNK_CHECK_HIP(hipDeviceSynchronize());
    std::vector<uint8_t> raw((size_t)L.stride);
    if (!h_slots) {
        std::vector<uint8_t> all((size_t)n * L.stride);
        if (n) SNK_CHECK_HIP(hipMemcpy(all.data(), e->d_state, all.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) raw_to_canonical(L, all.data() + (size_t)i * L.stride, &h_out[i]);
        return 0;
    }
    for (int i = 0; i < n; ++i) {
        SNK_REQUIRE(h_slots[i] >= 0 && h_slots[i] < e->n_slots, "snk_engine_export_sync: slot %d out of range", h_slots[i]);
        SNK_CHECK_HIP(hipMemcpy(raw.data(), e->d_state + (size_t)h_slots[i] * L.stride, (size_t)L.stride, hipMemcpyDeviceToHost));
        raw_to_canonical(L, raw.data(), &h_out[i]);
    }
    return 0;
}

static int engine_import(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in, int ring_start);
extern "C" int snk_engine_import_sync(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in)
{
    return engine_import(e, h_slots, n, h_in, 0);
}
// the same games with every snake's ring laid out from index ring_start on (tail = ring_start mod cap): the state a long game reaches
// -- after cap ticks a live segment straddles the ring's end -- without playing it (tests: the wrap of k_step's and k_observe's ring
// arithmetic).  Export gives back what was imported whatever ring_start is.
extern "C" int snk_engine_import_at_sync(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in, int ring_start)
{
    SNK_REQUIRE(ring_start >= 0, "snk_engine_import_at_sync: ring_start %d", ring_start);
    return engine_import(e, h_slots, n, h_in, ring_start);
}
static int engine_import(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in, int ring_start)
{
    SNK_REQUIRE(e != nullptr && h_in != nullptr, "snk_engine_import_sync: NULL argument");
    SNK_REQUIRE(n >= 0 && (h_slots || n <= e->n_slots), "snk_engine_import_sync: n=%d exceeds %d slots", n, e->n_slots);
    const Layout &L = e->L;
    SNK_CHECK_HIP(hipDeviceSynchronize());
    std::vector<uint8_t> raw((size_t)L.stride);
    for (int i = 0; i < n; ++i) {
        const int slot = h_slots ? h_slots[i] : i;
        SNK_REQUIRE(slot >= 0 && slot < e->n_slots, "snk_engine_import_sync: slot %d out of range", slot);
        const int rc = canonical_to_raw(L, &h_in[i], raw.data(), ring_start);
        SNK_REQUIRE(rc == 0, "snk_engine_import_sync: game %d is malformed (code %d)", i, rc);
        SNK_CHECK_HIP(hipMemcpy(e->d_state + (size_t)slot * L.stride, raw.data(), (size_t)L.stride, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int snk_engine_sum_counters_sync(const snk_engine *e, const int32_t *d_slots, int n, int64_t *h_out)
{
    SNK_REQUIRE(e != nullptr && h_out != nullptr, "snk_engine_sum_counters_sync: NULL argument");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_sum_counters_sync: n=%d exceeds %d slots", n, e->n_slots);
    SNK_CHECK_HIP(hipMemset(e->d_scratch64, 0, 8 * sizeof(unsigned long long)));
    if (n > 0) {
        int blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        k_sum_counters<<<blocks, 256>>>(e->d_state, e->L, d_slots, n, e->d_scratch64);
        SNK_CHECK_HIP(hipGetLastError());
    }
    unsigned long long tmp[6];
    SNK_CHECK_HIP(hipMemcpy(tmp, e->d_scratch64, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int q = 0; q < 6; ++q) h_out[q] = (int64_t)tmp[q];
    return 0;
}

extern "C" int snk_compact_scratch_elems(int n);

// (slot, snake id) of every alive snake, games in the given order, ids ascending: Game.get_ids over a batch
__global__ void k_ids_from_index(const int32_t *__restrict__ idx, const int32_t *__restrict__ count, const int32_t *__restrict__ slots,
                                 int S, int32_t *__restrict__ pairs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *count) return;
    const int f = idx[i], gi = f / S;
    pairs[2 * i] = slots ? slots[gi] : gi;
    pairs[2 * i + 1] = f - gi * S;
}

extern "C" int snk_engine_ids(const snk_engine *e, const int32_t *d_slots, int n, int32_t *d_pairs, int32_t *d_count,
                              uint8_t *d_alive_scratch, int32_t *d_scratch, void *stream)
{
    SNK_REQUIRE(e && d_count, "snk_engine_ids: NULL argument");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_ids: n=%d exceeds %d slots", n, e->n_slots);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { SNK_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return 0; }
    SNK_REQUIRE(d_pairs && d_alive_scratch && d_scratch, "snk_engine_ids: NULL argument");
    const int m = n * e->L.S;
    k_alive<<<(n + 255) / 256, 256, 0, st>>>(e->d_state, e->L, d_slots, n, d_alive_scratch, nullptr);
    int32_t *idx = d_scratch + snk_compact_scratch_elems(m);           // scratch: [tile sums | index list]
    const int tiles = (m + CMP_TILE - 1) / CMP_TILE;
    k_cmp_count<<<tiles, CMP_THREADS, 0, st>>>(d_alive_scratch, m, d_scratch);
    k_cmp_scan<<<1, CMP_THREADS, 0, st>>>(d_scratch, tiles, d_count);
    k_cmp_scatter<<<tiles, CMP_THREADS, 0, st>>>(d_alive_scratch, m, d_scratch, idx);
    k_ids_from_index<<<(m + 255) / 256, 256, 0, st>>>(idx, d_count, d_slots, e->L.S, d_pairs);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_compact_scratch_elems(int n) { return (n + CMP_TILE - 1) / CMP_TILE + 1; }

extern "C" int snk_compact_flags(const uint8_t *d_flags, int n, int32_t *d_out, int32_t *d_count,
                                 int32_t *d_scratch, void *stream)
{
    SNK_REQUIRE(d_count != nullptr, "snk_compact_flags: d_count is NULL");
    SNK_REQUIRE(n >= 0, "snk_compact_flags: negative n");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { SNK_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return 0; }
    SNK_REQUIRE(d_flags && d_out && d_scratch, "snk_compact_flags: NULL argument");
    if (n <= CMP_ONE_MAX) {
        k_cmp_one<<<1, CMP_THREADS, 0, st>>>(d_flags, n, d_out, d_count);
        SNK_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int tiles = (n + CMP_TILE - 1) / CMP_TILE;
    k_cmp_count<<<tiles, CMP_THREADS, 0, st>>>(d_flags, n, d_scratch);
    k_cmp_scan<<<1, CMP_THREADS, 0, st>>>(d_scratch, tiles, d_count);
    k_cmp_scatter<<<tiles, CMP_THREADS, 0, st>>>(d_flags, n, d_scratch, d_out);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
