/*
 * snake_engine.h -- C ABI of the MI355X-native batched Battlesnake self-play engine
 * (libsnake_engine.so, built from alphasnake-zero_amd/csrc/ for gfx950).
 *
 * The reference (Fool-Yang/AlphaSnake-Zero) has no FFI: its hot path is a set of Python
 * classes.  This header is the drop-in boundary a binding for that path needs; every entry
 * point names the reference code it replaces (paths relative to /root/reference/code/utils/).
 * The Python mirror of the reference's class API (alphasnake-zero_amd/utils/*.py) calls
 * exactly these symbols through ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE pointer (HBM), h_* is HOST memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *     on that stream unless the name ends in _sync or the call returns host data;
 *   - return value: 0 = ok, <0 = error (snk_last_error() gives the text).  Like the reference
 *     (single-threaded, not re-entrant) an engine must not be used from two threads at once;
 *   - a "slot" is one game (one board) living in HBM; snakes are addressed by their id 0..S-1,
 *     the reference's `snakes` list order is "alive ids ascending" (game.py:191 keeps order).
 *   - relative moves: 0 left, 1 straight, 2 right (game.py:92); absolute headings: 0 up, 1 right,
 *     2 down, 3 left (game.py:330-342); board cell index = y*W + x, (y, x) as in game.py.
 */
#ifndef SNAKE_ENGINE_H
#define SNAKE_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNK_MAX_SNAKES 8
#define SNK_MAX_CELLS 361
#define SNK_MAX_NODES 384

typedef struct snk_engine snk_engine;

/* canonical host-side picture of one game (export / import; the golden-vector format) */
typedef struct {
    int32_t H, W, S;
    uint32_t uid;
    uint8_t alive[SNK_MAX_SNAKES];
    int16_t health[SNK_MAX_SNAKES];
    int16_t length[SNK_MAX_SNAKES];                 /* node count == Snake.length (game.py:306) */
    uint8_t dir[SNK_MAX_SNAKES];                    /* Game.last_moves[id] (game.py:30,93) */
    int16_t nodes[SNK_MAX_SNAKES][SNK_MAX_NODES];   /* head..tail cell indices, -1 padded */
    uint8_t food[SNK_MAX_CELLS];                    /* Game.food as 0/1 per cell */
    int8_t rewards[SNK_MAX_SNAKES];                 /* Game.rewards: 0 None, +1, -1 */
    int32_t counters[6];                            /* wall, body, head collisions, starvation,
                                                       food_eaten, game_length (game.py:56-61) */
} snk_game_state;

const char *snk_last_error(void);
int snk_version(void);

/* ---- engine lifetime -------------------------------------------------------------------
 * Owns n_slots games of an H x W board with S snakes in HBM (struct-of-rings layout).
 * Replaces the dict of Game objects built at mp_game_runner.py:13 / agent.py:43-50.
 * Supported (H, W, S): (11,11,4) (7,7,2) (19,19,8) and any S in 2..8 for those boards.       */
int snk_engine_create(snk_engine **out, int n_slots, int H, int W, int S, int health_dec,
                      double food_spawn_chance, uint64_t seed, int device);
int snk_engine_destroy(snk_engine *e);
int snk_engine_info(const snk_engine *e, int *n_slots, int *H, int *W, int *S, int *slot_bytes);
/* raw device base of the slot array + stride, for zero-copy consumers (records arena, tests) */
int snk_engine_raw(const snk_engine *e, void **d_base, int *slot_bytes);
int snk_engine_set_params(snk_engine *e, int health_dec, double food_spawn_chance);

/* ---- Game.__init__ (game.py:13-61) ------------------------------------------------------
 * (Re)initialises games.  d_slots: int32[n] or NULL (= slots 0..n-1).
 * d_init_tape: NULL -> start cells, headings and food diagonals are drawn on device from a
 * counter-based Philox stream keyed by (seed, game uid); else uint8[n][3][S] =
 * {index into the 8 standard start cells, heading, diagonal 0..3} per snake: the recorded
 * outcome of sample()/choice() at game.py:25-30,46 (parity runs).                            */
int snk_engine_reset(snk_engine *e, const int32_t *d_slots, int n, const uint8_t *d_init_tape,
                     void *stream);

/* ---- Game.subgame (game.py:266-276) -----------------------------------------------------
 * dst[d_dst_slots[i*fanout + j]] = deep copy of src[d_src_slots[i]], j < fanout; counters are
 * zeroed, rewards copied (game.py:275).  NULL slot arrays mean identity / i*fanout + j.
 * src and dst may be the same engine; board geometry must match.                             */
int snk_engine_clone(const snk_engine *src, const int32_t *d_src_slots, int n, snk_engine *dst,
                     const int32_t *d_dst_slots, int fanout, void *stream);

/* ---- Game.tic (game.py:87-205) ----------------------------------------------------------
 * One env step for n games, one wavefront per game.
 * d_moves: uint8[n][S] relative moves indexed by snake id (entries of dead snakes ignored).
 * d_spawn_tape: NULL -> food spawn decided on device (Philox; chance = food_spawn_chance,
 *   uniform choice among empty cells, game.py:130-138); else int16[n]: cell to spawn or -1
 *   (the recorded outcome; parity runs).
 * d_done (optional): uint8[n], 1 when the game has ended (tic returned the rewards list).
 * d_spawned (optional): int16[n] the cell that received food or -1.
 * d_empty (optional): uint64[n][ceil(H*W/64)] bit mask of Game.empty_positions at spawn time.
 * Games that have already ended are left untouched (done stays 1).                            */
int snk_engine_step(snk_engine *e, const int32_t *d_slots, int n, const uint8_t *d_moves,
                    const int16_t *d_spawn_tape, uint8_t *d_done, int16_t *d_spawned,
                    uint64_t *d_empty, void *stream);

/* ---- Game.get_ids / alive bookkeeping (game.py:76-77) -----------------------------------
 * d_alive: uint8[n][S] (1 = snake alive), d_n_alive (optional): int32[n].                     */
int snk_engine_alive(const snk_engine *e, const int32_t *d_slots, int n, uint8_t *d_alive,
                     int32_t *d_n_alive, void *stream);

/* ---- Game.make_state / get_states (game.py:215-257, 68-69) + AlphaNNet.v's obstacle test
 *      (alpha_nnet.py:63-76) + the transposition key (agent.py:175) -------------------------
 * d_pairs: int32[m][2] = (slot, snake id) of the observations wanted, any order.
 * layout: SNK_NHWC_F32 writes the reference's exact bytes ((2H-1) x (2W-1) x 3 float32, rotated
 *   so the snake faces up); SNK_NCHW_F32 the same values channel-major.
 * d_planes (optional): float[m][...] observation planes.
 * d_mask (optional): uint8[m][3] 1 = left/straight/right blocked (obstacle test on channel 1;
 *   legacy_mask != 0 selects the float64 compare of the reference's pinned NumPy 1.18).
 * d_key (optional): uint64[m][2] 128-bit digest of the observation bytes (oracle/obs_key.py).
 * A pair naming a dead snake yields zero planes, mask 1,1,1 and key 0,0.                      */
enum { SNK_NHWC_F32 = 0, SNK_NCHW_F32 = 1 };
int snk_engine_observe(const snk_engine *e, const int32_t *d_pairs, int m, int layout,
                       float *d_planes, uint8_t *d_mask, uint64_t *d_key, int legacy_mask,
                       void *stream);

/* ---- host views (goldens, Game.snakes / .food / .rewards accessors, Game.draw) -----------
 * Synchronous.  h_slots: host int32[n] or NULL.                                               */
int snk_engine_export_sync(const snk_engine *e, const int32_t *h_slots, int n, snk_game_state *h_out);
int snk_engine_import_sync(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in);

/* ---- MPGameRunner's log counters (mp_game_runner.py:54-60, 71-76) ------------------------
 * Sums the six per-game counters over the given slots into h_out[6] (int64). Synchronous.     */
int snk_engine_sum_counters_sync(const snk_engine *e, const int32_t *d_slots, int n, int64_t *h_out);

/* ---- stream compaction helper ------------------------------------------------------------
 * d_out[0..count) = indices i (ascending) with d_flags[i] != 0; *d_count = count.
 * d_scratch: int32[snk_compact_scratch_elems(n)].  Deterministic (scan based).                */
int snk_compact_scratch_elems(int n);
int snk_compact_flags(const uint8_t *d_flags, int n, int32_t *d_out, int32_t *d_count,
                      int32_t *d_scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SNAKE_ENGINE_H */
