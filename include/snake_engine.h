/*
 * snake_engine.h -- C ABI of the MI355X-native batched Battlesnake self-play engine
 * (libsnake_engine.so, built from alphasnake-zero_amd/csrc/ for gfx950).
 *
 * The reference (Fool-Yang/AlphaSnake-Zero) has no FFI: its hot path is a set of Python
 * classes.  This header is the drop-in boundary a binding for that path needs; every entry
 * point names the reference code it replaces (paths relative to /root/reference/code/utils/).
 * The Python mirror of the reference's class API (alphasnake-zero_amd/utils/) calls
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
#define SNK_ABI_VERSION 112 /* 100: rounds 1-3; 110: round 5 (the gate argument d_skip of the tick kernels, round 4; the 16-bit
                             * towers' own weight image and rectangle plan, round 5); 111: the training step's deferred batch
                             * norm (fifteen entry points added, snk_conv3x3_stats_partials returns more); 112: round 6,
                             * snk_engine_import_at_sync and snk_engine_observe_rows added: a caller compares it with snk_version() */
int snk_version(void);

/* ---- engine lifetime -------------------------------------------------------------------
 * Owns n_slots games of an H x W board with S snakes in HBM (struct-of-rings layout).
 * Replaces the dict of Game objects built at mp_game_runner.py:13 / agent.py:43-50.
 * Supported: any SQUARE board H = W from 5 to 19 (the reference's observation is a rot90 of a square canvas, game.py:257) with
 * S = 2..8 snakes: tests/ run 5x5/2, 7x7/2, 9x9/3, 11x11/4, 15x15/5, 16x16/6, 19x19/8 against recordings of the reference and
 * random geometries against the C oracle.  11x11, 7x7 and 19x19 have compile-time-geometry kernels; the other sizes run the same
 * code with the geometry read from the layout.                                                  */
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
 * One env step for n games.  Lanes per game, by kernel form: boards of at most 4 snakes and 255 cells
 * (11x11/4, the judged configuration) run k_step_quad -- ONE LANE PER SNAKE, a quad per game, sixteen games per wavefront, the
 * cross-snake rules as DPP quad broadcasts and LDS bit planes; larger boards run k_step with 16 lanes per game (64 from 122
 * cells on, i.e. one wavefront per game on 19x19).  BASELINE.json's wording "one wavefront steps one game" is the 19x19 form;
 * on 11x11/4 it was measured instruction-bound (three quarters of the lanes idle: 73.7 us per 262 144 games) against 54.9 us for
 * the quad form, the time of a plain copy of the same records.
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
/* The lock-step form of MCTSMPGameRunner.run (mp_game_runner.py:104-113): slots 0..n-1, of which only those with
 * d_active[i] != 0 are stepped; the others (sub-games retired by their depth cap) are neither read nor written and
 * report d_done[i] = 0.  Food spawning follows the engine's food_spawn_chance (0 for sub-games, game.py:268).
 * d_skip (optional): the rollout tick's gate -- a device int32; when it is non-zero at launch time the call steps NO game
 * (every game reports d_done = 0).  The search passes the Q-net's range-guard word (snk_conv3x3_f16s_set_guard_word): a tick
 * whose leaf evaluation clamped an activation is not played, the host evaluates again and repeats it.            */
int snk_engine_step_active(snk_engine *e, const uint8_t *d_active, int n, const uint8_t *d_moves,
                           uint8_t *d_done, const int32_t *d_skip, void *stream);

/* ---- Game.get_ids / alive bookkeeping (game.py:76-77) -----------------------------------
 * d_alive: uint8[n][S] (1 = snake alive), d_n_alive (optional): int32[n].                     */
int snk_engine_alive(const snk_engine *e, const int32_t *d_slots, int n, uint8_t *d_alive,
                     int32_t *d_n_alive, void *stream);

/* ids of a batch (mp_game_runner.py:40-42): d_pairs int32[<= n*S][2] = (slot, snake id) of every alive snake, games in the
 * given order, snake ids ascending; *d_count = how many.  Scratch: d_alive_scratch uint8[n*S],
 * d_scratch int32[snk_compact_scratch_elems(n*S) + n*S].                                           */
int snk_engine_ids(const snk_engine *e, const int32_t *d_slots, int n, int32_t *d_pairs, int32_t *d_count,
                   uint8_t *d_alive_scratch, int32_t *d_scratch, void *stream);

/* ---- Game.make_state / get_states (game.py:215-257, 68-69) + AlphaNNet.v's obstacle test
 *      (alpha_nnet.py:63-76) + the transposition key (agent.py:175) -------------------------
 * d_pairs: int32[m][2] = (slot, snake id) of the observations wanted, any order.
 * layout: SNK_NHWC_F32 writes the reference's exact bytes ((2H-1) x (2W-1) x 3 float32, rotated
 *   so the snake faces up); SNK_NCHW_F32 the same values channel-major; SNK_NCHW_BF16 those rounded to bf16
 *   (nearest even), 2 bytes per value (d_planes then points to uint16[m][3][2H-1][2W-1]).
 * d_planes (optional): float[m][...] observation planes.
 * d_mask (optional): uint8[m][3] 1 = left/straight/right blocked (obstacle test on channel 1;
 *   legacy_mask != 0 selects the float64 compare of the reference's pinned NumPy 1.18).
 * d_key (optional): uint64[m][2] 128-bit digest of the observation bytes (oracle/obs_key.py).
 * A pair naming a dead snake yields zero planes, mask 1,1,1 and key 0,0.                      */
enum { SNK_NHWC_F32 = 0, SNK_NCHW_F32 = 1, SNK_NCHW_BF16 = 2 };
int snk_engine_observe(const snk_engine *e, const int32_t *d_pairs, int m, int layout,
                       float *d_planes, uint8_t *d_mask, uint64_t *d_key, int legacy_mask,
                       void *stream);
/* snk_engine_observe for the rows of a rollout tick (MCTSAgent.make_moves, agent.py:161-201; MCTSMPGameRunner.run,
 * mp_game_runner.py:99-103), two launches of bookkeeping folded into the one that holds the record anyway:
 * d_index (optional): int32[m]; output row i observes d_pairs[d_index[i]] (the rows whose keys were new: no gathered copy).
 * d_sub_active + d_row_active (optional, together): uint8[n_slots] / uint8[m]; d_row_active[i] = 1 when the observing snake is
 *   alive AND d_sub_active[its slot] != 0 -- the tick's live rows (what snk_engine_alive + snk_mcts_row_active computed).      */
int snk_engine_observe_rows(const snk_engine *e, const int32_t *d_pairs, const int32_t *d_index, int m, int layout,
                            float *d_planes, uint8_t *d_mask, uint64_t *d_key, int legacy_mask,
                            const uint8_t *d_sub_active, uint8_t *d_row_active, void *stream);

/* ---- host views (goldens, Game.snakes / .food / .rewards accessors, Game.draw) -----------
 * Synchronous.  h_slots: host int32[n] or NULL.                                               */
int snk_engine_export_sync(const snk_engine *e, const int32_t *h_slots, int n, snk_game_state *h_out);
int snk_engine_import_sync(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in);
/* snk_engine_import_sync with every snake's ring buffer (Snake / Node, game.py:329-365) laid out from ring index ring_start on
 * instead of 0: the same game as far as any entry point can tell (export returns h_in), in the memory state a game reaches after
 * ring_start moves -- used by the tests to put live ring segments across the ring's end without playing hundreds of ticks.    */
int snk_engine_import_at_sync(snk_engine *e, const int32_t *h_slots, int n, const snk_game_state *h_in, int ring_start);

/* ---- MPGameRunner's log counters (mp_game_runner.py:54-60, 71-76) ------------------------
 * Sums the six per-game counters over the given slots into h_out[6] (int64). Synchronous.     */
int snk_engine_sum_counters_sync(const snk_engine *e, const int32_t *d_slots, int n, int64_t *h_out);

/* ---- stream compaction helper ------------------------------------------------------------
 * d_out[0..count) = indices i (ascending) with d_flags[i] != 0; *d_count = count.
 * d_scratch: int32[snk_compact_scratch_elems(n)].  Deterministic (scan based).                */
int snk_compact_scratch_elems(int n);
int snk_compact_flags(const uint8_t *d_flags, int n, int32_t *d_out, int32_t *d_count,
                      int32_t *d_scratch, void *stream);

/* ---- AlphaNNet.v (alpha_nnet.py:19-56, 61-76): the Q-net's inference layers ---------------
 * Activations are NHWC float32: d_x[n_images][height][width][C], C = 128 (3 for the stem input).
 * Batch-norm (inference form, alpha_nnet.py:22 ...) is folded by the caller into per-channel
 * scale = gamma / sqrt(var + 1e-3), shift = beta - mean * scale.
 *
 * snk_conv3x3_prepare_weights: Keras Conv2D kernel (kh, kw, cin, cout) = (3,3,128,128) ->
 *   the K-contiguous (tap, cout, cin) layout the MFMA kernel streams.
 * snk_conv3x3_bn_f32: out = [relu]( conv3x3_same(x, w) * scale + shift [+ residual] ), the body of
 *   a residual block (alpha_nnet.py:25-47) as an implicit GEMM on v_mfma_f32_32x32x2_f32.
 * snk_stem_conv_bn_relu_f32: the first layer, 3 -> 128 channels (alpha_nnet.py:21-22); d_w is the
 *   Keras kernel (3,3,3,128) unchanged.
 * snk_head_f32: conv1x1 128 -> 1 + BN + ReLU, Flatten, Dense(128) + ReLU, Dense(3) + tanh
 *   (alpha_nnet.py:49-54); d_fc1_w (height*width, 128) and d_fc2_w (128, 3) are Keras Dense kernels.
 *   d_mask (optional) uint8[n][3]: entries set to 1 overwrite Q with -1.0 (alpha_nnet.py:67-72).  */
int snk_conv3x3_prepare_weights(const float *d_w_hwio, float *d_wT, void *stream);
int snk_conv3x3_bn_f32(const float *d_x, const float *d_wT, const float *d_scale, const float *d_shift,
                       const float *d_residual, float *d_out, int n_images, int height, int width,
                       int relu, void *stream);
/* The same layer in Winograd F(2x2,3x3) form (fp32 throughout, 2.05x fewer MFMA flops on 21x21 images):
 * snk_conv3x3_prepare_weights_winograd: Keras kernel (3,3,128,128) -> U = G g G^T laid out
 *   [16 positions][cin/4][cout][4] (float[16*128*128], evaluated in float64, stored float32). */
int snk_conv3x3_prepare_weights_winograd(const float *d_w_hwio, float *d_U, void *stream);
int snk_conv3x3_bn_f32_winograd(const float *d_x, const float *d_U, const float *d_scale,
                                const float *d_shift, const float *d_residual, float *d_out, int n_images,
                                int height, int width, int relu, void *stream);
/* The same layer at float32 accuracy on the f16 matrix pipe: every float32 operand is carried as hi + lo f16 numbers
 * and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32 accumulation (|dQ| of the whole
 * net vs float64 stays ~3e-7, the level of the f32 kernels).  Scaled activations beyond +-65504 are clamped AND
 * reported: the kernel sets the layer's range flag, the int32 at byte offset SNK_CONV_F16S_FLAG_OFFSET of d_wS (the
 * only word of the weight image a convolution launch writes); a caller that wants float32 accuracy must read it and
 * discard the result when it is non-zero (the net wrapper lowers x_scale and re-evaluates, or raises).
 * snk_conv3x3_prepare_weights_f16s: Keras kernel (3,3,128,128) -> the split weights in MFMA fragment order;
 *   d_wS: SNK_CONV_F16S_WEIGHT_BYTES bytes (9*128*128 hi/lo pairs, then the tail: float 2^-k, 2^k, x_scale,
 *   1/x_scale, int32 range flag (cleared here), 4 bytes of padding, the 8-byte address of an optional
 *   guard word shared by the net's layers (cleared here; snk_conv3x3_f16s_set_guard_word)).
 *   x_scale: a power of two the layer's INPUT activations are multiplied by before the split (undone exactly in
 *   the epilogue); choose it so that the largest |activation| * x_scale stays well below 65504 (1.0f is always
 *   valid; the net wrapper derives it from the producing layer's batch-norm parameters).
 * 3 <= width <= 80. */
#define SNK_CONV_F16S_WEIGHT_BYTES (9 * 128 * 128 * 4 + 32)
#define SNK_CONV_F16S_TAIL_OFFSET (9 * 128 * 128 * 4)
#define SNK_CONV_F16S_FLAG_OFFSET (9 * 128 * 128 * 4 + 16)
int snk_conv3x3_prepare_weights_f16s(const float *d_w_hwio, void *d_wS, float x_scale, void *stream);
/* One word for all layers of a net: snk_conv3x3_f16s_set_guard_word stores the address of a caller-owned DEVICE int32 in bytes
 * 24..31 of a weight image's tail (snk_conv3x3_prepare_weights_f16s clears them: register again after every prepare; NULL
 * unregisters).  A launch that clamps then also stores 1 to that word.  Two uses: (i) the kernels of a rollout tick that follow
 * the leaf evaluation take the word as their gate (d_skip of snk_tt_set_priors, snk_mcts_select, snk_mcts_backup,
 * snk_engine_step_active, snk_mcts_retire) and do nothing when it is non-zero, so a tick whose evaluation clamped leaves no
 * trace; (ii) the host copies the one word back (asynchronously, read at its next synchronisation) and looks at the per-layer
 * flags only when it is non-zero: widen that layer's x_scale, evaluate the batch again. */
int snk_conv3x3_f16s_set_guard_word(void *d_wS, int32_t *d_word, void *stream);
int snk_conv3x3_bn_f16s(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                        const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                        void *stream);
/* Reduced-precision form of the same kernel for BASELINE configs[4] ("bf16 MFMA conv"; NOT the 1e-5 parity path): the hi
 * parts only, one v_mfma_f32_32x32x16_f16 per product (f16 operands: 11 significand bits, float32 accumulate).  Takes
 * the weight image of snk_conv3x3_prepare_weights_f16s. */
int snk_conv3x3_bn_f16(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                       const float *d_residual, float *d_out, int n_images, int height, int width, int relu,
                       void *stream);
/* The sub-rectangle form of snk_conv3x3_bn_f16s (relu = 1).  The reference's observation (game.py:215-257) is ONE pixel value
 * (0, WALL, 0) everywhere outside the board window, and a 3x3 'same' convolution moves information by one pixel per layer, so
 * the output of the stem outside the window grown by 1 pixel, and of the tower's k-th layer (alpha_nnet.py:25-47) outside the
 * window grown by k + 2, does not depend on the state: it is what the layer gives on an all-background observation (the zero
 * padding at the canvas edge included) -- the layer's BACKGROUND IMAGE, [height][width][128].  Only the grown window is computed
 * and written per observation; a reader takes a pixel outside the producer's window from the producer's background image.
 * Results are bit-identical to the full convolution.
 * snk_conv_rect_plan: finds every observation's bounding box of non-background pixels (d_planes: [n][height][width][3], the
 *   stem's input; d_bbox: n uint32 = y0 | x0 << 8 | y1 << 16 | x1 << 24) and writes, for each of n_layers layers, the block
 *   descriptors of the rectangles box grown by grow[l] (cut to the canvas): d_desc = n_layers *
 *   snk_conv_rect_max_blocks(n, height, width) descriptors of 16 bytes, d_counts[2 l] = descriptors of layer l,
 *   d_counts[2 l + 1] = 32-row GEMM tiles they cover.
 * snk_conv3x3_bn_f16s_rect: one layer's launch: d_desc / d_count point at that layer's descriptors and count.  d_x is valid on
 *   the box grown by grow_in (what its producer computed) and read from d_bg_in outside it (d_bg_in NULL: d_x is valid
 *   wherever the taps reach); d_residual likewise with grow_res / d_bg_res.  d_out is written on the layer's rectangle only --
 *   and, with d_bg_out (the layer's own background image) given, everywhere else on the canvas too (the readers are full layers).
 * snk_stem_conv_bn_relu_f32_rect: snk_stem_conv_bn_relu_f32 (alpha_nnet.py:21-22) on the boxes grown by `grow` pixels. */
long snk_conv_rect_max_blocks(int n_images, int height, int width);
int snk_conv_rect_plan(const float *d_planes, float b0, float b1, float b2, int n_images, int height, int width,
                       int n_layers, const int *grow, void *d_bbox, void *d_desc, int *d_counts, void *stream);
int snk_stem_conv_bn_relu_f32_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                   float *d_out, const void *d_bbox, int grow, int n_images, int height, int width,
                                   void *stream);
/* The plan of the towers with 16-bit activations (snk_conv3x3_bn_f16_act16_rect / _bf16_act16_rect): their block frame holds
 * more pixels per LDS buffer (no lo parts), so a rectangle is cut into fewer, larger blocks; same arguments and descriptor
 * format, the descriptor array of a layer has snk_conv_rect_max_blocks_act16(n, height, width) entries. */
long snk_conv_rect_max_blocks_act16(int n_images, int height, int width);
int snk_conv_rect_plan_act16(const float *d_planes, float b0, float b1, float b2, int n_images, int height, int width,
                             int n_layers, const int *grow, void *d_bbox, void *d_desc, int *d_counts, void *stream);
/* the same two for the reduced-precision tower with f16 activations (snk_stem_conv_bn_relu_f16out, snk_conv3x3_bn_f16_act16 with
 * relu = 1 and f16 output; f16 background images) */
int snk_stem_conv_bn_relu_f16out_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                      void *d_out16, const void *d_bbox, int grow, int n_images, int height, int width,
                                      void *stream);
int snk_conv3x3_bn_f16_act16_rect(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                  const void *d_residual16, void *d_out16, const void *d_desc, const int *d_count,
                                  const void *d_bg_in16, int grow_in, const void *d_bg_res16, int grow_res,
                                  const void *d_bg_out16, int n_images, int height, int width, void *stream);
int snk_conv3x3_bn_f16s_rect(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                             const float *d_residual, float *d_out, const void *d_desc, const int *d_count,
                             const float *d_bg_in, int grow_in, const float *d_bg_res, int grow_res,
                             const float *d_bg_out, int n_images, int height, int width, void *stream);
/* The tower's LAST layer with the head's 1x1 stage fused into its epilogue (alpha_nnet.py:46-50): besides (or, with
 * d_out NULL, instead of) the layer output it writes d_h1[n][height*width] = relu(dot(out[pixel][:], w1x1) * bn_scale
 * + bn_shift); snk_head_dense_f32 finishes AlphaNNet.v from d_h1 (Flatten, Dense(128) + ReLU, Dense(3) + tanh,
 * obstacle overwrite) -- together they equal snk_conv3x3_bn_f16s(relu = 1) followed by snk_head_f32 without the
 * round trip of the last activation through HBM. */
int snk_conv3x3_bn_f16s_head(const float *d_x, const void *d_wS, const float *d_scale, const float *d_shift,
                             const float *d_residual, float *d_out, const float *d_w1x1, float bn_scale,
                             float bn_shift, float *d_h1, int n_images, int height, int width, void *stream);
/* Reduced precision with f16 ACTIVATIONS in HBM (BASELINE configs[4], "bf16-class MFMA conv"; outside the 1e-5
 * tolerance, never the default): d_x16 / d_residual16 are f16 [n][H][W][128] arrays, the output is f16 (out_f16 != 0:
 * every tower layer but the last) or float32 (the layer the head reads); weights = snk_conv3x3_prepare_weights_f16_act16
 * (f16 values in the 16-bit towers' image: [chunk of 32 input channels][tap][32-output tile][k step][lane] x 8, pre-scaled by
 * a power of two that the epilogue undoes).  snk_stem_conv_bn_relu_f16out is the stem that feeds it.  The block frame of
 * these towers works in chunks of 32 input channels (two MFMA k steps per LDS pixel) and its epilogue moves 16 bytes per
 * thread and row in and out.                                                                                      */
int snk_conv3x3_prepare_weights_f16_act16(const float *d_w_hwio, void *d_wS, void *stream);
int snk_conv3x3_bn_f16_act16(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                             const void *d_residual16, void *d_out, int out_f16, int n_images, int height, int width,
                             int relu, void *stream);
int snk_stem_conv_bn_relu_f16out(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                 void *d_out16, int n_images, int height, int width, void *stream);
/* BASELINE.json configs[4] AS IT IS WORDED -- "bf16 MFMA conv" (outside the 1e-5 tolerance, never the default): the same block
 * body instantiated for bf16.  Activations are bf16 [n][H][W][128] arrays in HBM (d_x16 / d_residual16; the output is bf16 when
 * out_bf16 != 0, float32 for the layer the head reads), weights are bf16 in the 16-bit towers' image
 * (snk_conv3x3_prepare_weights_bf16: a SNK_CONV_F16S_WEIGHT_BYTES buffer, first half used), products run on
 * v_mfma_f32_32x32x16_bf16 with float32 accumulation; batch norm, shortcut and ReLU in float32.  bf16 has float32's exponent
 * range: no activation scale, no clamp, no range flag.  snk_stem_conv_bn_relu_bf16out is the stem that feeds it; the
 * sub-rectangle forms (_rect) take bf16 background images.  (Rounds 1-3 had an im2col kernel with float32 activations under
 * the name snk_conv3x3_bn_bf16: 0.17 of the MFMA peak; retired in round 4.) */
int snk_conv3x3_prepare_weights_bf16(const float *d_w_hwio, void *d_wS, void *stream);
int snk_conv3x3_bn_bf16_act16(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                              const void *d_residual16, void *d_out, int out_bf16, int n_images, int height, int width,
                              int relu, void *stream);
int snk_conv3x3_bn_bf16_act16_rect(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                   const void *d_residual16, void *d_out16, const void *d_desc, const int *d_count,
                                   const void *d_bg_in16, int grow_in, const void *d_bg_res16, int grow_res,
                                   const void *d_bg_out16, int n_images, int height, int width, void *stream);
/* The last tower layer of the two towers with 16-bit activations with the head's 1x1 stage fused into its epilogue
 * (alpha_nnet.py:46-50; what snk_conv3x3_bn_f16s_head is to the float32 tower): d_h1[n][height * width], no layer output;
 * snk_head_dense_f32 finishes AlphaNNet.v from d_h1. */
int snk_conv3x3_bn_f16_act16_head(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                  const void *d_residual16, const float *d_w1x1, float bn_scale, float bn_shift,
                                  float *d_h1, int n_images, int height, int width, void *stream);
int snk_conv3x3_bn_bf16_act16_head(const void *d_x16, const void *d_wS, const float *d_scale, const float *d_shift,
                                   const void *d_residual16, const float *d_w1x1, float bn_scale, float bn_shift,
                                   float *d_h1, int n_images, int height, int width, void *stream);
int snk_stem_conv_bn_relu_bf16out(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                  void *d_out16, int n_images, int height, int width, void *stream);
int snk_stem_conv_bn_relu_bf16out_rect(const float *d_x, const float *d_w, const float *d_scale, const float *d_shift,
                                       void *d_out16, const void *d_bbox, int grow, int n_images, int height, int width,
                                       void *stream);
int snk_head_dense_f32(const float *d_h1, const float *d_fc1_w, const float *d_fc1_b, const float *d_fc2_w,
                       const float *d_fc2_b, const uint8_t *d_mask, float *d_q, int n_images, int height, int width,
                       void *stream);
int snk_stem_conv_bn_relu_f32(const float *d_x, const float *d_w, const float *d_scale,
                              const float *d_shift, float *d_out, int n_images, int height, int width,
                              void *stream);
int snk_head_f32(const float *d_x, const float *d_w1x1, float bn_scale, float bn_shift,
                 const float *d_fc1_w, const float *d_fc1_b, const float *d_fc2_w, const float *d_fc2_b,
                 const uint8_t *d_mask, float *d_q, int n_images, int height, int width, void *stream);

/* ---- the Agent-wide transposition cache (agent.py:16-19, 151-157) ------------------------
 * cached_values / total_rewards / visit_cnts / cache_hit as one open-addressing table in HBM keyed
 * by the 128-bit observation digest (snk_engine_observe d_key).  capacity: power of two.
 * An "entry" is a uint32 table slot; 0xFFFFFFFF = none (dead snake / inactive row).            */
typedef struct snk_tt snk_tt;
int snk_tt_create(snk_tt **out, uint64_t capacity, int device);
int snk_tt_destroy(snk_tt *t);
int snk_tt_clear(snk_tt *t, void *stream);                              /* Agent.clear, agent.py:140-147 */
int snk_tt_status_sync(snk_tt *t, int64_t *capacity, int64_t *occupied, int *overflowed);
/* eviction (agent.py:101-110): keeps entries with now_turn - last_touch <= max_age, re-hashed into a
 * table of new_capacity.  Entry indices change: call between root turns only.                    */
int snk_tt_rebuild_sync(snk_tt *t, uint64_t new_capacity, int now_turn, int max_age);
/* MCTSAgent.make_moves "get states without duplicates" (agent.py:170-186): find-or-insert m keys.
 * d_active (optional) uint8[m]: 0 rows are skipped (entry = none).  d_is_new[i] = 1 for exactly one
 * row of every key that has no live entry (never seen, or evicted: now_turn - touch > max_age + 1):
 * that row's observation must be evaluated by the net.  Every found entry is touched
 * (cache_hit[key] = 0, agent.py:185).                                                            */
int snk_tt_lookup_insert(snk_tt *t, const uint64_t *d_key, const uint8_t *d_active, int m, int now_turn,
                         int max_age, uint32_t *d_entry, uint8_t *d_is_new, void *stream);
/* Read-only probe -- `key in cached_values`, cached_values[key] / total_rewards[key] / visit_cnts[key] /
 * cache_hit[key] of the reference's four dicts (agent.py:16-19): d_entry[i] = entry index of key i, or 0xFFFFFFFF
 * when it does not exist (never inserted, or evicted: now_turn - touch > max_age + 1).  d_stat7 (optional)
 * float[m][7] = total[3], visit[3], age in root turns.  Nothing is inserted or touched.                          */
int snk_tt_find(snk_tt *t, const uint64_t *d_key, int m, int now_turn, int max_age, uint32_t *d_entry,
                float *d_stat7, void *stream);
/* new entries (agent.py:193-201): total = d_q[j], visit = 1,1,1 for entry d_entry[d_idx ? d_idx[j] : j] */
/* d_skip (here and in snk_mcts_select / _backup / _retire; optional): the rollout tick's gate, see snk_engine_step_active */
int snk_tt_set_priors(snk_tt *t, const uint32_t *d_entry, const int32_t *d_idx, int n, const float *d_q,
                      const int32_t *d_skip, void *stream);
/* d_q[i] = total/visit of entry d_entry[i * entry_stride]  (cached_values[first_key], agent.py:83-87) */
int snk_tt_read_q(snk_tt *t, const uint32_t *d_entry, int entry_stride, int m, float *d_q, void *stream);

/* ---- rollout tick of MCTSAgent.make_moves (agent.py:203-222) -----------------------------
 * Paths: d_path_entry uint32[m][path_depth], d_path_move uint8[m][path_depth], d_path_len int32[m]
 * are MCTSAgent.keys / .moves (agent.py:158-159) for row i = (subgame, snake).
 * Randomness: d_tape_u == NULL -> counter-based Philox keyed by (seed; row, ctr0, ctr1); else row i
 * uses the recorded uniform d_tape_u[tape_base + (d_rank ? d_rank[i] : i)] (parity runs: the draw
 * numpy.random.choice would have consumed).
 * snk_mcts_select: pmf = softermax(Q), move ~ pmf, est = pmf . Q, appends (entry, move) at
 *   path[len] (len is advanced by snk_mcts_backup).  d_est / d_pmf optional outputs.
 * snk_mcts_backup: every ancestor edge: visit += 1, total += est (agent.py:208-220).
 *   sequential = 0: one thread per row, float atomics, est as computed by select;
 *   sequential = 1: the reference's order (rows ascending, live Q re-read, needs d_pmf), one thread.
 * snk_mcts_terminal_backup: rows with reward +1/-1 add it along their whole path (agent.py:60-72);
 *   d_rewards int8[m], 0 = None.                                                                 */
int snk_mcts_select(snk_tt *t, const uint32_t *d_entry, int m, float softmax_base, const double *d_tape_u,
                    const int32_t *d_rank, int64_t tape_base, uint64_t seed, uint32_t ctr0, uint32_t ctr1,
                    uint8_t *d_moves, float *d_est, float *d_pmf, uint32_t *d_path_entry,
                    uint8_t *d_path_move, int32_t *d_path_len, int path_depth, const int32_t *d_skip, void *stream);
/* The rollout loop's bookkeeping around a tick (mp_game_runner.py:99-113), one launch each:
 * snk_mcts_row_active: d_row_active[b * n_snakes + s] = snake s of sub-game b is alive AND the sub-game is still active
 *   (d_alive_rows: what snk_engine_alive wrote; d_sub_active: uint8[n_subgames]).
 * snk_mcts_retire (after snk_engine_step_active): *d_sim_steps += number of active sub-games (the ones that just moved); a
 *   sub-game becomes inactive when its game is over (d_done, from the step) or tick >= d_sub_depth[b] (its depth cap).
 * snk_mcts_gather_rows: out_pairs[i] = pairs[idx[i]] ((sub-game, snake), int32 x 2), out_mask[i] = mask[idx[i]] (uint8 x 3): the
 *   rows whose observations go to the net (all_states, agent.py:172-186) out of the tick's row list. */
int snk_mcts_gather_rows(const int32_t *d_idx, int n, const int32_t *d_pairs, const uint8_t *d_mask, int32_t *d_out_pairs,
                         uint8_t *d_out_mask, void *stream);
int snk_mcts_row_active(const uint8_t *d_alive_rows, const uint8_t *d_sub_active, int n_subgames, int n_snakes,
                        uint8_t *d_row_active, void *stream);
int snk_mcts_retire(uint8_t *d_sub_active, const uint8_t *d_done, const int32_t *d_sub_depth, int tick, int n_subgames,
                    int64_t *d_sim_steps, const int32_t *d_skip, void *stream);
int snk_mcts_backup(snk_tt *t, const uint32_t *d_entry, int m, const float *d_est, const float *d_pmf,
                    uint32_t *d_path_entry, uint8_t *d_path_move, int32_t *d_path_len, int path_depth,
                    int sequential, const int32_t *d_skip, void *stream);
int snk_mcts_terminal_backup(snk_tt *t, const int8_t *d_rewards, int m, uint32_t *d_path_entry,
                             uint8_t *d_path_move, int32_t *d_path_len, int path_depth, int sequential,
                             void *stream);
/* root decision (agent.py:89-99): training -> move ~ softermax(V), else Agent.argmaxs(V) */
int snk_mcts_root_moves(const float *d_V, const uint8_t *d_alive, int m, float softmax_base, int training,
                        const double *d_tape_u, const int32_t *d_rank, int64_t tape_base, uint64_t seed,
                        uint32_t ctr0, uint32_t ctr1, uint8_t *d_moves, void *stream);
/* Agent.softermax / Agent.argmaxs on m rows of 3 (agent.py:114-137) */
int snk_softermax_argmax(const float *d_z, int m, float softmax_base, float *d_pmf, uint8_t *d_argmax,
                         void *stream);
/* Game.rewards of n games as int8[n][S]: 0 None, +1, -1 (mp_game_runner.py:110) */
int snk_engine_rewards(const snk_engine *e, const int32_t *d_slots, int n, int8_t *d_rewards, void *stream);

/* ---- training half: AlphaNNet.train = model.fit (alpha_nnet.py:58-59) -------------------------
 * Training-mode batch normalisation of a 128-channel channels-last float32 activation [rows = n * h * w][128] (the
 * BatchNormalization layers of alpha_nnet.py:23-46 as Keras runs them under fit), fused with the ReLU / residual add around
 * it.  Between the sums and the apply kernel of a direction sit snk_bn_train_finalize / snk_bn_train_grad_finalize (below),
 * and the caller's all-reduce of the float64 sums when it runs data-parallel.  d_partials: snk_bn_train_partials() floats.
 *   snk_bn_train_apply          out = y * scale + shift (+ residual), then ReLU when relu != 0; d_relu_mask (optional): one byte
 *                               per row and quad of channels, byte [row * 32 + c / 4] bit (c % 4) = out[row][c] > 0
 *   snk_bn_train_grad_sums_f64  with g = dout masked by the ReLU: d_sums = { sum g, sum g * xhat }, xhat = (y - mean) * inv.  The
 *                               mask is d_relu_mask (what snk_bn_train_apply wrote: 1/16 of the bytes) or, when that is NULL,
 *                               the sign of d_out (the activation itself, or any tensor whose sign is the mask)
 *   snk_bn_train_grad_apply     dx = a * (g - b - xhat * c); d_g (optional) = g, the gradient of the residual branch
 * The two apply kernels can hand on the power-of-two input scale of the convolution that reads their result (what
 * snk_conv3x3_f16s_input_scale would derive from it): d_out_scale_tail / d_dx_scale_tail (optional) = 4 floats
 * { ., ., scale, 1 / scale }, taken from the values as they are written -- the tensor is not read again for its maximum. */
int snk_bn_train_partials(void);
/* The split-f16 convolution's power-of-two input scale from the data itself, for tensors whose range is not known ahead
 * (gradients): x_scale with 2^11 <= max|x| * x_scale < 2^12 is written into the tail of the weight image d_wS
 * (after snk_conv3x3_prepare_weights_f16s), on the device.  n_floats: a multiple of 4; d_partials as above. */
int snk_conv3x3_f16s_input_scale(const float *d_x, long n_floats, void *d_wS, float *d_partials, void *stream);
/* Weight gradient of the tower convolution (the third convolution pass of a training step), float32 accuracy on the f16
 * matrix pipe: d_dw[3][3][128][128] (Keras layout) = sum over images and pixels of x[n][y + kh - 1][x + kw - 1][ci] *
 * dy[n][y][x][co]; x, dy channels-last float32 [n][h][w][128]; d_x_tail / d_dy_tail: the 4 floats { ., ., scale, 1 / scale }
 * at SNK_CONV_F16S_TAIL_OFFSET of a weight image on which snk_conv3x3_f16s_input_scale ran for that tensor.
 * d_partials: snk_conv3x3_wgrad_partials(h, w) floats (-1: the shape does not fit, use another path). */
long snk_conv3x3_wgrad_partials(int height, int width);
int snk_conv3x3_wgrad_f16s(const float *d_x, const float *d_dy, const float *d_x_tail, const float *d_dy_tail,
                           float *d_partials, float *d_dw, int n_images, int height, int width, void *stream);
int snk_bn_train_apply(const float *d_y, const float *d_scale, const float *d_shift, const float *d_residual,
                       float *d_out, long rows, int relu, float *d_partials, float *d_out_scale_tail, uint8_t *d_relu_mask,
                       void *stream);
/* snk_bn_train_apply (relu = 1) for the tower's LAST layer with the head's 1x1 convolution (alpha_nnet.py:49) and its batch-norm
 * sums taken from the values on their way out: d_z[rows] = dot(out[row][:], d_w1x1), d_hsums = { sum (z - center1), sum (z -
 * center1)^2 } (float64; d_center1: one float or NULL) -- snk_head_conv1x1_sums without its pass over d_out.  No scale tail: no
 * convolution reads this output.  d_partials: snk_bn_train_partials() floats. */
int snk_bn_train_apply_head(const float *d_y, const float *d_scale, const float *d_shift, const float *d_residual, float *d_out,
                            long rows, float *d_partials, uint8_t *d_relu_mask, const float *d_w1x1, const float *d_center1,
                            float *d_z, double *d_hsums, void *stream);
int snk_bn_train_grad_sums_f64(const float *d_dout, const float *d_out, const uint8_t *d_relu_mask, const float *d_y,
                               const float *d_mean, const float *d_inv, long rows, int relu, float *d_partials, double *d_sums,
                               void *stream);
int snk_bn_train_grad_apply(const float *d_dout, const float *d_out, const uint8_t *d_relu_mask, const float *d_y,
                            const float *d_mean, const float *d_inv, const float *d_a, const float *d_b, const float *d_c,
                            float *d_dx, float *d_g, long rows, int relu, float *d_partials, float *d_dx_scale_tail,
                            void *stream);

/* ---- the rest of the training step (csrc/train_net.hip; driven by snake_engine/train_step.py) ------------------------------
 * Everything AlphaNNet.train (alpha_nnet.py:58-59: Keras fit = forward, backward, Adam) needs besides the calls above, so that a
 * step runs on this library alone.  Sums travel as float64 (the ranks all-reduce them between a *_sums and its *_finalize).
 *   snk_conv3x3_prepare_weights_f16s_train  weight image of a tower layer for the forward pass (input_gradient = 0) or for its
 *        input gradient (1: taps mirrored, channel axes swapped); the input's power-of-two scale is read on the device from
 *        d_in_tail = { ., ., scale, 1 / scale } as snk_bn_train_apply / snk_bn_train_grad_apply left it; d_wS_same_kernel
 *        (optional): an image already made from the same kernel values -- its weight scale is reused
 *   snk_stem_conv_f32        the bare 3 -> 128 convolution of alpha_nnet.py:21 (no batch norm, no ReLU); d_w = Keras kernel (3,3,3,128)
 *   snk_stem_wgrad_f32       its weight gradient d_dw[3][3][3][128]; d_partials: snk_stem_wgrad_partials(n, h, w) floats
 *   snk_bn_train_sums_f64    d_sums[0..127] = sum (y - center), [128..255] = sum (y - center)^2 (d_center: 128 floats or NULL)
 *   snk_bn_train_finalize    sums of ALL ranks (count values per channel) -> mean, 1 / sqrt(var + eps), scale = gamma * inv, shift =
 *        beta - mean * scale; moving_mean / moving_var (optional) move with `momentum`, the variance taking the unbiased batch
 *        variance (TF fused batch norm).  d_center may be d_moving_mean.  Works for any channel count (the head's is 1).
 *   snk_bn_train_grad_sums_f64 / snk_bn_train_grad_finalize   { sum g, sum g xhat } -> a, b, c of snk_bn_train_grad_apply (from the
 *        sums of all ranks) and this rank's dgamma, dbeta (from its own sums)
 *   snk_head_conv1x1_sums    z[row] = dot(a[row][0..127], w1x1) (alpha_nnet.py:49) + d_sums[2] = sum (z - center), sum (z - center)^2
 *   snk_head_dense_train_fwd h = relu(z * scale + shift) [n][hw], d1 = relu(h W1 + b1) [n][128], q = tanh(d1 W2 + b2) [n][3]
 *        (alpha_nnet.py:50-54); d_scale_shift: 2 floats in device memory; d_sq_err (optional) = err_scale * sum (q - target)^2;
 *        d_h / d_d1 optional; d_partials: snk_bn_train_partials() floats
 *   snk_head_dense_train_bwd the way back: d_g[n][hw] (gradient at the 1-channel batch norm's output, ReLU applied), d_dw1[hw][128],
 *        d_small[515] = dW2[128][3], db2[3], db1[128], d_gsums[2] = sum g, sum g zhat; norm = 1 / (3 * rows of the global
 *        batch); d_h_mask / d_d1_mask (optional): tensors whose SIGN replaces h > 0 / d1 > 0 as the ReLU masks;
 *        d_partials: snk_head_dense_train_bwd_partials(n) floats, d_dpre1: n * 128 floats of scratch
 *   snk_head_conv1x1_bwd     dz = a (g - b - zhat c) with d_abc = {a, b, c}; d_da[row][c] = dz[row] * w1x1[c]; d_dw1x1[c] =
 *        sum_row a_last[row][c] * dz[row]; d_partials: snk_bn_train_partials() floats
 *   snk_adam_l2_step         tf.keras Adam on flat buffers: g += 2 * l2 * w where d_decay[i] != 0 (Conv2D / Dense kernels,
 *        kernel_regularizer = l2(1e-5)); m, v updated; w -= lr_t * m / (sqrt(v) + epsilon)
 *   snk_l2_sum               d_out[0] = scale * sum of w[i]^2 over d_decay[i] != 0; d_partials: 1 024 floats                      */
int snk_conv3x3_prepare_weights_f16s_train(const float *d_w_hwio, void *d_wS, const float *d_in_tail, int input_gradient,
                                           const void *d_wS_same_kernel, void *stream);
/* The weight images of ALL tower layers of a training step in two launches: h_w_hwio / h_wS_fwd / h_wS_bwd are HOST arrays of
 * n_layers (<= 40) device pointers -- the Keras kernels, their forward images and (array or entries may be NULL) their
 * input-gradient images.  Writes every image's fragments, weight scale and a cleared range flag, NOT its input scale: floats 2 and
 * 3 of the tail at SNK_CONV_F16S_TAIL_OFFSET are left to the kernel that writes the convolution's input (hand that address to
 * snk_bn_train_apply / snk_bn_train_grad_apply / snk_bn_train_finalize_range as their scale tail).  Images made this way stay
 * valid while the weights do: the forward-only steps at learning rate 0 (alpha_nnet.py:79-84) reuse them. */
int snk_conv3x3_prepare_weights_f16s_train_batch(const float *const *h_w_hwio, void *const *h_wS_fwd, void *const *h_wS_bwd,
                                                 int n_layers, void *stream);
int snk_stem_conv_f32(const float *d_x, const float *d_w, float *d_out, int n_images, int height, int width, void *stream);
/* snk_stem_conv_f32 with the sums its batch norm starts from taken in the kernel's epilogue (d_sums as snk_bn_train_sums_f64(d_out,
 * d_center) leaves them, without that pass over the output); d_partials: snk_bn_train_partials() floats */
int snk_stem_conv_f32_stats(const float *d_x, const float *d_w, float *d_out, const float *d_center, float *d_partials,
                            double *d_sums, int n_images, int height, int width, void *stream);
/* The forward convolution of a tower layer under training: d_out = conv3x3_same(d_x, w), bare (the batch norm comes after), and
 * from the same values on their way out of the kernel d_sums[0..127] = sum (out - center), [128..255] = sum (out - center)^2
 * over all n * h * w pixels (float64; d_center: 128 floats or NULL): snk_bn_train_sums_f64(d_out) without the extra pass.
 * d_wS: snk_conv3x3_prepare_weights_f16s_train(..., input_gradient = 0); d_partials: snk_conv3x3_stats_partials(n, h, w) floats. */
long snk_conv3x3_stats_partials(int n_images, int height, int width);
int snk_conv3x3_f16s_stats(const float *d_x, const void *d_wS, float *d_out, const float *d_center, float *d_partials,
                           double *d_sums, int n_images, int height, int width, void *stream);
/* The INPUT-GRADIENT convolution of the training step with the next batch-norm backward's two sums in its epilogue (what
 * snk_bn_train_grad_sums_f64 would compute from d_out in a pass of its own): d_out = conv3x3_same(d_x, mirrored kernel image)
 * (+ d_residual: the shortcut's gradient) = the gradient at the output of the layer below; d_sums[0..127] = sum(g),
 * d_sums[128..255] = sum(g * (y - mean) * inv), g = d_out where that layer's ReLU bit (d_mask, snk_bn_train_apply's bytes) is
 * set, d_y its pre-batch-norm output.  d_partials: snk_conv3x3_stats_partials floats.  (alpha_nnet.py:58-59: Keras fit) */
int snk_conv3x3_f16s_igrad_stats(const float *d_x, const void *d_wS, const float *d_residual, float *d_out, const float *d_y,
                                 const uint8_t *d_mask, const float *d_mean, const float *d_inv, float *d_partials,
                                 double *d_sums, int n_images, int height, int width, void *stream);
/* ---- the deferred batch norm of the training step (alpha_nnet.py:25-47 under Keras fit, alpha_nnet.py:58-59) ----------------
 * The activation BETWEEN the two convolutions of a residual block, relu(bn(conv1(x))), has exactly three readers: the block's
 * second convolution, that layer's weight gradient, and (as a sign) the batch-norm backward of the first.  These entry points let
 * all three take it from the first convolution's PRE-batch-norm output y as relu(y * scale[c] + shift[c]) -- the expression
 * snk_bn_train_apply(relu = 1, no residual) evaluates, bit for bit -- so that the activation and its mask bytes are never written
 * nor read: one element-wise pass over two 462 MB tensors less per block, in every step (also the forward-only steps at rate 0).
 *   snk_train_deferred_bn_supported(h, w)   1 when every kernel involved exists for the shape (the weight gradient's window form)
 *   snk_conv3x3_f16s_stats_deferred         snk_conv3x3_f16s_stats with d_in_scale / d_in_shift (both or neither: d_x is such a y)
 *                                           and d_amax (or NULL): 128 floats, largest |out - center| per channel
 *   snk_bn_train_finalize_range             snk_bn_train_finalize + d_out_scale_tail = { ., ., 2^k, 2^-k } with 2^11 <= bound 2^k <
 *                                           2^12, bound = max_c relu(scale_c (center_c +- amax_c) + shift_c) >= every value the
 *                                           deferred activation takes (what snk_bn_train_apply measures while it writes)
 *   snk_conv3x3_wgrad_f16s_deferred         snk_conv3x3_wgrad_f16s whose X operand is such a y (d_x_tail from finalize_range)
 *   snk_conv3x3_f16s_igrad_stats_deferred   snk_conv3x3_f16s_igrad_stats whose ReLU decision is d_y * d_scale + d_shift > 0
 *   snk_bn_train_grad_sums_f64_deferred / snk_bn_train_grad_apply_deferred   the same decision in the two element-wise kernels */
/*   The stem's own batch norm + ReLU output can be deferred the same way (readers: the first tower convolution, its weight gradient, the
 *   first block's shortcut, the stem's batch-norm backward):
 *   snk_stem_conv_f32_stats_deferred                snk_stem_conv_f32_stats + d_amax (128 floats: largest |out - center| per channel)
 *   snk_conv3x3_f16s_stats_deferred                 with BOTH input scale / shift and d_amax: the layer right above
 *   snk_bn_train_apply_res_deferred                 snk_bn_train_apply (relu = 1) whose shortcut d_res_y is PRE-batch-norm: relu(y s + t)
 *   snk_conv3x3_f16s_igrad_stats_masked_res_deferred   ..._masked_res whose sums take the ReLU decision d_y * d_scale + d_shift > 0 */
int snk_train_deferred_bn_supported(int height, int width);
int snk_stem_conv_f32_stats_deferred(const float *d_x, const float *d_w, float *d_out, const float *d_center, float *d_amax,
                                     float *d_partials, double *d_sums, int n_images, int height, int width, void *stream);
int snk_bn_train_apply_res_deferred(const float *d_y, const float *d_scale, const float *d_shift, const float *d_res_y,
                                    const float *d_res_scale, const float *d_res_shift, float *d_out, long rows, float *d_partials,
                                    float *d_out_scale_tail, uint8_t *d_relu_mask, void *stream);
int snk_conv3x3_f16s_igrad_stats_masked_res_deferred(const float *d_x, const void *d_wS, const float *d_residual,
                                                     const uint8_t *d_residual_mask, float *d_out, const float *d_y,
                                                     const float *d_scale, const float *d_shift, const float *d_mean,
                                                     const float *d_inv, float *d_partials, double *d_sums, int n_images, int height,
                                                     int width, void *stream);
int snk_conv3x3_f16s_stats_deferred(const float *d_x, const void *d_wS, float *d_out, const float *d_center, const float *d_in_scale,
                                    const float *d_in_shift, float *d_amax, float *d_partials, double *d_sums, int n_images,
                                    int height, int width, void *stream);
int snk_bn_train_finalize_range(const double *d_sums, double count, const float *d_center, const float *d_gamma, const float *d_beta,
                                float *d_moving_mean, float *d_moving_var, double momentum, double eps, float *d_mean, float *d_inv,
                                float *d_scale, float *d_shift, const float *d_amax, float *d_out_scale_tail, int channels,
                                void *stream);
int snk_conv3x3_wgrad_f16s_deferred(const float *d_y_below, const float *d_scale, const float *d_shift, const float *d_dy,
                                    const float *d_x_tail, const float *d_dy_tail, float *d_partials, float *d_dw, int n_images,
                                    int height, int width, void *stream);
int snk_conv3x3_f16s_igrad_stats_deferred(const float *d_x, const void *d_wS, const float *d_residual, float *d_out, const float *d_y,
                                          const float *d_scale, const float *d_shift, const float *d_mean, const float *d_inv,
                                          float *d_partials, double *d_sums, int n_images, int height, int width, void *stream);
int snk_bn_train_grad_sums_f64_deferred(const float *d_dout, const float *d_y, const float *d_scale, const float *d_shift,
                                        const float *d_mean, const float *d_inv, long rows, float *d_partials, double *d_sums,
                                        void *stream);
int snk_bn_train_grad_apply_deferred(const float *d_dout, const float *d_y, const float *d_scale, const float *d_shift,
                                     const float *d_mean, const float *d_inv, const float *d_a, const float *d_b, const float *d_c,
                                     float *d_dx, float *d_g, long rows, float *d_partials, float *d_dx_scale_tail, void *stream);
/* snk_conv3x3_f16s_igrad_stats whose shortcut gradient is d_residual WHERE d_residual_mask's bit is set: d_residual = the
 * gradient at the residual block's output as the layer above delivered it (before that output's ReLU), d_residual_mask = that
 * output's ReLU bits (snk_bn_train_apply's bytes).  The masked copy snk_bn_train_grad_apply(d_g) would write for this epilogue
 * (462 MB per block at 2 048 x 21 x 21) is then not needed.  d_out may be d_residual (in place). */
int snk_conv3x3_f16s_igrad_stats_masked_res(const float *d_x, const void *d_wS, const float *d_residual,
                                            const uint8_t *d_residual_mask, float *d_out, const float *d_y, const uint8_t *d_mask,
                                            const float *d_mean, const float *d_inv, float *d_partials, double *d_sums,
                                            int n_images, int height, int width, void *stream);
long snk_stem_wgrad_partials(int n_images, int height, int width);
int snk_stem_wgrad_f32(const float *d_x, const float *d_dy, float *d_partials, float *d_dw, int n_images, int height, int width,
                       void *stream);
int snk_bn_train_sums_f64(const float *d_y, long rows, const float *d_center, float *d_partials, double *d_sums, void *stream);
int snk_bn_train_finalize(const double *d_sums, double count, const float *d_center, const float *d_gamma, const float *d_beta,
                          float *d_moving_mean, float *d_moving_var, double momentum, double eps, float *d_mean, float *d_inv,
                          float *d_scale, float *d_shift, int channels, void *stream);
int snk_bn_train_grad_finalize(const double *d_sums_global, const double *d_sums_local, double count, const float *d_gamma,
                               const float *d_inv, float *d_a, float *d_b, float *d_c, float *d_dgamma, float *d_dbeta,
                               int channels, void *stream);
int snk_head_conv1x1_sums(const float *d_a, const float *d_w1x1, long rows, const float *d_center, float *d_z, float *d_partials,
                          double *d_sums, void *stream);
int snk_head_dense_train_fwd(const float *d_z, const float *d_scale_shift, const float *d_fc1_w, const float *d_fc1_b,
                             const float *d_fc2_w, const float *d_fc2_b, const float *d_target, float *d_h, float *d_d1,
                             float *d_q, float *d_partials, float *d_sq_err, double err_scale, int n_images, int height,
                             int width, void *stream);
int snk_head_dense_train_bwd_partials(int n_images);
int snk_head_dense_train_bwd(const float *d_q, const float *d_target, const float *d_h, const float *d_d1, const float *d_h_mask,
                             const float *d_d1_mask, const float *d_z, const float *d_mean_inv, const float *d_fc1_w,
                             const float *d_fc2_w, double norm, float *d_dpre1, float *d_g, float *d_dw1, float *d_small,
                             double *d_gsums, float *d_partials, int n_images, int height, int width, void *stream);
int snk_head_conv1x1_bwd(const float *d_g, const float *d_z, const float *d_mean_inv, const float *d_abc, const float *d_a_last,
                         const float *d_w1x1, float *d_da, float *d_dw1x1, float *d_partials, long rows, void *stream);
/* snk_head_conv1x1_bwd that also leaves the two sums the LAST tower layer's batch-norm backward starts from (d_sums as
 * snk_bn_train_grad_sums_f64(d_da, NULL, d_mask_last, d_y_last, d_mean_last, d_inv_last, ..) leaves them, without that pass over the
 * gradient it has just written); d_stat_partials: a second buffer of snk_bn_train_partials() floats */
int snk_head_conv1x1_bwd_stats(const float *d_g, const float *d_z, const float *d_mean_inv, const float *d_abc, const float *d_a_last,
                               const float *d_w1x1, float *d_da, float *d_dw1x1, float *d_partials, const float *d_y_last,
                               const uint8_t *d_mask_last, const float *d_mean_last, const float *d_inv_last,
                               float *d_stat_partials, double *d_sums, long rows, void *stream);
int snk_adam_l2_step(float *d_w, const float *d_g, float *d_m, float *d_v, const uint8_t *d_decay, long n, double lr_t,
                     double beta1, double beta2, double epsilon, double l2, void *stream);
int snk_l2_sum(const float *d_w, const uint8_t *d_decay, long n, double scale, float *d_partials, float *d_out, void *stream);

/* Measurement aid (bench.py): one wavefront that sits on a compute unit for ~microseconds and reports d_out[0] = shader cycles
 * (s_memtime) and d_out[1] = ticks of the constant 100 MHz counter (s_memrealtime) that passed meanwhile: launched on its
 * own stream beside the measured kernels it gives the clock the chip holds under their load, cycles / ticks x 100 MHz. */
int snk_clock_probe(uint64_t *d_out, int microseconds, void *stream);
/* sha-256 (64 hex digits) of a kernel source file of alphasnake-zero_amd/csrc/ ("conv_split.hip", "engine.hip", "common.h", ...)
 * as it was when the library was built; NULL for an unknown name.  The counter measurements kept under profiles/ carry the
 * hashes of the sources they were taken on; bench.py quotes them only when the loaded library carries the same ones. */
const char *snk_source_hash(const char *source_file);

#ifdef __cplusplus
}
#endif
#endif /* SNAKE_ENGINE_H */
