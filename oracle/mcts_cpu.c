/*
 * mcts_cpu.c -- CPU restatement, in C, of the reference's randomized parallel MCTS self-play loop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): this is the `cpu_baseline` of bench.py (the same workload on
 * the GPU box's host cores) and a second checker beside oracle/mcts_oracle.py.  It is never
 * linked into, imported by or executed from the product path.
 *
 * What it restates (paths relative to /root/reference/code/utils/), over the game primitives of snake_oracle.c:
 *   Agent.make_moves           agent.py:25-111       mc_begin_turn / mc_begin_epoch / mc_end_epoch / mc_end_turn
 *   MCTSMPGameRunner.run       mp_game_runner.py:85-115   the tick loop mc_collect / mc_apply
 *   MCTSAgent.make_moves       agent.py:161-223      mc_collect (cache de-duplication), mc_apply (priors, softermax,
 *                                                    draw, in-rollout back-up in ids order on live float32 statistics)
 *   Agent.softermax            agent.py:114-122      softermax3
 *   MPGameRunner.run           mp_game_runner.py:23-77    mc_end_turn (root Game.tic, retirement, six counters)
 * Differences from the reference that do not change results: the four cache dicts are one open-addressing table
 * keyed by the 128-bit digest of the observation bytes (oracle/obs_key.py) instead of the 5 292-byte string itself;
 * ageing is a time stamp (age = now - touch) instead of a loop over all keys, eviction is lazy.
 * The net is outside: mc_collect hands out the observations that need an evaluation (one batch per rollout tick,
 * agent.py:189-190) and mc_apply takes the Q rows back, so the caller can run one PyTorch-CPU batch for all worker
 * shards; with `stub` set the deterministic stub net of oracle/obs_key.py is evaluated here instead.
 *
 * Pinned by tests/test_mcts_cpu_baseline.py against the runs recorded from the unmodified reference
 * (tests/golden/mcts_tiny*.npz): same ids, moves, per-turn evaluation counts, draws consumed, record bytes, and root
 * Q values within 1e-5 (libm's powf/atanhf are not NumPy's, so the float statistics are not bit-identical).
 *
 * One mc_worker = one Agent + one MPGameRunner over its own shard of root games; workers share nothing, so N threads
 * drive N workers (the reference's only way to use N cores is N independent processes).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_MAX_S 8
#define ORC_MAX_CELLS 361
#define ORC_MAX_NODES 384

typedef struct {
    int32_t H, W, S, health_dec;
    double food_chance;
    uint8_t alive[ORC_MAX_S];
    int16_t health[ORC_MAX_S];
    int16_t length[ORC_MAX_S];
    uint8_t dir[ORC_MAX_S];
    int16_t nodes[ORC_MAX_S][ORC_MAX_NODES];
    uint8_t food[ORC_MAX_CELLS];
    int8_t rewards[ORC_MAX_S];
    int32_t counters[6];
} orc_game;                                             /* must match snake_oracle.c (checked by orc_sizeof_game) */

int orc_sizeof_game(void);
void orc_subgame(const orc_game *src, orc_game *dst);
int orc_tic(orc_game *g, const uint8_t *moves, int spawn_mode, int spawn_cell, double u1, double u2,
            uint8_t *empty_out, int *spawned_out);
void orc_make_state(const orc_game *g, int you, float *out);
void orc_obstacle_mask(const float *state, int gh, int gw, int legacy, uint8_t *mask3);
void orc_obs_key(const float *state, int npix, uint64_t *key2);
void orc_stub_q(const float *state, int gh, int gw, float *q3);

#define PARALLEL 8                                      /* agent.py:32 */

typedef struct {
    uint64_t lo, hi;
    float total[3], visit[3], q[3];
    int32_t touch;
    uint8_t state;                                      /* 0 empty, 1 pending (cache[key] = None), 2 filled */
} tt_entry;

typedef struct {
    /* configuration */
    int H, W, S, health_dec, n_games, training, max_depth, max_breadth, stub;
    float base;
    int obs_elems, gh;
    /* root games (MPGameRunner.games) */
    orc_game *roots;
    uint8_t *root_live;
    int n_live;
    int64_t totals[6];                                  /* sums over finished games (mp_game_runner.py:54-60) */
    int64_t env_steps, sim_steps, net_evals, lookups;
    /* transposition table (the four cache dicts, agent.py:16-19) */
    tt_entry *tt;
    uint64_t tt_cap, tt_used;
    int now;
    /* rollout state of the current epoch */
    orc_game *subs;
    int *sub_depth, *sub_live;                          /* sub_live: list of live sub-game indices */
    int n_subs, cap_subs, n_sub_live, tick, epoch, overflow;
    int32_t *path_entry;                                /* [n_subs][S][max_depth] */
    uint8_t *path_move;
    int32_t *path_len;                                  /* [n_subs][S]; -1 = snake not alive at the epoch's start */
    /* rows of the current tick */
    int *row_sub, *row_snake;
    int64_t *row_entry;
    int n_rows;
    int64_t *fresh_entry;
    int n_fresh;
    float *scratch_state;
    /* draws: numpy.random.choice's uniform (agent.py:91, 205) and Game.tic's random()/choice (game.py:131-133) */
    const double *tape;
    int64_t tape_len, tape_pos;
    uint64_t rng[4];
    /* training records (agent.py:95-97) */
    float *records;
    float *values;
    int64_t n_records, cap_records;
    int keep_records;
    /* last root turn's outputs */
    int32_t *last_ids;                                  /* [n_rows_root][2] */
    float *last_V;
    uint8_t *last_moves;
    int last_n;
    const int16_t *spawn_tape;                          /* per root game for the coming root tic, or NULL */
} mc_worker;

/* ------------------------------------------------------------------------------------------- RNG (xoshiro256**) */
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double next_u(mc_worker *w)
{
    uint64_t *s = w->rng;
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}
static double draw_u(mc_worker *w)
{
    if (w->tape) {
        double u = w->tape_pos < w->tape_len ? w->tape[w->tape_pos] : 0.5;
        w->tape_pos++;
        return u;
    }
    return next_u(w);
}

/* Agent.softermax (agent.py:114-122): base ** arctanh(z) / sum, uniform when the sum is 0 */
static void softermax3(float base, const float *z, float *p)
{
    float n[3], sigma = 0.0f;
    for (int i = 0; i < 3; ++i) { n[i] = powf(base, atanhf(z[i])); }
    sigma = (n[0] + n[1]) + n[2];                       /* python sum(): left to right in float32 */
    if (sigma == 0.0f) { p[0] = p[1] = p[2] = (float)(1.0 / 3.0); return; }
    for (int i = 0; i < 3; ++i) p[i] = n[i] / sigma;
}

/* numpy.random.choice([0,1,2], p=pmf): cdf = cumsum(p) in float64, normalised, searchsorted(cdf, u, 'right') */
static int choice3(mc_worker *w, const float *p)
{
    double c0 = (double)p[0], c1 = c0 + (double)p[1], c2 = c1 + (double)p[2];
    c0 /= c2; c1 /= c2;
    const double u = draw_u(w);
    return u < c0 ? 0 : (u < c1 ? 1 : 2);
}

/* ------------------------------------------------------------------------------------------- transposition table */
static inline int entry_valid(const mc_worker *w, const tt_entry *e)
{   /* evicted at the end of the turn in which age > max_depth (agent.py:101-110) <=> now - touch > max_depth + 1 at a later lookup */
    return e->state != 0 && (w->now - e->touch) <= w->max_depth + 1;
}

static void tt_rebuild(mc_worker *w, uint64_t new_cap)
{
    tt_entry *old = w->tt;
    const uint64_t old_cap = w->tt_cap;
    w->tt = (tt_entry *)calloc(new_cap, sizeof(tt_entry));
    w->tt_cap = new_cap;
    w->tt_used = 0;
    for (uint64_t i = 0; i < old_cap; ++i) {
        const tt_entry *e = &old[i];
        if (e->state == 0 || (w->now - e->touch) > w->max_depth) continue;      /* same test the reference applies at the turn's end */
        uint64_t j = e->lo & (new_cap - 1);
        while (w->tt[j].state) j = (j + 1) & (new_cap - 1);
        w->tt[j] = *e;
        w->tt_used++;
    }
    free(old);
}

/* returns the entry index of the key; *fresh = 1 when this call created it (cache miss, agent.py:181-184) */
static int64_t tt_lookup_insert(mc_worker *w, const uint64_t key[2], int *fresh)
{
    const uint64_t mask = w->tt_cap - 1;
    uint64_t j = key[0] & mask;
    *fresh = 0;
    for (;;) {
        tt_entry *e = &w->tt[j];
        if (e->state == 0) {
            if (w->tt_used + 2 >= w->tt_cap) { w->overflow = 1; return (int64_t)j; }   /* caller checks mc_overflow: table too small */
            e->lo = key[0]; e->hi = key[1]; e->state = 1; e->touch = w->now;
            w->tt_used++;
            *fresh = 1;
            return (int64_t)j;
        }
        if (e->lo == key[0] && e->hi == key[1]) {
            if (!entry_valid(w, e)) { e->state = 1; *fresh = 1; }      /* evicted earlier: a miss again, re-created in place */
            e->touch = w->now;                                          /* cache_hit[key] = 0 (agent.py:185) */
            return (int64_t)j;
        }
        j = (j + 1) & mask;
    }
}

/* ------------------------------------------------------------------------------------------- construction */
mc_worker *mc_create(int H, int W, int S, int health_dec, int n_games, double base, int training, int max_depth,
                     int max_breadth, int stub, int keep_records, uint64_t seed, int tt_log2)
{
    mc_worker *w = (mc_worker *)calloc(1, sizeof(mc_worker));
    w->H = H; w->W = W; w->S = S; w->health_dec = health_dec; w->n_games = n_games;
    w->base = (float)base; w->training = training; w->max_depth = max_depth; w->max_breadth = max_breadth;
    w->stub = stub; w->keep_records = keep_records;
    w->gh = 2 * H - 1;
    w->obs_elems = w->gh * (2 * W - 1) * 3;
    w->roots = (orc_game *)calloc((size_t)n_games, sizeof(orc_game));
    w->root_live = (uint8_t *)calloc((size_t)n_games, 1);
    const int par = max_breadth < PARALLEL ? max_breadth : PARALLEL;
    w->n_subs = w->cap_subs = n_games * par;
    w->subs = (orc_game *)calloc((size_t)w->n_subs, sizeof(orc_game));
    w->sub_depth = (int *)calloc((size_t)w->n_subs, sizeof(int));
    w->sub_live = (int *)calloc((size_t)w->n_subs, sizeof(int));
    const int D = max_depth > 1 ? max_depth : 1;
    w->path_entry = (int32_t *)calloc((size_t)w->n_subs * S * D, sizeof(int32_t));
    w->path_move = (uint8_t *)calloc((size_t)w->n_subs * S * D, 1);
    w->path_len = (int32_t *)calloc((size_t)w->n_subs * S, sizeof(int32_t));
    const size_t max_rows = (size_t)w->n_subs * S;
    w->row_sub = (int *)calloc(max_rows, sizeof(int));
    w->row_snake = (int *)calloc(max_rows, sizeof(int));
    w->row_entry = (int64_t *)calloc(max_rows, sizeof(int64_t));
    w->fresh_entry = (int64_t *)calloc(max_rows, sizeof(int64_t));
    w->scratch_state = (float *)calloc((size_t)w->obs_elems, sizeof(float));
    w->last_ids = (int32_t *)calloc((size_t)n_games * S * 2, sizeof(int32_t));
    w->last_V = (float *)calloc((size_t)n_games * S * 3, sizeof(float));
    w->last_moves = (uint8_t *)calloc((size_t)n_games * S, 1);
    w->tt_cap = 1ull << tt_log2;
    w->tt = (tt_entry *)calloc(w->tt_cap, sizeof(tt_entry));
    uint64_t s = seed;
    for (int i = 0; i < 4; ++i) w->rng[i] = splitmix(&s);
    return w;
}

void mc_destroy(mc_worker *w)
{
    if (!w) return;
    free(w->roots); free(w->root_live); free(w->subs); free(w->sub_depth); free(w->sub_live);
    free(w->path_entry); free(w->path_move); free(w->path_len); free(w->row_sub); free(w->row_snake);
    free(w->row_entry); free(w->fresh_entry); free(w->scratch_state); free(w->last_ids); free(w->last_V);
    free(w->last_moves); free(w->tt); free(w->records); free(w->values);
    free(w);
}

int mc_sizeof_game(void) { return (int)sizeof(orc_game); }
void mc_set_game(mc_worker *w, int g, const orc_game *src) { w->roots[g] = *src; if (!w->root_live[g]) { w->root_live[g] = 1; w->n_live++; } }
void mc_get_game(const mc_worker *w, int g, orc_game *dst) { *dst = w->roots[g]; }
void mc_set_tape(mc_worker *w, const double *tape, int64_t n) { w->tape = tape; w->tape_len = n; w->tape_pos = 0; }
int64_t mc_tape_pos(const mc_worker *w) { return w->tape_pos; }
int mc_max_rows(const mc_worker *w) { return w->cap_subs * w->S; }
int mc_overflow(const mc_worker *w) { return w->overflow; }
int mc_n_live(const mc_worker *w) { return w->n_live; }

static int n_alive(const orc_game *g)
{
    int n = 0;
    for (int s = 0; s < g->S; ++s) n += g->alive[s] != 0;
    return n;
}

/* "for key in self.cache_hit: self.cache_hit[key] += 1" (agent.py:30-31) */
void mc_begin_turn(mc_worker *w) { w->now += 1; w->epoch = 0; }

int mc_epochs(const mc_worker *w)
{
    const int par = w->max_breadth < PARALLEL ? w->max_breadth : PARALLEL;
    return w->max_breadth / par;                        /* agent.py:37 */
}

/* sub-game creation (agent.py:39-50) + MCTSAgent's empty paths (agent.py:158-159) */
void mc_begin_epoch(mc_worker *w)
{
    const int par = w->max_breadth < PARALLEL ? w->max_breadth : PARALLEL;
    int b = 0;
    w->n_sub_live = 0;
    for (int g = 0; g < w->n_games; ++g) {
        if (!w->root_live[g]) continue;
        const int depth = w->max_depth - 2 * (n_alive(&w->roots[g]) - 2);    /* agent.py:45 */
        for (int p = 0; p < par; ++p, ++b) {
            orc_subgame(&w->roots[g], &w->subs[b]);
            w->sub_depth[b] = depth;
            w->sub_live[w->n_sub_live++] = b;
            for (int s = 0; s < w->S; ++s) w->path_len[b * w->S + s] = w->subs[b].alive[s] ? 0 : -1;
        }
    }
    w->n_subs = b;
    w->tick = 0;
}

/* first half of MCTSAgent.make_moves (agent.py:172-186): observations, keys, cache de-duplication.
 * planes_out / mask_out receive the observations (and their obstacle masks) that need a net evaluation.
 * Returns their count, or -1 when no sub-game is live any more (mp_game_runner.py:91). */
int mc_collect(mc_worker *w, float *planes_out, uint8_t *mask_out)
{
    if (w->n_sub_live == 0) return -1;
    w->tick += 1;
    w->n_rows = 0;
    w->n_fresh = 0;
    const int gh = w->gh, gw = 2 * w->W - 1;
    for (int i = 0; i < w->n_sub_live; ++i) {
        const int b = w->sub_live[i];
        const orc_game *g = &w->subs[b];
        for (int s = 0; s < w->S; ++s) {
            if (!g->alive[s]) continue;
            orc_make_state(g, s, w->scratch_state);                       /* Game.get_states (agent.py:173) */
            uint64_t key[2];
            orc_obs_key(w->scratch_state, gh * gw, key);                  /* state.tostring() (agent.py:175) */
            int fresh;
            const int64_t e = tt_lookup_insert(w, key, &fresh);
            w->lookups++;
            const int r = w->n_rows++;
            w->row_sub[r] = b; w->row_snake[r] = s; w->row_entry[r] = e;
            if (fresh) {
                const int j = w->n_fresh++;
                w->fresh_entry[j] = e;
                if (w->stub) {
                    orc_stub_q(w->scratch_state, gh, gw, w->tt[e].total);     /* evaluated on the spot */
                } else {
                    memcpy(planes_out + (size_t)j * w->obs_elems, w->scratch_state, sizeof(float) * (size_t)w->obs_elems);
                    if (mask_out) orc_obstacle_mask(w->scratch_state, gh, gw, 0, mask_out + 3 * j);
                }
            }
        }
    }
    w->net_evals += w->n_fresh;
    return w->n_fresh;
}

/* second half of MCTSAgent.make_moves (agent.py:189-222) + the tic / retire part of MCTSMPGameRunner.run
 * (mp_game_runner.py:99-113).  q: n_fresh x 3 float32 = nnet.v(all_states) (already obstacle-masked), NULL in stub mode */
void mc_apply(mc_worker *w, const float *q)
{
    const int S = w->S, D = w->max_depth > 1 ? w->max_depth : 1;
    for (int j = 0; j < w->n_fresh; ++j) {                                /* agent.py:193-201 */
        tt_entry *e = &w->tt[w->fresh_entry[j]];
        for (int m = 0; m < 3; ++m) {
            if (q) e->total[m] = q[3 * j + m];
            e->visit[m] = 1.0f;
            e->q[m] = e->total[m] / e->visit[m];
        }
        e->state = 2;
    }
    /* pmf and move of every row first (agent.py:204-205 run before any back-up of this tick) */
    float *pmf = (float *)malloc(sizeof(float) * 3 * (size_t)(w->n_rows > 0 ? w->n_rows : 1));
    uint8_t *mv = (uint8_t *)malloc((size_t)(w->n_rows > 0 ? w->n_rows : 1));
    for (int r = 0; r < w->n_rows; ++r) softermax3(w->base, w->tt[w->row_entry[r]].q, pmf + 3 * r);
    for (int r = 0; r < w->n_rows; ++r) mv[r] = (uint8_t)choice3(w, pmf + 3 * r);
    /* in-rollout back-up, rows in ids order, statistics re-read live (agent.py:208-222) */
    for (int r = 0; r < w->n_rows; ++r) {
        const int b = w->row_sub[r], s = w->row_snake[r];
        const float *v = w->tt[w->row_entry[r]].q;
        const float est = (pmf[3 * r] * v[0] + pmf[3 * r + 1] * v[1]) + pmf[3 * r + 2] * v[2];
        int32_t *pe = w->path_entry + ((size_t)b * S + s) * D;
        uint8_t *pm = w->path_move + ((size_t)b * S + s) * D;
        int32_t *pl = &w->path_len[b * S + s];
        for (int k = *pl - 1; k >= 0; --k) {
            tt_entry *a = &w->tt[pe[k]];
            const int m = pm[k];
            a->visit[m] += 1.0f;
            a->total[m] += est;
            a->q[m] = a->total[m] / a->visit[m];
        }
        if (*pl < D) { pe[*pl] = (int32_t)w->row_entry[r]; pm[*pl] = mv[r]; *pl += 1; }
    }
    /* game.tic for every live sub-game (mp_game_runner.py:104-113) */
    uint8_t dense[ORC_MAX_S];
    int r = 0, n_next = 0;
    for (int i = 0; i < w->n_sub_live; ++i) {
        const int b = w->sub_live[i];
        for (int s = 0; s < S; ++s) dense[s] = 1;
        while (r < w->n_rows && w->row_sub[r] == b) { dense[w->row_snake[r]] = mv[r]; ++r; }
        const int ended = orc_tic(&w->subs[b], dense, 0, -1, 0.0, 0.0, 0, 0);
        w->sim_steps++;
        if (!(ended || w->tick >= w->sub_depth[b])) w->sub_live[n_next++] = b;
    }
    w->n_sub_live = n_next;
    free(pmf); free(mv);
}

/* terminal back-up (agent.py:60-72): sub-games in creation order, snakes alive at the epoch's start in id order */
void mc_end_epoch(mc_worker *w)
{
    const int S = w->S, D = w->max_depth > 1 ? w->max_depth : 1;
    for (int b = 0; b < w->n_subs; ++b)
        for (int s = 0; s < S; ++s) {
            const int32_t pl = w->path_len[b * S + s];
            const int8_t rw = w->subs[b].rewards[s];
            if (pl < 0 || rw == 0) continue;
            const float reward = (float)rw;
            const int32_t *pe = w->path_entry + ((size_t)b * S + s) * D;
            const uint8_t *pm = w->path_move + ((size_t)b * S + s) * D;
            for (int k = pl - 1; k >= 0; --k) {
                tt_entry *a = &w->tt[pe[k]];
                const int m = pm[k];
                a->visit[m] += 1.0f;
                a->total[m] += reward;
                a->q[m] = a->total[m] / a->visit[m];
            }
        }
    w->epoch += 1;
}

void mc_set_spawn_tape(mc_worker *w, const int16_t *tape) { w->spawn_tape = tape; }

/* rest of Agent.make_moves (agent.py:74-110) and of the root turn of MPGameRunner.run (mp_game_runner.py:44-66).
 * Returns the number of root games still live. */
int mc_end_turn(mc_worker *w)
{
    const int S = w->S, D = w->max_depth > 1 ? w->max_depth : 1;
    const int par = w->max_breadth < PARALLEL ? w->max_breadth : PARALLEL;
    /* V[i] = cached_values[first key of clone 0] (agent.py:74-87) */
    int n = 0, b = 0;
    for (int g = 0; g < w->n_games; ++g) {
        if (!w->root_live[g]) continue;
        for (int s = 0; s < S; ++s) {
            if (!w->roots[g].alive[s]) continue;
            const tt_entry *e = &w->tt[w->path_entry[((size_t)b * S + s) * D]];
            w->last_ids[2 * n] = g; w->last_ids[2 * n + 1] = s;
            memcpy(w->last_V + 3 * n, e->q, 3 * sizeof(float));
            ++n;
        }
        b += par;
    }
    w->last_n = n;
    if (w->training) {                                                    /* agent.py:90-97 */
        for (int i = 0; i < n; ++i) {
            float p[3];
            softermax3(w->base, w->last_V + 3 * i, p);
            w->last_moves[i] = (uint8_t)choice3(w, p);
        }
        if (w->keep_records) {
            if (w->n_records + n > w->cap_records) {
                w->cap_records = 2 * (w->n_records + n) + 64;
                w->records = (float *)realloc(w->records, sizeof(float) * (size_t)w->cap_records * w->obs_elems);
                w->values = (float *)realloc(w->values, sizeof(float) * (size_t)w->cap_records * 3);
            }
            for (int i = 0; i < n; ++i) {
                orc_make_state(&w->roots[w->last_ids[2 * i]], w->last_ids[2 * i + 1],
                               w->records + (size_t)(w->n_records + i) * w->obs_elems);
                memcpy(w->values + (size_t)(w->n_records + i) * 3, w->last_V + 3 * i, 3 * sizeof(float));
            }
        }
        w->n_records += n;
    } else {                                                              /* Agent.argmaxs (agent.py:124-137) */
        for (int i = 0; i < n; ++i) {
            const float *z = w->last_V + 3 * i;
            w->last_moves[i] = (uint8_t)(z[0] > z[1] ? (z[0] > z[2] ? 0 : 2) : (z[1] > z[2] ? 1 : 2));
        }
    }
    /* RAM recycle (agent.py:101-110) is lazy (entry_valid); the table is compacted when half full */
    if (w->tt_used * 2 > w->tt_cap) {
        tt_rebuild(w, w->tt_cap);
        if (w->tt_used * 2 > w->tt_cap) tt_rebuild(w, w->tt_cap * 2);
    }
    /* root Game.tic (mp_game_runner.py:50-66) */
    uint8_t dense[ORC_MAX_S];
    int i = 0;
    for (int g = 0; g < w->n_games; ++g) {
        if (!w->root_live[g]) continue;
        for (int s = 0; s < S; ++s) dense[s] = 1;
        while (i < n && w->last_ids[2 * i] == g) { dense[w->last_ids[2 * i + 1]] = w->last_moves[i]; ++i; }
        int ended;
        if (w->spawn_tape) ended = orc_tic(&w->roots[g], dense, 0, w->spawn_tape[g], 0.0, 0.0, 0, 0);
        else { const double u1 = next_u(w), u2 = next_u(w); ended = orc_tic(&w->roots[g], dense, 1, -1, u1, u2, 0, 0); }
        w->env_steps++;
        if (ended) {
            for (int k = 0; k < 6; ++k) w->totals[k] += w->roots[g].counters[k];
            w->root_live[g] = 0;
            w->n_live--;
        }
    }
    return w->n_live;
}

/* number of cache entries the reference would hold right now (after this turn's eviction) */
int64_t mc_cache_size(const mc_worker *w)
{
    int64_t n = 0;
    for (uint64_t i = 0; i < w->tt_cap; ++i)
        n += w->tt[i].state != 0 && (w->now - w->tt[i].touch) <= w->max_depth;
    return n;
}

/* whole root turns with the stub net, no caller in between (engine-only throughput) */
int mc_run_stub_turns(mc_worker *w, int turns)
{
    for (int t = 0; t < turns && w->n_live > 0; ++t) {
        mc_begin_turn(w);
        const int epochs = mc_epochs(w);
        for (int e = 0; e < epochs; ++e) {
            mc_begin_epoch(w);
            while (mc_collect(w, 0, 0) >= 0) mc_apply(w, 0);
            mc_end_epoch(w);
        }
        mc_end_turn(w);
    }
    return w->n_live;
}

/* accessors */
int mc_last(const mc_worker *w, int32_t *ids, float *V, uint8_t *moves)
{
    memcpy(ids, w->last_ids, sizeof(int32_t) * 2 * (size_t)w->last_n);
    memcpy(V, w->last_V, sizeof(float) * 3 * (size_t)w->last_n);
    memcpy(moves, w->last_moves, (size_t)w->last_n);
    return w->last_n;
}
void mc_stats(const mc_worker *w, int64_t *out8)
{
    out8[0] = w->env_steps; out8[1] = w->sim_steps; out8[2] = w->net_evals; out8[3] = w->lookups;
    out8[4] = w->n_records; out8[5] = (int64_t)w->tt_used; out8[6] = (int64_t)w->tt_cap; out8[7] = w->now;
}
void mc_totals(const mc_worker *w, int64_t *out6) { memcpy(out6, w->totals, sizeof(w->totals)); }
const float *mc_records(const mc_worker *w) { return w->records; }
const float *mc_values(const mc_worker *w) { return w->values; }
