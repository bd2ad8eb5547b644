// mcts.hip -- device side of the randomized parallel MCTS (reference: Agent.make_moves agent.py:25-111,
// MCTSAgent.make_moves agent.py:161-223, Agent.softermax / argmaxs agent.py:114-137).
//
// The reference keeps four Python dicts keyed by the 5 292-byte observation string
// (cached_values / total_rewards / visit_cnts / cache_hit, agent.py:16-19).  Here they are ONE
// open-addressing hash table in HBM keyed by the 128-bit observation digest of engine.hip:
//
//   key_lo[cap], key_hi[cap]   u64   (0 = empty; key_hi is published last, with release order)
//   stat[cap][8]               f32   total[3], visit[3], touch (i32 bits), pad -- one 32-byte line
//
// cached_values[key][m] is always total[m] / visit[m] (agent.py:72, 199, 220), so Q is not stored.
// cache_hit[key] is "root turns since last touch" = now - touch; the reference evicts at the end of a
// root turn when it exceeds max_depth (agent.py:101-110), so during turn `now` an entry exists iff
// now - touch <= max_age + 1.  Stale entries are treated as misses in place (the first finder
// re-creates the entry: fresh prior, visits 1,1,1) and are dropped physically by snk_tt_rebuild.
#include "common.h"

#define TT_NONE 0xFFFFFFFFu

struct snk_tt {
    uint64_t cap, mask;
    unsigned long long *key_lo, *key_hi;
    float *stat;
    int *d_ctrl;     // [0] occupied slots, [1] error flag (table full), [2] scratch counter
    int device;
};

__device__ static inline unsigned long long ld_u64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// MCTSAgent.make_moves, "get states without duplicates" (agent.py:170-186)
// ------------------------------------------------------------------------------------------
__global__ void k_tt_lookup_insert(snk_tt T, const unsigned long long *__restrict__ key, const uint8_t *__restrict__ active,
                                   int m, int now, int max_age, uint32_t *__restrict__ entry, uint8_t *__restrict__ is_new)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long klo = key[2 * (size_t)i], khi = key[2 * (size_t)i + 1];
    if ((active && !active[i]) || (klo == 0ull && khi == 0ull)) { entry[i] = TT_NONE; is_new[i] = 0; return; }
    if (klo == 0ull) klo = 1ull;          // 0 is the "empty" / "unpublished" sentinel of both words
    if (khi == 0ull) khi = 1ull;
    uint64_t slot = klo & T.mask;
    uint32_t found = TT_NONE;
    uint8_t fresh = 0;
    for (uint64_t probes = 0; probes <= T.cap;) {
        unsigned long long cur = ld_u64(&T.key_lo[slot]);
        if (cur == 0ull) {
            const unsigned long long prev = atomicCAS(&T.key_lo[slot], 0ull, klo);
            if (prev == 0ull) {            // claimed an empty slot: this lane creates the entry
                int *touch = (int *)&T.stat[slot * 8 + 6];
                __hip_atomic_store(touch, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&T.key_hi[slot], khi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&T.d_ctrl[0], 1);
                found = (uint32_t)slot; fresh = 1;
                break;
            }
            cur = prev;
        }
        if (cur == klo) {
            const unsigned long long hi = __hip_atomic_load(&T.key_hi[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (hi == 0ull) continue;      // the claimer has not published yet: look again (no inner spin)
            if (hi == khi) {
                int *touch = (int *)&T.stat[slot * 8 + 6];
                const int t = __hip_atomic_load(touch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (now - t > max_age + 1) {                 // evicted in the reference: a miss
                    if (atomicCAS(touch, t, now) == t) fresh = 1;
                } else if (t != now) {
                    __hip_atomic_store(touch, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // cache_hit[key] = 0
                }
                found = (uint32_t)slot;
                break;
            }
        }
        slot = (slot + 1) & T.mask;
        ++probes;
    }
    if (found == TT_NONE) atomicExch(&T.d_ctrl[1], 1);
    entry[i] = found;
    is_new[i] = fresh;
}

// read-only probe (the reference's `key in cache` / cache[key], agent.py:16-19): entry index and the raw statistics
// total[3], visit[3], age = now - touch of keys that exist and have not been evicted; TT_NONE otherwise.  No insert, no touch.
__global__ void k_tt_find(snk_tt T, const unsigned long long *__restrict__ key, int m, int now, int max_age,
                          uint32_t *__restrict__ entry, float *__restrict__ stat_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long klo = key[2 * (size_t)i], khi = key[2 * (size_t)i + 1];
    uint32_t found = TT_NONE;
    if (!(klo == 0ull && khi == 0ull)) {
        if (klo == 0ull) klo = 1ull;
        if (khi == 0ull) khi = 1ull;
        uint64_t slot = klo & T.mask;
        for (uint64_t probes = 0; probes <= T.cap; ++probes) {
            const unsigned long long cur = ld_u64(&T.key_lo[slot]);
            if (cur == 0ull) break;
            if (cur == klo && __hip_atomic_load(&T.key_hi[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == khi) {
                const int t = __float_as_int(T.stat[slot * 8 + 6]);
                if (now - t <= max_age + 1) found = (uint32_t)slot;
                break;
            }
            slot = (slot + 1) & T.mask;
        }
    }
    entry[i] = found;
    if (stat_out) {
        float *o = stat_out + 7 * (size_t)i;
        if (found == TT_NONE) { for (int k = 0; k < 7; ++k) o[k] = 0.f; }
        else {
            const float *s = &T.stat[(size_t)found * 8];
            for (int k = 0; k < 6; ++k) o[k] = s[k];
            o[6] = (float)(now - __float_as_int(s[6]));
        }
    }
}

// new entries: prior = net output, visits 1,1,1 (agent.py:193-201)
// TICK GATE (round 4): the kernels of a rollout tick that follow the leaf evaluation -- priors, move choice, back-up, the
// sub-games' tic, retirement -- take an optional device word and do NOTHING when it is non-zero.  The word is the Q-net's range
// guard (set by a convolution launch that had to clamp an input): a tick whose evaluation cannot be trusted leaves no trace, the
// host notices at the next tick's existing read-back and runs that evaluation and these kernels again (snake_engine/mcts.py).
#define TICK_GATE(skip) if ((skip) && *(volatile const int *)(skip)) return;

__global__ void k_tt_set_priors(snk_tt T, const uint32_t *__restrict__ entry, const int32_t *__restrict__ idx, int n,
                                const float *__restrict__ q, const int *__restrict__ skip)
{
    TICK_GATE(skip)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t e = entry[idx ? idx[j] : j];
    if (e == TT_NONE) return;
    float *s = &T.stat[(size_t)e * 8];
    s[0] = q[3 * (size_t)j]; s[1] = q[3 * (size_t)j + 1]; s[2] = q[3 * (size_t)j + 2];
    s[3] = 1.0f; s[4] = 1.0f; s[5] = 1.0f;
}

// ------------------------------------------------------------------------------------------
// Agent.softermax (agent.py:114-122) and numpy.random.choice([0,1,2], p=pmf) (agent.py:91, 205)
// ------------------------------------------------------------------------------------------
__device__ static inline void softermax3(float base, const float q[3], float pmf[3])
{
    float nrm[3];
    int inf_cnt = 0;
    for (int k = 0; k < 3; ++k) {
        nrm[k] = powf(base, atanhf(q[k]));      // power(base, arctanh(z)), float32
        if (isinf(nrm[k])) ++inf_cnt;
    }
    if (inf_cnt) {                              // z == +1.0: the reference divides inf/inf -> NaN and
        for (int k = 0; k < 3; ++k) pmf[k] = isinf(nrm[k]) ? 1.0f / (float)inf_cnt : 0.0f;   // numpy raises; clamp instead
        return;
    }
    const float sigma = (nrm[0] + nrm[1]) + nrm[2];        // python sum(): left to right
    if (sigma == 0.0f) { pmf[0] = pmf[1] = pmf[2] = (float)(1.0 / 3.0); return; }
    for (int k = 0; k < 3; ++k) pmf[k] = nrm[k] / sigma;
}

__device__ static inline int choice3(const float pmf[3], double u)
{
    double c0 = (double)pmf[0], c1 = c0 + (double)pmf[1], c2 = c1 + (double)pmf[2];   // cdf = p.cumsum()
    c0 /= c2; c1 /= c2;                                                                 // cdf /= cdf[-1]
    return (u >= c0) + (u >= c1);                                                       // searchsorted(cdf, u, 'right')
}

__device__ static inline int argmax3(const float z[3])      // Agent.argmaxs (agent.py:124-137)
{
    if (z[0] > z[1]) return (z[0] > z[2]) ? 0 : 2;
    return (z[1] > z[2]) ? 1 : 2;
}

__device__ static inline double uniform_draw(const double *tape, const int32_t *rank, long tape_base, int i,
                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t ctr0, uint32_t ctr1)
{
    if (tape) return tape[tape_base + (rank ? rank[i] : i)];
    uint32_t r[4];
    philox4x32((uint32_t)i, ctr0, ctr1, 0x4D435453u /* 'MCTS' */, seed_lo, seed_hi, r);
    return ((double)r[0] * 4294967296.0 + (double)r[1] + 0.5) * (1.0 / 18446744073709551616.0);
}

struct PathBufs {
    uint32_t *entry;   // [m][D]
    uint8_t *move;     // [m][D]
    int32_t *len;      // [m]
    int D;
};

// rollout move choice (agent.py:203-205) + append (key, move) to the snake's path (agent.py:221-222).
// est = pmf . Q (agent.py:214) is computed here from the statistics as they stand before this tick's back-ups.
__global__ void k_mcts_select(snk_tt T, const uint32_t *__restrict__ entry, int m, float base, const double *__restrict__ tape,
                              const int32_t *__restrict__ rank, long tape_base, uint32_t seed_lo, uint32_t seed_hi,
                              uint32_t ctr0, uint32_t ctr1, uint8_t *__restrict__ moves, float *__restrict__ est,
                              float *__restrict__ pmf_out, PathBufs P, const int *__restrict__ skip)
{
    TICK_GATE(skip)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t e = entry[i];
    if (e == TT_NONE) { moves[i] = 1; if (est) est[i] = 0.f; return; }
    const float *s = &T.stat[(size_t)e * 8];
    float q[3], pmf[3];
    for (int k = 0; k < 3; ++k) q[k] = s[k] / s[3 + k];
    softermax3(base, q, pmf);
    const double u = uniform_draw(tape, rank, tape_base, i, seed_lo, seed_hi, ctr0, ctr1);
    const int mv = choice3(pmf, u);
    moves[i] = (uint8_t)mv;
    if (est) est[i] = (pmf[0] * q[0] + pmf[1] * q[1]) + pmf[2] * q[2];
    if (pmf_out) { pmf_out[3 * (size_t)i] = pmf[0]; pmf_out[3 * (size_t)i + 1] = pmf[1]; pmf_out[3 * (size_t)i + 2] = pmf[2]; }
    const int L = P.len[i];
    if (L < P.D) { P.entry[(size_t)i * P.D + L] = e; P.move[(size_t)i * P.D + L] = (uint8_t)mv; }
    // len is advanced by the back-up kernel, after the ancestors (the path BEFORE this append) were updated
}

// in-rollout back-up (agent.py:208-220): every ancestor edge of the snake's path gets visit += 1, total += est
__global__ void k_mcts_backup(snk_tt T, const uint32_t *__restrict__ entry, int m, const float *__restrict__ est, PathBufs P,
                              const int *__restrict__ skip)
{
    TICK_GATE(skip)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (entry[i] == TT_NONE) return;
    const int L = P.len[i];
    const float r = est[i];
    for (int j = L - 1; j >= 0; --j) {
        float *s = &T.stat[(size_t)P.entry[(size_t)i * P.D + j] * 8];
        const int mv = P.move[(size_t)i * P.D + j];
        atomicAdd(&s[3 + mv], 1.0f);
        atomicAdd(&s[mv], r);
    }
    if (L < P.D) P.len[i] = L + 1;
}

// the same in the reference's sequential order (ids order, live Q reads): parity runs on tiny cases.
// pmf was fixed before the loop (agent.py:204); V[i] aliases the live cache row (agent.py:181, 214).
__global__ void k_mcts_backup_seq(snk_tt T, const uint32_t *__restrict__ entry, int m, const float *__restrict__ pmf, PathBufs P,
                                  const int *__restrict__ skip)
{
    TICK_GATE(skip)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int i = 0; i < m; ++i) {
        const uint32_t e = entry[i];
        if (e == TT_NONE) continue;
        const float *se = &T.stat[(size_t)e * 8];
        float q[3];
        for (int k = 0; k < 3; ++k) q[k] = se[k] / se[3 + k];
        const float r = (pmf[3 * (size_t)i] * q[0] + pmf[3 * (size_t)i + 1] * q[1]) + pmf[3 * (size_t)i + 2] * q[2];
        const int L = P.len[i];
        for (int j = L - 1; j >= 0; --j) {
            float *s = &T.stat[(size_t)P.entry[(size_t)i * P.D + j] * 8];
            const int mv = P.move[(size_t)i * P.D + j];
            s[3 + mv] += 1.0f;
            s[mv] += r;
        }
        if (L < P.D) P.len[i] = L + 1;
    }
}

// terminal back-up at the end of an epoch (agent.py:60-72); rewards: +1 / -1, 0 = None (alive at truncation)
__global__ void k_mcts_terminal(snk_tt T, const int8_t *__restrict__ rewards, int m, PathBufs P, int sequential)
{
    if (sequential) {
        if (blockIdx.x != 0 || threadIdx.x != 0) return;
        for (int i = 0; i < m; ++i) {
            if (!rewards[i]) continue;
            const float r = (float)rewards[i];
            for (int j = P.len[i] - 1; j >= 0; --j) {
                float *s = &T.stat[(size_t)P.entry[(size_t)i * P.D + j] * 8];
                const int mv = P.move[(size_t)i * P.D + j];
                s[3 + mv] += 1.0f;
                s[mv] += r;
            }
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !rewards[i]) return;
    const float r = (float)rewards[i];
    for (int j = P.len[i] - 1; j >= 0; --j) {
        float *s = &T.stat[(size_t)P.entry[(size_t)i * P.D + j] * 8];
        const int mv = P.move[(size_t)i * P.D + j];
        atomicAdd(&s[3 + mv], 1.0f);
        atomicAdd(&s[mv], r);
    }
}

// Q of arbitrary entries (root values: cached_values[first_key], agent.py:83-87)
__global__ void k_tt_read_q(snk_tt T, const uint32_t *__restrict__ entry, int stride, int m, float *__restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t e = entry[(size_t)i * stride];
    if (e == TT_NONE) { q[3 * (size_t)i] = q[3 * (size_t)i + 1] = q[3 * (size_t)i + 2] = 0.f; return; }
    const float *s = &T.stat[(size_t)e * 8];
    for (int k = 0; k < 3; ++k) q[3 * (size_t)i + k] = s[k] / s[3 + k];
}

// root move (agent.py:89-99): training -> sample from softermax(V); play -> argmaxs(V)
__global__ void k_root_moves(const float *__restrict__ V, const uint8_t *__restrict__ alive, int m, float base, int training,
                             const double *__restrict__ tape, const int32_t *__restrict__ rank, long tape_base,
                             uint32_t seed_lo, uint32_t seed_hi, uint32_t ctr0, uint32_t ctr1, uint8_t *__restrict__ moves)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (!alive[i]) { moves[i] = 1; return; }
    const float z[3] = {V[3 * (size_t)i], V[3 * (size_t)i + 1], V[3 * (size_t)i + 2]};
    if (training) {
        float pmf[3];
        softermax3(base, z, pmf);
        moves[i] = (uint8_t)choice3(pmf, uniform_draw(tape, rank, tape_base, i, seed_lo, seed_hi, ctr0, ctr1));
    } else {
        moves[i] = (uint8_t)argmax3(z);
    }
}

// eviction (agent.py:101-110): survivors of the end-of-turn test (now - touch <= max_age) move to a fresh table
__global__ void k_tt_rebuild(snk_tt Old, snk_tt New, int now, int max_age)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < Old.cap; s += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long klo = Old.key_lo[s];
        if (klo == 0ull) continue;
        const float *so = &Old.stat[s * 8];
        const int t = __float_as_int(so[6]);
        if (now - t > max_age) continue;
        uint64_t slot = klo & New.mask;
        for (uint64_t probes = 0; probes <= New.cap; ++probes) {
            if (atomicCAS(&New.key_lo[slot], 0ull, klo) == 0ull) {
                New.key_hi[slot] = Old.key_hi[s];
                float *sn = &New.stat[slot * 8];
                for (int k = 0; k < 8; ++k) sn[k] = so[k];
                atomicAdd(&New.d_ctrl[0], 1);
                break;
            }
            slot = (slot + 1) & New.mask;
        }
    }
}

__global__ void k_softermax_table(const float *__restrict__ z, int m, float base, float *__restrict__ pmf, uint8_t *__restrict__ am)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float q[3] = {z[3 * i], z[3 * i + 1], z[3 * i + 2]};
    float p[3];
    softermax3(base, q, p);
    pmf[3 * i] = p[0]; pmf[3 * i + 1] = p[1]; pmf[3 * i + 2] = p[2];
    am[i] = (uint8_t)argmax3(q);
}

// Game.rewards of n games as int8[n][S] (0 none / +1 / -1) -- feeds the terminal back-up
__global__ void k_engine_rewards(const uint8_t *__restrict__ state, Layout L, const int32_t *__restrict__ slots, int n,
                                 int8_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L.S) return;
    const int gi = t / L.S, s = t - gi * L.S;
    const int slot = slots ? slots[gi] : gi;
    out[t] = ((const int8_t *)(state + (size_t)slot * L.stride + L.rew_off))[s];
}

// ------------------------------------------------------------------------------------------ C ABI
static int tt_alloc(snk_tt *t, uint64_t cap)
{
    t->cap = cap; t->mask = cap - 1;
    t->key_lo = t->key_hi = nullptr; t->stat = nullptr; t->d_ctrl = nullptr;
    hipError_t e = hipMalloc((void **)&t->key_lo, cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&t->key_hi, cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&t->stat, cap * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_ctrl, 16);
    if (e == hipSuccess) e = hipMemset(t->key_lo, 0, cap * 8);
    if (e == hipSuccess) e = hipMemset(t->key_hi, 0, cap * 8);
    if (e == hipSuccess) e = hipMemset(t->d_ctrl, 0, 16);
    if (e != hipSuccess) {
        snk_set_error("transposition table: allocating %llu entries failed: %s", (unsigned long long)cap, hipGetErrorString(e));
        (void)hipFree(t->key_lo); (void)hipFree(t->key_hi); (void)hipFree(t->stat); (void)hipFree(t->d_ctrl);
        return -2;
    }
    return 0;
}

static void tt_free(snk_tt *t)
{
    (void)hipFree(t->key_lo); (void)hipFree(t->key_hi); (void)hipFree(t->stat); (void)hipFree(t->d_ctrl);
}

extern "C" int snk_tt_create(snk_tt **out, uint64_t capacity, int device)
{
    SNK_REQUIRE(out != nullptr, "snk_tt_create: out is NULL");
    SNK_REQUIRE(capacity >= 1024 && (capacity & (capacity - 1)) == 0 && capacity <= (1ull << 31),
                "snk_tt_create: capacity must be a power of two in [2^10, 2^31]");
    SNK_CHECK_HIP(hipSetDevice(device));
    snk_tt *t = new snk_tt();
    t->device = device;
    const int rc = tt_alloc(t, capacity);
    if (rc) { delete t; return rc; }
    *out = t;
    return 0;
}

extern "C" int snk_tt_destroy(snk_tt *t)
{
    if (!t) return 0;
    tt_free(t);
    delete t;
    return 0;
}

extern "C" int snk_tt_clear(snk_tt *t, void *stream)      /* Agent.clear (agent.py:140-147) */
{
    SNK_REQUIRE(t != nullptr, "snk_tt_clear: table is NULL");
    SNK_CHECK_HIP(hipMemsetAsync(t->key_lo, 0, t->cap * 8, (hipStream_t)stream));
    SNK_CHECK_HIP(hipMemsetAsync(t->key_hi, 0, t->cap * 8, (hipStream_t)stream));
    SNK_CHECK_HIP(hipMemsetAsync(t->d_ctrl, 0, 16, (hipStream_t)stream));
    return 0;
}

extern "C" int snk_tt_status_sync(snk_tt *t, int64_t *capacity, int64_t *occupied, int *overflowed)
{
    SNK_REQUIRE(t != nullptr, "snk_tt_status_sync: table is NULL");
    int ctrl[4];
    SNK_CHECK_HIP(hipDeviceSynchronize());
    SNK_CHECK_HIP(hipMemcpy(ctrl, t->d_ctrl, 16, hipMemcpyDeviceToHost));
    if (capacity) *capacity = (int64_t)t->cap;
    if (occupied) *occupied = ctrl[0];
    if (overflowed) *overflowed = ctrl[1];
    return 0;
}

extern "C" int snk_tt_rebuild_sync(snk_tt *t, uint64_t new_capacity, int now_turn, int max_age)
{
    SNK_REQUIRE(t != nullptr, "snk_tt_rebuild_sync: table is NULL");
    SNK_REQUIRE(new_capacity >= 1024 && (new_capacity & (new_capacity - 1)) == 0 && new_capacity <= (1ull << 31),
                "snk_tt_rebuild_sync: capacity must be a power of two in [2^10, 2^31]");
    snk_tt fresh = *t;
    const int rc = tt_alloc(&fresh, new_capacity);
    if (rc) return rc;
    k_tt_rebuild<<<4096, 256>>>(*t, fresh, now_turn, max_age);
    SNK_CHECK_HIP(hipGetLastError());
    SNK_CHECK_HIP(hipDeviceSynchronize());
    tt_free(t);
    *t = fresh;
    return 0;
}

extern "C" int snk_tt_lookup_insert(snk_tt *t, const uint64_t *d_key, const uint8_t *d_active, int m, int now_turn,
                                    int max_age, uint32_t *d_entry, uint8_t *d_is_new, void *stream)
{
    SNK_REQUIRE(t && d_key && d_entry && d_is_new, "snk_tt_lookup_insert: NULL argument");
    if (m <= 0) return 0;
    k_tt_lookup_insert<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, (const unsigned long long *)d_key, d_active, m, now_turn,
                                                                         max_age, d_entry, d_is_new);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_tt_find(snk_tt *t, const uint64_t *d_key, int m, int now_turn, int max_age, uint32_t *d_entry,
                           float *d_stat7, void *stream)
{
    SNK_REQUIRE(t && d_key && d_entry, "snk_tt_find: NULL argument");
    if (m <= 0) return 0;
    k_tt_find<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, (const unsigned long long *)d_key, m, now_turn, max_age, d_entry,
                                                                d_stat7);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_tt_set_priors(snk_tt *t, const uint32_t *d_entry, const int32_t *d_idx, int n, const float *d_q,
                                 const int32_t *d_skip, void *stream)
{
    SNK_REQUIRE(t && d_entry && d_q, "snk_tt_set_priors: NULL argument");
    if (n <= 0) return 0;
    k_tt_set_priors<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, d_entry, d_idx, n, d_q, d_skip);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_select(snk_tt *t, const uint32_t *d_entry, int m, float softmax_base, const double *d_tape_u,
                               const int32_t *d_rank, int64_t tape_base, uint64_t seed, uint32_t ctr0, uint32_t ctr1,
                               uint8_t *d_moves, float *d_est, float *d_pmf, uint32_t *d_path_entry, uint8_t *d_path_move,
                               int32_t *d_path_len, int path_depth, const int32_t *d_skip, void *stream)
{
    SNK_REQUIRE(t && d_entry && d_moves && d_path_entry && d_path_move && d_path_len, "snk_mcts_select: NULL argument");
    if (m <= 0) return 0;
    PathBufs P = {d_path_entry, d_path_move, d_path_len, path_depth};
    k_mcts_select<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, d_entry, m, softmax_base, d_tape_u, d_rank, (long)tape_base,
                                                                    (uint32_t)seed, (uint32_t)(seed >> 32), ctr0, ctr1, d_moves, d_est,
                                                                    d_pmf, P, d_skip);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_backup(snk_tt *t, const uint32_t *d_entry, int m, const float *d_est, const float *d_pmf,
                               uint32_t *d_path_entry, uint8_t *d_path_move, int32_t *d_path_len, int path_depth,
                               int sequential, const int32_t *d_skip, void *stream)
{
    SNK_REQUIRE(t && d_entry && d_path_entry && d_path_move && d_path_len, "snk_mcts_backup: NULL argument");
    SNK_REQUIRE(sequential ? d_pmf != nullptr : d_est != nullptr, "snk_mcts_backup: needs d_pmf (sequential) or d_est");
    if (m <= 0) return 0;
    PathBufs P = {d_path_entry, d_path_move, d_path_len, path_depth};
    if (sequential) k_mcts_backup_seq<<<1, 64, 0, (hipStream_t)stream>>>(*t, d_entry, m, d_pmf, P, d_skip);
    else k_mcts_backup<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, d_entry, m, d_est, P, d_skip);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- the rollout loop's bookkeeping (mp_game_runner.py:99-113), one launch each instead of a chain of tensor expressions ----
// row (sub-game b, snake s) takes part in a tick when its snake is alive and the sub-game has not been retired
__global__ void k_mcts_row_active(const uint8_t *__restrict__ alive_rows, const uint8_t *__restrict__ sub_active, int m, int S,
                                  uint8_t *__restrict__ row_active)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) row_active[i] = (alive_rows[i] != 0 && sub_active[i / S] != 0) ? 1 : 0;
}

// after the tick's step: the sub-games that moved are counted (sim_steps += number of active ones), then a sub-game retires
// when its game is over or its depth cap is reached (tick >= depth: mp_game_runner.py:108-113)
__global__ void k_mcts_retire(uint8_t *__restrict__ sub_active, const uint8_t *__restrict__ done, const int32_t *__restrict__ sub_depth,
                              int tick, int B, unsigned long long *__restrict__ sim_steps, const int *__restrict__ skip)
{
    TICK_GATE(skip)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < B && sub_active[i] != 0;
    const unsigned long long moved = __ballot(act);
    if ((threadIdx.x & 63) == 0 && moved) atomicAdd(sim_steps, (unsigned long long)__popcll(moved));
    if (act && (done[i] != 0 || tick >= sub_depth[i])) sub_active[i] = 0;
}

// the rows a tick has to evaluate (the compacted indices of the new keys): their (sub-game, snake) pairs and obstacle masks
__global__ void k_mcts_gather_rows(const int32_t *__restrict__ idx, int n, const int32_t *__restrict__ pairs, const uint8_t *__restrict__ mask,
                                   int32_t *__restrict__ out_pairs, uint8_t *__restrict__ out_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = idx[i];
    out_pairs[2 * i] = pairs[2 * r]; out_pairs[2 * i + 1] = pairs[2 * r + 1];
    out_mask[3 * i] = mask[3 * r]; out_mask[3 * i + 1] = mask[3 * r + 1]; out_mask[3 * i + 2] = mask[3 * r + 2];
}

extern "C" int snk_mcts_gather_rows(const int32_t *d_idx, int n, const int32_t *d_pairs, const uint8_t *d_mask, int32_t *d_out_pairs,
                                    uint8_t *d_out_mask, void *stream)
{
    SNK_REQUIRE(n >= 0, "snk_mcts_gather_rows: negative n");
    if (n == 0) return 0;
    SNK_REQUIRE(d_idx && d_pairs && d_mask && d_out_pairs && d_out_mask, "snk_mcts_gather_rows: NULL argument");
    k_mcts_gather_rows<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_idx, n, d_pairs, d_mask, d_out_pairs, d_out_mask);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_row_active(const uint8_t *d_alive_rows, const uint8_t *d_sub_active, int n_subgames, int n_snakes,
                                   uint8_t *d_row_active, void *stream)
{
    SNK_REQUIRE(d_alive_rows && d_sub_active && d_row_active && n_snakes >= 1, "snk_mcts_row_active: bad argument");
    const long m = (long)n_subgames * n_snakes;
    if (m <= 0) return 0;
    SNK_REQUIRE(m < (1l << 31), "snk_mcts_row_active: too many rows");
    k_mcts_row_active<<<(int)((m + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_alive_rows, d_sub_active, (int)m, n_snakes, d_row_active);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_retire(uint8_t *d_sub_active, const uint8_t *d_done, const int32_t *d_sub_depth, int tick, int n_subgames,
                               int64_t *d_sim_steps, const int32_t *d_skip, void *stream)
{
    SNK_REQUIRE(d_sub_active && d_done && d_sub_depth && d_sim_steps, "snk_mcts_retire: NULL argument");
    if (n_subgames <= 0) return 0;
    k_mcts_retire<<<(n_subgames + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_sub_active, d_done, d_sub_depth, tick, n_subgames,
                                                                             (unsigned long long *)d_sim_steps, d_skip);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_terminal_backup(snk_tt *t, const int8_t *d_rewards, int m, uint32_t *d_path_entry, uint8_t *d_path_move,
                                        int32_t *d_path_len, int path_depth, int sequential, void *stream)
{
    SNK_REQUIRE(t && d_rewards && d_path_entry && d_path_move && d_path_len, "snk_mcts_terminal_backup: NULL argument");
    if (m <= 0) return 0;
    PathBufs P = {d_path_entry, d_path_move, d_path_len, path_depth};
    if (sequential) k_mcts_terminal<<<1, 64, 0, (hipStream_t)stream>>>(*t, d_rewards, m, P, 1);
    else k_mcts_terminal<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, d_rewards, m, P, 0);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_tt_read_q(snk_tt *t, const uint32_t *d_entry, int entry_stride, int m, float *d_q, void *stream)
{
    SNK_REQUIRE(t && d_entry && d_q, "snk_tt_read_q: NULL argument");
    if (m <= 0) return 0;
    k_tt_read_q<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(*t, d_entry, entry_stride, m, d_q);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_mcts_root_moves(const float *d_V, const uint8_t *d_alive, int m, float softmax_base, int training,
                                   const double *d_tape_u, const int32_t *d_rank, int64_t tape_base, uint64_t seed, uint32_t ctr0,
                                   uint32_t ctr1, uint8_t *d_moves, void *stream)
{
    SNK_REQUIRE(d_V && d_alive && d_moves, "snk_mcts_root_moves: NULL argument");
    if (m <= 0) return 0;
    k_root_moves<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_V, d_alive, m, softmax_base, training, d_tape_u, d_rank,
                                                                   (long)tape_base, (uint32_t)seed, (uint32_t)(seed >> 32), ctr0, ctr1, d_moves);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_softermax_argmax(const float *d_z, int m, float softmax_base, float *d_pmf, uint8_t *d_argmax, void *stream)
{
    SNK_REQUIRE(d_z && d_pmf && d_argmax, "snk_softermax_argmax: NULL argument");
    if (m <= 0) return 0;
    k_softermax_table<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_z, m, softmax_base, d_pmf, d_argmax);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int snk_engine_rewards(const snk_engine *e, const int32_t *d_slots, int n, int8_t *d_rewards, void *stream)
{
    SNK_REQUIRE(e && d_rewards, "snk_engine_rewards: NULL argument");
    SNK_REQUIRE(n >= 0 && (d_slots || n <= e->n_slots), "snk_engine_rewards: n=%d exceeds %d slots", n, e->n_slots);
    if (n == 0) return 0;
    const int tot = n * e->L.S;
    k_engine_rewards<<<(tot + 255) / 256, 256, 0, (hipStream_t)stream>>>(e->d_state, e->L, d_slots, n, d_rewards);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}
