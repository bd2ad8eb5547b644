#!/usr/bin/env python3
"""bench.py -- self-play env-steps/sec of the MI355X engine (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] -- 11x11 board, 4 snakes, 4 096 parallel games per GPU,
max_MCTS_breadth 50 (= 48 rollouts, agent.py:32-37), max_MCTS_depth 8, health_dec 1, softmax_base 2,
training=True (trainer.py:52), Glorot-initialised gen-0 net (seed 0), synthetic start boards.
One "step" = one root turn of MPGameRunner.run over all live games (mp_game_runner.py:31-68): the full
MCTS behind every root Game.tic.  value = root env-steps of all ranks / max-over-ranks wall time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--games G | --games-total T] [--breadth B] [--no-cpu-baseline]
N > 1: one rank per GPU, either under torch.distributed.run (RANK / WORLD_SIZE in the environment) or started by
this script itself (`python bench.py --gpus N` spawns N fresh rank processes before touching the GPU and relays
rank 0's JSON line).  Games shard across ranks (weak scaling: G games per GPU), no collective in the self-play
path; the timed region ends with the iteration-end exchange over RCCL (all-gather of this rank's share of 10 240
sampled rows + all-reduce of the log counters).  With fewer GPUs than ranks the ranks share GPUs and use gloo.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "alphasnake-zero_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cpus():
    """host cpus this process may really use: the scheduler affinity capped by the cgroup cpu quota (the GPU boxes show
    256 cpus but grant 16: 128 PyTorch threads on a 16-cpu quota is why round 1's baseline was not reproducible)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(weights, breadth, n_games=16, n_turns=2):
    """The same workload on this box's host cores, as a FIXED amount of work: oracle/mcts_cpu.c (the C restatement of
    MPGameRunner.run + Agent/MCTSAgent.make_moves over oracle/snake_oracle.c, pinned to the reference's recorded runs by
    tests/test_mcts_cpu_baseline.py) with the PyTorch-CPU fp32 net (oracle/net_ref.py, identical weights), n_games
    seeded start boards x n_turns root turns, one net batch per rollout tick (n_games x 8 sub-games x <= 4 snakes rows).
    Also timed: the engine + MCTS alone (stub net inside the C code) on 1 and on all usable threads."""
    from oracle import net_ref
    from oracle.mcts_cpu import CpuSelfPlay, seeded_games
    cpus = usable_cpus()
    torch.set_num_threads(cpus)
    net_ref.forward(weights, np.zeros((64, 21, 21, 3), np.float32))        # thread pool / oneDNN primitives warm
    sp = CpuSelfPlay(seeded_games(n_games, seed=1), net=lambda X: net_ref.forward(weights, X), threads=min(cpus, n_games),
                     base=2, training=True, max_depth=8, max_breadth=breadth, seed=1)
    t0 = time.time()
    st = sp.run(max_turns=n_turns)
    dt = time.time() - t0
    sp.close()
    eng = {}
    for thr in sorted({1, cpus}):
        e = CpuSelfPlay(seeded_games(64 * thr, seed=2), net=None, threads=thr, base=2, training=True, max_depth=8,
                        max_breadth=breadth, seed=2)
        t1 = time.time()
        es = e.run(max_turns=2 if thr == 1 else 6)       # about a second of work either way
        eng[thr] = es["env_steps"] / (time.time() - t1)
        e.close()
    # bridge to the reference itself: the Python restatement (oracle/mcts_oracle.py: the
    # reference's loop statement for statement over the C game primitives) on the workload BASELINE.md timed the unmodified
    # reference on -- 8 games, breadth 50, stub net, one process.  In the survey container the reference does 6.46 and this
    # restatement 29.4 env-steps/s (ratio 0.22); the same ratio applied to this box's figure estimates the reference here.
    from oracle.mcts_oracle import SelfPlayOracle, Draws
    from oracle.obs_key import StubNet
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spo = SelfPlayOracle(StubNet(), 2, True, 8, 50, Draws(seed=1))
        t2 = time.time()
        _, py_steps = spo.run(seeded_games(8, seed=1), max_turns=6, rng=np.random.RandomState(0))
        py_rate = py_steps / (time.time() - t2)
    return {"value": st["env_steps"] / dt, "unit": "env-steps/s", "cores": cpus, "kind": "port",
            "sample": f"fixed work: {n_games} seeded games x {n_turns} root turns, breadth {breadth} = {st['env_steps']} env-steps, "
                      f"{st['net_evals']} net evals, {st['sim_steps']} rollout tics in {dt:.1f} s; oracle/mcts_cpu.c on "
                      f"{min(cpus, n_games)} threads + PyTorch-CPU fp32 net on {cpus} threads "
                      f"(os.cpu_count() = {os.cpu_count()}, usable = {cpus})",
            "net_evals_per_s": st["net_evals"] / dt,
            "python_restatement_stub_net_1_process": py_rate,
            "reference_estimate_stub_net_1_process": 6.46 / 29.4 * py_rate,
            "reference_estimate_note": "the unmodified reference cannot run on this box; BASELINE.md measured it at 6.46 env-steps/s "
                                       "(8 games, breadth 50, stub net, 1 core) where this Python restatement does 29.4",
            "engine_only_env_steps_per_s": {f"{k}_threads": v for k, v in eng.items()},
            "engine_only_note": "stub net inside the C code (zero net cost), 64 games per thread x 2 (1 thread) / 6 (all threads) root turns: what the "
                                "reference's Python loop does at 6.5 env-steps/s per core with a stub net (BASELINE.md)"}


def engine_kernel_rooflines(se, n=32768):
    """HBM-bound engine kernels in isolation: algorithmic bytes / measured time vs 8 TB/s"""
    out = {}
    eng = se.Engine(n, 11, 11, 4, 1, 0.15, seed=1234)
    eng.reset()
    g = torch.Generator(device="cuda").manual_seed(1234)
    sub = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(4)
    all_pairs = torch.stack([sub, torch.arange(4, dtype=torch.int32, device="cuda").repeat(n)], dim=1).contiguous()
    blocked = torch.empty((4 * n, 3), dtype=torch.uint8, device="cuda")

    def legal_moves():
        """uniform over the moves the obstacle mask leaves open ('uniformly random unmasked-legal moves'), so that
        the boards the kernels are timed on are mid-game boards, not finished games"""
        eng.observe(all_pairs, 4 * n, None, blocked, None)
        r = torch.rand((4 * n, 3), device="cuda", generator=g) - 2.0 * blocked.float()
        mv_ = torch.where(blocked.bool().all(dim=1), torch.ones((), dtype=torch.int64, device="cuda"), r.argmax(dim=1))
        return mv_.to(torch.uint8).reshape(n, 4).contiguous()
    for _ in range(32):
        eng.step(legal_moves())
    snap = se.Engine(n, 11, 11, 4, 1, 0.15)
    eng.clone_to(snap)
    G = eng.slot_bytes
    live_games = int((eng.alive().sum(dim=1) > 1).sum().item())       # k_step reads every record, writes back the unfinished ones

    def timed(fn, iters):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters
    mv = legal_moves()
    ts = []
    for _ in range(20):
        snap.clone_to(eng)
        torch.cuda.synchronize()
        ts.append(timed(lambda: eng.step(mv), 1))
    t = float(np.median(ts))
    step_bytes = (n + live_games) * G
    out["step"] = {"bound": "hbm", "achieved": step_bytes / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": step_bytes / t / 8e12,
                   "bytes_per_unit": 2 * G, "units": n, "us": t * 1e6, "unfinished_games": live_games,
                   "bytes_counted": "G read per game + G written per unfinished game after 32 warm-up ticks of uniform legal moves"}
    eng.set_params(food_spawn_chance=0.0)         # the form the rollout loop launches (sub-games never spawn food, game.py:268)
    ts = []
    for _ in range(20):
        snap.clone_to(eng)
        torch.cuda.synchronize()
        ts.append(timed(lambda: eng.step(mv), 1))
    t = float(np.median(ts))
    eng.set_params(food_spawn_chance=0.15)
    out["step_rollout_form"] = {"bound": "hbm", "achieved": step_bytes / t / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": step_bytes / t / 8e12, "bytes_per_unit": 2 * G, "units": n, "us": t * 1e6,
                                "form": "food_spawn_chance 0: no Philox draw, no empty-cell mask"}
    t = timed(lambda: snap.clone_to(eng), 20)
    out["clone"] = {"bound": "hbm", "achieved": n * 2 * G / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": n * 2 * G / t / 8e12,
                    "bytes_per_unit": 2 * G, "units": n, "us": t * 1e6}
    snap.clone_to(eng)
    alive = eng.alive()
    pairs = torch.nonzero(alive).to(torch.int32).contiguous()
    m = pairs.shape[0]
    planes = torch.empty((m, 21, 21, 3), device="cuda")
    mask = torch.empty((m, 3), dtype=torch.uint8, device="cuda")
    key = torch.empty((m, 2), dtype=torch.int64, device="cuda")
    # the two forms the MCTS loop launches (snake_engine/mcts.py): mask + key for every rollout state, planes for the misses
    t = timed(lambda: eng.observe(pairs, m, planes, None, None), 10)
    byts = m * (G + 5292)
    out["observe"] = {"bound": "hbm", "achieved": byts / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": byts / t / 8e12,
                      "bytes_per_unit": G + 5292, "units": m, "us": t * 1e6, "form": "planes only"}
    t = timed(lambda: eng.observe(pairs, m, None, mask, key), 10)
    byts = m * (G + 19)
    out["observe_mask_key"] = {"bound": "hbm", "achieved": byts / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": byts / t / 8e12,
                               "bytes_per_unit": G + 19, "units": m, "us": t * 1e6,
                               "form": "mask + key (instruction-bound: two splitmix64 per board cell)"}
    if n < 262144:      # 32 768 games move 39 MB: launch latency shows; the 8-GPU total of BASELINE configs[3] shows the kernel
        del eng, snap, planes, mask, key
        big = engine_kernel_rooflines(se, 262144)
        out["step_262144_games"] = big["step"]
        out["step_rollout_form_262144_games"] = big["step_rollout_form"]
        out["clone_262144_games"] = big["clone"]
        out["observe_19x19_8_snakes"] = observe_roofline_19(se)
    return out


def observe_roofline_19(se, n=16384):
    """k_observe on BASELINE configs[4]'s boards (19x19, 8 snakes: 16-bit ring entries, 8 352-byte records, 16 428-byte observations),
    planes form, mid-game boards -- the one engine kernel whose share of a configs[4] run is not negligible (5-8 % of the GPU time)"""
    S, B = 8, 19
    eng = se.Engine(n, B, B, S, 1, 0.15, seed=1234)
    eng.reset()
    g = torch.Generator(device="cuda").manual_seed(1234)
    sub = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(S)
    all_pairs = torch.stack([sub, torch.arange(S, dtype=torch.int32, device="cuda").repeat(n)], dim=1).contiguous()
    blocked = torch.empty((S * n, 3), dtype=torch.uint8, device="cuda")
    for _ in range(32):
        eng.observe(all_pairs, S * n, None, blocked, None)
        r = torch.rand((S * n, 3), device="cuda", generator=g) - 2.0 * blocked.float()
        mv = torch.where(blocked.bool().all(dim=1), torch.ones((), dtype=torch.int64, device="cuda"), r.argmax(dim=1))
        eng.step(mv.to(torch.uint8).reshape(n, S).contiguous())
    pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
    m = pairs.shape[0]
    planes = torch.empty((m, 2 * B - 1, 2 * B - 1, 3), device="cuda")
    eng.observe(pairs, m, planes, None, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.observe(pairs, m, planes, None, None)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-4
    G = eng.slot_bytes
    byts = m * (G + (2 * B - 1) ** 2 * 12)
    return {"bound": "hbm", "achieved": byts / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": byts / t / 8e12,
            "bytes_per_unit": G + (2 * B - 1) ** 2 * 12, "units": m, "us": t * 1e6, "form": "planes only, 19x19 / 8 snakes, 16 384 games"}


def conv_traffic_profile(L, algo, n_rect, flops_per_launch, profiles_dir=None, board=11):
    """(HBM bytes per average conv launch, profile file) from the newest profiles/*conv*traffic.json that was measured on the
    kernel sources the loaded library was built from and on the same form; (None, reason) otherwise"""
    import glob
    have = {f: (L.snk_source_hash(f.encode()) or b"").decode() for f in ("conv_split.hip", "common.h")}
    cands = []
    for path in glob.glob(os.path.join(profiles_dir or os.path.join(REPO, "profiles"), "*traffic.json")):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("source_sha256") == have and d.get("conv_algo") == algo and d.get("rect_layers") == n_rect \
                and d.get("board", 11) == board and "hbm_bytes_per_state_layer" in d:
            cands.append((os.path.getmtime(path), path, d))
    if not cands:
        return None, "no profile under profiles/ was measured on this library's csrc/conv_split.hip (snk_source_hash) in this form"
    _, path, d = max(cands)
    side = 2 * board - 1
    return d["hbm_bytes_per_state_layer"] * flops_per_launch / (2.0 * side * side * 9 * 128 * 128), os.path.relpath(path, REPO)


def conv_counters_profile(L, algo, board=11, profiles_dir=None):
    """(SQ-counter summary of the conv launches, profile file) from the newest profiles/*sq_counters.json taken on the kernel
    sources the loaded library was built from (tools/pmc_tower.sh -> tools/pmc_collect.py name every csrc hash); (None, reason)
    otherwise -- counters of another build say nothing about this one"""
    import glob
    have = {f: (L.snk_source_hash(f.encode()) or b"").decode() for f in ("conv_split.hip", "common.h")}
    want = "f16s<" if algo == "f16s" else "false, 3"            # the split form's symbols / the 16-bit frame's (SPLIT = false, IO16 = 3)
    cands = []
    for path in glob.glob(os.path.join(profiles_dir or os.path.join(REPO, "profiles"), "*sq_counters.json")):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        src = d.get("source_sha256", {})
        note = d.get("note", "")
        args = note.split("tower_only.py", 1)[1].split()[:3] if "tower_only.py" in note else []      # games, forwards[, board]
        note_board = int(args[2]) if len(args) > 2 and args[2].isdigit() else 11
        if all(src.get(f) == h for f, h in have.items()) and f"SNK_CONV_ALGO={algo}" in note and note_board == board:
            cands.append((os.path.getmtime(path), path, d))
    if not cands:
        return None, "no SQ-counter profile under profiles/ was taken on this library's csrc/conv_split.hip (snk_source_hash) for this tower"
    _, path, d = max(cands)
    out = {}
    for k, v in d.get("kernels", {}).items():
        p_, dv = v.get("per_dispatch", {}), v.get("derived", {})
        if p_.get("SQ_INSTS_MFMA", 0) < 1e5 or (want not in k and algo != "f16s"):
            continue
        row = {"dispatches": v.get("dispatches")}
        if "mfma_busy_per_busy_cycle" in dv:
            # SQ_VALU_MFMA_BUSY_CYCLES sums over 1 024 SIMDs, SQ_BUSY_CYCLES over 32 shader engines: 32 = every SIMD's pipe busy
            row["mfma_busy_share"] = dv["mfma_busy_per_busy_cycle"] / 32.0
        for a, b in (("lds_conflict_share", "lds_conflict_share"), ("wait_inst_any_share_of_wave_cycles", "sq_wait_inst_any_share_of_wave_cycles"),
                     ("wait_any_share_of_wave_cycles", "sq_wait_any_share_of_wave_cycles"), ("valu_per_mfma", "valu_instructions_per_mfma")):
            if b in dv:
                row[a] = dv[b]
        out[k] = row
    return out, os.path.relpath(path, REPO)


def workload_label(board, snakes, blocks, games, breadth, world):
    """which BASELINE.json config a run is (shape AND size), or 'custom'"""
    if (board, snakes, blocks) == (11, 4, 4):
        if (games, breadth) == (4096, 50):
            return "configs[1]" if world == 1 else f"configs[1] per GPU x{world} (weak scaling)"
        if (games, breadth) == (8, 25) and world == 1:
            return "configs[0] (on the GPU)"
        if (games, breadth) == (32768, 200):
            return "configs[2]" if world == 1 else f"configs[2] per GPU x{world}"
        if games * world == 262144:
            return f"configs[3] ({games} games per GPU x{world}, breadth {breadth})"
        return "custom (configs[1] shape)"
    if (board, snakes, blocks) == (19, 8, 10):
        return "configs[4] shape" + (f" x{world}" if world > 1 else "")
    return "custom"


def launch_ranks(n):
    """Parent of an N-rank run: one fresh child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (what torch.distributed.run would do), started BEFORE this process makes any HIP call.  Rank 0's stdout (the one
    JSON line) is relayed; the exit code is the first non-zero child code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports dmabuf IPC only; with the legacy mode RCCL's peer
        # buffers fail in hipIpcGetMemHandle ("invalid argument").  main() sets the same default for ranks started by
        # torch.distributed.run; a value the caller exported is left alone.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained by a thread so that a crashed rank is noticed while the others sit in a collective
    import threading
    chunks = []
    t = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    t.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):       # the exact children started above, nothing else
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.2)
    t.join(timeout=10)
    sys.stdout.write("".join(chunks))
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    if bad:
        log(f"bench.py: rank exit codes {codes}")
    return bad[0] if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="parallel root games per GPU")
    ap.add_argument("--games-total", type=int, default=None,
                    help="root games of the whole job, split evenly over the GPUs (BASELINE configs[3]: --gpus 8 --games-total 262144 "
                         "--breadth 200); overrides --games; the run is then strong scaling")
    ap.add_argument("--breadth", type=int, default=50)
    ap.add_argument("--chunk", type=int, default=8192, help="states per net forward chunk")
    ap.add_argument("--conv-algo", choices=["winograd", "direct", "bf16", "f16s", "f16", "f16a"], default=None,
                    help="default: f16s (float32-accurate split-f16 MFMA, the judged configuration); winograd, direct: f32 MFMA; f16 / f16a / bf16: reduced precision for configs[4] (bf16 = bf16 activations in HBM + v_mfma_f32_32x32x16_bf16), outside the 1e-5 tolerance")
    ap.add_argument("--conv-rect", choices=["0", "1"], default=None,
                    help="0: every tower layer convolves the whole canvas (the A/B arm of the sub-rectangle form; env SNK_CONV_RECT)")
    ap.add_argument("--board", type=int, default=11, choices=[7, 11, 19], help="board side; 19 with --snakes 8 --blocks 10 = BASELINE configs[4]")
    ap.add_argument("--snakes", type=int, default=4)
    ap.add_argument("--blocks", type=int, default=4, help="residual blocks of the Q-net")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-timing", action="store_true",
                    help="no HIP events around the conv launches (roofline.achieved is then null): for small, launch-bound runs")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    ap.add_argument("--trace-markers", action="store_true",
                    help="launch the one-wavefront k_clock_probe kernel (1 us) at the start and at the end of the timed self-play steps: "
                         "tools/gpu_busy.py finds the timed region in a rocprofv3 kernel trace by these two launches")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the untimed extra root turn that samples the shader clock beside the forward chunks (roofline.clock_mhz)")
    args = ap.parse_args()

    if args.games_total is not None:
        if args.games_total % args.gpus:
            sys.exit(f"bench.py: --games-total {args.games_total} does not split evenly over {args.gpus} GPUs")
        args.games = args.games_total // args.gpus
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process stays off the GPU (nothing above touches it;
        # torch.cuda.device_count() does not initialise HIP on this image) and starts N fresh rank processes.
        sys.exit(launch_ranks(args.gpus))
    # stdout carries exactly one JSON line: anything a library prints there (gloo's "[Gloo] Rank 0 is connected ...",
    # RCCL banners) is sent to stderr by pointing fd 1 at fd 2 and keeping a private handle on the real stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.conv_algo:
        os.environ["SNK_CONV_ALGO"] = args.conv_algo
    if args.conv_rect is not None:
        os.environ["SNK_CONV_RECT"] = args.conv_rect
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # checked before any GPU call: a process that has touched the GPU must never be re-launched
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before the first HIP call of this rank (see launch_ranks)
    # the ranks of one node share its host cores: each keeps its share of the cpus this job may use (a rank's host side is one
    # Python thread sequencing launches; torch's default of one intra-op thread per visible cpu would put N x 256 threads on
    # a 16-cpu quota)
    host_threads = max(1, usable_cpus() // world)
    torch.set_num_threads(host_threads)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    n_dev = torch.cuda.device_count()
    # RCCL needs one GPU per rank; with fewer GPUs than ranks (a 1-GPU box) the ranks share GPUs and talk over gloo
    backend = os.environ.get("SNK_DIST_BACKEND") or ("nccl" if n_dev >= world else "gloo")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import snake_engine as se
    from snake_engine import net, dist as sdist
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner

    B, OBS = args.board, 2 * args.board - 1
    judged = (B, args.snakes, args.blocks) == (11, 4, 4)          # configs[1]'s shape: the one cpu_baseline / rooflines describe
    weights = net.glorot_uniform_weights((OBS, OBS, 3), blocks=args.blocks, seed=0)
    nnet = AlphaNNet(input_shape=(OBS, OBS, 3), _weights=weights)
    nnet._qnet.max_chunk = args.chunk
    MPGameRunner.verbose = False
    MPGameRunner.init = "device"
    alice = Agent(nnet, 2, True, 8, args.breadth, seed=1234 + rank)
    gr = MPGameRunner(B, B, args.snakes, 1, args.games, seed=1234 + rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()


    t_w = time.time()
    gr.run(alice, max_turns=args.warmup)
    barrier()
    log(f"[rank {rank}] warmup {args.warmup} turns in {time.time() - t_w:.1f} s")
    m = alice._mcts
    ev0, sim0 = (m.stats["net_evals"], m.stats["sim_steps"]) if m is not None else (0, 0)     # --warmup 0: nothing ran yet
    tick0 = m.stats["rollout_ticks"] if m is not None else 0
    nnet._qnet.conv_timing = None if args.no_conv_timing else []
    nnet._qnet.rect_tiles = None if args.no_conv_timing or not nnet._qnet.n_rect else []      # GEMM tiles the sub-rectangle layers execute
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    def sync_clock():
        torch.cuda.synchronize()
        return time.time()
    marker = None
    if args.trace_markers:
        marker = torch.zeros((2, 2), dtype=torch.int64, device="cuda")

    def mark(i):
        if marker is not None:
            se.check(se.lib().snk_clock_probe(marker[i].data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
    barrier()
    t0 = time.time()
    cpu0 = time.process_time()
    env_steps = 0
    mark(0)
    for k in range(args.steps):                # one root turn per call: the same launches, plus a progress line per step
        gr.run(alice, max_turns=1)
        env_steps += gr.env_steps
        if rank == 0:
            log(f"[rank 0] step {k + 1}/{args.steps}: {gr.env_steps} env-steps, {time.time() - t0:.1f} s since the start of the timed region")
    mark(1)
    t_play = sync_clock() - t0
    cpu_play = time.process_time() - cpu0      # user + system cpu seconds of this rank's process over the self-play steps
    # iteration-end exchange (trainer.py:63-75 across ranks): the row count comes from the records of ALL ranks, every rank
    # sends its share of the sampled rows (all-gather), the six log counters are all-reduced
    counts, seed = sdist.gather_counts(len(alice.records))
    wanted, _, share = sdist.sample_plan(max(sum(counts), world), world)
    wanted = min(wanted, sum(counts) // world * world)          # (a run too short to have recorded `world` rows samples what there is)
    rows = sdist.share_counts(counts, wanted, seed)             # equal shares; a rank short of its share is topped up by the others
    share = max(rows)
    idx = sdist.sample_share(len(alice.records), rows[rank], np.random.RandomState(rank))
    X = alice.records.fetch_device(idx) if len(idx) else torch.zeros((0, OBS, OBS, 3), dtype=torch.float32, device="cuda")
    Vs = torch.as_tensor(alice._values_host()[idx] if len(idx) else np.zeros((0, 3), np.float32), device="cuda")
    t1 = sync_clock()
    Xg, Vg = sdist.all_gather_samples(X, Vs, rows)
    t2 = sync_clock()
    sdist.all_reduce_counters(gr.engine.sum_counters(), args.games, "cuda")
    t3 = sync_clock()
    wall_rank = t3 - t0                        # this rank's own time, before it waits for the others
    barrier()
    dt = time.time() - t0
    m = alice._mcts
    evals = m.stats["net_evals"] - ev0
    sims = m.stats["sim_steps"] - sim0
    rollout_ticks = m.stats["rollout_ticks"] - tick0
    n_records = len(alice.records)
    tm = nnet._qnet.conv_timing or []
    nnet._qnet.conv_timing = None
    conv_s = sum(a.elapsed_time(b) for a, b, _ in tm) * 1e-3
    rect_tiles, nnet._qnet.rect_tiles = nnet._qnet.rect_tiles, None
    # the clock the chip holds under this loop, sampled OUTSIDE the timed region (one more root turn of the same games with a
    # one-wavefront probe kernel on its own stream beside every forward chunk): nothing extra runs while the metric is timed
    clk = np.zeros(0)
    if not args.no_clock_probe and len(gr.games):
        probe = net.ClockProbe(torch.device("cuda", dev_index))
        nnet._qnet.clock_probe = probe
        gr.run(alice, max_turns=1)
        torch.cuda.synchronize()
        nnet._qnet.clock_probe = None
        clk = probe.mhz()
    clk_med = float(np.median(clk)) if len(clk) else 0.0
    # per-rank figures for reading a scaling curve: every rank's env-steps, self-play time, sampling, the two collectives, the
    # host cpu seconds its process used, the GPU seconds of its conv launches, the clock its GPU held
    mine = torch.tensor([float(env_steps), t_play, t1 - t0 - t_play, t2 - t1, t3 - t2, wall_rank, float(evals), float(n_records),
                         cpu_play, conv_s, clk_med, float(host_threads)], dtype=torch.float64, device=coll_dev)
    per_rank = mine.unsqueeze(0)
    if world > 1:
        flat_rows = torch.empty(world * mine.numel(), dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(flat_rows, mine)
        per_rank = flat_rows.view(world, mine.numel())
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    per_rank = per_rank.cpu().numpy()
    env_steps_rank0 = env_steps
    env_steps = float(per_rank[:, 0].sum())

    if rank == 0:
        conv_flops = sum(f for _, _, f in tm)
        achieved = conv_flops / conv_s / 1e12 if conv_s > 0 else None      # None: --no-conv-timing
        algo = nnet._qnet.conv_algo
        peak = 2500.0 if algo in ("bf16", "f16s", "f16", "f16a") else 157.3      # dense MFMA peaks (bf16 / f16, f32), MI355X_MICROARCH.md
        # MFMA flops the kernel executes per algorithmic (direct-convolution) flop: MFMAs per product x GEMM rows per pixel
        T_full = (OBS * OBS + 31) // 32
        per_product = 3.0 if algo == "f16s" else 1.0           # hi*hi + hi*lo + lo*hi; the reduced-precision forms issue one
        executed = {"f16s": per_product * T_full * 32 / (OBS * OBS), "f16": T_full * 32 / (OBS * OBS), "f16a": T_full * 32 / (OBS * OBS),
                    "bf16": T_full * 32 / (OBS * OBS), "winograd": 16 * 121 / (441 * 9.0)}.get(algo, 1.0)
        qn = nnet._qnet
        rect = None
        qn.rect_tiles = rect_tiles
        if qn.rect_tiles:
            # sub-rectangle form: the first n_rect layers convolve the board window grown by one pixel per layer only; what
            # the launches executed = the 32-row GEMM tiles the device-side plans counted + the full layers' tiles
            n_layers = 2 * qn.blocks
            imgs = sum(mm for mm, _ in qn.rect_tiles)
            per_layer = torch.stack([c[:, 1] for _, c in qn.rect_tiles]).to(torch.float64).sum(dim=0).cpu().numpy()
            tiles = float(per_layer.sum()) + float(imgs) * T_full * (n_layers - qn.n_rect)
            executed = per_product * tiles * 32 / (float(imgs) * n_layers * OBS * OBS)
            rect = {"layers": qn.n_rect, "of": n_layers,
                    "tiles_executed_vs_full_per_layer": [float(v) / (imgs * T_full) for v in per_layer],
                    "tiles_executed_vs_full": tiles / (float(imgs) * T_full * n_layers),
                    "what": "tower layer i < layers convolves only the bounding box of an observation's non-background pixels (the "
                            "board window, game.py:215-257) grown by i + 2 pixels; the rest of its output is a per-layer constant "
                            "(bit-identical to the full convolution, tests/test_rect_conv_gpu.py); `achieved` still counts the "
                            "reference's full direct convolution"}
            qn.rect_tiles = None
        # HBM bytes per average launch, from the committed rocprofv3 --pmc passes (profiles/, made by tools/conv_traffic.py).  A
        # profile is quoted only for the kernel it was measured on: it names the sha-256 of csrc/conv_split.hip + common.h and the
        # form (algorithm, sub-rectangle layers), and the LOADED library reports the hashes of the sources it was built from
        traffic, traffic_src = None, None
        if tm:
            traffic, traffic_src = conv_traffic_profile(se.lib(), algo, qn.n_rect, conv_flops / len(tm), board=B)
        counters, counters_src = conv_counters_profile(se.lib(), algo, board=B)
        res = {
            "metric": "self-play env-steps/sec (11x11, 4 snakes, 50 MCTS sims)",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.games_total is not None else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16 operands and bf16 activations in HBM, f32 accumulate (outside the 1e-5 parity tolerance)",
                      "f16": "f16 operands, f32 accumulate (outside the 1e-5 parity tolerance)",
                      "f16a": "f16 operands and f16 activations in HBM, f32 accumulate (outside the 1e-5 parity tolerance)",
                      "f16s": "f32 (tower convolutions: each f32 operand split into f16 hi + lo, 3 f16 MFMAs per product, f32 accumulate)"}.get(algo, "f32"),
            "data": "synthetic",
            "config": {"workload": f"{workload_label(B, args.snakes, args.blocks, args.games, args.breadth, world)}: "
                                   f"{B}x{B}, {args.snakes} snakes, {args.blocks}-block net, {args.games} parallel games per GPU, max_MCTS_breadth "
                                   f"{args.breadth} (= {args.breadth // 8 * 8} rollouts), depth 8, health_dec 1, softmax_base 2, "
                                   "training=True, gen-0 Glorot net (seed 0), fp32 Q-net",
                       "games_per_gpu": args.games, "breadth": args.breadth, "parallelism": f"games sharded x{world}", "dist_backend": backend if world > 1 else None,
                       "games_total": args.games * world, "host_threads_per_rank": host_threads,
                       "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
                       "net_evals_per_env_step": evals / max(1, env_steps_rank0), "sim_steps_per_env_step": sims / max(1, env_steps_rank0),
                       "net_evals_per_s_rank0": evals / dt, "rollout_ticks_per_step": rollout_ticks / max(1, args.steps),
                       "net_evals_per_rollout_tick": evals / max(1, rollout_ticks), "sample_rows_gathered": int(Xg.shape[0])},
            "roofline": {"bound": "mfma",
                         "kernel": {"winograd": "k_conv3x3_wino_f32", "bf16": "k_conv3x3_f16s<SPLIT = false, IO16, BF = true>" + (" + its sub-rectangle form" if rect else ""),
                                    "f16s": "k_conv3x3_f16s" + (" + k_conv3x3_f16s_rect (the same body on sub-rectangles)" if rect else ""),
                                    "f16": "k_conv3x3_f16s<SPLIT = false>",
                                    "f16a": "k_conv3x3_f16s<SPLIT = false, IO16>"}.get(algo, "k_conv3x3_f32"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "counters": counters, "counters_source": counters_src,
                         "launches": len(tm),
                         "flops_convention": "algorithmic = direct 3x3 convolution, 2*441*1152*128 per state and layer (SURVEY 8d)",
                         "executed_frac": achieved / peak * executed if achieved else None,
                         "algorithm": {"winograd": "Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: executes 16*121/(441*9) = 0.488 of the "
                                                   "algorithmic flops, fp32 throughout",
                                       "bf16": "implicit GEMM on v_mfma_f32_32x32x16_bf16 (the split kernel's block body), bf16 operands, bf16 activations in HBM (reduced precision)",
                                       "f16": "implicit GEMM on v_mfma_f32_32x32x16_f16, f16-rounded operands (reduced precision)",
                                       "f16a": "implicit GEMM on v_mfma_f32_32x32x16_f16, f16 operands, f16 activations in HBM (reduced precision)",
                                       "f16s": "implicit GEMM on v_mfma_f32_32x32x16_f16 with split operands: 3 MFMAs (hi*hi, hi*lo, lo*hi) per "
                                               "product over the GEMM tiles actually launched (executed_frac; the full form pads 441 rows to "
                                               "448, the sub-rectangle form computes fewer rows, see sub_rectangles); peak = dense f16 MFMA "
                                               "at 2.4 GHz, the chip holds less in this loop (clock_mhz)"}.get(
                                                   algo, "implicit GEMM on v_mfma_f32_32x32x2_f32"),
                         "avg_launch_ms": conv_s / max(1, len(tm)) * 1e3,
                         "share_of_step_time": conv_s / dt, "sub_rectangles": rect},
            # what a non-linear scaling curve is made of: one row per rank (rank r = GPU r of the node)
            "ranks": [{"rank": r, "env_steps": int(v[0]), "self_play_s": v[1], "sample_rows_s": v[2], "all_gather_s": v[3],
                       "all_reduce_s": v[4], "wall_s": v[5], "env_steps_per_s": v[0] / v[5], "net_evals": int(v[6]),
                       "records": int(v[7]), "host_cpu_s": v[8], "conv_gpu_s": v[9] if tm else None,
                       "self_play_minus_conv_s": v[1] - v[9] if tm else None,
                       "clock_mhz_median": v[10] or None, "host_threads": int(v[11])} for r, v in enumerate(per_rank)],
            "ranks_note": "host_cpu_s = user + system cpu seconds of the rank's process over the self-play steps (one Python thread "
                          "sequencing launches; host_threads = torch intra-op threads = usable cpus // ranks); conv_gpu_s = HIP-event "
                          "time of its conv launches; self_play_minus_conv_s = every other kernel + the time its GPU waited for the host",
            "exchange": {"dist_backend": backend if world > 1 else None, "rows_per_rank": int(share), "rows_gathered": int(Xg.shape[0]),
                         "bytes_per_rank": int(share) * (OBS * OBS * 3 + 3) * 4,
                         "all_gather_s_max": float(per_rank[:, 3].max()), "all_reduce_s_max": float(per_rank[:, 4].max())},
        }
        if len(clk):
            held = float(np.median(clk))
            res["roofline"]["clock_mhz"] = {"median": held, "p10": float(np.percentile(clk, 10)), "p90": float(np.percentile(clk, 90)),
                                            "samples": int(len(clk)), "nominal": 2400.0,
                                            "how": "one-wavefront probe kernel on its own stream beside every forward chunk of ONE EXTRA, "
                                                   "UNTIMED root turn after the timed region (the same games, the next turn): "
                                                   "s_memtime / s_memrealtime x 100 MHz over 200 us (csrc/probe.hip)"}
            if res["roofline"]["executed_frac"]:
                res["roofline"]["executed_frac_of_held_clock_peak"] = res["roofline"]["executed_frac"] * 2400.0 / held
        if not args.no_kernel_rooflines and world == 1 and judged:
            del alice, gr
            torch.cuda.empty_cache()
            res["engine_kernels"] = engine_kernel_rooflines(se)
        if not args.no_cpu_baseline and world == 1 and judged:          # reported at N = 1 only
            res["cpu_baseline"] = cpu_baseline(weights, args.breadth)
        print(json.dumps(res), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
