"""Which committed profiles describe the kernels as they are NOW?

    python3 tools/profiles_index.py [prefix, e.g. r6_]

For every profiles/<prefix>*.json that names the sources it was measured on (`source_sha256`: tools/conv_traffic.py,
tools/pmc_collect.py write it from the library's snk_source_hash), prints the first 10 hex digits of its csrc/conv_split.hip hash and
whether that is the file in the tree ("current") or an earlier one ("older": bench.py will not quote it; roofline.traffic_source /
counters_source say so).  Needs no GPU and no built library: the tree's files are hashed directly.
"""
import glob
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_hash(name):
    return hashlib.sha256(open(os.path.join(REPO, "alphasnake-zero_amd", "csrc", name), "rb").read()).hexdigest()


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else ""
    now = {f: tree_hash(f) for f in ("conv_split.hip", "common.h", "engine.hip", "mcts.hip", "net.hip")}
    rows = []
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", prefix + "*.json"))):
        try:
            d = json.loads(open(path).read().strip().splitlines()[0]) if path.endswith(".json") else {}
        except (ValueError, IndexError):
            try:
                d = json.load(open(path))
            except ValueError:
                continue
        src = d.get("source_sha256") if isinstance(d, dict) else None
        if not isinstance(src, dict):
            rows.append((os.path.basename(path), "-", "(a bench line or log: names no sources)"))
            continue
        conv = ("conv_split.hip", "common.h")                   # what the conv profiles describe and bench.py compares
        other = [f for f, h in now.items() if f in src and src[f] != h and f not in conv]
        if all(src.get(f) == now[f] for f in conv if f in src):
            state = "current" + (f" (since then changed, not a conv source: {', '.join(other)})" if other else "")
        else:
            state = "older: " + ", ".join(f for f in conv if f in src and src[f] != now[f])
        rows.append((os.path.basename(path), (src.get("conv_split.hip") or "-")[:10], state))
    w = max(len(r[0]) for r in rows) if rows else 10
    for r in rows:
        print(f"{r[0]:{w}s}  {r[1]:10s}  {r[2]}")
    print(f"tree: conv_split.hip {now['conv_split.hip'][:10]}")


if __name__ == "__main__":
    main()
