"""HBM traffic of the tower's conv launches from two rocprofv3 --pmc passes over tools/tower_only.py (FETCH_SIZE and WRITE_SIZE
cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") -> one JSON for profiles/ that NAMES what it was measured on:

  conv_traffic.py <fetch dir> <write dir> <tower_only.py log> <out.json> [note]

The JSON carries the sha-256 of csrc/conv_split.hip and common.h as compiled into the library that ran (snk_source_hash), the
conv algorithm and the number of sub-rectangle layers (read from the tower log), and the kernel symbols the counters were summed
over.  bench.py quotes `roofline.traffic` from such a file only when the library it loaded reports the same hashes and form.
FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes, MI355X_MICROARCH.md
"HBM"); WRITE_SIZE is taken as reported (exact for 16-byte-per-lane streaming stores)."""
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]


def sums(d, counter):
    """{kernel symbol: (dispatches, counter total)} over the conv kernels of one pass"""
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if r.get("Counter_Name") == counter and "k_conv3x3" in k:
                k = re.sub(r"\(.*$", "", k.replace("void ", "")).strip()
                n, t = out.get(k, (0, 0.0))
                out[k] = (n + 1, t + float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, log, out_path = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    m = re.search(r"observations (\d+) forwards (\d+) n_rect (\d+) layers (\d+)(?: algo (\w+))?(?: board (\d+))?", open(log).read())
    assert m, "tower_only.py line not found in " + log
    n_obs, fwd, n_rect, layers = (int(m.group(i)) for i in range(1, 5))
    algo = m.group(5) or "f16s"
    board = int(m.group(6) or 11)
    side = 2 * board - 1
    elem = 2 if algo in ("bf16", "f16a") else 4           # bytes of an activation element in HBM
    # the FULL convolution's algorithmic bytes per state and layer: read x + write the output + the shortcut on every second layer
    # - the last layer's output, which never goes to HBM (its epilogue feeds the head)
    algorithmic = side * side * 128 * elem * (2.5 - 1.0 / layers)
    from snake_engine._lib import lib
    L = lib()
    fe, wr = sums(fetch_dir, "FETCH_SIZE"), sums(write_dir, "WRITE_SIZE")
    assert fe and wr and set(fe) == set(wr), (sorted(fe), sorted(wr))
    launches = sum(n for n, _ in fe.values())
    assert launches % (fwd * layers) == 0 or launches >= fwd * layers, (launches, fwd, layers)
    # counters are reported in KB (rocprofv3 derived metric): FETCH_SIZE x 2 (gfx950), WRITE_SIZE as is
    rd = 2.0 * 1024.0 * sum(t for _, t in fe.values())
    wb = 1024.0 * sum(t for _, t in wr.values())
    per = (rd + wb) / (fwd * n_obs * layers)
    d = {
        "conv_algo": algo, "rect_layers": n_rect, "tower_layers": layers, "board": board,
        "source_sha256": {f: (L.snk_source_hash(f.encode()) or b"").decode() for f in ("conv_split.hip", "common.h")},
        "kernel_symbols": sorted(fe),
        "workload": f"tools/tower_only.py: {fwd} whole forwards of {n_obs} mid-game {board}x{board} observations (one chunk), {layers} tower layers",
        "states_per_forward": n_obs, "forwards": fwd, "conv_launches_counted": launches,
        "FETCH_SIZE_KB_total": {k: t for k, (_, t) in sorted(fe.items())},
        "WRITE_SIZE_KB_total": {k: t for k, (_, t) in sorted(wr.items())},
        "hbm_read_bytes_per_forward_corrected": rd / fwd, "hbm_write_bytes_per_forward": wb / fwd,
        "hbm_bytes_per_state_layer": per,
        # (11x11 float32 tower: 225 792 + 225 792 + 112 896 - 28 224 = 536 256 B)
        "algorithmic_bytes_per_state_layer_full_form": algorithmic,
        "vs_full_form_algorithmic": per / algorithmic,
        "note": ("separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' "
                 "(gfx950 reports half the bytes of 16-B-per-lane streaming reads); WRITE_SIZE exact.  " + note).strip(),
    }
    json.dump(d, open(out_path, "w"), indent=1)
    print(json.dumps({k: d[k] for k in ("conv_algo", "rect_layers", "hbm_bytes_per_state_layer", "vs_full_form_algorithmic", "kernel_symbols")}))


if __name__ == "__main__":
    main()
