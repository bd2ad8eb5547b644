"""Turns the per-pass output directories of a set of `rocprofv3 --pmc ...` runs into ONE JSON for profiles/:

  pmc_collect.py <dir with one sub-directory per pass> <out.json> <kernel substring>[,<substring>...] [note] [pass-directory prefix]

Per kernel symbol that contains one of the substrings: dispatches counted and, per counter, the total and the per-dispatch
average (counters of different passes see the same dispatches of the same tool run again).  The JSON names the sources the
counters were taken on: the sha-256 of every csrc file as compiled into the library found under SNK_LIB_PATH / the default path
(snk_source_hash), and the command line of each pass when its log holds one.  Derived figures where their counters are present:
MFMA busy share of the SIMD cycles, LDS conflict share, wait shares of the wave cycles, vector instructions per MFMA."""
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]


def clean(k):
    return re.sub(r"\(.*$", "", k.replace("void ", "")).strip()


def main():
    top, out_path, subs = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    prefix = sys.argv[5] if len(sys.argv) > 5 else ""
    per = {}
    passes = []
    for d in sorted(glob.glob(os.path.join(top, "*"))):
        if not os.path.isdir(d) or not os.path.basename(d).startswith(prefix):
            continue
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        seen = set()
        for f in files:
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                if not any(s in k for s in subs):
                    continue
                k = clean(k)
                c = r["Counter_Name"]
                seen.add(c)
                n, t = per.setdefault(k, {}).get(c, (0, 0.0))
                per[k][c] = (n + 1, t + float(r["Counter_Value"]))
        log = d + ".log"
        tail = ""
        if os.path.exists(log):
            lines = [ln.strip() for ln in open(log, errors="replace") if ln.strip()]
            tail = lines[-1] if lines else ""
        passes.append({"pass": os.path.basename(d), "counters": sorted(seen), "tool_output": tail[:300]})
    assert per, f"no counter rows for {subs} under {top}"
    hashes = {}
    try:
        from snake_engine._lib import lib
        L = lib()
        for f in sorted(os.listdir(os.path.join(REPO, "alphasnake-zero_amd", "csrc"))):
            if f.endswith((".hip", ".h")) and f != "build_id.h":
                h = L.snk_source_hash(f.encode())
                if h:
                    hashes[f] = h.decode()
    except Exception as e:      # noqa: BLE001 -- the summary is still worth keeping without the hashes
        hashes = {"error": repr(e)}
    kernels = {}
    for k, cs in sorted(per.items()):
        row = {"dispatches": max(n for n, _ in cs.values()),
               "per_dispatch": {c: t / n for c, (n, t) in sorted(cs.items())}}
        a = row["per_dispatch"]
        dv = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
            # SQ_BUSY_CYCLES counts per SE (32 of them tick together); MFMA busy is summed over the SIMDs of all CUs: the ratio
            # below is meaningful as a trend between builds of the same launch, the absolute share comes from the stamps
            dv["mfma_busy_per_busy_cycle"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CYCLES"], 1.0)
        if "SQ_LDS_BANK_CONFLICT" in a and "SQ_LDS_IDX_ACTIVE" in a:
            dv["lds_conflict_share"] = a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1.0)
        if "SQ_WAVE_CYCLES" in a:
            for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in a:
                    dv[c.lower() + "_share_of_wave_cycles"] = a[c] / max(a["SQ_WAVE_CYCLES"], 1.0)
        if "SQ_INSTS_MFMA" in a and "SQ_INSTS_VALU" in a:
            dv["valu_instructions_per_mfma"] = a["SQ_INSTS_VALU"] / max(a["SQ_INSTS_MFMA"], 1.0)
        if "SQ_INSTS_MFMA" in a and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            dv["mfma_busy_cycles_per_mfma"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_INSTS_MFMA"], 1.0)
        row["derived"] = dv
        kernels[k] = row
    d = {"source_sha256": hashes, "passes": passes, "kernels": kernels,
         "note": ("one rocprofv3 --pmc pass per counter group (at most four SQ counters per pass), --kernel-trace only beside it; "
                  "values are sums over all shader engines as rocprofv3 reports them, per dispatch.  " + note).strip()}
    json.dump(d, open(out_path, "w"), indent=1)
    for k, row in kernels.items():
        print(k, row["dispatches"], json.dumps(row["derived"]))


if __name__ == "__main__":
    main()
