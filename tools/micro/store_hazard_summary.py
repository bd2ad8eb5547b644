"""Turns the output of tools/micro/store_hazard (one plain run, optionally one run beside the stress kernel) into the profile that
tools/store_hazard_scan.py's rule and tests/test_store_hazard_cpu.py cite:

    store_hazard_summary.py plain.json [stress.json] > profiles/r6_store_hazard_micro.json
"""
import json
import sys

GAP_WS = {"d0": 0, "v1": 1, "v2": 2, "v3": 3, "v4": 4, "n0": 1, "n1": 2, "n2": 3}
FORMS = {"bsg4": "buffer_store_dwordx4, SGPR soffset", "bim4": "buffer_store_dwordx4, soffset 0", "bsg3": "buffer_store_dwordx3, SGPR soffset",
         "bsg2": "buffer_store_dwordx2, SGPR soffset", "gsa4": "global_store_dwordx4, saddr", "gva4": "global_store_dwordx4, vaddr",
         "ldw4": "ds_write_b128", "ldw3": "ds_write_b96", "ldw2": "ds_write_b64"}


def table_of(d):
    table = {}
    for v in d["variants"]:
        t = table.setdefault(v["store"], {}).setdefault(v["gap"], {"wait_states": GAP_WS[v["gap"]], "dwords_stored": 0, "dwords_new": 0,
                                                                    "dwords_other": 0, "lanes_mod16": 0})
        width = int(v["store"][-1])
        t["dwords_stored"] += v["stored_per_dword"] * width
        t["dwords_new"] += sum(v["new"])
        t["dwords_other"] += sum(v["other"])
        t["lanes_mod16"] |= int(v["lanes_mod16_with_new"], 16)
    for s in table.values():
        for g in s.values():
            g["lanes_mod16"] = "0x%04x" % g["lanes_mod16"]
    return table


def main():
    runs = [json.load(open(p)) for p in sys.argv[1:3]]
    tables = [table_of(d) for d in runs]
    rule = {}
    for s in FORMS:
        if not all(s in t for t in tables):
            continue
        worst = max([g["wait_states"] for t in tables for g in t[s].values() if g["dwords_new"]] + [-1])
        rule[s] = {"form": FORMS[s], "largest_wait_states_with_wrong_data": worst if worst >= 0 else None, "wait_states_needed": worst + 1}
    out = {"what": "tools/micro/store_hazard.hip on an MI355X (gfx950), round 6: <data registers := OLD>; STORE; GAP; VALU write of data "
                   "register(s) := NEW; a checker classes every stored dword.  dwords_new = the store carried the NEW value (the hazard).  "
                   "gap d0 = nothing, vN = N independent VALU instructions, nK = s_nop K (K + 1 wait states).  Six overwrite forms (v_mov_b32 of "
                   "dword 0..3, v_pk_mul_f32 of dwords 0:1 / 2:3) x two grids (8 and 2 048 workgroups) x the repetitions, summed per cell.  "
                   "'stress' = the same sequences beside a kernel that keeps one workgroup per compute unit busy with back-to-back MFMAs, LDS "
                   "traffic, global loads and 16-byte global stores on another stream.",
           "device": runs[0]["device"], "rule": rule, "by_store_and_gap": tables[0],
           "range_check": runs[0]["range_check"],
           "range_check_note": "a raw buffer load / store whose SGPR soffset alone carries the address past num_records: all 64 lanes read "
                               "zero, no lane stored -- the range check of gfx950 covers soffset (LLVM documents it as excluded; ADVICE round 5 "
                               "finding 1 asked)",
           "variants_with_wrong_data": [v for v in runs[0]["variants"] if sum(v["new"]) + sum(v["other"])]}
    if len(runs) > 1:
        out["by_store_and_gap_stress"] = tables[1]
        out["variants_with_wrong_data_stress"] = [v for v in runs[1]["variants"] if sum(v["new"]) + sum(v["other"])]
        out["range_check_stress"] = runs[1]["range_check"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
