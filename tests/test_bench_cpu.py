"""host logic of bench.py that needs no GPU: which committed counter profile a run may quote for roofline.traffic"""
import json
import os
import sys

from conftest import REPO

sys.path.insert(0, REPO)


class _Lib:
    def __init__(self, hashes):
        self.h = hashes

    def snk_source_hash(self, name):
        v = self.h.get(name.decode())
        return None if v is None else v.encode()


def test_traffic_profile_is_quoted_only_for_the_sources_and_form_it_was_measured_on(tmp_path):
    import bench
    have = {"conv_split.hip": "a" * 64, "common.h": "b" * 64}
    prof = {"conv_algo": "f16s", "rect_layers": 6, "source_sha256": dict(have), "hbm_bytes_per_state_layer": 400000.0}
    json.dump(prof, open(tmp_path / "rX_conv_traffic.json", "w"))
    json.dump({"hbm_bytes_per_state_layer": 1.0}, open(tmp_path / "old_traffic.json", "w"))      # a profile without provenance
    per_launch = 2.0 * 441 * 9 * 128 * 128 * 1000                                               # a launch of 1 000 states
    t, src = bench.conv_traffic_profile(_Lib(have), "f16s", 6, per_launch, str(tmp_path))
    assert t == 400000.0 * 1000 and src.endswith("rX_conv_traffic.json")
    for lib_, algo, n_rect in ((_Lib(dict(have, **{"conv_split.hip": "c" * 64})), "f16s", 6),     # the kernel source changed
                               (_Lib(have), "f16a", 6), (_Lib(have), "f16s", 0), (_Lib({}), "f16s", 6)):
        t, why = bench.conv_traffic_profile(lib_, algo, n_rect, per_launch, str(tmp_path))
        assert t is None and "no profile" in why


def test_committed_traffic_profiles_with_provenance_are_well_formed():
    import glob
    for path in glob.glob(os.path.join(REPO, "profiles", "*traffic.json")):
        d = json.load(open(path))
        if "source_sha256" in d:
            assert set(d["source_sha256"]) == {"conv_split.hip", "common.h"} and all(len(v) == 64 for v in d["source_sha256"].values())
            assert d["conv_algo"] in ("f16s", "f16a", "bf16", "f16") and d["kernel_symbols"] and d["hbm_bytes_per_state_layer"] > 0


def test_counter_profile_is_quoted_only_for_the_sources_tower_and_board_it_was_taken_on(tmp_path):
    """round 6: roofline.counters / counters_source follow the rule of traffic / traffic_source -- SQ counters of
    another build of csrc/conv_split.hip, another tower or another canvas are not this run's counters"""
    import bench
    have = {"conv_split.hip": "a" * 64, "common.h": "b" * 64}
    kern = {"k_conv3x3_f16s<7, 1, true, 0, false>": {"dispatches": 4, "per_dispatch": {"SQ_INSTS_MFMA": 8.0e7},
                                                     "derived": {"mfma_busy_per_busy_cycle": 25.6, "lds_conflict_share": 0.46,
                                                                 "sq_wait_inst_any_share_of_wave_cycles": 0.58, "valu_instructions_per_mfma": 3.0}},
            "k_f16s_wscale": {"dispatches": 1, "per_dispatch": {"SQ_INSTS_MFMA": 0.0}, "derived": {}}}
    prof = {"source_sha256": dict(have, **{"engine.hip": "e" * 64}), "kernels": kern,
            "note": "tools/pmc_tower.sh r6_f16s f16s 2300 11: tools/tower_only.py 2300 2 11 with SNK_CONV_ALGO=f16s (two forwards ...)"}
    json.dump(prof, open(tmp_path / "rX_conv_f16s_sq_counters.json", "w"))
    c, src = bench.conv_counters_profile(_Lib(have), "f16s", 11, str(tmp_path))
    assert src.endswith("rX_conv_f16s_sq_counters.json") and list(c) == ["k_conv3x3_f16s<7, 1, true, 0, false>"]
    row = c["k_conv3x3_f16s<7, 1, true, 0, false>"]
    assert abs(row["mfma_busy_share"] - 0.8) < 1e-12 and row["lds_conflict_share"] == 0.46 and row["valu_per_mfma"] == 3.0
    for lib_, algo, board in ((_Lib(dict(have, **{"conv_split.hip": "c" * 64})), "f16s", 11), (_Lib(have), "bf16", 11),
                              (_Lib(have), "f16s", 19), (_Lib({}), "f16s", 11)):
        c, why = bench.conv_counters_profile(lib_, algo, board, str(tmp_path))
        assert c is None and "no SQ-counter profile" in why
    # round 5's file (pmc_a16.sh's note: "... tower_only.py 500 2 19 with SNK_CONV_ALGO=bf16 ...") is found for the bf16 tower at 19
    prof19 = dict(prof, note="tools/pmc_a16.sh r5_bf16: tools/tower_only.py 500 2 19 with SNK_CONV_ALGO=bf16 (two forwards ...)",
                  kernels={"k_conv3x3_f16s<8, 1, false, 3, true>": kern["k_conv3x3_f16s<7, 1, true, 0, false>"]})
    json.dump(prof19, open(tmp_path / "rY_a16_sq_counters.json", "w"))
    c, src = bench.conv_counters_profile(_Lib(have), "bf16", 19, str(tmp_path))
    assert src.endswith("rY_a16_sq_counters.json") and list(c) == ["k_conv3x3_f16s<8, 1, false, 3, true>"]
