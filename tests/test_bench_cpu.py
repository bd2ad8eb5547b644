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
