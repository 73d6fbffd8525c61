"""bench.py's CPU-side helpers: the committed PMC profiles are read per workload, the slot-time summary adds up, the known answers and inputs that the
command-line legs of the driver's line compare with exist."""
import json
import os

import bench
from tests import golden_util as gu


def test_committed_profiles_belong_to_their_workloads():
    for cfg, name in ((None, "pmc_latest.json"), ("c3", "pmc_c3.json")):
        with open(os.path.join(bench.ROOT, "profiles", name)) as fh:
            raw = json.load(fh)
        assert raw["config"] == (cfg or "headline2k")
        prof, tag = bench.profiled_counters(cfg)
        if prof is None:                                         # the kernel sources moved on since the profile: the line then carries nulls, by design
            assert tag.startswith("stale") and raw["kernel_src_sha"] != bench.kernel_sources_sha()
            continue
        st = bench.slot_time(prof)
        assert abs(sum(prof["chain_per_launch"][k].get("wave_cycles", 0.0) for k in prof["chain_per_launch"]) - st["sum_wave_cycles"]) < 1.0
        assert 0 < st["service_kernels_wave_cycles"] < st["sum_wave_cycles"]
        assert abs(st["ms_on_4096_slots_at_2.4GHz"] - st["sum_wave_cycles"] / (4096 * 2.4e9) * 1e3) < 1e-9
    assert bench.profiled_counters("c2")[0] is None              # no profile of config 2: its line says null, not the headline's numbers


def test_known_answers_of_the_command_line_legs_exist():
    for f in ("c2_1000_stdout.json", "c4_100000_stdout.json", "c2_1000_wire.json", "c3_100_wire.json", "c4_100k_wire.json"):
        with open(os.path.join(gu.GOLDEN, f)) as fh:
            k = json.load(fh)
        assert len(k["sha256"]) == 64
    assert len(bench.C5_FILES) == 15
    for n in bench.C5_FILES:
        assert os.path.exists(gu.input_path(n)) and os.path.exists(os.path.join(gu.GOLDEN, f"{n}.p.stdout"))
