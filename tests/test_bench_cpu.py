"""bench.py without a GPU: the helpers that bring the committed profile summaries into the bench line
find them and read them (a renamed file or key would otherwise silently drop a field), and the split of
a step over ranks covers every sample exactly once in each mode."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_profile_summaries_reach_the_bench_line():
    bench = importlib.import_module("bench")
    for name in ("cfg2", "cfg3"):
        c = bench.measured_counters(name)
        assert c and 0.3 < c["valu_lane_utilisation"] < 1.0 and 8.0 < c["resident_waves_per_cu"] <= 32.0
        assert 0.0 < c["wave_cycles_waiting"] < 1.0 and c["source"].startswith("profiles/")
        per_path, src = bench.measured_traffic(name)
        assert 10.0 < per_path < 200.0 and "profiles/" in src
        p = bench.predicted_scaling(name)
        assert set(p["strong_spp_split_efficiency"]) == {"1", "2", "4", "8"}
        assert all(0.5 < v <= 1.05 for v in p["strong_spp_split_efficiency"].values())
    # the million-object scene has counters of the bench's own 8K launch
    per_path, src = bench.measured_traffic("cfg5_8k")
    assert 100.0 < per_path < 5000.0 and "7680x4320" in src


def test_workloads_are_the_configs_baseline_names():
    bench = importlib.import_module("bench")
    w = bench.WORKLOADS
    assert (w["cfg2"]["width"], w["cfg2"]["height"], w["cfg2"]["spp"], w["cfg2"]["depth"]) == (1920, 1080, 100, 10)
    assert (w["cfg3"]["spp"], w["cfg3"]["depth"]) == (500, 50)
    assert (w["cfg4"]["width"], w["cfg4"]["height"]) == (3840, 2160)
    assert (w["cfg5"]["width"], w["cfg5"]["height"], w["cfg5"]["scene"]) == (7680, 4320, "synthetic:1000000")


def test_sample_and_tile_shares_cover_a_step_once():
    from raytracer_amd import distributed as D
    for world in (1, 2, 3, 4, 7, 8):
        for spp in (1, 7, 13, 100, 500):
            seen = []
            for rank in range(world):
                begin, count = D.spp_slice(spp, rank, world)
                seen += list(range(begin, begin + count))
            assert seen == list(range(spp)), (world, spp)
        parts = [D.tile_part(rank, world) for rank in range(world)]
        assert sorted(p[0] for p in parts) == list(range(world)) and all(p[1] == world for p in parts)
