"""bench.py's host-side bookkeeping (no GPU): what the JSON line derives from the kernel profile and from profiles/."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_end_to_end_fraction_against_the_implemented_sort():
    """One histogram pass + 24 B per scatter pass launched + the run pass, instead of SURVEY's eight-pass 200 B/pt."""
    args = argparse.Namespace(steps=3, sampler="RANDOM_GRID")
    # (12 brackets per step: four passes over 1 B points and the eight tiny passes over the 32 768-key sample)
    prof = {"radix_scatter": {"launches": 36, "total_ms": 99.0, "algorithmic_bytes": 3 * (4 * 24 * 10**9 + 8 * 24 * 32768)},
            "radix_runs": {"launches": 3, "total_ms": 17.0}}
    r = bench.implemented_sort_frac(prof, args, 4.0, 1e9, 0.1)
    assert r["scatter_passes"] == 4.0 and abs(r["sort_bytes_per_point"] - (8 + 4 * 24 + 24)) < 0.1
    assert abs(r["bytes_per_point"] - (32 + 128 + 4 * 33)) < 0.1
    assert abs(r["frac"] - (292.0 * 1e9 / 0.1) / (bench.HBM_PEAK_GBS * 1e9)) < 1e-4
    args.sampler = "MIN_DISTANCE"
    assert abs(bench.implemented_sort_frac(prof, args, 4.0, 1e9, 0.1)["bytes_per_point"] - (32 + 128 + 4 * 57)) < 0.1
    assert bench.implemented_sort_frac({}, args, 4.0, 1e9, 0.1) is None


def test_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/rNN/traffic.json only while schwarzwald_amd/csrc still hashes to the stamp."""
    sha = bench.library_source_sha16()
    assert len(sha) == 16
    value, source = bench.measured_traffic("sample_min_distance", 1_000_000_000, "MIN_DISTANCE")
    assert value is None or "profiles/" in source
    if value is None:
        assert "stale" in source or "no " in source or source


def _args(argv):
    import sys
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse_args()
    finally:
        sys.argv = old


def test_baseline_config_presets():
    """--config 4 / 5 are BASELINE.json's multi-GPU configurations as one command each: total sizes split over the ranks
    (strong scaling), sampler / payload / staging set; --total-points scales them down for a dry run."""
    a = _args(["--config", "4", "--gpus", "8"])
    assert (a.sampler, a.strategy, a.batches, a.total_points, a.points) == ("JITTERED", "ACCURATE", 1, 1_000_000_000, 125_000_000)
    a = _args(["--config", "5", "--gpus", "8"])
    assert (a.sampler, a.total_points, a.points, a.staged, a.payload, a.md_mode) == ("MIN_DISTANCE", 4_000_000_000, 500_000_000, True,
                                                                                     "rgb,intensity", "exact")
    assert a.batches == 10  # about 50 M points per batch and rank
    a = _args(["--config", "5", "--gpus", "8", "--total-points", "200000000", "--one-device"])
    assert a.points == 25_000_000 and a.batches == 2 and a.one_device
    a = _args(["--gpus", "2", "--total-points", "1000"])
    assert a.points == 500 and a.config == 0


def test_shard_stamps_that_do_not_fit_a_step_are_refused():
    """ADVICE / VERDICT r5: a stamp taken against an unset clock printed exchange_ms = 1.85e8 into a committed line."""
    import bench
    ok = [{"shard": 0, "exchange_done_ms": 3.0, "root_begun_ms": 3.1, "root_done_ms": 20.0, "levels_done_ms": 41.0}]
    assert bench.check_shard_stamps(ok, 42.0) is None
    assert "185461977" in bench.check_shard_stamps([{"shard": 1, "exchange_done_ms": 185461977.9, "root_begun_ms": 0.0}], 820.0)
    assert bench.check_shard_stamps([{"shard": 0, "exchange_ms": -1.0}], 10.0) is not None
    assert bench.check_shard_stamps([{"shard": 0, "exchange_ms": float("nan")}], 10.0) is not None
    assert bench.shard_stamp_error([{"rank": 0, "exchange_ms": 2.0, "levels_ms": 5.0, "root_mode": "joint", "points": 7}], 9.0) is None
    assert bench.shard_stamp_error([{"rank": 0, "exchange_ms": 2.0e9, "root_mode": "joint"}], 9.0) is not None


def test_bench_looks_for_the_newest_committed_profile_first():
    import bench
    rounds = bench.profile_rounds()
    assert rounds == sorted(rounds, reverse=True) and all(r[0] == "r" and r[1:].isdigit() for r in rounds)
    newest = sorted(d for d in os.listdir(os.path.join(bench.ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit())[-1]
    assert rounds[0] == newest
