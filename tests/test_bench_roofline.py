"""bench.py's roofline object: built from the committed per-ray counter records, a fraction of the
largest of three measured bounds, never above 1, and explicit about records that are missing or were
taken on other kernel sources."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402


def _roofline(workload, build, form, rays_per_s, n_gpus=1):
    return bench.roofline(workload, build, form, rays_per_s, n_gpus, 1000.0, 100.0, {"box": 1.0, "tri": 1.0, "sph": 0.0},
                          {"kernel": "k_path", "rays_per_launch": 1_000_000})


def test_committed_records_are_complete():
    """Every benchmark workload has a record, with the form it was profiled with and the hash of the
    kernel sources it was taken on (bench.py compares that hash with the tree's and flags a stale
    record in its output; a kernel edit is not a test failure)."""
    with open(bench.PMC_FILE) as f:
        records = json.load(f)["records"]
    assert {"cornell_plane_1024_1k/reference", "world1_1080p_256/reference", "thing_glass_2048_512/reference",
            "cornell_full_512_64/reference", "soup_10m_4k_256/reference"} <= set(records)
    for key, rec in records.items():
        assert len(rec["source_hash"]) == 16, key
        opt_in = key.endswith("/wide")             # records of the opt-in wide walk: <workload>/<build>/wide
        assert rec["form"] in (("wide", "wide_refill") if opt_in else ("sweep", "walk", "walk_lds", "walk_refill", "walk_lds_refill")), key
        for field in ("hbm_bytes_per_ray", "valu_insts_per_ray", "vmem_insts_per_ray", "ta_busy", "clock_ghz", "mrays_profiled"):
            assert rec[field] > 0, (key, field)


@pytest.mark.parametrize("workload,form", [("cornell_plane_1024_1k", "sweep"), ("world1_1080p_256", "walk_lds"),
                                           ("cornell_full_512_64", "walk_refill"), ("soup_10m_4k_256", "walk")])
def test_fraction_is_the_largest_measured_bound(workload, form):
    rec = bench.pmc_record(workload, "reference")
    r = _roofline(workload, "reference", form, rec["mrays_unprofiled"] * 1e6)
    parts = {"hbm": r["hbm"]["frac"], "valu_issue": r["valu_issue"]["frac"], "address_unit": r["address_unit"]["busy_share"]}
    assert r["bound"] == max(parts, key=parts.get)
    assert r["frac"] == pytest.approx(parts[r["bound"]], abs=1e-4) and 0.0 < r["frac"] <= 1.0
    assert r["pmc"]["stale"] == (rec["source_hash"] != bench.kernel_source_hash())
    assert ("note" in r) == r["pmc"]["stale"]
    assert r["traffic"] == int(rec["hbm_bytes_per_ray"] * 1_000_000)
    # the fractions scale with the measured rate and are clamped, the algorithmic figure is kept apart
    fast = _roofline(workload, "reference", form, rec["mrays_unprofiled"] * 1e6 * 50)
    assert fast["frac"] == 1.0 and fast["algorithmic"]["GB/s"] == 1000.0
    # N GPUs: per-GPU rates against per-GPU peaks
    two = _roofline(workload, "reference", form, rec["mrays_unprofiled"] * 2e6, n_gpus=2)
    assert two["frac"] == pytest.approx(r["frac"], abs=1e-4)


def test_missing_and_mismatched_records_are_said_so():
    r = _roofline("no_such_workload", "reference", "walk", 1e9)
    assert r["frac"] is None and r["traffic"] is None and "no PMC record" in r["note"]
    r = _roofline("cornell_plane_1024_1k", "reference", "walk_lds", 1e10)     # the record is the sweep's
    assert r["pmc"]["profiled_form"] == "sweep" and "another kernel form" in r["note"]


def test_the_wide_walk_has_records_of_its_own():
    """bench.py --traversal wide looks its counters up under <workload>/<build>/wide: the soup's record
    there says HBM is the bound of that kernel."""
    rec = bench.pmc_record("soup_10m_4k_256", "reference/wide")
    assert rec is not None and rec["form"] == "wide_refill"
    r = _roofline("soup_10m_4k_256", "reference/wide", "wide_refill", rec["mrays_unprofiled"] * 1e6)
    assert r["bound"] in ("hbm", "address_unit") and 0.4 < r["frac"] <= 1.0
    assert rec["hbm_bytes_per_ray"] < 0.5 * bench.pmc_record("soup_10m_4k_256", "reference")["hbm_bytes_per_ray"]
