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


def _roofline(workload, key, form, rays_per_s, n_gpus=1):
    return bench.roofline(workload, key, form, rays_per_s, n_gpus, 1000.0, 100.0, {"box": 1.0, "tri": 1.0, "sph": 0.0},
                          {"kernel": "k_path", "rays_per_launch": 1_000_000})


STRICT_FORMS = ("sweep", "walk", "walk_lds", "walk_refill", "walk_lds_refill")
WORKLOAD_NAMES = ("cornell_plane_1024_1k", "world1_1080p_256", "thing_glass_2048_512", "cornell_full_512_64", "soup_10m_4k_256")


def _records():
    with open(bench.PMC_FILE) as f:
        return json.load(f)["records"]


def _strict_key(workload):
    """the record of the workload's reference-order form: <workload>/reference/<form>"""
    keys = [k for k, r in _records().items() if k.startswith(workload + "/reference/") and r["form"] in STRICT_FORMS]
    assert len(keys) == 1, (workload, keys)
    return keys[0].split("/", 1)[1]


def test_committed_records_are_complete():
    """Every benchmark workload has a record of its reference-order form and one of the form the
    library picks by itself (keys <workload>/<tree>/<kernel form>), each with the hash of the kernel
    sources it was taken on (bench.py compares that hash with the tree's and flags a stale record in
    its output; a kernel edit is not a test failure) and the split of its instructions over the stages."""
    records = _records()
    for wl in WORKLOAD_NAMES:
        _strict_key(wl)
        assert any(k.startswith(wl + "/") and r.get("picked_by_auto") for k, r in records.items()), wl
    for key, rec in records.items():
        wl, tree, form = key.split("/")
        assert wl in WORKLOAD_NAMES and tree in ("reference", "sah") and form == rec["form"] and form in bench.rt.FORM_NAMES, key
        assert len(rec["source_hash"]) == 16, key
        for field in ("hbm_bytes_per_ray", "valu_insts_per_ray", "vmem_insts_per_ray", "ta_busy", "clock_ghz", "mrays_profiled"):
            assert rec[field] > 0, (key, field)
        stages = rec["stages"]
        assert set(stages) == {"raygen", "closest_hit", "shade", "accumulate"}, key
        assert abs(sum(v["time_share"] for v in stages.values()) - 1.0) < 0.02
        assert stages["closest_hit"]["valu_insts_per_ray"] > stages["shade"]["valu_insts_per_ray"] > 0


@pytest.mark.parametrize("workload", WORKLOAD_NAMES)
def test_fraction_is_the_largest_measured_bound(workload):
    key = _strict_key(workload)
    form = key.split("/")[1]
    rec = bench.pmc_record(workload, key)
    r = _roofline(workload, key, form, rec["mrays_unprofiled"] * 1e6)
    parts = {"hbm": r["hbm"]["frac"], "valu_issue": r["valu_issue"]["frac"], "address_unit": r["address_unit"]["busy_share"]}
    assert r["bound"] == max(parts, key=parts.get)
    assert r["frac"] == pytest.approx(parts[r["bound"]], abs=1e-4) and 0.0 < r["frac"] <= 1.0
    assert r["pmc"]["stale"] == (rec["source_hash"] != bench.kernel_source_hash())
    assert ("note" in r) == r["pmc"]["stale"]
    assert r["traffic"] == int(rec["hbm_bytes_per_ray"] * 1_000_000)
    assert r["stages"] == rec["stages"]
    # the clock: this run's where it was measured, else the profiled run's (and the line says which)
    assert r["valu_issue"]["clock_ghz"] == rec["clock_ghz"] and "profiled" in r["valu_issue"]["clock_source"]
    own = bench.roofline(workload, key, form, rec["mrays_unprofiled"] * 1e6, 1, 1000.0, 100.0, {"box": 1.0, "tri": 1.0, "sph": 0.0},
                         {"kernel": "k_path", "rays_per_launch": 1_000_000}, clock_ghz=2.0 * rec["clock_ghz"])
    assert own["valu_issue"]["frac"] == pytest.approx(r["valu_issue"]["frac"] / 2.0, rel=1e-3)
    assert "this run" in own["valu_issue"]["clock_source"] and own["valu_issue"]["clock_ghz_profiled_run"] == rec["clock_ghz"]
    # the fractions scale with the measured rate and are clamped, the algorithmic figure is kept apart
    fast = _roofline(workload, key, form, rec["mrays_unprofiled"] * 1e6 * 50)
    assert fast["frac"] == 1.0 and fast["algorithmic"]["GB/s"] == 1000.0
    # N GPUs: per-GPU rates against per-GPU peaks
    two = _roofline(workload, key, form, rec["mrays_unprofiled"] * 2e6, n_gpus=2)
    assert two["frac"] == pytest.approx(r["frac"], abs=1e-4)


def test_missing_and_mismatched_records_are_said_so():
    r = _roofline("no_such_workload", "reference/walk", "walk", 1e9)
    assert r["frac"] is None and r["traffic"] is None and "no PMC record" in r["note"]
    key = _strict_key("cornell_plane_1024_1k")
    r = _roofline("cornell_plane_1024_1k", key, "walk_lds", 1e10)            # the record is the sweep's
    assert r["pmc"]["profiled_form"] == "sweep" and "another kernel form" in r["note"]


def test_the_soup_is_bound_by_memory_once_it_walks_the_wide_tree():
    """The form the library picks for the 10 M-triangle soup is the wide walk over the surface-area trees:
    its record says HBM (or the address units) is what bounds that kernel, at a fraction of the
    bytes per ray the reference-order walk fetches."""
    picked = [(k, r) for k, r in _records().items() if k.startswith("soup_10m_4k_256/") and r.get("picked_by_auto")]
    assert len(picked) == 1 and picked[0][1]["form"] == "wide_refill"
    key, rec = picked[0][0].split("/", 1)[1], picked[0][1]
    r = _roofline("soup_10m_4k_256", key, "wide_refill", rec["mrays_unprofiled"] * 1e6)
    assert r["bound"] in ("hbm", "address_unit") and 0.4 < r["frac"] <= 1.0
    assert rec["hbm_bytes_per_ray"] < 0.5 * bench.pmc_record("soup_10m_4k_256", _strict_key("soup_10m_4k_256"))["hbm_bytes_per_ray"]
