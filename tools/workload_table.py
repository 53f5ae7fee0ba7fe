"""Per-workload hardware table from PMC passes of k_path (rocpd databases written by rocprofv3):
    python tools/workload_table.py <dir> name ...
expects <dir>/<name>_{sq,f,w}/*_results.db and the bench lines <name>_{plain,sq}.json, prints the
table and writes <dir>/pmc_workloads.json.

Columns: rays/s under the profiler, HBM GB/s (2*FETCH_SIZE + WRITE_SIZE: FETCH_SIZE reads half of a
16-byte-per-lane stream on gfx950), vector instructions per SIMD cycle (the part sustains one per
2.1 cycles, 3.07 for packed f32), active lanes per vector instruction, resident waves per SIMD,
share of wave time in s_waitcnt, wave-wide vector memory instructions per ray and the share of
time the CUs' address units (TA) were busy.

The JSON holds, per "<workload>/<tree>/<kernel form>", what does not depend on how fast the run was:
HBM bytes and vector instructions per ray, with the clock the chip held; bench.py multiplies them by
the rate (and divides by the clock) it measures itself.  Where <name>_st holds the same counters for
the staged launches (raygen, closest hit and shade as kernels of their own), the record also says
which stage the instructions, the memory instructions and the time belong to."""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sums(path, pattern):
    db = sqlite3.connect(glob.glob(path + "/*_results.db")[0])
    out = {}
    for name, val in db.execute("select counter_name, sum(counter_value) from pmc_events where name like ? group by counter_name", (pattern,)):
        out[name] = val
    dur = db.execute("select sum(duration) from (select distinct dispatch_id, duration from pmc_events where name like ?)", (pattern,)).fetchone()[0]
    return out, (dur or 0) * 1e-9


def main():
    d = sys.argv[1]
    try:
        from bench import kernel_source_hash
        src_hash = kernel_source_hash()
    except Exception:
        src_hash = None
    records = {}
    stage_rows = []
    print("| workload | form | Mrays/s (unprofiled) | Mrays/s (profiled) | HBM bytes / ray | HBM GB/s | of 8 TB/s | VALU instr / ray | "
          "VALU instr / SIMD cycle | active lanes / 64 | waves / SIMD | wait share | clock GHz | VMEM instr / ray | TA busy |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name in sys.argv[2:]:
        j = json.loads(open(f"{d}/{name}_sq.json").read().strip().splitlines()[-1])
        plain = json.loads(open(f"{d}/{name}_plain.json").read().strip().splitlines()[-1])
        form = j["config"]["kernel_form"]
        tree = j["config"]["tree"]
        assert (tree, form) == (plain["config"]["tree"], plain["config"]["kernel_form"]), (name, tree, form, plain["config"])
        sq, t = sums(f"{d}/{name}_sq", "%k_path%")
        f, tf = sums(f"{d}/{name}_f", "%k_path%")
        w, tw = sums(f"{d}/{name}_w", "%k_path%")
        rays = j["config"]["rays_per_step"]
        cycles_xcd = sq["GRBM_GUI_ACTIVE"] / 8                     # summed over the 8 XCDs
        simd_cycles = cycles_xcd * 1024
        clock = cycles_xcd / t / 1e9
        hbm_per_ray = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / rays
        hbm = (2 * f["FETCH_SIZE"] / tf + w["WRITE_SIZE"] / tw) * 1024 / 1e9
        valu_per_ray = sq["SQ_INSTS_VALU"] / rays
        lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
        vmem_per_ray = (sq.get("SQ_INSTS_VMEM_RD", 0) + sq.get("SQ_INSTS_VMEM_WR", 0)) / rays
        # TA_TA_BUSY_sum adds up the 256 address units, GRBM_GUI_ACTIVE the 8 XCDs
        ta_busy = sq.get("TA_TA_BUSY_sum", 0) / 256 / cycles_xcd
        wl = j["config"]["workload"].split(":")[0]
        print(f"| {wl} ({tree} tree) | {form} | {plain['value']:.0f} | {rays / t / 1e6:.0f} | {hbm_per_ray:.0f} | {hbm:.0f} | {hbm / 8000:.2f} | "
              f"{valu_per_ray:.0f} | {sq['SQ_INSTS_VALU'] / simd_cycles:.3f} | {lanes:.1f} | "
              f"{sq['SQ_WAVE_CYCLES'] * 4 / simd_cycles:.1f} | {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.2f} | {clock:.2f} | "
              f"{vmem_per_ray:.2f} | {ta_busy:.2f} |")
        key = f"{wl}/{tree}/{form}"
        stages = None
        if glob.glob(f"{d}/{name}_st/*_results.db"):
            # the staged launches of the same job: one kernel per stage, so the counters say where the
            # per-batch kernel's instructions go (their sum per ray is the fused kernel's, up to the loop
            # around the stages)
            js = json.loads(open(f"{d}/{name}_st.json").read().strip().splitlines()[-1])
            rays_st = js["config"]["rays_per_step"]
            stages, total_t = {}, 0.0
            for stage, pat in (("raygen", "%k_raygen%"), ("closest_hit", "%k_trace<%"), ("shade", "%k_shade%"), ("accumulate", "%k_accumulate%")):
                c, ts = sums(f"{d}/{name}_st", pat)
                if not c:
                    continue
                total_t += ts
                stages[stage] = {"valu_insts_per_ray": round(c["SQ_INSTS_VALU"] / rays_st, 2),
                                 "salu_insts_per_ray": round(c["SQ_INSTS_SALU"] / rays_st, 2),
                                 "vmem_insts_per_ray": round((c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / rays_st, 3),
                                 "active_lanes": round(c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1), 1),
                                 "wait_share": round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
                                 "kernel_seconds": ts}
            for v in stages.values():
                v["time_share"] = round(v.pop("kernel_seconds") / total_t, 3) if total_t else None
            stage_rows.append((wl, tree, form, stages))
        records[key] = {
            "form": form, "tree": tree, "stages": stages, "rays_profiled": rays,
            # the library took this tree and form by itself (the plain run was an RT_BUILD_AUTO / RT_TRAVERSAL_AUTO one)
            "picked_by_auto": plain["config"]["build"] == "auto" and plain["config"]["traversal"] == "auto", "spp_profiled": j["config"]["workload"].split(", ")[1],
            "hbm_bytes_per_ray": round(hbm_per_ray, 2),
            "fetch_size_kib": f["FETCH_SIZE"], "write_size_kib": w["WRITE_SIZE"],
            "valu_insts_per_ray": round(valu_per_ray, 2), "salu_insts_per_ray": round(sq["SQ_INSTS_SALU"] / rays, 2),
            "vmem_insts_per_ray": round(vmem_per_ray, 3), "ta_busy": round(ta_busy, 3),
            "active_lanes": round(lanes, 1), "waves_per_simd": round(sq["SQ_WAVE_CYCLES"] * 4 / simd_cycles, 2),
            "wait_share": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 3), "clock_ghz": round(clock, 3),
            # SIMD cycles one vector instruction occupies with 8 waves per SIMD (tools/valu_rate.hip): 2.1 for
            # plain f32 / select / compare, 3.07 for packed f32, of which the sweep's stream holds 45 %
            "simd_cycles_per_valu_inst": 2.54 if form == "sweep" else 2.1,
            "mrays_unprofiled": plain["value"], "mrays_profiled": round(rays / t / 1e6, 1),
            "git_sha": os.environ.get("RT_GIT_SHA"), "source_hash": src_hash,
            "command": f"tools/profile_workloads.sh (bench.py --no-cpu --steps 1 --warmup 0 --workload {wl} --build {tree}, form pinned)",
        }
    if stage_rows:
        print()
        print("Stages of the same jobs as kernels of their own (RT_RENDER_STAGED), per ray of the job:")
        print()
        print("| workload | form | stage | share of kernel time | VALU instr / ray | SALU instr / ray | VMEM instr / ray | active lanes / 64 | wait share |")
        print("|---|---|---|---|---|---|---|---|---|")
        for wl, tree, form, stages in stage_rows:
            for stage, v in stages.items():
                print(f"| {wl} ({tree} tree) | {form} | {stage} | {v['time_share']:.2f} | {v['valu_insts_per_ray']:.1f} | {v['salu_insts_per_ray']:.1f} | "
                      f"{v['vmem_insts_per_ray']:.2f} | {v['active_lanes']:.1f} | {v['wait_share']:.2f} |")
    with open(f"{d}/pmc_workloads.json", "w") as out:
        json.dump({"note": "per-ray PMC records of the per-batch kernel k_path, written by tools/workload_table.py; "
                           "hbm_bytes_per_ray = (2*FETCH_SIZE + WRITE_SIZE) KiB / rays of the profiled step "
                           "(the launches of that step only: the form is pinned, so the first render probes nothing)",
                   "records": records}, out, indent=1)
        out.write("\n")


if __name__ == "__main__":
    main()
