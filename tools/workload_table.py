"""Per-workload hardware table from PMC passes of k_path (rocpd databases written by rocprofv3):
    python tools/workload_table.py <dir> name ...      # expects <dir>/<name>_{sq,f,w}/*_results.db and <name>_sq.json
Columns: rays/s under the profiler, HBM GB/s (2*FETCH_SIZE + WRITE_SIZE), vector instructions per
SIMD cycle (the part sustains one per 2.1-3.1 cycles), active lanes per vector instruction,
resident waves per SIMD, share of wave time in s_waitcnt."""
import glob
import json
import sqlite3
import sys


def sums(path):
    db = sqlite3.connect(glob.glob(path + "/*_results.db")[0])
    out, dur = {}, 0
    for name, val in db.execute("select counter_name, sum(counter_value) from pmc_events where name like '%k_path%' group by counter_name"):
        out[name] = val
    dur = db.execute("select sum(duration) from (select distinct dispatch_id, duration from pmc_events where name like '%k_path%')").fetchone()[0]
    return out, dur * 1e-9


d = sys.argv[1]
print("| workload | Mrays/s (profiled) | HBM GB/s | of 8 TB/s | VALU instr / SIMD cycle | active lanes / 64 | waves / SIMD | wait share |")
print("|---|---|---|---|---|---|---|---|")
for name in sys.argv[2:]:
    sq, t = sums(f"{d}/{name}_sq")
    f, tf = sums(f"{d}/{name}_f")
    w, tw = sums(f"{d}/{name}_w")
    j = json.loads(open(f"{d}/{name}_sq.json").read().strip().splitlines()[-1])
    rays = j["config"]["rays_per_step"]
    simd_cycles = sq["GRBM_GUI_ACTIVE"] / 8 * 1024
    hbm = (2 * f["FETCH_SIZE"] / tf + w["WRITE_SIZE"] / tw) * 1024 / 1e9
    print(f"| {j['config']['workload'].split(':')[0]} | {rays / t / 1e6:.0f} | {hbm:.0f} | {hbm / 8000:.2f} | "
          f"{sq['SQ_INSTS_VALU'] / simd_cycles:.3f} | {sq['SQ_THREAD_CYCLES_VALU'] / sq['SQ_ACTIVE_INST_VALU']:.1f} | "
          f"{sq['SQ_WAVE_CYCLES'] * 4 / simd_cycles:.1f} | {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.2f} |")
