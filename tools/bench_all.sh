#!/bin/bash
# Every benchmark workload on one GPU (run on the GPU box from the repository root): one bench.py JSON
# line per run into <out file>.
#   bash tools/bench_all.sh <out file>
# Per workload: the defaults (both trees built, the first render keeps the fastest exact form; the line
# also carries the rate of the reference's trees in the reference's order), then the surface-area trees
# alone, the wide walk over either tree and the wide tree built on the GPU.
set -e
out=${1:-gpurun_out/all_workloads.jsonl}
: > $out
small="cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64"
soup="--workload soup_10m_4k_256 --steps 1 --warmup 1"
for wl in $small; do
    timeout -k 10 400 python3 -u bench.py --no-cpu --workload $wl | tail -n 1 >> $out
done
timeout -k 10 900 python3 -u bench.py --no-cpu $soup | tail -n 1 >> $out
for extra in "--build sah" "--build reference --traversal wide" "--build sah --traversal wide" "--build reference --traversal wide --wide-build device"; do
    for wl in $small; do
        timeout -k 10 400 python3 -u bench.py --no-cpu --no-reference-leg --workload $wl $extra | tail -n 1 >> $out
    done
    timeout -k 10 900 python3 -u bench.py --no-cpu --no-reference-leg $soup $extra | tail -n 1 >> $out
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]; t = r["algorithmic"]["tests_per_ray"]; c = d["config"]
    ref = d.get("reference_order") or {}
    print(f'{c["workload"].split(":")[0]:24s} build {c["build"]:9s} traversal {c["traversal"]:5s} {c.get("wide_build", ""):8s} -> {c["tree"]:9s} {c["kernel_form"]:16s} '
          f'{d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.1f} ms/frame  tests/ray box {t["box"]} tri {t["tri"]} sph {t["sph"]}  '
          f'bound {r["bound"]} {r["frac"]}' + (f'   reference order: {ref["value"]:.1f} ({ref["kernel_form"]})' if "value" in ref else ""))
PY
