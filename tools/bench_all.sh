#!/bin/bash
# Every benchmark workload on one GPU, reference and SAH trees (run on the GPU box from the
# repository root): one bench.py JSON line per run into <out file>.
#   bash tools/bench_all.sh <out file>
set -e
out=${1:-gpurun_out/all_workloads.jsonl}
: > $out
for build in reference sah; do
    for wl in cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64; do
        timeout -k 10 400 python3 -u bench.py --no-cpu --workload $wl --build $build | tail -n 1 >> $out
    done
    timeout -k 10 900 python3 -u bench.py --no-cpu --workload soup_10m_4k_256 --build $build --steps 1 --warmup 0 | tail -n 1 >> $out
done
timeout -k 10 900 python3 -u bench.py --no-cpu --workload soup_10m_4k_256 --build sah --traversal octant --steps 1 --warmup 0 | tail -n 1 >> $out
# the opt-in wide walk (eight-child nodes collapsed from either tree)
for build in reference sah; do
    for wl in cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64; do
        timeout -k 10 400 python3 -u bench.py --no-cpu --workload $wl --build $build --traversal wide | tail -n 1 >> $out
    done
    timeout -k 10 900 python3 -u bench.py --no-cpu --workload soup_10m_4k_256 --build $build --traversal wide --steps 1 --warmup 0 | tail -n 1 >> $out
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]; t = r["algorithmic"]["tests_per_ray"]
    print(f'{d["config"]["workload"].split(":")[0]:24s} build {d["config"].get("build","?"):9s} traversal {d["config"].get("traversal","?"):8s} form {d["config"]["kernel_form"]:16s} '
          f'{d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.1f} ms/frame  tests/ray box {t["box"]} tri {t["tri"]} sph {t["sph"]}  '
          f'bound {r["bound"]} {r["frac"]}')
PY
