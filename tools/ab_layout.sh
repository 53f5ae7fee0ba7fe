#!/bin/bash
# A/B of the per-lane walks' memory layouts on one GPU (run on the GPU box from the repository root):
#   bash tools/ab_layout.sh <out file> [workload ...]
# One bench.py line per (workload, layout): pre-order (a box, then its first child), and the linked
# layout with sibling pairs side by side for several alignments / breadth-first budgets.
out=${1:-gpurun_out/ab_layout.jsonl}
shift
wls=${@:-"cornell_full_512_64 soup_10m_4k_256"}
: > $out
run() {   # label, env assignments..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 400 python3 -u bench.py --no-cpu "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); d['label']='$label'; print(json.dumps(d))" >> $out || echo "FAILED $label" >> $out
}
for wl in $wls; do
    extra=""
    [ "$wl" = soup_10m_4k_256 ] && extra="--steps 1 --warmup 0 --spp 32"
    run "$wl preorder" RT_LAYOUT_PREORDER=1 -- --workload $wl $extra
    run "$wl pairs top0 align4" RT_LAYOUT_TOP=0 -- --workload $wl $extra
    run "$wl pairs top2048 align4" X=1 -- --workload $wl $extra
    run "$wl pairs top0 align1" RT_LAYOUT_TOP=0 RT_LAYOUT_ALIGN=1 -- --workload $wl $extra
    run "$wl pairs top0 align8" RT_LAYOUT_TOP=0 RT_LAYOUT_ALIGN=8 -- --workload $wl $extra
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        print(l.strip()); continue
    d = json.loads(l)
    print(f'{d["label"]:45s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.1f} ms  form {d["config"]["kernel_form"]}')
PY
