#!/bin/bash
# A/B of builds of the library and of memory layouts on one GPU (run on the GPU box from the
# repository root):  bash tools/ab_kernels.sh <out file> "<lib ...>" "<layout env ...>" [workload ...]
# One bench.py line per (workload, library, layout).  A library is a path for RT_AMD_LIB ("default"
# = the in-tree one); a layout is a comma-separated list of environment assignments ("-" = none).
out=$1; libs=$2; layouts=$3; shift 3
wls=${@:-"cornell_full_512_64 soup_10m_4k_256"}
: > $out
for wl in $wls; do
    extra=""
    [ "$wl" = soup_10m_4k_256 ] && extra="--steps 1 --warmup 0 --spp 32"
    [ "$wl" = world1_1080p_256 ] && extra="--spp 64"
    [ "$wl" = thing_glass_2048_512 ] && extra="--spp 64"
    [ "$wl" = cornell_plane_1024_1k ] && extra="--spp 256"
    for lib in $libs; do
        for lay in $layouts; do
            envs=()
            [ "$lib" != default ] && envs+=("RT_AMD_LIB=$PWD/$lib")
            [ "$lay" != "-" ] && IFS=, read -ra more <<< "$lay" && envs+=("${more[@]}")
            label="$wl $lib $lay"
            env "${envs[@]}" timeout -k 10 400 python3 -u bench.py --no-cpu --no-reference-leg --workload $wl $extra 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); d['label']='$label'; print(json.dumps(d))" >> $out || echo "FAILED $label" >> $out
        done
    done
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        print(l.strip()); continue
    d = json.loads(l)
    print(f'{d["label"]:70s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.1f} ms  form {d["config"]["kernel_form"]}')
PY
