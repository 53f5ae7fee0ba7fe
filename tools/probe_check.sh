#!/bin/bash
# What the first render's timing says about EVERY form on either tree (RT_AUTO_PROBE_ALL=1) on every benchmark workload (run on the GPU box from the
# repository root): one line per workload with the rate of the frame, then the probe's timings.
#   bash tools/probe_check.sh [build]          # auto (default) | reference | sah
build=${1:-auto}
for wl in cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64 cornell_full_512_256 soup_10m_4k_8 soup_10m_4k_256; do
  case $wl in
    cornell_full_512_256) w=cornell_full_512_64; extra="--spp 256";;
    soup_10m_4k_8) w=soup_10m_4k_256; extra="--spp 8";;
    *) w=$wl; extra="";;
  esac
  RT_AUTO_PROBE_ALL=1 RT_AUTOTUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu --no-reference-leg --build $build --workload $w $extra --steps 2 2> gpurun_out/probe.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['config']['tree'], d['config']['kernel_form'], d['config']['scene_build_s'])" || exit 1
  grep "rt probe" gpurun_out/probe.err
done
