for wl in cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64 cornell_full_512_256 soup_10m_4k_256; do
  w=$wl; extra=""
  if [ $wl = cornell_full_512_256 ]; then w=cornell_full_512_64; extra="--spp 256"; fi
  RT_AUTOTUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu --no-reference-leg --workload $w $extra --steps 2 2> gpurun_out/probe.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['config']['tree'], d['config']['kernel_form'], d['config']['scene_build_s'])" || exit 1
  grep "rt probe" gpurun_out/probe.err
done
