# Wide walk on the 10 M-triangle soup, both build modes (run on the GPU box from the repository root)
set -e
out=gpurun_out/w2_bench.jsonl
: > $out
for b in reference sah; do
    RT_BUILD_TIMING=1 timeout -k 10 600 python3 -u bench.py --no-cpu --workload soup_10m_4k_256 --build $b --traversal wide --steps 1 --warmup 0 2> gpurun_out/w2_$b.err | tail -n 1 >> $out
done
python3 - <<'PY'
import json
for l in open('gpurun_out/w2_bench.jsonl'):
    d=json.loads(l); print(d['config']['workload'].split(':')[0], d['config']['build'], d['config']['kernel_form'], d['value'], d['config'].get('scene_build_s'), d['roofline']['algorithmic']['tests_per_ray'])
PY
