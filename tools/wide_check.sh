# Wide walk: its GPU tests, then every workload on both build modes (run on the GPU box from the repository root)
set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_ordered.py -x -q -k wide --timeout 300 --timeout-method thread -s > gpurun_out/w4_tests.txt 2>&1 || { tail -40 gpurun_out/w4_tests.txt; exit 1; }
tail -3 gpurun_out/w4_tests.txt
out=gpurun_out/w4_bench.jsonl
: > $out
for refill in 0 16; do
for b in reference sah; do
    for wl in cornell_full_512_64 thing_glass_2048_512 world1_1080p_256; do
        RT_REFILL=$refill timeout -k 10 300 python3 -u bench.py --no-cpu --workload $wl --build $b --traversal wide 2>/dev/null | tail -n 1 >> $out
    done
    RT_REFILL=$refill timeout -k 10 600 python3 -u bench.py --no-cpu --workload soup_10m_4k_256 --build $b --traversal wide --steps 1 --warmup 0 2>/dev/null | tail -n 1 >> $out
done
done
python3 - <<'PY'
import json
for l in open('gpurun_out/w4_bench.jsonl'):
    d=json.loads(l); print(d['config']['workload'].split(':')[0], d['config']['build'], d['config']['kernel_form'], d['value'], d['roofline']['algorithmic']['tests_per_ray'])
PY
