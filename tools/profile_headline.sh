#!/bin/bash
# Kernel-trace profiles of the default benchmark command (run on the GPU box from the repository
# root):  bash tools/profile_headline.sh <out dir>
# rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu` and of its --staged form; the
# average duration of the per-batch kernel k_path there is what bench.py's avg_launch_ms must agree with.  (The PMC
# passes of every workload, the headline included, are tools/profile_workloads.sh.)
set -e
out=${1:-gpurun_out/headline}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_staged -o s --output-format csv -- python3 bench.py --no-cpu --staged > $out/bench_staged_under_rocprof.json 2> $out/stats_staged.err
cp $out/stats/s_kernel_stats.csv $out/kernel_stats.csv
cp $out/stats_staged/s_kernel_stats.csv $out/kernel_stats_staged.csv
rm -rf $out/stats $out/stats_staged
head -3 $out/kernel_stats.csv
