#!/bin/bash
# Profiles of the default benchmark command (run on the GPU box from the repository root):
#   bash tools/profile_headline.sh <out dir>
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu` (and of --staged);
# 2. three --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE: separate runs) of
#    `python3 bench.py --no-cpu --steps 1 --warmup 0 --traversal sweep` (the form the unprofiled run
#    takes: counters serialise kernels, which would tip the first render's timing of the candidate
#    forms the other way), summarised per kernel by tools/pmc_summary.py.
set -e
out=${1:-gpurun_out/headline}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_staged -o s --output-format csv -- python3 bench.py --no-cpu --staged > $out/bench_staged_under_rocprof.json 2> $out/stats_staged.err
for pass in sq fetch write; do
    case $pass in
        sq) ctr="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY";;
        fetch) ctr="FETCH_SIZE";;
        write) ctr="WRITE_SIZE";;
    esac
    timeout -s KILL 300 rocprofv3 --pmc $ctr -d $out/$pass -o p --output-format csv -- python3 bench.py --no-cpu --steps 1 --warmup 0 --traversal sweep > $out/$pass.json 2> $out/$pass.err
    python3 tools/pmc_summary.py $out/$pass/p_counter_collection.csv > $out/pmc_$pass.txt
done
cp $out/stats/s_kernel_stats.csv $out/kernel_stats.csv
cp $out/stats_staged/s_kernel_stats.csv $out/kernel_stats_staged.csv
rm -rf $out/stats $out/stats_staged $out/sq $out/fetch $out/write
head -3 $out/kernel_stats.csv; grep -A3 "k_path<1>" $out/pmc_fetch.txt $out/pmc_write.txt
