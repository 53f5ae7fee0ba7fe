#!/bin/bash
# Copies what tools/profile_workloads.sh (reference, then auto) and tools/profile_headline.sh left under
# gpurun_out/ into profiles/ and merges the per-ray counter records (run here, after the GPU call).
#   bash tools/install_profiles.sh <round tag, e.g. r03>
set -e
r=${1:-r03}
python3 tools/merge_pmc.py profiles/pmc_workloads.json gpurun_out/${r}_workloads_ref/pmc_workloads.json gpurun_out/${r}_workloads_auto/pmc_workloads.json
rm -rf profiles/${r}_workloads && mkdir -p profiles/${r}_workloads
for d in ref auto; do
    cp gpurun_out/${r}_workloads_$d/*_plain.json gpurun_out/${r}_workloads_$d/*_sq.json gpurun_out/${r}_workloads_$d/*_st.json profiles/${r}_workloads/
done
cp gpurun_out/${r}_workloads_ref/pmc_workloads.json profiles/${r}_workloads/pmc_reference.json
cp gpurun_out/${r}_workloads_auto/pmc_workloads.json profiles/${r}_workloads/pmc_auto.json
cp gpurun_out/${r}_workloads_ref/table.md profiles/${r}_workloads_table_reference.md
cp gpurun_out/${r}_workloads_auto/table.md profiles/${r}_workloads_table.md
if [ -f gpurun_out/headline/kernel_stats.csv ]; then
    cp gpurun_out/headline/kernel_stats.csv profiles/${r}_kernel_stats.csv
    cp gpurun_out/headline/kernel_stats_staged.csv profiles/${r}_kernel_stats_staged.csv
    cp gpurun_out/headline/bench_under_rocprof.json profiles/${r}_bench_under_rocprof.json
    cp gpurun_out/headline/bench_staged_under_rocprof.json profiles/${r}_bench_staged_under_rocprof.json
fi
