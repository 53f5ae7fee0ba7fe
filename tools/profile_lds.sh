#!/bin/bash
# LDS counters of the walks out of LDS (run on the GPU box from the repository root):
#   bash tools/profile_lds.sh <out dir>
# One rocprofv3 --pmc pass per workload with the surface-area tree and the per-lane walk pinned; prints, per
# k_path kernel, the LDS array's active cycles, the cycles bank conflicts added, and the LDS instructions.
out=${1:-gpurun_out/lds}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
for c in "cornell_plane_1024_1k 256" "world1_1080p_256 64" "thing_glass_2048_512 32"; do
    set -- $c
    ( export RT_NO_AUTOTUNE=1; timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
        -d $out/$1 -o lds --output-format csv -- python3 bench.py --no-cpu --no-reference-leg --steps 1 --warmup 0 \
        --workload $1 --spp $2 --build sah --traversal per_lane > $out/$1.json 2> $out/$1.err ) || { tail -3 $out/$1.err; exit 1; }
    f=$(find $out/$1 -name "*counter_collection.csv" | head -1)
    echo "== $1"
    python3 tools/pmc_summary.py $f | grep -A8 "k_path"
    rm -rf $out/$1
done
