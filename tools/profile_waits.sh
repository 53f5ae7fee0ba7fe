#!/bin/bash
# What the waves of the walks out of LDS spend their cycles on (run on the GPU box from the repository root):
#   bash tools/profile_waits.sh <out dir>
#   WORKLOADS="cornell_full_512_64:256 soup_10m_4k_256:8" TRAVERSAL=wide bash tools/profile_waits.sh <out dir>
# Two rocprofv3 --pmc passes per workload with the surface-area tree and the per-lane walk pinned: wave
# cycles split into waiting (any reason / for an instruction slot / for the LDS) and issuing (by kind), and
# the LDS instructions in flight (level counter / instructions = their mean latency).
out=${1:-gpurun_out/waits}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
p1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
p2="SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for c in ${WORKLOADS:-cornell_plane_1024_1k:256 world1_1080p_256:64 thing_glass_2048_512:32}; do
    set -- ${c/:/ }
    n=0
    for ctr in "$p1" "$p2"; do
        n=$((n + 1))
        ( export RT_NO_AUTOTUNE=1; timeout -s KILL 200 rocprofv3 --pmc $ctr \
            -d $out/$1.p$n -o w --output-format csv -- python3 bench.py --no-cpu --no-reference-leg --steps 1 --warmup 0 \
            --workload $1 --spp $2 --build sah --traversal ${TRAVERSAL:-per_lane} > $out/$1.p$n.json 2> $out/$1.p$n.err ) || { tail -3 $out/$1.p$n.err; exit 1; }
        f=$(find $out/$1.p$n -name "*counter_collection.csv" | head -1)
        echo "== $1 pass $n"
        python3 tools/pmc_summary.py $f | grep -A10 "k_path"
        rm -rf $out/$1.p$n
    done
done
