#!/bin/bash
# Memory-path counters of the per-batch kernel (run on the GPU box from the repository root):
#   bash tools/profile_mempath.sh <out dir> <workload> <spp> [extra bench.py arguments]
# Three rocprofv3 --pmc passes (address unit / L1 / TLB counters do not fit one), each summarised per
# kernel by tools/pmc_summary.py.
out=$1; wl=$2; spp=$3; shift 3
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
p1="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
p2="GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
p3="GRBM_GUI_ACTIVE TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"
n=0
for ctr in "$p1" "$p2" "$p3"; do
    n=$((n + 1))
    timeout -s KILL 400 rocprofv3 --pmc $ctr -d $out/p$n -o p --output-format csv -- python3 bench.py --no-cpu --steps 1 --warmup 0 \
        --workload $wl --spp $spp "$@" > $out/p$n.json 2> $out/p$n.err || { tail -5 $out/p$n.err; exit 1; }
    python3 tools/pmc_summary.py $(find $out/p$n -name "*counter_collection.csv" | head -1) > $out/pmc_p$n.txt
    rm -rf $out/p$n
done
cat $out/pmc_p*.txt | grep -A12 "k_path"
