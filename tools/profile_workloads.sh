#!/bin/bash
# PMC passes of the per-batch kernel for every benchmark workload (run on the GPU box from the
# repository root):
#   bash tools/profile_workloads.sh <out dir> [build] [workload ...]      # build: auto (default) | reference | sah
# TRAVERSAL=wide (or ordered, octant, per_lane, sweep) in the environment profiles that walk instead of
# the form the library picks by itself.  Records are keyed <workload>/<tree>/<kernel form>.
# Per workload: one unprofiled run to learn which tree and which form of the closest-hit loop the first
# render's timing takes (the profiler serialises kernels, which would tip that timing), then four
# rocprofv3 --pmc runs with that tree and form pinned (SQ counters; FETCH_SIZE; WRITE_SIZE -- separate
# runs, the counter blocks do not all fit one; and the SQ counters once more on the staged launches,
# which attributes the instructions to raygen / closest hit / shade / accumulate), then
# tools/workload_table.py, which prints the tables and writes <out dir>/pmc_workloads.json: the per-ray
# records bench.py turns into roofline fractions.
set -e
out=${1:-gpurun_out/workloads}
build=${2:-auto}
shift 2 || true
wls=${@:-"cornell_plane_1024_1k world1_1080p_256 thing_glass_2048_512 cornell_full_512_64 soup_10m_4k_256"}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
names=""
for wl in $wls; do
    case $wl in
        cornell_plane_1024_1k) spp=256;; world1_1080p_256) spp=64;; thing_glass_2048_512) spp=32;;
        cornell_full_512_64) spp=256;; soup_10m_4k_256) spp=8;; *) spp=16;;
    esac
    name=${wl}_$build${TRAVERSAL:+_$TRAVERSAL}
    timeout -k 10 600 python3 -u bench.py --no-cpu --no-reference-leg --steps 1 --warmup 1 --workload $wl --spp $spp --build $build ${TRAVERSAL:+--traversal $TRAVERSAL} > $out/${name}_plain.json 2> $out/${name}_plain.err \
        || { tail -5 $out/${name}_plain.err; exit 1; }
    form=$(python3 -c "import json,sys; print(json.loads(open('$out/${name}_plain.json').read().strip().splitlines()[-1])['config']['kernel_form'])")
    tree=$(python3 -c "import json,sys; print(json.loads(open('$out/${name}_plain.json').read().strip().splitlines()[-1])['config']['tree'])")
    case $form in
        sweep) pin="--traversal sweep"; envs="RT_NO_AUTOTUNE=1";;
        walk_lds) pin="--traversal per_lane"; envs="RT_NO_AUTOTUNE=1";;
        walk_lds_refill) pin="--traversal per_lane"; envs="RT_REFILL=16";;
        walk) pin="--traversal per_lane"; envs="RT_REFILL=0";;
        walk_refill) pin="--traversal per_lane"; envs="RT_REFILL=16 RT_PRIM_WAIT=16";;
        ordered) pin="--traversal $form"; envs="RT_NO_AUTOTUNE=1";;
        wide) pin="--traversal wide"; envs="RT_REFILL=0";;
        wide_refill) pin="--traversal wide"; envs="RT_REFILL=16";;
        octant) pin="--traversal octant"; envs="RT_REFILL=0";;
        octant_refill) pin="--traversal octant"; envs="RT_REFILL=16 RT_PRIM_WAIT=16";;
        *) echo "unexpected form $form"; exit 1;;
    esac
    for pass in sq f w st; do
        staged=""
        case $pass in
            sq) ctr="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_TA_BUSY_sum";;
            f) ctr="FETCH_SIZE";;
            w) ctr="WRITE_SIZE";;
            st) ctr="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_TA_BUSY_sum"; staged="--staged";;
        esac
        # (the variables go through export: nothing may sit between rocprofv3's -- and the program)
        ( export $envs; timeout -s KILL 400 rocprofv3 --pmc $ctr -d $out/${name}_$pass -o $pass -- python3 bench.py --no-cpu --no-reference-leg --steps 1 --warmup 0 \
            --workload $wl --spp $spp --build $tree $pin $staged > $out/${name}_$pass.json 2> $out/${name}_$pass.err ) || { tail -5 $out/${name}_$pass.err; exit 1; }
    done
    names="$names $name"
    echo "$name done: tree $tree, form $form"
done
python3 tools/workload_table.py $out $names | tee $out/table.md
rm -rf $out/*_sq $out/*_f $out/*_w $out/*_st          # the databases are large; the table, the records and the bench lines stay
