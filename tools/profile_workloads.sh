#!/bin/bash
# PMC passes of the per-batch kernel for every benchmark workload (run on the GPU box from the
# repository root): three rocprofv3 --pmc runs per workload (SQ counters; FETCH_SIZE; WRITE_SIZE --
# separate runs, the counter blocks do not all fit one), then tools/workload_table.py.
#   bash tools/profile_workloads.sh <out dir> [build]      # build: reference (default) | sah
set -e
out=${1:-gpurun_out/workloads}
build=${2:-reference}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
names=""
run() {   # name workload spp [extra bench.py arguments]
    local name=$1 wl=$2 spp=$3 more=$4
    for pass in sq f w; do
        case $pass in
            sq) ctr="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU";;
            f) ctr="FETCH_SIZE";;
            w) ctr="WRITE_SIZE";;
        esac
        timeout -s KILL 300 rocprofv3 --pmc $ctr -d $out/${name}_$pass -o $pass -- python3 bench.py --no-cpu --steps 1 --warmup 0 \
            --workload $wl --spp $spp --build $build $more > $out/${name}_$pass.json 2> $out/${name}_$pass.err || { tail -5 $out/${name}_$pass.err; exit 1; }
    done
    names="$names $name"
    echo "$name done"
}
# (the profiler serialises kernels, and a scene's first render times its candidate forms with two
# kernels in flight: pin the form the unprofiled run takes)
if [ "$build" = reference ]; then run c3_cornell_plane cornell_plane_1024_1k 256 "--traversal sweep"; else run c3_cornell_plane cornell_plane_1024_1k 256; fi
run c2_world1 world1_1080p_256 64
run c4_thing_glass thing_glass_2048_512 32
run c1_cornell_full cornell_full_512_64 256
RT_SOUP_TRIS=${RT_SOUP_TRIS:-10000000} run c5_soup soup_10m_4k_256 4
python3 tools/workload_table.py $out $names | tee $out/table.md
rm -rf $out/*_sq $out/*_f $out/*_w          # the databases are large; the table and the bench lines stay
