#!/bin/bash
# Runs tools/fetch_calib under the HBM byte counters (on the GPU box, from the repository root):
#   bash tools/fetch_calib.sh <out dir>
out=${1:-gpurun_out/fetch_calib}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
timeout -k 10 120 ./tools/fetch_calib > $out/plain.txt 2>&1 || { cat $out/plain.txt; exit 1; }
cat $out/plain.txt
n=0
for ctr in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "WRITE_SIZE"; do
    n=$((n + 1))
    timeout -s KILL 120 rocprofv3 --pmc $ctr -d $out/p$n -o p --output-format csv -- ./tools/fetch_calib > $out/p$n.txt 2> $out/p$n.err || { echo "pass $n ($ctr) failed"; tail -3 $out/p$n.err; continue; }
    python3 - $(find $out/p$n -name "*counter_collection.csv" | head -1) <<'PY' | tee $out/pmc_p$n.txt
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    for c, x in sorted(v.items()):
        print('%-12s %-28s %.6g' % (k, c, x))
PY
    rm -rf $out/p$n
done
