#!/bin/bash
# tools/ta_rate under the address-unit counters (on the GPU box, from the repository root): what
# TA_TA_BUSY reads when the vector memory path is the only thing a kernel uses.
#   bash tools/ta_rate_pmc.sh <out dir>
out=${1:-gpurun_out/ta_rate_pmc}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $out/p -o p --output-format csv -- ./tools/ta_rate > $out/ta_rate.txt 2> $out/p.err || { tail -3 $out/p.err; exit 1; }
python3 - $(find $out/p -name "*counter_collection.csv" | head -1) $out/ta_rate.txt <<'PY' | tee $out/ta_busy.txt
import csv, collections, sys
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
lines = [l.rstrip() for l in open(sys.argv[2]) if l.startswith(('dword', 'ds_b128'))]
ids = sorted(rows)
# two dispatches per printed line (the first warms up): take the second
for k, line in enumerate(lines):
    c = rows[ids[2 * k + 1]]
    cu_cycles = c['GRBM_GUI_ACTIVE'] / 8 * 256
    print(f"{line[:66]}  TA busy {c['TA_TA_BUSY_sum'] / cu_cycles:5.2f}  busy cycles / load {c['TA_TA_BUSY_sum'] / max(c['SQ_INSTS_VMEM_RD'], 1):6.1f}")
PY
rm -rf $out/p
