#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_sf -o sf -- python $R/tools/diag_stem_final.py > $R/gpurun_out/diag_stem_final.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_sf/**/*kernel_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        n=r['Kernel_Name']
        if 'conv' in n and 'pack' not in n:
            print(f"{n[:70]:70s} grid {r['Grid_Size']:>8s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
