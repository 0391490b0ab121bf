#!/bin/bash
# calibration: what SQ_VALU_MFMA_BUSY_CYCLES reads for a SATURATED matrix pipe (tools/ubench_mfma_chain: nothing but v_mfma_f32_32x32x16_bf16)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/pmc_chain -o chain -- $R/tools/bin/ubench_mfma_chain > $R/gpurun_out/ubench_mfma_chain.txt 2>&1
cd $R
cat gpurun_out/ubench_mfma_chain.txt | grep accumulators
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_chain/**/*counter_collection.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    d=collections.OrderedDict()
    for r in rows:
        key=(r['Dispatch_Id'], r['Kernel_Name'][:40], r['Workgroup_Size'])
        d.setdefault(key,{})[r['Counter_Name']]=float(r['Counter_Value'])
        d[key]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    for k,v in d.items():
        gui=v.get('GRBM_GUI_ACTIVE',0); mf=v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)
        print(k, 'dur %.1f us'%v['dur'], 'clk %.2f GHz'%(gui/8/v['dur']/1e3 if v['dur'] else 0), 'mfma_busy %.3f'%(mf/(gui/8*1024) if gui else 0), {n:x for n,x in v.items() if n not in('dur',)})
PY
