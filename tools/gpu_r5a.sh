#!/bin/bash
# round 5, call A: suite + bench (with --pmc-traffic) + stream / process co-residency experiments
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --pmc-traffic > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], 'valid', d.get('valid'), 'selfcheck', d['selfcheck_max_abs'], 'traffic', r.get('traffic'), r.get('traffic_measured_in_run'), r.get('traffic_note'), r.get('traffic_vs_algorithmic'), 'fwd_vs_step', r.get('forward_events_vs_step'))
print({k:(v['launches'],v['ms']) for k,v in r['breakdown'].items()})
print('f16x2', d.get('f16x2_leg',{}).get('value'), 'cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','speedup','port_vs_reference','reference_estimate')})
PY
SECS=5 timeout 300 python tools/diag_concurrent_streams.py > gpurun_out/diag_streams.log 2>&1; tail -12 gpurun_out/diag_streams.log
# two PROCESSES, victim fir beside attn_h2<3,3> (the round-4 corruption, control), then the same with the processes on disjoint CU halves
SECS=5 PLAN="fir:1;attnD96:4" timeout 120 python tools/diag_concurrent_ops.py > gpurun_out/diag_2proc_control.log 2>&1; tail -4 gpurun_out/diag_2proc_control.log
cat > /tmp/masked.sh <<'SH'
#!/bin/bash
cd $1
HSA_CU_MASK="0:0-127" SECS=5 PLAN="fir:1;attnD96:4" python tools/diag_concurrent_ops.py worker p0of2 &
HSA_CU_MASK="0:128-255" SECS=5 PLAN="fir:1;attnD96:4" python tools/diag_concurrent_ops.py worker p1of2 &
wait
SH
timeout 120 bash /tmp/masked.sh $PWD > gpurun_out/diag_2proc_cumask.log 2>&1; tail -4 gpurun_out/diag_2proc_cumask.log
