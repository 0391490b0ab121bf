#!/bin/bash
# What clock does the GPU sustain under the benchmark?  rocm-smi sampled every 0.25 s while bench.py runs (one gpurun call) -> gpurun_out/clock_under_load.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/clock_under_load.txt
: > $OUT
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -20 >> $OUT
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n')"; sleep 0.25; done ) > gpurun_out/clock_samples.txt &
SPID=$!
timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_clock.json 2> gpurun_out/bench_clock.err
kill $SPID
python - <<'PY' >> gpurun_out/clock_under_load.txt
import json,re
rows=[]
for ln in open('gpurun_out/clock_samples.txt'):
    m=re.match(r't=([\d.]+) (\{.*\})\s*$', ln)
    if not m: continue
    try: d=json.loads(m.group(2))
    except Exception: continue
    c=d.get('card0',{})
    def num(k):
        for kk,v in c.items():
            if k in kk.lower():
                mm=re.search(r'([\d.]+)', str(v))
                if mm: return float(mm.group(1))
        return None
    rows.append((float(m.group(1)), num('sclk'), num('mclk'), num('power')))
if rows:
    t0=rows[0][0]
    print("# t (s)   sclk (MHz)   mclk (MHz)   power (W)")
    for t,s,mc,p in rows: print(f"{t-t0:7.2f}   {s}   {mc}   {p}")
    busy=[r for r in rows if r[3] and r[3]>0.6*max(x[3] for x in rows if x[3])]
    if busy:
        import statistics as st
        print(f"# samples with power > 60 % of the maximum: {len(busy)}; sclk median {st.median([b[1] for b in busy if b[1]])} MHz, min {min(b[1] for b in busy if b[1])}, max {max(b[1] for b in busy if b[1])}; power median {st.median([b[3] for b in busy])} W")
d=json.load(open('gpurun_out/bench_clock.json'))
print('# bench:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, roofline frac', d['roofline']['frac'])
PY
tail -5 $OUT
