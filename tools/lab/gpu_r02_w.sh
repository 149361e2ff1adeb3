#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02w
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd /tmp
for i in 1 2 3 4; do
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t$i -o b -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b$i.json 2> $OUT/b$i.err
  python3 - <<PY
import json,csv,glob
d=json.loads([l for l in open('$OUT/b$i.json') if l.startswith('{')][-1]); s=d['stage_us_per_step']
print('run $i', round(d['value'],1), 'gen us/step', round(s['primary_ray_gen']), 'ptrace', round(s['primary_trace']))
rows=[]
for f in glob.glob('$OUT/t$i/**/*kernel_trace.csv', recursive=True):
    rows+=list(csv.DictReader(open(f)))
cp=[]
for f in glob.glob('$OUT/t$i/**/*memory_copy_trace.csv', recursive=True):
    cp+=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void ','').replace('rt::','')[:44]) for r in rows]
ev+=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size',''))) for r in cp]
ev.sort()
t0=ev[0][0]
# the timed pass: the third-from... find raygen launches; print events around the last 'big' raygen (20 layers)
gens=[k for k,e in enumerate(ev) if e[2].startswith('k_raygen')]
big=[k for k in gens if (ev[k][1]-ev[k][0])>4e5]
k0=big[1] if len(big)>1 else big[-1]
prev_end=max(e[1] for e in ev[:k0-6])
for e in ev[k0-6:k0+6]:
    print(f"   {(e[0]-t0)/1e6:10.2f} ms  gap {(e[0]-prev_end)/1e6:8.2f}  dur {(e[1]-e[0])/1e6:8.2f}  {e[2]}")
    prev_end=max(prev_end,e[1])
PY
done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
