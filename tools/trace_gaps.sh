#!/bin/bash
# kernel timeline of one small prove: kernel durations and the gaps between them (host round trips + launch latency)
#   gpurun -- 'bash tools/trace_gaps.sh r03 20'      (log_n; stderr kept under gpurun_out/)
set -o pipefail
TAG=${1:-r03}
LOGN=${2:-20}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_gaps_$LOGN
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_gaps_$LOGN -- python3 bench.py --no-live-traffic --log-n $LOGN --no-absorb --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_gaps_$LOGN.log 2>&1 || { echo "exit $?"; tail -30 $OUT/${TAG}_gaps_$LOGN.log; exit 1; }
python3 - <<PY > $OUT/${TAG}_gaps_$LOGN.txt
import csv,glob,re
f=glob.glob('$OUT/${TAG}_gaps_$LOGN/**/*_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
def short(n):
    m=re.match(r'(?:void )?(?:zkk::)?([A-Za-z0-9_]+)',n); return m.group(1) if m else n[:30]
# the timed steps: take the last 20 proves' worth: find repeating pattern starting at k_block_sums*
names=[short(r['Kernel_Name']) for r in rows]
starts=[i for i,n in enumerate(names) if n.startswith('k_block_sums')]
# use proves 5..24 (skip generate/warmup)
sel=starts[-21:-1]
import collections
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for a,b in zip(sel[:-1],sel[1:]):
    for i in range(a,b):
        r=rows[i]; dur[(i-a,names[i])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        if i+1<=b: gap[(i-a,names[i])].append((int(rows[i+1]['Start_Timestamp'])-int(r['End_Timestamp']))/1e3)
tot=0
print('# kernel timeline of one 2^$LOGN prove (mean over 19 proves, us): duration, then gap to the next kernel start')
for k in sorted(dur):
    d=sum(dur[k])/len(dur[k]); g=sum(gap[k])/len(gap[k]); tot+=d+g
    print(f'{k[0]:2d} {k[1]:28s} dur {d:8.1f}  gap {g:8.1f}')
print('sum', round(tot,1))
PY
cat $OUT/${TAG}_gaps_$LOGN.txt
