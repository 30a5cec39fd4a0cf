# lib.sh -- shared by the tools/*.sh sweeps (source it after setting LABEL).  Nothing a bench run prints is thrown away:
# stdout (the JSON line) goes to the caller, stderr is appended to gpurun_out/$LABEL.err together with the command line
# and its exit code, and a failing run shows the tail of that file.  (A GKR A/B run of round 2 printed nothing because its
# stderr went to /dev/null: DESIGN.md section 9.)
OUT=gpurun_out
mkdir -p $OUT
: "${LABEL:=$(basename "$0" .sh)}"
: > $OUT/$LABEL.err
bench_json() { # bench_json <seconds> <command ...>
    local t=$1; shift
    echo "+ $*" >> $OUT/$LABEL.err
    timeout -k 10 "$t" "$@" 2>> $OUT/$LABEL.err
    local rc=$?
    echo "exit $rc" >> $OUT/$LABEL.err
    if [ $rc -ne 0 ]; then
        echo "FAILED (exit $rc): $*  -- stderr kept in $OUT/$LABEL.err:" >&2
        tail -15 $OUT/$LABEL.err >&2
    fi
    return $rc
}
# summaries of one bench line on stdin (an empty stdin -- the run failed and said why above -- is not an error of its own)
summ_deg2() { python3 -c "
import json,sys
s=sys.stdin.read().strip()
if not s: sys.exit(1)
d=json.loads(s.splitlines()[-1]); k=d['steps']
print('prove_ms', round(d['ms_per_step'],4), {n:(v['launches'], round(v['total_ms']/k,4), round(v['achieved_GBs'],0)) for n,v in d['kernels'].items()})"; }
summ_gkr() { python3 -c "
import json,sys
s=sys.stdin.read().strip()
if not s: sys.exit(1)
d=json.loads(s.splitlines()[-1]); g=d.get('gkr', d)
print({k:round(g[k],4) for k in ('prove_ms','layers_ms_total','ms_per_round','circuit_evaluate_ms','output_absorb_ms','host_other_ms') if k in g})"; }
summ_mle() { python3 -c "
import json,sys
s=sys.stdin.read().strip()
if not s: sys.exit(1)
d=json.loads(s.splitlines()[-1])
print(round(d['ms_per_step'],4), {k:(v['launches'], round(v['total_ms'],3)) for k,v in d.get('kernels',{}).items()})"; }
