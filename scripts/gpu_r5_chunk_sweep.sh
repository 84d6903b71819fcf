# round 5: the time-split decode's unit size and warm-up on cfg3 / cfg1, ONE box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # workload, env...
  w=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w $*', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step; crit', round(r['critical_launch_ms'] or 0, 3), 'rest', r['rest_launch_ms'] and round(r['rest_launch_ms'], 3), j.get('time_split'), 'mismatch', j['parity'].get('label_mismatches'))"
}
run cfg1 SMM_CHUNK=0
run cfg1 SMM_CHUNK=1
run cfg1 SMM_CHUNK=1 SMM_CHUNK_WC=1024
run cfg3 SMM_CHUNK=0
for wc in 512 1024; do
  for p in 0 5000 7000 9000; do
    run cfg3 SMM_CHUNK=1 SMM_CHUNK_WC=$wc SMM_CHUNK_P=$p
  done
done
run cfg3 SMM_CHUNK=0
