# same-box A/B of an environment switch of libsmmdp.so through bench.py: bash scripts/ab_env.sh VAR [workload]
cd $GRAFT_REPO_ROOT
V=$1; W=${2:-cfg3}
for rep in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $V=1; else unset $V; fi
    timeout -k 10 300 python bench.py --workload $W --steps 12 --warmup 3 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); print('$W $V=$on', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step dp', round(j['roofline']['kernel_ms'],3))"
  done
done
