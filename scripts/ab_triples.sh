# same-box alternation: cfg3 with the host's choice vs forced numbers of triples among the gangs
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for cfg in "default" "32 0" "32 4" "32 8" "24 8"; do
    set -- $cfg
    if [ "$1" = "default" ]; then unset SMM_PAIRS SMM_TRIPLES; else export SMM_PAIRS=$1 SMM_TRIPLES=$2; fi
    timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); print('cfg3 gangs/triples $cfg:', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step dp', round(j['roofline']['kernel_ms'],3))"
  done
done
