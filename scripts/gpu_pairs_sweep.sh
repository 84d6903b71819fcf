cd $GRAFT_REPO_ROOT
for n in auto 0 16 32 48 64 96; do
  if [ $n = auto ]; then unset SMM_PAIRS; else export SMM_PAIRS=$n; fi
  timeout -k 10 300 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('pairs $n', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3), 'mof', round(r['mof'],4))"
done
unset SMM_PAIRS
for w in cfg1 cfg2; do timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$w', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3), 'mof', round(r['mof'],4))"; done
