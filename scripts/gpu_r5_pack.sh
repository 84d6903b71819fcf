# cfg3 (CU-time-bound): does a better packing of the launch pay?  time split forced at several unit sizes, with and without the
# four-wave workgroups for <= 16-state videos.  One box, same command each.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_pack.txt
: > $O
run() {
  echo "== $*" >> $O
  env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('ms_per_step %.3f crit %s rest %s small %s split %s mismatches %s' % (r['ms_per_step'], rf.get('critical_launch_ms'), rf.get('rest_launch_ms'), rf.get('small_wg_launch_ms'), r.get('time_split',{}).get('videos_split'), r['parity']['label_mismatches']))" >> $O 2>&1
  tail -1 $O
}
run SMM_X=0
run SMM_CHUNK_P=7000
run SMM_CHUNK_P=7000 SMM_SMALL_WG=2
run SMM_SMALL_WG=2
run SMM_CHUNK_P=6000 SMM_SMALL_WG=2
run SMM_CHUNK_P=8000 SMM_SMALL_WG=2
run SMM_X=0
