# round 4: the GPU suite through the shipped library, then bench lines (cfg3 default, cfg5 strong leg on one rank) of the shipped
# library and of variant libraries (BAND-only builds: scripts/build_variants.sh)     usage: bash scripts/gpu_r4_s.sh "<variant tags>"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4s_tests.txt 2>&1; rc=$?
tail -4 gpurun_out/r4s_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for tag in shipped $1; do
  if [ $tag = shipped ]; then unset SMM_LIB_PATH; else export SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$tag.so; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 2>gpurun_out/r4s_${tag}_cfg3.err | tail -1 > gpurun_out/r4s_${tag}_cfg3.json
  SMM_DIST_SINGLE_RANK=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --strong-leg --scaling strong --second-seed -1 2>gpurun_out/r4s_${tag}_cfg5.err | tail -1 > gpurun_out/r4s_${tag}_cfg5.json
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
for w in ('cfg3', 'cfg5'):
    try:
        r = json.load(open('gpurun_out/r4s_%s_%s.json' % (t, w)))
        rf = r['roofline']
        print(t, w, round(r['value'] / 1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step; DP launch mean', round(rf.get('kernel_ms', 0), 3),
              'critical', rf.get('critical_launch_ms'), 'rest', rf.get('rest_launch_ms'), 'mof', round(r.get('mof', 0), 5))
    except Exception as e:
        print(t, w, 'failed', e)
PY
done
