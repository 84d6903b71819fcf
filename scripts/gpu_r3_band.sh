set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_fullsize.py tests/test_gpu_random_sweeps.py -q -x -k "not log_partition" > gpurun_out/r3e_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3e_pytest.log
(cd action-segmentation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DSMM_PROFILE -DSMM_DEV_R=16 -mllvm -pragma-unroll-threshold=1048576 -mllvm -unroll-threshold=1048576 -o ../libsmmdp_prof16.so smm_api.hip smm_emission.hip smm_viterbi.hip smm_logz.hip smm_logz_bwd.hip smm_dense.hip smm_eval.hip smm_fit.hip smm_tables.hip)
SMM_ONLY_BAND=1 timeout -k 10 300 python scripts/prof_band.py prof plain > gpurun_out/r3e_band_stamps.txt 2>&1; echo rc=$?
grep -B1 -A9 "SMM_BAND 1" gpurun_out/r3e_band_stamps.txt | head -80
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3e_cfg3.err | tail -1 > gpurun_out/r3e_cfg3.json
timeout -k 10 300 python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3e_cfg1.err | tail -1 > gpurun_out/r3e_cfg1.json
python - <<'PY'
import json
for w in ('cfg3', 'cfg1'):
    try:
        r = json.load(open('gpurun_out/r3e_%s.json' % w))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'mof', r['mof'], 'other', r.get('other_draw', {}).get('dp_kernel_ms'))
    except Exception as e:
        print(w, 'failed', e)
PY
