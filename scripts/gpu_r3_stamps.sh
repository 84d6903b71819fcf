# profiles/round3_band_stamps.txt: the first BAND kernel of round 3 (blocks of 4 positions, commit 1f784c9: b4 / profb4) against the
# shipped one on ONE box: kernel time, block stamps of the workgroup of cfg3's longest video, who reaches the barrier last,
# and the CrossTask-like lattices by state count.  Variants: scripts/build_variants.sh (+ the b4 build described in profiles/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( echo "== DP kernel on the bench's cfg3 corpus (seed 2): HIP events, 5 launches each; b4 = round 3's first BAND kernel (blocks of 4 positions)"
  timeout -k 10 300 python scripts/prof_cfg3.py b4 base b4 base
  echo "== block stamps (-DSMM_PROFILE builds: cycles per hand-over block, workgroup 0 = the launch's longest video; a block is 4 positions in profb4, 8 in profbase)"
  timeout -k 10 300 python scripts/prof_cfg3.py profb4 profbase
  echo "== -DSMM_PROFILE=2: how often each wave is the one the barrier waits for (waited < 150 cycles itself)"
  SMM_PROF_LAST=1 timeout -k 10 300 python scripts/prof_cfg3.py prof2
  echo "== CrossTask-like lattices, 64 videos x 4096 frames, by state count (scripts/time_variants.py)"
  timeout -k 10 300 python scripts/time_variants.py b4 base ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_band_stamps.txt
echo rc=$?
