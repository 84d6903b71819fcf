"""Development aid: DP kernel time of libsmmdp_<tag>.so variants (scripts/build_variants.sh) on the CrossTask-like lattices
of scripts/prof_band.py, all on one box.  usage: python scripts/time_variants.py base abl1 ..."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import prof_band

os.environ['SMM_BAND'] = '1'
shapes = ((64, 4096, 23), (64, 4096, 20), (64, 4096, 16), (64, 4096, 11))
for rep in range(2):
    for tag in sys.argv[1:]:
        print('==', tag, 'pass', rep); sys.stdout.flush()
        for b, t, c in shapes:
            prof_band.run(b, t, c, 1024, 'libsmmdp_%s.so' % tag)
