"""Fold the FETCH_SIZE calibration runs (scripts/gpu_r5_final.sh) into <root>/pmc_calibration.json:
  <root>/pmc_calib_ubench   rocprofv3 --pmc FETCH_SIZE of scripts/ubench/_bin/fetch_calib (three access shapes, 1 GiB each)
  <root>/pmc_calib_emission rocprofv3 --pmc FETCH_SIZE of scripts/probe_emission_alone.py (ops.emission on cfg3: 4 D frames of x)
factor = known bytes / (FETCH_SIZE x 1024)."""
import csv, glob, json, os, re, sys

root = sys.argv[1]
out = {}


def mean_fetch(sub, kernel_substr):
    vals = []
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row.get('Kernel_Name', '') and row.get('Counter_Name') == 'FETCH_SIZE':
                vals.append(float(row['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


log = open(os.path.join(root, 'pmc_calib_ubench.log')).read()
for shape in ('b16_coalesced', 'b16_rowpieces', 'b8_coalesced'):
    m = re.search(r'calib_%s bytes (\d+)' % shape, log)
    fetch, n = mean_fetch('pmc_calib_ubench', 'calib_' + shape)
    if m and fetch:
        out[shape] = dict(known_bytes=int(m.group(1)), FETCH_SIZE_KiB=fetch, launches=n, factor=int(m.group(1)) / (fetch * 1024.0),
                          source='scripts/ubench/fetch_calib.hip: 1 GiB read once')
log = open(os.path.join(root, 'pmc_calib_emission.log')).read()
m = re.search(r'emission_alone frames (\d+) x_bytes (\d+)', log)
fetch, n = mean_fetch('pmc_calib_emission', 'smm_emission')
if m and fetch:
    out['emission_alone'] = dict(known_bytes=int(m.group(2)), FETCH_SIZE_KiB=fetch, launches=n, factor=int(m.group(2)) / (fetch * 1024.0),
                                 source='ops.emission alone on cfg3 (x = %d bytes = 7.7 x the Infinity Cache); known bytes = 4 D frames, '
                                        'the weights a workgroup re-reads from L2 not counted' % int(m.group(2)))
json.dump(out, open(os.path.join(root, 'pmc_calibration.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
