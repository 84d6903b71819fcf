"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into profiles/pmc_summary.json (per kernel, per launch).

FETCH_SIZE is scaled per kernel by a CALIBRATED factor (profiles/pmc_calibration.json: known bytes / (FETCH_SIZE x 1024) of the
kernel's access shape, measured with scripts/ubench/fetch_calib.hip and -- the emission kernel -- with ops.emission alone on a
corpus of 7.7 x the Infinity Cache, scripts/probe_emission_alone.py); a kernel without a calibration gets the guide's factor for
wide coalesced streaming reads, 2.0, and says so.  WRITE_SIZE is taken as it is (the guide: exact for streaming stores)."""
import csv, glob, json, os, sys

root, workload = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
calib_path = os.path.join(root, 'pmc_calibration.json')
if not os.path.exists(calib_path):
    calib_path = os.path.join(here, 'profiles', 'pmc_calibration.json')
calib = json.load(open(calib_path)) if os.path.exists(calib_path) else {}
# which calibrated shape each kernel's reads have
# (the emission kernels: the SHAPE's factor, 16-byte loads in 64-byte row pieces -- not the factor of the kernel's own known-bytes run,
# which would make its traffic equal its algorithmic bytes by construction; that run is the cross-check in pmc_calibration.json)
SHAPE = {'smm_viterbi_kernel': 'b8_coalesced', 'smm_emission_pair_kernel': 'b16_rowpieces', 'smm_emission_kernel': 'b16_rowpieces',
         'smm_class_sums_kernel': 'b16_coalesced', 'smm_chunk_stitch_kernel': 'b8_coalesced', 'smm_cum_anchor_kernel': 'b8_coalesced'}
res = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, 'pmc_%s' % ctr, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')
            if 'smm_' not in name or row.get('Counter_Name') != ctr:
                continue
            key = name.split('(')[0].replace('void ', '')
            # the repair launch of a time-split decode (TAG = 1: the last template argument) is a kernel of its own
            if key.startswith('smm_viterbi_kernel<') and key.rstrip('>').rstrip().endswith(', 1'):
                key = 'smm_viterbi_kernel_repair_launch'
            else:
                key = key.split('<')[0]
            res.setdefault(key, {}).setdefault(ctr, []).append(float(row['Counter_Value']))
out = {}
for k, v in res.items():
    fetch = sum(v.get('FETCH_SIZE', [0])) / max(1, len(v.get('FETCH_SIZE', [])))
    write = sum(v.get('WRITE_SIZE', [0])) / max(1, len(v.get('WRITE_SIZE', [])))
    shape = SHAPE.get(k)
    c = calib.get(shape) if shape else None
    factor = c['factor'] if c else 2.0
    out[k] = dict(FETCH_SIZE_KiB_per_launch=fetch, WRITE_SIZE_KiB_per_launch=write, fetch_factor=factor,
                  fetch_factor_source=("calibrated: %s (%s)" % (shape, c['source'])) if c else
                  "uncalibrated: the guide's factor for 16-B-per-lane coalesced streaming reads",
                  hbm_bytes_per_launch=(factor * fetch + write) * 1024.0)
path = os.path.join(here, 'profiles', 'pmc_summary.json')
allr = json.load(open(path)) if os.path.exists(path) else {}
allr[workload] = dict(out)
if len(sys.argv) > 3:          # provenance label (the caller knows the commit; the GPU box has no .git)
    allr[workload]['_source'] = sys.argv[3]
if 'smm_viterbi_kernel' in out:
    allr[workload]['smm_viterbi_kernel_hbm_bytes_per_launch'] = out['smm_viterbi_kernel']['hbm_bytes_per_launch']
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(allr, open(os.path.join(root, 'pmc_summary.json'), 'w'), indent=1)
print(json.dumps(allr, indent=1))
