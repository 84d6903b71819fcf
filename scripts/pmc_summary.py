"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into profiles/pmc_summary.json (per kernel, per launch)."""
import csv, glob, json, os, sys

root, workload = sys.argv[1], sys.argv[2]
res = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, 'pmc_%s' % ctr, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')
            if 'smm_' not in name or row.get('Counter_Name') != ctr:
                continue
            key = name.split('(')[0].replace('void ', '')
            # (rounds 1-3: the recovery launches behind a gang launch were rows of their own; round 4 has neither)
            key = key.split('<')[0]
            res.setdefault(key, {}).setdefault(ctr, []).append(float(row['Counter_Value']))
out = {}
for k, v in res.items():
    fetch = sum(v.get('FETCH_SIZE', [0])) / max(1, len(v.get('FETCH_SIZE', [])))
    write = sum(v.get('WRITE_SIZE', [0])) / max(1, len(v.get('WRITE_SIZE', [])))
    out[k] = dict(FETCH_SIZE_KiB_per_launch=fetch, WRITE_SIZE_KiB_per_launch=write,
                  hbm_bytes_per_launch=(2.0 * fetch + write) * 1024.0,
                  note="FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), KiB -> bytes")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_summary.json')
allr = json.load(open(path)) if os.path.exists(path) else {}
allr[workload] = dict(out)
if len(sys.argv) > 3:          # provenance label (the caller knows the commit; the GPU box has no .git)
    allr[workload]['_source'] = sys.argv[3]
if 'smm_viterbi_kernel' in out:
    allr[workload]['smm_viterbi_kernel_hbm_bytes_per_launch'] = out['smm_viterbi_kernel']['hbm_bytes_per_launch']
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(allr, open(os.path.join(root, 'pmc_summary.json'), 'w'), indent=1)
print(json.dumps(allr, indent=1))
