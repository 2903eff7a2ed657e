"""Turn the two rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE) into
profiles/lookup_pmc.json: HBM bytes per corr-lookup launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE
reports half of the bytes of a coalesced read stream -> doubled; the factor is re-checked here on
instance_norm_kernel<16> (plain variant), whose traffic is exactly one read + one write of its
tensor."""
import csv, json, re, sys, collections

fetch_csv, write_csv, out_json, tag = sys.argv[1:5]
stats_csv = sys.argv[5] if len(sys.argv) > 5 else None      # --kernel-trace --stats pass (no PMC)


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')].append(float(r['Counter_Value']))
    return agg


f = per_kernel(fetch_csv, 'FETCH_SIZE')
w = per_kernel(write_csv, 'WRITE_SIZE')
lk = [k for k in f if 'corr_lookup' in k][0]
inorm = [k for k in f if 'instance_norm_kernel<16>' in k][0]
cal_w = min(w[inorm]) * 1024                      # exact: one write of the tensor
cal_f = min(f[inorm]) * 1024                      # plain variant: one read of the same tensor
factor = cal_w / cal_f
fetch = sum(f[lk]) / len(f[lk]) * 1024 * 2
write = sum(w[lk]) / len(w[lk]) * 1024
out = {
    'kernel': lk, 'launches': len(f[lk]),
    'fetch_bytes_per_launch': round(fetch), 'write_bytes_per_launch': round(write),
    'traffic_bytes_per_launch': round(fetch + write),
    'algorithmic_bytes_per_launch': 2904 * 32 * 1024,
    'fetch_size_calibration': {'kernel': inorm, 'write_bytes': cal_w, 'raw_fetch_bytes': cal_f,
                               'bytes_per_reported_byte': round(factor, 3)},
    'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py batch 32, '
              f'{tag}; KiB units, FETCH_SIZE x2 (gfx950 correction, calibration factor measured '
              f'{factor:.2f})',
}
if stats_csv:
    for r in csv.DictReader(open(stats_csv)):
        if 'corr_lookup' in r['Name']:
            out['rocprof_kernel_trace'] = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 2),
                                           'min_us': round(float(r['MinNs']) / 1e3, 2),
                                           'max_us': round(float(r['MaxNs']) / 1e3, 2)}
json.dump(out, open(out_json, 'w'), indent=1)
print(json.dumps(out, indent=1))
