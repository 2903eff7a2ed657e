"""Turn the rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE; separate runs) into
profiles/lookup_pmc.json: HBM bytes per corr-lookup launch, at batch 32 (256x256) and for the
configs[4] launches (8 x 60 x 80 queries) of the same run.

Both counters are in KiB.  FETCH_SIZE under-reports by a pattern-dependent factor on gfx950
(MI355X_MICROARCH.md, HBM section: exactly 1/2 for wide coalesced reads, "other widths
uncalibrated"), so the factor is MEASURED on the lookup's own access type with
tools/lab/fetch_calib.hip (dword global_load_lds `nt` gathers and `sc1` dword stores over a 1 GiB
buffer with known byte counts):

    python tools/summarize_pmc.py <pmc_fetch.csv> <pmc_write.csv> <out.json> <tag> \
        [stats.csv] [calib_fetch.csv calib_write.csv]
"""
import collections
import csv
import json
import re
import sys

fetch_csv, write_csv, out_json, tag = sys.argv[1:5]
stats_csv = sys.argv[5] if len(sys.argv) > 5 else None      # --kernel-trace --stats pass (no PMC)
calib_f = sys.argv[6] if len(sys.argv) > 7 else None
calib_w = sys.argv[7] if len(sys.argv) > 7 else None


def rows(path, counter):
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            yield r


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in rows(path, counter):
        agg[re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')].append(float(r['Counter_Value']))
    return agg


GIB = float(1 << 30)
calibration = None
fetch_factor, write_factor = 2.0, 1.0      # the guide's figures, replaced by measured ones below
if calib_f and calib_w:
    cf, cw = per_kernel(calib_f, 'FETCH_SIZE'), per_kernel(calib_w, 'WRITE_SIZE')
    true_bytes = {'calib_f4_stream': GIB, 'calib_dma<4, 1, 1>': GIB, 'calib_dma<64, 1, 1>': GIB / 16,
                  'calib_dma<128, 1, 1>': GIB / 32, 'calib_dma<4, 10, 32>': GIB * 40 / 128}
    calibration = {}
    for k, tb in true_bytes.items():
        rep = min(cf[k]) * 1024
        calibration[k] = {'bytes_requested': tb, 'fetch_size_reported_bytes': rep,
                          'requested_per_reported': round(tb / rep, 3),
                          'reported_bytes_per_64B_sector_touched': None}
    # sectors touched: contig -> all; sector64 -> all 64-B sectors; line128 -> half of them;
    # 40 B of a 128-B line -> the first sector only
    sectors = {'calib_dma<4, 1, 1>': GIB / 64, 'calib_dma<64, 1, 1>': GIB / 64, 'calib_dma<128, 1, 1>': GIB / 128,
               'calib_dma<4, 10, 32>': GIB / 128, 'calib_f4_stream': GIB / 64}
    for k, ns in sectors.items():
        calibration[k]['reported_bytes_per_64B_sector_touched'] = round(calibration[k]['fetch_size_reported_bytes'] / ns, 2)
    for k in ('calib_store<0>', 'calib_store<1>'):
        rep = min(cw[k]) * 1024
        calibration[k] = {'bytes_written': GIB, 'write_size_reported_bytes': rep, 'written_per_reported': round(GIB / rep, 3)}
    # the lookup reads whole 64-byte sectors' worth of useful dwords per gather: its counting factor
    # is the one of the contiguous dword LDS-DMA pattern
    fetch_factor = calibration['calib_dma<4, 1, 1>']['requested_per_reported']
    write_factor = calibration['calib_store<1>']['written_per_reported']

# every lookup dispatch of the run, whatever instantiation took it (r5: the batch-32 launches run four groups per
# 1024-thread block, configs[4] one group per 256-thread block).  The two workloads are told apart by the launch's
# TOTAL thread count: 32 x 1024 queries x 4 waves x 64 lanes / 32 = 262144 at batch 32 (1024 x 256 or 256 x 1024).
MAIN_THREADS = 262144


def lookup_rows(path, counter):
    return [(float(r['Counter_Value']), int(r['Grid_Size']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
            for r in rows(path, counter) if 'corr_lookup' in r['Kernel_Name']]


fr, wr = lookup_rows(fetch_csv, 'FETCH_SIZE'), lookup_rows(write_csv, 'WRITE_SIZE')
fm, f4 = [v for v, g, _ in fr if g == MAIN_THREADS], [v for v, g, _ in fr if g != MAIN_THREADS]
wm, w4 = [v for v, g, _ in wr if g == MAIN_THREADS], [v for v, g, _ in wr if g != MAIN_THREADS]
lk = sorted({n for _, g, n in fr if g == MAIN_THREADS})[0]
fetch = sum(fm) / len(fm) * 1024 * fetch_factor
write = sum(wm) / len(wm) * 1024 * write_factor
out = {
    'kernel': lk, 'launches': len(fm),
    'fetch_bytes_per_launch': round(fetch), 'write_bytes_per_launch': round(write),
    'traffic_bytes_per_launch': round(fetch + write),
    'algorithmic_bytes_per_launch': 2904 * 32 * 1024,
    'fetch_counting_factor': fetch_factor, 'write_counting_factor': write_factor,
    'calibration': calibration,
    'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py batch 32, '
              f'{tag}; KiB units; counting factors measured on the kernel\'s own access types '
              f'(tools/lab/fetch_calib.hip): FETCH x{fetch_factor}, WRITE x{write_factor}',
}
if f4 and w4:
    out['config4'] = {'launches': len(f4),
                      'fetch_bytes_per_launch': round(sum(f4) / len(f4) * 1024 * fetch_factor),
                      'write_bytes_per_launch': round(sum(w4) / len(w4) * 1024 * write_factor),
                      'algorithmic_bytes_per_launch': 2904 * 8 * 60 * 80}
    out['config4']['traffic_bytes_per_launch'] = out['config4']['fetch_bytes_per_launch'] + out['config4']['write_bytes_per_launch']
if stats_csv:
    # a --kernel-trace per-dispatch CSV (preferred: separates the batch-32 launches from the configs[4]
    # ones by grid size) or a --stats summary CSV
    rd = list(csv.DictReader(open(stats_csv)))
    if rd and 'Start_Timestamp' in rd[0]:
        nthr = lambda r: int(r['Grid_Size_X'])
        d32 = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in rd
               if 'corr_lookup' in r['Kernel_Name'] and nthr(r) == MAIN_THREADS]
        d4 = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in rd
              if 'corr_lookup' in r['Kernel_Name'] and nthr(r) != MAIN_THREADS]
        if d32:
            out['rocprof_kernel_trace'] = {'calls': len(d32), 'avg_us': round(sum(d32) / len(d32), 2),
                                           'min_us': round(min(d32), 2), 'max_us': round(max(d32), 2),
                                           'note': 'batch-32 launches (262144 threads) of the traced bench.py run'}
        if d4 and 'config4' in out:
            out['config4']['rocprof_kernel_trace'] = {'calls': len(d4), 'avg_us': round(sum(d4) / len(d4), 2),
                                                      'min_us': round(min(d4), 2), 'max_us': round(max(d4), 2)}
    else:
        for r in rd:
            if 'corr_lookup' in r['Name']:
                out['rocprof_kernel_trace'] = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 2),
                                               'min_us': round(float(r['MinNs']) / 1e3, 2),
                                               'max_us': round(float(r['MaxNs']) / 1e3, 2),
                                               'note': 'all launches of the kernel in the traced run (batch-32 steps AND the configs[4] block)'}
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import bench as _bench      # kernel_source_hashes: bench.py attaches this summary only while these files are unchanged
out['kernel_source_hashes'] = _bench.kernel_source_hashes(_bench.LOOKUP_SOURCES)
json.dump(out, open(out_json, 'w'), indent=1)
print(json.dumps(out, indent=1))
