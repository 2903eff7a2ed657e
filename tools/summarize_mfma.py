"""MFMA utilisation of the matrix-core kernels from one rocprofv3 counter pass over bench.py
(tools/collect_profiles.sh: --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; own run, no other
trace domains).

    python tools/summarize_mfma.py <counter_collection.csv> <kernel_trace.csv> <out.json> <tag>

Per kernel (correlation GEMM, every conv_dma / conv_mfma instantiation): launches, mean duration,
counter sums per launch, and
  mfma_busy_vs_chip_peak = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)
      -- busy cycles of the matrix pipes against what the chip offers at its peak clock (the
         north star's "MFMA utilisation ... against chip peak"); SQ_VALU_MFMA_BUSY_CYCLES is summed
         over all SIMDs and counts 64 cycles per v_mfma_f32_32x32x2_f32
  mfma_busy_vs_active_clock = the same against GRBM_GUI_ACTIVE (the clock the kernel actually ran
         at under the profiler), when that counter is present
The fp32 MFMA peak (157.3 TFLOP/s) = 1024 SIMDs x 64 flop/clk x 2.4 GHz, so mfma_busy_vs_chip_peak
is directly the fraction of that peak the kernel's MFMA instructions occupy."""
import collections
import csv
import json
import re
import sys

pmc_csv, trace_csv, out_json, tag = sys.argv[1:5]
SIMDS, PEAK_HZ = 1024, 2.4e9


def short(name):
    name = re.sub(r'\(.*', '', name).replace('void ', '')
    return name


cnt = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(pmc_csv)):
    k = short(r['Kernel_Name'])
    if not any(t in k for t in ('corr_gemm', 'conv_dma', 'conv_mfma', 'conv_f16x3', 'conv_taps', 'conv_wino', 'fc_splitk')):
        continue
    cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key)
        calls[k] += 1
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace_csv)):
    k = short(r['Kernel_Name'])
    if k in cnt:
        dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9)

rows = []
for k, c in cnt.items():
    n = calls[k]
    d = sum(dur[k]) / max(len(dur[k]), 1)
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n
    row = {'kernel': k, 'launches': n, 'mean_duration_us': round(d * 1e6, 2),
           'total_us': round(sum(dur[k]) * 1e6, 1),
           'counters_per_launch': {a: round(v / n, 1) for a, v in sorted(c.items())},
           'mfma_busy_vs_chip_peak': round(busy / (SIMDS * d * PEAK_HZ), 4) if d else None}
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / n
    if gui:
        # GRBM_GUI_ACTIVE is reported per XCD and summed over the 8 XCDs
        row['active_clock_ghz'] = round(gui / 8 / d / 1e9, 3)
        row['mfma_busy_vs_active_clock'] = round(busy / (SIMDS * gui / 8), 4)
    rows.append(row)
rows.sort(key=lambda r: -r['total_us'])
tot = sum(r['total_us'] for r in rows)
busy_tot = sum(r['counters_per_launch'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) * r['launches'] for r in rows)
out = {'tag': tag, 'source': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE '
                             'SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY over bench.py (own pass)',
       'all_matrix_kernels': {'total_us': round(tot, 1),
                              'mfma_busy_vs_chip_peak': round(busy_tot / (SIMDS * tot * 1e-6 * PEAK_HZ), 4)},
       'kernels': rows}
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import bench as _bench      # kernel_source_hashes: bench.py attaches this summary only while these files are unchanged
out['kernel_source_hashes'] = _bench.kernel_source_hashes(_bench.CONV_SOURCES)
json.dump(out, open(out_json, 'w'), indent=1)
print(f"{'kernel':70s} {'n':>5s} {'mean us':>9s} {'MFMA busy / chip peak':>22s} {'/ active clk':>12s} {'clk GHz':>8s}")
for r in rows:
    print(f"{r['kernel'][:70]:70s} {r['launches']:5d} {r['mean_duration_us']:9.1f} {r['mfma_busy_vs_chip_peak']:22.3f} "
          f"{r.get('mfma_busy_vs_active_clock', float('nan')):12.3f} {r.get('active_clock_ghz', float('nan')):8.3f}")
print('all matrix kernels:', out['all_matrix_kernels'])
