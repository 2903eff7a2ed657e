"""r6: the timeline of ONE batch-N pass out of a rocprofv3 --kernel-trace CSV (steady_trace.py window): every launch with its
start offset, duration, queue and the idle time of the device before it -- where the critical path of configs[1] really is.
    python tools/lab/b1_timeline.py <kernel_trace.csv> [max rows printed]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''), r.get('Queue_Id', '?'))
             for r in rows), key=lambda e: e[0])
# one pass = from a corr_gemm launch back to the first stem launch before it ... simpler: between two consecutive
# conv_taps stems of the feature encoder (first launch of a pass)
# a replay starts with the graph wrapper's input copies (__amd_rocclr_copyBuffer x 7): pass = from the first copy of one replay to
# the first copy of the next
starts = [i for i, e in enumerate(ev) if 'copyBuffer' in e[2] and (i == 0 or 'copyBuffer' not in ev[i - 1][2])]
if len(starts) < 4:
    sys.exit('trace too short')
a, b = starts[1], starts[2]
one = ev[a:b]
t0 = one[0][0]
print(f'pass: {len(one)} launches, {(one[-1][1] - t0) / 1e3:.1f} us from first start to last end; next pass starts {(ev[b][0] - t0) / 1e3:.1f} us after')
busy_end = t0
idle = 0.0
agg = {}
for i, (s, e, name, q) in enumerate(one):
    gap = s - busy_end
    if gap > 0:
        idle += gap
    if i < lim:
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  q{q:>3s}  idle-before {max(gap, 0) / 1e3:5.1f}  {name[:70]}')
    busy_end = max(busy_end, e)
    k = name[:60]
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
print(f'device idle inside the pass (no kernel running): {idle / 1e3:.1f} us')
# union of busy time per queue
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{v[1]:8.1f} us  x{v[0]:4d}  {k}')
