"""Copy the evidence of one `tools/gpu_suite.sh <tag> ... bench profiles` run from gpurun_out/ (scratch) into profiles/
(tracked): bench line, per-kernel stats, the lookup / correlation rows of the FETCH / WRITE passes, the SQ summaries and
the raw SQ rows of the Winograd dispatches, the co-issue lab output.   python tools/publish_profiles.py <tag>"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P, D = os.path.join(R, 'gpurun_out', tag), os.path.join(R, 'gpurun_out', 'profiles_' + tag), os.path.join(R, 'profiles')


def first(pattern):
    m = sorted(glob.glob(pattern, recursive=True))
    return m[0] if m else None


def copy(src, name):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(D, name)); print('  ', name)


def rows(src, name, keep):
    if not src:
        return
    with open(src) as f, open(os.path.join(D, name), 'w', newline='') as o:
        r = csv.DictReader(f); w = csv.DictWriter(o, r.fieldnames); w.writeheader()
        n = 0
        for row in r:
            if keep(row, n):
                w.writerow(row); n += 1
    print('  ', name, n, 'rows')


line = open(os.path.join(G, 'bench.json')).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(D, tag + '_bench.json'), 'w').write(line + '\n'); print('  ', tag + '_bench.json')
copy(first(P + '/stats_f32/**/*kernel_stats.csv'), tag + '_bench_f32_kernel_stats.csv')
copy(first(P + '/stats_f16x3/**/*kernel_stats.csv'), tag + '_bench_f16x3_kernel_stats.csv')
copy(first(os.path.join(R, 'gpurun_out', tag + '_b32') + '/**/*kernel_stats.csv'), tag + '_batch32_only_kernel_stats.csv')
lk = lambda row, n: 'corr_lookup' in row['Kernel_Name'] or 'corr_gemm' in row['Kernel_Name']
rows(first(P + '/pmc_fetch/**/*counter_collection.csv'), tag + '_pmc_fetch_lookup_and_corr_build.csv', lk)
rows(first(P + '/pmc_write/**/*counter_collection.csv'), tag + '_pmc_write_lookup_and_corr_build.csv', lk)
rows(first(P + '/pmc_mfma/**/*counter_collection.csv'), tag + '_pmc_mfma_conv_wino_dispatches.csv',
     lambda row, n: 'conv_wino' in row['Kernel_Name'] and n < 7 * 96)
copy(os.path.join(G, 'mfma_pmc.json'), tag + '_mfma_pmc.json')
copy(os.path.join(G, 'lookup_pmc.json'), 'lookup_pmc.json')
copy(os.path.join(G, 'coissue.txt'), tag + '_coissue_mfma_vs_other_instructions.txt')
