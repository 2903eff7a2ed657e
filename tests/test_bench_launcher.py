"""bench.py --gpus N must start N ranks itself.  CPU check: the same launcher,
process group, barrier / max-over-ranks timing and pose gather, with a CPU stand-in step over gloo
(``--standin``); the JSON line must report n_gpus == 2 and one entry per rank."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_self_launches_two_ranks():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--standin',
              '--min-seconds', '0.2', '--min-warmup-seconds', '0.05'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # rank 0 prints exactly one JSON line
    out = json.loads(lines[0])
    assert out['standin'] is True and 'SELF-TEST' in out['metric']
    assert out['n_gpus'] == 2 and out['collective_world_size'] == 2
    assert [e['rank'] for e in out['ranks']] == [0, 1]
    assert len({e['pid'] for e in out['ranks']}) == 2          # two processes
    assert out['steps'] == 2 and out['config']['global_batch'] == 8
    assert out['timing']['blocks'] >= 1 and out['timing']['timed_seconds'] >= 0.2
    assert out['scaling'] == 'weak' and out['value'] > 0


def test_bench_refuses_mismatched_world():
    # a torchrun-style environment whose WORLD_SIZE disagrees with --gpus must not produce a line
    r = _run(['--gpus', '2', '--standin', '--steps', '1', '--warmup', '0'],
             env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_refuses_without_gpus():
    # no GPU in the CPU container: --gpus 2 (real step) must refuse instead of running 1 rank
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode == 2 and 'refusing' in r.stderr


def test_counter_summaries_are_bound_to_the_kernel_sources():
    """bench.py attaches profiles/lookup_pmc.json / *_mfma_pmc.json (separate rocprofv3 --pmc passes) only when they
    record the hashes of today's kernel sources: no hashes -> stale; one changed file -> stale, named."""
    sys.path.insert(0, ROOT)
    import bench
    now = bench.kernel_source_hashes(bench.LOOKUP_SOURCES)
    assert set(now) == set(bench.LOOKUP_SOURCES) and all(v and len(v) == 16 for v in now.values())
    assert bench._fresh({'kernel_source_hashes': dict(now)}, bench.LOOKUP_SOURCES) == (True, None)
    ok, why = bench._fresh({'traffic_bytes_per_launch': 1}, bench.LOOKUP_SOURCES)
    assert not ok and 'no kernel_source_hashes' in why
    other = dict(now, **{'corr_lookup.hip': '0' * 16})
    ok, why = bench._fresh({'kernel_source_hashes': other}, bench.LOOKUP_SOURCES)
    assert not ok and 'corr_lookup.hip' in why
    # whatever is committed under profiles/ is either fresh or refused -- never silently attached
    p = os.path.join(ROOT, 'profiles', 'lookup_pmc.json')
    if os.path.exists(p):
        d = json.load(open(p))
        ok, why = bench._fresh(d, bench.LOOKUP_SOURCES)
        assert ok or why
