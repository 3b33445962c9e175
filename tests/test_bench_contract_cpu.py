"""bench.py's output contract, checked without a GPU: the defaults of the command line and the shape of the JSON line (on the
lines kept under profiles/ -- they are bench.py's own output on the MI355X)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {'metric': str, 'value': float, 'unit': str, 'n_gpus': int, 'steps': int, 'warmup': int, 'ms_per_step': float,
            'higher_is_better': bool, 'scaling': str, 'dtype': str, 'data': str, 'config': dict, 'roofline': dict}


def test_defaults_are_one_gpu_and_a_run_of_minutes(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse()
    assert bench.METRIC == json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    assert a.gpus == 1 and 1 <= a.steps <= 50 and a.warmup >= 1 and a.dtype == 'fp32' and a.points == 100_000
    assert not a.no_prefetch and not a.no_optimizer and not a.no_cpu_baseline


def test_kept_bench_lines_follow_the_contract():
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round2_final_bench_*.json')))
    assert files, 'no bench line kept under profiles/'
    for f in files:
        d = json.load(open(f))
        for k, t in REQUIRED.items():
            assert k in d and isinstance(d[k], t), (f, k)
        assert 'vs_baseline' in d and d['vs_baseline'] is None            # BASELINE.md publishes no number for this metric
        # (lines written before the metric string carried BASELINE.json's ", 1/2/4/8 MI355X" suffix are kept as they were printed)
        assert d['metric'] in (base['metric'], base['metric'].rsplit(',', 1)[0])
        assert d['unit'] == 'scenes/s' and d['higher_is_better'] and d['scaling'] == 'weak'
        assert abs(d['value'] - d['config']['global_batch'] / d['ms_per_step'] * 1e3) < 1e-6 * d['value']
        assert 'workload' in d['config'] and 'model' not in d['config']
        r = d['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in r, (f, k)
        assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
        assert d['dtype'] in ('f32', 'bf16') and r['peak'] == (157.3 if d['dtype'] == 'f32' else 2500.0)
        if 'cpu_baseline' in d and d['cpu_baseline']:
            c = d['cpu_baseline']
            assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    main = json.load(open(os.path.join(ROOT, 'profiles', 'round2_final_bench_fp32.json')))
    assert main['cpu_baseline'] and main['n_gpus'] == 1 and main['config']['global_batch'] == 8
