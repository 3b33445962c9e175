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
    assert not a.no_prefetch and not a.no_optimizer and not a.no_cpu_baseline and not a.no_cfg3 and not a.no_mfma_line and a.optimizer == 'adamw'
    assert not a.no_extra_configs and a.config == 'cfg2'          # the default line carries cfg2 (headline), native MFMAs, cfg3, cfg4, cfg5


def test_gpus_n_turns_itself_into_a_torchrun_launch(monkeypatch):
    """`python bench.py --gpus N` (how the driver invokes it) must not need an outer torchrun: with WORLD_SIZE unset it execs
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`; inside a
    torchrun environment (WORLD_SIZE set) and for N = 1 it does nothing."""
    sys.path.insert(0, ROOT)
    import bench
    argv = ['--gpus', '4', '--steps', '3', '--warmup', '1']
    cmd = bench.launch_command(4, argv, 12345)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '12345'
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, 'bench.py') and cmd[-len(argv):] == argv
    seen = {}
    monkeypatch.setattr(os, 'execvpe', lambda f, a, e: seen.update(file=f, args=a, env=e))
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench.self_launch(bench.parse())
    assert seen['args'][-len(argv):] == argv and '--nproc-per-node=4' in seen['args'] and seen['file'] == sys.executable
    seen.clear()
    monkeypatch.setenv('WORLD_SIZE', '4')
    bench.self_launch(bench.parse())                       # already inside torchrun
    assert not seen
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    bench.self_launch(bench.parse())                       # N = 1
    assert not seen


def test_pmc_traffic_is_refused_on_other_kernel_sources(tmp_path, monkeypatch):
    """a PMC summary is quoted only when the hashes of EVERY kernel source match the tree (the file describes all
    kernels of the step, not only the convolutions), it names the commit it was taken at, and the cfg3 block quotes the cfg3 pass."""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.csrc_hashes()
    assert set(bench.PMC_SOURCES) == set(h) and all(len(v) == 16 for v in h.values())
    assert {'spconv.hip', 'spconv_wg.hip', 'gemm.hip', 'attn_x3.hip', 'bn.hip', 'radix.hip', 'u3d_common.h'} <= set(bench.PMC_SOURCES)
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'csrc_hashes', lambda names=None: {n: h[n] for n in (names or h)})
    assert bench._pmc_traffic(False) == (None, None) and bench._pmc_traffic(True) == (None, None)        # no file at all
    rec = {'_spconv_gmm_all': {'hbm_MB_per_launch': 12.5}, '_meta': {'csrc_sha16': dict(h), 'git_head': 'abc123'}}
    (prof / 'round9_pmc_traffic.json').write_text(json.dumps(rec))
    t, src = bench._pmc_traffic(False)
    assert t == 12.5e6 and 'round9_pmc_traffic.json' in src and 'abc123' in src
    t, src = bench._pmc_traffic(True)                       # the file holds no cfg3 pass
    assert t is None and 'no pass of this command' in src
    rec['cfg3'] = {'_spconv_gmm_all': {'hbm_MB_per_launch': 40.25}}
    (prof / 'round9_pmc_traffic.json').write_text(json.dumps(rec))
    t, src = bench._pmc_traffic(True)
    assert t == 40.25e6 and 'cfg3 command' in src
    for stale in ('spconv_wg.hip', 'gemm.hip'):            # ANY source: a GEMM edit makes the file stale too
        rec['_meta']['csrc_sha16'] = dict(h, **{stale: '0' * 16})
        (prof / 'round9_pmc_traffic.json').write_text(json.dumps(rec))
        for bf in (False, True):
            t, src = bench._pmc_traffic(bf)
            assert t is None and 'different kernel sources' in src
    # the passes are passes of the cfg2 / cfg3 commands: other workloads never quote them, matching sources or not
    rec['_meta']['csrc_sha16'] = dict(h)
    (prof / 'round9_pmc_traffic.json').write_text(json.dumps(rec))
    assert bench._pmc_traffic(False, 'cfg2')[0] == 12.5e6
    for wl in ('cfg4', 'cfg5'):
        t, src = bench._pmc_traffic(False, wl)
        assert t is None and 'cfg2' in src


def test_git_head_falls_back_to_the_stamp_file(tmp_path, monkeypatch):
    """On the GPU box the snapshot has no .git: bench.git_head() reads the .git_head file tools/grun.sh wrote."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.git_head() not in ('', None)
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.git_head() == '?'
    (tmp_path / '.git_head').write_text('deadbeef1234+dirty(ab12cd34)\n')
    assert bench.git_head() == 'deadbeef1234+dirty(ab12cd34)'


def test_tree_hash_is_a_function_of_the_shipped_sources_only(tmp_path, monkeypatch):
    """The stamp of a bench line: computed from the files themselves -- the same in the build container, on the GPU
    box (no .git there) and in the driver's checkout; changes with any source byte, not with build products or caches."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    h = bench.tree_hash()
    assert len(h) == 16 and h == bench.tree_hash()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--tree-hash'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == h
    (tmp_path / 'unidet3d_amd' / 'csrc').mkdir(parents=True)
    (tmp_path / 'include').mkdir()
    (tmp_path / 'bench.py').write_text('x')
    (tmp_path / 'include' / 'u3d.h').write_text('h')
    (tmp_path / 'unidet3d_amd' / 'a.py').write_text('a')
    (tmp_path / 'unidet3d_amd' / 'csrc' / 'k.hip').write_text('k')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    h0 = bench.tree_hash()
    (tmp_path / 'unidet3d_amd' / 'csrc' / 'k.o').write_text('object code')          # build products do not count
    (tmp_path / 'unidet3d_amd' / 'csrc' / 'libu3d_hip.so').write_text('library')
    assert bench.tree_hash() == h0
    (tmp_path / 'unidet3d_amd' / 'csrc' / 'k.hip').write_text('k2')
    assert bench.tree_hash() != h0


def test_kept_bench_lines_follow_the_contract():
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round2_final_bench_*.json')) + glob.glob(os.path.join(ROOT, 'profiles', 'round3_bench_*.json')) +
                   glob.glob(os.path.join(ROOT, 'profiles', 'round4_bench_*.json')) + glob.glob(os.path.join(ROOT, 'profiles', 'round5_bench_*.json')))
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
        if ('round4' in os.path.basename(f) or 'round5' in os.path.basename(f)) and d['config'].get('fp32_math') == 'bf16x3':
            # round 4 on: a three-plane line prices frac against the instruction ceiling, the fp32-peak figure rides along
            assert abs(r['peak'] - 2500.0 / 6) < 1e-9 and r['dtype_peak'] == 157.3 and abs(r['frac_of_dtype_peak'] - r['achieved'] / 157.3) < 1e-9
            assert 0 < r['frac'] < 1 and r['frac_of_dtype_peak'] > r['frac']
        else:
            assert d['dtype'] in ('f32', 'bf16') and r['peak'] == (157.3 if d['dtype'] == 'f32' else 2500.0)
        if 'cpu_baseline' in d and d['cpu_baseline']:
            c = d['cpu_baseline']
            assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    for name in ('round2_final_bench_fp32.json', 'round3_bench_fp32.json'):
        main = json.load(open(os.path.join(ROOT, 'profiles', name)))
        assert main['cpu_baseline'] and main['n_gpus'] == 1 and main['config']['global_batch'] == 8 and main['dtype'] == 'f32'
    # round 3: the driver-visible line carries BASELINE configs[2] as a block of its own
    c3 = main['cfg3']
    assert c3['dtype'] == 'bf16' and c3['config']['global_batch'] == 16 and 'cfg3' in c3['config']['workload']
    assert abs(c3['value'] - 16 / c3['ms_per_step'] * 1e3) < 1e-6 * c3['value'] and c3['roofline']['peak'] == 2500.0
    assert 'median of' in main['cpu_baseline']['sample'] and 'warm-up' in main['cpu_baseline']['sample']      # SURVEY 8(d) protocol
    # the headline forms its fp32 products from three bf16 planes per operand and says so; the same workload on the
    # native fp32 MFMA kernels is a block of the same line
    assert main['config']['fp32_math'] == 'bf16x3' and 'math' in main['roofline'] and main['roofline']['instruction_peak'] > main['roofline']['peak']
    nm = main['fp32_native_mfma']
    assert nm['value'] > 0 and abs(nm['value'] - 8 / nm['ms_per_step'] * 1e3) < 1e-6 * nm['value']
    assert nm['roofline']['peak'] == 157.3 and 'math' not in nm['roofline']
    assert nm['warmup_losses'][0] == main['config']['warmup_losses'][0]          # same initial weights and scenes


def test_round5_line_prices_every_family_by_the_mfma_it_issues_and_ends_with_a_summary():
    """cfg3's weight gradient runs on bf16 MFMAs and is priced against the bf16 peak (round 4 quoted 0.82 of the fp32
    peak for it); the cfg3 block carries PMC traffic; the line names its commit and ends with the numbers a reader wants first."""
    d = json.load(open(os.path.join(ROOT, 'profiles', 'round5_bench_fp32.json')))
    assert list(d)[-1] == 'summary' and d['git_head'] not in ('', '?')
    sm = d['summary']
    assert abs(sm['scenes_per_s'] - d['value']) < 0.01 and abs(sm['cfg3_bf16_scenes_per_s'] - d['cfg3']['value']) < 0.01
    assert abs(sm['fp32_native_mfma_scenes_per_s'] - d['fp32_native_mfma']['value']) < 0.01 and sm['cpu_baseline_scenes_per_s'] == d['cpu_baseline']['value']
    assert d['config']['wgrad_overlap'] == 2 and 'side-stream chain' in d['config']['workload']
    for k, v in d['cfg3']['kernels'].items():
        assert v['mfma_peak'] == 2500.0 and 'bf16' in v['mfma_instr'] and 0 < v['frac_mfma'] < 1, k
    for k, v in d['kernels'].items():
        assert v['mfma_peak'] == 157.3 and 0 < v['frac_mfma'] < 1, k
    assert 'f32' in d['kernels']['conv_wgrad']['mfma_instr'] and 'bf16x3_ceiling' not in d['kernels']['conv_wgrad']
    for blk in (d, d['cfg3']):
        r = blk['roofline']
        assert r['traffic'] and r['traffic'] > 0.5 * r['algorithmic_bytes_per_launch'] and d['git_head'][:12] in r['traffic_source']
    pm = json.load(open(os.path.join(ROOT, 'profiles', 'round5_pmc_traffic.json')))
    assert pm['_meta']['git_head'] == d['git_head'] and 'cfg3' in pm and '_spconv_wgrad_all' in pm
