"""The N > 1 path of bench.py end to end on ONE GPU: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run ... bench.py --gpus 2 ...), sharing device 0 and using gloo instead of RCCL
(FRZ_BENCH_SHARE_DEVICE=1, a rehearsal switch the driver never sets).  Checks the contract of the JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_bench_prints_one_contract_line():
    env = dict(os.environ, FRZ_BENCH_SHARE_DEVICE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29531',
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '150', '--warmup', '50']
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, done.stdout[-2000:]  # rank 0 only
    line = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline'):
        assert key in line, key
    assert line['n_gpus'] == 2 and line['steps'] == 150 and line['warmup'] == 50 and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['value'] > 0 and abs(line['value'] - 2 * 65536 * 150 / (line['ms_per_step'] * 150 / 1e3)) / line['value'] < 1e-6
    assert 'cpu_baseline' not in line and 'secondary_workloads' not in line  # rank 0 at N = 1 only
    assert set(line['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    d = line['distributed']
    assert d['world_size'] == 2 and d['ranks_in_all_reduce'] == 2 and d['collectives_in_timed_region'] == 0
    # the one collective of the job (after the timed loop) saw both ranks' blocks
    assert line['timing']['job_metrics']['env_steps_counted'] == 2 * 65536 * 150 * line['timing']['blocks']


@pytest.mark.gpu
def test_one_rank_bench_over_rccl():
    """bench.py under torch.distributed.run with ONE rank and no rehearsal switch: backend 'nccl' (= RCCL) is initialised on device 0 and the
    job's collectives (the barriers, the max over ranks, the ONE metrics all-reduce after the timed loop) run on the GPU — the N > 1 code
    path with RCCL itself, as far as one GPU can take it.  No collective sits inside a timed block; the line prices what one
    per block would cost (`block_ms_median_with_a_collective_per_block`)."""
    env = {k: v for k, v in os.environ.items() if k != 'FRZ_BENCH_SHARE_DEVICE'}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29533',
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-secondary']
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['steps'] == 20 and line['warmup'] == 5
    assert line['timing']['blocks'] >= 10 and line['timing']['block_ms_min'] <= line['timing']['block_ms_median'] <= line['timing']['block_ms_max']
    assert abs(line['ms_per_step'] * 20 - line['timing']['block_ms_median']) < 1e-9
    # every timed block stepped every env 20 times and the all-reduced metrics saw all of them
    assert line['timing']['job_metrics']['env_steps_counted'] == 65536 * 20 * line['timing']['blocks']
    d = line['distributed']
    assert d['backend'] == 'nccl' and d['collectives_in_timed_region'] == 0 and d['collective_probe_blocks'] >= 10
    assert d['block_ms_median_without_collective'] > 0 and d['block_ms_median_with_a_collective_per_block'] > 0
    print('one-rank RCCL: block median %.4f ms without, %.4f ms with an all-reduce per block' % (
        d['block_ms_median_without_collective'], d['block_ms_median_with_a_collective_per_block']))


@pytest.mark.gpu
def test_examples_run(tmp_path):
    """examples/ are the reference's documented loops on this package: they must keep running as shipped."""
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'baselines_rollout.py'), '3', str(tmp_path / 'logs')], cwd=ROOT,
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert 'StrongestBaseline' in done.stdout and sorted(os.listdir(tmp_path / 'logs')) == ['0.csv', '1.csv', '2.csv']
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'random_rollout.py'), '2048'], cwd=ROOT, capture_output=True, text=True,
                          timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert all(name in done.stdout for name in ('wildfire', 'cybersecurity', 'rideshare'))
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'rollout_api.py')], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert all(text in done.stdout for text in ('auto-reset rollouts', 'recorded trajectory', 'reference-shaped loop on an exclusive device', 'rideshare rollout'))
