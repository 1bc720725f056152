"""The REAL DetMatch step with world size 2 on one GPU: `bench.py --gpus 2` as a fresh child process, both ranks on
cuda:0 (DM_FORCE_DEVICE=0), gloo process group (RCCL refuses two ranks on one device; device tensors of the
collectives are staged through the host, mm3d/parallel.py:_host_staged).  Everything that exists per rank runs
together for the first time here: look-ahead geometry, early backward passes, the deferred 2D trunk backward,
bucket order of the gradient exchange, the log all-reduce, the fused optimizers — and the ranks must end with
identical parameters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('mode,lanes', [('collect', '1'), ('hooks', '1'), ('collect', '0')])
def test_two_ranks_on_one_gpu_stay_in_sync(dev, mode, lanes):
    # lanes '1': the order that ships (three stream lanes per rank — round 5 ran this test in the one-lane order only:
    # two processes with the lanes on one device dead-locked in a third of its runs, DESIGN.md 6.R6 says why)
    env = dict(os.environ, DM_FORCE_DEVICE='0', DM_DIST_BACKEND='gloo', DM_BENCH_CHECK_SYNC='1', DM_GRAD_MODE=mode,
               DM_TWO_LANES=lanes)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    assert out['last_loss'] is not None and out['last_loss'] == out['last_loss'] and abs(out['last_loss']) < 1e9
    ps = out['param_sync']
    assert ps['finite'] and ps['n_values'] > 1e7
    assert ps['max_abs_diff_between_ranks'] == 0.0, ps


def test_bench_watchdog_falls_back_to_one_lane():
    """`python bench.py` measures in a child process under a watchdog: a first attempt that never finishes (test hook) is
    killed and the bench repeats once in the one-lane order, saying so in its line."""
    import json
    env = dict(os.environ, DM_BENCH_FAKE_HANG='1', DM_BENCH_WATCHDOG_FIRST_S='3')
    env['DM_TWO_LANES'] = '1'      # the first attempt is the bench default; the fallback switches it off
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert 'one-lane order' in line['note'] and line['config']['stream_order'] == 'glue' and line['value'] > 0
    assert 'did not start within' in r.stderr and 'killed' in r.stderr
