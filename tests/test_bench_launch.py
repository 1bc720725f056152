"""`python bench.py --gpus N` launches N ranks itself (reference
tools/dist_train.sh:7-9).  CPU test of the launcher control flow: DM_BENCH_DRYRUN skips the GPU
workload, the ranks rendezvous over gloo and rank 0 reports how many joined."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ, DM_BENCH_DRYRUN='1', DM_DIST_BACKEND='gloo', **extra_env)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env,
                          capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout           # exactly one JSON line, from rank 0
    assert json.loads(lines[0])['n_gpus'] == 2


def test_world_size_mismatch_fails():
    env = dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    e = dict(os.environ, DM_BENCH_DRYRUN='1', **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=e,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_stall_guard_ends_a_rank_that_makes_no_progress():
    """Ranks under a launcher have no watching parent: bench.start_stall_guard ends the process (exit code 17, a
    diagnostic naming the one-lane switch) when no step completes within the limit, and stays quiet otherwise."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "g = bench.start_stall_guard(0.4, rank=3); "
            "[ (time.sleep(0.1), g.__setitem__('t', time.monotonic())) for _ in range(8) ]; "      # progress: no exit
            "print('alive', flush=True); time.sleep(30)") % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 17, (r.returncode, r.stderr[-1000:])
    assert 'alive' in r.stdout and 'rank 3' in r.stderr and 'DM_TWO_LANES=0' in r.stderr
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "g = bench.start_stall_guard(0.4); time.sleep(0.2); g['done'] = True; time.sleep(1.5); print('done')") % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'done' in r.stdout, (r.returncode, r.stderr[-1000:])


def test_watchdog_kills_a_child_without_progress_and_retries_in_one_lane():
    """watched_single_gpu_run on CPU: a first attempt that never starts / stops beating is killed by the progress
    watchdog; the second attempt runs with DM_TWO_LANES=0 (and ends at the 'needs a GPU' assertion here)."""
    for fake, msg in (('1', 'did not start within'), ('2', 'no step or phase completed')):
        env = dict(os.environ, DM_BENCH_FAKE_HANG=fake, DM_BENCH_WATCHDOG_FIRST_S='2')
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DM_TWO_LANES', 'DM_BENCH_DRYRUN', 'DM_BENCH_CHILD'):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                           env=env, capture_output=True, text=True, timeout=300)
        assert msg in r.stderr and '(attempt 1): killed' in r.stderr, r.stderr[-1500:]
        assert 'needs a GPU' in r.stderr and r.returncode != 0, (r.returncode, r.stderr[-1500:])
        assert '(attempt 2)' not in r.stderr


def test_watchdog_parent_forwards_termination_to_its_child():
    """`timeout N python bench.py` (SIGTERM to the parent) must not orphan the measuring child."""
    import signal
    import time
    import psutil
    env = dict(os.environ, DM_BENCH_FAKE_HANG='1')         # the child sleeps without ever beating
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DM_TWO_LANES', 'DM_BENCH_DRYRUN', 'DM_BENCH_CHILD',
              'DM_BENCH_WATCHDOG_FIRST_S'):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                         env=env, stderr=subprocess.PIPE, text=True)
    try:
        kids = []
        for _ in range(200):
            time.sleep(0.1)
            kids = psutil.Process(p.pid).children(recursive=True)
            if kids:
                break
        assert kids, 'the watchdog parent started no child'
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=30)
        err = p.stderr.read()
        assert rc == 128 + signal.SIGTERM and 'measuring child killed' in err, (rc, err[-500:])
        gone, alive = psutil.wait_procs(kids, timeout=10)
        assert not alive
    finally:
        if p.poll() is None:
            p.kill()
