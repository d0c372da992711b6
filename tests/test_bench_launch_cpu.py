"""bench.py's launch contract without a GPU (--stub-trainer: CPU ranks under gloo).  What is real here: the
self-launch of ``python bench.py --gpus N`` (N fresh children, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by the
parent, which itself never joins the job), the torch.distributed.run entry, the barrier + max-over-ranks timing, ONE
JSON line from rank 0, and a non-zero exit code when a rank dies.  The GPU version of the same path is
tests/test_ddp_gpu.py::test_bench_self_launch_two_ranks_one_gpu."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return env


def _one_json_line(stdout):
    # gloo prints its own connection chatter on the ranks' stdout at rendezvous ("[Gloo] Rank 0 is connected to ..."), and
    # the lines of two ranks can interleave into fragments that no longer start with "[Gloo]"; the benchmark's line is the
    # one that is a JSON object (RCCL, the backend of the GPU runs, prints nothing)
    lines = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '5', '--warmup', '1', '--stub-trainer'],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 2 and out['steps'] == 5 and out['warmup'] == 1 and out['config']['parallelism'] == 'dp2'
    assert out['collective'] == {'backend': 'gloo', 'ranks': 2, 'allreduce_sum': 2.0 ** 5}      # both ranks took part
    # the reported time is the slowest rank's: rank 1 sleeps twice as long per step as rank 0
    assert out['rank_time']['max_s'] >= out['rank_time']['min_s'] > 0
    assert abs(out['ms_per_step'] - out['rank_time']['max_s'] / 5 * 1e3) < 1e-6
    assert out['rank_time']['max_s'] >= 5 * 0.004


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, '--steps', '3', '--warmup', '0', '--stub-trainer'], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)['n_gpus'] == 1


def test_self_launch_propagates_a_dead_rank():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '0', '--stub-trainer',
                        '--stub-fail-rank', '1'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]       # no JSON line from a failed job


def test_torchrun_entry_still_works():
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29631', BENCH, '--gpus', '2', '--steps', '3',
                        '--warmup', '1', '--stub-trainer'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 2 and out['collective']['ranks'] == 2


def test_launcher_and_flag_must_agree():
    env = dict(_env(), WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29632')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--stub-trainer'], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr


def test_self_launch_eight_ranks():
    """The N = 8 line of the driver's scaling run, rehearsed on CPU ranks: eight children, one JSON line, the slowest rank's time."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '4', '--warmup', '1', '--stub-trainer'],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 8 and out['config']['parallelism'] == 'dp8'
    assert out['collective'] == {'backend': 'gloo', 'ranks': 8, 'allreduce_sum': 8.0 ** 4}
    assert out['rank_time']['max_s'] >= 4 * 0.002 * 8 > out['rank_time']['min_s'] * 0 
