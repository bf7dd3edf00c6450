"""CPU: `python bench.py --gpus N` (N > 1) without a torchrun environment must never run one rank silently (review r05, weak
point 8a): it re-executes itself under torch.distributed.run, or -- on a node with fewer GPUs than asked for, like this
container -- exits non-zero saying so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_world_size_refuses_instead_of_running_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NLT_BENCH_SHARE_GPU')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                            # (a real multi-GPU node: the relaunch path runs the bench itself)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert 'refusing to run fewer ranks' in p.stderr and '"n_gpus"' not in p.stdout


def test_world_size_that_disagrees_with_gpus_is_an_error():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and '--gpus 4 but WORLD_SIZE=1' in (p.stderr + p.stdout)
