"""bench.py's launcher: `--gpus N` must start its own N ranks (the driver runs it bare for the scaling bench).

CPU: the launch + rendezvous + blob broadcast rehearsal (`--dry-run`, gloo, measures nothing).
GPU: two ranks sharing the box's one GPU over gloo with the real kernels, and one rank under
torch.distributed.run with the nccl (= RCCL) backend: broadcast_blob -> melf_ctx_create(blob_on_device=1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_gpus_2_self_launch_dry_run():
    line = _run([sys.executable, BENCH, '--gpus', '2', '--dry-run'])
    assert line['dry_run'] is True and line['n_gpus'] == 2 and line['collective_ranks'] == 2 and line['value'] is None


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='1')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '3', '--dry-run'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b'WORLD_SIZE' in p.stderr


SMALL = ['--steps', '3', '--warmup', '1', '--batch', '64', '--nbuf', '2', '--preheat', '0']


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_over_gloo():
    line = _run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--share-gpu', '--only', 'none'] + SMALL)
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 128 and line['config']['parallelism'] == 'dp2'
    assert len(line['per_rank_ms_per_step']) == 2 and line['value'] > 0
    assert line['config']['frames_read_ok_batch0'] >= 60
    assert line['roofline']['launches'] == 3


@pytest.mark.gpu
def test_one_rank_nccl_broadcast_and_parity_gate():
    """RCCL with one rank: all_reduce, broadcast of the calibration blob GPU-side, context from the device-resident
    bytes; the CPU-oracle sample (checker) must agree with the records."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29533', BENCH, '--gpus', '1', '--only', 'cpu,config4', '--cpu-sample', '24'] + SMALL
    line = _run(cmd)
    assert line['rccl_ranks'] == 1 and line['backend'] == 'nccl' and line['n_gpus'] == 1
    assert line['cpu_baseline']['parity_mismatches_vs_gpu'] == 0
    assert line['config4']['cpu_baseline']['parity_mismatches_vs_gpu'] == 0
    assert line['config4']['frames_read_ok_batch0'] >= 60


@pytest.mark.gpu
def test_four_ranks_rehearsal_on_one_gpu():
    """The code paths of rank != 0 beyond two ranks (config-5 params made on rank 0 only and broadcast, the rank-0-only blocks
    with the other ranks waiting at barriers, max-over-ranks timing), rehearsed on the box's one GPU over gloo at a reduced
    batch: the only way on this pool to execute them before a real SCALE run.  Four ranks, not eight: the pool allows at most
    six processes on a card (the test runner and the launcher count); nothing in bench.py depends on the rank count beyond
    `rank == 0`."""
    line = _run([sys.executable, BENCH, '--gpus', '4', '--backend', 'gloo', '--share-gpu', '--skip', 'sustained,twostream,hostfed,jpeg',
                 '--batch5', '16', '--cpu-sample', '16'] + SMALL, timeout=1200)
    assert line['n_gpus'] == 4 and line['config']['global_batch'] == 4 * 64 and line['config']['parallelism'] == 'dp4'
    assert len(line['per_rank_ms_per_step']) == 4 and all(t > 0 for t in line['per_rank_ms_per_step']) and line['value'] > 0
    assert line['cpu_baseline']['parity_mismatches_vs_gpu'] == 0
    assert line['fused_mask']['roofline']['launches'] >= 3
    c4 = line['config4']
    assert len(c4['per_rank_ms_per_step']) == 4 and c4['cpu_baseline']['parity_mismatches_vs_gpu'] == 0 and c4['frames_read_ok_batch0'] >= 60
    c5 = line['config5']
    assert c5['full_path']['dials'] == 6 and c5['full_path']['frames_read_ok'] >= 12
    assert c5['full_path']['parity_gate']['parity_mismatches_vs_gpu'] == 0 and c5['full_path']['parity_gate']['oracle_frames'] == 16
    assert c5['fused_mask']['roofline']['frac'] > 0


@pytest.mark.gpu
def test_two_ranks_every_default_block():
    """What a SCALE run executes: EVERY default block with more than one rank (the four-rank rehearsal above skips the long ones).
    Round 6 put a time-bounded preheat loop around calls that hold barriers and the ranks left it after different counts: this
    test is here so that such a loop hangs a test, not the driver's scaling run."""
    line = _run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--share-gpu', '--sustained', '0.3', '--batch5', '16',
                 '--cpu-sample', '16'] + SMALL, timeout=900)
    assert line['n_gpus'] == 2 and line['value'] > 0 and len(line['per_rank_ms_per_step']) == 2
    assert line['sustained']['steps'] > 0 and line['two_streams']['records_identical_to_timed_region']
    assert line['config5']['full_path']['untimed_preheat_steps'] > 0 and line['fused_mask']['untimed_preheat_launches'] > 0
    assert line['cpu_baseline']['parity_mismatches_vs_gpu'] == 0
