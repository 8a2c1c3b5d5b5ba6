"""Two data-parallel ranks with the real GPU model.  RCCL wants one device per rank and the test box
has one, so the ranks share cuda:0 and talk over gloo (gradients staged through host memory): every
part of the multi-rank step except the transport runs as in production.  The child processes check
that the reduced gradients equal the mean of the ranks' local gradients, that the ranks hold
identical parameters after every step, and that the schedule vote of ``GanTrainer.tune`` agrees."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rehearsal.py")], cwd=ROOT,
                          capture_output=True, text=True, timeout=170)
    assert done.returncode == 0 and "dp rehearsal OK" in done.stdout, done.stdout[-2000:] + done.stderr[-4000:]


@pytest.mark.gpu
def test_two_ranks_with_the_hooks_of_the_overlapped_reduction():
    """The same rehearsal with the chunked reductions issued from grad-ready hooks during the backward pass (the
    RCCL schedule; over gloo on one GPU nothing actually overlaps): the deferred weight gradients are issued chunk
    by chunk before the hooks read them, ranks stay identical and the recorded step still equals the eager one."""
    env = dict(os.environ, HIVT_REHEARSAL_OVERLAP="1")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rehearsal.py")], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=170)
    assert done.returncode == 0 and "dp rehearsal OK" in done.stdout, done.stdout[-2000:] + done.stderr[-4000:]


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_two_gpus_are_present():
    """The same rehearsal over RCCL, one GPU per rank, with the chunked all-reduces overlapping the backward
    pass: needs two visible GPUs, skipped otherwise."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HIVT_REHEARSAL_BACKEND="nccl")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rehearsal.py")], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=170)
    assert done.returncode == 0 and "dp rehearsal OK" in done.stdout, done.stdout[-2000:] + done.stderr[-4000:]
