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
