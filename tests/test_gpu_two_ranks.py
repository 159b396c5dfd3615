"""Two ranks against the real RCCL on one GPU: as far as one device allows."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_agree_on_the_collective_and_match_single_rank():
    """Two processes share cuda:0 (process group over gloo) and ask for collective="rccl".  RCCL does not accept two ranks
    of one communicator on the same device, so the collective ncclCommInitRank behind mocap_comm_init fails -- on both ranks;
    pipeline.negotiate_rccl has to notice on every rank, destroy what was created and fall back to torch.distributed on BOTH
    (a rank left alone inside a collective would hang: the workers run under a timeout).  Whichever road the ranks take, they
    take the same one, and their time steps equal a single-rank tracker's bit for bit.  The counterpart of the reference's
    thread hand-off (RealtimeTracking_FLIR.py:107-113,180-183,304-312) with a real multi-rank set-up behind it; an
    ncclAllGather between two DEVICES needs a second GPU and is not run here."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(port)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    roads = {line.split("collective=")[1] for o in outs for line in o.splitlines() if line.startswith("OK rank")}
    assert len(roads) == 1 and roads <= {"rccl", "torch"}, outs
