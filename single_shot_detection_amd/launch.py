"""One process per GPU, started from a plain ``python script.py --gpus N`` -- the role of bf/training/helpers.py:129-142 (`launch`:
spawn ``nproc`` workers on the local machine) and bf/training/env.py:55-67 (tcp://127.0.0.1:port rendezvous, device = rank).

The parent never touches the GPU: it only picks a free port, starts N fresh interpreters of the same command line with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what ``torch.distributed.run`` would set), lets them inherit
stdout / stderr (rank 0 prints the result line) and returns the worst exit code.  Nothing is re-exec'ed."""
import os
import socket
import subprocess
import sys
import time


def under_launcher():
    """True inside a rank started by this launcher or by torch.distributed.run."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(nproc, argv, env=None, poll=0.2):
    """Run ``argv`` (a full command line, e.g. [sys.executable, 'bench.py', '--gpus', '8']) as ``nproc`` ranks; returns the exit code
    (0 only if every rank returned 0).  When one rank fails the others are terminated (their own PIDs, nothing else)."""
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL's intra-node transport on this driver
    procs = []
    for rank in range(nproc):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen(list(argv), env=e))
    code = 0
    alive = list(procs)
    while alive:
        time.sleep(poll)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in alive:   # a dead rank would leave the others waiting in a collective for ever (helpers.py:142-143 just joins)
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return code


def self_launch_if_needed(n_gpus, script):
    """Call first thing in main(): with --gpus N > 1 outside a launcher, become the parent of N ranks and exit with their code."""
    if n_gpus > 1 and not under_launcher():
        sys.exit(launch(n_gpus, [sys.executable, os.path.abspath(script)] + sys.argv[1:]))
