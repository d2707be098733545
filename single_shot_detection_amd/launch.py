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


EADDRINUSE_HINTS = ('EADDRINUSE', 'Address already in use', 'address already in use')


def _is_listen_error(line, port):
    """A c10d / TCPStore complaint that the RENDEZVOUS port could not be bound -- not any line that mentions EADDRINUSE (another socket of
    the job, late in a long run, is not a lost port race)."""
    return any(h in line for h in EADDRINUSE_HINTS) and ('server socket' in line or 'listen' in line or 'bind' in line or str(port) in line)


def _run_once(nproc, argv, base, poll, grace, race_window=180.0):
    """One attempt: returns (exit code, True when a rank died with the rendezvous port taken).  A lost race shows within the ranks'
    start-up (interpreter start + imports + init_process_group): a failure later than ``race_window`` seconds after the launch is never
    taken for one, whatever its stderr says -- re-running a job that has already done work would duplicate its side effects."""
    t_launch = time.monotonic()
    t_first_failure = None
    procs = []
    for rank in range(nproc):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen(list(argv), env=e, stderr=subprocess.PIPE, text=True, errors='replace'))
    tails = [[] for _ in procs]

    def pump(p, tail):   # relay the rank's stderr line by line, remembering its end (to recognise a lost port race)
        for line in p.stderr:
            sys.stderr.write(line)
            tail.append(line)
            del tail[:-20]
    import threading
    threads = [threading.Thread(target=pump, args=(p, t), daemon=True) for p, t in zip(procs, tails)]
    for t in threads:
        t.start()
    code = 0
    alive = list(procs)
    deadline = None   # set when the first rank fails: the survivors get SIGTERM at once and SIGKILL once it has passed
    while alive:
        time.sleep(poll)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and code == 0:
                code = rc
                t_first_failure = time.monotonic()
                deadline = time.monotonic() + grace
                for q in alive:   # a dead rank would leave the others waiting in a collective for ever (helpers.py:142-143 just joins)
                    q.terminate()
        if deadline is not None and time.monotonic() > deadline:
            for q in alive:       # a rank stuck in a collective / driver call, or with a SIGTERM handler: its own PID, nothing else
                q.kill()
            deadline = float('inf')
    for t in threads:
        t.join(timeout=5)
    port = base.get('MASTER_PORT', '')
    port_race = (code != 0 and t_first_failure is not None and t_first_failure - t_launch <= race_window and
                 any(_is_listen_error(line, port) for tail in tails for line in tail))
    return code, port_race


def launch(nproc, argv, env=None, poll=0.2, grace=30.0, port_retries=2):
    """Run ``argv`` (a full command line, e.g. [sys.executable, 'bench.py', '--gpus', '8']) as ``nproc`` ranks; returns the exit code
    (0 only if every rank returned 0).  When one rank fails the others are terminated (their own PIDs, nothing else) and killed if they
    have not exited ``grace`` seconds later.  The rendezvous port is picked by binding port 0 and released before the ranks bind it: if
    another process takes it in between (EADDRINUSE in a rank's stderr) the whole launch is retried on a new port."""
    code = 1
    for attempt in range(port_retries + 1):
        base = dict(os.environ if env is None else env)
        base.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc))
        base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL's intra-node transport on this driver
        code, port_race = _run_once(nproc, argv, base, poll, grace)
        if not port_race:
            break
    return code


def self_launch_if_needed(n_gpus, script):
    """Call first thing in main(): with --gpus N > 1 outside a launcher, become the parent of N ranks and exit with their code."""
    if n_gpus > 1 and not under_launcher():
        sys.exit(launch(n_gpus, [sys.executable, os.path.abspath(script)] + sys.argv[1:]))
