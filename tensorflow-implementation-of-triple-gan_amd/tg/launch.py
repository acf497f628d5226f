"""One process per GPU without an external launcher: `spawn_ranks` starts N fresh children of a script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what `python -m torch.distributed.run` would export) and waits for them.

The parent must not have initialised the GPU (it only counts devices), it never replaces itself with another program, children are
stopped by their exact PIDs, and a failed rank makes the whole job exit non-zero."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices():
    """number of HIP devices this process would see; counting does not initialise the GPU."""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(n, argv, env=None, need_devices=True, poll=0.2):
    """Run `python argv...` as ranks 0..n-1 on 127.0.0.1.  Rank 0 inherits stdout (its one JSON line goes straight through); every
    rank inherits stderr.  Returns the job's exit code: 0 when every rank exited 0, else the first non-zero one (the other ranks
    are terminated as soon as one fails — a collective they wait in would never complete)."""
    if need_devices:
        have = visible_devices()
        if have < n:
            sys.stderr.write("tg.launch: %d ranks requested but only %d HIP device(s) visible\n" % (n, have))
            return 2
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), TG_SPAWNED='1')
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(poll)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                      # exact PIDs of our own children
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc
