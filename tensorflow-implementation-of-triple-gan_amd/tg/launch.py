"""One process per GPU without an external launcher: `spawn_ranks` starts N fresh children of a script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what `python -m torch.distributed.run` would export) and waits for them.

The parent never touches HIP: devices are counted from the kernel's topology files (sysfs), not through torch / hipGetDeviceCount
(which opens /dev/kfd in the launcher — a process that then starts N children must not hold the GPU).  It never replaces itself with
another program, children are stopped by their exact PIDs, and a failed rank makes the whole job exit non-zero."""
import glob
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


KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def _visible_filter(n, env):
    """apply HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (comma lists of indices or UUIDs; an empty string
    hides every device, an out-of-range or negative index ends the list — the runtime's rule — and a repeated entry counts once)."""
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = env.get(var)
        if v is None:
            continue
        seen = set()
        for tok in [t.strip() for t in v.split(',')] if v.strip() else []:
            if tok.lstrip('+-').isdigit():
                if int(tok) < 0 or int(tok) >= n:     # "-1" (the usual way to hide everything / end the list) and an out-of-range index end it
                    break
                seen.add(int(tok))                    # a repeated index names the same device again
            elif tok:                                 # a UUID ("GPU-...") names one device
                seen.add(tok)
        n = min(n, len(seen))
    return n


def visible_devices(env=None, nodes_dir=None, render_glob='/dev/dri/renderD*'):
    """Number of GPUs a child process would see, WITHOUT any HIP / HSA call in this process: KFD topology nodes with simd_count > 0
    (CPU nodes have 0), capped by the render nodes present in /dev/dri (a container sees the host's whole topology but only the
    device files passed through to it; the render nodes alone when the topology is not mounted), then the *_VISIBLE_DEVICES filters.
    An over-count is harmless — every rank checks its own device in tg.dist.init() and fails the job — an under-count is not."""
    env = os.environ if env is None else env
    nodes_dir = KFD_NODES if nodes_dir is None else nodes_dir
    n = 0
    found = False
    for prop in glob.glob(os.path.join(nodes_dir, '*', 'properties')):
        try:
            with open(prop) as f:
                for line in f:
                    parts = line.split()
                    if len(parts) == 2 and parts[0] == 'simd_count':
                        found = True
                        if int(parts[1]) > 0:
                            n += 1
        except OSError:
            continue
    render = len(glob.glob(render_glob))
    if not found:
        n = render
    elif render:
        n = min(n, render)
    return _visible_filter(n, env)


def holds_gpu():
    """True when this process has the GPU driver's device file open (any HIP call does that) — spawn_ranks refuses to fork then."""
    try:
        for fd in os.listdir('/proc/self/fd'):
            try:
                if os.readlink('/proc/self/fd/' + fd) == '/dev/kfd':
                    return True
            except OSError:
                continue
    except OSError:
        pass
    return False


def spawn_ranks(n, argv, env=None, need_devices=True, poll=0.2):
    """Run `python argv...` as ranks 0..n-1 on 127.0.0.1.  Rank 0 inherits stdout (its one JSON line goes straight through); every
    rank inherits stderr.  Returns the job's exit code: 0 when every rank exited 0, else the first non-zero one (the other ranks
    are terminated as soon as one fails — a collective they wait in would never complete)."""
    if holds_gpu():
        sys.stderr.write("tg.launch: this process has already initialised the GPU (/dev/kfd is open); start ranks from a process that has not\n")
        return 2
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
