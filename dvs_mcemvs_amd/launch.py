"""One process per GPU: how `bench.py --gpus N` (and anything else that wants N ranks on one node) gets them.

Two ways in, one result:

* a launcher already started the ranks (`python -m torch.distributed.run --nproc-per-node N ...` exports RANK /
  LOCAL_RANK / WORLD_SIZE / MASTER_*): `Dist()` joins the group;
* nobody did (`python bench.py --gpus N` with no WORLD_SIZE in the environment): `spawn_ranks(N, argv)` starts N
  copies of the same command itself, one per device, with that environment, relays rank 0's standard output and
  returns the worst exit status.  It REFUSES (no fallback to fewer ranks) when the node has fewer than N devices.

`Dist` is the side channel of the ranks -- rendezvous, barriers, a few scalars, the 128-byte RCCL unique id --
over a gloo (CPU) torch.distributed group.  It never carries volume data: the data-path collective is the engine's
RCCL communicator (include/dsi_engine.h, dsi_comm_*).
"""
import os
import socket
import subprocess
import sys
import tempfile
import time

SPAWNED_ENV = "DSI_LAUNCH_SPAWNED"      # set in the ranks spawn_ranks() starts (they must not spawn again)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_a_launcher(env=None):
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


# Two settings the ranks get unless the caller's environment already has them -- guesses until an 8-GPU node has run
# (RCCL with more than one rank never has, in any round): whatever the environment says wins, DSI_LAUNCH_NO_ENV_DEFAULTS=1
# sets neither, and bench.py prints what the ranks actually ran with (environment_report) in its JSON line.
ENV_DEFAULTS = {
    # the host driver only supports dmabuf IPC (without it RCCL fails with hipIpcGetMemHandle: invalid argument)
    "HSA_ENABLE_IPC_MODE_LEGACY": "0",
    # one node: RCCL's bootstrap sockets on the loopback interface (the container's hostname may not resolve)
    "NCCL_SOCKET_IFNAME": "lo",
}
NO_DEFAULTS_ENV = "DSI_LAUNCH_NO_ENV_DEFAULTS"


def apply_env_defaults(env):
    if env.get(NO_DEFAULTS_ENV) == "1":
        return env
    for k, v in ENV_DEFAULTS.items():
        env.setdefault(k, v)
    return env


def environment_report(env=None):
    """What a rank runs with for the settings above: value and whether it is this module's default."""
    env = os.environ if env is None else env
    rep = {}
    for k, v in ENV_DEFAULTS.items():
        have = env.get(k)
        rep[k] = {"value": have, "is_launch_default": have == v and env.get(NO_DEFAULTS_ENV) != "1"}
    rep["defaults_disabled"] = env.get(NO_DEFAULTS_ENV) == "1"
    return rep


def rank_environment(base, rank, world, port):
    env = dict(base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), SPAWNED_ENV: "1"})
    return apply_env_defaults(env)


def spawn_ranks(world, argv, n_devices=None, env=None, timeout=None, poll_s=0.05, out=None, err=None):
    """Start `world` copies of the command `argv` (a list, e.g. [sys.executable, "bench.py", ...]), rank r with
    RANK = LOCAL_RANK = r, and wait for them.  Rank 0's stdout is relayed to `out` (default sys.stdout) -- the ONE
    JSON line of bench.py is printed by rank 0 --, every other rank's stdout and all stderr go to `err`.  If a rank
    fails, the others are terminated (by the exact PIDs started here) and its status is returned.
    n_devices: devices visible on the node; fewer than `world` is an error (status 2), never a smaller job."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    if world < 1:
        print("launch: --gpus must be >= 1 (got %d)" % world, file=err)
        return 2
    if n_devices is not None and n_devices < world:
        print("launch: %d ranks requested but this node has %d GPU device(s); refusing to run a smaller job "
              "(one rank per device, no fallback)" % (world, n_devices), file=err)
        return 2
    port = free_port()
    base = dict(os.environ if env is None else env)
    procs = []
    rank0_out = tempfile.TemporaryFile(mode="w+")    # (a file, not a pipe: nothing reads it before the ranks are done)
    for r in range(world):
        procs.append(subprocess.Popen(argv, env=rank_environment(base, r, world, port),
                                      stdout=rank0_out if r == 0 else err, stderr=err, text=True))
    t0 = time.time()
    status = 0
    try:
        alive = set(range(world))
        while alive:
            for r in list(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and status == 0:
                    status = rc
                    print("launch: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=err)
            if status != 0 or (timeout is not None and time.time() - t0 > timeout):
                if status == 0:
                    status = 124
                    print("launch: timeout after %.0f s; stopping the ranks" % timeout, file=err)
                break
            if alive:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    rank0_out.seek(0)
    text = rank0_out.read()
    rank0_out.close()
    if text:
        out.write(text)
        out.flush()
    return status


class Dist:
    """Rendezvous / barrier / scalar reductions / object broadcast over the ranks (gloo: CPU).  With one rank
    every method is the identity and torch is not imported."""

    def __init__(self, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.spawned = os.environ.get(SPAWNED_ENV) == "1"
        self.dist = self.torch = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if os.environ.get(NO_DEFAULTS_ENV) != "1":
                if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
                    os.environ.setdefault("NCCL_SOCKET_IFNAME", ENV_DEFAULTS["NCCL_SOCKET_IFNAME"])
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", ENV_DEFAULTS["HSA_ENABLE_IPC_MODE_LEGACY"])
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, v, op):
        if self.dist is None:
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX if self.dist else None)

    def min(self, v):
        return self._reduce(v, self.dist.ReduceOp.MIN if self.dist else None)

    def sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM if self.dist else None)

    def broadcast(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.dist is None:
            return [obj]
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def aggregate(D, units_this_rank, elapsed_this_rank, steps):
    """The bench contract's whole-job figure: (units all ranks processed per step) x steps / MAX over ranks of the
    timed region.  Returns (value per second, max elapsed, total units per step)."""
    elapsed = D.max(elapsed_this_rank)
    units = D.sum(units_this_rank)
    return units * steps / elapsed, elapsed, units
