"""VectorEnv with one worker process per env, talking over Pipes.

Protocol as in the reference (pfrl/envs/multiprocess_vector_env.py): the
parent sends a command to every worker first and only then collects the
replies, so the envs step concurrently; workers ignore SIGINT so that Ctrl-C is
handled once, by the parent."""
import multiprocessing as mp
import signal

import numpy as np

from pfrl_amd.env import VectorEnv


class _Worker:
    """Command loop run inside the child process."""

    def __init__(self, conn, env_fn):
        self.conn = conn
        self.env = env_fn()
        self.handlers = {
            "step": lambda a: self.env.step(a),
            "reset": lambda _: self.env.reset(),
            "seed": lambda s: self.env.seed(s),
            "spaces": lambda _: (getattr(self.env, "action_space", None),
                                 getattr(self.env, "observation_space", None)),
            "spec": lambda _: getattr(self.env, "spec", None),
        }

    def serve(self):
        try:
            while True:
                cmd, payload = self.conn.recv()
                if cmd == "close":
                    self.conn.close()
                    return
                self.conn.send(self.handlers[cmd](payload))
        finally:
            self.env.close()


def _child_main(conn, env_fn):
    signal.signal(signal.SIGINT, signal.SIG_IGN)
    _Worker(conn, env_fn).serve()


class MultiprocessVectorEnv(VectorEnv):
    def __init__(self, env_fns):
        # One pipe + one daemonic worker at a time, and the parent drops its copy of the worker's
        # end right away: if a worker dies (its env constructor raised, it was killed), the
        # parent's recv() sees EOF instead of blocking forever, and a parent that exits without
        # close() does not hang in multiprocessing's atexit join.
        self.remotes, self.procs = [], []
        for fn in env_fns:
            parent_end, worker_end = mp.Pipe()
            proc = mp.Process(target=_child_main, args=(worker_end, fn), daemon=True)
            proc.start()
            worker_end.close()
            self.remotes.append(parent_end)
            self.procs.append(proc)
        self.closed = False
        self.last_obs = [None] * self.num_envs
        self.action_space, self.observation_space = self._ask(0, "spaces")

    @property
    def num_envs(self):
        return len(self.remotes)

    def _ask(self, i, cmd, payload=None):
        self.remotes[i].send((cmd, payload))
        return self._recv(i)

    def _recv(self, i):
        try:
            return self.remotes[i].recv()
        except (EOFError, ConnectionResetError):
            raise RuntimeError("env worker %d exited (see its traceback above)" % i) from None

    @property
    def spec(self):
        """The ``spec`` of the first env (fetched from its worker once)."""
        if not hasattr(self, "_spec"):
            self._check_open()
            self._spec = self._ask(0, "spec")
        return self._spec

    def _check_open(self):
        assert not self.closed, "This env is already closed"

    def step(self, actions):
        self._check_open()
        for remote, action in zip(self.remotes, actions):
            remote.send(("step", action))
        replies = [self._recv(i) for i in range(self.num_envs)]
        obs, rewards, dones, infos = zip(*replies)
        self.last_obs = obs
        return obs, rewards, dones, infos

    def reset(self, mask=None):
        self._check_open()
        keep = np.zeros(self.num_envs, dtype=bool) if mask is None else np.asarray(mask, bool)
        restarting = [i for i in range(self.num_envs) if not keep[i]]
        for i in restarting:
            self.remotes[i].send(("reset", None))
        obs = list(self.last_obs)
        for i in restarting:
            obs[i] = self._recv(i)
        self.last_obs = obs
        return obs

    def seed(self, seeds=None):
        self._check_open()
        if seeds is None:
            seeds = [None] * self.num_envs
        elif isinstance(seeds, int):
            seeds = [seeds] * self.num_envs
        elif not isinstance(seeds, list):
            raise TypeError("Type of Seeds {} is not supported.".format(type(seeds)))
        elif len(seeds) != self.num_envs:
            raise ValueError("length of seeds must be same as num_envs {}".format(self.num_envs))
        for remote, s in zip(self.remotes, seeds):
            remote.send(("seed", s))
        return [self._recv(i) for i in range(self.num_envs)]

    def close(self):
        self._check_open()
        self.closed = True
        for remote in self.remotes:
            try:
                remote.send(("close", None))
            except (BrokenPipeError, OSError):      # that worker is already gone
                pass
        for proc in self.procs:
            proc.join()

    def __del__(self):
        if not getattr(self, "closed", True):
            try:
                self.close()
            except Exception:     # interpreter teardown: the workers are daemonic anyway
                pass
