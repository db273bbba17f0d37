"""Replay buffers whose contents survive the process
(reference pfrl/replay_buffers/persistent.py:10-165).

The queues are ``PersistentRandomAccessQueue`` logs (chunk / index / CRC-32 files, see
``pfrl_amd.collections.persistent_collections``), so a directory written by either implementation
can be re-opened by the other.  Items go through ``pickle`` when they are appended, so -- unlike the
in-memory buffers -- observations shared between consecutive transitions are stored once per
transition.  These are host containers: an agent's ``bind(device)`` leaves them on the host, and
``batch_experiences`` takes the list-of-dicts route.  ``save`` / ``load`` are no-ops because the
directory already is the checkpoint.
"""
import os
import warnings

from pfrl_amd.collections.persistent_collections import PersistentRandomAccessQueue
from pfrl_amd.replay_buffers.episodic import EpisodicReplayBuffer
from pfrl_amd.replay_buffers.replay_buffer import ReplayBuffer


def _open_queue(kind, dirname, capacity, ancestor, logger, distributed):
    if distributed:
        # the multi-node queue of the reference lives in a private package (:60-78)
        raise RuntimeError("`pfrlmn` private package is required to enable distributed "
                           "execution support of {}.".format(kind))
    return PersistentRandomAccessQueue(dirname, capacity, ancestor=ancestor, logger=logger)


class PersistentReplayBuffer(ReplayBuffer):
    """1-step ``ReplayBuffer`` logged under ``dirname`` (and re-loaded from it on construction)."""

    def __init__(self, dirname, capacity, *, ancestor=None, logger=None, distributed=False,
                 group=None):
        super().__init__(capacity)
        self.memory = _open_queue("PersistentReplayBuffer", dirname, capacity, ancestor, logger,
                                  distributed)

    def bind(self, device, phi=None):
        return self                       # host container; see module docstring

    def save(self, _):
        pass

    def load(self, _):
        warnings.warn("{}.load() has been ignored, as it is persistent replay buffer".format(self))


class PersistentEpisodicReplayBuffer(EpisodicReplayBuffer):
    """``EpisodicReplayBuffer`` with two logs: ``dirname/memory`` (single transitions) and
    ``dirname/episodic_memory`` (whole episodes).  Both are opened with ``maxlen=capacity``."""

    def __init__(self, dirname, capacity, *, ancestor=None, logger=None, distributed=False,
                 group=None):
        super().__init__(capacity)
        self.memory_dir = os.path.join(dirname, "memory")
        self.episodic_memory_dir = os.path.join(dirname, "episodic_memory")
        kind = "PersistentEpisodicReplayBuffer"
        self.memory = _open_queue(kind, self.memory_dir, capacity, ancestor, logger, distributed)
        self.episodic_memory = _open_queue(kind, self.episodic_memory_dir, capacity, ancestor,
                                           logger, distributed)

    def save(self, _):
        pass

    def load(self, _):
        warnings.warn("PersistentEpisodicReplayBuffer.load() is called but it has not effect.")
