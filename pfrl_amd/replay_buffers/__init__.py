from pfrl_amd.replay_buffers.replay_buffer import ReplayBuffer  # NOQA
from pfrl_amd.replay_buffers.prioritized import PrioritizedReplayBuffer, PriorityWeightError  # NOQA
from pfrl_amd.replay_buffers.episodic import EpisodicReplayBuffer  # NOQA
from pfrl_amd.replay_buffers.prioritized_episodic import PrioritizedEpisodicReplayBuffer  # NOQA
from pfrl_amd.replay_buffers.persistent import (PersistentEpisodicReplayBuffer,  # NOQA
                                                PersistentReplayBuffer)
