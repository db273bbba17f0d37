from pfrl_amd.replay_buffers.replay_buffer import ReplayBuffer  # NOQA
from pfrl_amd.replay_buffers.prioritized import PrioritizedReplayBuffer, PriorityWeightError  # NOQA
