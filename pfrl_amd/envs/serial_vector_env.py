"""VectorEnv over a list of in-process envs, stepped one after another.

Same behaviour as the reference's SerialVectorEnv (pfrl/envs/
serial_vector_env.py): ``reset(mask)`` restarts the envs whose mask entry is
False and returns the remembered observation for the rest."""
import numpy as np

from pfrl_amd.env import VectorEnv


class SerialVectorEnv(VectorEnv):
    def __init__(self, envs):
        self.envs = list(envs)
        first = self.envs[0]
        self.action_space = getattr(first, "action_space", None)
        self.observation_space = getattr(first, "observation_space", None)
        self.spec = getattr(first, "spec", None)
        self.last_obs = [None for _ in self.envs]

    @property
    def num_envs(self):
        return len(self.envs)

    def step(self, actions):
        obs, rewards, dones, infos = [], [], [], []
        for env, action in zip(self.envs, actions):
            o, r, d, info = env.step(action)
            obs.append(o)
            rewards.append(r)
            dones.append(d)
            infos.append(info)
        self.last_obs = tuple(obs)
        return self.last_obs, tuple(rewards), tuple(dones), tuple(infos)

    def reset(self, mask=None):
        keep = np.zeros(self.num_envs, dtype=bool) if mask is None else np.asarray(mask, bool)
        self.last_obs = [old if k else env.reset()
                         for k, env, old in zip(keep, self.envs, self.last_obs)]
        return self.last_obs

    def seed(self, seeds):
        for env, s in zip(self.envs, seeds):
            env.seed(s)

    def close(self):
        for env in self.envs:
            env.close()
