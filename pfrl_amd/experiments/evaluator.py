"""Periodic evaluation.

Same contract as ``pfrl.experiments.evaluator``
(/root/reference/pfrl/experiments/evaluator.py): ``eval_performance`` returns
mean / median / stdev / max / min of episode returns (:218-266), batch
evaluation walks a VectorEnv and takes the first n finished episodes or the
episodes that fit in n_steps (:100-215), ``Evaluator`` evaluates every
``eval_interval`` steps, appends a tab-separated row to ``scores.txt``
(columns: steps, episodes, elapsed, mean, median, stdev, max, min, agent
statistics..., env statistics...; :375-393) and saves the best agent
(:396-521).  Evaluation uses the agent's normal ``batch_act`` in eval mode, so
with a device VectorEnv observations never leave HBM.
"""
import logging
import os
import statistics
import time

import numpy as np

from pfrl_amd.env import VectorEnv
from pfrl_amd.experiments.train_agent_batch import save_agent


def _run_episodes(env, agent, n_steps, n_episodes, max_episode_len=None, logger=None):
    assert (n_steps is None) != (n_episodes is None)
    logger = logger or logging.getLogger(__name__)
    scores, lengths = [], []
    timestep = 0
    reset = True
    test_r = episode_len = 0
    while True:
        if reset:
            obs = env.reset()
            test_r, episode_len = 0, 0
        a = agent.act(obs)
        obs, r, done, info = env.step(a)
        test_r += r
        episode_len += 1
        timestep += 1
        reset = done or episode_len == max_episode_len or info.get("needs_reset", False)
        agent.observe(obs, r, done, reset)
        if reset:
            logger.info("evaluation episode %s length:%s R:%s", len(scores), episode_len, test_r)
            scores.append(float(test_r))
            lengths.append(float(episode_len))
        if (len(scores) >= n_episodes) if n_steps is None else (timestep >= n_steps):
            break
    if not scores:   # all steps went into one unfinished episode
        scores.append(float(test_r))
        lengths.append(float(episode_len))
    return scores, lengths


def _batch_run_episodes(env, agent, n_steps, n_episodes, max_episode_len=None, logger=None):
    assert (n_steps is None) != (n_episodes is None)
    logger = logger or logging.getLogger(__name__)
    num_envs = env.num_envs
    finished_r, finished_len = {}, {}
    episode_indices = np.arange(num_envs, dtype="i")
    next_idx = num_envs
    episode_r = np.zeros(num_envs, dtype=np.float64)
    episode_len = np.zeros(num_envs, dtype="i")
    obss = env.reset()
    while True:
        actions = agent.batch_act(obss)
        obss, rs, dones, infos = env.step(actions)
        episode_r += rs
        episode_len += 1
        if max_episode_len is None:
            resets = np.zeros(num_envs, dtype=bool)
        else:
            resets = episode_len == max_episode_len
        resets = np.logical_or(resets, [info.get("needs_reset", False) for info in infos])
        end = np.logical_or(resets, dones)
        not_end = np.logical_not(end)
        for i in np.flatnonzero(end):
            finished_r[episode_indices[i]] = episode_r[i]
            finished_len[episode_indices[i]] = episode_len[i]
            episode_indices[i] = next_idx
            next_idx += 1
        episode_r[end] = 0
        episode_len[end] = 0
        first_unfinished = 0
        while first_unfinished in finished_r:
            first_unfinished += 1
        rets, lens = [], []
        if n_steps is not None:
            total = 0
            for idx in range(first_unfinished):
                total += finished_len[idx]
                if total > n_steps:
                    break
                rets.append(finished_r[idx])
                lens.append(finished_len[idx])
            stop = total >= n_steps
            if not stop:
                cur = np.where(episode_indices == first_unfinished)[0]
                if total + episode_len[cur] >= n_steps:
                    stop = True
                    if first_unfinished == 0:
                        rets.append(episode_r[cur])
                        lens.append(episode_len[cur])
        else:
            stop = first_unfinished >= n_episodes
            if stop:
                rets = [finished_r[i] for i in range(n_episodes)]
                lens = [finished_len[i] for i in range(n_episodes)]
        if stop:
            resets.fill(True)   # the agent must see every episode end
        agent.batch_observe(obss, rs, dones, resets)
        if stop:
            break
        obss = env.reset(not_end)
    for i, (ln, r) in enumerate(zip(lens, rets)):
        logger.info("evaluation episode %s length: %s R: %s", i, ln, r)
    return [float(r) for r in rets], [float(ln) for ln in lens]


def run_evaluation_episodes(env, agent, n_steps, n_episodes, max_episode_len=None, logger=None):
    with agent.eval_mode():
        return _run_episodes(env, agent, n_steps, n_episodes, max_episode_len, logger)


def batch_run_evaluation_episodes(env, agent, n_steps, n_episodes, max_episode_len=None,
                                  logger=None):
    with agent.eval_mode():
        return _batch_run_episodes(env, agent, n_steps, n_episodes, max_episode_len, logger)


def eval_performance(env, agent, n_steps, n_episodes, max_episode_len=None, logger=None):
    assert (n_steps is None) != (n_episodes is None)
    run = batch_run_evaluation_episodes if isinstance(env, VectorEnv) else run_evaluation_episodes
    scores, lengths = run(env, agent, n_steps, n_episodes, max_episode_len=max_episode_len,
                          logger=logger)
    return dict(
        episodes=len(scores), mean=statistics.mean(scores), median=statistics.median(scores),
        stdev=statistics.stdev(scores) if len(scores) >= 2 else 0.0,
        max=np.max(scores), min=np.min(scores), length_mean=statistics.mean(lengths),
        length_median=statistics.median(lengths),
        length_stdev=statistics.stdev(lengths) if len(lengths) >= 2 else 0,
        length_max=np.max(lengths), length_min=np.min(lengths))


_BASIC_COLUMNS = ("steps", "episodes", "elapsed", "mean", "median", "stdev", "max", "min")


def write_header(outdir, agent, env):
    env_stats = getattr(env, "get_statistics", lambda: [])
    names = (_BASIC_COLUMNS + tuple(k for k, _ in agent.get_statistics())
             + tuple(k for k, _ in env_stats()))
    with open(os.path.join(outdir, "scores.txt"), "w") as f:
        print("\t".join(names), file=f)


def record_stats(outdir, values):
    with open(os.path.join(outdir, "scores.txt"), "a+") as f:
        print("\t".join(str(x) for x in values), file=f)


class Evaluator(object):
    def __init__(self, agent, env, n_steps, n_episodes, eval_interval, outdir,
                 max_episode_len=None, step_offset=0, evaluation_hooks=(),
                 save_best_so_far_agent=True, logger=None, use_tensorboard=False):
        assert (n_steps is None) != (n_episodes is None), \
            "One of n_steps or n_episodes must be None."
        if use_tensorboard:
            raise NotImplementedError("TensorBoard logging is outside the hot path (SURVEY.md 2)")
        self.agent = agent
        self.env = env
        self.max_score = np.finfo(np.float32).min
        self.start_time = time.time()
        self.n_steps = n_steps
        self.n_episodes = n_episodes
        self.eval_interval = eval_interval
        self.outdir = outdir
        self.max_episode_len = max_episode_len
        self.step_offset = step_offset
        self.prev_eval_t = self.step_offset - self.step_offset % self.eval_interval
        self.evaluation_hooks = evaluation_hooks
        self.save_best_so_far_agent = save_best_so_far_agent
        self.logger = logger or logging.getLogger(__name__)
        self.env_get_stats = getattr(self.env, "get_statistics", lambda: [])
        self.env_clear_stats = getattr(self.env, "clear_statistics", lambda: None)
        write_header(self.outdir, self.agent, self.env)

    def evaluate_and_update_max_score(self, t, episodes):
        self.env_clear_stats()
        eval_stats = eval_performance(self.env, self.agent, self.n_steps, self.n_episodes,
                                      max_episode_len=self.max_episode_len, logger=self.logger)
        elapsed = time.time() - self.start_time
        agent_stats = self.agent.get_statistics()
        env_stats = self.env_get_stats()
        mean = eval_stats["mean"]
        record_stats(self.outdir, (t, episodes, elapsed, mean, eval_stats["median"],
                                   eval_stats["stdev"], eval_stats["max"], eval_stats["min"])
                     + tuple(v for _, v in agent_stats) + tuple(v for _, v in env_stats))
        for hook in self.evaluation_hooks:
            hook(env=self.env, agent=self.agent, evaluator=self, step=t, eval_stats=eval_stats,
                 agent_stats=agent_stats, env_stats=env_stats)
        if mean > self.max_score:
            self.logger.info("The best score is updated %s -> %s", self.max_score, mean)
            self.max_score = mean
            if self.save_best_so_far_agent:
                save_agent(self.agent, "best", self.outdir, self.logger)
        return mean

    def evaluate_if_necessary(self, t, episodes):
        if t >= self.prev_eval_t + self.eval_interval:
            score = self.evaluate_and_update_max_score(t, episodes)
            self.prev_eval_t = t - t % self.eval_interval
            return score
        return None


class LinearInterpolationHook(object):
    """Step hook that linearly interpolates a value over training
    (reference pfrl/experiments/hooks.py:26-57), e.g. a learning-rate decay."""

    support_train_agent = True
    support_train_agent_batch = True

    def __init__(self, total_steps, start_value, stop_value, setter):
        self.total_steps = total_steps
        self.start_value = start_value
        self.stop_value = stop_value
        self.setter = setter

    def __call__(self, env, agent, step):
        value = np.interp(step, [1, self.total_steps], [self.start_value, self.stop_value])
        self.setter(env, agent, value)
