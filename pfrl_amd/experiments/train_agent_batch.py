"""Batched training driver.

Same contract as ``pfrl.experiments.train_agent_batch``
(/root/reference/pfrl/experiments/train_agent_batch.py:10-154): the loop is
``batch_act -> env.step -> batch_observe -> env.reset(not_end)``; ``resets`` is
``episode_len == max_episode_len`` or the env's ``needs_reset`` info flag
(:74-80); every env step advances ``t`` by one and fires the step hooks
(:98-104); the agent is saved on exceptions (``_except``) and at the end
(``_finish``).  Observations are passed through untouched, so a device
VectorEnv's ``DeviceObsBatch`` reaches the agent without a host copy.
"""
import logging
import os
from collections import deque

import numpy as np


def save_agent(agent, t, outdir, logger, suffix=""):
    dirname = os.path.join(outdir, "{}{}".format(t, suffix))
    agent.save(dirname)
    logger.info("Saved the agent to %s", dirname)


class _EpisodeBook:
    """Per-env episode accounting of the batched loop: running return and length, finished
    episode count, and the window of recent returns the progress line averages."""

    def __init__(self, num_envs, return_window_size):
        self.ret = np.zeros(num_envs, dtype=np.float64)
        self.length = np.zeros(num_envs, dtype="i")
        self.count = np.zeros(num_envs, dtype="i")
        self.recent_returns = deque(maxlen=return_window_size)

    def after_step(self, rewards, dones, infos, max_episode_len):
        """Fold one env step in; returns (resets, end): which envs are cut without being terminal
        (length limit or the env's own ``needs_reset`` flag, reference :74-80) and which episodes
        are over for either reason."""
        self.ret += rewards
        self.length += 1
        asked = np.asarray([info.get("needs_reset", False) for info in infos], dtype=bool)
        if max_episode_len is None:
            resets = asked
        else:
            resets = np.logical_or(self.length == max_episode_len, asked)
        end = np.logical_or(resets, dones)
        self.count += end
        self.recent_returns.extend(self.ret[end])
        return resets, end

    def start_new_episodes(self, end):
        self.ret[end] = 0
        self.length[end] = 0

    @property
    def episodes(self):
        return np.sum(self.count)

    def progress(self):
        if not self.recent_returns:
            return np.nan, np.nan
        return self.recent_returns[-1], np.mean(self.recent_returns)


def train_agent_batch(agent, env, steps, outdir, checkpoint_freq=None, log_interval=None,
                      max_episode_len=None, step_offset=0, evaluator=None, successful_score=None,
                      step_hooks=(), return_window_size=100, logger=None):
    logger = logger or logging.getLogger(__name__)
    num_envs = env.num_envs
    book = _EpisodeBook(num_envs, return_window_size)
    eval_stats_history = []

    obss = env.reset()
    t = step_offset
    if hasattr(agent, "t"):
        agent.t = step_offset
    try:
        while True:
            actions = agent.batch_act(obss)
            obss, rs, dones, infos = env.step(actions)
            resets, end = book.after_step(rs, dones, infos, max_episode_len)
            agent.batch_observe(obss, rs, dones, resets)

            # one global step per env: checkpoints and hooks see every value of t (:98-104)
            for t in range(t + 1, t + num_envs + 1):
                if checkpoint_freq and t % checkpoint_freq == 0:
                    save_agent(agent, t, outdir, logger, suffix="_checkpoint")
                for hook in step_hooks:
                    hook(env, agent, t)

            if log_interval is not None and t >= log_interval and t % log_interval < num_envs:
                last_r, average_r = book.progress()
                logger.info("outdir:%s step:%s episode:%s last_R: %s average_R:%s", outdir, t,
                            book.episodes, last_r, average_r)
                logger.info("statistics: %s", agent.get_statistics())
            if evaluator:
                eval_score = evaluator.evaluate_if_necessary(t=t, episodes=book.episodes)
                if eval_score is not None:
                    eval_stats_history.append(dict(agent.get_statistics(), eval_score=eval_score))
                    if successful_score is not None and evaluator.max_score >= successful_score:
                        break
            if t >= steps:
                break

            book.start_new_episodes(end)
            obss = env.reset(np.logical_not(end))

    except (Exception, KeyboardInterrupt):
        save_agent(agent, t, outdir, logger, suffix="_except")
        env.close()
        if evaluator:
            evaluator.env.close()
        raise
    else:
        save_agent(agent, t, outdir, logger, suffix="_finish")

    return eval_stats_history


def train_agent_batch_with_evaluation(agent, env, steps, eval_n_steps, eval_n_episodes,
                                      eval_interval, outdir, checkpoint_freq=None,
                                      max_episode_len=None, step_offset=0, eval_max_episode_len=None,
                                      return_window_size=100, eval_env=None, log_interval=None,
                                      successful_score=None, step_hooks=(), evaluation_hooks=(),
                                      save_best_so_far_agent=True, use_tensorboard=False,
                                      logger=None):
    """train_agent_batch + periodic evaluation (reference :157-263)."""
    from pfrl_amd.experiments.evaluator import Evaluator

    logger = logger or logging.getLogger(__name__)
    for hook in evaluation_hooks:
        if not getattr(hook, "support_train_agent_batch", True):
            raise ValueError(
                "{} does not support train_agent_batch_with_evaluation().".format(hook))
    os.makedirs(outdir, exist_ok=True)
    if eval_env is None:
        eval_env = env
    if eval_max_episode_len is None:
        eval_max_episode_len = max_episode_len
    evaluator = Evaluator(agent=agent, n_steps=eval_n_steps, n_episodes=eval_n_episodes,
                          eval_interval=eval_interval, outdir=outdir,
                          max_episode_len=eval_max_episode_len, env=eval_env,
                          step_offset=step_offset, evaluation_hooks=evaluation_hooks,
                          save_best_so_far_agent=save_best_so_far_agent,
                          use_tensorboard=use_tensorboard, logger=logger)
    eval_stats_history = train_agent_batch(
        agent, env, steps, outdir, checkpoint_freq=checkpoint_freq,
        max_episode_len=max_episode_len, step_offset=step_offset, evaluator=evaluator,
        successful_score=successful_score, return_window_size=return_window_size,
        log_interval=log_interval, step_hooks=step_hooks, logger=logger)
    return agent, eval_stats_history
