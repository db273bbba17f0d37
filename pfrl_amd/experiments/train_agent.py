"""Single-environment training loop (reference pfrl/experiments/train_agent.py:25-113,
:116-218) -- BASELINE configs[0] plumbing (CartPole, one env, host path).

Order of events per step, as in the reference: ``agent.act`` -> ``env.step`` ->
``agent.observe(obs, r, done, reset)`` with ``reset = episode_len ==
max_episode_len or info['needs_reset']`` -> step hooks -> episode bookkeeping ->
evaluator -> ``env.reset`` -> checkpoint.  The model is saved as ``<t>_except``
before an exception propagates and as ``<t>_finish`` at the end.
"""
import logging
import os

from pfrl_amd.experiments.train_agent_batch import save_agent


def save_agent_replay_buffer(agent, t, outdir, suffix="", logger=None):
    logger = logger or logging.getLogger(__name__)
    path = os.path.join(outdir, "{}{}.replay.pkl".format(t, suffix))
    agent.replay_buffer.save(path)
    logger.info("Saved the current replay buffer to %s", path)


def ask_and_save_agent_replay_buffer(agent, t, outdir, suffix=""):
    """Offer, on the terminal, to dump the replay buffer next to the agent (reference :15-21)."""
    from pfrl_amd.utils.ask_yes_no import ask_yes_no

    if hasattr(agent, "replay_buffer") and ask_yes_no(
            "Replay buffer has {} transitions. Do you save them to a file?".format(
                len(agent.replay_buffer))):
        save_agent_replay_buffer(agent, t, outdir, suffix=suffix)


def train_agent(agent, env, steps, outdir, checkpoint_freq=None, max_episode_len=None,
                step_offset=0, evaluator=None, successful_score=None, step_hooks=(),
                eval_during_episode=False, logger=None):
    logger = logger or logging.getLogger(__name__)
    t = step_offset
    if hasattr(agent, "t"):
        agent.t = step_offset
    history = []
    n_episodes = 0
    ep_return, ep_len = 0, 0
    obs = env.reset()
    try:
        while t < steps:
            obs, r, done, info = env.step(agent.act(obs))
            t += 1
            ep_return += r
            ep_len += 1
            reset = ep_len == max_episode_len or info.get("needs_reset", False)
            agent.observe(obs, r, done, reset)
            for hook in step_hooks:
                hook(env, agent, t)
            ended = done or reset or t == steps
            if ended:
                logger.info("outdir:%s step:%s episode:%s R:%s", outdir, t, n_episodes, ep_return)
                logger.info("statistics:%s", agent.get_statistics())
                n_episodes += 1
            if evaluator is not None and (ended or eval_during_episode):
                score = evaluator.evaluate_if_necessary(t=t, episodes=n_episodes)
                if score is not None:
                    stats = dict(agent.get_statistics())
                    stats["eval_score"] = score
                    history.append(stats)
                if successful_score is not None and evaluator.max_score >= successful_score:
                    break
            if ended:
                if t == steps:
                    break
                ep_return, ep_len = 0, 0
                obs = env.reset()
            if checkpoint_freq and t % checkpoint_freq == 0:
                save_agent(agent, t, outdir, logger, suffix="_checkpoint")
    except (Exception, KeyboardInterrupt):
        save_agent(agent, t, outdir, logger, suffix="_except")
        raise
    save_agent(agent, t, outdir, logger, suffix="_finish")
    return history


def train_agent_with_evaluation(agent, env, steps, eval_n_steps, eval_n_episodes, eval_interval,
                                outdir, checkpoint_freq=None, train_max_episode_len=None,
                                step_offset=0, eval_max_episode_len=None, eval_env=None,
                                successful_score=None, step_hooks=(), evaluation_hooks=(),
                                save_best_so_far_agent=True, use_tensorboard=False,
                                eval_during_episode=False, logger=None):
    """train_agent + periodic evaluation; returns (agent, eval_stats_history)."""
    from pfrl_amd.experiments.evaluator import Evaluator

    logger = logger or logging.getLogger(__name__)
    for hook in evaluation_hooks:
        if not getattr(hook, "support_train_agent", True):
            raise ValueError("{} does not support train_agent_with_evaluation().".format(hook))
    os.makedirs(outdir, exist_ok=True)
    if eval_env is None:
        assert not eval_during_episode, (
            "To run evaluation during training episodes, you need to specify `eval_env`"
            " that is independent from `env`.")
        eval_env = env
    if eval_max_episode_len is None:
        eval_max_episode_len = train_max_episode_len
    evaluator = Evaluator(agent=agent, n_steps=eval_n_steps, n_episodes=eval_n_episodes,
                          eval_interval=eval_interval, outdir=outdir,
                          max_episode_len=eval_max_episode_len, env=eval_env,
                          step_offset=step_offset, evaluation_hooks=evaluation_hooks,
                          save_best_so_far_agent=save_best_so_far_agent,
                          use_tensorboard=use_tensorboard, logger=logger)
    history = train_agent(agent, env, steps, outdir, checkpoint_freq=checkpoint_freq,
                          max_episode_len=train_max_episode_len, step_offset=step_offset,
                          evaluator=evaluator, successful_score=successful_score,
                          step_hooks=step_hooks, eval_during_episode=eval_during_episode,
                          logger=logger)
    return agent, history
