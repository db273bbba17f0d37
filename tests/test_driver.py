"""Driver contract (reference tests/experiments_tests/test_train_agent_batch.py:
11-138, 180-233): call counts of batch_act / batch_observe / env.reset / env.step,
hook step numbers, evaluation cadence, needs_reset handling; VectorFrameStack
frame sharing (tests/wrappers_tests/test_vector_frame_stack.py:98-99)."""
import os
import tempfile
from unittest import mock

import numpy as np
import pytest
import torch

import pfrl_amd as pfrl
from pfrl_amd.envs import SerialVectorEnv


def _make_env(n_steps_per_episode, needs_reset_at=None):
    env = mock.Mock()
    state = {"t": 0}

    def reset():
        state["t"] = 0
        return ("state", 0)

    def step(a):
        state["t"] += 1
        done = state["t"] == n_steps_per_episode
        info = {}
        if needs_reset_at is not None and state["t"] == needs_reset_at:
            info["needs_reset"] = True
        return ("state", state["t"]), 0.5, done, info

    env.reset.side_effect = reset
    env.step.side_effect = step
    return env


def test_train_agent_batch_call_counts_and_hooks():
    outdir = tempfile.mkdtemp()
    agent = mock.Mock()
    agent.batch_act.side_effect = lambda obs: [0] * len(obs)
    agent.get_statistics.return_value = []
    envs = [_make_env(5), _make_env(5)]
    venv = SerialVectorEnv(envs)
    hook = mock.Mock()
    pfrl.experiments.train_agent_batch(agent=agent, env=venv, steps=10, outdir=outdir,
                                       step_hooks=[hook])
    # 2 envs x 5 iterations = 10 steps
    assert agent.batch_act.call_count == 5
    assert agent.batch_observe.call_count == 5
    for e in envs:
        assert e.step.call_count == 5
        assert e.reset.call_count == 1   # episode ends exactly when training ends
    assert hook.call_count == 10
    for i, call in enumerate(hook.call_args_list):
        assert call[0][0] is venv and call[0][1] is agent and call[0][2] == i + 1
    agent.save.assert_called_once_with(os.path.join(outdir, "10_finish"))


def test_train_agent_batch_needs_reset_and_masks():
    outdir = tempfile.mkdtemp()
    agent = mock.Mock()
    agent.batch_act.side_effect = lambda obs: [0] * len(obs)
    agent.get_statistics.return_value = []
    envs = [_make_env(100, needs_reset_at=2), _make_env(3)]
    venv = SerialVectorEnv(envs)
    pfrl.experiments.train_agent_batch(agent=agent, env=venv, steps=12, outdir=outdir)
    resets = [c[0][3] for c in agent.batch_observe.call_args_list]
    dones = [c[0][2] for c in agent.batch_observe.call_args_list]
    assert [bool(r[0]) for r in resets] == [False, True, False, True, False, True]
    assert [bool(d[1]) for d in dones] == [False, False, True, False, False, True]
    assert envs[0].reset.call_count == 3   # initial + two needs_reset restarts (third ends run)
    assert envs[1].reset.call_count == 2


def test_train_agent_batch_with_evaluation_writes_scores():
    outdir = tempfile.mkdtemp()
    agent = mock.MagicMock()
    agent.batch_act.side_effect = lambda obs: [0] * len(obs)
    agent.get_statistics.return_value = [("average_q", 1.5)]
    venv = SerialVectorEnv([_make_env(4), _make_env(4)])
    eval_env = SerialVectorEnv([_make_env(4), _make_env(4)])
    _, hist = pfrl.experiments.train_agent_batch_with_evaluation(
        agent=agent, env=venv, steps=16, eval_n_steps=None, eval_n_episodes=2, eval_interval=8,
        outdir=outdir, eval_env=eval_env)
    rows = open(os.path.join(outdir, "scores.txt")).read().strip().split("\n")
    assert rows[0].split("\t")[:8] == ["steps", "episodes", "elapsed", "mean", "median", "stdev",
                                       "max", "min"]
    assert rows[0].split("\t")[8] == "average_q"
    assert len(rows) == 3 and len(hist) == 2
    assert float(rows[1].split("\t")[3]) == 2.0   # 4 steps x 0.5 reward
    agent.save.assert_any_call(os.path.join(outdir, "best"))


def test_vector_frame_stack_shares_frames_by_identity():
    from pfrl_amd.wrappers import VectorFrameStack

    class Env:
        def __init__(self, seed):
            self.rs = np.random.RandomState(seed)

        def reset(self):
            return self.rs.randint(0, 256, size=(1, 8, 8)).astype(np.uint8)

        def step(self, a):
            return self.rs.randint(0, 256, size=(1, 8, 8)).astype(np.uint8), 0.0, False, {}

        def close(self):
            pass

    venv = VectorFrameStack(SerialVectorEnv([Env(0), Env(1)]), 4, stack_axis=0)
    o0 = venv.reset()
    assert np.asarray(o0[0]).shape == (4, 8, 8)
    assert all(f is o0[0]._frames[0] for f in o0[0]._frames)   # reset = k copies
    o1, _, _, _ = venv.step([0, 0])
    o2, _, _, _ = venv.step([0, 0])
    for i in range(3):
        assert o2[0]._frames[i] is o1[0]._frames[i + 1]          # shared, not copied
    np.testing.assert_array_equal(np.asarray(o2[1])[:3], np.asarray(o1[1])[1:])


def test_cartpole_dqn_train_agent_matches_reference(tmp_path):
    """BASELINE configs[0]: examples/gym/train_dqn_gym.py settings, one env, host path,
    through pfrl_amd.experiments.train_agent -- against the trace the reference's
    train_agent + DQN produced on the same CartPole (tests/golden/make_golden.py)."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents, experiments, explorers, q_functions, replay_buffers
    from pfrl_amd.envs import CartPoleEnv

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "agent_trace_cartpole_dqn.npz"))
    pfrl.utils.set_random_seed(0)
    env = CartPoleEnv(seed=0)
    torch.manual_seed(77)
    q = q_functions.FCStateQFunctionWithDiscreteAction(4, 2, n_hidden_channels=100,
                                                       n_hidden_layers=2)
    opt = torch.optim.Adam(q.parameters())
    rbuf = replay_buffers.ReplayBuffer(5 * 10 ** 5)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 1000, env.action_space.sample)
    ag = agents.DQN(q, opt, rbuf, gpu=-1, gamma=0.99, explorer=ex, replay_start_size=200,
                    target_update_interval=100, update_interval=1, minibatch_size=32,
                    target_update_method="hard", soft_update_tau=1e-2)
    actions, losses = [], []
    orig_act = ag.act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append(int(a))
        return a

    ag.act = spy_act
    orig_core = ag._update_from_batch

    def spy_core(*a, **kw):
        orig_core(*a, **kw)
        losses.append(float(ag.loss_record.values()[-1]))

    ag._update_from_batch = spy_core
    experiments.train_agent(ag, env, 1500, str(tmp_path), max_episode_len=200)
    assert os.path.isdir(os.path.join(str(tmp_path), "1500_finish"))
    np.testing.assert_array_equal(np.asarray(actions), g["actions"])
    np.testing.assert_allclose(np.asarray(losses), g["losses"], rtol=1e-5, atol=1e-6)
    flat = np.concatenate([p.detach().numpy().ravel() for p in q.parameters()])
    np.testing.assert_allclose(flat, g["final_params"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(env.state, g["final_state"], rtol=0, atol=0)
    np.testing.assert_allclose([float(v) for _, v in ag.get_statistics()], g["stats"], rtol=1e-5,
                               atol=1e-6)


def _scripted_env(reset_obs, transitions):
    env = mock.Mock()
    env.reset.side_effect = list(reset_obs)
    env.step.side_effect = list(transitions)
    return env


def test_train_agent_call_contract(tmp_path):
    """Reference tests/experiments_tests/test_train_agent.py:11-50: five steps to a
    terminal state -- act/observe/step five times, one reset, hooks see steps 1..5."""
    agent, hook = mock.Mock(), mock.Mock()
    env = _scripted_env(["s0"], [("s1", 0, False, {}), ("s2", 0, False, {}),
                                 ("s3", -0.5, False, {}), ("s4", 0, False, {}),
                                 ("s5", 1, True, {})])
    hist = pfrl.experiments.train_agent(agent=agent, env=env, steps=5, outdir=str(tmp_path),
                                        step_hooks=[hook])
    assert hist == []
    assert agent.act.call_count == agent.observe.call_count == env.step.call_count == 5
    assert env.reset.call_count == 1
    assert agent.observe.call_args_list[4][0][2] is True          # done at s5
    assert [c[0][2] for c in hook.call_args_list] == [1, 2, 3, 4, 5]
    assert all(c[0][0] is env and c[0][1] is agent for c in hook.call_args_list)
    agent.save.assert_called_once()                                 # <t>_finish


def test_train_agent_needs_reset_and_max_episode_len(tmp_path):
    """:52-93 -- info['needs_reset'] ends the episode with done=False, reset=True and a
    second env.reset(); max_episode_len does the same by step count."""
    agent = mock.Mock()
    env = _scripted_env(["s0", "s4"], [("s1", 0, False, {}), ("s2", 0, False, {}),
                                       ("s3", 0, False, {"needs_reset": True}),
                                       ("s5", -0.5, False, {}), ("s6", 0, False, {}),
                                       ("s7", 1, True, {})])
    pfrl.experiments.train_agent(agent=agent, env=env, steps=5, outdir=str(tmp_path))
    assert env.reset.call_count == 2 and env.step.call_count == 5
    third = agent.observe.call_args_list[2][0]
    assert third[2] is False and third[3] is True
    agent2 = mock.Mock()
    env2 = _scripted_env(["a", "b", "c"], [("x", 0, False, {})] * 6)
    pfrl.experiments.train_agent(agent=agent2, env=env2, steps=6, outdir=str(tmp_path),
                                 max_episode_len=2)
    resets = [bool(c[0][3]) for c in agent2.observe.call_args_list]
    assert resets == [False, True, False, True, False, True]
    assert env2.reset.call_count == 3


def test_train_agent_saves_on_exception(tmp_path):
    agent = mock.Mock()
    env = _scripted_env(["s0"], [("s1", 0, False, {}), RuntimeError("boom")])
    try:
        pfrl.experiments.train_agent(agent=agent, env=env, steps=5, outdir=str(tmp_path))
    except RuntimeError:
        pass
    else:
        raise AssertionError("exception swallowed")
    assert agent.save.call_args[0][0].endswith("1_except")


def test_train_agent_with_evaluation_rejects_unsupported_hook(tmp_path):
    class Hook:
        support_train_agent = False

    import pytest

    with pytest.raises(ValueError):
        pfrl.experiments.train_agent_with_evaluation(
            agent=mock.Mock(), env=mock.Mock(), steps=1, eval_n_steps=1, eval_n_episodes=None,
            eval_interval=1, outdir=str(tmp_path), evaluation_hooks=[Hook()])


def test_prepare_output_dir(tmp_path, monkeypatch):
    """Reference tests/experiments_tests/test_prepare_output_dir.py:70-130: files written,
    explicit and generated ids, backup of an existing directory."""
    import argparse
    import json
    import subprocess

    work = tmp_path / "work"
    work.mkdir()
    monkeypatch.chdir(work)
    args = argparse.Namespace(a=1, b="two")
    # not under git: the id is the timestamp
    d = pfrl.experiments.prepare_output_dir(args, basedir=str(tmp_path / "out"), argv=["x", "--y"])
    assert os.path.dirname(d) == str(tmp_path / "out")
    assert json.load(open(os.path.join(d, "args.txt"))) == {"a": 1, "b": "two"}
    assert open(os.path.join(d, "command.txt")).read() == "x --y"
    assert "PATH" in json.load(open(os.path.join(d, "environ.txt")))
    assert len(open(os.path.join(d, "start.txt")).read().splitlines()) == 1
    assert not os.path.exists(os.path.join(d, "git-head.txt"))
    # explicit id, second start: appended timestamp and a backup of the first state
    d1 = pfrl.experiments.prepare_output_dir({"k": 3}, basedir=str(tmp_path / "out"), exp_id="run")
    d2 = pfrl.experiments.prepare_output_dir({"k": 4}, basedir=str(tmp_path / "out"), exp_id="run")
    assert d1 == d2 and len(open(os.path.join(d2, "start.txt")).read().splitlines()) == 2
    assert json.load(open(os.path.join(d2, "args.txt"))) == {"k": 4}
    assert any(n.startswith("run.") and n.endswith(".backup")
               for n in os.listdir(str(tmp_path / "out")))
    # under git: deterministic id from HEAD, the diff and argv; git records are saved
    env = dict(os.environ, GIT_AUTHOR_NAME="t", GIT_AUTHOR_EMAIL="t@t", GIT_COMMITTER_NAME="t",
               GIT_COMMITTER_EMAIL="t@t")
    subprocess.check_call(["git", "init", "-q"], cwd=str(work))
    (work / "f.txt").write_text("hello")
    subprocess.check_call(["git", "add", "f.txt"], cwd=str(work))
    subprocess.check_call(["git", "commit", "-q", "-m", "init"], cwd=str(work), env=env)
    assert pfrl.experiments.is_under_git_control()
    i1 = pfrl.experiments.generate_exp_id(prefix="p", argv=["a"])
    assert i1 == pfrl.experiments.generate_exp_id(prefix="p", argv=["a"]) and i1.startswith("p-")
    assert i1 != pfrl.experiments.generate_exp_id(prefix="p", argv=["b"])
    d3 = pfrl.experiments.prepare_output_dir(args, basedir=str(tmp_path / "out2"))
    for name in ("git-head.txt", "git-status.txt", "git-log.txt", "git-diff.txt"):
        assert os.path.exists(os.path.join(d3, name))


@pytest.mark.parametrize("name,fire,flicker,scale", [("plain", False, False, False),
                                                    ("fire_flicker_scaled", True, True, True)])
def test_atari_wrappers_follow_reference_on_scripted_game(name, fire, flicker, scale):
    """tests/golden/atari_wrappers.npz: the reference's wrapper stack over tests/_fake_ale.py.
    Same observations (checksum + newest frame), rewards, dones, needs_reset flags, the same
    number of real game resets / raw steps, and the same k-1 frames shared by identity."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from _fake_ale import FakeALE

    from pfrl_amd.wrappers import ContinuingTimeLimit, atari_wrappers as aw

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "atari_wrappers.npz"))
    game = FakeALE(seed=3)
    env = ContinuingTimeLimit(game, max_episode_steps=90)
    env = aw.MaxAndSkipEnv(aw.NoopResetEnv(env, noop_max=5), skip=4)
    env = aw.EpisodicLifeEnv(env)
    if fire:
        env = aw.FireResetEnv(env)
    if scale:
        env = aw.ScaledFloatFrame(env)
    env = aw.ClipRewardEnv(env)
    if flicker:
        env = aw.FlickerFrame(env)
    env = aw.FrameStack(env, 4, channel_order="chw")
    np.testing.assert_array_equal(env.observation_space.low, g[name + "_space_low"])
    np.testing.assert_array_equal(env.observation_space.high, g[name + "_space_high"])
    assert env.observation_space.dtype == g[name + "_space_high"].dtype
    rs = np.random.RandomState(11)
    obs = env.reset()
    assert isinstance(obs, aw.LazyFrames) and all(f is obs._frames[0] for f in obs._frames)
    prev = obs
    for t in range(len(g[name + "_reward"])):
        obs, r, done, info = env.step(int(rs.randint(3)))
        assert float(np.asarray(obs, dtype=np.float64).sum()) == g[name + "_obs_sum"][t], t
        np.testing.assert_array_equal(np.asarray(obs)[-1], g[name + "_newest"][t])
        assert (float(r), bool(done), bool(info.get("needs_reset", False))) == (
            g[name + "_reward"][t], g[name + "_done"][t], g[name + "_needs_reset"][t]), t
        shared = sum(a is b for a, b in zip(obs._frames[:-1], prev._frames[1:]))
        assert shared == g[name + "_shared"][t] == 3, t
        if done or info.get("needs_reset", False):
            obs = env.reset()
        prev = obs
        assert (game.n_resets, game.n_steps) == (g[name + "_resets"][t], g[name + "_steps"][t]), t
    assert set(np.unique(g[name + "_reward"])) <= {-1.0, 0.0, 1.0}


def test_atari_wrappers_fail_at_the_point_of_use_without_their_dependencies():
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from _fake_ale import FakeALE

    from pfrl_amd.wrappers import atari_wrappers as aw

    if aw.cv2 is None:
        with pytest.raises(RuntimeError):
            aw.WarpFrame(FakeALE(0))
        with pytest.raises(RuntimeError):
            aw.wrap_deepmind(FakeALE(0))
    try:
        import gym  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            aw.make_atari("PongNoFrameskip-v4")


@pytest.mark.gpu
def test_batch_evaluation_on_the_device_path(tmp_path):
    """train_agent_batch_with_evaluation with a DEVICE eval env (frames in HBM,
    observations as frame-slot refs).  During evaluation ``batch_act`` runs in eval mode --
    greedy w.r.t. the current Q-network, no exploration draw, nothing appended to the replay
    buffer -- and scores.txt has the reference's columns (pfrl/experiments/evaluator.py:375-393:
    eight basic columns, then the agent's statistics) and one row per evaluation."""
    import numpy as np
    import torch

    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DiscreteActionValueHead

    dev = torch.device("cuda:0")
    pfrl.utils.set_random_seed(0)
    N = 4

    def make_env(seed):
        store = DeviceFrameStore(4096, (12, 12), torch.uint8, dev, stack=4)
        return SyntheticAtariVectorEnv(N, store=store, seed=seed, n_actions=5, p_done=0.05)

    env, eval_env = make_env(1), make_env(2)
    torch.manual_seed(3)
    q = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
                            torch.nn.Linear(32, 5), DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=1e-3)
    rbuf = replay_buffers.ReplayBuffer(500)
    ex = explorers.ConstantEpsilonGreedy(1.0, lambda: np.random.randint(5))   # training: all random
    phi = lambda x: np.asarray(x, dtype=np.float32) / 255   # noqa: E731
    ag = agents.DQN(q, opt, rbuf, 0.99, ex, gpu=0, replay_start_size=32, minibatch_size=8,
                    update_interval=4, target_update_interval=40, phi=phi)
    eval_calls = []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        if not ag.training:
            with torch.no_grad():
                want = ag.model(pfrl.utils.batch_states(obs, ag.device, phi)).greedy_actions
            eval_calls.append((np.asarray(a).copy(), want.cpu().numpy(), len(rbuf),
                               np.random.get_state()[2]))
        return a

    ag.batch_act = spy_act
    outdir = str(tmp_path)
    pfrl.experiments.train_agent_batch_with_evaluation(
        ag, env, steps=240, eval_n_steps=None, eval_n_episodes=6, eval_interval=80, outdir=outdir,
        eval_env=eval_env, log_interval=10 ** 9)
    assert len(eval_calls) > 20
    for got, want, _, _ in eval_calls:
        np.testing.assert_array_equal(got, want)          # greedy, never the explorer's action
    # evaluation neither appended transitions nor drew from the explorer's stream in between
    by_len = {}
    for _, _, n, pos in eval_calls:
        by_len.setdefault(n, set()).add(pos)
    assert all(len(v) == 1 for v in by_len.values())
    lines = open(os.path.join(outdir, "scores.txt")).read().strip().split("\n")
    header = lines[0].split("\t")
    assert header[:8] == ["steps", "episodes", "elapsed", "mean", "median", "stdev", "max", "min"]
    assert header[8:] == [name for name, _ in ag.get_statistics()]
    rows = [ln.split("\t") for ln in lines[1:]]
    assert len(rows) == 3 and all(len(r) == len(header) for r in rows)
    assert [int(r[0]) for r in rows] == [80, 160, 240]
    assert os.path.isdir(os.path.join(outdir, "best")) and os.path.isdir(os.path.join(outdir, "240_finish"))
    assert ag.training
