"""The drop-in boundary as a list of names (SURVEY.md 8b and the modules the hot-path example
scripts import), and a JSON-able description of a callable's signature.  Shared by
tests/golden/make_golden.py (which describes the REFERENCE) and tests/test_host_logic.py (which
describes pfrl_amd and compares)."""
API_SURFACE = {
    "replay_buffers": ["ReplayBuffer", "PrioritizedReplayBuffer", "EpisodicReplayBuffer",
                       "PrioritizedEpisodicReplayBuffer", "PersistentReplayBuffer",
                       "PersistentEpisodicReplayBuffer", "PriorityWeightError"],
    "replay_buffer": ["batch_experiences", "batch_recurrent_experiences", "ReplayUpdater",
                      "random_subseq"],
    "utils": ["batch_states", "set_random_seed", "clip_l2_grad_norm_"],
    "utils.copy_param": ["synchronize_parameters", "soft_copy_param", "copy_param"],
    "utils.recurrent": ["one_step_forward", "pack_and_forward", "mask_recurrent_state_at",
                        "get_recurrent_state_at", "concatenate_recurrent_states",
                        "flatten_sequences_time_first", "recurrent_state_from_numpy"],
    "agents": ["DQN", "DoubleDQN", "CategoricalDQN", "CategoricalDoubleDQN", "IQN", "AL", "PAL",
               "DoublePAL", "DPP", "PPO", "A2C", "SoftActorCritic", "TD3", "DDPG"],
    "experiments": ["train_agent_batch", "train_agent_batch_with_evaluation", "train_agent",
                    "train_agent_with_evaluation", "eval_performance", "prepare_output_dir",
                    "LinearInterpolationHook"],
    "envs": ["MultiprocessVectorEnv", "SerialVectorEnv"],
    "envs.abc": ["ABC"],
    "agents.iqn": ["ImplicitQuantileQFunction", "RecurrentImplicitQuantileQFunction",
                   "CosineBasisLinear", "cosine_basis_functions",
                   "compute_eltwise_huber_quantile_loss"],
    "agents.ppo": ["_make_dataset", "_make_dataset_recurrent", "_yield_minibatches",
                   "_limit_sequence_length",
                   "_yield_subset_of_sequences_with_fixed_number_of_items",
                   "_add_advantage_and_value_target_to_episode"],
    "agents.dqn": ["compute_value_loss", "compute_weighted_value_loss"],
    "experiments.evaluation_hooks": ["EvaluationHook", "OptunaPrunerHook"],
    "testing": ["torch_assert_allclose"],
    "wrappers": ["VectorFrameStack", "ContinuingTimeLimit", "CastObservationToFloat32",
                 "ScaleReward", "RandomizeAction", "NormalizeActionSpace"],
    "wrappers.atari_wrappers": ["FrameStack", "MaxAndSkipEnv", "NoopResetEnv", "wrap_deepmind",
                                "make_atari", "LazyFrames"],
    "explorers": ["LinearDecayEpsilonGreedy", "ConstantEpsilonGreedy", "ExponentialDecayEpsilonGreedy",
                  "Greedy", "AdditiveGaussian", "AdditiveOU", "Boltzmann"],
    "nn": ["EmpiricalNormalization", "FactorizedNoisyLinear", "to_factorized_noisy", "MLP",
           "LargeAtariCNN", "SmallAtariCNN", "BoundByTanh", "ConcatObsAndAction", "Lambda", "Branched",
           "RecurrentSequential", "RecurrentBranched"],
    "q_functions": ["DuelingDQN", "DistributionalDuelingDQN", "FCStateQFunctionWithDiscreteAction",
                    "DistributionalFCStateQFunctionWithDiscreteAction", "DiscreteActionValueHead",
                    "FCQuadraticStateQFunction"],
    "policies": ["SoftmaxCategoricalHead", "GaussianHeadWithStateIndependentCovariance",
                 "GaussianHeadWithDiagonalCovariance", "GaussianHeadWithFixedCovariance",
                 "DeterministicHead"],
    "collections.random_access_queue": ["RandomAccessQueue"],
    "collections.persistent_collections": ["PersistentRandomAccessQueue"],
    "collections.prioritized": ["PrioritizedBuffer"],
    "optimizers": ["RMSpropEpsInsideSqrt", "SharedRMSpropEpsInsideSqrt"],
    "action_value": ["DiscreteActionValue", "DistributionalDiscreteActionValue",
                     "QuantileDiscreteActionValue", "QuadraticActionValue", "SingleActionValue"],
    "functions.lower_triangular_matrix": ["lower_triangular_matrix"],
    "functions.bound_by_tanh": ["bound_by_tanh"],
    "utils.env_modifiers": ["make_rendered", "make_timestep_limited", "make_action_filtered",
                            "make_reward_filtered"],
    "utils.conjugate_gradient": ["conjugate_gradient"],
}
API_METHODS = ["append", "sample", "sample_episodes", "update_errors", "stop_current_episode", "save",
               "load", "batch_act", "batch_observe", "act", "observe", "get_statistics",
               "update_if_necessary", "select_action", "step", "reset", "forward", "experience"]


def describe_signature(obj):
    """[[name, kind, default-or-marker], ...] with non-literal defaults reduced to a marker."""
    import inspect

    try:
        sig = inspect.signature(obj)
    except (TypeError, ValueError):
        return None
    out = []
    for prm in sig.parameters.values():
        d = prm.default
        if d is inspect.Parameter.empty:
            d = "<required>"
        elif d is None or isinstance(d, (bool, int, float, str)):
            d = repr(d)
        elif isinstance(d, tuple) and all(isinstance(x, (bool, int, float, str)) for x in d):
            d = repr(d)
        else:
            d = "<object>"
        out.append([prm.name, prm.kind.name, d])
    return out


def describe_api(root):
    """Signatures of API_SURFACE under the package ``root`` (the reference here, pfrl_amd in the test)."""
    import importlib
    import inspect

    desc = {}
    for mod_name, names in API_SURFACE.items():
        mod = importlib.import_module(root + "." + mod_name)
        for name in names:
            obj = getattr(mod, name)
            key = mod_name + "." + name
            desc[key] = describe_signature(obj)
            if inspect.isclass(obj):
                for meth in API_METHODS:
                    if callable(getattr(obj, meth, None)):
                        desc[key + "." + meth] = describe_signature(getattr(obj, meth))
    return desc
