from pfrl_amd.experiments.evaluator import (Evaluator, LinearInterpolationHook,  # NOQA
                                            eval_performance)
from pfrl_amd.experiments.train_agent_batch import (save_agent, train_agent_batch,  # NOQA
                                                    train_agent_batch_with_evaluation)
from pfrl_amd.experiments.train_agent import (ask_and_save_agent_replay_buffer,  # NOQA
                                              save_agent_replay_buffer, train_agent,
                                              train_agent_with_evaluation)
from pfrl_amd.experiments.prepare_output_dir import (generate_exp_id,  # NOQA
                                                     is_under_git_control, prepare_output_dir)
from pfrl_amd.experiments.hooks import EvaluationHook, StepHook  # NOQA
from pfrl_amd.experiments import evaluation_hooks  # NOQA
from pfrl_amd.experiments.evaluation_hooks import OptunaPrunerHook  # NOQA
