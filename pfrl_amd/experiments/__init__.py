from pfrl_amd.experiments.evaluator import (Evaluator, LinearInterpolationHook,  # NOQA
                                            eval_performance)
from pfrl_amd.experiments.train_agent_batch import (save_agent, train_agent_batch,  # NOQA
                                                    train_agent_batch_with_evaluation)
