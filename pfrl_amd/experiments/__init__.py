from pfrl_amd.experiments.train_agent_batch import save_agent, train_agent_batch  # NOQA
