from pfrl_amd.utils.batch_states import batch_states  # NOQA
from pfrl_amd.utils.contexts import evaluating  # NOQA
from pfrl_amd.utils.copy_param import copy_grad, copy_param, soft_copy_param, synchronize_parameters  # NOQA
from pfrl_amd.utils.random_seed import set_random_seed  # NOQA
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_  # NOQA
