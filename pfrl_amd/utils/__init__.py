from pfrl_amd.utils.batch_states import batch_states  # NOQA
from pfrl_amd.utils.contexts import evaluating  # NOQA
from pfrl_amd.utils import copy_param  # NOQA  (the MODULE, as in the reference: utils.copy_param.soft_copy_param)
from pfrl_amd.utils.random_seed import set_random_seed  # NOQA
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_  # NOQA
from pfrl_amd.utils import ask_yes_no  # NOQA  (the MODULE, as in the reference)
from pfrl_amd.utils import env_modifiers  # NOQA
from pfrl_amd.utils.conjugate_gradient import conjugate_gradient  # NOQA
from pfrl_amd.utils.is_return_code_zero import is_return_code_zero  # NOQA
from pfrl_amd.utils.stoppable_thread import StoppableThread  # NOQA
