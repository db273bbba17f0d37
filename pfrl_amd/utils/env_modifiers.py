"""In-place env patches (reference pfrl/utils/env_modifiers.py): each replaces ``env.step`` /
``reset`` / ``close`` of ONE env object with a closure around the original bound method."""


def make_rendered(env, *render_args, **render_kwargs):
    """Render after every step; closing renders once more with ``close=True``."""
    step, close = env.step, env.close

    def rendered_step(action):
        result = step(action)
        env.render(*render_args, **render_kwargs)
        return result

    def rendered_close():
        env.render(*render_args, close=True, **render_kwargs)
        close()

    env.step, env.close = rendered_step, rendered_close


def make_timestep_limited(env, timestep_limit):
    """Report ``done`` from the ``timestep_limit``-th step of an episode on."""
    step, reset = env.step, env.reset
    elapsed = [0]

    def limited_step(action):
        observation, reward, done, info = step(action)
        elapsed[0] += 1
        return observation, reward, done or elapsed[0] >= timestep_limit, info

    def limited_reset():
        elapsed[0] = 0
        return reset()

    env.step, env.reset = limited_step, limited_reset


def make_action_filtered(env, action_filter):
    step = env.step
    env.step = lambda action: step(action_filter(action))


def make_reward_filtered(env, reward_filter):
    step = env.step

    def filtered_step(action):
        observation, reward, done, info = step(action)
        return observation, reward_filter(reward), done, info

    env.step = filtered_step
