from .sampler import DeviceRandomState, Sampler, global_random_state  # noqa: F401
