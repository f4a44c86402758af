from .sampler import DeviceRandomState, Sampler, global_random_state, host_numpy_stream  # noqa: F401
