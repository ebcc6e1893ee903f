"""Environment ids of the reference's examples (examples/__init__.py:6-18).

`DiscreteSteps-v0` is registered with gymnasium when gymnasium is importable (same id,
entry_point + vector_entry_point, max_episode_steps=20), and is always available through
the local make()/make_vec() below, which accept the same arguments as
gymnasium.make / gymnasium.make_vec(id, num_envs, vectorization_mode="custom", vector_kwargs=...).
`ContinuousJumps-v0` (examples/__init__.py:14-18) is the single-environment continuous task.

The vector entry point builds the device-resident environment (harness.DeviceVectorDiscreteSteps,
the one bench.py measures); `glue="host"` selects the numpy-glue VectorDiscreteSteps (identical
results) and `devices=[...]` the environment sharded over several GPUs.
"""

from reinfocus_amd.environments import harness

ENTRY_POINTS = {
    "DiscreteSteps-v0": {
        "entry_point": "reinfocus_amd.environments.harness:DiscreteSteps",
        "vector_entry_point": "reinfocus_amd.registration:vector_discrete_steps",
        "max_episode_steps": 20,
    },
    "ContinuousJumps-v0": {
        "entry_point": "reinfocus_amd.environments.harness:ContinuousJumps",
        "max_episode_steps": 20,
    },
}


def vector_discrete_steps(max_episode_steps=20, num_envs=1, render_mode=None, *, glue="device", devices=None,
                          **kwargs):
    """vector_entry_point of DiscreteSteps-v0 (examples/__init__.py:9,
    custom_environments.py:148-153: max_episode_steps, num_envs, render_mode).  Extensions:
    glue = "device" (default: whole step on the GPU) or "host" (numpy glue around render + focus);
    devices = list of GPU indices -> one environment sharded over them; frame_height,
    samples_per_pixel, seed, device, first_state_index as in harness."""
    if devices is not None:
        assert glue == "device", "the sharded environment is device-resident"
        return harness.ShardedVectorDiscreteSteps(max_episode_steps, num_envs, render_mode, devices=devices, **kwargs)
    if glue == "host":
        return harness.VectorDiscreteSteps(max_episode_steps, num_envs, render_mode, **kwargs)
    assert glue == "device", f"glue must be 'device' or 'host', not {glue!r}"
    return harness.DeviceVectorDiscreteSteps(max_episode_steps, num_envs, render_mode, **kwargs)


def register_with_gymnasium():
    """Registers the ids with gymnasium; returns False when gymnasium is not installed."""
    try:
        from gymnasium.envs import registration
    except ImportError:
        return False
    for env_id, spec in ENTRY_POINTS.items():
        if env_id not in registration.registry:
            registration.register(id=env_id, **spec)
    return True


def make(env_id, **kwargs):
    """Local stand-in for gymnasium.make (without the TimeLimit wrapper gymnasium adds
    from max_episode_steps)."""
    if env_id == "DiscreteSteps-v0":
        return harness.DiscreteSteps(**kwargs)
    if env_id == "ContinuousJumps-v0":
        return harness.ContinuousJumps(**kwargs)
    raise KeyError(f"{env_id} is not provided by reinfocus_amd ({sorted(ENTRY_POINTS)})")


def make_vec(env_id, num_envs=1, vectorization_mode="custom", vector_kwargs=None, **kwargs):
    if env_id != "DiscreteSteps-v0":
        raise KeyError(f"{env_id} has no vector entry point (only DiscreteSteps-v0, examples/__init__.py:6-11)")
    assert vectorization_mode == "custom", "DiscreteSteps-v0 brings its own vector environment"
    args = {"max_episode_steps": ENTRY_POINTS[env_id]["max_episode_steps"]}
    args.update(vector_kwargs or {})
    args.update(kwargs)
    return vector_discrete_steps(num_envs=num_envs, **args)


register_with_gymnasium()
