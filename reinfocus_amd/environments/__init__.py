"""Caller side of the hot path: FocusObserver (the reference's only GPU-touching
strategy) and a thin vector-environment harness that reproduces the call order of
reinfocus.environments.VectorEnvironment for the DiscreteSteps-v0 task."""
