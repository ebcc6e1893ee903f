"""The reference's own known answers for the strategy objects of DiscreteSteps-v0
(tests/golden/reference_strategy_cases.json, transcribed from the reference's
tests/environments/*_test.py by tests/golden/make_reference_strategy_cases.py) asserted
against the host glue in reinfocus_amd/environments/harness.py.  tests/test_gpu_strategy_cases.py
asserts the same numbers against rf_env_* through the C ABI.

harness.py holds the task's fixed combination (TimeLimit | Diverging ender, Delta + Observation +
OnTarget reward, Normalized(Delta([IndexedElement, Focus]))), so each component is isolated by
neutral settings of the others (an unreachable time limit, a zero observation, a span nothing is
inside of).  float32 states: tolerances 1e-6 where the reference computes in float64."""

import json
import os
import types

import numpy as np
import pytest

from reinfocus_amd.environments import harness

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_strategy_cases.json")))["cases"]
NEVER = 10 ** 9


def cases_of(*components):
    return [pytest.param(c, id=c["name"]) for c in CASES if c["component"] in components]


def f32(rows):
    return np.array(rows, dtype=np.float32)


def mask_of(op):
    return None if "mask" not in op else np.array(op["mask"], dtype=bool)


def test_fixture_is_what_the_script_writes(tmp_path):
    """The committed JSON is exactly the committed script's output."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_reference_strategy_cases.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert json.loads(json.dumps(mk.CASES)) == CASES
    assert len(CASES) >= 20 and all(c["source"] and c["ops"] for c in CASES)


@pytest.mark.parametrize("case", cases_of("diverging_ender", "time_limit_ender"))
def test_enders(case):
    p = case["params"]
    if case["component"] == "diverging_ender":
        ender = harness._Ender(p["num_envs"], None, p["threshold"], p["early_end_steps"])
    else:
        ender = harness._Ender(p["num_envs"], p["max_steps"], NEVER, NEVER)
    for op in case["ops"]:
        states = f32(op["states"])
        if op["op"] == "reset":
            ender.reset(states, mask_of(op))
            continue
        ender.step(states)
        if "truncated" in op:
            assert ender.is_truncated().tolist() == op["truncated"], op
        if "terminated" in op:
            assert ender.is_terminated().tolist() == op["terminated"], op
        if "status" in op:
            assert [ender.status(i) for i in range(p["num_envs"])] == op["status"], op


@pytest.mark.parametrize("case", cases_of("op_ender_or"))
def test_ender_combination(case):
    """TimeLimitEnder | DivergingEnder (custom_environments.py:186-190 on episode_ender.py:79-104):
    the truth table and the status join of the reference's OpEnder tests."""
    for op in case["ops"]:
        if op["op"] == "combine":
            n = len(op["left"])
            ender = harness._Ender(n, 2, 0.0, 2)
            ender._steps[:] = np.where(op["left"], 2, 1)             # left: the time limit is reached
            ender._diverging_steps[:] = np.where(op["right"], 2, 0)  # right: diverged long enough
            assert ender.is_truncated().tolist() == op["or"]
            assert not ender.is_terminated().any()
        else:
            # the left ender is the time limit, the right one the diverging rule; an ender that
            # reports "" in the reference's case is one that has nothing to report here
            ender = harness._Ender(1, 2 if op["left"] else None, 0.0, 2)
            ender._steps[:] = 1
            ender._diverging_steps[:] = 1 if op["right"] else 0
            want = op["joined"]
            if op["left"]:
                want = want.replace(op["left"], "step 1 / 2")
            if op["right"]:
                want = want.replace(op["right"], "diverging 1 / 2")
            assert ender.status(0) == want


@pytest.mark.parametrize("case", cases_of("delta_rewarder"))
def test_delta_rewarder(case):
    p = case["params"]
    assert p["check_index"] == harness.FOCUS
    rewarder = harness._Rewarder(p["scale"], -1.0)  # span -1: nothing is ever on target
    n = len(case["ops"][0]["states"])
    zeros = np.zeros((n, 4), dtype=np.float32)
    for op in case["ops"]:
        states = f32(op["states"])
        if op["op"] == "reset":
            rewarder.reset(states, zeros[: len(states)], mask_of(op))
        else:
            np.testing.assert_allclose(rewarder.reward(states, zeros), op["rewards"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", cases_of("observation_rewarder"))
def test_observation_rewarder(case):
    for op in case["ops"]:
        observations = f32(op["observations"])
        states = np.zeros((len(observations), 2), dtype=np.float32)
        rewarder = harness._Rewarder(1.0, -1.0, focus_value_o_index=op["index"])
        rewarder.reset(states, observations)
        np.testing.assert_allclose(rewarder.reward(states, observations), op["rewards"])


@pytest.mark.parametrize("case", cases_of("on_target_rewarder"))
def test_on_target_rewarder(case):
    p = case["params"]
    for op in case["ops"]:
        states = f32(op["states"])
        zeros = np.zeros((len(states), 4), dtype=np.float32)
        rewarder = harness._Rewarder(1.0, p["span"])
        rewarder.reset(states, zeros)
        on_target = rewarder.reward(states, zeros)  # 1.0 inside the span, 0.0 outside (the task's values)
        np.testing.assert_allclose(p["off"] + (p["on"] - p["off"]) * on_target, op["rewards"])


def test_reward_is_the_sum_of_its_parts():
    """OpRewarder '+' (episode_rewarder_test.py:46-55: [1, 2] + [3, 4] = [4, 6]) on the task's three
    rewarders: the delta case's numbers + the observation case's numbers + the on-target flags."""
    by_name = {c["name"]: c for c in CASES}
    plus = by_name["op_rewarder_plus"]["ops"][0]
    assert (np.array(plus["left"]) + np.array(plus["right"])).tolist() == plus["add"]
    delta = by_name["delta_rewarder_reward"]
    observed = f32(by_name["observation_rewarder_reward"]["ops"][1]["observations"])
    rewarder = harness._Rewarder(delta["params"]["scale"], 1.5)  # |target - focus| < 1.5
    rewarder.reset(f32(delta["ops"][0]["states"]), observed)
    states = f32(delta["ops"][1]["states"])
    on_target = (np.abs(states[:, 0] - states[:, 1]) < 1.5) * 1.0
    assert on_target.tolist() == [0.0, 1.0, 1.0, 0.0]
    np.testing.assert_allclose(rewarder.reward(states, observed),
                               np.array(delta["ops"][1]["rewards"]) + observed[:, 1] + on_target, rtol=1e-6)


@pytest.mark.parametrize("case", cases_of("discrete_move_transformer"))
def test_discrete_move_transformer(case):
    p = case["params"]
    glue = types.SimpleNamespace(_action_set=np.asarray(p["action_set"]), _limits=tuple(p["limits"]))
    # harness moves the focus-plane element (index 1); the reference's move_index 0 cases are the
    # same arithmetic with the columns swapped
    swap = p["move_index"] != harness.FOCUS
    for op in case["ops"]:
        states, want = f32(op["states"]), f32(op["new_states"])
        if swap:
            states, want = states[:, ::-1].copy(), want[:, ::-1]
        got = harness._HostGlue._transform(glue, states, np.array(op["actions"]).reshape(-1, 1))
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, want)


@pytest.mark.parametrize("case", cases_of("delta_observer_spaces"))
def test_delta_observer_spaces(case):
    for op in case["ops"]:
        max_change = None if op["max_change"] is None else [np.nan if m is None else m for m in op["max_change"]]
        low, high = harness.delta_bounds(op["lows"], op["highs"], max_change, op["include_original"])
        np.testing.assert_allclose(low, op["low"])
        np.testing.assert_allclose(high, op["high"])


class _NegatedPosition:
    """Stands in for FocusObserver: observes -position, the second wrapped observer of the
    reference's multidimensional DeltaObserver test."""

    def observe(self, states, indices):
        return -states[:, harness.FOCUS].reshape((indices.sum(), 1))


def _observer(num_envs, mid, scale):
    observer = harness._Observer.__new__(harness._Observer)
    observer._focus = _NegatedPosition()
    observer._mid = np.asarray(mid, dtype=np.float32)
    observer._scale = np.asarray(scale, dtype=np.float32)
    observer._old = np.full((num_envs, 2), np.nan, dtype=np.float32)
    observer._num_envs = num_envs
    return observer


@pytest.mark.parametrize("case", cases_of("delta_observer"))
def test_delta_observer(case):
    n = case["params"]["num_envs"]
    scale = 100.0  # nothing clips: observation * scale = the unnormalised [original, -original, delta, -delta]
    observer = _observer(n, np.zeros(4), np.full(4, scale))
    for op in case["ops"]:
        mask = mask_of(op)
        states = np.zeros((len(op["values"]), 2), dtype=np.float32)
        states[:, harness.FOCUS] = op["values"]
        got = (observer.reset if op["op"] == "reset" else observer.observe)(states, mask) * scale
        np.testing.assert_allclose(got[:, 2], op["deltas"], atol=1e-5)
        np.testing.assert_allclose(got[:, 3], op.get("negated_deltas", [-d for d in op["deltas"]]), atol=1e-5)
        np.testing.assert_allclose(got[:, 0], op.get("originals", op["values"]), atol=1e-5)


@pytest.mark.parametrize("case", cases_of("indexed_element_observer"))
def test_indexed_element_observer(case):
    n = case["params"]["num_envs"]
    for op in case["ops"]:
        observer = _observer(n, np.zeros(4), np.full(4, 100.0))
        mask = mask_of(op)
        given = f32(op["states"])
        states = np.zeros_like(given)
        states[:, harness.FOCUS] = given[:, op["index"]]  # harness observes the focus-plane element
        got = observer.reset(states, mask) * 100.0
        np.testing.assert_allclose(got[:, 0], op["values"], atol=1e-5)


@pytest.mark.parametrize("case", cases_of("normalized_observer"))
def test_normalized_observer(case):
    p = case["params"]
    mid, scale = harness.normaliser_from_bounds(f32(p["lows"]), f32(p["highs"]))
    observer = _observer(5, mid, scale)
    for op in case["ops"]:
        values = f32(op["values"]).reshape(-1, 1)
        got = observer._normalize(np.hstack([values, values]))
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, op["normalized"], rtol=1e-6)


def test_task_normaliser_uses_these_rules():
    """custom_environments.py:196-218: Normalized(Delta([Indexed(5, 10), Focus(min, max)], True,
    [max_move, nan])) -- mid / scale of the task are delta_bounds + normaliser_from_bounds."""
    mid, scale = harness.normaliser_constants((5.0, 10.0), 5.0, 30.0, 350.0)
    np.testing.assert_allclose(mid, [7.5, 190.0, 0.0, 0.0])
    np.testing.assert_allclose(scale, [2.5, 160.0, 5.0, 320.0])
