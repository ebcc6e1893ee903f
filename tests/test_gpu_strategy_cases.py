"""The reference's known answers for the strategy objects of DiscreteSteps-v0
(tests/golden/reference_strategy_cases.json) asserted against the device-resident step:
rf_env_configure / rf_env_reset / rf_env_step through the C ABI (csrc/rf_env.h kernels around
the real render and focus kernels).  tests/test_reference_strategy_cases.py asserts the same
numbers against the numpy glue on the CPU.

How a case becomes a device run.  rf_env_* is the task's fixed pipeline -- move the focus plane
by action_set[action], clip, TimeLimit | Diverging, render + focus, Normalized(Delta(..)),
Delta + Observation + OnTarget reward, same-step auto-reset -- with every constant in
rf_env_config, so a component is isolated by neutral constants for the others, and the
reference's input states are reached with an action set made of exactly the moves the case
needs.  Three things differ from the bare strategy objects and are handled explicitly:
  * only the focus plane (element 1) moves: cases that move element 0 keep their
    |element 0 - element 1| sequence (all the enders / OnTargetRewarder read) by moving element 1;
  * an environment that ends is reset in the same step, so its flags are compared up to and
    including the step it ends in (`alive`), and a reference case's partial reset is reproduced
    only where an ending can trigger it (time_limit_ender_reset); other cases run up to it;
  * all values are shifted by +5 so that the scenes are renderable (the arithmetic under test
    depends on differences only; limits / normaliser constants shift with them).
The focus value the kernels measure enters rewards and observations: its normaliser scale is
1e12 by default (a contribution below 1e-8), and where the reward's observation part is under test
it is taken from the returned observation.
"""

import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "reference_strategy_cases.json")))["cases"]}
SHIFT = 5.0
NEVER = 10 ** 9
FRAME, SPP = 8, 1


class Device:
    """One rf_ctx configured as a generic instance of the DiscreteSteps pipeline."""

    def __init__(self, n, action_set, limits=(-20.0, 40.0), max_steps=0, threshold=1e9, early_end_steps=NEVER,
                 mid=(0, 0, 0, 0), scale=(1, 1e12, 1, 1e12), reward_scale=1.0, span=-1.0):
        import math

        from reinfocus_amd import _native
        from reinfocus_amd.graphics import camera

        self.n = n
        self.action_set = [float(a) for a in action_set]
        assert 1 <= len(self.action_set) <= 32
        self.ctx = _native.Context(0)
        cams = camera.FastCameras()
        cfg = _native.EnvConfig()
        cfg.n = n
        cfg.n_actions = len(self.action_set)
        for i, a in enumerate(self.action_set):
            cfg.action_set[i] = a
        cfg.limit_lo, cfg.limit_hi = limits
        cfg.max_steps = max_steps
        cfg.diverge_threshold = threshold
        cfg.early_end_steps = early_end_steps
        for i in range(4):
            cfg.mid[i] = mid[i]
            cfg.scale[i] = scale[i]
        cfg.reward_scale = reward_scale
        cfg.on_target_span = span
        cfg.half_width = cams._half_width
        cfg.half_height = cams._half_height
        cfg.tan_half_r = math.tan(math.radians(10))
        for i in range(3):
            cfg.look_from[i] = float(cams._look_from[i])
            cfg.cam_u[i] = float(cams._u[i])
            cfg.cam_v[i] = float(cams._v[i])
            cfg.cam_w[i] = float(cams._w[i])
        cfg.lens_radius = float(cams._half_aperture)
        cfg.frame_height, cfg.spp, cfg.gray_mode = FRAME, SPP, _native.GRAY_15BIT
        self.ctx.seed(n * FRAME * FRAME, 0, 0)
        self.ctx.env_configure(cfg)

    def reset(self, states):
        return self.ctx.env_reset(np.asarray(states, dtype=np.float32))

    def step(self, moves, pool=None):
        """Moves the focus planes by `moves` (each must be in the action set)."""
        actions = [self.action_set.index(float(m)) for m in moves]
        if pool is None:
            pool = np.full((self.n, 2), SHIFT, dtype=np.float32)
        return self.ctx.env_step(np.array(actions, dtype=np.int32), np.asarray(pool, dtype=np.float32))

    def close(self):
        self.ctx.close()


def focus_moves(sequence):
    """Per-step moves of a sequence of focus-plane vectors, and the set of all of them."""
    moves = [np.asarray(b, dtype=np.float64) - np.asarray(a, dtype=np.float64) for a, b in zip(sequence, sequence[1:])]
    return moves, sorted({float(m) for step in moves for m in step} | {0.0})


def gap_positions(case):
    """Focus-plane vectors with the case's |element 0 - element 1| per op, targets fixed at
    SHIFT + 2; ops up to (not including) the first partial reset."""
    target = SHIFT + 2.0
    positions = []
    for op in case["ops"]:
        if "mask" in op:
            break
        rows = np.asarray(op["states"], dtype=np.float64)
        positions.append(target - np.abs(rows[:, 0] - rows[:, 1]))
    return target, positions


def statuses(dev, max_steps, early_end_steps):
    steps, diverging = dev.ctx.env_counters()
    out = []
    for s, d in zip(steps, diverging):
        left = f"step {s} / {max_steps}" if max_steps else ""
        right = f"diverging {d} / {early_end_steps}" if d > 0 else ""
        out.append(left + (", " if left and right else "") + right)
    return out


@pytest.mark.parametrize("name", ["diverging_ender_is_truncated_diverge", "diverging_ender_is_truncated_threshold",
                                  "diverging_ender_reset", "diverging_ender_status", "time_limit_ender_one_step",
                                  "time_limit_ender_two_steps"])
def test_enders_on_device(name):
    case = CASES[name]
    p = case["params"]
    n = p["num_envs"]
    target, positions = gap_positions(case)
    moves, action_set = focus_moves(positions)
    max_steps = p.get("max_steps", 0)
    early = p.get("early_end_steps", NEVER)
    dev = Device(n, action_set, max_steps=max_steps, threshold=p.get("threshold", 1e9), early_end_steps=early)
    try:
        dev.reset(np.column_stack([np.full(n, target), positions[0]]))
        alive = np.full(n, True)
        compared = 0
        for op, move in zip(case["ops"][1:], moves):
            _, _, truncated, _ = dev.step(np.where(alive, move, 0.0))
            if "truncated" in op:
                assert truncated[alive].tolist() == np.array(op["truncated"])[alive].tolist(), (name, op)
                compared += int(alive.sum())
            if "status" in op:
                got = statuses(dev, max_steps, early)
                keep = alive & ~truncated  # an environment that ended has been reset: counters are 0 again
                assert [g for g, k in zip(got, keep) if k] == [s for s, k in zip(op["status"], keep) if k], (name, op)
                compared += int(keep.sum())
            alive &= ~truncated
        assert compared >= n  # every environment's flags were pinned at least once
    finally:
        dev.close()


def test_time_limit_reset_on_device():
    """time_limit_ender_reset: the reference resets environments 0 and 2 after the first step; here
    they end in that step (they diverge, early_end_steps 1) and are reset by the device itself.
    From then on the expected flags and statuses are the reference's."""
    case = CASES["time_limit_ender_reset"]
    ops = case["ops"]
    mask = np.array(ops[2]["mask"])
    assert mask.tolist() == [True, False, True, False] and case["params"]["max_steps"] == 2
    dev = Device(4, [0.0, -1.0], max_steps=2, threshold=0.0, early_end_steps=1)
    try:
        target = SHIFT + 2.0
        dev.reset(np.column_stack([np.full(4, target), np.full(4, target - 1.0)]))
        pool = np.column_stack([np.full(4, target), np.full(4, target - 1.0)])
        _, _, truncated, used = dev.step(np.where(mask, -1.0, 0.0), pool)  # 0 and 2 diverge: |gap| 1 -> 2
        assert truncated.tolist() == mask.tolist() and used == 2
        assert truncated[~mask].tolist() == np.array(ops[1]["truncated"])[~mask].tolist()
        _, _, truncated, _ = dev.step(np.zeros(4), pool)
        assert truncated.tolist() == ops[3]["truncated"]                   # [F, T, F, T]
        got = statuses(dev, 2, 1)
        assert [got[0], got[2]] == [ops[3]["status"][0], ops[3]["status"][2]]
        _, _, truncated, _ = dev.step(np.zeros(4), pool)
        assert truncated[mask].tolist() == np.array(ops[4]["truncated"])[mask].tolist()  # 0 and 2: T, T
    finally:
        dev.close()


def test_ender_combination_on_device():
    """op_ender_or_truth_table: time limit [T, F, T, F] | diverging [F, F, T, T] = [T, F, T, T]."""
    table = CASES["op_ender_or_truth_table"]["ops"][0]
    left, right = np.array(table["left"]), np.array(table["right"])
    dev = Device(4, [0.0, -1.0], max_steps=2, threshold=0.0, early_end_steps=1)
    try:
        target = SHIFT + 3.0
        dev.reset(np.column_stack([np.full(4, target), np.full(4, target - 1.0)]))
        pool = np.column_stack([np.full(4, target), np.full(4, target - 1.0)])
        # step 1: the environments that must NOT reach the time limit next step end now (they diverge)
        _, _, truncated, _ = dev.step(np.where(~left, -1.0, 0.0), pool)
        assert truncated.tolist() == (~left).tolist()
        # step 2: time limit for `left`; `right` diverges in this very step
        _, _, truncated, _ = dev.step(np.where(right, -1.0, 0.0), pool)
        assert truncated.tolist() == table["or"]
    finally:
        dev.close()


@pytest.mark.parametrize("name", ["delta_rewarder_reward", "delta_rewarder_reset"])
def test_delta_rewarder_on_device(name):
    case = CASES[name]
    scale = case["params"]["scale"]
    ops = case["ops"]
    n = len(ops[0]["states"])
    focus = [np.asarray(ops[0]["states"], dtype=np.float64)[:, 1] + SHIFT]
    partial = None
    for op in ops[1:]:
        rows = np.asarray(op["states"], dtype=np.float64)
        if "mask" in op:
            partial = (np.array(op["mask"]), rows[:, 1] + SHIFT)
            after = focus[-1].copy()
            after[partial[0]] = partial[1]
            focus.append(after)  # pseudo-position: the state after the partial reset
        else:
            focus.append(rows[:, 1] + SHIFT)
    # moves between consecutive real positions (the partial reset itself is not a move)
    action_values = {0.0}
    for a, b, op in zip(focus, focus[1:], ops[1:]):
        if "mask" not in op:
            action_values |= {float(m) for m in b - a}
    # the environments of the partial reset end (diverge, early_end_steps 1) in the step before it:
    # targets are free for DeltaRewarder (it reads element 1 only), so they are placed to make
    # exactly those environments' gaps grow in that step
    targets = np.full(n, SHIFT)
    early = NEVER
    if partial is not None:
        early = 1
        before, after = focus[0], focus[1]
        for e in range(n):
            grows = partial[0][e]
            # gap after > gap before  <=>  target on the far side of the move's start
            if after[e] == before[e]:
                assert not grows
                targets[e] = before[e]
            else:
                direction = np.sign(after[e] - before[e])
                targets[e] = before[e] - direction * 0.5 if grows else after[e] + direction * 0.5
    dev = Device(n, sorted(action_values), threshold=0.0, early_end_steps=early, reward_scale=scale)
    try:
        dev.reset(np.column_stack([targets, focus[0]]))
        position = focus[0]
        for op, new in zip(ops[1:], focus[1:]):
            if "mask" in op:
                position = new
                continue
            pool = np.column_stack([targets, new])[: n]
            if partial is not None and position is focus[0]:
                rows = [[targets[e], partial[1][i]] for i, e in enumerate(np.flatnonzero(partial[0]))]
                pool = np.array(rows + [[SHIFT, SHIFT]] * (n - len(rows)))
            _, rewards, truncated, used = dev.step(new - position, pool)
            if partial is not None and position is focus[0]:
                assert truncated.tolist() == partial[0].tolist() and used == int(partial[0].sum())
            # reward = delta part (+ < 1e-8 of observation, + 0: nothing is on target with span -1)
            np.testing.assert_allclose(rewards, op["rewards"], rtol=1e-6, atol=1e-6)
            position = new
    finally:
        dev.close()


def test_on_target_and_observation_rewarders_on_device():
    case = CASES["on_target_rewarder_reward"]
    p, op = case["params"], case["ops"][0]
    rows = np.asarray(op["states"], dtype=np.float64) + SHIFT
    dev = Device(len(rows), [0.0], span=p["span"], scale=(1, 1e4, 1, 1e4))
    try:
        dev.reset(rows)
        obs, rewards, truncated, _ = dev.step(np.zeros(len(rows)))
        assert not truncated.any()
        # nothing moved: reward = observation element 1 (ObservationRewarder(1): the reward IS that
        # element, episode_rewarder_test.py:141-157) + the on-target flag
        assert np.all(obs[:, 1] > 0) and np.all(obs[:, 1] < 1)  # a real, unclipped focus value
        on_target = rewards - obs[:, 1].astype(np.float64)
        np.testing.assert_allclose(p["off"] + (p["on"] - p["off"]) * on_target, op["rewards"], atol=1e-5)
    finally:
        dev.close()


def test_discrete_move_transformer_on_device():
    case = CASES["discrete_move_transformer_right"]
    p = case["params"]
    limits = (p["limits"][0] + SHIFT, p["limits"][1] + SHIFT)
    dev = Device(3, p["action_set"], limits=limits)
    try:
        for op in case["ops"]:
            dev.reset(np.asarray(op["states"], dtype=np.float64) + SHIFT)
            dev.ctx.env_step(np.array(op["actions"], dtype=np.int32), np.full((3, 2), SHIFT, dtype=np.float32))
            np.testing.assert_allclose(dev.ctx.env_states() - SHIFT, op["new_states"], atol=1e-6)
    finally:
        dev.close()


@pytest.mark.parametrize("name", ["delta_observer_observation", "delta_observer_observation_with_original",
                                  "delta_observer_observation_with_reset", "delta_observer_multidimensional",
                                  "indexed_element_observer"])
def test_observers_on_device(name):
    """Observation elements 0 (IndexedElementObserver of the focus plane) and 2 (its DeltaObserver
    change) with mid 0 (shifted) and scale 100: nothing clips."""
    case = CASES[name]
    scale = 100.0
    if name == "indexed_element_observer":
        op = case["ops"][1]  # index 1 = the element the task observes
        rows = np.asarray(op["states"], dtype=np.float64) + SHIFT
        dev = Device(len(rows), [0.0], mid=(SHIFT, 0, 0, 0), scale=(scale, 1e12, scale, 1e12))
        try:
            obs = dev.reset(rows)
            np.testing.assert_allclose(obs[:, 0] * scale, op["values"], atol=1e-4)
        finally:
            dev.close()
        return
    ops = case["ops"]
    n = case["params"]["num_envs"]
    # Targets sit far above every focus plane, and every move the cases observe is upwards: gaps
    # only shrink, nobody diverges.  A partial reset (DeltaObserver.reset(states, mask)) is an
    # ending: the masked environments step DOWN by one (gap grows, threshold 0, early_end_steps 1),
    # end, and take the reference's new values from the pool; the others do not move.
    target = SHIFT + 30.0
    position = np.asarray(ops[0]["values"], dtype=np.float64) + SHIFT
    action_values = {0.0, -1.0}
    walk = position.copy()
    for op in ops[1:]:
        new = np.asarray(op["values"], dtype=np.float64) + SHIFT
        if "mask" in op:
            walk[np.array(op["mask"])] = new
        else:
            assert np.all(new >= walk)
            action_values |= {float(m) for m in new - walk}
            walk = new
    dev = Device(n, sorted(action_values), threshold=0.0, early_end_steps=1, mid=(SHIFT, 0, 0, 0),
                 scale=(scale, 1e12, scale, 1e12))
    try:
        obs = dev.reset(np.column_stack([np.full(n, target), position]))
        np.testing.assert_allclose(obs[:, 2] * scale, ops[0]["deltas"], atol=1e-4)
        np.testing.assert_allclose(obs[:, 0] * scale, ops[0].get("originals", ops[0]["values"]), atol=1e-4)
        for op in ops[1:]:
            new = np.asarray(op["values"], dtype=np.float64) + SHIFT
            if "mask" in op:
                mask = np.array(op["mask"])
                pool = np.array([[target, v] for v in new] + [[SHIFT, SHIFT]] * (n - len(new)))
                obs, _, truncated, used = dev.step(np.where(mask, -1.0, 0.0), pool)
                assert truncated.tolist() == mask.tolist() and used == int(mask.sum())
                np.testing.assert_allclose(obs[mask, 2] * scale, op["deltas"], atol=1e-4)  # zero right after a reset
                position = position.copy()
                position[mask] = new
            else:
                obs, _, truncated, _ = dev.step(new - position)
                assert not truncated.any()
                np.testing.assert_allclose(obs[:, 2] * scale, op["deltas"], atol=1e-4)
                np.testing.assert_allclose(obs[:, 0] * scale, op.get("originals", op["values"]), atol=1e-4)
                position = new
    finally:
        dev.close()


@pytest.mark.parametrize("which", [0, 1])
def test_normalized_observer_on_device(which):
    case = CASES["normalized_observer_observation"]
    p = case["params"]
    low, high = p["lows"][which] + SHIFT, p["highs"][which] + SHIFT
    from reinfocus_amd.environments import harness

    mid, scale = harness.normaliser_from_bounds(np.float32([low]), np.float32([high]))
    for op in (case["ops"][0], case["ops"][2]):  # observe and reset: all five environments
        values = np.asarray(op["values"], dtype=np.float64) + SHIFT
        dev = Device(len(values), [0.0, 1.0], mid=(float(mid[0]), 0, 0, 0), scale=(float(scale[0]), 1e12, 1, 1e12))
        try:
            want = np.asarray(op["normalized"])[:, which]
            if op["op"] == "reset":
                obs = dev.reset(np.column_stack([values, values]))
            else:  # reach the values by a step: start one below, move up by one
                dev.reset(np.column_stack([values, values - 1.0]))
                obs, *_ = dev.step(np.ones(len(values)))
            np.testing.assert_allclose(obs[:, 0], want, rtol=1e-6, atol=1e-7)
        finally:
            dev.close()
