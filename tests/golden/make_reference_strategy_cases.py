"""Writes tests/golden/reference_strategy_cases.json: the NUMBERS of the reference's own unit
tests for the strategy objects whose arithmetic rf_env_* (csrc/rf_env.h) and
reinfocus_amd/environments/harness.py execute for DiscreteSteps-v0.

Each case names the reference test it was transcribed from (file:lines under
/root/reference/tests/environments/) and holds only data -- constructor parameters, input
states, expected outputs.  Nothing is imported from the reference; run this file to regenerate
the JSON (python tests/golden/make_reference_strategy_cases.py).

Conventions: `states` rows are [element 0, element 1]; a `mask` selects the environments a
partial reset / observation applies to (its rows are the selected environments, in order).
"""

import json
import os

T, F = True, False

CASES = [
    # ---- DivergingEnder(num_envs, (0, 1), threshold, early_end_steps) --------------------------
    {
        "name": "diverging_ender_is_truncated_diverge",
        "source": "episode_ender_test.py:127-143",
        "component": "diverging_ender",
        "params": {"num_envs": 3, "threshold": 0, "early_end_steps": 2},
        "ops": [
            {"op": "reset", "states": [[-1, -1], [0, 0], [1, 1]]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.5], [1.5, 1]], "truncated": [F, F, F], "terminated": [F, F, F]},
            {"op": "step", "states": [[-1, -0.6], [0, -0.6], [1.5, 0.5]], "truncated": [F, T, T]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.6], [1.5, 0.5]], "truncated": [T, T, T]},
        ],
    },
    {
        "name": "diverging_ender_is_truncated_threshold",
        "source": "episode_ender_test.py:145-161",
        "component": "diverging_ender",
        "params": {"num_envs": 3, "threshold": 0.25, "early_end_steps": 2},
        "ops": [
            {"op": "reset", "states": [[-1, -1], [0, 0], [1, 1]]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.5], [1.5, 1]], "truncated": [F, F, F]},
            {"op": "step", "states": [[-1, -0.6], [0, -0.6], [1.5, 0.5]], "truncated": [F, F, T]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.9], [1.5, 0.5]], "truncated": [F, T, T]},
        ],
    },
    {
        "name": "diverging_ender_reset",
        "source": "episode_ender_test.py:163-181",
        "component": "diverging_ender",
        "params": {"num_envs": 3, "threshold": 0, "early_end_steps": 2},
        "ops": [
            {"op": "reset", "states": [[-1, -1], [0, 0], [1, 1]]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.5], [1.5, 1]], "truncated": [F, F, F]},
            {"op": "reset", "states": [[1.5, 1]], "mask": [F, F, T]},
            {"op": "step", "states": [[-1, -0.6], [0, -0.6], [1.5, 0.5]], "truncated": [F, T, F]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.6], [2, 0.5]], "truncated": [T, T, T]},
        ],
    },
    {
        "name": "diverging_ender_status",
        "source": "episode_ender_test.py:183-207",
        "component": "diverging_ender",
        "params": {"num_envs": 3, "threshold": 0, "early_end_steps": 2},
        "ops": [
            {"op": "reset", "states": [[-1, -1], [0, 0], [1, 1]]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.5], [1.5, 1]],
             "status": ["diverging 1 / 2", "diverging 1 / 2", "diverging 1 / 2"]},
            {"op": "reset", "states": [[1.5, 0.5]], "mask": [F, F, T]},
            {"op": "step", "states": [[-1, -0.6], [0, -0.6], [1.5, 0.5]],
             "status": ["diverging 1 / 2", "diverging 2 / 2", ""]},
            {"op": "step", "states": [[-1, -0.5], [0, -0.6], [1.5, 0.5]],
             "status": ["diverging 2 / 2", "diverging 2 / 2", ""]},
        ],
    },
    # ---- TimeLimitEnder(num_envs, max_steps) -----------------------------------------------------
    {
        "name": "time_limit_ender_one_step",
        "source": "episode_ender_test.py:539-550",
        "component": "time_limit_ender",
        "params": {"num_envs": 4, "max_steps": 1},
        "ops": [
            {"op": "reset", "states": [[1, 0], [2, 0], [3, 0], [4, 0]]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [T, T, T, T], "terminated": [F, F, F, F]},
        ],
    },
    {
        "name": "time_limit_ender_two_steps",
        "source": "episode_ender_test.py:552-566",
        "component": "time_limit_ender",
        "params": {"num_envs": 4, "max_steps": 2},
        "ops": [
            {"op": "reset", "states": [[1, 0], [2, 0], [3, 0], [4, 0]]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [F, F, F, F]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [T, T, T, T]},
        ],
    },
    {
        "name": "time_limit_ender_reset",
        "source": "episode_ender_test.py:568-610",
        "component": "time_limit_ender",
        "params": {"num_envs": 4, "max_steps": 2},
        "ops": [
            {"op": "reset", "states": [[1, 0], [2, 0], [3, 0], [4, 0]]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [F, F, F, F],
             "status": ["step 1 / 2", "step 1 / 2", "step 1 / 2", "step 1 / 2"]},
            {"op": "reset", "states": [[1, 0], [3, 0]], "mask": [T, F, T, F]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [F, T, F, T],
             "status": ["step 1 / 2", "step 2 / 2", "step 1 / 2", "step 2 / 2"]},
            {"op": "step", "states": [[1, 0], [2, 0], [3, 0], [4, 0]], "truncated": [T, T, T, T]},
        ],
    },
    # ---- OpEnder: l | r of the two above (the DiscreteSteps-v0 combination) -------------------------
    {
        "name": "op_ender_or_truth_table",
        "source": "episode_ender_test.py:86-108, :324-356",
        "component": "op_ender_or",
        "params": {},
        "ops": [
            {"op": "combine", "left": [T, F, T, F], "right": [F, F, T, T], "or": [T, F, T, T], "and": [F, F, T, F]},
            {"op": "status", "left": "A", "right": "B", "joined": "A, B"},
            {"op": "status", "left": "B", "right": "A", "joined": "B, A"},
            {"op": "status", "left": "A", "right": "", "joined": "A"},
            {"op": "status", "left": "", "right": "B", "joined": "B"},
        ],
    },
    # ---- DeltaRewarder(check_index, scale) ---------------------------------------------------------
    {
        "name": "delta_rewarder_reward",
        "source": "episode_rewarder_test.py:72-91",
        "component": "delta_rewarder",
        "params": {"check_index": 1, "scale": 2},
        "ops": [
            {"op": "reset", "states": [[4, 1], [3, 2], [2, 3], [1, 4]]},
            {"op": "reward", "states": [[4, 1], [3, 4], [2, 1], [1, 5]], "rewards": [0, -1, -1, -0.5]},
            {"op": "reward", "states": [[4, 0.6], [3, 1], [2, 4], [1, 3.5]], "rewards": [-0.2, -1.5, -1.5, -0.75]},
        ],
    },
    {
        "name": "delta_rewarder_reset",
        "source": "episode_rewarder_test.py:93-117",
        "component": "delta_rewarder",
        "params": {"check_index": 1, "scale": 2},
        "ops": [
            {"op": "reset", "states": [[4, 1], [3, 2], [2, 3], [1, 4]]},
            {"op": "reward", "states": [[4, 1], [3, 4], [2, 1], [1, 5]], "rewards": [0, -1, -1, -0.5]},
            {"op": "reset", "states": [[3, 2], [1, 4]], "mask": [F, T, F, T]},
            {"op": "reward", "states": [[4, 0.6], [3, 1], [2, 4], [1, 3.5]], "rewards": [-0.2, -0.5, -1.5, -0.25]},
        ],
    },
    # ---- ObservationRewarder(index) --------------------------------------------------------------
    {
        "name": "observation_rewarder_reward",
        "source": "episode_rewarder_test.py:141-157",
        "component": "observation_rewarder",
        "params": {},
        "ops": [
            {"op": "reward", "index": 0, "observations": [[4, 1], [3, 2], [2, 3], [1, 4]], "rewards": [4, 3, 2, 1]},
            {"op": "reward", "index": 1, "observations": [[4, 1], [3, 2], [2, 3], [1, 4]], "rewards": [1, 2, 3, 4]},
        ],
    },
    # ---- OnTargetRewarder((0, 1), span, off, on) -----------------------------------------------------
    {
        "name": "on_target_rewarder_reward",
        "source": "episode_rewarder_test.py:160-173",
        "component": "on_target_rewarder",
        "params": {"span": 0.1, "off": -3, "on": 7},
        "ops": [
            {"op": "reward", "states": [[0.5, 0.65], [0.5, 0.55], [0.5, 0.45], [0.5, 0.35]], "rewards": [-3, 7, 7, -3]},
        ],
    },
    # ---- OpRewarder: l + r (the combination DiscreteSteps-v0 uses) ---------------------------------
    {
        "name": "op_rewarder_plus",
        "source": "episode_rewarder_test.py:46-55",
        "component": "op_rewarder_plus",
        "params": {},
        "ops": [{"op": "combine", "left": [1, 2], "right": [3, 4], "add": [4, 6]}],
    },
    # ---- DiscreteMoveTransformer(num_envs, move_index, limits, action_set) ----------------------------
    {
        "name": "discrete_move_transformer_left",
        "source": "state_transformer_test.py:106-121",
        "component": "discrete_move_transformer",
        "params": {"move_index": 0, "limits": [0, 1], "action_set": [-0.5, 0, 0.5]},
        "ops": [
            {"op": "transform", "states": [[1, 0], [1, 0.5], [1, 1]], "actions": [0, 1, 2],
             "new_states": [[0.5, 0], [1, 0.5], [1, 1]]},
            {"op": "transform", "states": [[1, 0], [1, 0.5], [1, 1]], "actions": [2, 1, 0],
             "new_states": [[1, 0], [1, 0.5], [0.5, 1]]},
        ],
    },
    {
        "name": "discrete_move_transformer_right",
        "source": "state_transformer_test.py:123-141",
        "component": "discrete_move_transformer",
        "params": {"move_index": 1, "limits": [0, 1], "action_set": [-0.5, 0, 0.5]},
        "ops": [
            {"op": "transform", "states": [[1, 0], [1, 0.5], [1, 1]], "actions": [0, 0, 0],
             "new_states": [[1, 0], [1, 0], [1, 0.5]]},
            {"op": "transform", "states": [[1, 0], [1, 0.5], [1, 1]], "actions": [1, 1, 1],
             "new_states": [[1, 0], [1, 0.5], [1, 1]]},
            {"op": "transform", "states": [[1, 0], [1, 0.5], [1, 1]], "actions": [2, 2, 2],
             "new_states": [[1, 0.5], [1, 1], [1, 1]]},
        ],
    },
    # ---- DeltaObserver(observer(s), include_original, max_change) ----------------------------------
    # the wrapped observer of these tests returns the (one-element) state itself
    {
        "name": "delta_observer_spaces",
        "source": "state_observer_test.py:205-291",
        "component": "delta_observer_spaces",
        "params": {},
        "ops": [
            {"op": "space", "lows": [2], "highs": [5], "max_change": None, "include_original": F,
             "low": [-3], "high": [3]},
            {"op": "space", "lows": [2], "highs": [5], "max_change": [1], "include_original": F,
             "low": [-1], "high": [1]},
            {"op": "space", "lows": [2, 3, 6], "highs": [5, 7, 9], "max_change": [1, 2, None], "include_original": F,
             "low": [-1, -2, -3], "high": [1, 2, 3]},
            {"op": "space", "lows": [1, 2, 3], "highs": [2, 4, 6], "max_change": None, "include_original": T,
             "low": [1, 2, 3, -1, -2, -3], "high": [2, 4, 6, 1, 2, 3]},
        ],
    },
    {
        "name": "delta_observer_observation",
        "source": "state_observer_test.py:293-312",
        "component": "delta_observer",
        "params": {"num_envs": 3},
        "ops": [
            {"op": "reset", "values": [0, 1, 2], "deltas": [0, 0, 0]},
            {"op": "observe", "values": [0, 2, 4], "deltas": [0, 1, 2]},
        ],
    },
    {
        "name": "delta_observer_partial_observation",
        "source": "state_observer_test.py:314-340",
        "component": "delta_observer",
        "params": {"num_envs": 4},
        "ops": [
            {"op": "reset", "values": [0, 1, 2, 3], "deltas": [0, 0, 0, 0]},
            {"op": "observe", "values": [3, 6], "mask": [F, T, F, T], "deltas": [2, 3]},
            {"op": "observe", "values": [-2, 1], "mask": [T, T, F, F], "deltas": [-2, -2]},
        ],
    },
    {
        "name": "delta_observer_observation_with_original",
        "source": "state_observer_test.py:342-367",
        "component": "delta_observer",
        "params": {"num_envs": 3},
        # the wrapped observer returns [-state, state]; the second column is what is kept here
        "ops": [
            {"op": "reset", "values": [0, 1, 2], "deltas": [0, 0, 0], "originals": [0, 1, 2]},
            {"op": "observe", "values": [0, 3, 6], "deltas": [0, 2, 4], "originals": [0, 3, 6]},
        ],
    },
    {
        "name": "delta_observer_observation_with_reset",
        "source": "state_observer_test.py:369-391",
        "component": "delta_observer",
        "params": {"num_envs": 2},
        "ops": [
            {"op": "reset", "values": [0, 1], "deltas": [0, 0]},
            {"op": "reset", "values": [3], "mask": [F, T], "deltas": [0]},
            {"op": "observe", "values": [2, 4], "deltas": [2, 1]},
        ],
    },
    {
        "name": "delta_observer_multidimensional",
        "source": "state_observer_test.py:393-420",
        "component": "delta_observer",
        "params": {"num_envs": 4},
        # second wrapped observer returns -state: its deltas are the negated ones
        "ops": [
            {"op": "reset", "values": [0, 1, 2, 3], "deltas": [0, 0, 0, 0]},
            {"op": "observe", "values": [0, 2, 4, 6], "deltas": [0, 1, 2, 3], "negated_deltas": [0, -1, -2, -3]},
        ],
    },
    # ---- IndexedElementObserver(num_envs, index, min, max) ------------------------------------------
    {
        "name": "indexed_element_observer",
        "source": "state_observer_test.py:486-528",
        "component": "indexed_element_observer",
        "params": {"num_envs": 5},
        "ops": [
            {"op": "observe", "states": [[0, 1], [1, 3], [2, 5], [3, 7], [4, 9]], "index": 0, "values": [0, 1, 2, 3, 4]},
            {"op": "observe", "states": [[0, 1], [1, 3], [2, 5], [3, 7], [4, 9]], "index": 1, "values": [1, 3, 5, 7, 9]},
            {"op": "observe", "states": [[0, 1], [2, 5], [4, 9]], "mask": [T, F, T, F, T], "index": 0, "values": [0, 2, 4]},
            {"op": "observe", "states": [[1, 3], [3, 7]], "mask": [F, T, F, T, F], "index": 1, "values": [3, 7]},
        ],
    },
    # ---- NormalizedObserver([Box(0, 2), Box(1, 4)]) ------------------------------------------------
    {
        "name": "normalized_observer_observation",
        "source": "state_observer_test.py:578-634",
        "component": "normalized_observer",
        "params": {"lows": [0, 1], "highs": [2, 4]},
        "ops": [
            {"op": "observe", "values": [0, 1, 2, 3, 4],
             "normalized": [[-1.0, -1.0], [0.0, -1.0], [1.0, -0.3333333333333333], [1.0, 0.3333333333333333], [1.0, 1.0]]},
            {"op": "observe", "values": [0, 2, 4], "mask": [T, F, T, F, T],
             "normalized": [[-1.0, -1.0], [1.0, -0.3333333333333333], [1.0, 1.0]]},
            {"op": "reset", "values": [0, 1, 2, 3, 4],
             "normalized": [[-1.0, -1.0], [0.0, -1.0], [1.0, -0.3333333333333333], [1.0, 0.3333333333333333], [1.0, 1.0]]},
        ],
    },
]


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_strategy_cases.json")
    with open(path, "w") as out:
        json.dump({"note": "numbers of the reference's strategy-object tests; see make_reference_strategy_cases.py",
                   "cases": CASES}, out, indent=1)
        out.write("\n")
    print(f"{len(CASES)} cases -> {path}")


if __name__ == "__main__":
    main()
