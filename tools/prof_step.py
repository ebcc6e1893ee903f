"""Where a small environment's step goes on the host (development tool): the whole
DeviceVectorDiscreteSteps.step, Context.env_step alone (fixed pool), and the bare rf_env_step call with pointers made once.
usage: python tools/prof_step.py [envs] [frame] [spp] [steps]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinfocus_amd.environments import harness  # noqa: E402

n, frame, spp, steps = (int(v) for v in (sys.argv[1:5] + ["1", "64", "1", "3000"][len(sys.argv) - 1:]))
env = harness.DeviceVectorDiscreteSteps(num_envs=n, frame_height=frame, samples_per_pixel=spp, seed=0)
env.reset(seed=0)
rng = np.random.default_rng(1)
actions = rng.integers(0, 13, size=(steps + 50, n)).astype(np.int32)


def timed(f, count):
    for i in range(50):
        f(i)
    t0 = time.perf_counter()
    for i in range(50, 50 + count):
        f(i)
    return (time.perf_counter() - t0) / count * 1e6


print("harness step        %.1f us" % timed(lambda i: env.step(actions[i]), steps))
ctx = env._ctx
pool = np.full((n, 2), 7.5, dtype=np.float32)
print("Context.env_step    %.1f us" % timed(lambda i: ctx.env_step(actions[i], pool), steps))
lib, h = ctx._lib, ctx._h
obs = np.empty((n, 4), dtype=np.float32)
rew = np.empty(n, dtype=np.float64)
tr = np.empty(n, dtype=np.uint8)
k = ctypes.c_int(0)
ptrs = [ctypes.c_void_p(a.ctypes.data) for a in (pool, obs, rew, tr)]
a0 = np.ascontiguousarray(actions[0])
pa = ctypes.c_void_p(a0.ctypes.data)
kp = ctypes.byref(k)
print("bare rf_env_step    %.1f us" % timed(lambda i: lib.rf_env_step(h, pa, ptrs[0], ptrs[1], ptrs[2], ptrs[3], kp), steps))
