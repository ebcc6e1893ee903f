"""Development tool: cProfile of the numpy-glue environment (harness.VectorDiscreteSteps) at a small configuration."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, ".")
from reinfocus_amd.environments import harness  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = harness.VectorDiscreteSteps(num_envs=n, frame_height=128, samples_per_pixel=4, seed=0, device=0)
env.reset()
rng = np.random.default_rng(0)
acts = [rng.integers(0, 13, n) for _ in range(300)]
for a in acts[:20]:
    env.step(a)
pr = cProfile.Profile()
pr.enable()
for a in acts[20:]:
    env.step(a)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
