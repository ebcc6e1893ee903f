"""Development tool: randomised soak of the device-resident environment (rf_env_*, incl. the
hipGraph replay) against the numpy-glue harness.  usage: python tools/soak_env.py [cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from reinfocus_amd.environments import harness  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time()
    for case in range(cases):
        n = int(rng.integers(1, 400))
        height = int(rng.choice([8, 16, 30, 32, 50, 64, 100]))
        spp = int(rng.integers(1, 8))
        steps = int(rng.integers(3, 30))
        max_steps = int(rng.integers(1, 25))
        kw = dict(num_envs=n, frame_height=height, samples_per_pixel=spp, seed=int(rng.integers(0, 1 << 30)),
                  device=0, max_episode_steps=max_steps)
        host = harness.VectorDiscreteSteps(**kw)
        dev = harness.DeviceVectorDiscreteSteps(**kw)
        assert np.array_equal(host.reset()[0], dev.reset()[0])
        for step in range(steps):
            actions = rng.integers(0, 13, n)
            want = host.step(actions)
            got = dev.step(actions)
            for a, b in zip(want[:4], got[:4]):
                assert np.array_equal(a, b), (case, step, n, height, spp)
            if step % 5 == 4:
                assert np.array_equal(host._state, dev._state), (case, step, "state")
        host.close()
        dev.close()
        print(f"case {case}: n={n} h={height} spp={spp} steps={steps} limit={max_steps} ok ({time.time() - t0:.0f} s)",
              flush=True)
    print("env soak ok")


if __name__ == "__main__":
    main()
