#!/usr/bin/env python3
"""Is the 2^23-2^24 fall-off (0.78 of peak at 2^22 envs, 0.62-0.70 at 2^24) a matter of how far apart a batch's rows lie?
2^24 C1 envs as ONE handle (rows 64-128 MB apart) against the same envs as 4 handles of 2^22 / 16 of 2^20 (each handle's rows
16-32 MB / 4-8 MB apart), stepped back to back on one stream and - where it fits the 8-member limit - as one nsg_step_group launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ns_gym_amd import workloads as W  # noqa: E402
from ns_gym_amd.vec_env import step_group  # noqa: E402


def timed(fn, iters=60, reps=3):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(best)[1]


def main():
    total = 1 << 24
    out = {}
    for parts in (1, 4, 8, 16):
        n = total // parts
        envs = [W.build("c1", n, specialize=True, seed=7 + k) for k in range(parts)]
        acts = [W.random_actions(e) for e in envs]

        def seq():
            for e, a in zip(envs, acts):
                e.step(a)
        out[f"{parts} x 2^{n.bit_length() - 1} envs, back-to-back launches"] = timed(seq)
        if 1 < parts <= 8:
            out[f"{parts} x 2^{n.bit_length() - 1} envs, one nsg_step_group launch"] = timed(lambda: step_group(envs, acts))
        for e in envs:
            e.close()
        del envs, acts
        torch.cuda.empty_cache()
    for k, v in out.items():
        print(f"{k}: {v:.1f} us per 2^24 env-steps = {120 * total / (v * 1e-6) / 8e12:.3f} of 8 TB/s", flush=True)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
