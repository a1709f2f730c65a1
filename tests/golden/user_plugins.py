"""User-defined schedulers and update functions in the reference's extension idiom (ns_gym/base.py:50-203; tutorial.ipynb
cells 38-44): subclasses of `base.Scheduler` with `_check(t)` and of `base.UpdateFn` / `base.UpdateDistributionFn` with
`_update(param, t)`.

ONE definition, two worlds: `plugin_classes(base)` builds the classes on whichever `base` module it is given -
`ns_gym.base` (the reference: `make_golden.py` drives the reference wrappers with them and records the numbers) or
`ns_gym_amd.base` (this package: the tests rebuild the same objects and must reproduce those numbers through the kernels).
Test data, not product code.
"""
import numpy as np


def plugin_classes(base):
    class Every(base.Scheduler):
        """Pure function of t."""

        def __init__(self, every=5, start=0, end=np.inf):
            super().__init__(start, end)
            self.every = every

        def _check(self, t):
            return t % self.every == 0

    class EveryNthCall(base.Scheduler):
        """Stateful: fires on every n-th CALL (in range).  The wrapper restarts an episode from a deep copy of the objects it was
        built with (base.py:381-384), so the count restarts with the episode; an object shared by two update functions is called
        twice per step, in dict order."""

        def __init__(self, n, start=0, end=np.inf):
            super().__init__(start, end)
            self.n = n
            self.calls = 0

        def _check(self, t):
            self.calls += 1
            return self.calls % self.n == 0

    class SeededCoin(base.Scheduler):
        """Owns a seeded generator.  It lives inside the deep-copied objects, so every episode replays the same flips (nothing in
        the wrapper re-seeds a scheduler)."""

        def __init__(self, p, seed, start=0, end=np.inf):
            super().__init__(start, end)
            self.p = p
            self.gen = np.random.default_rng(seed)

        def _check(self, t):
            return self.gen.random() < self.p

    class GlobalCoin(base.Scheduler):
        """The tutorial's StochasticScheduler (cell 40): draws from NumPy's GLOBAL generator - not reproducible, not fusable."""

        def _check(self, t):
            return np.random.choice([True, False], p=[0.25, 0.75])

    class Sawtooth(base.UpdateFn):
        """Value- and t-dependent; walks below zero now and then, so the constraint checker rejects some proposals and the NEXT
        proposal starts from the un-updated value (classic_control.py:87-92)."""

        def __init__(self, scheduler, up=0.25, down=0.6, block=5):
            super().__init__(scheduler)
            self.up, self.down, self.block = up, down, block

        def _update(self, param, t):
            return param + self.up if (t // self.block) % 2 == 0 else param - self.down

    class Momentum(base.UpdateFn):
        """Reads the bookkeeping `UpdateFn.__call__` keeps on the object (prev_param: the value handed in at the previous call,
        fired or not; base.py:143-148)."""

        def __init__(self, scheduler, k, beta=0.5):
            super().__init__(scheduler)
            self.k, self.beta = k, beta

        def _update(self, param, t):
            last = param if self.prev_param is None else self.prev_param
            return param + self.beta * (param - last) + self.k

    class OscillatingSlip(base.UpdateDistributionFn):
        """The tutorial's oscillating slip updater (cell 43): deterministic <-> slippery, mutating the list it is handed."""

        def __init__(self, scheduler, head=0.4):
            super().__init__(scheduler)
            self.head = head

        def _update(self, param, t):
            n = len(param)
            head, rest = (self.head, (1 - self.head) / (n - 1)) if param[0] == 1 else (1, 0)
            param[0] = head
            for i in range(1, n):
                param[i] = rest
            return param

    class Sharpen(base.UpdateDistributionFn):
        """Value-dependent distribution update: squares and renormalises (returns a NEW list of NumPy floats)."""

        def __init__(self, scheduler, floor=0.05):
            super().__init__(scheduler)
            self.floor = floor

        def _update(self, param, t):
            w = np.asarray(param, dtype=float) ** 2 + self.floor * (1 + t % 3)
            return list(w / w.sum())

    return {c.__name__: c for c in (Every, EveryNthCall, SeededCoin, GlobalCoin, Sawtooth, Momentum, OscillatingSlip, Sharpen)}


def _dec(v):
    if v == "inf":
        return np.inf
    if isinstance(v, dict) and "__set__" in v:
        return set(v["__set__"])
    if isinstance(v, dict) and "__tuples__" in v:
        return [tuple(x) for x in v["__tuples__"]]
    return v


def build_params(base, S, U, params_spec):
    """tunable_params from a neutral spec.  Class names prefixed "user:" come from `plugin_classes(base)`, the others from the
    scheduler / update-function modules S / U; `{"scheduler_of": name, "update": ...}` re-uses the Scheduler OBJECT of `name`."""
    import copy

    P = plugin_classes(base)

    def cls(mod, name):
        return P[name[5:]] if name.startswith("user:") else getattr(mod, name)

    out = {}
    for name, fs in params_spec.items():
        if "scheduler_of" in fs:
            sched = out[fs["scheduler_of"]].scheduler
        else:
            sname, skw = fs["scheduler"]
            sched = cls(S, sname)(**{k: _dec(v) for k, v in skw.items()})
        uname, ukw = fs["update"]
        out[name] = cls(U, uname)(sched, **copy.deepcopy({k: _dec(v) for k, v in ukw.items()}))
    return out
