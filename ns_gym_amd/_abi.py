"""ctypes mirror of include/nsgym_hip.h (the C-ABI data format).  Host-side only."""
from __future__ import annotations

import ctypes as C

NSG_ABI_VERSION = 4
NSG_MAX_PARAMS = 8
NSG_MAX_THETA = 8
NSG_MAX_SEGMENTS = 8
MAX_TABLE_BYTES = 16384   # kMaxTableBytes (csrc/nsg_kernels.hip.h): the constant-table blob a batch may stage in LDS

# env types
(ENV_CARTPOLE, ENV_PENDULUM, ENV_ACROBOT, ENV_MOUNTAINCAR, ENV_MOUNTAINCAR_CONT, ENV_FROZENLAKE,
 ENV_CLIFFWALKING, ENV_BRIDGE) = range(8)
GRID_ENVS = (ENV_FROZENLAKE, ENV_CLIFFWALKING, ENV_BRIDGE)
N_DIST = {ENV_FROZENLAKE: 3, ENV_CLIFFWALKING: 4, ENV_BRIDGE: 3}

# schedulers
(SCHED_CONTINUOUS, SCHED_PERIODIC, SCHED_BURST, SCHED_TABLE, SCHED_RANDOM, SCHED_DECAYING,
 SCHED_MEMORYLESS) = range(7)

# scalar update fns
(UPD_INCREMENT, UPD_DECREMENT, UPD_TREND, UPD_POLY, UPD_GEOMETRIC, UPD_EXPDECAY, UPD_OSCILLATING,
 UPD_SIGMOID, UPD_LERP, UPD_STEPWISE, UPD_CYCLIC, UPD_NOUPDATE, UPD_RANDOMWALK, UPD_RW_DRIFT,
 UPD_RW_DRIFT_TREND, UPD_OU, UPD_BOUNDED_RW) = range(17)
# distribution update fns
(UPD_D_INCREMENT, UPD_D_DECREMENT, UPD_D_STEPWISE, UPD_D_CYCLIC, UPD_D_NOUPDATE, UPD_D_UNIFORMDRIFT,
 UPD_D_TARGETREV, UPD_D_LERP, UPD_D_RANDOMCAT, UPD_D_LCBOUNDED) = range(32, 42)

F_CHANGE_NOTIFICATION = 0x1
F_DELTA_NOTIFICATION = 0x2
F_PERSISTENT_PARAMS = 0x4
F_TRACK_RETURNS = 0x8
F_MODIFIED_REWARDS = 0x10
F_VIOLATION_MASK = 0x200
F_TERMINAL_CLIFF = 0x100
F_SIM_ENV = 0x40
F_IN_SIM_CHANGE = 0x80
F_NO_AUTORESET = 0x400
F_LIBM_EXACT = 0x800   # sin / cos / scalar ** 2 / exp evaluated as glibc's libm does, bit for bit (specialised units only)

ST_NEEDS_RESET = 0x1
EP_COUNT_SHIFT = 1          # episode word of the classic-control envs: resets drawn so far << 1 | needs-reset
STREAM_AFFINE = 1 << 63     # descriptor word 0 of buffers.rng_env (classic-control envs)

CNT_DONE, CNT_FIRED, CNT_VIOLATION, CNT_STEPS, CNT_LC_EXHAUSTED, CNT_SCHED_OVERRUN = 0, 1, 2, 3, 4, 5
CNT_COUNT = 6
CNT_SHARDS = 16384


class ParamCfg(C.Structure):
    _fields_ = [
        ("theta_slot", C.c_int32),
        ("sched_kind", C.c_int32),
        ("upd_kind", C.c_int32),
        ("rng_child", C.c_int32),
        ("sched_start", C.c_double),
        ("sched_end", C.c_double),
        ("sched_i0", C.c_int64),
        ("sched_i1", C.c_int64),
        ("sched_p0", C.c_double),
        ("sched_p1", C.c_double),
        ("sched_tab_off", C.c_int32),
        ("sched_tab_len", C.c_int32),
        ("val_tab_off", C.c_int32),
        ("val_tab_len", C.c_int32),
        ("u", C.c_double * 10),
        ("fn_seed", C.c_uint64),
        ("has_fn_seed", C.c_int32),
        ("uses_rng", C.c_int32),
        ("sched_seed", C.c_uint64),
        ("has_sched_seed", C.c_int32),
        ("fn_slot", C.c_int32),
        ("sched_slot", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("env_type", C.c_int32),
        ("n_params", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("flags", C.c_uint32),
        ("nrow", C.c_int32),
        ("ncol", C.c_int32),
        ("desc_tab_off", C.c_int32),
        ("base_theta", C.c_double * NSG_MAX_THETA),
        ("initial_prob", (C.c_double * 4) * 2),
        ("letter_reward", C.c_double * 4),
        ("params", ParamCfg * NSG_MAX_PARAMS),
    ]


BUFFER_FIELDS = [
    ("phys", C.c_double), ("cell", C.c_int32), ("theta", C.c_double), ("table_prob", C.c_double), ("derived", C.c_double), ("t", C.c_int32), ("t_fork", C.c_int32),
    ("status", C.c_uint8), ("episode", C.c_int32), ("rng_env", C.c_uint64), ("rng_upd", C.c_uint64), ("rng_sched", C.c_uint64), ("sched_next", C.c_int32), ("cursor", C.c_int32),
    ("obs", C.c_float), ("reward", C.c_float), ("terminated", C.c_uint8), ("truncated", C.c_uint8),
    ("env_change", C.c_uint8), ("delta_change", C.c_float), ("violation", C.c_uint8), ("prob", C.c_float),
    ("ep_return", C.c_float), ("ep_length", C.c_int32), ("last_return", C.c_float),
    ("last_length", C.c_int32), ("counters", C.c_uint64), ("done_bits", C.c_uint64),
]


class Buffers(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in BUFFER_FIELDS]


class Layout(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("phys_dim", C.c_int32), ("obs_dim", C.c_int32), ("n_params", C.c_int32),
        ("n_theta_rows", C.c_int32), ("action_is_float", C.c_int32), ("n_actions", C.c_int32),
    ] + [(name, C.c_int64) for name, _ in BUFFER_FIELDS]


class TraceState(C.Structure):
    """nsg_trace_state: the per-object state nsg_theta_trace_stateful carries across calls."""
    _fields_ = [("rng", C.c_void_p), ("cursor", C.c_void_p), ("sched_rng", C.c_void_p), ("sched_next", C.c_void_p),
                ("resume", C.c_int32), ("reserved0", C.c_int32)]


class RolloutOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")]


# nsg_policy.kind: where a fused rollout takes its actions from (nsg_rollout_policy)
NSG_POL_TABLE, NSG_POL_UNIFORM, NSG_POL_BY_STATE, NSG_POL_LINEAR = 0, 1, 2, 3


class Policy(C.Structure):
    """nsg_policy"""
    _fields_ = [("kind", C.c_int32), ("step0", C.c_int32), ("seed", C.c_uint64), ("index0", C.c_int64), ("data", C.c_void_p),
                ("n_data", C.c_int32), ("reserved0", C.c_int32), ("actions_out", C.c_void_p)]


class EpisodeAcc(C.Structure):
    """nsg_episode_acc"""
    _fields_ = [("ret", C.c_void_p), ("length", C.c_void_p), ("alive", C.c_void_p), ("discount", C.c_void_p),
                ("n_discount", C.c_int32), ("reserved0", C.c_int32)]
