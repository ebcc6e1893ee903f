"""ctypes binding of libreinfocus_hip.so (include/reinfocus_hip.h).

The HIP library is the only compute path of this package: if it is missing or no GPU
is usable the calls below raise, loudly.  There is no CPU fallback (the CPU oracle
under oracle/ is test infrastructure and is never imported from here).
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# REINFOCUS_HIP_LIB: an alternative build of the same library (tests use it for a build with a
# tiny cooperative list); never a different implementation -- there is no CPU fallback
LIB_PATH = os.environ.get("REINFOCUS_HIP_LIB") or os.path.join(_HERE, "libreinfocus_hip.so")

RF_OK = 0
RF_ERR_INVALID = -1
RF_ERR_HIP = -2
RF_ERR_NO_DEVICE = -3
RF_ERR_OOM = -4

GRAY_15BIT = 15
GRAY_14BIT = 14

# every symbol include/reinfocus_hip.h declares
SYMBOLS = (
    "rf_last_error",
    "rf_abi_version",
    "rf_device_count",
    "rf_device_info",
    "rf_create",
    "rf_destroy",
    "rf_seed",
    "rf_num_states",
    "rf_get_states",
    "rf_set_states",
    "rf_set_scene",
    "rf_render",
    "rf_get_frames",
    "rf_upload_frames",
    "rf_focus",
    "rf_step",
    "rf_synchronize",
    "rf_timing",
    "rf_timing_read",
    "rf_render_general",
    "rf_env_configure",
    "rf_env_reset",
    "rf_env_step",
    "rf_env_step_begin",
    "rf_env_step_end",
    "rf_env_step_abort",
    "rf_env_step_plan",
    "rf_env_step_run",
    "rf_env_render_states",
    "rf_env_step_end_given",
    "rf_env_get_states",
    "rf_env_scene_len",
    "rf_env_render",
    "rf_env_get_counters",
    "rf_env_last_step_branch",
    "rf_render_kernel_name",
    "rf_pixels_rendered",
    "rf_allocations_poisoned",
    "rf_general_redo_pixels",
)


class EnvConfig(ctypes.Structure):
    """rf_env_config (include/reinfocus_hip.h)."""

    _fields_ = [
        ("n", ctypes.c_int),
        ("n_actions", ctypes.c_int),
        ("action_set", ctypes.c_double * 32),
        ("limit_lo", ctypes.c_float),
        ("limit_hi", ctypes.c_float),
        ("max_steps", ctypes.c_int),
        ("diverge_threshold", ctypes.c_float),
        ("early_end_steps", ctypes.c_int),
        ("mid", ctypes.c_float * 4),
        ("scale", ctypes.c_float * 4),
        ("reward_scale", ctypes.c_float),
        ("on_target_span", ctypes.c_float),
        ("half_width", ctypes.c_double),
        ("half_height", ctypes.c_double),
        ("tan_half_r", ctypes.c_double),
        ("look_from", ctypes.c_float * 3),
        ("cam_u", ctypes.c_float * 3),
        ("cam_v", ctypes.c_float * 3),
        ("cam_w", ctypes.c_float * 3),
        ("lens_radius", ctypes.c_double),
        ("frame_height", ctypes.c_int),
        ("spp", ctypes.c_int),
        ("gray_mode", ctypes.c_int),
    ]


class NativeLibraryMissing(ImportError):
    """libreinfocus_hip.so has not been built (python -c 'import __graft_entry__ as g; g.build()')."""


_lib = None


def load():
    """Loads the shared library (once) and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C reinfocus_amd/csrc` "
            "(or __graft_entry__.build()); reinfocus_amd has no CPU fallback"
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, u64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_double
    lib.rf_last_error.restype = ctypes.c_char_p
    lib.rf_last_error.argtypes = []
    lib.rf_abi_version.argtypes = []
    lib.rf_device_count.argtypes = [ctypes.POINTER(i32)]
    lib.rf_device_info.argtypes = [i32, ctypes.c_char_p, i32, ctypes.POINTER(i32)]
    lib.rf_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.rf_destroy.argtypes = [vp]
    lib.rf_seed.argtypes = [vp, u64, u64, u64]
    lib.rf_num_states.argtypes = [vp, ctypes.POINTER(u64)]
    lib.rf_get_states.argtypes = [vp, u64, u64, vp]
    lib.rf_set_states.argtypes = [vp, u64, u64, vp]
    lib.rf_set_scene.argtypes = [vp, i32, vp, vp, vp, vp, vp, dbl]
    lib.rf_render.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.rf_get_frames.argtypes = [vp, i32, i32, vp]
    lib.rf_upload_frames.argtypes = [vp, i32, i32, i32, vp]
    lib.rf_focus.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.rf_step.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.rf_synchronize.argtypes = [vp]
    lib.rf_timing.argtypes = [vp, i32]
    lib.rf_timing_read.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(u64), ctypes.POINTER(dbl),
                                   ctypes.POINTER(u64)]
    lib.rf_render_general.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp]
    lib.rf_env_configure.argtypes = [vp, ctypes.POINTER(EnvConfig)]
    lib.rf_env_reset.argtypes = [vp, vp, vp]
    lib.rf_env_step.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.POINTER(i32)]
    lib.rf_env_step_begin.argtypes = [vp, vp, vp, vp, ctypes.POINTER(i32)]
    lib.rf_env_step_end.argtypes = [vp, vp, vp]
    lib.rf_env_step_abort.argtypes = [vp]
    lib.rf_env_step_plan.argtypes = [vp, vp, ctypes.POINTER(i32)]
    lib.rf_env_step_run.argtypes = [vp, vp, vp, vp, vp]
    lib.rf_env_render_states.argtypes = [vp, i32, vp, vp]
    lib.rf_env_step_end_given.argtypes = [vp, vp, vp, vp]
    lib.rf_env_get_states.argtypes = [vp, vp]
    lib.rf_env_scene_len.argtypes = [vp, ctypes.POINTER(i32)]
    lib.rf_env_render.argtypes = [vp, i32, i32, vp]
    lib.rf_env_get_counters.argtypes = [vp, vp, vp]
    lib.rf_env_last_step_branch.argtypes = [vp, ctypes.POINTER(i32)]
    lib.rf_render_kernel_name.restype = ctypes.c_char_p
    lib.rf_render_kernel_name.argtypes = [vp]
    lib.rf_pixels_rendered.restype = ctypes.c_ulonglong
    lib.rf_pixels_rendered.argtypes = []
    lib.rf_allocations_poisoned.restype = ctypes.c_int
    lib.rf_allocations_poisoned.argtypes = []
    lib.rf_general_redo_pixels.restype = ctypes.c_uint
    lib.rf_general_redo_pixels.argtypes = [vp]
    _lib = lib
    return lib


def _check(rc):
    if rc == RF_OK:
        return
    msg = load().rf_last_error().decode("utf-8", "replace")
    if rc == RF_ERR_INVALID:
        raise AssertionError(msg)
    if rc == RF_ERR_OOM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def device_count():
    n = ctypes.c_int(0)
    _check(load().rf_device_count(ctypes.byref(n)))
    return n.value


def device_info(device):
    """{"device", "pci_bus_id", "numa_node"} of a HIP device index (rf_device_info); numa_node is -1 when the
    host does not say."""
    bus = ctypes.create_string_buffer(32)
    node = ctypes.c_int(-1)
    _check(load().rf_device_info(int(device), bus, 32, ctypes.byref(node)))
    return {"device": int(device), "pci_bus_id": bus.value.decode(), "numa_node": node.value}


def numa_cpus(node, sysfs="/sys/devices/system/node"):
    """CPUs of a NUMA node ("0-15,128-143" in <sysfs>/node<N>/cpulist) as a set; empty if unknown."""
    cpus = set()
    try:
        text = open(os.path.join(sysfs, f"node{int(node)}", "cpulist")).read().strip()
    except (OSError, ValueError):
        return cpus
    for part in filter(None, text.split(",")):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_numa_node(node, whole_process=False, sysfs="/sys/devices/system/node"):
    """Restricts the calling thread (or every thread of the process: the HIP runtime's helper threads exist by the
    time a device can be asked where it sits) to the CPUs of `node` that the process may use at all.  Returns the
    sorted CPU list now in force, or None when nothing was changed (unknown node, no such CPUs, no permission)."""
    if node is None or int(node) < 0 or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = numa_cpus(node, sysfs) & set(os.sched_getaffinity(0))
    if not allowed:
        return None
    try:
        tasks = [int(t) for t in os.listdir("/proc/self/task")] if whole_process else [0]
    except OSError:
        tasks = [0]
    changed = False
    for task in tasks:
        try:
            os.sched_setaffinity(task, allowed)
            changed = True
        except OSError:  # a thread that has ended meanwhile, or no permission
            pass
    return sorted(allowed) if changed else None


def pixels_rendered():
    """Pixels all render launches of this process were made for (rf_pixels_rendered)."""
    return int(load().rf_pixels_rendered())


def allocations_poisoned():
    """True when the library fills its allocations with 0xA5 bytes first (REINFOCUS_POISON_ALLOC; rf_allocations_poisoned)."""
    return bool(load().rf_allocations_poisoned())


def default_device():
    """One process per GPU: torchrun's LOCAL_RANK selects the device."""
    return int(os.environ.get("REINFOCUS_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """Owner of one rf_ctx (one renderer's device state on one GPU)."""

    def __init__(self, device=None):
        self._lib = load()
        self._h = ctypes.c_void_p()
        dev = default_device() if device is None else int(device)
        _check(self._lib.rf_create(dev, ctypes.byref(self._h)))
        self.device = dev

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rf_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # --- RNG ---------------------------------------------------------------------------
    def seed(self, n_states, seed=0, first_state_index=0):
        _check(self._lib.rf_seed(self._h, int(n_states), int(seed), int(first_state_index)))

    def num_states(self):
        n = ctypes.c_uint64(0)
        _check(self._lib.rf_num_states(self._h, ctypes.byref(n)))
        return n.value

    def get_states(self, first=0, count=None):
        count = self.num_states() - first if count is None else count
        out = np.zeros((int(count), 2), dtype=np.uint64)
        _check(self._lib.rf_get_states(self._h, int(first), int(count), _ptr(out)))
        return out

    def set_states(self, states, first=0):
        states = np.ascontiguousarray(states, dtype=np.uint64)
        assert states.ndim == 2 and states.shape[1] == 2
        _check(self._lib.rf_set_states(self._h, int(first), states.shape[0], _ptr(states)))

    # --- scene / render / focus ----------------------------------------------------------
    def set_scene(self, cam_dyn, rect, origin, u, v, lens_radius):
        cam_dyn = np.ascontiguousarray(cam_dyn, dtype=np.float32)
        rect = np.ascontiguousarray(rect, dtype=np.float32)
        n = rect.shape[0]
        assert cam_dyn.shape == (n, 3, 3), "cam_dyn must be float32[n,3,3]"
        assert rect.shape == (n, 2), "rect must be float32[n,2]"
        origin = np.ascontiguousarray(origin, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        _check(self._lib.rf_set_scene(self._h, n, _ptr(cam_dyn), _ptr(rect), _ptr(origin), _ptr(u), _ptr(v),
                                      float(lens_radius)))

    def render(self, n, h, w, spp, to_host=False):
        out = np.empty((n, h, w, 3), dtype=np.uint8) if to_host else None
        _check(self._lib.rf_render(self._h, n, h, w, spp, _ptr(out) if to_host else None))
        return out

    def get_frames(self, shape, first_env=0, n_envs=None):
        n, h, w = shape[:3]
        n_envs = n - first_env if n_envs is None else n_envs
        out = np.empty((n_envs, h, w, 3), dtype=np.uint8)
        _check(self._lib.rf_get_frames(self._h, first_env, n_envs, _ptr(out)))
        return out

    def upload_frames(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        assert frames.ndim == 4 and frames.shape[3] == 3, "frames must be uint8[n,h,w,3]"
        n, h, w = frames.shape[:3]
        _check(self._lib.rf_upload_frames(self._h, n, h, w, _ptr(frames)))

    def focus(self, n, h, w, gray_mode=GRAY_15BIT):
        out = np.empty(n, dtype=np.float64)
        _check(self._lib.rf_focus(self._h, n, h, w, gray_mode, _ptr(out)))
        return out

    def step(self, n, h, w, spp, gray_mode=GRAY_15BIT):
        out = np.empty(n, dtype=np.float64)
        _check(self._lib.rf_step(self._h, n, h, w, spp, gray_mode, _ptr(out)))
        return out

    # --- general renderer ------------------------------------------------------------------
    def render_general(self, cameras, params, types, sizes, h, w, spp, to_host=True):
        """rf_render_general; to_host=False leaves the frames on the device (rf_get_frames fetches them)."""
        cameras = np.ascontiguousarray(cameras, dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float32)
        types = np.ascontiguousarray(types, dtype=np.int32)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        n, most, width = params.shape
        assert cameras.shape == (n, 19) and types.shape == (n, most) and sizes.shape == (n,)
        if width < 7:  # the kernel reads up to 7 parameters per shape
            params = np.ascontiguousarray(np.pad(params, ((0, 0), (0, 0), (0, 7 - width))))
            width = 7
        out = np.empty((n, h, w, 3), dtype=np.uint8) if to_host else None
        _check(self._lib.rf_render_general(self._h, n, h, w, spp, _ptr(cameras), _ptr(params), _ptr(types),
                                           _ptr(sizes), most, width, _ptr(out) if to_host else None))
        return out

    # --- device-resident env step ----------------------------------------------------------
    def env_configure(self, cfg):
        self._env_n = cfg.n
        self._env_k = ctypes.c_int(0)  # (env_step's count of ended environments, and its reference, made once)
        self._env_k_ref = ctypes.byref(self._env_k)
        _check(self._lib.rf_env_configure(self._h, ctypes.byref(cfg)))

    def env_reset(self, states):
        states = np.ascontiguousarray(states, dtype=np.float32).reshape(self._env_n, 2)
        obs = np.empty((self._env_n, 4), dtype=np.float32)
        _check(self._lib.rf_env_reset(self._h, _ptr(states), _ptr(obs)))
        return obs

    def env_step(self, actions, pool):
        # (addresses as plain integers: the ctypes casts of _ptr cost 2 us each, a sixth of a small environment's step)
        n = self._env_n
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(n)
        pool = np.ascontiguousarray(pool, dtype=np.float32).reshape(n, 2)
        obs = np.empty((n, 4), dtype=np.float32)
        rewards = np.empty(n, dtype=np.float64)
        truncated = np.empty(n, dtype=np.bool_)  # (the library writes 0 / 1 bytes)
        k = self._env_k
        rc = self._lib.rf_env_step(self._h, actions.ctypes.data, pool.ctypes.data, obs.ctypes.data, rewards.ctypes.data,
                                   truncated.ctypes.data, self._env_k_ref)
        if rc != 0:
            _check(rc)
        return obs, rewards, truncated, k.value

    def env_step_begin(self, actions):
        """First half of a two-phase step: (rewards, truncated, number of environments that ended)."""
        n = self._env_n
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(n)
        rewards = np.empty(n, dtype=np.float64)
        truncated = np.empty(n, dtype=np.uint8)
        k = ctypes.c_int(0)
        _check(self._lib.rf_env_step_begin(self._h, _ptr(actions), _ptr(rewards), _ptr(truncated), ctypes.byref(k)))
        return rewards, truncated.astype(bool), k.value

    def env_step_end(self, pool_rows):
        """Second half: the environments that ended take pool_rows float32[k, 2]; observations."""
        n = self._env_n
        pool_rows = np.ascontiguousarray(pool_rows, dtype=np.float32).reshape(-1, 2)
        obs = np.empty((n, 4), dtype=np.float32)
        _check(self._lib.rf_env_step_end(self._h, _ptr(pool_rows) if len(pool_rows) else None, _ptr(obs)))
        return obs

    def env_step_plan(self, actions):
        """First half of a step cut BEFORE its render (transform, enders, ranking): how many environments end."""
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self._env_n)
        k = ctypes.c_int(0)
        _check(self._lib.rf_env_step_plan(self._h, _ptr(actions), ctypes.byref(k)))
        return k.value

    def env_step_run(self, pool_rows):
        """The rest of a planned step with pool_rows float32[k, 2] for the environments that end:
        (observations, rewards, truncated)."""
        n = self._env_n
        pool_rows = np.ascontiguousarray(pool_rows, dtype=np.float32).reshape(-1, 2)
        obs = np.empty((n, 4), dtype=np.float32)
        rewards = np.empty(n, dtype=np.float64)
        truncated = np.empty(n, dtype=np.uint8)
        _check(self._lib.rf_env_step_run(self._h, _ptr(pool_rows) if len(pool_rows) else None, _ptr(obs), _ptr(rewards),
                                         _ptr(truncated)))
        return obs, rewards, truncated.astype(bool)

    def env_render_states(self, states):
        """Exact mode of a sharded environment: float32[k, 2] states rendered and scored as compacted rows
        0..k-1 from this context's RNG state 0; float64[k] focus values."""
        states = np.ascontiguousarray(states, dtype=np.float32).reshape(-1, 2)
        focus = np.empty(len(states), dtype=np.float64)
        _check(self._lib.rf_env_render_states(self._h, len(states), _ptr(states), _ptr(focus)))
        return focus

    def env_step_end_given(self, pool_rows, focus):
        """Second half of a two-phase step whose reset renders happened elsewhere: observations."""
        n = self._env_n
        pool_rows = np.ascontiguousarray(pool_rows, dtype=np.float32).reshape(-1, 2)
        focus = np.ascontiguousarray(focus, dtype=np.float64).reshape(-1)
        assert len(focus) == len(pool_rows)
        obs = np.empty((n, 4), dtype=np.float32)
        some = len(pool_rows) > 0
        _check(self._lib.rf_env_step_end_given(self._h, _ptr(pool_rows) if some else None,
                                               _ptr(focus) if some else None, _ptr(obs)))
        return obs

    def env_step_abort(self):
        """Drops an open two-phase step (another shard failed): env_reset must come next."""
        _check(self._lib.rf_env_step_abort(self._h))

    def env_scene_len(self):
        n = ctypes.c_int(0)
        _check(self._lib.rf_env_scene_len(self._h, ctypes.byref(n)))
        return n.value

    def env_render(self, frame_height, spp):
        """uint8[len, h, h, 3] of the scene set the environment uploaded last."""
        out = np.empty((self.env_scene_len(), frame_height, frame_height, 3), dtype=np.uint8)
        _check(self._lib.rf_env_render(self._h, int(frame_height), int(spp), _ptr(out)))
        return out

    def env_counters(self):
        steps = np.empty(self._env_n, dtype=np.int32)
        diverging = np.empty(self._env_n, dtype=np.int32)
        _check(self._lib.rf_env_get_counters(self._h, _ptr(steps), _ptr(diverging)))
        return steps, diverging

    def env_last_step_branch(self):
        """'none' | 'one-sync' | 'graph' | 'count-sized': how the last rf_env_step was scheduled."""
        branch = ctypes.c_int(0)
        _check(self._lib.rf_env_last_step_branch(self._h, ctypes.byref(branch)))
        return ("none", "one-sync", "graph", "count-sized", "fused", "fused-graph")[branch.value]

    def render_kernel_name(self):
        return self._lib.rf_render_kernel_name(self._h).decode()

    def general_redo_pixels(self):
        """Pixels the last render_general call left to its fix-up kernel (rf_general_redo_pixels)."""
        return int(self._lib.rf_general_redo_pixels(self._h))

    def env_states(self):
        out = np.empty((self._env_n, 2), dtype=np.float32)
        _check(self._lib.rf_env_get_states(self._h, _ptr(out)))
        return out

    def synchronize(self):
        _check(self._lib.rf_synchronize(self._h))

    def timing(self, enable=True):
        _check(self._lib.rf_timing(self._h, 1 if enable else 0))

    def timing_read(self):
        rm, fm = ctypes.c_double(0), ctypes.c_double(0)
        rn, fn = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self._lib.rf_timing_read(self._h, ctypes.byref(rm), ctypes.byref(rn), ctypes.byref(fm),
                                        ctypes.byref(fn)))
        return {"render_ms": rm.value, "render_launches": rn.value, "focus_ms": fm.value,
                "focus_launches": fn.value}
