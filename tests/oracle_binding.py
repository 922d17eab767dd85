"""ctypes binding of the CPU oracle (oracle/_build/libmirt_oracle*.so).  TEST INFRASTRUCTURE:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

import weekend_raytracer_wgpu_amd as mirt
from weekend_raytracer_wgpu_amd import _abi

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "oracle" / "_build"
CLEAN, FAITHFUL = 0, 1


def _cpu_has_fma() -> bool:
    try:
        return " fma " in (" " + Path("/proc/cpuinfo").read_text().replace("\n", " ") + " ")
    except OSError:
        return False


def _load() -> C.CDLL:
    name = "libmirt_oracle.so" if _cpu_has_fma() else "libmirt_oracle_nofma.so"
    target = []
    if os.environ.get("MIRT_ORACLE_VARIANT") == "asan":      # tests/test_oracle_asan.py: ASan + UBSan build of the oracle
        name, target = "libmirt_oracle_asan.so", ["asan"]
    path = BUILD / name
    if not path.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")] + target, check=True, capture_output=True)
    lib = C.CDLL(str(path))
    P = C.POINTER
    fp = P(C.c_float)
    sigs = {
        "mirt_oracle_render": (C.c_int, [P(_abi.MirtScene), P(_abi.MirtParams), C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
        "mirt_oracle_render_pt_sums": (C.c_int, [P(_abi.MirtScene), P(_abi.MirtParams), C.c_void_p, C.c_size_t, C.c_int]),
        "mirt_oracle_get_stats": (C.c_int, [P(_abi.MirtStats)]),
        "mirt_oracle_camera_new": (C.c_int, [P(_abi.MirtCamera), C.c_uint32, C.c_uint32, P(_abi.MirtGpuCamera)]),
        "mirt_oracle_camera_from_fly_pose": (C.c_int, [fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, P(_abi.MirtCamera)]),
        "mirt_oracle_degrees_to_radians": (C.c_float, [C.c_float]),
        "mirt_oracle_radians_to_degrees": (C.c_float, [C.c_float]),
        "mirt_oracle_validate_render_params": (C.c_int, [P(_abi.MirtCamera), P(_abi.MirtSamplingParams), C.c_uint32, C.c_uint32]),
        "mirt_oracle_math_sincos": (None, [fp, fp, fp, C.c_size_t]),
        "mirt_oracle_math_acos": (None, [fp, fp, C.c_size_t]),
        "mirt_oracle_math_atan2": (None, [fp, fp, fp, C.c_size_t]),
        "mirt_oracle_math_log2": (None, [fp, fp, C.c_size_t]),
        "mirt_oracle_math_exp2": (None, [fp, fp, C.c_size_t]),
        "mirt_oracle_math_pow": (None, [fp, fp, fp, C.c_size_t]),
        "mirt_oracle_rng_stream": (None, [C.c_uint32, C.c_uint32, C.c_uint64, fp, C.c_size_t]),
        "mirt_params_out_rows_impl": (C.c_uint32, [P(_abi.MirtParams)]),
        "mirt_params_out_row_index_impl": (C.c_uint32, [P(_abi.MirtParams), C.c_uint32]),
    }
    for n, (r, a) in sigs.items():
        f = getattr(lib, n)
        f.restype, f.argtypes = r, a
    return lib


LIB = _load()


class OracleError(RuntimeError):
    def __init__(self, status: int):
        self.status = status
        self.status_name = _abi.STATUS.get(status, str(status))
        super().__init__(self.status_name)


def host_cores() -> int:
    """CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota (a GPU box gives a one-GPU job 16 of its 256
    logical CPUs: 256 OpenMP threads on that quota run the oracle at half the rate of 16)."""
    import os
    from pathlib import Path
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                    # cgroup v2
        q, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                                # cgroup v1
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


DEFAULT_THREADS = host_cores()       # n_threads = 0 in the calls below: one thread per usable CPU, not OpenMP's one per logical CPU


def out_rows(params) -> int:
    return int(LIB.mirt_params_out_rows_impl(C.byref(params)))


def out_row_index(params, i: int) -> int:
    return int(LIB.mirt_params_out_row_index_impl(C.byref(params), i))


def render(scene: "mirt.SceneData", params, n_threads: int = 0, variant: int = CLEAN) -> np.ndarray:
    """Oracle render -> uint8 [rows, width, 4]."""
    rows = out_rows(params)
    out = np.zeros((rows, params.width, 4), dtype=np.uint8)
    c = scene.as_c()
    rc = LIB.mirt_oracle_render(C.byref(c), C.byref(params), out.ctypes.data_as(C.c_void_p), out.nbytes, n_threads or DEFAULT_THREADS, variant)
    if rc != 0:
        raise OracleError(rc)
    return out


def render_status(scene_c, params, out_nbytes: int = 1 << 20) -> int:
    """Raw status of an oracle render with a scratch output buffer (error-behaviour tests)."""
    buf = (C.c_uint8 * out_nbytes)()
    return int(LIB.mirt_oracle_render(C.byref(scene_c) if scene_c is not None else None,
                                      C.byref(params) if params is not None else None, buf, out_nbytes, 1, CLEAN))


def render_pt_sums(scene: "mirt.SceneData", params, n_threads: int = 0) -> np.ndarray:
    rows = out_rows(params)
    out = np.zeros((rows, params.width, 3), dtype=np.uint64)
    c = scene.as_c()
    rc = LIB.mirt_oracle_render_pt_sums(C.byref(c), C.byref(params), out.ctypes.data_as(C.c_void_p), out.size, n_threads or DEFAULT_THREADS)
    if rc != 0:
        raise OracleError(rc)
    return out


def stats() -> dict:
    st = _abi.MirtStats()
    LIB.mirt_oracle_get_stats(C.byref(st))
    return st.as_dict()


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    LIB.mirt_oracle_math_sincos(_fp(x), _fp(s), _fp(c), x.size)
    return s, c


def _unary(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fn(_fp(x), _fp(y), x.size)
    return y


def acos(x): return _unary(LIB.mirt_oracle_math_acos, x)
def log2(x): return _unary(LIB.mirt_oracle_math_log2, x)
def exp2(x): return _unary(LIB.mirt_oracle_math_exp2, x)


def atan2(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty_like(x)
    LIB.mirt_oracle_math_atan2(_fp(y), _fp(x), _fp(r), x.size)
    return r


def pow_pos(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    r = np.empty_like(x)
    LIB.mirt_oracle_math_pow(_fp(x), _fp(y), _fp(r), x.size)
    return r


def rng_stream(pixel_index: int, sample: int, seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    LIB.mirt_oracle_rng_stream(pixel_index, sample, seed, _fp(out), n)
    return out
