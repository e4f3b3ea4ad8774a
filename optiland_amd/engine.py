"""Device-side engine: owns an `ol_system` handle and launches traces.

PyTorch is used only as the device-memory container and stream provider; all
arithmetic happens in the hand-written HIP kernels behind the C ABI.
"""

from __future__ import annotations

import collections
import ctypes as C
import logging
import threading
import os
import time
import weakref

import numpy as np
import torch

from . import _capi
from . import system as S
from .system import SystemTable

PLANES = ("x", "y", "z", "L", "M", "N", "i", "opd")
_DT = {torch.float32: _capi.F32, torch.float64: _capi.F64}
_VEC_PAD = 64  # record row stride padded to 64 elements (256 B fp32 / 512 B fp64)


def _require_gpu(device) -> torch.device:
    if not torch.cuda.is_available():
        raise _capi.HipExtensionError(
            "no HIP device visible (torch.cuda.is_available() is False); the fused "
            "trace has no CPU fallback"
        )
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise ValueError(f"HipSystem needs a cuda (HIP) device, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class TraceResult:
    """Outputs of one launch.  `record` is (rows, 8, stride) with [:, :, :n] valid."""

    __slots__ = ("n", "_rays", "record", "prt", "status", "first", "last", "updated_intensity")

    def __init__(self, n, rays, record, prt, status, first, last):
        self.n, self._rays, self.record, self.prt = n, rays, record, prt
        self.status, self.first, self.last = status, first, last
        self.updated_intensity = None  # ol_trace_generate's update_intensity epilogue (ABI 7)

    @property
    def rays(self):
        """The eight ray planes of the launch; for a generating launch that recorded row 0,
        the planes of that row (made when asked for)."""
        if self._rays is None and self.record is not None and self.first == 0:
            self._rays = list(self.record[0, :, : self.n].unbind(0))
        return self._rays

    @rays.setter
    def rays(self, value):
        self._rays = value

    def row(self, s: int, plane: str | int) -> torch.Tensor:
        k = PLANES.index(plane) if isinstance(plane, str) else plane
        return self.record[s - self.first, k, : self.n]

    def rows(self, s: int):
        """All 8 plane views of recorded surface `s` (two tensor ops instead of 8)."""
        return self.record[s - self.first, :, : self.n].unbind(0)

    def stack(self, plane: str | int) -> torch.Tensor:
        """(rows, n) view of one plane for all recorded surfaces (no copy)."""
        k = PLANES.index(plane) if isinstance(plane, str) else plane
        return self.record[:, k, : self.n]


# Record blocks that are handed to a user (`alloc_record`).  "auto" (the default since round 5):
# a shape of `min_bytes` or more that is asked for a SECOND time -- a loop, not a one-off trace --
# gets a `RecordPool` of two placed windows, provided the device has memory to spare (see
# `_auto_arena_bytes`); an int = that many windows per shape from the first request on; 0 = off.
# OPTILAND_HIP_PLACED_RECORDS (auto | 0 | n) seeds it.
def _default_slots():
    env = os.environ.get("OPTILAND_HIP_PLACED_RECORDS", "auto").strip().lower()
    if env in ("", "auto"):
        return "auto"
    try:
        return max(int(env), 0)
    except ValueError:
        return "auto"


def _default_idle_s() -> float:
    try:
        return max(float(os.environ.get("OPTILAND_HIP_POOL_IDLE_S", "30")), 0.0)
    except ValueError:
        return 30.0


# idle_s: a pool none of whose blocks is in a user's hands and that has not been asked for one
# for this many seconds gives its arenas back to the device (0 = never).
_POOL_CONFIG = {"slots": _default_slots(), "min_bytes": 256 << 20, "max_pools": 2,
                "cooldown": 64, "idle_s": _default_idle_s()}
_LOG = logging.getLogger("optiland_amd")


class _Arena:
    """Device memory of the library's OWN (`ol_arena_alloc`: hipMalloc), in which record windows
    are looked for.  Not a block of torch's caching allocator: handing an arena back is one
    hipFree and never touches the user's cache (until round 6 the arenas were torch tensors and
    `torch.cuda.empty_cache()` was how they went back to the driver)."""

    __slots__ = ("lib", "device", "ptr", "nbytes", "__weakref__")
    live_bytes = 0          # all arenas of the process (`record_pool_stats`)
    _lock = threading.RLock()   # (re-entrant: __del__ may run inside __init__'s critical section)

    def __init__(self, lib, device, nbytes: int):
        self.lib, self.device, self.nbytes, self.ptr = lib, device, int(nbytes), 0
        out = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.ol_arena_alloc(int(nbytes), C.byref(out))
        if rc != 0 or not out.value:
            raise MemoryError(f"no arena of {nbytes >> 20} MiB on {device}")
        self.ptr = int(out.value)
        with _Arena._lock:
            _Arena.live_bytes += self.nbytes

    def data_ptr(self) -> int:
        return self.ptr

    def numel(self) -> int:
        return self.nbytes

    def view(self, offset: int, nbytes: int) -> torch.Tensor:
        """uint8 tensor over [offset, offset + nbytes) whose storage keeps this arena alive."""
        return torch.as_tensor(_ArenaView(self, self.ptr + int(offset), int(nbytes)),
                               device=self.device)

    def __del__(self):
        ptr, self.ptr = self.ptr, 0
        if not ptr:
            return
        try:
            self.lib.ol_arena_free(C.c_void_p(ptr))
            with _Arena._lock:
                _Arena.live_bytes -= self.nbytes
            # the windows that lay in it are no windows any more: the driver hands these
            # addresses to the next allocation (torch's, typically), which is an ORDINARY block
            _forget_placed(self.device, ptr, ptr + self.nbytes)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class _ArenaView:
    """CUDA-array-interface handle on a piece of an `_Arena`; torch keeps it (and through it
    the arena) for as long as the storage made from it -- any view of it -- lives."""

    __slots__ = ("arena", "ptr", "nbytes")

    def __init__(self, arena, ptr, nbytes):
        self.arena, self.ptr, self.nbytes = arena, ptr, nbytes

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                "strides": None, "version": 2}


def _new_arena(hip, nbytes: int):
    """An `_Arena` of `nbytes` on the system's device, or None when the device cannot spare it
    (host-math engines of the tests, which have no hipMalloc behind them, use a tensor)."""
    if not hasattr(hip.lib, "ol_arena_alloc") or hip.device.type != "cuda":
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=hip.device)
        except RuntimeError:
            return None
    try:
        return _Arena(hip.lib, hip.device, nbytes)
    except MemoryError:
        return None


def _arena_view(arena, offset: int, nbytes: int) -> torch.Tensor:
    if isinstance(arena, _Arena):
        return arena.view(offset, nbytes)
    return arena[offset: offset + nbytes]
_RECORD_POOLS: "collections.OrderedDict" = collections.OrderedDict()  # key -> RecordPool (LRU)
_POOL_LOCK = threading.RLock()  # pools are process-wide; traces may come from several threads
_SHAPE_SEEN: dict = {}     # key -> requests so far ("auto": the second one builds the pool)
_POOL_COOLDOWN: dict = {}  # device index -> big allocations left before another pool is built
# Placed windows handed out so far: (device index, first byte) -> bytes.  A record launch into
# a big block that lies in NONE of them says so (`TRACE_FEW_WAVES`: the fp32 conic-only kernels
# write ~3 % faster into an ordinary allocation with fewer workgroups resident).  An entry whose
# arena has since been freed only costs that hint; a few dozen entries at most.
_PLACED_WINDOWS: "collections.OrderedDict" = collections.OrderedDict()
_PLACED_MAX = 64
_PLACED_LOCK = threading.RLock()  # (re-entrant: an arena finalised by the garbage collector takes it too)
_FEW_WAVES_MIN_BYTES = 256 << 20


def _note_placed(device, ptr: int, nbytes: int) -> None:
    key = (device.index or 0, int(ptr))
    with _PLACED_LOCK:
        _PLACED_WINDOWS[key] = int(nbytes)
        _PLACED_WINDOWS.move_to_end(key)
        while len(_PLACED_WINDOWS) > _PLACED_MAX:
            _PLACED_WINDOWS.popitem(last=False)


def _forget_placed(device, lo: int, hi: int) -> None:
    """Drop the placed windows that start in [lo, hi) on `device` (their arena has been freed)."""
    dev = getattr(device, "index", None) or 0
    with _PLACED_LOCK:
        for key in [k for k in _PLACED_WINDOWS if k[0] == dev and lo <= k[1] < hi]:
            del _PLACED_WINDOWS[key]


def _few_waves_flag(rec) -> int:
    """`TRACE_FEW_WAVES` for a record block of >= 256 MB outside every placed window."""
    if rec is None or rec.numel() * rec.element_size() < _FEW_WAVES_MIN_BYTES:
        return 0
    dev, ptr = rec.device.index or 0, rec.data_ptr()
    with _PLACED_LOCK:
        for (d, lo), nb in _PLACED_WINDOWS.items():
            if d == dev and lo <= ptr < lo + nb:
                return 0
    return S.TRACE_FEW_WAVES


# Hot loops (round 6).  The fp32 conic-only record-all kernels write a PLACED block 3.8 % faster
# with at most two workgroups resident per CU -- once the part's clocks have settled; in the
# first ~25 launches after an idle gap the same cap costs 11-13 % (profiles/r05_ab_wgcap.txt).
# Round 5 therefore left it to the caller (`ol_set_tuning`).  The engine can see a loop itself:
# launch after launch into the SAME block, enqueued without a pause.  From the `after`-th such
# launch on the block is traced with `TRACE_FEW_WAVES` (the kernels' word for "two workgroups");
# a gap of `gap_s` (5 ms) between two enqueues -- the device may have gone idle -- starts the count
# again.  OPTILAND_HIP_HOT_LOOP=0 turns it off; =N sets `after`.
def _hot_loop_after() -> int:
    try:
        return max(int(os.environ.get("OPTILAND_HIP_HOT_LOOP", "32")), 0)
    except ValueError:
        return 32


_HOT_LOOP = {"after": _hot_loop_after(), "gap_s": 0.005}
_HOT_BLOCKS: "collections.OrderedDict" = collections.OrderedDict()  # (dev, ptr) -> [count, last]


def _hot_loop_flag(rec) -> int:
    """`TRACE_FEW_WAVES` for a big record block that has been the target of `after` launches in
    a row without an idle gap, else 0."""
    after = _HOT_LOOP["after"]
    if not after or rec is None or rec.numel() * rec.element_size() < _FEW_WAVES_MIN_BYTES:
        return 0
    key = (rec.device.index or 0, rec.data_ptr())
    now = time.perf_counter()
    ent = _HOT_BLOCKS.get(key)
    if ent is None:
        ent = _HOT_BLOCKS[key] = [0, now]
        while len(_HOT_BLOCKS) > 8:
            _HOT_BLOCKS.popitem(last=False)
    elif now - ent[1] > _HOT_LOOP["gap_s"]:
        ent[0] = 0
    else:
        ent[0] += 1
    ent[1] = now
    return S.TRACE_FEW_WAVES if ent[0] >= after else 0


class _Lease:
    """One window of a placement arena while it is the storage of a record block handed to a
    user.  `torch.as_tensor(lease)` (CUDA array interface) makes a tensor on the window and
    keeps a reference to this object for as long as that storage -- any view of it -- lives;
    when the last one dies the window goes back to its pool.  The lease holds the arena, so
    the memory outlives the pool if it must."""

    __slots__ = ("pool", "index", "arena", "ptr", "nbytes")

    def __init__(self, pool, index, arena, ptr, nbytes):
        self.pool, self.index, self.arena, self.ptr, self.nbytes = pool, index, arena, ptr, nbytes

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                "strides": None, "version": 2}

    def __del__(self):
        try:
            self.pool._give_back(self.index)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class RecordPool:
    """A few PLACED record blocks of one shape for traces whose record is handed to a user
    (the drop-in's `Optic.trace`): windows of device memory in which the record-all store
    pattern writes fastest (`HipSystem.alloc_record_placed`), lent out one trace at a time and
    returned when the user's last view of the block dies.  A loop that keeps one result alive
    while it makes the next needs two.  The arenas behind the windows stay allocated for the
    life of the pool -- ~40 GiB for two 4 GiB windows on the boxes measured
    (`HipSystem.enable_record_pool(0)` / `integration.enable(placed_records=0)` gives them
    back).  WHY a window is fast is still not known (five experiments of round 5
    say what it is NOT); it is therefore found, not built."""

    def __init__(self, hip, n: int, dtype, rows: int, slots: int = 2, arena_bytes=None,
                 min_gain: float = 0.04, max_arenas: int = 3):
        self.device, self.dtype, self.rows = hip.device, dtype, rows  # (`hip`: for the probe only)
        b = torch.empty((), dtype=dtype).element_size()
        self.stride = hip.record_stride(n, b)
        self.need = need = rows * 8 * self.stride * b
        self.windows = []          # (arena, address)
        self.free = []
        self.lock = threading.RLock()  # (a lease's finaliser may run while this thread holds it)
        self.last_used = time.monotonic()
        self.info = {"block_bytes": need, "slots_wanted": slots, "probes": 0, "arenas": 0}
        if arena_bytes is None:
            arena_bytes = max(3 * need, 40 << 30)
        coarse_ms, candidates, held = [], [], []
        picked = []
        # A small arena first (3 blocks: found or not in a tenth of the time and memory), then
        # up to `max_arenas` of the full size; each is allocated while the earlier ones are held
        # (a freed arena would come back at the same addresses).
        sizes = ([3 * need] if arena_bytes > 3 * need else []) + [arena_bytes] * max(1, int(max_arenas))
        for want in sizes:
            free, _total = torch.cuda.mem_get_info(hip.device)
            size = min(want, int(free * 0.45)) // (2 << 20) * (2 << 20)
            if size < 2 * need:
                break
            arena = _new_arena(hip, size)
            if arena is None:
                break
            held.append(arena)
            times, offs, pad = hip._probe_windows(arena, size, need, b, rows * 8)
            self.info["probes"] += len(times)
            coarse_ms += [times[o] for o in offs]
            candidates += [(t, len(held) - 1, pad + o) for o, t in times.items()]
            med = float(np.median(coarse_ms))
            picked = self._pick(candidates, med * (1.0 - min_gain), need, slots)
            if len(picked) >= slots:
                break
        self.info["arenas"] = len(held)
        if coarse_ms:
            med = float(np.median(coarse_ms))
            self.info["probe_median_GBps"] = need / (med * 1e-3) / 1e9
            self.info["window_GBps"] = [need / (t * 1e-3) / 1e9 for t, _a, _o in picked]
            for _t, a, off in picked:
                self.windows.append((held[a], held[a].data_ptr() + off))
                _note_placed(hip.device, held[a].data_ptr() + off, need)
        # arenas without a window go back to the device with this frame (one hipFree each)
        del held, candidates
        self.free = list(range(len(self.windows)))
        self.info["slots"] = len(self.windows)
        self.info["arena_bytes_kept"] = sum({id(a): a.numel() for a, _p in self.windows}.values())
        if self.windows:
            _LOG.info("optiland_amd: record pool for %s blocks of %d MiB on %s keeps %d MiB of "
                      "device memory (given back after %g s without use; "
                      "HipSystem.enable_record_pool(0) or OPTILAND_HIP_PLACED_RECORDS=0 turns "
                      "pools off)", len(self.windows), need >> 20, hip.device,
                      self.info["arena_bytes_kept"] >> 20, _POOL_CONFIG["idle_s"])

    def idle(self) -> bool:
        """No block lent out (the windows' arenas may go)."""
        with self.lock:
            return len(self.free) == len(self.windows)

    @staticmethod
    def _pick(candidates, limit_ms, need, slots):
        """The fastest non-overlapping windows at least `min_gain` under the median."""
        out = []
        for t, a, off in sorted(candidates):
            if t > limit_ms or len(out) >= slots:
                break
            if all(a != a2 or abs(off - o2) >= need for _t2, a2, o2 in out):
                out.append((t, a, off))
        return out

    def _give_back(self, index):
        with self.lock:
            self.free.append(index)
            self.last_used = time.monotonic()

    def acquire(self):
        """A (rows, 8, stride) record block on a free window, or None when all are lent out."""
        with self.lock:
            self.last_used = time.monotonic()
            if not self.free:
                return None
            index = self.free.pop()
        arena, ptr = self.windows[index]
        lease = _Lease(self, index, arena, ptr, self.need)
        flat = torch.as_tensor(lease, device=self.device)
        del lease  # (the tensor's storage holds the only reference now)
        return flat.view(self.dtype).view(self.rows, 8, self.stride)


def release_record_pools(only_idle: bool = False) -> int:
    """Drop the record pools of this process (their arenas go back to the device as soon as no
    lent block holds them); `only_idle`: those with no block lent out and no request for
    `idle_s` seconds.  Returns the number of pools dropped."""
    now, limit, dropped = time.monotonic(), _POOL_CONFIG["idle_s"], 0
    with _POOL_LOCK:
        for key in list(_RECORD_POOLS):
            pool = _RECORD_POOLS[key]
            if only_idle and not (getattr(pool, "windows", None) and pool.idle()
                                  and now - getattr(pool, "last_used", now) > limit):
                continue
            del _RECORD_POOLS[key]
            _SHAPE_SEEN[key] = 1          # the next request of the shape is "the second" again
            dropped += 1
    return dropped


def record_pool_stats() -> dict:
    """What the placed-record machinery holds: `placed_bytes` = device memory of the library's
    arenas (pools AND blocks of `alloc_record_placed` that are still alive), `pools` = one
    entry per pooled shape."""
    with _POOL_LOCK:
        pools = [{"device": k[0], "rays": k[1], "dtype": str(k[2]).split(".")[-1], "rows": k[3],
                  "slots": len(getattr(p, "windows", ())),
                  "lent": len(getattr(p, "windows", ())) - len(getattr(p, "free", ())),
                  "arena_bytes": getattr(p, "info", {}).get("arena_bytes_kept", 0)}
                 for k, p in _RECORD_POOLS.items()]
    return {"placed_bytes": int(_Arena.live_bytes), "pools": pools,
            "idle_s": _POOL_CONFIG["idle_s"]}


_SWEEPER = {"thread": None}


def _start_sweeper() -> None:
    """A daemon thread that gives idle pools back (a process that stops tracing never calls
    `alloc_record` again, so nobody else would)."""
    if _SWEEPER["thread"] is not None or not _POOL_CONFIG["idle_s"]:
        return

    def run():
        while True:
            time.sleep(max(_POOL_CONFIG["idle_s"] / 4.0, 0.05) if _POOL_CONFIG["idle_s"] else 5.0)
            try:
                if _POOL_CONFIG["idle_s"]:
                    release_record_pools(only_idle=True)
            except Exception:  # noqa: BLE001 - never take the process down from here
                pass

    t = threading.Thread(target=run, name="optiland_amd-pool-sweeper", daemon=True)
    _SWEEPER["thread"] = t
    t.start()


class HipSystem:
    """A surface table resident on one GPU (wraps `ol_system`)."""

    def __init__(self, table: SystemTable, device=None):
        self.lib = _capi.load()
        self.device = _require_gpu(device)
        self.table = table
        surf = np.ascontiguousarray(table.surfaces)
        assert surf.dtype.itemsize == C.sizeof(_capi.SurfaceDesc), "ABI struct mismatch"
        optics = np.ascontiguousarray(table.optics)
        coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
        handle = C.c_void_p()
        with self._device_ctx():
            rc = self.lib.ol_system_create(
                surf.ctypes.data, surf.shape[0],
                coeffs.ctypes.data if coeffs.size else None, coeffs.size,
                optics.ctypes.data, optics.shape[1], C.byref(handle))
        self._check(rc, "ol_system_create")
        self._handle = handle
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)

    # The three places that know the library is the HIP one and the tensors are device
    # tensors.  (tests/_hostmath.HostMathEngine overrides them -- and the constructor -- to
    # run this class's argument checking and marshalling on CPU tensors against the
    # host build of the kernel source; the product class only ever accepts a HIP device.)
    def _device_ctx(self):
        return torch.cuda.device(self.device)

    def _stream(self) -> int:
        return _stream_ptr(self.device)

    def _check(self, rc: int, what: str) -> None:
        _capi.check(rc, what, self.lib)

    def update(self, table: SystemTable) -> bool:
        """`ol_system_update`: rewrite this system's device tables IN PLACE from `table`
        (no allocation; the copies are queued on the current stream, after everything that
        still reads the old tables).  False -- nothing changed -- when the new table does not
        fit the existing allocations or the library predates ABI 6; the caller then builds a
        new `HipSystem`."""
        if not getattr(self, "_handle", None) or not hasattr(self.lib, "ol_system_update"):
            return False
        surf = np.ascontiguousarray(table.surfaces)
        optics = np.ascontiguousarray(table.optics)
        coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
        if surf.shape[0] != self.table.num_surfaces or optics.shape[1] != self.table.optics.shape[1]:
            return False
        with self._device_ctx():
            rc = self.lib.ol_system_update(
                self._handle, surf.ctypes.data, surf.shape[0],
                coeffs.ctypes.data if coeffs.size else None, coeffs.size,
                optics.ctypes.data, optics.shape[1], self._stream())
        if rc == -2:  # OL_EUNSUPPORTED: does not fit
            return False
        self._check(rc, "ol_system_update")
        self.table = table
        return True

    def close(self):
        if getattr(self, "_handle", None):
            self.lib.ol_system_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __reduce__(self):
        """copy.deepcopy / pickle: a device handle cannot be copied bit for bit, so the
        copy is a fresh `ol_system` built from the same table on the same device.  The
        drop-in caches engines on reference objects (tracer, SurfaceGroup) that the
        reference deep-copies (tolerancing, optimisation, paraxial_to_thick ...)."""
        return (HipSystem, (self.table, str(self.device)))

    @property
    def num_surfaces(self) -> int:
        return self.table.num_surfaces

    # ------------------------------------------------------------------ trace
    @staticmethod
    def record_stride(n: int, itemsize: int) -> int:
        """Plane stride (elements) of the record block for n rays.

        Big planes are aligned to 2 MiB: on MI355X the 104-plane store pattern of a
        1e7-ray record-all trace sustains ~4 % more HBM bandwidth that way than with a
        256-byte-aligned stride (tools/microbench/rw_stride.hip: 5.29 vs 5.07 TB/s).
        Small planes only pay 256 B / 16 KiB so that a 100-ray trace does not allocate
        hundreds of MiB of padding.  OPTILAND_RECORD_ALIGN (bytes) overrides.
        """
        plane = n * itemsize
        env = os.environ.get("OPTILAND_RECORD_ALIGN")
        if env:
            align = max(int(env), 256)
        elif plane >= 16 << 20:
            align = 2 << 20
        elif plane >= 256 << 10:
            align = 16 << 10
        else:
            align = 256
        a = align // itemsize
        stride = max((n + a - 1) // a * a, _VEC_PAD)
        skew = os.environ.get("OPTILAND_RECORD_SKEW")  # bytes added to the aligned stride
        if skew:
            stride += max(int(skew), 0) // 256 * 256 // itemsize
        return stride

    @staticmethod
    def enable_record_pool(slots=2, min_bytes: int = 256 << 20) -> None:
        """How `alloc_record` (of every system of this process) serves block shapes of at least
        `min_bytes` -- the blocks of traces whose record goes to a user.  `slots` = n: n PLACED
        blocks (`RecordPool`) per shape may be alive at a time from the first request on;
        "auto" (the default): two, from the SECOND request of a shape on and only while the
        device has memory to spare; 0 = off (the pools and their arenas go).  Further blocks
        are ordinary allocations; the two most recently used shapes per device are kept."""
        with _POOL_LOCK:
            _POOL_CONFIG["slots"] = "auto" if slots == "auto" else max(int(slots), 0)
            _POOL_CONFIG["min_bytes"] = int(min_bytes)
            if not _POOL_CONFIG["slots"]:
                _RECORD_POOLS.clear()
            _SHAPE_SEEN.clear()
            _POOL_COOLDOWN.clear()

    @staticmethod
    def reset_record_pool() -> None:
        """Back to the default policy (OPTILAND_HIP_PLACED_RECORDS, else "auto"); pools go."""
        with _POOL_LOCK:
            _RECORD_POOLS.clear()
            HipSystem.enable_record_pool(_default_slots())

    def _auto_arena_bytes(self, need: int):
        """Arena size of an "auto" pool, or None when the device cannot spare one: at least
        half of the device memory must be free, and the arena takes at most a quarter of
        that."""
        free, total = torch.cuda.mem_get_info(self.device)
        if free < total // 2:
            return None
        size = min(max(3 * need, 40 << 30), free // 4)
        return size if size >= 2 * need else None

    def _pool_for(self, n: int, dtype, rows: int, need: int):
        """The `RecordPool` that serves this shape, or None (policy: see `enable_record_pool`)."""
        slots = _POOL_CONFIG["slots"]
        if not slots or need < _POOL_CONFIG["min_bytes"] or self.device.type != "cuda" \
                or not hasattr(self.lib, "ol_stream_fill"):
            return None
        with _POOL_LOCK:
            return self._pool_for_locked(n, dtype, rows, need, slots)

    def _pool_for_locked(self, n, dtype, rows, need, slots):
        dev = self.device.index
        key = (dev, int(n), dtype, rows)
        pool = _RECORD_POOLS.get(key)
        if pool is not None:
            _RECORD_POOLS.move_to_end(key)     # least recently USED goes first
            return pool
        auto = slots == "auto"
        seen = _SHAPE_SEEN[key] = _SHAPE_SEEN.get(key, 0) + 1
        if auto and seen < 2:
            return None                        # a one-off trace does not pay for a probe
        left = _POOL_COOLDOWN.get(dev, 0)
        if left > 0:                           # a pool was evicted a moment ago: a workload that
            _POOL_COOLDOWN[dev] = left - 1     # cycles through many shapes is served plain
            return None
        arena_bytes = None
        if auto:
            arena_bytes = self._auto_arena_bytes(need)
            if arena_bytes is None:
                return None
        mine = [k for k in _RECORD_POOLS if k[0] == dev]
        if len(mine) >= _POOL_CONFIG["max_pools"]:
            _RECORD_POOLS.pop(mine[0])         # the least recently used shape (and its arenas)
            _POOL_COOLDOWN[dev] = _POOL_CONFIG["cooldown"]
        # (a pool whose probe found no window stays registered -- empty, without arenas -- so
        # that the shape is not probed again)
        # (one arena in three to seven holds no fast window, profiles/r05_vmm_junctions.txt,
        # r04_placement_attempts.txt: up to three are tried; those without one go straight back)
        pool = _RECORD_POOLS[key] = RecordPool(self, n, dtype, rows, 2 if auto else slots,
                                               arena_bytes=arena_bytes, max_arenas=3)
        if getattr(pool, "windows", None):
            _start_sweeper()
        return pool

    def alloc_record(self, n: int, dtype, rows: int | None = None) -> torch.Tensor:
        rows = self.num_surfaces if rows is None else rows
        b = torch.empty((), dtype=dtype).element_size()
        stride = self.record_stride(n, b)
        pool = self._pool_for(n, dtype, rows, rows * 8 * stride * b)
        if pool is not None:
            block = pool.acquire()
            if block is not None:
                return block
        try:
            return torch.empty((rows, 8, stride), dtype=dtype, device=self.device)
        except torch.OutOfMemoryError:
            # the device is full: the pools' arenas are the first thing to go
            if not release_record_pools():
                raise
            import gc
            gc.collect()
            return torch.empty((rows, 8, stride), dtype=dtype, device=self.device)

    def _probe_windows(self, arena: torch.Tensor, size: int, need: int, b: int, planes: int):
        """(times [ms] by byte offset, the coarse offsets, pad): `ol_stream_fill` -- the record
        block's own store pattern without arithmetic -- timed over candidate windows of `arena`
        (coarse pass in quarter-block steps, a fine pass around the best)."""
        stream = self._stream()
        # (a block the caching allocator carved out of an older segment is only 512 B
        # aligned: windows start on 2 MiB boundaries of the ADDRESS, like plain blocks)
        pad = (-arena.data_ptr()) % (2 << 20)
        base = arena.data_ptr() + pad
        size -= pad + (2 << 20)

        def fill_ms(off, reps=2):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            for k in range(1 + reps):
                if k == 1:
                    e0.record()
                self._check(self.lib.ol_stream_fill(C.c_void_p(base + off), need, b, planes, 0,
                                                    stream), "ol_stream_fill")
            e1.record()
            torch.cuda.synchronize(self.device)
            return e0.elapsed_time(e1) / reps

        with self._device_ctx():
            for _ in range(30):  # past the clock transient of the first launches
                self.lib.ol_stream_fill(C.c_void_p(base), need, b, planes, 0, stream)
            last = size - need
            coarse = max(need // 4 // (2 << 20) * (2 << 20), 2 << 20)
            offs = list(range(0, last + 1, coarse))
            times = {o: fill_ms(o) for o in offs}
            best = min(times, key=times.get)
            fine = max(coarse // 4 // (2 << 20) * (2 << 20), 2 << 20)
            for o in range(max(best - coarse + fine, 0), min(best + coarse, last + 1), fine):
                if o not in times:
                    times[o] = fill_ms(o)
        return times, offs, pad

    def alloc_record_placed(self, n: int, dtype, rows: int | None = None,
                            arena_bytes: int | None = None, min_gain: float = 0.04,
                            max_arenas: int = 3, time_budget_s: float | None = None):
        """A record block PLACED where this part writes it fastest -- for callers that reuse
        one block over many traces (`bench.py`, a sharded step loop, `GraphedTrace`).

        Measured on MI355X (profiles/r04_window_scan_*.txt, r04_block_placement_*.txt): the
        record-all store pattern -- 104 planes of 40 MB written concurrently -- sustains
        5.7-5.8 TB/s inside most 4 GiB windows of device memory and **7.0 TB/s** in windows
        that straddle certain boundaries of the physical backing (every 32 GiB of a 128 GiB
        allocation on the boxes measured; the same windows for the arithmetic-free pattern of
        `ol_stream_fill`, correlation 0.96-0.998; stable while the allocation lives).  Where
        such a boundary falls is the driver's business, so it is FOUND: an arena is allocated,
        `ol_stream_fill` times this block's own store pattern over candidate offsets (coarse
        pass, then a fine one around the best; ~0.1 s), and the block is a view of the fastest
        window.  An arena need not contain such a window (one process in five on the boxes
        measured, profiles/r04_placement_attempts.txt): up to `max_arenas` are tried, each
        allocated while the earlier ones are still held; the others are released once the choice
        is made.  A plain allocation when no window is at least `min_gain` faster than
        the median one.  `time_budget_s`: no further arena is tried once that much time has
        gone into probing (`info["gave_up"]`; a sharded run does not wait for its slowest
        rank's third arena).  Returns (record, info).  The view pins its arena: not for results
        that are handed to a user (those come from `alloc_record`, i.e. an ordinary
        allocation or, when `enable_record_pool` is on, a window LENT by a `RecordPool`)."""
        rows = self.num_surfaces if rows is None else rows
        b = torch.empty((), dtype=dtype).element_size()
        stride = self.record_stride(n, b)
        need = rows * 8 * stride * b
        info = {"placed": False, "block_bytes": need}
        if not hasattr(self.lib, "ol_stream_fill") or self.device.type != "cuda" \
                or need < (256 << 20):
            return self.alloc_record(n, dtype, rows), info
        if arena_bytes is None:
            env = os.environ.get("OPTILAND_HIP_RECORD_ARENA_GIB")
            arena_bytes = int(float(env) * (1 << 30)) if env else max(3 * need, 40 << 30)
        def probe(arena, size):
            return self._probe_windows(arena, size, need, b, rows * 8)

        held, coarse_ms, chosen, probes = [], [], None, 0
        t_start = time.perf_counter()
        for _attempt in range(max(1, int(max_arenas))):
            if held and time_budget_s is not None \
                    and time.perf_counter() - t_start > float(time_budget_s):
                info["gave_up"] = True
                break
            free, _total = torch.cuda.mem_get_info(self.device)
            size = min(arena_bytes, int(free * 0.45)) // (2 << 20) * (2 << 20)
            if size < 2 * need:
                break
            arena = _new_arena(self, size)   # (the library's own memory: never torch's cache)
            if arena is None:
                break
            held.append(arena)
            times, offs, pad = probe(arena, size)
            probes += len(times)
            coarse_ms += [times[o] for o in offs]
            best = min(times, key=times.get)
            if chosen is None or times[best] < chosen[0]:
                chosen = (times[best], len(held) - 1, pad, best)
            if chosen[0] <= float(np.median(coarse_ms)) * (1.0 - min_gain):
                break
        if chosen is None:
            del held
            return self.alloc_record(n, dtype, rows), info
        med = float(np.median(coarse_ms))
        t_best, which, pad, off = chosen
        info.update(arena_bytes=int(held[which].numel()), arenas_tried=len(held), probes=probes,
                    probe_best_ms=t_best, probe_median_ms=med, window_offset_bytes=off,
                    probe_best_GBps=need / (t_best * 1e-3) / 1e9,
                    probe_median_GBps=need / (med * 1e-3) / 1e9)
        placed = t_best <= med * (1.0 - min_gain)
        info["probe_seconds"] = time.perf_counter() - t_start
        arena = held[which] if placed else None
        del held   # arenas that were not chosen go back to the device (one hipFree each)
        if not placed:
            return self.alloc_record(n, dtype, rows), info
        info["placed"] = True
        rec = _arena_view(arena, pad + off, need).view(dtype).view(rows, 8, stride)
        _note_placed(self.device, rec.data_ptr(), need)
        return rec, info

    def trace(self, rays, wavelength_index: int = 0, record=True, prt: torch.Tensor | None = None,
              first: int = 0, last: int | None = None, write_rays: bool | None = None,
              check_status: bool = True, prt_identity: bool = False,
              defer_status: bool = False, zero_status: bool = True,
              spot=None, record_first: int | None = None,
              nonunit_directions: bool = False) -> TraceResult:
        """Launch the fused trace.

        rays: sequence of 8 contiguous 1-D device tensors (x,y,z,L,M,N,i,opd) of one
        dtype -- the initial state.  record: True (allocate), False/None, or a
        preallocated (rows, 8, stride) tensor.  prt: (9, n) real or (18, n) real+imaginary
        tensor (the latter is required behind retarder coatings), read-modify-write.
        prt_identity: the PRT buffer is uninitialised and the trace starts from the
        identity (a fresh PolarizedRays) -- no fill, no read.
        With write_rays (default: only when nothing is recorded) the final state is
        written back into `rays` in place, like SurfaceGroup.trace mutates its rays.
        spot: optional (slots, cx, cy) -- `slots` from `alloc_spot_slots()`; the launch
        then also accumulates the masked spot moments of the final state about
        (cx, cy) as an epilogue of the same kernel (`ol_trace_ex`); read them with
        `reduce_spot_slots(slots)`.
        nonunit_directions (polarised traces, `OL_TRACE_NONUNIT_K`): the caller's direction
        cosines are not unit vectors (the reference's iterative / robust aimers) -- the PRT
        update then reproduces the reference's algebra on them as they are
        (rays/polarized_rays.py:136-202).
        """
        rays = list(rays)
        if len(rays) != 8:
            raise ValueError("rays must be 8 planes: x,y,z,L,M,N,i,opd")
        n = int(rays[0].numel())
        dtype = rays[0].dtype
        if dtype not in _DT:
            raise TypeError(f"unsupported ray dtype {dtype}")
        for t in rays:
            if t.device != self.device or t.dtype != dtype or t.numel() != n or not t.is_contiguous():
                raise ValueError("ray planes must be contiguous, same dtype/size, on the system's device")
        last = self.num_surfaces - 1 if last is None else last
        # record_first (ABI 6): rows start at that surface instead of `first`
        rec_first = first if record_first is None else max(int(record_first), first)
        if rec_first > last:
            raise ValueError("record_first beyond the last traced surface")
        rows = last - rec_first + 1
        rec = None
        if record is True:
            rec = self.alloc_record(n, dtype, rows)
        elif isinstance(record, torch.Tensor):
            rec = record
            if rec.dtype != dtype or rec.dim() != 3 or rec.shape[0] < rows or rec.shape[1] != 8 \
                    or rec.shape[2] < n or not rec.is_contiguous():
                raise ValueError("record must be a contiguous (rows, 8, stride>=n) tensor")
        if write_rays is None:
            write_rays = rec is None
        flags = (S.TRACE_WRITE_RAYS if write_rays else 0) | S.TRACE_COMPACT \
            | _few_waves_flag(rec) | _hot_loop_flag(rec)
        if prt is not None:
            if prt.dtype != dtype or prt.dim() != 2 or prt.shape[0] not in (9, 18) \
                    or prt.shape[1] != n or not prt.is_contiguous():
                raise ValueError("prt must be a contiguous (9, n) [real] or (18, n) "
                                 "[real + imaginary] tensor of the ray dtype")
            if prt.shape[0] == 18:
                flags |= S.TRACE_PRT_COMPLEX
            if prt_identity:  # write-only PRT: starts from I inside the kernel
                flags |= S.TRACE_PRT_IDENTITY
            if nonunit_directions:
                flags |= S.TRACE_NONUNIT_K
        ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in rays])
        extras = ex = None
        if rec_first != first:
            ex = _capi.TraceExtras(None, 0.0, 0.0, rec_first, 0)
        if spot is not None:
            if rec_first != first:
                raise ValueError("spot epilogue and record_first cannot be combined")
            slots, cx, cy = spot
            if slots.dtype != torch.float64 or slots.numel() != 8 * _capi.SPOT_SLOTS \
                    or slots.device != self.device or not slots.is_contiguous():
                raise ValueError("spot slots must come from alloc_spot_slots()")
            ex = _capi.TraceExtras(slots.data_ptr(), float(cx), float(cy), 0, 0)
        # opt-in reference stop rule (SURF_REFERENCE_NEWTON): the iteration counts of this batch
        ref_newton = self.table.reference_newton_surfaces(first, last) if n else []
        if ref_newton:
            if spot is not None:
                raise ValueError("the spot epilogue is not available on a range with "
                                 "reference-rule Newton surfaces")
            iters = self._newton_counts(ptrs, _DT[dtype], n, int(wavelength_index), int(first),
                                        ref_newton)
            if ex is None:
                ex = _capi.TraceExtras(None, 0.0, 0.0, 0, 0)
            ex.newton_iterations = iters.data_ptr()
        if ex is not None:
            extras = C.byref(ex)
        if check_status and zero_status:  # zero_status=False: keep bits set by ray generation
            self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_trace_ex(
                self._handle, _DT[dtype], n, ptrs, int(wavelength_index),
                rec.data_ptr() if rec is not None else None,
                int(rec.shape[2]) if rec is not None else 0,
                prt.data_ptr() if prt is not None else None,
                int(first), int(last), flags,
                self._status.data_ptr() if check_status else None,
                extras, self._stream())
        self._check(rc, "ol_trace")
        # defer_status: the kernel still ORs its bits into self._status, but the caller
        # reads them back later (together with other device-side checks)
        status = int(self._status.item()) if (check_status and not defer_status) else 0
        self.raise_for_status(status)
        return TraceResult(n, rays, rec, prt, status, rec_first if rec is not None else first,
                           last)

    # ---- reference-rule Newton surfaces (ABI 11; newton_raphson.py:137-166) -----------------
    # `newton_count_hook(iters)`: called after every counting / verifying launch with the
    # (2 S,) int32 device tensor -- a batch that is SHARDED over ranks takes the maximum over
    # its shards there (distributed.py: one small all-reduce), because the reference's count is
    # a property of the whole batch.
    newton_count_hook = None

    def _newton_counts(self, ptrs, dt: int, n: int, wl: int, first: int, surfaces):
        """The (2 S,) int32 device tensor `ol_trace_extras.newton_iterations` of one trace call:
        for every reference-rule Newton surface of the range, in order, the number K of updates
        the reference's lockstep loop makes on THIS batch -- the first k at which every ray is
        below the surface's tolerance, `max_iter` when a ray is NaN or never gets there."""
        S_ = self.num_surfaces
        iters = torch.zeros(2 * S_, dtype=torch.int32, device=self.device)
        hook = self.newton_count_hook
        stream = self._stream()

        def launch(s, verify):
            if n:   # (an EMPTY shard launches nothing and still takes part in the exchange)
                with self._device_ctx():
                    rc = self.lib.ol_newton_count(self._handle, dt, n, ptrs, wl, first, int(s),
                                                  iters.data_ptr(), 1 if verify else 0, stream)
                self._check(rc, "ol_newton_count")
            if hook is not None:
                hook(iters)

        for s in surfaces:
            launch(s, False)
        # A ray's first k below the tolerance bounds K from below; it IS K unless a ray that was
        # below is above again at that k (rounding noise of the order of the tolerance).  One
        # launch checks the rule at the counts found, for all surfaces at once; the rare repair
        # moves the first offending count up by one and takes the later ones again.
        max_iter = self.table.surfaces["max_iter"]
        while True:
            launch(surfaces[-1], True)
            host = iters.cpu().numpy()   # (the one synchronisation of the chain)
            bad = [s for s in surfaces if host[S_ + s] != 0 and host[s] < int(max_iter[s])]
            if not bad:
                return iters
            s0 = bad[0]
            iters[s0] += 1
            later = [s for s in surfaces if s > s0]
            for s in later:
                iters[s] = 0
            iters[S_:] = 0
            for s in later:
                launch(s, False)

    def newton_counts_of_an_empty_shard(self, first: int = 0, last: int | None = None):
        """A rank whose shard of a batch is empty traces nothing, but the counting chain of the
        other ranks exchanges after every launch (`newton_count_hook`): take part in exactly
        those exchanges (the loop control is the same on every rank because it reads the
        REDUCED counts)."""
        surfaces = self.table.reference_newton_surfaces(first, last)
        if surfaces and self.newton_count_hook is not None:
            self._newton_counts(None, 0, 0, 0, int(first), surfaces)

    def can_trace_generate(self, field_planes: bool = False) -> bool:
        """`ol_trace_generate` serves this launch: the table carries generator scalars.  Per-ray
        field planes (`field_planes`) and apodized pupils are one launch too -- unpolarised
        traces since ABI 8, polarised ones since ABI 10.  Not with reference-rule Newton
        surfaces: their counting launches read the generated rays (two launches)."""
        rg = self.table.raygen
        if not rg or not hasattr(self.lib, "ol_trace_generate"):
            return False
        if self.table.reference_newton_surfaces():
            return False
        if field_planes or int(rg.get("apod_kind", 0)) != 0:
            return (self.table.polarization is None and not self.table.uses_polarization) \
                or hasattr(self.lib, "ol_trace_spot_batch")   # (an ABI-10 library)
        return True

    def trace_generate(self, px, py, wavelength_index: int = 0, *, field, vig=(1.0, 1.0),
                       record=True, record_first: int = 0, prt: torch.Tensor | None = None,
                       rays_out=None, flags: int = 0, zero_status: bool = True,
                       defer_status: bool = False, update_intensity=None,
                       spot=None) -> TraceResult:
        """`ol_trace_generate`: rays generated from the normalised pupil planes and traced
        through the whole system in one launch.  field: (hx, hy) floats -- ONE field point --
        or two device planes (per-ray fields; `vig` then floats or two planes, None =
        unvignetted).  spot: optional (slots, cx, cy) as in `trace` -- the masked
        image-plane moments as an epilogue of the same launch (one field point, unpolarised,
        no apodization).  record: True (allocate)
        or a preallocated (rows, 8, stride) block; rows start at surface `record_first`
        (0 = the generated rays themselves as the object row).  prt: write-only (9 | 18, n)
        buffer of a polarised trace.  rays_out: optional 8 planes for the final state.
        flags: `_capi.RAYGEN_*` (pupil range check, trace_generic pre-scaling).
        update_intensity: polarisation state dict (`is_polarized`, `Ex`, `Ey`, `phase_x`,
        `phase_y`) -- with a `prt`, `PolarizedRays.update_intensity` runs as an epilogue of the
        same launch and the result carries `updated_intensity` (n values).
        The result's `rays` are the planes of record row 0 when that row is recorded."""
        p = self._raygen_params()
        n = int(px.numel())
        dtype = px.dtype
        if dtype not in _DT:
            raise TypeError(f"unsupported ray dtype {dtype}")
        last = self.num_surfaces - 1
        record_first = int(record_first)
        if not 0 <= record_first <= last:
            raise ValueError("record_first outside the surface range")
        rows = last - record_first + 1
        if record is True:
            rec = self.alloc_record(n, dtype, rows)
        else:
            rec = record
            if not isinstance(rec, torch.Tensor) or rec.dtype != dtype or rec.dim() != 3 \
                    or rec.shape[0] < rows or rec.shape[1] != 8 or rec.shape[2] < n \
                    or not rec.is_contiguous():
                raise ValueError("record must be a contiguous (rows, 8, stride>=n) tensor")
        tflags = _few_waves_flag(rec) | _hot_loop_flag(rec)
        if prt is not None:
            if prt.dtype != dtype or prt.dim() != 2 or prt.shape[0] not in (9, 18) \
                    or prt.shape[1] != n or not prt.is_contiguous():
                raise ValueError("prt must be a contiguous (9, n) [real] or (18, n) "
                                 "[real + imaginary] tensor of the ray dtype")
            if prt.shape[0] == 18:
                tflags |= S.TRACE_PRT_COMPLEX
        res = TraceResult(n, None, rec, prt, 0, record_first, last)  # .rays: row 0, on demand
        if n == 0:
            return res
        planes = isinstance(field[0], torch.Tensor)
        if planes:
            vx, vy = vig if vig is not None else (None, None)
            if not isinstance(vx, torch.Tensor) and vx is not None:
                vx, vy = float(vx), float(vy)
            inp, keep = self._raygen_inputs(field[0], field[1], px, py, vx, vy, flags)
        else:
            inp, keep = self._raygen_inputs(float(field[0]), float(field[1]), px, py,
                                            float(vig[0]), float(vig[1]), flags)
        outp = None
        if rays_out is not None:
            self._check_out_planes(list(rays_out), n, dtype, "trace_generate")
            outp = (C.c_void_p * 8)(*[t.data_ptr() for t in rays_out])
        ex = _capi.TraceExtras(None, 0.0, 0.0, record_first, 0, None, None)
        if spot is not None:
            slots, cx, cy = spot
            if slots.dtype != torch.float64 or slots.numel() != 8 * _capi.SPOT_SLOTS \
                    or slots.device != self.device or not slots.is_contiguous():
                raise ValueError("spot slots must come from alloc_spot_slots()")
            ex.spot_slots, ex.cx, ex.cy = slots.data_ptr(), float(cx), float(cy)
        if update_intensity is not None and prt is not None and self.can_fuse_update_intensity():
            pol = update_intensity
            if pol.get("is_polarized"):
                st = _capi.PolarizationStateC(1, 0, pol["Ex"], pol["Ey"], pol["phase_x"],
                                              pol["phase_y"])
            else:
                st = _capi.PolarizationStateC(0, 0, 0.0, 0.0, 0.0, 0.0)
            res.updated_intensity = torch.empty(n, dtype=dtype, device=self.device)
            keep.append(st)
            ex.update_intensity_state = C.addressof(st)
            ex.updated_intensity = res.updated_intensity.data_ptr()
        extras = C.byref(ex)
        if zero_status:
            self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_trace_generate(
                self._handle, _DT[dtype], n, C.byref(p), C.byref(inp), int(wavelength_index),
                rec.data_ptr(), int(rec.shape[2]), outp,
                prt.data_ptr() if prt is not None else None, tflags,
                self._status.data_ptr(), extras, self._stream())
        self._check(rc, "ol_trace_generate")
        if not defer_status:
            res.status = int(self._status.item())
            self.raise_for_status(res.status)
        return res

    def alloc_spot_slots(self) -> torch.Tensor:
        """Zeroed [OL_SPOT_SLOTS, 8] float64 buffer for the spot epilogue of `trace`."""
        return torch.zeros((_capi.SPOT_SLOTS, 8), dtype=torch.float64, device=self.device)

    @staticmethod
    def reduce_spot_slots(slots: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """[slots, 8] -> the seven doubles of `ol_trace_spot` (sum of 0..5, max of 6)."""
        if out is None:
            out = torch.empty(7, dtype=torch.float64, device=slots.device)
        torch.sum(slots[:, :6], dim=0, out=out[:6])
        torch.amax(slots[:, 6], dim=0, out=out[6])
        return out

    def can_fuse_update_intensity(self) -> bool:
        """The library runs `update_intensity` inside `ol_trace_generate` (ABI >= 7)."""
        ok = getattr(self, "_abi7", None)
        if ok is None:
            ok = self._abi7 = bool(hasattr(self.lib, "ol_trace_generate")
                                   and self.lib.ol_abi_version() >= 7
                                   and os.environ.get("OPTILAND_HIP_FUSE_INTENSITY", "1") != "0")
        return ok

    def row0_planes(self, record: torch.Tensor, n: int):
        """The 8 planes of record row 0 as ray planes (zero-copy object row)."""
        return list(record[0, :, :n].unbind(0))

    @staticmethod
    def raise_for_status(status: int) -> None:
        """Device status bits -> the reference's exceptions (same texts)."""
        # raytrace/real_ray_tracer.py:156-173 (field is validated before the pupil)
        if status & S.STATUS_FIELD_RANGE:
            raise ValueError("Normalized field coordinates must be within (-1, 1)")
        if status & S.STATUS_PUPIL_RANGE:
            raise ValueError("Normalized pupil coordinates must be within (-1, 1)")
        if status & S.STATUS_ZERNIKE_RANGE:
            # optiland/geometries/zernike.py:262-266
            raise ValueError(
                "Zernike coordinates must be normalized "
                "to [-1, 1]. Consider updating the normalization "
                "radius to 1.1x the surface aperture."
            )
        if status & S.STATUS_CHEBYSHEV_RANGE:
            # optiland/geometries/chebyshev.py:235-240
            raise ValueError(
                "Chebyshev input coordinates must be normalized "
                "to [-1, 1]. Consider updating the normalization "
                "factors."
            )

    # ------------------------------------------------------------- ray source
    def _raygen_params(self):
        rg = self.table.raygen
        if not rg:
            raise ValueError("this SystemTable carries no ray-generation scalars")
        return _capi.RaygenParams(int(rg["object_infinite"]), int(rg.get("field_kind", 0)),
                                  rg["EPL"], rg["EPD"],
                                  float(rg.get("field_scale", rg["max_field"])), rg["offset"],
                                  rg["z_first"], float(rg.get("tele_dz", 0.0)),
                                  float(rg.get("apod_a", 0.0)), float(rg.get("apod_b", 0.0)),
                                  int(rg.get("apod_kind", 0)), 0)

    def _check_out_planes(self, planes, n, dtype, what):
        for t in planes:
            if t.device != self.device or t.dtype != dtype or t.numel() != n \
                    or not t.is_contiguous():
                raise ValueError(f"{what}: output planes must be contiguous, of the ray dtype, "
                                 f"{n} long and on the system's device")

    def _raygen_inputs(self, hx, hy, px, py, vx, vy, flags):
        """`ol_raygen_inputs` from tensors (per-ray planes) or floats (launch-uniform).
        Returns (struct, keep-alive list)."""
        n, dtype = int(px.numel()), px.dtype
        keep, ptr, scal = [], {}, {"hx": 0.0, "hy": 0.0, "vx": 1.0, "vy": 1.0}
        for name, v in (("hx", hx), ("hy", hy), ("px", px), ("py", py), ("vx", vx), ("vy", vy)):
            if isinstance(v, torch.Tensor):
                if v.dtype != dtype or v.numel() != n or v.device != self.device:
                    raise ValueError("ray generation: coordinate planes must share dtype and "
                                     "length and live on the system's device")
                v = v.contiguous()
                keep.append(v)
                ptr[name] = v.data_ptr()
            else:
                ptr[name] = None
                if v is not None:
                    scal[name] = float(v)
        for a, b in (("hx", "hy"), ("vx", "vy")):
            if (ptr[a] is None) != (ptr[b] is None):
                raise ValueError(f"ray generation: {a} and {b} must both be planes or both scalars")
        st = _capi.RaygenInputs(ptr["hx"], ptr["hy"], ptr["px"], ptr["py"], ptr["vx"], ptr["vy"],
                                scal["hx"], scal["hy"], scal["vx"], scal["vy"], int(flags), 0)
        return st, keep

    def generate_rays(self, hx, hy, px, py, vx=None, vy=None, out=None, *, flags: int = 0,
                      zero_status: bool = True):
        """On-device ray generation (paraxial aiming, angle fields).  `hx, hy` and
        `vx, vy` are per-ray tensors or floats (launch-uniform; None = unvignetted).
        Returns 7 planes.  `out`: optional list of >= 7 preallocated planes (an 8th, if
        present, is zero-filled as the opd plane) -- e.g. row 0 of a record block, so
        that the trace need not copy the object-surface row.  `flags`: `_capi.RAYGEN_*`
        (range checks land in the status word read by `raise_for_status`)."""
        p = self._raygen_params()
        n = int(px.numel())
        dtype = px.dtype
        if out is None:
            buf = torch.empty((7, max(n, 1)), dtype=dtype, device=self.device)
            planes = [buf[k, :n] for k in range(7)]
        else:
            planes = list(out[:8])
            self._check_out_planes(planes, n, dtype, "generate_rays")
        if n == 0:
            return planes[:7]
        inp, keep = self._raygen_inputs(hx, hy, px, py, vx, vy, flags)
        ptrs = (C.c_void_p * 8)(*([p_.data_ptr() for p_ in planes] + [None] * (8 - len(planes))))
        if zero_status and flags & (_capi.RAYGEN_CHECK_FIELD | _capi.RAYGEN_CHECK_PUPIL):
            self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_generate_rays(C.byref(p), _DT[dtype], n, C.byref(inp), ptrs,
                                           self._status.data_ptr(), self._stream())
        self._check(rc, "ol_generate_rays")
        return planes[:7]

    def can_pupil_points(self) -> bool:
        return hasattr(self.lib, "ol_pupil_points") \
            and os.environ.get("OPTILAND_HIP_DEVICE_PUPIL", "1") != "0"

    def pupil_points(self, name: str, num: int, dtype):
        """`ol_pupil_points`: the "hexapolar" (num = rings) or "uniform" (num = grid side)
        sampler of the reference evaluated on the device -- (x, y) planes of `dtype`, same
        values and order as `distribution.create_distribution(name).generate_points(num)`."""
        from . import distribution as D

        num = int(num)
        if name == "hexapolar":
            n, kind, first, offset = D.hexapolar_count(num), 0, None, None
        elif name == "uniform":
            f, o = D.uniform_rows(num)
            n, kind = int(o[-1]), 1
            first = torch.as_tensor(f, device=self.device)
            offset = torch.as_tensor(o, device=self.device)
        else:
            raise ValueError(f"no device sampler for {name!r}")
        x = torch.empty(n, dtype=dtype, device=self.device)
        y = torch.empty(n, dtype=dtype, device=self.device)
        with self._device_ctx():
            rc = self.lib.ol_pupil_points(
                kind, num, _DT[dtype], n,
                first.data_ptr() if first is not None else None,
                offset.data_ptr() if offset is not None else None,
                x.data_ptr(), y.data_ptr(), self._stream())
        self._check(rc, "ol_pupil_points")
        return x, y

    def polarized_intensity(self, prt, k0, i0, polarization: dict | None):
        n = int(i0.numel())
        dtype = i0.dtype
        if polarization and polarization.get("is_polarized"):
            st = _capi.PolarizationStateC(1, 0, polarization["Ex"], polarization["Ey"],
                                          polarization["phase_x"], polarization["phase_y"])
        else:
            st = _capi.PolarizationStateC(0, 0, 0.0, 0.0, 0.0, 0.0)
        out = torch.empty(n, dtype=dtype, device=self.device)
        kp = (C.c_void_p * 3)(*[t.data_ptr() for t in k0])
        self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_polarized_intensity(
                _DT[dtype], n, prt.data_ptr(), 1 if prt.shape[0] == 18 else 0, kp,
                i0.data_ptr(), C.byref(st), out.data_ptr(),
                self._status.data_ptr(), self._stream())
        self._check(rc, "ol_polarized_intensity")
        if int(self._status.item()) & S.STATUS_K_PARALLEL_X:
            # rays/polarized_rays.py:221-222
            raise ValueError("k-vector parallel to x-axis is not currently supported.")
        return out

    def wavefront_opd(self, params: dict, rays7, px, py, want_pupil: bool = True):
        """OPD in waves (+ pupil coordinates) against the chief-ray reference sphere."""
        n = int(px.numel())
        dtype = px.dtype
        p = _capi.WavefrontParams(**{k: float(params.get(k, 0.0)) for k, _ in
                                     _capi.WavefrontParams._fields_})  # nx,ny,nz default 0
        opd = torch.empty(n, dtype=dtype, device=self.device)
        pupil = torch.empty((3, n), dtype=dtype, device=self.device) if want_pupil else None
        rp = (C.c_void_p * 7)(*[t.data_ptr() for t in rays7])
        pp = (C.c_void_p * 3)(*[pupil[k].data_ptr() for k in range(3)]) if want_pupil else None
        with self._device_ctx():
            rc = self.lib.ol_wavefront_opd(C.byref(p), _DT[dtype], n, rp, px.data_ptr(),
                                           py.data_ptr(), opd.data_ptr(), pp,
                                           self._stream())
        self._check(rc, "ol_wavefront_opd")
        return opd, pupil

    def trace_opd(self, params: dict | None, px, py, wl_index: int, *, field, vig=(1.0, 1.0),
                  want_pupil: bool = True, moments: torch.Tensor | None = None,
                  check_status: bool = True, reference: torch.Tensor | None = None,
                  zero_status: bool = True):
        """Fused generate -> trace -> OPD (`ol_trace_opd`, fp64): the pupil points
        (px, py) of ONE field point straight to the OPD map against the reference sphere /
        plane `params` -- no ray planes.  Returns (opd_waves, intensity, pupil (3, n) or
        None, moments): `moments` = the 12 device sums of `ol_trace_opd` (tilt-fit moments,
        count / sum / sum of squares of the OPD over rays with i > 0), accumulated into
        the tensor passed in (zero it to start) or a fresh one."""
        p = self._raygen_params()
        n = int(px.numel())
        dtype = px.dtype
        if dtype != torch.float64:
            raise ValueError("trace_opd: wavefront work is fp64 only")
        if moments is None:
            moments = torch.zeros(_capi.OPD_MOMENTS, dtype=torch.float64, device=self.device)
        opd = torch.empty(n, dtype=dtype, device=self.device)
        inten = torch.empty(n, dtype=dtype, device=self.device)
        pupil = torch.empty((3, n), dtype=dtype, device=self.device) if want_pupil else None
        if n == 0:
            return opd, inten, pupil, moments
        inp, keep = self._raygen_inputs(float(field[0]), float(field[1]), px, py,
                                        float(vig[0]), float(vig[1]), 0)
        pp = (C.c_void_p * 3)(*[pupil[k].data_ptr() for k in range(3)]) if want_pupil else None
        if check_status and zero_status:
            self._status.zero_()
        with self._device_ctx():
            if reference is not None:
                # `reference`: what `wavefront_reference` left on the device (ol_trace_opd_dev)
                rc = self.lib.ol_trace_opd_dev(
                    self._handle, _DT[dtype], n, C.byref(p), C.byref(inp), reference.data_ptr(),
                    int(wl_index), opd.data_ptr(), inten.data_ptr(), pp, moments.data_ptr(),
                    self._status.data_ptr(), self._stream())
            else:
                w = _capi.WavefrontParams(**{k: float(params.get(k, 0.0)) for k, _ in
                                             _capi.WavefrontParams._fields_})
                rc = self.lib.ol_trace_opd(self._handle, _DT[dtype], n, C.byref(p), C.byref(inp),
                                           C.byref(w), int(wl_index), opd.data_ptr(),
                                           inten.data_ptr(), pp, moments.data_ptr(),
                                           self._status.data_ptr(), self._stream())
        self._check(rc, "ol_trace_opd")
        if check_status:
            self.raise_for_status(int(self._status.item()))
        return opd, inten, pupil, moments

    def can_wavefront_reference(self) -> bool:
        return hasattr(self.lib, "ol_wavefront_reference") \
            and os.environ.get("OPTILAND_HIP_DEVICE_REFERENCE", "1") != "0"

    def wavefront_reference(self, params: dict, wl_index: int, *, field, vig=(1.0, 1.0),
                            pupil_z: float = 0.0, planar: bool = False, want_chief: bool = False,
                            zero_status: bool = True):
        """`ol_wavefront_reference`: the chief ray of one field point traced on the device and
        turned into the reference sphere (centre = its image point, radius to (0, 0, pupil_z))
        or plane (`planar`) of the wavefront kernels -- LEFT ON THE DEVICE.  `params`: n_image,
        wavelength_um, ux, uy, half_epd.  Returns (reference, chief): `reference` = a float64
        device tensor (`[0:3]` centre, `[3]` radius) for `trace_opd(reference=...)`, `chief` = the
        chief ray's x, y, z, L, M, N, i, opd (8 values) or None.  No read-back."""
        p = self._raygen_params()
        ref = torch.empty(_capi.WAVEFRONT_REFERENCE_DOUBLES, dtype=torch.float64,
                          device=self.device)
        chief = torch.empty(8, dtype=torch.float64, device=self.device) if want_chief else None
        zero = torch.zeros(1, dtype=torch.float64, device=self.device)
        inp, keep = self._raygen_inputs(float(field[0]), float(field[1]), zero, zero,
                                        float(vig[0]), float(vig[1]), 0)
        w = _capi.WavefrontParams(**{k: float(params.get(k, 0.0)) for k, _ in
                                     _capi.WavefrontParams._fields_})
        if zero_status:
            self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_wavefront_reference(
                self._handle, _DT[torch.float64], C.byref(p), C.byref(inp), C.byref(w),
                float(pupil_z), 1 if planar else 0, int(wl_index), ref.data_ptr(),
                chief.data_ptr() if chief is not None else None, self._status.data_ptr(),
                self._stream())
        self._check(rc, "ol_wavefront_reference")
        return ref, chief

    def can_wavefront_fit(self) -> bool:
        return hasattr(self.lib, "ol_wavefront_fit")

    def wavefront_fit(self, kind: str, params: dict, rays8, px, py, *, trim_std: float = 3.0,
                      flavour: str = "torch", planar: bool = False) -> torch.Tensor:
        """`ol_wavefront_fit`: the reference sphere / plane `CentroidStrategy` (kind "centroid")
        or `BestFitStrategy` ("best_fit") derive from the traced bundle `rays8` = x, y, z, L, M,
        N, opd, intensity at the image surface (fp64 device planes), and the piston, as a chain
        of device reductions -- LEFT ON THE DEVICE.  `params`: n_image, wavelength_um, ux, uy,
        half_epd; `flavour`: which of the reference's backends to follow where they differ
        ("torch": std with n - 1, NaN-ignoring piston mean; "numpy": n, plain mean).
        Returns `reference`, a float64 device tensor for `wavefront_opd_fitted`: `[0:3]` centre
        / plane point, `[3]` radius, `[10:13]` plane normal, and in its last slot the status
        word of the fit (`fit_result` reads both back in one copy).  No read-back here."""
        n = int(px.numel())
        for t in (*rays8, px, py):
            if t.dtype != torch.float64 or t.numel() != n or not t.is_contiguous():
                raise ValueError("wavefront_fit: contiguous float64 planes of one length")
        nref = _capi.WAVEFRONT_REFERENCE_DOUBLES
        code = {"centroid": _capi.FIT_CENTROID, "best_fit": _capi.FIT_BEST_FIT}[kind]
        if n == 0:  # an empty bundle has no planes to point at: strategy.py:387 directly
            ref = torch.full((nref + 1,), float("nan"), dtype=torch.float64, device=self.device)
            ref[nref:].view(torch.int32)[:] = 0
            ref[nref:].view(torch.int32)[0] = _capi.FIT_NO_VALID
            return ref
        ref = torch.empty(nref + 1, dtype=torch.float64, device=self.device)
        work = torch.empty(_capi.WAVEFRONT_FIT_WORKSPACE_DOUBLES, dtype=torch.float64,
                           device=self.device)
        w = _capi.WavefrontParams(**{k: float(params.get(k, 0.0)) for k, _ in
                                     _capi.WavefrontParams._fields_})
        rp = (C.c_void_p * 8)(*[t.data_ptr() for t in rays8])
        flags = {"torch": _capi.FIT_STD_DDOF1 | _capi.FIT_PISTON_SKIPS_NAN, "numpy": 0}[flavour]
        with self._device_ctx():
            rc = self.lib.ol_wavefront_fit(code, C.byref(w), float(trim_std or 0.0),
                                           flags, 1 if planar else 0, n, rp,
                                           px.data_ptr(), py.data_ptr(), work.data_ptr(),
                                           ref.data_ptr(), ref[nref:].data_ptr(), self._stream())
        self._check(rc, "ol_wavefront_fit")
        return ref

    @staticmethod
    def fit_result(reference: torch.Tensor):
        """(radius, status bits) of a `wavefront_fit` reference: ONE device-to-host copy."""
        host = reference.cpu()
        return float(host[3]), int(host[-1:].view(torch.int32)[0])

    @staticmethod
    def raise_for_fit_status(bits: int) -> None:
        """The reference's own errors (wavefront/strategy.py:387, 536, 334) for the bits
        `ol_wavefront_fit` left."""
        if bits & _capi.FIT_NO_VALID:
            raise ValueError("No valid ray samples found for best-fit geometry.")
        if bits & _capi.FIT_TOO_FEW:
            raise ValueError("Need at least 4 valid ray samples for best-fit.")
        if bits & _capi.FIT_SINGULAR:
            raise RuntimeError("Least-squares sphere fit failed: singular normal equations")
        if bits & _capi.FIT_NO_ALIVE:
            raise ValueError("No valid rays with non-zero intensity for OPD calculation.")

    def wavefront_opd_fitted(self, reference: torch.Tensor, rays7, px, py,
                             want_pupil: bool = True):
        """`ol_wavefront_opd_fitted`: OPD in waves (+ the reference-surface intersection
        points) of the bundle against a device-resident fitted reference."""
        n = int(px.numel())
        opd = torch.empty(n, dtype=torch.float64, device=self.device)
        pupil = torch.empty((3, n), dtype=torch.float64, device=self.device) if want_pupil \
            else None
        if n == 0:
            return opd, pupil
        rp = (C.c_void_p * 7)(*[t.data_ptr() for t in rays7])
        pp = (C.c_void_p * 3)(*[pupil[k].data_ptr() for k in range(3)]) if want_pupil else None
        with self._device_ctx():
            rc = self.lib.ol_wavefront_opd_fitted(n, rp, px.data_ptr(), py.data_ptr(),
                                                  reference.data_ptr(), opd.data_ptr(), pp,
                                                  self._stream())
        self._check(rc, "ol_wavefront_opd_fitted")
        return opd, pupil

    def pupil_fill(self, opd, intensity, cell: torch.Tensor, n_side: int, grid_size: int,
                   pupil_xy=None, plane=None) -> torch.Tensor:
        """`ol_pupil_fill`: A exp(-i 2 pi OPD) of the compacted samples scattered into the
        zero-padded (grid_size, grid_size) complex128 FFT grid (psf/fft.py:101-160);
        `cell` = int32 row-major cell index of every sample in the n_side x n_side sample
        grid; `pupil_xy` + `plane` = (a, b, c): subtract a + b X + c Y (tilt removal)."""
        n = int(opd.numel())
        if cell.dtype != torch.int32 or cell.numel() != n or cell.device != self.device:
            raise ValueError("pupil_fill: cell must be an int32 device tensor, one entry per sample")
        grid = torch.zeros((grid_size, grid_size), dtype=torch.complex128, device=self.device)
        co = None
        px_ptr = py_ptr = None
        if pupil_xy is not None:
            co = (C.c_double * 3)(*[float(v) for v in plane])
            px_ptr, py_ptr = pupil_xy[0].data_ptr(), pupil_xy[1].data_ptr()
        with self._device_ctx():
            rc = self.lib.ol_pupil_fill(_DT[opd.dtype], n, opd.data_ptr(), intensity.data_ptr(),
                                        px_ptr, py_ptr, co, cell.contiguous().data_ptr(),
                                        int(n_side), int(grid_size), grid.data_ptr(),
                                        self._stream())
        self._check(rc, "ol_pupil_fill")
        return grid

    def trace_spot(self, px, py, wl_index: int, *, field=None, hx=None, hy=None, vig=(1.0, 1.0),
                   vx=None, vy=None, center=(0.0, 0.0), hits=None, out=None,
                   check_status: bool = True, flags: int = 0):
        """Fused generate -> trace -> reduce (`ol_trace_spot`): one kernel, no ray planes.

        `field` = (Hx, Hy) launch-uniform, or per-ray `hx`, `hy` tensors; `vig` =
        launch-uniform (1 - vx, 1 - vy) or per-ray `vx`, `vy`; `center` = (cx, cy) the
        moments are taken about (global image coordinates); `hits` = optional list of 3
        preallocated planes for the image-plane x, y, intensity.  Returns a float64
        tensor {count, sum dx, sum dy, sum dx^2, sum dy^2, sum i, max r^2} (`out`, if
        given, is ACCUMULATED into -- zero it to start a new spot)."""
        p = self._raygen_params()
        if (field is None) == (hx is None):
            raise ValueError("give either field=(Hx, Hy) or per-ray hx, hy")
        n = int(px.numel())
        dtype = px.dtype
        if out is None:
            out = torch.zeros(7, dtype=torch.float64, device=self.device)
        if n == 0:  # nothing to add (an empty tensor has no device pointer to hand over)
            return out
        if field is not None:
            hx, hy = float(field[0]), float(field[1])
        if vx is None:
            vx, vy = float(vig[0]), float(vig[1])
        inp, keep = self._raygen_inputs(hx, hy, px, py, vx, vy, flags)
        hp = None
        if hits is not None:
            self._check_out_planes(hits, n, dtype, "trace_spot")
            hp = (C.c_void_p * 3)(*[h.data_ptr() for h in hits])
        if check_status:
            self._status.zero_()
        with self._device_ctx():
            rc = self.lib.ol_trace_spot(self._handle, _DT[dtype], n, C.byref(p), C.byref(inp),
                                        float(center[0]), float(center[1]), int(wl_index), hp,
                                        out.data_ptr(), self._status.data_ptr(),
                                        self._stream())
        self._check(rc, "ol_trace_spot")
        if check_status:
            self.raise_for_status(int(self._status.item()))
        return out

    def trace_spot_batch(self, px, py, cells, *, hits: bool = False, out=None,
                         check_status: bool = True, flags: int = 0):
        """A grid of fused spots in ONE launch (`ol_trace_spot_batch`, ABI 10): `cells` is a
        sequence of (Hx, Hy, vig_x, vig_y, cx, cy, wavelength index[, engine]); every cell
        traces the same pupil planes.  `engine`: another `HipSystem` of the SAME optic (one
        packed table per wavelength, as the drop-in keeps them) whose index rows the cell reads.  Returns (moments, hits): moments a (cells, 8) float64 device tensor (per
        cell the seven of `trace_spot`; `out`, if given, is accumulated into), hits None or a
        (cells, 3, stride) block of the image-plane x, y, intensity (use `[..., :n]`).  More
        cells than one launch takes (32) go out as several launches; the status word is read
        back ONCE."""
        p = self._raygen_params()
        n, dtype = int(px.numel()), px.dtype
        cells = list(cells)
        k = len(cells)
        if out is None:
            out = torch.zeros((k, 8), dtype=torch.float64, device=self.device)
        elif out.shape != (k, 8) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous (cells, 8) float64 tensor")
        hb = None
        stride = 0
        if hits:
            b = px.element_size()
            stride = (n + 16 // b - 1) // (16 // b) * (16 // b)   # planes stay 16-byte aligned
            hb = torch.empty((k, 3, max(stride, 1)), dtype=dtype, device=self.device)
        if n == 0 or k == 0:
            return out, hb
        inp, keep = self._raygen_inputs(0.0, 0.0, px, py, 1.0, 1.0, flags)  # (flags: _capi.RAYGEN_* | SPOT_*)
        if check_status:
            self._status.zero_()
        step = _capi.SPOT_BATCH_MAX_CELLS
        for c in cells:
            # a cell whose engine has been closed must not silently read THIS system's index
            # rows (another wavelength's) through a NULL `optics_of`
            if len(c) > 7 and c[7] is not None and getattr(c[7], "_handle", True) is None:
                raise _capi.HipExtensionError("trace_spot_batch: a cell's engine is closed")
        if getattr(self, "_handle", True) is None:
            raise _capi.HipExtensionError("trace_spot_batch: this engine is closed")
        with self._device_ctx():
            for lo in range(0, k, step):
                part = cells[lo: lo + step]
                arr = (_capi.SpotCell * len(part))(*[
                    _capi.SpotCell(float(c[0]), float(c[1]), float(c[2]), float(c[3]), float(c[4]),
                                   float(c[5]), int(c[6]), 0,
                                   c[7]._handle if len(c) > 7 and c[7] is not None
                                   and c[7] is not self else None) for c in part])
                rc = self.lib.ol_trace_spot_batch(
                    self._handle, _DT[dtype], n, C.byref(p), C.byref(inp), len(part), arr,
                    hb[lo].data_ptr() if hb is not None else None, stride, out[lo].data_ptr(),
                    self._status.data_ptr(), self._stream())
                self._check(rc, "ol_trace_spot_batch")
        if check_status:
            self.raise_for_status(int(self._status.item()))
        return out, hb

    def spot_moments(self, x, y, intensity, out=None):
        """Device reduction: returns float64 tensor [count, sx, sy, sxx, syy, count]
        (`out`: optional preallocated 6-element float64 tensor, zeroed here)."""
        if out is None:
            out = torch.zeros(6, dtype=torch.float64, device=self.device)
        else:
            out.zero_()
        with self._device_ctx():
            rc = self.lib.ol_spot_moments(_DT[x.dtype], int(x.numel()), x.data_ptr(),
                                          y.data_ptr(), intensity.data_ptr(), out.data_ptr(),
                                          self._stream())
        self._check(rc, "ol_spot_moments")
        return out

    def irradiance(self, x, y, power, x_edges: torch.Tensor, y_edges: torch.Tensor,
                   out: torch.Tensor | None = None) -> torch.Tensor:
        """numpy.histogram2d(x, y, [x_edges, y_edges], weights=power) over rays with
        power > 0 (`ol_irradiance`, analysis/irradiance.py:341-353).  Edges: ascending
        float64 device tensors; returns (nx, ny) float64 (`out` is accumulated into)."""
        for e in (x_edges, y_edges):
            if e.dtype != torch.float64 or e.device != self.device or e.numel() < 2:
                raise ValueError("edges must be float64 tensors on the system's device")
        xe, ye = x_edges.contiguous(), y_edges.contiguous()
        nx, ny = xe.numel() - 1, ye.numel() - 1
        if out is None:
            out = torch.zeros((nx, ny), dtype=torch.float64, device=self.device)
        with self._device_ctx():
            rc = self.lib.ol_irradiance(_DT[x.dtype], int(x.numel()), x.data_ptr(), y.data_ptr(),
                                        power.data_ptr(), xe.data_ptr(), nx, ye.data_ptr(), ny,
                                        out.data_ptr(), self._stream())
        self._check(rc, "ol_irradiance")
        return out

    def radial_energy(self, x, y, intensity, cx: float, cy: float, r_step: torch.Tensor,
                      out: torch.Tensor | None = None):
        """Energy per radius step about (cx, cy) (`ol_radial_energy`); `torch.cumsum` of
        the result is the reference's encircled-energy curve
        (analysis/encircled_energy.py:147-160).  `r_step`: ascending float64 device
        tensor; `out` (accumulated) or a new zeroed tensor of the same length."""
        if r_step.dtype != torch.float64 or r_step.device != self.device:
            raise ValueError("r_step must be a float64 tensor on the system's device")
        r_step = r_step.contiguous()
        if out is None:
            out = torch.zeros_like(r_step)
        with self._device_ctx():
            rc = self.lib.ol_radial_energy(_DT[x.dtype], int(x.numel()), x.data_ptr(),
                                           y.data_ptr(), intensity.data_ptr(), float(cx),
                                           float(cy), r_step.data_ptr(), int(r_step.numel()),
                                           out.data_ptr(), self._stream())
        self._check(rc, "ol_radial_energy")
        return out

    def spot_max_r2(self, x, y, intensity, cx: float, cy: float):
        out = torch.zeros(1, dtype=torch.float64, device=self.device)
        with self._device_ctx():
            rc = self.lib.ol_spot_max_r2(_DT[x.dtype], int(x.numel()), x.data_ptr(),
                                         y.data_ptr(), intensity.data_ptr(), float(cx),
                                         float(cy), out.data_ptr(), self._stream())
        self._check(rc, "ol_spot_max_r2")
        return out
