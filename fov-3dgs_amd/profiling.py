"""Per-stage kernel timing with HIP events recorded by the C library on the launch stream.

`StageTimer` hands the rasterizer one set of FR_NUM_STAGE_EVENTS events per forward call; nothing is
synchronised while the timed region runs. Durations are read afterwards (fr_event_elapsed_ms).
Stage names: _native.STAGES. `stages` limits the events to the boundaries of the named stages: every event
record is a command of its own on the stream (~3 us each on MI355X: eight of them are 3.5 % of a 0.72 ms frame), so a
timed region that only needs one kernel's duration should only record that kernel's two.
"""
import ctypes as C

from . import _native, rasterizer


class StageTimer:
    def __init__(self, max_calls, stages=None):
        self.lib = _native.load()
        self.n = _native.NUM_STAGE_EVENTS
        self.stages = tuple(_native.STAGES) if stages is None else tuple(stages)
        wanted = set()
        for name in self.stages:
            i = self._first(name)
            wanted.update((i, i + 1))
        self.sets = []
        for _ in range(max_calls):
            arr = (C.c_void_p * self.n)()
            for i in range(self.n):
                arr[i] = self.lib.fr_event_create() if i in wanted else None
            self.sets.append(arr)
        self.used = 0

    @staticmethod
    def _first(name):
        return _native.STAGES.index(name)

    def _next(self):
        if self.used >= len(self.sets):
            return None
        arr = self.sets[self.used]
        self.used += 1
        return C.cast(arr, C.POINTER(C.c_void_p))

    def __enter__(self):
        rasterizer._stage_events_hook = self._next
        return self

    def __exit__(self, *exc):
        rasterizer._stage_events_hook = None

    def stage_ms(self):
        """-> list (one per recorded call) of dicts stage -> milliseconds."""
        out = []
        ms = C.c_float()
        for arr in self.sets[:self.used]:
            d = {}
            for name in self.stages:
                i = self._first(name)
                rc = self.lib.fr_event_elapsed_ms(arr[i], arr[i + 1], C.byref(ms))
                d[name] = ms.value if rc == 0 else float("nan")
            out.append(d)
        return out

    def close(self):
        for arr in self.sets:
            for i in range(self.n):
                if arr[i]:
                    self.lib.fr_event_destroy(arr[i])
        self.sets = []


class BackwardTimer:
    """The same for fr_backward calls: events around the tile pass (k_render_bwd), the per-Gaussian pass (k_preprocess_bwd) and,
    on the helper stream it runs on, the zero fill of the gradient tensors. stage_ms() -> dicts render_bwd / preprocess_bwd /
    fill_zero in milliseconds."""
    NAMES = (("render_bwd", 0, 1), ("preprocess_bwd", 1, 2), ("fill_zero", 3, 4))

    def __init__(self, max_calls):
        self.lib = _native.load()
        self.sets = []
        for _ in range(max_calls):
            arr = (C.c_void_p * 5)()
            for i in range(5):
                arr[i] = self.lib.fr_event_create()
            self.sets.append(arr)
        self.used = 0

    def _next(self):
        if self.used >= len(self.sets):
            return None
        arr = self.sets[self.used]
        self.used += 1
        return C.cast(arr, C.POINTER(C.c_void_p))

    def __enter__(self):
        rasterizer._bwd_events_hook = self._next
        return self

    def __exit__(self, *exc):
        rasterizer._bwd_events_hook = None

    def stage_ms(self):
        out = []
        ms = C.c_float()
        for arr in self.sets[:self.used]:
            d = {}
            for name, i, j in self.NAMES:
                rc = self.lib.fr_event_elapsed_ms(arr[i], arr[j], C.byref(ms))
                d[name] = ms.value if rc == 0 else float("nan")
            out.append(d)
        return out

    def close(self):
        for arr in self.sets:
            for i in range(5):
                if arr[i]:
                    self.lib.fr_event_destroy(arr[i])
        self.sets = []


class NativeCallTimer:
    """Rasterizer-only time of a training step ("plus rasterizer-only fwd and bwd ms"): torch events on the current
    stream around every native forward (fr_forward_begin + finish) and backward (fr_backward) call of the plain variants' autograd
    function while the context is active. ms() -> {"fwd": [...], "bwd": [...]} after a synchronize."""

    def __enter__(self):
        self.events = []
        rasterizer._call_events = self.events
        return self

    def __exit__(self, *exc):
        rasterizer._call_events = None

    def ms(self):
        out = {"fwd": [], "bwd": [], "fill": []}  # fill: the gradient tensors' zero fill on its side stream (rasterizer.prefill_gradients)
        for kind, e0, e1 in self.events:
            e1.synchronize()
            out[kind].append(e0.elapsed_time(e1))
        return out
