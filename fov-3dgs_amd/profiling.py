"""Per-stage kernel timing with HIP events recorded by the C library on the launch stream.

`StageTimer` hands the rasterizer one set of FR_NUM_STAGES + 1 events per forward call; nothing is
synchronised while the timed region runs. Durations are read afterwards (fr_event_elapsed_ms).
Stage names: _native.STAGES. Note: the "emit" interval also contains the frame's single host
synchronisation (instance-count read-back) and the binning-buffer callback.
"""
import ctypes as C

from . import _native, rasterizer


class StageTimer:
    def __init__(self, max_calls):
        self.lib = _native.load()
        self.n = len(_native.STAGES) + 1
        self.sets = []
        for _ in range(max_calls):
            arr = (C.c_void_p * self.n)()
            for i in range(self.n):
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
            for i, name in enumerate(_native.STAGES):
                rc = self.lib.fr_event_elapsed_ms(arr[i], arr[i + 1], C.byref(ms))
                d[name] = ms.value if rc == 0 else float("nan")
            out.append(d)
        return out

    def close(self):
        for arr in self.sets:
            for i in range(self.n):
                self.lib.fr_event_destroy(arr[i])
        self.sets = []
