"""Wall-clock `Timer` with the call shape the reference's examples use
(`with prim3d.Timer("cuda marching cubes: {:.6f}s"):`, examples/sphere.py:14; class at
prim3d/misc/utils.py:41-116).  Like the reference it does NOT synchronise the device."""
import re
from time import perf_counter

_HAS_FLOAT_FIELD = re.compile(r"\{:[^}]*\df\}")


class TimerError(Exception):
    """Raised when a stopped timer is queried (same name and role as prim3d/misc/utils.py:34-38)."""

    def __init__(self, message="timer is not running"):
        self.message = message
        super().__init__(message)


class Timer:
    """Context manager / stopwatch.  `print_tmpl` without a `{:.nf}` field gets ` {:.3f}` appended."""

    def __init__(self, print_tmpl=None, start=True):
        if print_tmpl is None:
            print_tmpl = "{:.3f}"
        elif not _HAS_FLOAT_FIELD.search(print_tmpl):
            print_tmpl = print_tmpl + " {:.3f}"
        self.print_tmpl = print_tmpl
        self._origin = None  # None <=> stopped
        self._mark = None
        if start:
            self.start()

    @property
    def is_running(self):
        return self._origin is not None

    def start(self):
        now = perf_counter()
        if self._origin is None:
            self._origin = now
        self._mark = now

    def _require_running(self):
        if self._origin is None:
            raise TimerError("timer is not running")

    def since_start(self):
        self._require_running()
        self._mark = perf_counter()
        return self._mark - self._origin

    def since_last_check(self):
        self._require_running()
        now = perf_counter()
        elapsed, self._mark = now - self._mark, now
        return elapsed

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, exc_type, exc, tb):
        print(self.print_tmpl.format(self.since_last_check()))
        self._origin = None
