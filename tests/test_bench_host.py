"""Host-side pieces of bench.py that need no GPU: the watchdog around the N > 1 diagnostic legs and the cgroup CPU quota."""
import threading
import time

import bench


def test_watchdog_fires_only_on_a_call_that_does_not_return():
    fired = threading.Event()
    assert bench.run_with_watchdog(lambda: 7, 0.05, fired.set) == 7
    time.sleep(0.15)
    assert not fired.is_set()                       # returned in time: never
    release = threading.Event()

    def stuck():                                    # stands for a collective that hangs; the real on_timeout ends the process
        release.wait(5.0)
        return "late"

    t0 = time.perf_counter()
    out = bench.run_with_watchdog(stuck, 0.1, lambda: (fired.set(), release.set()))
    assert fired.is_set() and out == "late" and time.perf_counter() - t0 < 2.0


def test_watchdog_lets_exceptions_through_and_disarms():
    fired = threading.Event()
    try:
        bench.run_with_watchdog(lambda: 1 / 0, 0.05, fired.set)
    except ZeroDivisionError:
        pass
    else:
        raise AssertionError("the exception was swallowed")
    time.sleep(0.15)
    assert not fired.is_set()


def test_cpu_quota_is_none_or_positive():
    q = bench.cpu_quota()
    assert q is None or q > 0
    cores, logical = bench.physical_cores()
    assert 1 <= cores <= logical
