"""Ordering logic of api.FramePipeline on the CPU, with stand-ins for the device scenes (the GPU tests check the frames
themselves: tests/test_gpu_parity.py::test_frame_pipeline_*).  What must hold: frame k does not start before frame k-1
has raised its tail flag; every scene renders its frames in order; stats come back per frame; a failing render releases
whoever waits for it and the error reaches the caller instead of a hang."""
import ctypes as C
import threading
import time

import pytest

from rust_raytracer_amd import api


class FakeLib:
    def __init__(self):
        self.flag_addr = {}

    def rt_scene_set_tail_flag(self, handle, addr):
        self.flag_addr[handle] = addr
        return api.RT_OK


class FakeScene:
    """render_device: 'steady' phase, raise the tail flag, 'tail' phase; like librt_mi355.so the flag is also raised on errors."""

    def __init__(self, lib, handle, log, lock, steady=0.03, tail=0.03, fail_on=None):
        self._lib, self._h, self.log, self.lock = lib, handle, log, lock
        self.steady, self.tail, self.fail_on = steady, tail, fail_on
        self.last = None

    def _raise_flag(self):
        addr = self._lib.flag_addr.get(self._h)
        if addr:
            C.c_int32.from_address(addr).value = 1

    def render_device(self, camera, params, out_ptr, stream):
        frame = params  # the tests pass the frame number as "params"
        with self.lock:
            self.log.append(("start", frame, self._h, time.perf_counter()))
        try:
            time.sleep(self.steady)
            if self.fail_on == frame:
                raise api.RtError(api.RT_E_DEVICE, "injected failure")
            with self.lock:
                self.log.append(("tail", frame, self._h, time.perf_counter()))
            self._raise_flag()
            time.sleep(self.tail)
        finally:
            self._raise_flag()
            with self.lock:
                self.log.append(("end", frame, self._h, time.perf_counter()))
        self.last = frame

    def stats(self):
        return ("stats", self.last)

    def close(self):
        pass


def run(depth, n_frames, fail_on=None):
    lib, log, lock = FakeLib(), [], threading.Lock()
    scenes = [FakeScene(lib, 100 + i, log, lock, fail_on=fail_on) for i in range(depth)]
    pipe = api.FramePipeline(None, 0, depth, scenes=scenes)
    t0 = time.perf_counter()
    stats = pipe.render_frames(None, list(range(n_frames)), [0] * n_frames, list(range(depth)))
    return stats, log, time.perf_counter() - t0


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_frames_start_in_order_after_the_previous_tail(depth):
    n = 7
    stats, log, elapsed = run(depth, n)
    assert stats == [("stats", k) for k in range(n)]
    t = {(what, frame): when for what, frame, _, when in log}
    scene_of = {frame: h for what, frame, h, _ in log if what == "start"}
    for k in range(n):
        assert scene_of[k] == 100 + k % depth                      # round robin over the device scenes
        if k > 0:
            assert t[("start", k)] >= t[("tail", k - 1)]           # never before the previous frame's tail
        if k >= depth:
            assert t[("start", k)] >= t[("end", k - depth)]        # a scene renders one frame at a time
    if depth > 1:
        # tails overlap the next frame: the batch is shorter than the frames one after the other would have been (their own
        # measured durations, so that a loaded machine's longer sleeps do not matter)
        one_after_the_other = sum(t[("end", k)] - t[("start", k)] for k in range(n))
        assert elapsed < 0.95 * one_after_the_other
        assert any(t[("start", k)] < t[("end", k - 1)] for k in range(1, n))


def test_a_failing_frame_releases_the_waiters_and_raises():
    t0 = time.perf_counter()
    with pytest.raises(api.RtError, match="injected failure"):
        run(2, 6, fail_on=2)
    assert time.perf_counter() - t0 < 5.0      # no thread is left waiting for a flag that never comes


def test_argument_checks():
    lib, log, lock = FakeLib(), [], threading.Lock()
    pipe = api.FramePipeline(None, 0, 2, scenes=[FakeScene(lib, 1, log, lock), FakeScene(lib, 2, log, lock)])
    with pytest.raises(ValueError):
        pipe.render_frames(None, [0, 1], [0], [0, 1])          # one output buffer per frame
    with pytest.raises(ValueError):
        pipe.render_frames(None, [0, 1], [0, 0], [0])          # one stream per device scene
    assert pipe.render_frames(None, [], [], [0, 1]) == []
