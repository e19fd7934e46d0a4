"""Bookkeeping of the frame-staging path (fsgs_amd/staging.py) on CPU tensors: residency, least-recently-used eviction,
hit / miss / prefetch counters, the list protocol the step drivers rely on, and that Runner.global_run draws its frames in
the order of the reference's loop (train.py:378-384: one random.choice per iteration) although it now draws one iteration
ahead for the prefetch."""
import random

import numpy as np
import pytest
import torch

from fsgs_amd.staging import RecentWindow, StagedFrames, StagedLane


def _items(n, shape=(3, 4, 5)):
    return [torch.full(shape, float(i)) for i in range(n)]


def test_lane_is_list_like_and_returns_the_frames_it_was_given():
    lane = StagedLane(_items(6), "cpu", capacity=3)
    assert len(lane) == 6 and lane.shape == (3, 4, 5)
    assert [float(t[0, 0, 0]) for t in lane] == [0, 1, 2, 3, 4, 5]
    assert float(lane[-1].mean()) == 5.0
    assert lane.misses == 6 and lane.hits == 1  # the iteration missed everywhere (capacity 3 of 6); [-1] found frame 5 resident


def test_least_recently_used_frame_is_the_one_replaced():
    lane = StagedLane(_items(8), "cpu", capacity=3)
    a, b, c = lane[0], lane[1], lane[2]
    lane[0]                      # touch 0: frame 1 is now the oldest
    lane[3]                      # takes frame 1's buffer
    assert set(lane.cache) == {0, 2, 3} and float(b.mean()) == 3.0  # the buffer object was reused, as on the device
    assert float(a.mean()) == 0.0 and float(c.mean()) == 2.0
    assert (lane.hits, lane.misses) == (1, 4)


def test_prefetch_turns_the_following_lookup_into_a_hit():
    lane = StagedLane(_items(5), "cpu", capacity=3)  # (with 2, frame 0's miss after prefetch(1) would leave 1 the oldest)
    for t in range(5):
        lane.prefetch(t + 1)     # one past the end and the frames already resident are no-ops
        assert float(lane[t].mean()) == float(t)
    assert lane.misses == 1 and lane.hits == 4 and lane.prefetched == 4
    lane.prefetch(-1)
    lane.prefetch(None)
    assert lane.prefetched == 4


def test_a_protected_prefetch_survives_the_keyframes_pulled_through_the_lane_until_it_is_read():
    """progressive_run asks for frame t+1 before frame t's 50 tracking + 30 two-view mapping iterations; the
    random keyframes of those iterations must not push it out of a 4-buffer lane"""
    lane = StagedLane(_items(12), "cpu", capacity=4)
    lane[5]                                 # the current frame
    lane.prefetch(6)                        # the next frame, a whole cycle ahead
    keys = (0, 1, 2, 3, 0, 2, 1, 3, 0, 1)   # keyframes of the mapping iterations, each prefetched one iteration ahead:
    lane.prefetch(keys[0])                  # Runner.mapping asks for iteration i+1's BEFORE iteration i reads its own
    misses = lane.misses
    for i, k in enumerate(keys):
        if i + 1 < len(keys):
            lane.prefetch(keys[i + 1])
        lane[k]
        lane[5]
    assert lane.misses == misses            # not one keyframe was evicted between its prefetch and its read
    assert 6 in lane.cache and 6 in lane.protected
    misses = lane.misses
    assert float(lane[6].mean()) == 6.0 and lane.misses == misses  # a hit, and the protection ends with the first read
    assert 6 not in lane.protected
    for k in (7, 8, 9, 10):
        lane[k]
    assert 6 not in lane.cache              # ... after which it ages out like any other frame
    # never more than capacity - 1 protected frames: one buffer always stays evictable
    small = StagedLane(_items(6), "cpu", capacity=2)
    small.prefetch(0)
    small.prefetch(1)
    assert set(small.protected) == {0}
    small.prefetch(2, protect=False)
    assert set(small.cache) == {0, 2} and float(small[0].mean()) == 0.0


def test_the_lookups_of_a_progressive_run_all_find_their_frame_resident_in_four_buffers():
    """the order in which Runner.progressive_run / mapping ask for and read frames (trainer.py), replayed on CPU lanes with
    FOUR buffers each over 17 frames with a random keyframe per two-view iteration: apart from frame 0's colours nothing
    is ever missed, and no protection is left behind (a frame's mono-depth is only fetched for its own mapping)"""
    n, H, W = 17, 4, 6
    rng = np.random.default_rng(0)
    fr = StagedFrames([rng.random((3, H, W), dtype=np.float32) for _ in range(n)],
                      [rng.random((H, W), dtype=np.float32) for _ in range(n)],
                      flows_fw=[rng.random((2, H, W), dtype=np.float32) for _ in range(n - 1)],
                      K=np.eye(3, dtype=np.float32), device="cpu", capacity=4)
    r, keys, train = random.Random(0), [], set(int(i) for i in fr.i_train)
    for t in range(n):
        if t + 1 < n:
            fr.prefetch(t + 1, monodeps=False)       # progressive_run, top of the frame cycle
        if t in train:
            fr.prefetch(t, flows=False, colors=False)
        if t > 0:                                    # tracking(t)
            if t > 1:
                fr.flows_fw[t - 2]
            fr.flows_fw[t - 1]
            for _ in range(6):
                fr.colors[t]
        if t in train:                               # mapping(t)
            iters, views = (10, 1) if t == 0 else (12, 2)
            nxt = r.choice(keys) if views == 2 else None
            fr.prefetch(nxt, flows=False)
            for k in range(iters):
                ts = [nxt, t] if views == 2 else [t]
                if views == 2:
                    nxt = r.choice(keys) if k + 1 < iters else None
                    fr.prefetch(nxt, flows=False)
                for v in ts:
                    fr.colors[v]
                    fr.monodeps[v]
            keys.append(t)
    st = fr.stats()
    assert st["colors"]["misses"] == 1 and st["monodeps"]["misses"] == 0 and st["flows_fw"]["misses"] == 0, st
    assert not (fr.colors.protected or fr.monodeps.protected or fr.flows_fw.protected)


def test_absent_entries_and_mismatched_shapes():
    lane = StagedLane([None, torch.ones(2, 2), None], "cpu", capacity=4)
    assert lane[0] is None and lane[2] is None and float(lane[1].sum()) == 4.0
    lane.prefetch(0)
    assert lane.prefetched == 0
    with pytest.raises(ValueError, match="one shape"):
        StagedLane([torch.ones(2, 2), torch.ones(2, 3)], "cpu", capacity=2)
    with pytest.raises(ValueError, match="float32"):
        StagedLane([torch.ones(2, 2, dtype=torch.float64)], "cpu", capacity=2)
    assert len(StagedLane([], "cpu", capacity=2)) == 0


def test_recent_window_keeps_the_last_written_frames():
    w = RecentWindow(10, keep=3)
    assert len(w) == 10 and w[4] is None
    for t in range(6):
        w[t] = torch.tensor(float(t))
    assert [None if w[t] is None else float(w[t]) for t in range(6)] == [None, None, None, 3.0, 4.0, 5.0]
    w[4] = torch.tensor(40.0)    # rewriting refreshes the entry
    w[6] = torch.tensor(6.0)
    assert w[3] is None and float(w[4]) == 40.0 and float(w[-4]) == 6.0
    with pytest.raises(IndexError):
        w[10]


def test_staged_frames_prefetch_covers_what_tracking_a_frame_reads():
    n, H, W = 9, 4, 6
    rng = np.random.default_rng(0)
    cols = [rng.random((3, H, W), dtype=np.float32) for _ in range(n)]
    deps = [rng.random((H, W), dtype=np.float32) for _ in range(n)]
    fws = [rng.random((2, H, W), dtype=np.float32) for _ in range(n - 1)]
    fr = StagedFrames(cols, deps, flows_fw=fws, K=np.eye(3, dtype=np.float32), device="cpu", capacity=4)
    assert list(fr.i_test) == [4] and len(fr.pred_depths) == n and fr.pred_depths[0] is None
    fr.prefetch(3)
    assert set(fr.colors.cache) == {3} and set(fr.monodeps.cache) == {3} and set(fr.flows_fw.cache) == {1, 2}
    np.testing.assert_array_equal(fr.colors[3].numpy(), cols[3])
    np.testing.assert_array_equal(fr.flows_fw[2].numpy(), fws[2])
    np.testing.assert_array_equal(fr.flows_fw[1].numpy(), fws[1])
    np.testing.assert_array_equal(fr.monodeps[3].numpy(), deps[3])
    st = fr.stats()
    assert all(v["misses"] == 0 for v in st.values()) and st["flows_fw"] == {"hits": 2, "misses": 0, "prefetched": 2}
    fr.prefetch(0)               # frame 0 has no earlier flows
    fr.prefetch(n)               # past the end: colours / depths ignore it, the flows n-1 .. n-2 exist only partly
    assert set(fr.flows_fw.cache) == {1, 2, n - 2}


def test_global_run_draws_the_frames_in_the_reference_order_and_prefetches_one_iteration_ahead():
    from fsgs_amd.trainer import Runner

    class Frames:
        i_train = np.array([0, 1, 2, 3, 5, 6, 7])
        i_test = np.array([], dtype=int)
        colors = [None] * 8

        def __init__(self):
            self.events = []

        def prefetch(self, t, flows=True):
            assert not flows  # a mapping view reads colours and mono-depth only
            self.events.append(("prefetch", t))

    class Cloud:
        def initialize_optimizer(self): pass
        def oneupSHdegree(self): pass
        def update_learning_rate(self, it): pass

    run = Runner.__new__(Runner)
    run.rng, run.frames, run.pc = random.Random(7), Frames(), Cloud()
    run.mapping = lambda ts, views, progressive, want_pkg=True: run.frames.events.append(("map", ts))
    run.global_run(11, eval_every=0)
    want_rng = random.Random(7)
    want = [int(want_rng.choice(list(Frames.i_train))) for _ in range(12)]  # range(0, iterations + 1): 12 steps
    ev = run.frames.events
    assert [t for k, t in ev if k == "map"] == want
    assert [t for k, t in ev if k == "prefetch"] == want[1:]
    for i in range(11):  # the copy of step i + 1 is started before step i is enqueued
        assert ev[2 * i] == ("prefetch", want[i + 1]) and ev[2 * i + 1] == ("map", want[i])
    assert run.rng.random() == want_rng.random()  # not one draw more than the reference's loop


def test_progressive_mapping_draws_its_keyframes_in_the_reference_order_too():
    """train.py:239: one random.choice(keyframe_list) per two-view iteration -- now drawn one iteration early"""
    from fsgs_amd.trainer import Runner

    class Frames:
        colors = [None] * 8

        def __init__(self):
            self.prefetched = []

        def prefetch(self, t, flows=True):
            self.prefetched.append(t)

    class Opt:
        def zero_grad(self, set_to_none=True): pass

    class Cloud:
        optimizer = Opt()
        num_points = 0

    class Fast:
        last = None

        def __init__(self):
            self.seen = []

        def mapping_step(self, ts, step_optimizer=True, collect_stats=False):
            self.seen.append(list(ts))
            return 0.0

    run = Runner.__new__(Runner)
    run.rng, run.frames, run.pc, run.fast = random.Random(3), Frames(), Cloud(), Fast()
    run.keyframes, run.iteration, run.densify, run.trace, run.fused = [0, 1, 2, 3, 5], 0, False, None, True
    run.densify_interval, run.opacity_reset_interval, run.densify_until = 300, 3000, 15000
    run.fast.last = {"image": torch.zeros(1), "depth_sil": torch.zeros(2, 1)}
    run.mapping(6, 9, progressive=True)
    want_rng = random.Random(3)
    want = [want_rng.choice([0, 1, 2, 3, 5]) for _ in range(9)]
    # every keyframe is asked for before its iteration, the first one of the call included
    assert run.fast.seen == [[k, 6] for k in want] and run.frames.prefetched == want
    assert run.rng.random() == want_rng.random()
    run.mapping(0, 4, progressive=True)  # frame 0 is mapped alone: no draw at all
    assert run.fast.seen[-4:] == [[0]] * 4 and run.rng.random() == want_rng.random()


def test_a_prefetch_that_is_never_read_does_not_pin_its_buffer_forever():
    """protections end with the first lookup, with clear_protected() (Runner: end of a phase), or by age --
    PROTECT_LOADS loads after they were made -- so an abandoned look-ahead cannot make the frame in use the only
    eviction victim of a small lane"""
    import torch

    from fsgs_amd.staging import StagedLane

    host = [torch.full((2, 2), float(i)) for i in range(40)]
    lane = StagedLane(host, "cpu", 3)
    lane.prefetch(30)
    lane.prefetch(31)  # capacity - 1 protections at most
    assert set(lane.protected) == {30, 31}
    for rep in range(4):  # the frame in use keeps missing: it is the only victim
        assert float(lane[1][0, 0]) == 1.0 and float(lane[2][0, 0]) == 2.0
    assert lane.misses == 8 and 30 in lane.cache and 31 in lane.cache
    lane.clear_protected()
    assert not lane.protected
    m = lane.misses
    for rep in range(4):
        lane[1], lane[2]
    assert lane.misses - m <= 2  # both fit beside each other now
    # ageing: a fresh protection expires after PROTECT_LOADS further loads
    lane.prefetch(33)
    assert 33 in lane.protected
    for i in range(lane.PROTECT_LOADS + 4):
        lane[i % 20]
    assert 33 not in lane.protected and 33 not in lane.cache
