"""train._StreamSets decision logic without a GPU (the calibration and the library call are stubbed): the probe of both sets,
the drift move, and -- ADVICE r5 -- the re-baseline when a lasting workload shift makes BOTH sets look high."""
import collections

from convdr_amd import train as TR


class _Sets(TR._StreamSets):
    def __init__(self):
        self.device, self.scores = None, None
        self.sets = [(1, 2, 3), (4, 5, 6)]
        self.decisions = collections.deque(maxlen=64)
        self.active, self.enabled, self.phase, self.skip, self.high = 0, True, "probe0", 2, 0
        self.pending, self.samples, self.median = collections.deque(), {}, {}
        self.moves, self.nsteps = collections.deque(maxlen=4), 0
        self.applied = 0

    def _apply(self):
        self.applied += 1


def _settle(ss, c0=1.0, c1=0.9):
    for _ in range(ss.PROBE):
        ss._feed(0, c0)
    for _ in range(ss.PROBE):
        ss._feed(1, c1)
    assert ss.phase == "steady"


def test_probe_keeps_the_cheaper_set_and_drift_moves():
    ss = _Sets()
    _settle(ss)
    assert ss.active == 1 and "keeping set 1" in ss.decisions[-1]
    for c in (0.90, 1.02, 0.90, 1.02, 1.03):
        ss._feed(1, c)
    assert ss.active == 1                      # two high steps in a row are not a drift
    ss._feed(1, 1.02)
    assert ss.active == 0 and "moving to set 0" in ss.decisions[-1]


def test_a_lasting_workload_shift_rebaselines_instead_of_hopping_forever():
    ss = _Sets()
    _settle(ss)
    k, moves = 1, 0
    for i in range(200):                       # every step now costs 15 % more per token on EITHER set (longer sequences)
        before = ss.active
        ss._feed(ss.active, 1.035 if ss.active == 1 else 1.15)
        moves += int(ss.active != before)
    assert moves == 1, list(ss.decisions)      # one move, then "both sets high" -> re-baseline, and quiet from there on
    assert any("re-baselining" in d for d in ss.decisions)
    assert ss.median and max(ss.median.values()) < 1.2
    assert len(ss.decisions) <= 64 and len(ss.samples.get(ss.active, [])) <= 64
    # ... and a REAL single-set degradation afterwards is still caught
    good = ss.active
    for _ in range(3):
        ss._feed(good, 1.15 * 1.12)
    assert ss.active != good
