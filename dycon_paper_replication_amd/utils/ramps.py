"""Host-side ramp schedules, same names/semantics as code/utils/ramps.py:19-41."""
import math


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t)^2), t = clip(current, 0, length)/length   (ramps.py:19-26)."""
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length)) / rampup_length
    return float(math.exp(-5.0 * (1.0 - t) * (1.0 - t)))


def linear_rampup(current, rampup_length):
    """ramps.py:29-35."""
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def cosine_rampdown(current, rampdown_length):
    """ramps.py:38-41."""
    assert 0 <= current <= rampdown_length
    return float(0.5 * (math.cos(math.pi * current / rampdown_length) + 1))
