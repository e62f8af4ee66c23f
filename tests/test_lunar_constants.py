"""Constants of the LunarLander solver that are derived, not copied: re-derive them on the CPU."""
import os
import re

import numpy as np

HPP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gymrl_amd", "csrc", "env_lunar_device.hpp")


def _const(name):
    text = open(HPP).read()
    m = re.search(r"constexpr float %s = ([0-9a-fxp.\-+]+)f" % name, text)
    assert m, name
    return np.float32(float.fromhex(m.group(1)) if m.group(1).startswith("0x") else float(m.group(1)))


def test_linear_slop_square_threshold():
    """b2RevoluteJoint::SolvePositionConstraints returns positionError <= b2_linearSlop with positionError = C.Length()
    (the oracle's lunar_oracle.c keeps the sqrtf); the kernel compares the squared length with the largest float whose
    correctly rounded square root is still <= the slop.  numpy's float32 sqrt is correctly rounded."""
    slop, thr = _const("kLinearSlop"), _const("kLinearSlopSqMax")
    assert np.sqrt(thr, dtype=np.float32) <= slop
    assert np.sqrt(np.nextafter(thr, np.float32(1), dtype=np.float32), dtype=np.float32) > slop
    # monotonic on both sides of the threshold over a window of neighbours
    x = thr
    for _ in range(2000):
        x = np.nextafter(x, np.float32(0), dtype=np.float32)
        assert np.sqrt(x, dtype=np.float32) <= slop
    x = thr
    for _ in range(2000):
        x = np.nextafter(x, np.float32(1), dtype=np.float32)
        assert np.sqrt(x, dtype=np.float32) > slop
