#!/usr/bin/env python
"""tests/golden/jet256.npy: the 256 BGR byte triples Visualization::DrawPoint paints with, computed from the colour tables of
the reference (esvo_core/src/tools/Visualization.cpp:128-226: CV_RGB(255.0f * r[i], 255.0f * g[i], 255.0f * b[i]) stored
into an 8-bit BGR image, i.e. saturate_cast<uchar> = round half to even of the float product).  Runs only where
/root/reference exists; the table itself is not copied into this repository, only the bytes it produces."""
import os
import re
import sys

import numpy as np

REF = os.environ.get("ESVO_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "esvo_core/src/tools/Visualization.cpp")).read()


def table(name):
    m = re.search(r"const float Visualization::%s\[\] = \{(.*?)\};" % name, src, re.S)
    return np.array([float(t) for t in m.group(1).replace("\n", " ").split(",")], np.float32)


r, g, b = table("r"), table("g"), table("b")
assert len(r) == len(g) == len(b) == 256
to_u8 = lambda t: np.clip(np.rint((np.float32(255.0) * t).astype(np.float64)), 0, 255).astype(np.uint8)
bgr = np.stack([to_u8(b), to_u8(g), to_u8(r)], axis=1)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jet256.npy"), bgr)
# the closed form used by the oracle and the device: byte = rint(clamp(min(4 i + a, -4 i + b), 0, 255)) (jet on i / 255)
i = np.arange(256, dtype=np.float64)
form = lambda a, b: np.rint(np.clip(np.minimum(4 * i + a, -4 * i + b), 0, 255)).astype(np.uint8)
ok = (np.array_equal(form(-382.5, 1147.5), bgr[:, 2]) and np.array_equal(form(-127.5, 892.5), bgr[:, 1])
      and np.array_equal(form(127.5, 637.5), bgr[:, 0]))
print("closed form == bytes of the reference's tables:", ok)
assert ok
