"""ros::Time arithmetic used by the mapper's host glue (roscpp time.h semantics).

The reference compares stamps through `toSec()` doubles (utils.h:43-71) and builds the virtual
view stamps with `ros::Time(t.toSec() + 0.05 * BM_half_slice_thickness)`
(esvo_Mapping.cpp:585-599); reproducing those roundings keeps the event->pose association
identical.
"""
import math

import numpy as np

NS = 1_000_000_000


def ns_to_sec(ns):
    ns = int(ns)
    return float(ns // NS) + 1e-9 * float(ns % NS)


def ros_time_from_sec(t):
    """ros::TimeBase::fromSec"""
    sec = int(math.floor(t))
    frac = (t - sec) * 1e9
    nsec = int(math.floor(frac + 0.5))  # boost::math::round: half away from zero (frac >= 0)
    sec += nsec // NS
    nsec %= NS
    return sec * NS + nsec


def pose_stamps(t_ns, half_slice):
    """st_map_ stamps: from t_begin = t_end - 10*half_slice in steps of 0.05*half_slice while
    t.toSec() <= t_end.toSec()  (esvo_Mapping.cpp:563,585-599)."""
    t_end = ns_to_sec(t_ns)
    t_tmp = ros_time_from_sec(max(0.0, t_end - 10 * half_slice))
    out = []
    while ns_to_sec(t_tmp) <= t_end:
        out.append(t_tmp)
        t_tmp = ros_time_from_sec(ns_to_sec(t_tmp) + 0.05 * half_slice)
    return np.array(out, dtype=np.uint64)


def pose_table(pose_fn, t_ns, half_slice):
    st = pose_stamps(t_ns, half_slice)
    T = np.stack([np.asarray(pose_fn(int(s)), np.float64).reshape(16) for s in st])
    return st, np.ascontiguousarray(T)
