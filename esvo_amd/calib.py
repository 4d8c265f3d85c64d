"""Stereo calibration products without OpenCV.

The reference computes, once at start-up and with OpenCV, the raw->rectified look-up table
(cv::undistortPoints), the rectifying remap tables (cv::initUndistortRectifyMap) and the
validity mask (remap of an all-ones image + threshold):
  esvo_core/src/container/CameraSystem.cpp:36-111, esvo_time_surface/src/TimeSurface.cpp:313-401.
The C-ABI takes those arrays as inputs (a ROS build passes its own CameraSystem's), so this
numpy restatement is host-side set-up for the ROS-free harness: both the oracle and the GPU
receive the same arrays, hence last-bit deviations from real OpenCV cannot affect parity.
"""
import os

import numpy as np
import yaml

from .abi import Calib


def _poly_fisheye(theta, D):
    t2 = theta * theta
    return theta * (1 + D[0] * t2 + D[1] * t2**2 + D[2] * t2**3 + D[3] * t2**4)


def rect_to_raw(u, v, K, D, R, P, model):
    """Continuous rectified pixel -> raw (distorted) pixel: the per-pixel formula of
    cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    R = np.asarray(R, np.float64).reshape(3, 3)
    P = np.asarray(P, np.float64).reshape(3, 4)
    iR = np.linalg.inv(P[:, :3] @ R)
    u = np.asarray(u, np.float64)
    v = np.asarray(v, np.float64)
    X = iR[0, 0] * u + iR[0, 1] * v + iR[0, 2]
    Y = iR[1, 0] * u + iR[1, 1] * v + iR[1, 2]
    Wc = iR[2, 0] * u + iR[2, 1] * v + iR[2, 2]
    x, y = X / Wc, Y / Wc
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if model == "plumb_bob":
        k1, k2, p1, p2 = D[:4]
        r2 = x * x + y * y
        kr = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return fx * xd + cx, fy * yd + cy
    if model == "equidistant":
        r = np.sqrt(x * x + y * y)
        theta = np.arctan(r)
        theta_d = _poly_fisheye(theta, D)
        scale = np.where(r == 0, 1.0, theta_d / np.where(r == 0, 1.0, r))
        return fx * x * scale + cx, fy * y * scale + cy
    raise ValueError(f"unsupported distortion model {model!r}")


def init_undistort_rectify_map(K, D, R, P, width, height, model):
    """cv::initUndistortRectifyMap(..., CV_32FC1): map1 (x) and map2 (y), float32."""
    vv, uu = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    mx, my = rect_to_raw(uu, vv, K, D, R, P, model)
    return mx.astype(np.float32), my.astype(np.float32)


def undistort_points(K, D, R, P, width, height, model):
    """cv::undistortPoints / cv::fisheye::undistortPoints on every raw pixel centre
    (float32 in, float32 out; the 4th column of P is ignored): the raw->rectified LUT."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    R = np.asarray(R, np.float64).reshape(3, 3)
    P = np.asarray(P, np.float64).reshape(3, 4)
    vv, uu = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (uu.astype(np.float64) - cx) / fx
    y0 = (vv.astype(np.float64) - cy) / fy
    if model == "plumb_bob":
        k1, k2, p1, p2 = D[:4]
        x, y = x0.copy(), y0.copy()
        for _ in range(5):  # TermCriteria(COUNT, 5, 0.01)
            r2 = x * x + y * y
            icdist = 1.0 / (1 + k1 * r2 + k2 * r2 * r2)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
        ok = np.ones_like(x, dtype=bool)
    elif model == "equidistant":
        theta_d = np.sqrt(x0 * x0 + y0 * y0)
        theta_d = np.clip(theta_d, -np.pi / 2, np.pi / 2)
        theta = theta_d.copy()
        active = np.abs(theta_d) > 1e-8
        done = ~active
        for _ in range(10):  # TermCriteria(COUNT + EPS, 10, 1e-8)
            t2 = theta * theta
            t4, t6, t8 = t2 * t2, t2 * t2 * t2, t2 * t2 * t2 * t2
            fix = (theta * (1 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8) - theta_d) / (
                1 + 3 * D[0] * t2 + 5 * D[1] * t4 + 7 * D[2] * t6 + 9 * D[3] * t8)
            theta = np.where(done, theta, theta - fix)
            done = done | (np.abs(fix) < 1e-8)
        safe_td = np.where(active, theta_d, 1.0)
        scale = np.where(active, np.tan(theta) / safe_td, 1.0)
        flipped = ((theta_d < 0) & (theta > 0)) | ((theta_d > 0) & (theta < 0))
        ok = done & ~flipped
        x, y = x0 * scale, y0 * scale
    else:
        raise ValueError(f"unsupported distortion model {model!r}")
    X = R[0, 0] * x + R[0, 1] * y + R[0, 2]
    Y = R[1, 0] * x + R[1, 1] * y + R[1, 2]
    Wc = R[2, 0] * x + R[2, 1] * y + R[2, 2]
    xr, yr = X / Wc, Y / Wc
    up = P[0, 0] * xr + P[0, 2]
    vp = P[1, 1] * yr + P[1, 2]
    up = np.where(ok, up, -1000000.0)
    vp = np.where(ok, vp, -1000000.0)
    return np.stack([up, vp], axis=-1).astype(np.float32)


def rectify_mask(map_x, map_y, model):
    """CameraSystem.cpp:67-72 / :87-92: remap (INTER_LINEAR, float source, BORDER_CONSTANT 0) of
    an all-ones image, threshold(>0.999 plumb_bob / >0.1 equidistant) -> 0/255."""
    h, w = map_x.shape
    sx = np.rint(map_x * np.float32(32)).astype(np.int64)
    sy = np.rint(map_y * np.float32(32)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    ax = ((sx & 31).astype(np.float32)) / np.float32(32)
    ay = ((sy & 31).astype(np.float32)) / np.float32(32)
    one = np.float32(1)
    w00, w01 = (one - ax) * (one - ay), ax * (one - ay)
    w10, w11 = (one - ax) * ay, ax * ay

    def inside(xx, yy):
        return ((xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)).astype(np.float32)

    val = inside(ix, iy) * w00 + inside(ix + 1, iy) * w01 + inside(ix, iy + 1) * w10 + inside(ix + 1, iy + 1) * w11
    thr = np.float32(0.999) if model == "plumb_bob" else np.float32(0.1)
    return np.where(val > thr, 255, 0).astype(np.uint8)


def make_camera(width, height, K, D, R, P, model):
    map_x, map_y = init_undistort_rectify_map(K, D, R, P, width, height, model)
    lut = undistort_points(K, D, R, P, width, height, model)
    mask = rectify_mask(map_x, map_y, model)
    return Calib(width, height, P, lut, mask, map_x, map_y)


class StereoRig:
    """CameraSystem (CameraSystem.cpp:152-212): left/right Calib + raw intrinsics."""

    def __init__(self, left, right, intr_left=None, intr_right=None, name="rig"):
        self.left, self.right, self.name = left, right, name
        self.intr_left, self.intr_right = intr_left, intr_right
        Pr = right.P.reshape(3, 4)
        self.baseline = float(np.linalg.norm(np.linalg.inv(Pr[:, :3]) @ Pr[:, 3]))  # computeBaseline
        self.width, self.height = left.width, left.height

    @property
    def focal(self):
        P = self.left.P.reshape(3, 4)
        return 0.5 * (P[0, 0] + P[1, 1])


def load_yaml_camera(path):
    with open(path) as f:
        y = yaml.safe_load(f)
    intr = dict(
        width=int(y["image_width"]), height=int(y["image_height"]),
        K=np.array(y["camera_matrix"]["data"], np.float64).reshape(3, 3),
        D=np.array(y["distortion_coefficients"]["data"], np.float64),
        R=np.array(y["rectification_matrix"]["data"], np.float64).reshape(3, 3),
        P=np.array(y["projection_matrix"]["data"], np.float64).reshape(3, 4),
        model=str(y["distortion_model"]),
    )
    return intr


def load_rig(calib_dir, name=None):
    """CameraSystem::loadCalibInfo (CameraSystem.cpp:168-212): <dir>/left.yaml + right.yaml."""
    il = load_yaml_camera(os.path.join(calib_dir, "left.yaml"))
    ir = load_yaml_camera(os.path.join(calib_dir, "right.yaml"))
    return rig_from_intrinsics(il, ir, name or os.path.basename(calib_dir.rstrip("/")))


def rig_from_intrinsics(il, ir, name="rig"):
    left = make_camera(il["width"], il["height"], il["K"], il["D"], il["R"], il["P"], il["model"])
    right = make_camera(ir["width"], ir["height"], ir["K"], ir["D"], ir["R"], ir["P"], ir["model"])
    return StereoRig(left, right, il, ir, name)


def ideal_rig(width, height, focal, baseline, cx=None, cy=None, name="ideal"):
    """Rectified pin-hole rig with D=0, R=I (identity LUT / maps, all-valid mask)."""
    cx = (width - 1) / 2.0 if cx is None else cx
    cy = (height - 1) / 2.0 if cy is None else cy
    K = np.array([[focal, 0, cx], [0, focal, cy], [0, 0, 1]], np.float64)
    Pl = np.hstack([K, np.zeros((3, 1))])
    Pr = Pl.copy()
    Pr[0, 3] = -focal * baseline
    il = dict(width=width, height=height, K=K, D=np.zeros(4), R=np.eye(3), P=Pl, model="plumb_bob")
    ir = dict(width=width, height=height, K=K, D=np.zeros(4), R=np.eye(3), P=Pr, model="plumb_bob")
    return rig_from_intrinsics(il, ir, name)


# Calibrations of the datasets the reference ships (esvo_core/calib/*), embedded as data so
# the GPU box (which has no /root/reference) can build the same rigs.
_DATASETS = {
    "upenn": dict(
        width=346, height=260, model="equidistant",
        K_l=[226.38018519795807, 0.0, 173.6470807871759, 0.0, 226.15002947047415, 133.73271487507847, 0, 0, 1],
        D_l=[-0.048031442223833355, 0.011330957517194437, -0.055378166304281135, 0.021500973881459395],
        R_l=[0.999877311526236, 0.015019439766575743, -0.004447282784398257,
             -0.014996983873604017, 0.9998748347535599, 0.005040367172759556,
             0.004522429630305261, -0.004973052949604937, 0.9999774079320989],
        P_l=[199.6530123165822, 0.0, 177.43276376280926, 0.0, 0.0, 199.6530123165822, 126.81215684365904, 0.0,
             0.0, 0.0, 1.0, 0.0],
        K_r=[226.0181418548734, 0, 174.5433576736815, 0, 225.7869434267677, 124.21627572590607, 0, 0, 1],
        D_r=[-0.04846669832871334, 0.010092844338123635, -0.04293073765014637, 0.005194706897326005],
        R_r=[0.9999922706537476, 0.003931701344419404, -1.890238450965101e-05,
             -0.003931746704476347, 0.9999797362744968, -0.005006836150689904,
             -7.83382948021244e-07, 0.0050068717705076754, 0.9999874655386736],
        P_r=[199.6530123165822, 0.0, 177.43276376280926, -19.941771812941038, 0.0, 199.6530123165822,
             126.81215684365904, 0.0, 0.0, 0.0, 1.0, 0.0],
    ),
    "rpg": dict(  # DAVIS240C pair, esvo_core/calib/rpg/{left,right}.yaml
        width=240, height=180, model="plumb_bob",
        K_l=[196.639, 0, 105.064, 0, 196.733, 72.4717, 0.0, 0.0, 1.0],
        D_l=[-0.336733, 0.111789, -0.00140053, -0.000459594],
        R_l=[0.999791, -0.018779, -0.00802416, 0.0187767, 0.999824, -0.000360707, 0.00802952, 0.000209964, 0.999968],
        P_l=[156.925, 0, 108.167, 0, 0, 156.925, 78.4205, 0, 0, 0, 1, 0],
        K_r=[196.426, 0, 110.745, 0, 196.564, 88.1131, 0.0, 0.0, 1.0],
        D_r=[-0.346294, 0.12772, -0.000272051, -0.000195801],
        R_r=[0.999589, 0.0222217, -0.0181009, -0.0222166, 0.999753, 0.000486491, 0.0181073, -8.41512e-05, 0.999836],
        P_r=[156.925, 0, 108.167, -23.2327, 0, 156.925, 78.4205, 0, 0, 0, 1, 0],
    ),
    "hkust": dict(  # DAVIS346 pair, esvo_core/calib/hkust/{left,right}.yaml (the right camera_matrix's last row is
        # not [0 0 1] in the shipped file; OpenCV reads fx, fy, cx, cy only)
        width=346, height=260, model="plumb_bob",
        K_l=[263.796, 0, 176.994, 0, 263.738, 124.373, 0, 0, 1],
        D_l=[-0.386589, 0.157241, 0.000322143, 6.13759e-06],
        R_l=[0.999809, 0.0161928, 0.0109163, -0.0162088, 0.999868, 0.0013701, -0.0108927, -0.00154678, 0.999939],
        P_l=[189.705, 0, 165.382, 0, 0, 189.705, 121.295, 0, 0, 0, 1, 0],
        K_r=[263.485, 0, 162.942, 0, 263.276, 118.029, -0.0151344, 0.00133093, 0.999885],
        D_r=[-0.383425, 0.152823, -0.000257745, 0.000268432],
        R_r=[0.9993960957463914, 0.0034732142808621717, -0.03457427641222047,
             -0.0035085878889783376, 0.9999933816804096, -0.0009625000798637905,
             0.03457070461958685, 0.0010832257094615543, 0.9994016675011942],
        P_r=[189.705, 0, 165.382, -13.8634, 0, 189.705, 121.295, 0, 0, 0, 1, 0],
    ),
    "dsec": dict(
        width=640, height=480, model="plumb_bob",
        K_l=[553.469, 0, 346.653, 0, 553.399, 216.521, 0, 0, 1],
        D_l=[-0.0935648, 0.194458, 7.64243e-05, 0.00195639],
        R_l=[0.999866, -0.00319364, 0.0160517, 0.00322964, 0.999992, -0.00221712, -0.0160445, 0.00226867, 0.999869],
        P_l=[534.094, 0, 335.446, 0, 0, 534.094, 223.233, 0, 0, 0, 1, 0],
        K_r=[552.182, 0, 336.874, 0, 551.445, 226.326, 0, 0, 1],
        D_r=[-0.0949368, 0.202115, 0.000582129, 0.00145529],
        R_r=[0.999963, 0.00818053, -0.00267849, -0.0081745, 0.999964, 0.00225394, 0.00269683, -0.00223196, 0.999994],
        P_r=[534.094, 0, 335.446, -319.94, 0, 534.094, 223.233, 0, 0, 0, 1, 0],
    ),
}


def dataset_rig(name):
    """Real-distortion rig of a shipped dataset ('upenn' 346x260 equidistant, 'rpg' 240x180, 'hkust' 346x260 and
    'dsec' 640x480 plumb_bob), values from esvo_core/calib/<name>/{left,right}.yaml; 'hd' is SURVEY.md §8's synthetic
    1280x720 stress rig (ideal, f = 1000 px, baseline 0.3 m)."""
    if name == "hd":
        return ideal_rig(1280, 720, 1000.0, 0.3, name="hd")
    d = _DATASETS[name]
    mk = lambda s: dict(width=d["width"], height=d["height"], K=np.array(d["K_" + s], np.float64).reshape(3, 3),
                        D=np.array(d["D_" + s], np.float64), R=np.array(d["R_" + s], np.float64).reshape(3, 3),
                        P=np.array(d["P_" + s], np.float64).reshape(3, 4), model=d["model"])
    return rig_from_intrinsics(mk("l"), mk("r"), name)
