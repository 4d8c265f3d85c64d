"""A SECOND, independent restatement of the OpenCV arithmetic on the hot path (test infrastructure).

OpenCV is absent from the image and from /root/reference, so `cv::remap`, `cv::GaussianBlur`, `cv::medianBlur` and
`cv::Mat::convertTo` cannot be pinned to the real library.  The CPU oracle (oracle/esvo_oracle.cpp) restates them in closed form;
this module restates them again, from the STRUCTURE of OpenCV's own implementation (imgproc/src/imgwarp.cpp, smooth.dispatch.cpp,
median_blur.simd.hpp as published) rather than from the closed forms -- table-driven fixed-point interpolation, two-pass 8.8
fixed-point filtering, sorting networks replaced by a plain sort -- in numpy, sharing no code with the oracle.  Two sources that
agree bit for bit on random and calibrated inputs (tests/test_third_party_pin.py) is what can be had without the library; the same
is done for Eigen's LevenbergMarquardt in oracle/ref_shim (tests/test_ref_pin.py) and scipy's MINPACK.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS            # 32
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS


def cv_round(v):
    """cvRound: round half to even (SSE cvtss2si / lrint under the default rounding mode)"""
    return np.rint(v)


def bilinear_tab_i():
    """imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt = true): for every (fy, fx) of the 32 x 32 sub-pixel grid the 2 x 2 weights
    saturate_cast<short>(vy[k1] * vx[k2] * 32768) with the 1-D linear taps (1 - x, x), x = i / 32 computed in float.
    The only entry whose float product does not fit a short is (0, 0): 1 * 1 * 32768 -> 32767."""
    tab = np.zeros((INTER_TAB_SIZE, INTER_TAB_SIZE, 2, 2), np.int32)
    t1 = np.zeros((INTER_TAB_SIZE, 2), np.float32)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * np.float32(1.0 / INTER_TAB_SIZE)
        t1[i] = (np.float32(1.0) - x, x)
    for i in range(INTER_TAB_SIZE):          # fy
        for j in range(INTER_TAB_SIZE):      # fx
            for k1 in range(2):
                for k2 in range(2):
                    v = np.float32(t1[i, k1] * t1[j, k2]) * np.float32(INTER_REMAP_COEF_SCALE)
                    tab[i, j, k1, k2] = int(min(max(cv_round(float(v)), -32768), 32767))   # saturate_cast<short>
    return tab


_TAB = None


def cv_remap_linear_u8(src, map_x, map_y, border_value=0):
    """cv::remap(src 8UC1, map1 32FC1, map2 32FC1, INTER_LINEAR, BORDER_CONSTANT): the float maps are quantised to 1 / 32 pixel
    (sx = cvRound(mapx * 32), float multiply), the integer part addresses the 2 x 2 block, the fraction the weight table;
    remapBilinear's three cases: block fully inside / fully outside (constant) / straddling (taps outside read the constant);
    FixedPtCast<int, uchar, 15>: (sum + 2^14) >> 15, saturated."""
    global _TAB
    if _TAB is None:
        _TAB = bilinear_tab_i()
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape
    sx = cv_round(np.asarray(map_x, np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    sy = cv_round(np.asarray(map_y, np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    ix, iy = sx >> INTER_BITS, sy >> INTER_BITS
    w = _TAB[sy & (INTER_TAB_SIZE - 1), sx & (INTER_TAB_SIZE - 1)].astype(np.int64)   # [..., k1 (y), k2 (x)]
    out = np.zeros(ix.shape, np.int64)
    acc = np.zeros(ix.shape, np.int64)
    outside = (ix >= W) | (ix + 1 < 0) | (iy >= H) | (iy + 1 < 0)
    for k1 in range(2):
        for k2 in range(2):
            x, y = ix + k2, iy + k1
            ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
            v = np.where(ok, src[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)].astype(np.int64), border_value)
            acc += v * w[..., k1, k2]
    out = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS
    out = np.where(outside, border_value, out)
    return np.clip(out, 0, 255).astype(np.uint8)


def cv_gaussian5_u8(src):
    """cv::GaussianBlur(8UC1, Size(5, 5), sigma = 0): sigma <= 0 and ksize <= 7 select the fixed small_gaussian_tab row
    {0.0625, 0.25, 0.375, 0.25, 0.0625}; the 8-bit path filters in 8.8 fixed point (ufixedpoint16: the taps are 16, 64, 96, 64,
    16), rows then columns, and rounds once at the end: (v + 2^15) >> 16.  BORDER_REFLECT_101 (the default)."""
    src = np.ascontiguousarray(src, np.uint8)
    k = np.array([16, 64, 96, 64, 16], np.int64)   # 8.8
    H, W = src.shape
    if min(H, W) < 3:
        raise ValueError("image too small for reflect-101 padding of 2")
    p = np.pad(src.astype(np.int64), 2, mode="reflect")
    rows = sum(k[i] * p[:, i:i + W] for i in range(5))             # (H + 4) x W, 8.8
    cols = sum(k[i] * rows[i:i + H, :] for i in range(5))          # H x W, 16.16
    return np.clip((cols + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def cv_median3_u8(src):
    """cv::medianBlur(8UC1, 3): the median of the 3 x 3 neighbourhood, BORDER_REPLICATE"""
    p = np.pad(np.ascontiguousarray(src, np.uint8), 1, mode="edge")
    H, W = src.shape
    stack = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 0)
    return np.sort(stack, axis=0)[4]


def cv_convert_to_u8(img_f64):
    """cv::Mat::convertTo(CV_8U) of a CV_64F image: saturate_cast<uchar>(cvRound(v))"""
    return np.clip(cv_round(np.asarray(img_f64, np.float64)), 0, 255).astype(np.uint8)
