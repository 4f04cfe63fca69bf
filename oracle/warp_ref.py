"""ORACLE (test infrastructure only; never imported by the product path) -- the landmark-Delaunay warp post-process of
1024_warp_morphs.py:78-113,163-210, with the OpenCV calls it makes restated from OpenCV's published sources (4.x):

    cv2.boundingRect(np.float32([tri]))                          -> imgproc/src/shapedescr.cpp  pointSetBoundingRect (CV_32F points)
    cv2.getAffineTransform(np.float32(src), np.float32(dst))     -> imgproc/src/imgwarp.cpp     getAffineTransform + core LU (hal::LU64f, LUImpl)
    cv2.warpAffine(patch, M, size, INTER_LINEAR, REFLECT_101)    -> imgwarp.cpp                 warpAffine, WarpAffineInvoker, remapBilinear<float>,
                                                                                               initInterTab1D/2D, core borderInterpolate
    cv2.fillConvexPoly(mask_f32, np.int32(tri), 1, 16, 0)        -> imgproc/src/drawing.cpp     fillConvexPoly (LINE_AA on a non-8U image falls back to
                                                                                               line_type 8), FillConvexPoly, Line, LineIterator

written as LITERAL loop transcriptions (one scanline / one Bresenham step / one LU pivot at a time), so that the product's closed-form,
vectorised host set-up and its device kernel (drivers.warp_plan, csrc/warp.hip) are checked against an independent statement.

PARITY UNPINNED: OpenCV itself is absent offline (no cv2, no fixture of it in the reference), so no output of the real library pins this
file; what is restated is the documented / published algorithm -- the 1/32-pixel fixed-point coordinate grid (INTER_BITS = 5, AB_BITS = 10,
round_delta = 16), the 32 x 32 bilinear weight table in float, BORDER_REFLECT_101 at the PATCH, the scanline + Bresenham polygon fill --
and the KATs of tests/test_oracle_golden.py are hand-computed from those rules.  One thing the sources leave open: hal::LU64f may dispatch to
LAPACK in a given build; the last bits of the 2x3 matrix then differ, which moves a coordinate only when it sits within ~1e-13 of a rounding
boundary of the 1/1024 grid.
"""
import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS


def cv_floor(v):
    return int(np.floor(v))


def cv_round(v):
    """cvRound(double) = lrint: round half to even."""
    return int(np.rint(v))


def bounding_rect_f32(pts):
    """pointSetBoundingRect for CV_32F points: Rect(floor(min x), floor(min y), floor(max x) - floor(min x) + 1, ...)."""
    p = np.asarray(pts, np.float32)
    xmin, ymin = cv_floor(p[:, 0].min()), cv_floor(p[:, 1].min())
    xmax, ymax = cv_floor(p[:, 0].max()), cv_floor(p[:, 1].max())
    return xmin, ymin, xmax - xmin + 1, ymax - ymin + 1


def lu_solve(a, b):
    """LUImpl (core/src/matrix_decomp.cpp): Gaussian elimination with partial pivoting, in place, double."""
    a = np.array(a, np.float64)
    b = np.array(b, np.float64)
    m = a.shape[0]
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(a[j, i]) > abs(a[k, i]):
                k = j
        if abs(a[k, i]) < np.finfo(np.float64).eps * 100:
            return None                                  # LUImpl returns 0: "singular"
        if k != i:
            for j in range(i, m):
                a[i, j], a[k, j] = a[k, j], a[i, j]
            b[i], b[k] = b[k], b[i]
        d = -1.0 / a[i, i]
        for j in range(i + 1, m):
            alpha = a[j, i] * d
            for kk in range(i + 1, m):
                a[j, kk] += alpha * a[i, kk]
            b[j] += alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, m):
            s -= a[i, k] * b[k]
        b[i] = s / a[i, i]
    return b


def get_affine_transform(src, dst):
    """getAffineTransform: the 6 x 6 system [x y 1 0 0 0; 0 0 0 x y 1] X = [u; v] from float32 points, solved in double -> M [2,3]."""
    s, d = np.asarray(src, np.float32), np.asarray(dst, np.float32)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (float(s[i, 0]), float(s[i, 1]), 1.0)
        a[2 * i + 1, 3:6] = (float(s[i, 0]), float(s[i, 1]), 1.0)
        b[2 * i], b[2 * i + 1] = float(d[i, 0]), float(d[i, 1])
    x = lu_solve(a, b)
    if x is None:                                        # cv::solve: `if( !result ) dst = Scalar(0);` -- and getAffineTransform ignores the result
        x = np.zeros(6)
    return x.reshape(2, 3)


def invert_affine(M):
    """warpAffine without WARP_INVERSE_MAP inverts the matrix first (imgwarp.cpp, cv::warpAffine)."""
    M = np.array(M, np.float64).reshape(6)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def border_reflect_101(p, n):
    """borderInterpolate(p, len, BORDER_REFLECT_101)."""
    if 0 <= p < n:
        return p
    if n == 1:
        return 0
    while True:
        p = -p if p < 0 else n - 1 - (p - n) - 1
        if 0 <= p < n:
            return p


def bilinear_tab():
    """initInterTab1D / initInterTab2D for INTER_LINEAR in float: tab[fy][fx] = [(1-fy)(1-fx), (1-fy) fx, fy (1-fx), fy fx], f = i / 32.f."""
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    t1 = [(np.float32(1.0) - np.float32(i) * scale, np.float32(i) * scale) for i in range(INTER_TAB_SIZE)]
    tab = np.zeros((INTER_TAB_SIZE, INTER_TAB_SIZE, 4), np.float32)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            for k1 in range(2):
                for k2 in range(2):
                    tab[i, j, k1 * 2 + k2] = t1[i][k1] * t1[j][k2]
    return tab


_TAB = bilinear_tab()


def warp_affine_linear_reflect101(src, M, size):
    """cv2.warpAffine(src [h,w,c] float32, M [2,3], (width, height), flags=INTER_LINEAR, borderMode=BORDER_REFLECT_101) -> [height,width,c] float32:
    WarpAffineInvoker's fixed-point coordinates + remapBilinear<float>."""
    src = np.asarray(src, np.float32)
    sh, sw, cn = src.shape
    dw, dh = size
    iM = invert_affine(M)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    adelta = [cv_round(iM[0] * x * AB_SCALE) for x in range(dw)]
    bdelta = [cv_round(iM[3] * x * AB_SCALE) for x in range(dw)]
    dst = np.zeros((dh, dw, cn), np.float32)
    for y in range(dh):
        X0 = cv_round((iM[1] * y + iM[2]) * AB_SCALE) + round_delta
        Y0 = cv_round((iM[4] * y + iM[5]) * AB_SCALE) + round_delta
        for x in range(dw):
            X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS)
            Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS)
            sx = int(np.clip(X >> INTER_BITS, -32768, 32767))
            sy = int(np.clip(Y >> INTER_BITS, -32768, 32767))
            w = _TAB[Y & (INTER_TAB_SIZE - 1), X & (INTER_TAB_SIZE - 1)]
            if 0 <= sx < sw - 1 and 0 <= sy < sh - 1:
                v0, v1, v2, v3 = src[sy, sx], src[sy, sx + 1], src[sy + 1, sx], src[sy + 1, sx + 1]
            else:
                sx0, sx1 = border_reflect_101(sx, sw), border_reflect_101(sx + 1, sw)
                sy0, sy1 = border_reflect_101(sy, sh), border_reflect_101(sy + 1, sh)
                v0, v1, v2, v3 = src[sy0, sx0], src[sy0, sx1], src[sy1, sx0], src[sy1, sx1]
            dst[y, x] = ((v0 * w[0] + v1 * w[1]) + v2 * w[2]) + v3 * w[3]           # float32, left to right
    return dst


def line8(pt1, pt2):
    """Line(img, pt1, pt2, color, 8) -> the pixels LineIterator(connectivity 8, left_to_right=true) visits (end points inside the image)."""
    x1, y1 = pt1
    x2, y2 = pt2
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                                  # left_to_right: start from the left end point
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    ystep = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    pts = []
    x, y = x1, y1
    for _ in range(dx + 1):
        pts.append((x, y))
        mask = err < 0
        err += minus_delta + (plus_delta if mask else 0)
        if steep:                               # major axis y (its sign is ystep), minor axis x (always +1: the walk goes left to right)
            y += ystep
            if mask:
                x += 1
        else:
            x += 1
            if mask:
                y += ystep
    return pts


def fill_convex_poly(height, width, pts):
    """cv2.fillConvexPoly(mask [height,width] non-8U, np.int32 points, color, lineType=16 -> 8, shift=0) -> bool mask of the written pixels:
    FillConvexPoly with line_type 8 = the Bresenham outline (Line) + the scanline fill with edges tracked in 16.16 fixed point."""
    v = [(int(p[0]), int(p[1])) for p in pts]
    npts = len(v)
    mask = np.zeros((height, width), bool)
    delta1 = delta2 = XY_ONE >> 1
    p0 = v[npts - 1]
    xmin = xmax = v[0][0]
    ymin = ymax = v[0][1]
    imin = 0
    for i in range(npts):
        p = v[i]
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax, xmax, xmin = max(ymax, p[1]), max(xmax, p[0]), min(xmin, p[0])
        for (x, y) in line8(p0, p):
            mask[y, x] = True
        p0 = p
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= width or ymin >= height:
        return mask
    ymax = min(ymax, height - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=npts - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = npts
    y = ymin
    while True:
        for e in edge:
            if y >= e["ye"]:
                idx0, di = e["idx"], e["di"]
                idx = idx0 + di
                if idx >= npts:
                    idx -= npts
                while True:
                    go = edges > 0
                    edges -= 1
                    if not go:
                        break
                    ty = v[idx][1]
                    if ty > y:
                        xs, xe = v[idx0][0] << XY_SHIFT, v[idx][0] << XY_SHIFT
                        e["ye"] = ty
                        num, den = (xe - xs) * 2 + (ty - y), 2 * (ty - y)
                        e["dx"] = abs(num) // den * (1 if num >= 0 else -1)              # C integer division truncates toward zero
                        e["x"], e["idx"] = xs, idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= npts:
                        idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = (edge[left]["x"] + delta1) >> XY_SHIFT
            xx2 = (edge[right]["x"] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < width:
                xx1, xx2 = max(xx1, 0), min(xx2, width - 1)
                mask[y, xx1:xx2 + 1] = True
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break
    return mask


def morph_triangle(img_G, img_avg, t_G, t_avg):
    """morphTriangle (1024_warp_morphs.py:90-113), in place on img_avg [H,W,3] float32."""
    r1 = bounding_rect_f32(t_G)
    r = bounding_rect_f32(t_avg)
    t_rect = [(t_avg[i][0] - r[0], t_avg[i][1] - r[1]) for i in range(3)]
    t1_rect = [(t_G[i][0] - r1[0], t_G[i][1] - r1[1]) for i in range(3)]
    mask = fill_convex_poly(r[3], r[2], np.int32(t_rect))
    img1 = img_G[r1[1]:r1[1] + r1[3], r1[0]:r1[0] + r1[2]]
    warp_mat = get_affine_transform(np.float32(t1_rect), np.float32(t_rect))
    patch = warp_affine_linear_reflect101(img1, warp_mat, (r[2], r[3]))
    m = mask[:, :, None].astype(np.float32)
    sl = (slice(r[1], r[1] + r[3]), slice(r[0], r[0] + r[2]))
    img_avg[sl] = img_avg[sl] * (np.float32(1) - m) + patch * m


def warp_morph_ref(img_hwc_f32, points_G, points_avg, simplices):
    """The triangle loop of 1024_warp_morphs.py:186-203 -> imgMorph [H,W,3] float32 (np.uint8(imgMorph) is what the script writes)."""
    img = np.asarray(img_hwc_f32, np.float32)
    out = np.zeros_like(img)
    for tri in simplices:
        x, y, z = (int(t) for t in tri)
        morph_triangle(img, out, [points_G[x], points_G[y], points_G[z]], [points_avg[x], points_avg[y], points_avg[z]])
    return out


def warp_affine_linear_reflect101_rows(src, M, size):
    """The same arithmetic as warp_affine_linear_reflect101 with the per-pixel loop carried by numpy (whole destination patch at once):
    used for the 1024^2 comparisons; tests/test_oracle_golden.py checks it bit for bit against the literal loop."""
    src = np.asarray(src, np.float32)
    sh, sw, cn = src.shape
    dw, dh = size
    iM = invert_affine(M)
    xs, ys = np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64)
    adelta = np.rint(iM[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(iM[3] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((iM[1] * ys + iM[2]) * AB_SCALE).astype(np.int64) + AB_SCALE // INTER_TAB_SIZE // 2
    Y0 = np.rint((iM[4] * ys + iM[5]) * AB_SCALE).astype(np.int64) + AB_SCALE // INTER_TAB_SIZE // 2
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy = np.clip(X >> INTER_BITS, -32768, 32767), np.clip(Y >> INTER_BITS, -32768, 32767)
    w = _TAB[Y & (INTER_TAB_SIZE - 1), X & (INTER_TAB_SIZE - 1)]                       # [dh, dw, 4]

    def refl(p, n):
        if n == 1:
            return np.zeros_like(p)
        p = p.copy()
        for _ in range(64):
            bad = (p < 0) | (p >= n)
            if not bad.any():
                break
            p = np.where(p < 0, -p, np.where(p >= n, 2 * n - 2 - p, p))
        return p

    x0, x1, y0, y1 = refl(sx, sw), refl(sx + 1, sw), refl(sy, sh), refl(sy + 1, sh)   # (inliers are their own reflection)
    v0, v1, v2, v3 = src[y0, x0], src[y0, x1], src[y1, x0], src[y1, x1]
    return ((v0 * w[..., 0:1] + v1 * w[..., 1:2]) + v2 * w[..., 2:3]) + v3 * w[..., 3:4]


def warp_morph_ref_rows(img_hwc_f32, points_G, points_avg, simplices):
    """warp_morph_ref with the vectorised patch warp (the rasteriser stays the literal one)."""
    img = np.asarray(img_hwc_f32, np.float32)
    out = np.zeros_like(img)
    for tri in simplices:
        t_G = [points_G[int(t)] for t in tri]
        t_avg = [points_avg[int(t)] for t in tri]
        r1, r = bounding_rect_f32(t_G), bounding_rect_f32(t_avg)
        t_rect = [(t_avg[i][0] - r[0], t_avg[i][1] - r[1]) for i in range(3)]
        t1_rect = [(t_G[i][0] - r1[0], t_G[i][1] - r1[1]) for i in range(3)]
        mask = fill_convex_poly(r[3], r[2], np.int32(t_rect))
        img1 = img[r1[1]:r1[1] + r1[3], r1[0]:r1[0] + r1[2]]
        patch = warp_affine_linear_reflect101_rows(img1, get_affine_transform(np.float32(t1_rect), np.float32(t_rect)), (r[2], r[3]))
        m = mask[:, :, None].astype(np.float32)
        sl = (slice(r[1], r[1] + r[3]), slice(r[0], r[0] + r[2]))
        out[sl] = out[sl] * (np.float32(1) - m) + patch * m
    return out
