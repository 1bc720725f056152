"""Independent geometry for the tests: area of the intersection of two rotated rectangles by
Sutherland-Hodgman clipping of their corner polygons (float64).  The corners are built by the CALLER with
the formula of the code under comparison (e.g. the clockwise rotation of kitti_utils/rotate_iou.py or of
mmdet3d's iou3d kernel), so a wrong angle convention in the implementation shows up as a mismatch."""
import numpy as np


def corners_clockwise(cx, cy, dx, dy, angle):
    """rotate_iou.py:205-227 / box_np_ops.rotation_2d / iou3d_kernel.cu:111-118:
    x' = cos a * x + sin a * y,  y' = -sin a * x + cos a * y  (clockwise by `angle`)."""
    c, s = np.cos(angle), np.sin(angle)
    local = np.array([[-dx / 2, -dy / 2], [-dx / 2, dy / 2], [dx / 2, dy / 2], [dx / 2, -dy / 2]])
    return np.stack([c * local[:, 0] + s * local[:, 1] + cx, -s * local[:, 0] + c * local[:, 1] + cy], 1)


def _area(poly):
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _ccw(poly):
    x, y = poly[:, 0], poly[:, 1]
    return poly if (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) > 0 else poly[::-1]


def intersection_area(pa, pb):
    subject, clip = _ccw(np.asarray(pa, float)), _ccw(np.asarray(pb, float))
    out = subject
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        inside = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0
        nxt = []
        for j in range(len(out)):
            p, q = out[j], out[(j + 1) % len(out)]
            if inside(p) != inside(q):
                d1, d2 = q - p, b - a
                t = ((a[0] - p[0]) * d2[1] - (a[1] - p[1]) * d2[0]) / (d1[0] * d2[1] - d1[1] * d2[0])
                x = p + t * d1
                if inside(p):
                    nxt += [p, x]
                else:
                    nxt += [x]
            elif inside(p):
                nxt.append(p)
        out = np.array(nxt) if nxt else np.zeros((0, 2))
        if len(out) < 3:
            return 0.0
    return _area(out)
