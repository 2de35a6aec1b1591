"""CPU: the crop oracle (oracle/crop_oracle.c) against hand-derived answers of panoramaCropper.m:73-165 and against a
literal pure-Python transcription of its scan loops on small images."""
import numpy as np

import oracle


def _img(mask, value=200):
    a = np.zeros(mask.shape + (3,), np.uint8)
    a[mask] = value
    return a


def _reference_scan(inside):
    """panoramaCropper.m:99-157, line by line, 1-based arrays, quirks included."""
    h, w = inside.shape
    height = [0] * (w + 2)
    left = [0] * (w + 2)
    right = [0] * (w + 2)
    maxarea = ll = rr = hh = nl = 0
    for line in range(1, h + 1):
        for k in range(1, w + 1):
            height[k] = height[k] + 1 if inside[line - 1, k - 1] else 0
        for k in range(1, w + 1):
            left[k] = k
            while left[k] > 1 and height[k] <= height[left[k] - 1]:
                left[k] = left[left[k] - 1]
        for k in range(w - 1, 0, -1):
            right[k] = k
            while right[k] < w - 1 and height[k] <= height[right[k] + 1]:
                right[k] = right[right[k] + 1]
        for k in range(1, w + 1):
            val = (right[k] - left[k] + 1) * height[k]
            if maxarea < val:
                maxarea, ll, rr, hh, nl = val, left[k], right[k], height[k], line
    return (ll, nl - hh + 1, rr - ll + 1, hh + 1), (ll, rr, hh, nl)


def test_solid_rectangle_and_the_last_column_quirk():
    m = np.zeros((40, 60), bool)
    m[5:25, 10:50] = True  # rows 6..25, columns 11..50 (1-based)
    rect, ok, dbg = oracle.crop_rect(_img(m))
    assert dbg == (11, 50, 20, 25) and rect == (11, 6, 40, 21) and ok
    # content up to the last column: `right` never reaches column w, the rectangle stops one short
    m2 = np.zeros((30, 50), bool)
    m2[2:22, 5:50] = True
    rect2, ok2, dbg2 = oracle.crop_rect(_img(m2))
    assert dbg2[:2] == (6, 49) and rect2[2] == 44
    # content over the whole image: rows offsety..offsety+cropH overshoot -> the reference returns its input
    full = np.ones((20, 30), bool)
    rect3, ok3, _ = oracle.crop_rect(_img(full))
    assert not ok3


def test_holes_are_filled_but_bays_are_not():
    m = np.zeros((50, 70), bool)
    m[5:45, 5:65] = True
    m[20:30, 20:40] = False  # enclosed hole -> filled: the crop spans it
    inside = oracle.crop_inside(_img(m))
    assert inside[25, 30] and not inside[0, 0]
    rect, ok, dbg = oracle.crop_rect(_img(m))
    assert dbg == (6, 65, 40, 45) and ok
    m[20:30, 5:40] = False  # now the hole opens to the left border region: a bay, not a hole
    inside = oracle.crop_inside(_img(m))
    assert not inside[25, 30]
    rect_b, _, dbg_b = oracle.crop_rect(_img(m))
    assert dbg_b != dbg
    # a diagonal-only connection does not open a hole (4-connected background)
    d = np.ones((9, 9), bool)
    d[0, :] = d[-1, :] = d[:, 0] = d[:, -1] = False
    d[4, 4] = False
    d[1, 1] = False  # touches the border ring's corner only diagonally... (1,1) IS 4-adjacent to (0,1): opens nothing inside
    inside = oracle.crop_inside(_img(d))
    assert inside[4, 4] and not inside[1, 1]


def test_threshold_and_white_canvas():
    a = np.full((20, 20, 3), 255, np.uint8)
    a[4:16, 3:17] = 120
    rect, ok, dbg = oracle.crop_rect(a, canvas_white=True, rng=250)
    assert dbg == (4, 17, 12, 16)
    b = np.zeros((20, 20, 3), np.uint8)
    b[4:16, 3:17] = 9  # gray 9 > blackRange 8, not > 9
    assert oracle.crop_rect(b, False, 8)[2] == (4, 17, 12, 16)
    assert oracle.crop_rect(b, False, 9)[2] == (0, 0, 0, 0) and not oracle.crop_rect(b, False, 9)[1]
    # rgb2gray weights: pure blue 255 -> round(29.07) = 29
    c = np.zeros((6, 6, 3), np.uint8)
    c[1:5, 1:5, 2] = 255
    assert oracle.crop_rect(c, False, 28)[2][2] > 0 and oracle.crop_rect(c, False, 29)[2] == (0, 0, 0, 0)


def test_scan_equals_the_literal_loops_on_random_shapes():
    rng = np.random.default_rng(5)
    for trial in range(40):
        h, w = int(rng.integers(1, 28)), int(rng.integers(1, 34))
        m = rng.random((h, w)) < rng.uniform(0.3, 0.95)
        if trial % 3 == 0:  # blobs with ties: many equal heights
            m = np.zeros((h, w), bool)
            for _ in range(3):
                r0, c0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                m[r0:r0 + int(rng.integers(1, h + 1)), c0:c0 + int(rng.integers(1, w + 1))] = True
        img = _img(m)
        inside = oracle.crop_inside(img)
        want_rect, want_dbg = _reference_scan(inside)
        rect, ok, dbg = oracle.crop_rect(img)
        assert dbg == want_dbg and rect == want_rect, (trial, h, w)
        ox, oy, cw, ch = rect
        assert ok == (ox >= 1 and oy >= 1 and oy + ch <= h and ox + cw <= w)
