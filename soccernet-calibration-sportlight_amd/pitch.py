"""57-point 3-D pitch template and point sets (host constants of the solve).

PITCH_POINTS / INTERSECTON_TO_PITCH_POINTS mirror /root/reference/src/datatools/ellipse.py:16-157 on top of
SoccerPitch.point_dict (/root/reference/baseline/soccerpitch.py:109-263); the point sets mirror
/root/reference/src/models/hrnet/prediction.py:15-41.  World frame: metres, origin at the centre mark,
x along the pitch length (left goal at x = -52.5), +y towards the main-camera ("bottom") touch line,
z = -height.  The device copy of the same table lives in csrc/solve.hip (build_pitch).
"""
import math

import numpy as np

PITCH_LENGTH, PITCH_WIDTH = 105.0, 68.0
PENALTY_AREA_LENGTH, PENALTY_AREA_WIDTH = 16.5, 40.32
GOAL_AREA_LENGTH, GOAL_AREA_WIDTH = 5.5, 18.32
GOAL_LINE_TO_PENALTY_MARK = 11.0
CENTER_CIRCLE_RADIUS = 9.15
GOAL_LENGTH, GOAL_HEIGHT = 7.32, 2.44

INTERSECTON_TO_PITCH_POINTS = {
    0: 'L_GOAL_TL_POST', 1: 'L_GOAL_TR_POST', 2: 'L_GOAL_BL_POST', 3: 'L_GOAL_BR_POST',
    4: 'L_GOAL_AREA_BR_CORNER', 5: 'L_GOAL_AREA_TR_CORNER', 6: 'L_GOAL_AREA_BL_CORNER', 7: 'L_GOAL_AREA_TL_CORNER',
    8: 'L_PENALTY_AREA_BR_CORNER', 9: 'L_PENALTY_AREA_TR_CORNER', 10: 'L_PENALTY_AREA_BL_CORNER',
    11: 'L_PENALTY_AREA_TL_CORNER', 12: 'BL_PITCH_CORNER', 13: 'TL_PITCH_CORNER',
    14: 'B_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 15: 'T_TOUCH_AND_HALFWAY_LINES_INTERSECTION',
    16: 'R_PENALTY_AREA_BL_CORNER', 17: 'R_PENALTY_AREA_TL_CORNER', 18: 'R_PENALTY_AREA_BR_CORNER',
    19: 'R_PENALTY_AREA_TR_CORNER', 20: 'R_GOAL_AREA_BL_CORNER', 21: 'R_GOAL_AREA_TL_CORNER',
    22: 'R_GOAL_AREA_BR_CORNER', 23: 'R_GOAL_AREA_TR_CORNER', 24: 'R_GOAL_TL_POST', 25: 'R_GOAL_TR_POST',
    26: 'R_GOAL_BL_POST', 27: 'R_GOAL_BR_POST', 28: 'BR_PITCH_CORNER', 29: 'TR_PITCH_CORNER',
    30: 'CENTER_CIRCLE_TANGENT_TR', 31: 'CENTER_CIRCLE_TANGENT_TL', 32: 'CENTER_CIRCLE_TANGENT_BR',
    33: 'CENTER_CIRCLE_TANGENT_BL', 34: 'CENTER_CIRCLE_TR', 35: 'CENTER_CIRCLE_TL', 36: 'CENTER_CIRCLE_BR',
    37: 'CENTER_CIRCLE_BL', 38: 'CENTER_CIRCLE_R', 39: 'CENTER_CIRCLE_L',
    40: 'T_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION', 41: 'B_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION',
    42: 'CENTER_MARK', 43: 'LEFT_CIRCLE_R', 44: 'BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION',
    45: 'TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 46: 'LEFT_CIRCLE_TANGENT_T', 47: 'LEFT_CIRCLE_TANGENT_B',
    48: 'L_PENALTY_MARK', 49: 'L_MIDDLE_PENALTY', 50: 'RIGHT_CIRCLE_L',
    51: 'BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 52: 'TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION',
    53: 'RIGHT_CIRCLE_TANGENT_T', 54: 'RIGHT_CIRCLE_TANGENT_B', 55: 'R_PENALTY_MARK', 56: 'R_MIDDLE_PENALTY',
}
PITCH_POINTS_TO_INTERSECTON = {v: k for k, v in INTERSECTON_TO_PITCH_POINTS.items()}


def _tangents(cx, cy, r, px, py):
    hyp = math.sqrt((px - cx) ** 2 + (py - cy) ** 2)
    th = math.acos(r / hyp)
    d = math.atan2(py - cy, px - cx)
    return ((cx + r * math.cos(d + th), cy + r * math.sin(d + th)), (cx + r * math.cos(d - th), cy + r * math.sin(d - th)))


def _template():
    hl, hw, R = PITCH_LENGTH / 2, PITCH_WIDTH / 2, CENTER_CIRCLE_RADIUS
    gy, gh = GOAL_LENGTH / 2, GOAL_HEIGHT
    xs = {'goal': hl, 'ga': hl - GOAL_AREA_LENGTH, 'pa': hl - PENALTY_AREA_LENGTH, 'pm': hl - GOAL_LINE_TO_PENALTY_MARK}
    pts = {}

    def put(i, x, y, z=0.0):
        pts[i] = (float(x), float(y), float(z))
    put(0, -hl, gy, -gh); put(1, -hl, -gy, -gh); put(2, -hl, gy); put(3, -hl, -gy)
    put(4, -xs['ga'], GOAL_AREA_WIDTH / 2); put(5, -xs['ga'], -GOAL_AREA_WIDTH / 2)
    put(6, -hl, GOAL_AREA_WIDTH / 2); put(7, -hl, -GOAL_AREA_WIDTH / 2)
    put(8, -xs['pa'], PENALTY_AREA_WIDTH / 2); put(9, -xs['pa'], -PENALTY_AREA_WIDTH / 2)
    put(10, -hl, PENALTY_AREA_WIDTH / 2); put(11, -hl, -PENALTY_AREA_WIDTH / 2)
    put(12, -hl, hw); put(13, -hl, -hw); put(14, 0, hw); put(15, 0, -hw)
    put(16, xs['pa'], PENALTY_AREA_WIDTH / 2); put(17, xs['pa'], -PENALTY_AREA_WIDTH / 2)
    put(18, hl, PENALTY_AREA_WIDTH / 2); put(19, hl, -PENALTY_AREA_WIDTH / 2)
    put(20, xs['ga'], GOAL_AREA_WIDTH / 2); put(21, xs['ga'], -GOAL_AREA_WIDTH / 2)
    put(22, hl, GOAL_AREA_WIDTH / 2); put(23, hl, -GOAL_AREA_WIDTH / 2)
    put(24, hl, -gy, -gh); put(25, hl, gy, -gh); put(26, hl, -gy); put(27, hl, gy)
    put(28, hl, hw); put(29, hl, -hw)
    t_top = _tangents(0, 0, R, 0, -hw)
    t_bot = _tangents(0, 0, R, 0, hw)
    put(30, *t_top[0]); put(31, *t_top[1]); put(32, *t_bot[1]); put(33, *t_bot[0])
    s = math.sqrt(2.0) * R / 2
    put(34, s, -s); put(35, -s, -s); put(36, s, s); put(37, -s, s)
    put(38, R, 0); put(39, -R, 0); put(40, 0, -R); put(41, 0, R); put(42, 0, 0)
    dx = PENALTY_AREA_LENGTH - GOAL_LINE_TO_PENALTY_MARK
    ay = math.sqrt(R * R - dx * dx)
    put(43, -xs['pm'] + R, 0); put(44, -xs['pa'], ay); put(45, -xs['pa'], -ay)
    put(46, *_tangents(-xs['pm'], 0, R, pts[9][0], pts[9][1])[0])
    put(47, *_tangents(-xs['pm'], 0, R, pts[8][0], pts[8][1])[1])
    put(48, -xs['pm'], 0); put(49, pts[8][0], 0)
    put(50, xs['pm'] - R, 0); put(51, xs['pa'], ay); put(52, xs['pa'], -ay)
    put(53, *_tangents(xs['pm'], 0, R, pts[17][0], pts[17][1])[1])
    put(54, *_tangents(xs['pm'], 0, R, pts[16][0], pts[16][1])[0])
    put(55, xs['pm'], 0); put(56, pts[16][0], 0)
    return np.array([pts[i] for i in range(57)], dtype=np.float64)


PITCH_ARRAY = _template()                                  # (57,3), row = keypoint id
PITCH_POINTS = {INTERSECTON_TO_PITCH_POINTS[i]: PITCH_ARRAY[i].copy() for i in range(57)}

top_gates = [0, 1, 24, 25]
point_sets = {
    'groundplane': [i for i in range(58) if i not in top_gates],
    'goal_left': [0, 1, 2, 3, 6, 7, 10, 11, 12, 13],
    'goal_right': [18, 19, 22, 23, 24, 25, 26, 27, 28, 29],
}
IMG_SIZE = (960, 540)
keep_points = list(range(29)) + [40, 41, 42, 44, 45, 48, 51, 52, 55]


def get_pitch():
    return PITCH_POINTS
