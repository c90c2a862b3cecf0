"""Oracle: the 57-point 3-D pitch template, recomputed from the FIFA dimensions.

TEST INFRASTRUCTURE ONLY.  Follows
  * /root/reference/baseline/soccerpitch.py:6-13, 109-263 (39 dev-kit points; x along the pitch
    length with the left goal at x=-52.5, +y = "bottom" touch line, z = -height)
  * /root/reference/src/datatools/ellipse.py:16-92 (18 derived circle / tangent / mid-penalty points)
  * /root/reference/src/datatools/ellipse.py:99-157 (index -> name map)
  * /root/reference/src/models/hrnet/prediction.py:15-41 (point sets, keep_points, swap_z_y)
Pinned by tests/golden/pitch.npz (captured from the imported reference).
"""
import numpy as np

L, W = 105.0, 68.0
PEN_L, PEN_W = 16.5, 40.32
GA_L, GA_W = 5.5, 18.32
PEN_MARK = 11.0
R = 9.15
GOAL_W, GOAL_H = 7.32, 2.44


def _tangent_points(center, radius, point):
    """ellipse.py:20-33: the two tangent points on a circle seen from an outside point."""
    hyp = np.sqrt((point[0] - center[0]) ** 2 + (point[1] - center[1]) ** 2)
    th = np.arccos(radius / hyp)
    d = np.arctan2(point[1] - center[1], point[0] - center[0])
    return (np.array([center[0] + radius * np.cos(d + th), center[1] + radius * np.sin(d + th), 0.0]),
            np.array([center[0] + radius * np.cos(d - th), center[1] + radius * np.sin(d - th), 0.0]))


def pitch_points() -> np.ndarray:
    """(57,3) float64, row i = keypoint id i (ellipse.py:99-157)."""
    hl, hw = L / 2.0, W / 2.0
    P = np.zeros((57, 3))
    gy = GOAL_W / 2.0
    # goals, left: TL/TR posts are the crossbar ends (z=-2.44); "left"/"right" as seen from the pitch
    P[0] = (-hl, gy, -GOAL_H)      # L_GOAL_TL_POST
    P[1] = (-hl, -gy, -GOAL_H)     # L_GOAL_TR_POST
    P[2] = (-hl, gy, 0)            # L_GOAL_BL_POST
    P[3] = (-hl, -gy, 0)           # L_GOAL_BR_POST
    P[4] = (-hl + GA_L, GA_W / 2, 0)    # L_GOAL_AREA_BR_CORNER
    P[5] = (-hl + GA_L, -GA_W / 2, 0)   # L_GOAL_AREA_TR_CORNER
    P[6] = (-hl, GA_W / 2, 0)           # L_GOAL_AREA_BL_CORNER
    P[7] = (-hl, -GA_W / 2, 0)          # L_GOAL_AREA_TL_CORNER
    P[8] = (-hl + PEN_L, PEN_W / 2, 0)  # L_PENALTY_AREA_BR_CORNER
    P[9] = (-hl + PEN_L, -PEN_W / 2, 0)  # L_PENALTY_AREA_TR_CORNER
    P[10] = (-hl, PEN_W / 2, 0)         # L_PENALTY_AREA_BL_CORNER
    P[11] = (-hl, -PEN_W / 2, 0)        # L_PENALTY_AREA_TL_CORNER
    P[12] = (-hl, hw, 0)                # BL_PITCH_CORNER
    P[13] = (-hl, -hw, 0)               # TL_PITCH_CORNER
    P[14] = (0, hw, 0)                  # B_TOUCH_AND_HALFWAY
    P[15] = (0, -hw, 0)                 # T_TOUCH_AND_HALFWAY
    P[16] = (hl - PEN_L, PEN_W / 2, 0)  # R_PENALTY_AREA_BL_CORNER
    P[17] = (hl - PEN_L, -PEN_W / 2, 0)  # R_PENALTY_AREA_TL_CORNER
    P[18] = (hl, PEN_W / 2, 0)          # R_PENALTY_AREA_BR_CORNER
    P[19] = (hl, -PEN_W / 2, 0)         # R_PENALTY_AREA_TR_CORNER
    P[20] = (hl - GA_L, GA_W / 2, 0)    # R_GOAL_AREA_BL_CORNER
    P[21] = (hl - GA_L, -GA_W / 2, 0)   # R_GOAL_AREA_TL_CORNER
    P[22] = (hl, GA_W / 2, 0)           # R_GOAL_AREA_BR_CORNER
    P[23] = (hl, -GA_W / 2, 0)          # R_GOAL_AREA_TR_CORNER
    P[24] = (hl, -gy, -GOAL_H)          # R_GOAL_TL_POST
    P[25] = (hl, gy, -GOAL_H)           # R_GOAL_TR_POST
    P[26] = (hl, -gy, 0)                # R_GOAL_BL_POST
    P[27] = (hl, gy, 0)                 # R_GOAL_BR_POST
    P[28] = (hl, hw, 0)                 # BR_PITCH_CORNER
    P[29] = (hl, -hw, 0)                # TR_PITCH_CORNER
    top = _tangent_points((0.0, 0.0), R, P[15][:2])
    bot = _tangent_points((0.0, 0.0), R, P[14][:2])
    P[30], P[31] = top[0], top[1]       # CENTER_CIRCLE_TANGENT_TR / TL
    P[32], P[33] = bot[1], bot[0]       # CENTER_CIRCLE_TANGENT_BR / BL
    s = np.sqrt(2.0) * R / 2
    P[34] = (s, -s, 0)
    P[35] = (-s, -s, 0)
    P[36] = (s, s, 0)
    P[37] = (-s, s, 0)
    P[38] = (R, 0, 0)
    P[39] = (-R, 0, 0)
    P[40] = (0, -R, 0)                  # T_HALFWAY_LINE_AND_CENTER_CIRCLE
    P[41] = (0, R, 0)                   # B_HALFWAY_LINE_AND_CENTER_CIRCLE
    P[42] = (0, 0, 0)                   # CENTER_MARK
    lpm = np.array([-hl + PEN_MARK, 0.0, 0.0])
    rpm = np.array([hl - PEN_MARK, 0.0, 0.0])
    dx = PEN_L - PEN_MARK
    ay = np.sqrt(R * R - dx * dx)
    P[43] = lpm + P[38]                 # LEFT_CIRCLE_R
    P[44] = (-hl + PEN_L, ay, 0)        # BL_16M_LINE_AND_PENALTY_ARC
    P[45] = (-hl + PEN_L, -ay, 0)       # TL_16M_...
    P[46] = _tangent_points(lpm[:2], R, P[9][:2])[0]    # LEFT_CIRCLE_TANGENT_T
    P[47] = _tangent_points(lpm[:2], R, P[8][:2])[1]    # LEFT_CIRCLE_TANGENT_B
    P[48] = lpm
    P[49] = (P[8][0], 0.0, 0.0)         # L_MIDDLE_PENALTY
    P[50] = rpm + P[39]                 # RIGHT_CIRCLE_L
    P[51] = (hl - PEN_L, ay, 0)         # BR_16M_...
    P[52] = (hl - PEN_L, -ay, 0)        # TR_16M_...
    P[53] = _tangent_points(rpm[:2], R, P[17][:2])[1]   # RIGHT_CIRCLE_TANGENT_T
    P[54] = _tangent_points(rpm[:2], R, P[16][:2])[0]   # RIGHT_CIRCLE_TANGENT_B
    P[55] = rpm
    P[56] = (P[16][0], 0.0, 0.0)        # R_MIDDLE_PENALTY
    return P


TOP_GATES = [0, 1, 24, 25]                                   # prediction.py:15
GROUND = [i for i in range(57) if i not in TOP_GATES]        # prediction.py:19 (id 57 never exists, Q6)
GOAL_LEFT = [0, 1, 2, 3, 6, 7, 10, 11, 12, 13]               # prediction.py:20
GOAL_RIGHT = [18, 19, 22, 23, 24, 25, 26, 27, 28, 29]        # prediction.py:21
KEEP_POINTS = list(range(29)) + [40, 41, 42, 44, 45, 48, 51, 52, 55]   # prediction.py:25-26 (Q7: no 29)


def swap_z_y(p):
    """prediction.py:29-34: express a goal-plane point as a z=0 planar point (y, z, 0)."""
    return np.array([p[1], p[2], 0.0])
