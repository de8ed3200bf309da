import numpy as np

F, BL = 721.5377, 0.54


def solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, box_left, box_right, kpts):
    disp = max(float(box_left[0] - box_right[0]), 1.0)
    z = F * BL / disp
    x = ((box_left[0] + box_left[2]) / 2 - 609.5593) * z / F
    return 1, np.array([x, 1.6, z, float(alpha)])


def solve_x_y_theta_from_kpt(im_shape, calib, alpha, dim, box_left, disparity, kpts):
    z = F * BL / float(disparity)
    x = ((box_left[0] + box_left[2]) / 2 - 609.5593) * z / F
    return np.array([x, 1.6, float(alpha)]), z
