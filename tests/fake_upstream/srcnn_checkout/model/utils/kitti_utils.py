import os
import types

import numpy as np

F, BL = 721.5377, 0.54


def read_obj_calibration(path):
    p2 = np.array([[F, 0, 609.5593, 44.85728], [0, F, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    p3 = p2.copy()
    p3[0, 3] = p2[0, 3] - F * BL
    return types.SimpleNamespace(p2=p2, p3=p3, path=path)


def infer_boundary(im_shape, boxes):
    return np.stack([boxes[:, 0], boxes[:, 2]], 1).astype(np.float32)


def write_detection_results(result_dir, image_number, calib, box, xyz, dim, theta, score):
    with open(os.path.join(result_dir, image_number + ".txt"), "a") as f:
        f.write("Car -1 -1 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n"
                % (0.0, box[0], box[1], box[2], box[3], dim[0], dim[1], dim[2], xyz[0], xyz[1], xyz[2], theta, float(score)))
