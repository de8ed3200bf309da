import torch


def bbox_transform_inv(boxes, deltas, batch_size):
    w = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    h = boxes[:, :, 3] - boxes[:, :, 1] + 1.0
    cx, cy = boxes[:, :, 0] + 0.5 * w, boxes[:, :, 1] + 0.5 * h
    out = deltas.clone()
    for k in range(deltas.shape[2] // 4):
        dx, dy, dw, dh = deltas[:, :, 4 * k], deltas[:, :, 4 * k + 1], deltas[:, :, 4 * k + 2], deltas[:, :, 4 * k + 3]
        pcx, pcy, pw, ph = dx * w + cx, dy * h + cy, torch.exp(dw) * w, torch.exp(dh) * h
        out[:, :, 4 * k], out[:, :, 4 * k + 1] = pcx - 0.5 * pw, pcy - 0.5 * ph
        out[:, :, 4 * k + 2], out[:, :, 4 * k + 3] = pcx + 0.5 * pw, pcy + 0.5 * ph
    return out


def kpts_transform_inv(boxes, delta, grid):
    w = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    pos = (delta[:, :, 0] % grid).float()
    return (boxes[:, :, 0] + (pos + 0.5) / grid * w).unsqueeze(2), (delta // grid).float()


def border_transform_inv(boxes, delta, grid):
    w = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    return (boxes[:, :, 0] + (delta[:, :, 0].float() + 0.5) / grid * w).unsqueeze(2)


def clip_boxes(boxes, im_info, batch_size):
    boxes[:, :, 0::4].clamp_(0, float(im_info[0, 1]) - 1)
    boxes[:, :, 1::4].clamp_(0, float(im_info[0, 0]) - 1)
    boxes[:, :, 2::4].clamp_(0, float(im_info[0, 1]) - 1)
    boxes[:, :, 3::4].clamp_(0, float(im_info[0, 0]) - 1)
    return boxes
