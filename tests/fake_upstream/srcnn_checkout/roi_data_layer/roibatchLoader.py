import numpy as np
import torch
import torch.utils.data

H, W = 600, 1987
MEANS = np.array([102.9801, 115.9465, 122.7717], np.float32)


class roibatchLoader(torch.utils.data.Dataset):
    def __init__(self, roidb, ratio_list, ratio_index, batch_size, num_classes, training=True, normalize=None):
        self._roidb, self.training = roidb, training

    def __len__(self):
        return len(self._roidb)

    def __getitem__(self, index):
        rs = np.random.RandomState(100 + index)
        base = rs.randint(0, 256, (H // 8 + 1, W // 8 + 1, 3)).astype(np.float32)
        img = np.repeat(np.repeat(base, 8, 0), 8, 1)[:H, :W] + rs.randint(-5, 6, (H, W, 3)).astype(np.float32) * 0.37   # not 8-bit levels
        left = torch.from_numpy(np.ascontiguousarray((img - MEANS).transpose(2, 0, 1)))
        right = torch.roll(left, -38, 2).contiguous()
        im_info = torch.tensor([float(H), float(W), 1.6])
        if not self.training:
            g = torch.FloatTensor([1, 1, 1, 1, 1])
            return left, right, im_info, g, g, g, g, g, 0
        boxes = torch.zeros(30, 5)
        boxes[0] = torch.tensor([300.0, 250.0, 500.0, 400.0, 1.0])
        return left, right, im_info, boxes, boxes.clone(), boxes.clone(), torch.zeros(30, 5), torch.zeros(30, 6), 1
