from dsgn import _C

nms = _C.nms
