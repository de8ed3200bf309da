def get_dimensions(corners):
    """corners [3,8] centred -> (h, w, l, ry)"""
    ext = corners.max(dim=1)[0] - corners.min(dim=1)[0]
    return float(ext[1]), float(ext[0]), float(ext[2]), -1.57
