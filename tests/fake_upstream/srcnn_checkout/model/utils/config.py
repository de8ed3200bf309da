import types

import numpy as np

cfg = types.SimpleNamespace(
    RNG_SEED=3, PIXEL_MEANS=np.array([[[102.9801, 115.9465, 122.7717]]]), MAX_NUM_GT_BOXES=30, KPTS_GRID=28,
    TRAIN=types.SimpleNamespace(USE_FLIPPED=True, SCALES=(600,), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2), BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0),
                                DIM_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.3, 0.3), DIM_NORMALIZE_MEANS=(1.5, 1.6, 3.9, 0.0, 0.0)),
    TEST=types.SimpleNamespace(NMS=0.3))
