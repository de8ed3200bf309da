"""Host-side operators over torch ROCm tensors -> libadvengine.so, one module per kernel family:

  pixel    a1-a13: pixel-space maps, the PGD step, 8-bit import / export, disc mask, patch paste / update (csrc/advengine.hip)
  psv      K7: the plane-sweep cost volume (csrc/psv.hip)
  roi      RoIAlign forward / ordered backward, pyramid pooling, NMS (csrc/roi.hip)
  conv3d   3x3x3 convolutions on the float32 matrix cores: direct, strided, transposed, Winograd (csrc/conv3d.hip, csrc/wino2d.hip)
  conv2d   2D convolutions on the float32 matrix cores: 1x1, 3x3, Winograd, the route choice; fused bias / ReLU pass (csrc/conv2d.hip, csrc/wino2d.hip)
  wino4    3x3 / 3x3x3 stride-1 convolutions by Winograd F(4x4,3x3) (csrc/wino4.hip)
  align    dense photometric box alignment (csrc/align.hip)
  volume   after the 3D convolutions: depth regression, 5-D grid sampling, bilinear up-sampling, bird's-eye-view fold, focal loss (csrc/volume.hip, csrc/resize.hip)
  boxes    the proposal / target stage of a Stereo R-CNN step: IoU rows, regression targets, decode + clip, size partition, roi sampling,
           the RPN head's list packing, the attack objective's six-term chain (csrc/boxes.hip)
  elementwise  ReLU backward, the ResNet stem's bias + ReLU + max-pooling (csrc/volume.hip)

Each function is the counterpart of one inline block of the reference's attack scripts (cited per function) and keeps its argument
meaning; tensors stay where they are, kernels are enqueued on the caller's current torch stream.  There is no CPU or eager-torch fallback:
CPU tensors raise ``TypeError``, a missing ``libadvengine.so`` raises ``ImportError`` at import."""
from ._base import *       # noqa: F401,F403
from .elementwise import *       # noqa: F401,F403
from .pixel import *       # noqa: F401,F403
from .psv import *       # noqa: F401,F403
from .roi import *       # noqa: F401,F403
from .conv3d import *       # noqa: F401,F403
from .conv2d import *       # noqa: F401,F403
from .wino4 import *       # noqa: F401,F403
from .align import *       # noqa: F401,F403
from .volume import *       # noqa: F401,F403
from .boxes import *       # noqa: F401,F403

SOURCES = tuple(__import__("os").path.join(__import__("os").path.dirname(__file__), n + ".py") for n in ("_base", "elementwise", "pixel", "psv", "roi", "conv3d", "conv2d", "wino4", "align", "volume", "boxes"))
