"""eval_driving_safety_amd - MI355X-native perturbation engine for the stereo-detector attacks
of DexterJZ/eval_driving_safety (PGD/FGSM and universal-patch loops on DSGN / Stereo R-CNN).

The compute path is ``libadvengine.so`` (hand-written HIP for gfx950 behind the C ABI in
``include/advengine.h``).  There is no CPU or eager-PyTorch fallback: importing
``eval_driving_safety_amd.ops`` without the built library raises.
"""
__version__ = "0.1.0"

import os as _os

# the host driver of the MI355X pools supports dmabuf IPC only; ROCr reads this at hsa_init, i.e. it must be in the
# environment before ANY torch.cuda call of the process (RCCL and cross-process tensor sharing fail without it)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
