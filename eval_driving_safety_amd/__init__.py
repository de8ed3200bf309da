"""eval_driving_safety_amd - MI355X-native perturbation engine for the stereo-detector attacks
of DexterJZ/eval_driving_safety (PGD/FGSM and universal-patch loops on DSGN / Stereo R-CNN).

The compute path is ``libadvengine.so`` (hand-written HIP for gfx950 behind the C ABI in
``include/advengine.h``).  There is no CPU or eager-PyTorch fallback: importing
``eval_driving_safety_amd.ops`` without the built library raises.
"""
__version__ = "0.1.0"
