"""Command-line counterparts of the reference's attack scripts (same flags, defaults and output
directories):

    python -m eval_driving_safety_amd.cli.dsgn_pgd_attack    <- attack/DSGN/pgd_attack.py
    python -m eval_driving_safety_amd.cli.dsgn_patch_attack  <- attack/DSGN/patch_attack.py
    python -m eval_driving_safety_amd.cli.srcnn_pgd_attack   <- attack/Stereo-RCNN/pgd_attack.py
    python -m eval_driving_safety_amd.cli.srcnn_patch_attack <- attack/Stereo-RCNN/patch_attack.py

Launch with ``python -m torch.distributed.run --nproc-per-node N ...`` to shard the stereo pairs over N GPUs.
"""
