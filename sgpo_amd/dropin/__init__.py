"""Host-side mirror of the reference's plugin interface for the sampling path.

The reference registers its components through hydra `_target_` strings
(`models.mdlm.MDLM`, `sampling.cls_guidance.Classifier_Guidance_Inpaint`,
`problem.protein.ProteinPredictor`, ... -- SURVEY.md section 8b).  Putting this directory
on `sys.path` (or calling `sgpo_amd.install_dropin()`) makes those dotted paths
resolve to the HIP-backed implementations, so the reference's drivers and yaml
configs run unchanged.
"""
import os
import sys

DROPIN_DIR = os.path.dirname(os.path.abspath(__file__))


def install():
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    return DROPIN_DIR
