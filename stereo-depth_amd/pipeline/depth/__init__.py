"""Stereo-matching backends.  Only the hand-written HIP engine ("cuda" in the reference's
configuration vocabulary) is part of this build; the names of the reference's traced-DNN backends
(dnn_stereo_matching_backend.py: MSNet2D/3D, GwcNet -- out of scope) are
kept so that selecting one fails with a clear message."""
from pipeline.depth.cuda_stereo_matching_backend import CudaStereoMatchingBackend
from pipeline.depth.stereo_matching import StereoMatching

AVAILABLE_DNN_BACKENDS = ("msnet2d", "msnet3d", "gwcnet")

__all__ = ["StereoMatching", "CudaStereoMatchingBackend", "AVAILABLE_DNN_BACKENDS"]
