from pipeline.depth.stereo_matching import StereoMatching
from pipeline.depth.cuda_stereo_matching_backend import CudaStereoMatchingBackend

# The DNN backends of the reference (dnn_stereo_matching_backend.py: traced MSNet2D/3D,
# GwcNet) are out of scope for this build.
AVAILABLE_DNN_BACKENDS = ("msnet2d", "msnet3d", "gwcnet")
