"""Frame sources: the two abstract camera types and the dataset readers of row f4."""
from pipeline.camera.camera import Camera, EvaluationCamera
from pipeline.camera.kitti_single_view_camera import KittiSingleViewCamera
from pipeline.camera.middlebury_stereo_camera import MiddleBuryStereoCamera

__all__ = ["Camera", "EvaluationCamera", "KittiSingleViewCamera", "MiddleBuryStereoCamera"]
