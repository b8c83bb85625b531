from pipeline.camera.camera import Camera, EvaluationCamera
from pipeline.camera.middlebury_stereo_camera import MiddleBuryStereoCamera
