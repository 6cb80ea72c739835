"""Sensor configs (reference shifu/configs/sensor_config.py).  Kept so user configs
import; rasterised cameras themselves are out of scope on this backend."""
from shifu_amd.isaacgym import gymapi

from .base_config import BaseConfig


class BaseSensorConfig(BaseConfig):
    name = "DummySensor"
    frequency = 30
    data_shape = 0


class CameraSensorConfig(BaseSensorConfig):
    name = "DummyCameraSensor"
    image_types = [gymapi.IMAGE_COLOR, gymapi.IMAGE_DEPTH, gymapi.IMAGE_SEGMENTATION, gymapi.IMAGE_OPTICAL_FLOW]
    image_normalization = False
    local_lookat_positions = None
    transform = None
    attach_local_transform = None

    def __init__(self):
        props = gymapi.CameraProperties()
        for attr in dir(self.camera_props):
            if '__' not in attr:
                setattr(props, attr, getattr(self.camera_props, attr))
        self.camera_props = props
        super().__init__()

    class camera_props:
        enable_tensors = True
        use_collision_geometry = False
        width = 256
        height = 256
        near_plane = 0.1
        far_plane = 3
        horizontal_fov = 87
