"""Nested-class config system (reference shifu/configs/base_config.py:37-58): every
class-valued attribute is replaced by an instance, recursively, so that user configs
written as nested classes (`class sim(TerrainEnvConfig.sim): dt = 0.005`) resolve by
ordinary inheritance.  `name` mirrors the reference's BaseConfig.name."""
import inspect


class BaseConfig:
    name = None

    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for key in dir(obj):
            if key == "__class__":
                continue
            member = getattr(obj, key)
            if inspect.isclass(member):
                instance = member()
                setattr(obj, key, instance)
                BaseConfig.init_member_classes(instance)
