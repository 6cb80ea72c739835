from .base_config import BaseConfig
from .asset_config import ActorConfig, BoxActorConfig, ArmRobotActorConfig, LeggedRobotActorConfig
from .env_config import BaseEnvConfig, TerrainEnvConfig
from .sensor_config import BaseSensorConfig, CameraSensorConfig
from .policy_config import PPOConfig
