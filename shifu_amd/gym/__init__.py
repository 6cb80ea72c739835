from .isaac_gym import IsaacGymEnv, TerrainGymEnv
from .env import ShifuVecEnv
