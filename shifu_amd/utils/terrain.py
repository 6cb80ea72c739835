"""Procedural terrain map: the host/NumPy generator behind TerrainGymEnv.

Mirrors the behaviour of the reference's `Terrain` (shifu/utils/terrain.py:42-173):
a (tot_rows, tot_cols) int16 height map made of num_rows x num_cols square
sub-terrains inside a flat border, `env_origins[row, col]` = centre of each
sub-terrain with z = highest sample of its central 2 m x 2 m, curriculum layout
difficulty = row/num_rows, type choice = col/num_cols + 0.001 (:91-98).  Only
the int16 samples and env_origins ever reach the GPU.
"""
import numpy as np

from shifu_amd.isaacgym import terrain_utils


def gap_terrain(terrain, gap_size, platform_size=1.):
    """Square moat of width gap_size around a centre platform (terrain.py:176-188)."""
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - plat) // 2
    y1 = (terrain.width - plat) // 2
    x2, y2 = x1 + gap, y1 + gap
    terrain.height_field_raw[cx - x2: cx + x2, cy - y2: cy + y2] = -1000
    terrain.height_field_raw[cx - x1: cx + x1, cy - y1: cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.):
    """Square pit of `depth` metres in the centre (terrain.py:191-198)."""
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2 = terrain.length // 2 - half, terrain.length // 2 + half
    y1, y2 = terrain.width // 2 - half, terrain.width // 2 + half
    terrain.height_field_raw[x1:x2, y1:y2] = -d


class Terrain:
    def __init__(self, cfg, num_robots) -> None:
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self.curiculum()
        elif cfg.selected:
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":
            self.vertices, self.triangles = terrain_utils.convert_heightfield_to_trimesh(
                self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale, cfg.slope_treshold)

    # layout policies (terrain.py:81-113) -----------------------------------
    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):  # (sic) the reference's spelling is part of its surface
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                difficulty = i / self.cfg.num_rows
                choice = j / self.cfg.num_cols + 0.001
                self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def selected_terrain(self):
        kwargs = dict(self.cfg.terrain_kwargs)
        generator = getattr(terrain_utils, kwargs.pop("type"))
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            terrain = self._blank()
            generator(terrain, **kwargs)
            self.add_terrain_to_map(terrain, i, j)

    def _blank(self):
        return terrain_utils.SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                                        vertical_scale=self.cfg.vertical_scale,
                                        horizontal_scale=self.cfg.horizontal_scale)

    def make_terrain(self, choice, difficulty):
        """Type by `choice` against the cumulative proportions, hardness by `difficulty`
        (terrain.py:115-154)."""
        terrain = self._blank()
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        obstacle_height = 0.05 + difficulty * 0.2
        stone_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        p = self.proportions
        if choice < p[0]:
            if choice < p[0] / 2:
                slope *= -1
            terrain_utils.pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.)
        elif choice < p[1]:
            terrain_utils.pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.)
            terrain_utils.random_uniform_terrain(terrain, min_height=-0.05, max_height=0.05, step=0.005,
                                                 downsampled_scale=0.2)
        elif choice < p[3]:
            if choice < p[2]:
                step_height *= -1
            terrain_utils.pyramid_stairs_terrain(terrain, step_width=0.31, step_height=step_height, platform_size=3.)
        elif choice < p[4]:
            terrain_utils.discrete_obstacles_terrain(terrain, obstacle_height, 1., 2., 20, platform_size=3.)
        elif len(p) > 5 and choice < p[5]:
            terrain_utils.stepping_stones_terrain(terrain, stone_size=stone_size, stone_distance=stone_distance,
                                                  max_height=0., platform_size=4.)
        elif len(p) > 6 and choice < p[6]:
            gap_terrain(terrain, gap_size=1. * difficulty, platform_size=3.)
        else:
            pit_terrain(terrain, depth=1. * difficulty, platform_size=4.)
        return terrain

    def add_terrain_to_map(self, terrain, row, col):
        i, j = row, col
        sx, ex = self.border + i * self.length_per_env_pixels, self.border + (i + 1) * self.length_per_env_pixels
        sy, ey = self.border + j * self.width_per_env_pixels, self.border + (j + 1) * self.width_per_env_pixels
        self.height_field_raw[sx:ex, sy:ey] = terrain.height_field_raw
        x1 = int((self.env_length / 2. - 1) / terrain.horizontal_scale)
        x2 = int((self.env_length / 2. + 1) / terrain.horizontal_scale)
        y1 = int((self.env_width / 2. - 1) / terrain.horizontal_scale)
        y2 = int((self.env_width / 2. + 1) / terrain.horizontal_scale)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]
