from .units import Unit, Actor, Sensor
from .object import Object, Box
from .robot import Robot, ArmRobot, LeggedRobot
