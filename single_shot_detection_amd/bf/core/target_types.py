"""bf/core/target_types.py"""
from enum import Enum, auto


class TargetTypes(Enum):
    Boxes = auto()
    NoTarget = auto()
