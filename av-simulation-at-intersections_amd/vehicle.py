"""Minimal stand-ins for the two reference types the MPC surface touches, so the drop-in can be used
(and tested) without the reference tree: `State` (main/lib/simulation.py:11-19, field order x, y, yaw, v)
and a car-dimensions object exposing `distance_back_to_front_wheel` (main/lib/car_dimensions.py:82-90).
Any object with those attributes works; the reference's own classes are accepted unchanged."""
from dataclasses import dataclass


@dataclass
class State:
    x: float = 0.0
    y: float = 0.0
    yaw: float = 0.0
    v: float = 0.0


class BicycleModelDimensions:
    @property
    def distance_back_to_front_wheel(self) -> float:
        return 2.86
