"""waterlily_amd -- MI355X-native backend for WaterLily's `sim_step!` hot path."""
from .body import AutoBody, NoBody, measure, norm2  # noqa: F401
