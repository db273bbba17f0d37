"""Module path of the reference (pfrl/wrappers/cast_observation.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import CastObservation, CastObservationToFloat32  # NOQA
