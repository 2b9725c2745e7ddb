"""MI355X-native batched EEPACC MPC engine (hot path of stefavpolito/EEPACC_MPC_CasADi_MATLAB)."""
from .settings import Settings, SetVehicleParameters, GenerateUseCase, SimplifyPWA, Run_DrivingCycle  # noqa: F401

__all__ = ["Settings", "SetVehicleParameters", "GenerateUseCase", "SimplifyPWA", "Run_DrivingCycle"]
