"""Drop-in `madrona_gpudrive` module (reference src/bindings.cpp) backed by the MI355X-native
engine in gpudrive_lab_amd.  `import madrona_gpudrive` keeps working for gpudrive.env / datatypes."""
from gpudrive_lab_amd.madrona_gpudrive_impl import (  # noqa: F401
    CollisionBehaviour, DynamicsModel, EntityType, FindRoadObservationsWith, Parameters, RewardParams,
    RewardType, SimManager, episodeLen, kMaxAgentCount, kMaxAgentMapObservationsCount,
    kMaxRoadEntityCount, numLidarSamples, vehicleScale,
)
from . import madrona  # noqa: F401
