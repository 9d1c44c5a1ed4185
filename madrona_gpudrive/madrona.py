"""`madrona_gpudrive.madrona` submodule (reference madrona::py::setupMadronaSubmodule):
ExecMode and the Tensor view type."""
from gpudrive_lab_amd.madrona_gpudrive_impl import ExecMode, Tensor  # noqa: F401
