"""MI355X-native GPUDrive step engine: host-side package.

`gpudrive_lab_amd.madrona_gpudrive_impl` mirrors the reference's `madrona_gpudrive` nanobind
module (reference src/bindings.cpp) on top of the C ABI in include/gpudrive_amd.h; the top-level
`madrona_gpudrive` package re-exports it so existing `gpudrive.env` code imports it unchanged.
"""
from ._capi import build, lib_path  # noqa: F401
