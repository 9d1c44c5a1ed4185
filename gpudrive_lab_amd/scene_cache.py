"""Binary scene cache: scene JSON -> ".gdsm" (the parsed, polyline-reduced
map), so that `SimManager(...)` / `set_maps(...)` stop re-parsing megabytes of JSON on one host thread
(reference src/mgr.cpp:630-647 -> src/MapReader.cpp:46-61).  A ".gdsm" path is accepted wherever a scene
path is; worlds built from it are bit-identical to worlds built from the JSON."""
import os

from . import _capi


def cache_path(scene, out_dir=None):
    base = os.path.splitext(os.path.basename(scene))[0] + ".gdsm"
    return os.path.join(out_dir if out_dir is not None else os.path.dirname(os.path.abspath(scene)), base)


def build_cache(scenes, polyline_reduction_threshold, out_dir=None, force=False):
    """Write (or reuse, when newer than its JSON) one cache file per scene; returns the cache paths in the
    order of `scenes`, ready to be passed to SimManager / set_maps.  The threshold is baked into the file
    and must be the `Parameters.polylineReductionThreshold` used later (ValueError otherwise)."""
    L = _capi.lib()
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
    out = []
    done = {}
    for s in scenes:
        if s in done:
            out.append(done[s])
            continue
        if s.endswith(".gdsm"):
            done[s] = s
            out.append(s)
            continue
        dst = cache_path(s, out_dir)
        fresh = os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(s)
        if force or not fresh:
            _capi.check(L.gd_scene_cache_write(os.fsencode(s), float(polyline_reduction_threshold), os.fsencode(dst)),
                        "gd_scene_cache_write")
        done[s] = dst
        out.append(dst)
    return out
