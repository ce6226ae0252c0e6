"""MI355X-native ORB front-end + LocalBA hot path of Active-ORB-SLAM2 (C-ABI HIP library + bindings).

The directory name contains '-', so import it through `__graft_entry__.load_package()` (or
importlib) under the module name `active_orb_slam2_amd`.
"""
from . import capi, sharding, synth, scenario, chain, datasets  # noqa: F401
from .capi import (Extractor, Matcher, LocalBA, Frames, ComputeStereoMatches, Vocabulary, LibraryMissing, AosError, device_count, device_local_cpus, bind_to_device_node, lib_path, host_empty)  # noqa: F401
