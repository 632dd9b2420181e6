"""yolact_amd — host-side mirror of the reference's `src/yolact.rs` surface over libyolact_hip.so.

The arithmetic lives in the hand-written HIP library (csrc/); this package is the thin binding a
Python caller uses, and `Yolact` mirrors the reference's public items one for one:

    reference (Rust)                         here
    Yolact::init() -> Yolact      (:17)      Yolact.init(...)
    Yolact::classify(&mut [u32])  (:39)      Yolact.classify(frame_buffer)   # in place

There is no CPU fallback: importing works anywhere, but creating an engine without the built
library or without a GPU raises.
"""
from .capi import (Engine, Group, TfliteEngine, Scene, Tuning, rccl_unique_id, setup_audit, group_broadcast_weights, PRECISION_F16, PRECISION_FP8, tfl_validate, YhError, Config, Detection, lib_path, load_library, version,  # noqa: F401
                   COMPAT_STRICT, COMPAT_SANE, EDIVERGE)
from .yolact import Yolact  # noqa: F401
