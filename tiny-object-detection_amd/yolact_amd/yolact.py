"""`Yolact` — mirror of the reference's struct (src/yolact.rs:13-41) over the HIP library."""
import os

import numpy as np

from .capi import COMPAT_STRICT, Engine, TfliteEngine


class Yolact:
    """reference: `pub struct Yolact<'a> { interpreter }` (src/yolact.rs:13-15)."""

    def __init__(self, engine, compat_mode):
        self.interpreter = engine
        self.compat_mode = compat_mode

    @classmethod
    def init(cls, model_path=None, weights=None, seed=1, input_size=224, compat_mode=COMPAT_STRICT, device=0, backbone=50):
        """reference: `Yolact::init()` (src/yolact.rs:17-37), which hard-codes
        "data/FRC_model_edgetpu.tflite". Here:
          * model_path = a non-EdgeTPU .tflite (the reference's data/FRC_model.tflite family: uint8
            MobileNetV2-style graph) -> the TFLite model path runs it on the GPU;
          * otherwise the YOLACT R50/R101 engine at `input_size` (224 = the reference's tile,
            yolact.rs:143-144) with `weights` (a YHW1 blob) or the seeded synthetic blob.
        Errors raise (the reference `.expect`s every failure)."""
        if model_path is not None:
            if not os.path.exists(model_path):
                raise FileNotFoundError(f"failed to load model: {model_path}")   # yolact.rs:20
            with open(model_path, "rb") as f:
                return cls(TfliteEngine(f.read(), device=device), compat_mode)
        eng = Engine(input_size=input_size, backbone=backbone, max_batch=2, use_graph=True, device=device)
        eng.load_weights(weights if weights is not None else eng.generate_weights(seed))
        return cls(eng, compat_mode)

    def classify(self, frame_buffer, width=640, height=480):
        """reference: `classify(&mut self, frame_buffer: &mut [u32])` (src/yolact.rs:39-41, :192-234):
        overwrites the packed camera frame (r<<24|g<<16|b<<8, src/scene.rs:86) in place."""
        assert isinstance(frame_buffer, np.ndarray) and frame_buffer.dtype == np.uint32
        self.interpreter.classify_frame(frame_buffer.reshape(-1), width, height, self.compat_mode)
