"""`Yolact` — mirror of the reference's struct (src/yolact.rs:13-41) over the HIP engine."""
import numpy as np

from .capi import COMPAT_STRICT, Engine


class Yolact:
    """reference: `pub struct Yolact<'a> { interpreter }` (src/yolact.rs:13-15)."""

    def __init__(self, engine, compat_mode):
        self.interpreter = engine
        self.compat_mode = compat_mode

    @classmethod
    def init(cls, weights=None, seed=1, input_size=224, compat_mode=COMPAT_STRICT, device=0, backbone=50):
        """reference: `Yolact::init()` (src/yolact.rs:17-37). The reference hard-codes its model
        path and tile size (224, :143-144); the weights file is absent from the checkout, so
        `weights=None` loads the seeded synthetic blob. Errors raise (the reference `.expect`s)."""
        eng = Engine(input_size=input_size, backbone=backbone, max_batch=2, use_graph=True, device=device)
        eng.load_weights(weights if weights is not None else eng.generate_weights(seed))
        return cls(eng, compat_mode)

    def classify(self, frame_buffer, width=640, height=480):
        """reference: `classify(&mut self, frame_buffer: &mut [u32])` (src/yolact.rs:39-41, :192-234):
        overwrites the packed camera frame (r<<24|g<<16|b<<8, src/scene.rs:86) in place."""
        assert isinstance(frame_buffer, np.ndarray) and frame_buffer.dtype == np.uint32
        self.interpreter.classify_frame(frame_buffer.reshape(-1), width, height, self.compat_mode)
