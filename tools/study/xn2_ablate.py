"""Timing study of bneck_xn2_f16<128> (batch 64): tune.ablate bits 4.. drop residual DMA (16), y stores (32), W_c DMA (64), W_a' loads (128),
GEMM 2's MFMAs (256), GEMM 3's MFMAs (512), GEMM 2's epilogue (1024). Results are garbage; times only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tm = int(sys.argv[2]) if len(sys.argv) > 2 else 128
frames = np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8)
blob = None
for name, ab in (("everything", 0), ("no residual DMA", 16), ("no y stores", 32), ("no W_c DMA", 64), ("no W_a' loads", 128), ("no memory at all", 16 + 32 + 64 + 128),
                 ("no GEMM 2 MFMAs", 256), ("no GEMM 3 MFMAs", 512), ("no MFMAs", 768), ("no GEMM 2 epilogue", 1024), ("no MFMAs, no epilogue", 768 + 1024),
                 ("no memory, no MFMAs, no epilogue", 16 + 32 + 64 + 128 + 768 + 1024)):
    e = ya.Engine(input_size=550, max_batch=batch, use_graph=False, tune=dict(ablate=ab, xn_tm=tm, xn_pipe=1))
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob); e.set_input(frames); e.evaluate(); e.sync()
    ms = [p["ms"] for p in e.profile(True, 3) if p["name"].startswith("bneck_xn2")]
    print(f"{name:34s}: {len(ms)} launches, mean {1e3 * np.mean(ms):7.1f} us", flush=True)
    e.close()
