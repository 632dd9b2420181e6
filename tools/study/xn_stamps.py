"""In-kernel timeline of bneck_xn_f16: prints, per step, what each wave role spent in its phases - from
s_memtime stamps of ONE workgroup. Needs a STAMPED build: the shipped kernel executes no stamp. To reproduce, put
`if (blockIdx.x == 7 && (tid & 255) == 0) xn_stamps[(loader ? 128 : 0) + i] = __builtin_amdgcn_s_memtime();` (a __device__
long long xn_stamps[256]) at the phase boundaries named in the print below and export
`extern "C" int yh_debug_xn_stamps(long long* out)` (hipMemcpyFromSymbol); the output of the round-3 run is
profiles/r03_xn_stamps.txt."""
import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
eng = ya.Engine(input_size=700, backbone=101, max_batch=8, use_graph=False)
eng.load_weights(eng.generate_weights(1))
eng.set_input(np.random.default_rng(0).integers(0, 256, (8, 700, 700, 3), dtype=np.uint8))
for _ in range(3): eng.evaluate()
eng.sync()
out = (ctypes.c_longlong * 256)()
so = ctypes.CDLL(ya.lib_path())
so.yh_debug_xn_stamps(out)
t = np.array(out[:], dtype=np.int64)
c, l = t[:128], t[128:]
t0 = c[7]
print("kernel start -> compute at E(-1):", c[2] - t0, " loader: start", l[0] - t0, "issue(0)", l[1] - l[0], "landed", l[2] - l[1], " E(-1) passed at", c[3] - t0, "; loop", c[4] - c[3])
for oc in range(16):
    b = 8 + oc * 6
    print(f"oc {oc:2d} compute: GEMM2+epi {c[b+2]-c[b]:5d} wait M {c[b+3]-c[b+2]:5d} GEMM3 {c[b+5]-c[b+3]:5d} wait E {(c[b+6] if oc<15 else c[4])-c[b+5]:5d} | loader: front issue {l[b+1]-l[b] if oc<15 else 0:5d} wait back {l[b+2]-(l[b+1] if oc<15 else l[b]):5d} wait M {l[b+3]-l[b+2]:5d} stores+back issue {(l[b+4] if oc<15 else l[b+5])-l[b+3]:5d} wait front {l[b+5]-(l[b+4] if oc<15 else l[b+5]):5d} wait E {(l[b+6] if oc<15 else l[b+5])-l[b+5]:5d}")
