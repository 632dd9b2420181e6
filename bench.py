#!/usr/bin/env python3
"""bench.py — YOLACT-550 frames/s on N MI355X (one process per GPU), plus the roofline of the
dominant kernel and the CPU oracle timed on the host cores.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
yh_set_input_u8_device (device->device) + yh_evaluate (preprocess, 86 convs, FPN, protonet, heads,
softmax/top-k/Fast-NMS/mask assembly), hipGraph-replayed. Frames shard across ranks (weak scaling,
--batch frames per GPU per step); the only collective is the one-time RCCL broadcast of the weight
blob from rank 0. PyTorch is plumbing here (device buffers, torch.distributed), not the product.
"""
import argparse
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16
MFMA_FP8_DENSE_PEAK_TFLOPS = 5000.0   # same guide: ~5 PF dense fp8 (block-scaled MFMA)
HBM_PEAK_GBS = 8000.0


def shard_frames(total_frames, world_size, rank):
    """Contiguous block partition of frame indices: returns (start, count)."""
    base, rem = divmod(total_frames, world_size)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def broadcast_weights(dist, blob):
    """Replicates the weight blob from rank 0 with torch.distributed (gloo in the CPU test and the one-GPU
    rehearsal; the fallback on GPUs). `blob` is a uint8 tensor of identical size on every rank."""
    if dist is not None:
        dist.broadcast(blob, src=0)
    return blob


def all_ranks_ok(dist, ok, device):
    """True iff `ok` holds on every rank (one MIN all-reduce), so that all ranks take the same branch."""
    if dist is None:
        return bool(ok)
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


LIBRARY_BROADCAST_DEADLINE_S = 120
_LIBRARY_CALL_STUCK = False   # on THIS rank a helper thread is still inside yh_rank_broadcast_weights: leave through os._exit


def leave_if_stuck():
    """A rank whose helper thread is still blocked inside the library's RCCL call must not run any teardown that could wait for
    it (interpreter exit joins nothing, but HIP / RCCL teardown with a thread inside ncclCommInitRank is the hang the deadline
    exists to avoid): flush and leave. Called by EVERY rank right after its last collective."""
    if _LIBRARY_CALL_STUCK:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def replicate_weights(ya, torch, dist, rank, world, local_rank, src, seed, use_library=True, device=None, make_engine=None):
    """The path's ONE collective: rank 0's weights -> every rank's weight-holding engine. Preferred: the library's own RCCL
    broadcast (yh_rank_broadcast_weights: ncclCommInitRank + ncclBroadcast of the canonical blob over xGMI - the
    call a Rust host would make); if librccl cannot be opened on some rank, or the call fails on
    some rank, every rank falls back to torch.distributed.broadcast + yh_load_weights_device. Returns (how it went, the
    engine that holds the weights): that is `src`, unless this rank's library call is STUCK - then `src` is never touched
    again (the blocked thread still owns it: no load, no close - the handle is leaked on purpose) and the fallback loads
    into a fresh engine from make_engine()."""
    dev = device or f"cuda:{local_rank}"
    blob_host = None
    if rank == 0:
        blob_host = src.generate_weights(seed)
        src.load_weights(blob_host)
    if world == 1:
        return "single GPU (no collective)", src
    why = ""
    stuck_here = False
    if use_library:
        ident = None
        try:
            ident = ya.rccl_unique_id()          # every rank opens librccl here, before anyone commits to the path
        except Exception as e:                   # noqa: BLE001 - reported, not swallowed
            why = f"librccl unavailable on rank {rank}: {e}"
        if all_ranks_ok(dist, ident is not None, dev):
            box = [ident if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)   # the id travels by the host's own means (128 bytes)
            # The library's RCCL path has never run with more than one rank on hardware this project could reach (INTEGRATION.md
            # §4), and a rank that fails inside ncclCommInitRank leaves its peers blocked there. A measurement must not hang on
            # it: the call runs in a helper thread under a deadline; a rank that misses it reports so, every rank falls back to
            # torch.distributed.broadcast, and the process leaves through os._exit at the end (the blocked call is never joined).
            res = {}

            def _call():
                try:
                    src.rank_broadcast_weights(box[0], rank, world, 0)
                    res["ok"] = True
                except Exception as e:           # noqa: BLE001
                    res["err"] = str(e)
            th = threading.Thread(target=_call, daemon=True)
            th.start()
            th.join(timeout=LIBRARY_BROADCAST_DEADLINE_S)
            ok = bool(res.get("ok"))
            if th.is_alive():
                global _LIBRARY_CALL_STUCK
                _LIBRARY_CALL_STUCK = stuck_here = True
                why = f"yh_rank_broadcast_weights did not return within {LIBRARY_BROADCAST_DEADLINE_S} s on rank {rank}"
            elif not ok:
                why = f"yh_rank_broadcast_weights failed on rank {rank}: {res.get('err')}"
            if all_ranks_ok(dist, ok, dev):
                return "yh_rank_broadcast_weights (library RCCL: ncclCommInitRank + ncclBroadcast)", src
            if not why:
                why = "yh_rank_broadcast_weights failed or timed out on another rank"
    if stuck_here:
        # the blocked thread may write into `src` whenever its peers' sockets close: a fresh handle takes the weights
        if make_engine is None:
            raise RuntimeError("the library's RCCL call is stuck on this rank and no engine factory was given")
        src = make_engine()
        if rank == 0:
            src.load_weights(blob_host)
    nbytes = src.weights_nbytes()
    blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if rank == 0:
        blob.copy_(torch.from_numpy(blob_host))
    broadcast_weights(dist, blob)
    if blob.is_cuda:
        torch.cuda.synchronize()
    if rank != 0:
        src.load_weights_device(blob.data_ptr(), nbytes)
    return "torch.distributed.broadcast + yh_load_weights_device" + (f" (library path not taken: {why})" if why else ""), src


def max_over_ranks(dist, seconds, device):
    """The step time the contract asks for: MAX over ranks of the locally measured duration."""
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def kernel_family(sym):
    """Tile family of a kernel symbol: `conv_igemm_f16<256,256,2,4,0,2,mfma16>[+1x1]` -> `conv_igemm_f16<256,256,2,4,0,2>` - the
    bracket suffixes ([ml] multi-level, [+1x1] fused tail, [3x3] the streaming tile's 3x3 form) and the flag arguments (mfma16 ...)
    name template flags of ONE tile (conv_igemm.hip), which rocprofv3 lists as symbols of their own; a roofline chosen by symbol
    string split the 256 x 256 tile three ways and never reported the step's largest family."""
    base = sym.split("[")[0]
    if base.startswith("conv_igemm_f16<") and base.endswith(">"):
        args = base[len("conv_igemm_f16<"):-1].split(",")
        fam = "conv_igemm_f16<" + ",".join(args[:6]) + ">"
        # Round 5: ONE bound per family. The single-stage streaming tiles (STAGES = 1: <128,128,2,2,0,1>, <64,256,1,4,0,1>)
        # carry two populations - HBM-bound 1x1 launches (3.2-3.4 TB/s, 0.13 of the MFMA peak) and MFMA-bound 3x3 launches (850-870
        # TFLOP/s, 0.13 of HBM) - which one "hbm 0.30" label described neither of: the kernel extent is part of their family name.
        if len(args) >= 6 and args[5] == "1" and "[3x3]" in sym:
            fam += "[3x3]"
        return fam
    return base


def dominant_kernel(prof, key=None):
    """Groups per-launch hipEvent timings by kernel symbol (key = kernel_family: by tile family); returns the group with most time."""
    by = {}
    for p in prof:
        sym = p["name"].split(":")[0]
        k = key(sym) if key else sym
        d = by.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, symbols={}))
        d["ms"] += p["ms"]; d["flops"] += p["flops"]; d["bytes"] += p["bytes"]; d["launches"] += 1
        d["symbols"][sym] = d["symbols"].get(sym, 0) + 1
    sym = max(by, key=lambda k: by[k]["ms"])
    return sym, by


def measured_traffic(sym, batch):
    """HBM-side bytes per launch of `sym` from the committed PMC passes (profiles/*_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied by
    tools/make_traffic.py). PMC cannot be collected from inside this process; None if the file has
    no entry for this kernel at this batch size."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        ks = t.get("kernels", {})
        syms = sym if isinstance(sym, dict) else {sym: 1}      # a family: {symbol: launches per step}
        if t.get("batch") == batch and all(k in ks for k in syms):
            n = sum(syms.values())
            return round(sum(ks[k]["hbm_bytes_per_launch"] * c for k, c in syms.items()) / n), os.path.relpath(path, ROOT)
    return None, None


def roofline_of(prof, batch=None, rank=0):
    """The step's dominant TILE FAMILY (kernel_family; rank = 1: the second largest) with the roofline that binds it. The two largest
    families of a batch-64 step tie at about a third of the step each and swap places from box to box (round 4: the driver's record
    named the streaming tile, the profile box's the 256 x 256 tile), so the bench line carries both - `roofline` and
    `roofline_second` - and the documents quote both by name."""
    _, by = dominant_kernel(prof, key=kernel_family)
    order = sorted(by, key=lambda k: -by[k]["ms"])
    if rank >= len(order):
        return None
    sym = order[rank]
    d = by[sym]
    total_ms = sum(p["ms"] for p in prof)
    # the roofline that binds this kernel: the larger of its MFMA and HBM fractions (a 1x1 conv on MFMA instructions is
    # HBM-bound: 0.15 of the matrix peak but 0.6 of the memory peak)
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    mfma_peak = MFMA_FP8_DENSE_PEAK_TFLOPS if sym.startswith("conv_igemm_fp8") else MFMA_F16_DENSE_PEAK_TFLOPS   # the kernel's own dtype
    if tf / mfma_peak >= gbs / HBM_PEAK_GBS:
        r = dict(bound="mfma", achieved=round(tf, 2), peak=mfma_peak, unit="TFLOP/s", frac=round(tf / mfma_peak, 4), traffic=None)
    else:
        r = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None)
    r.update(kernel=sym, symbols=d["symbols"], launches=d["launches"], avg_launch_ms=round(d["ms"] / d["launches"], 5),
             share_of_step=round(d["ms"] / total_ms, 3), algorithmic_gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 3),
             algorithmic_gbytes_hbm=round(d["bytes"] / 1e9, 4),
             hbm_gbs_algorithmic=round(gbs, 1), mfma_tflops=round(tf, 2),
             timing="avg_launch_ms: hipEvent-bracketed launches of the live engine, one kernel on the chip at a time. In the replayed step "
                    "the head / tail run on a second stream beside the protonet: a kernel trace of the same command (profiles/rNN_rocprofv3_"
                    "kernel_stats.csv) shows those launches' CONTENDED durations, longer than these by the share of the chip the other stream takes; "
                    "launches of the backbone phase (one stream) agree with the trace")
    traffic, src = measured_traffic(d["symbols"], batch)
    if traffic is not None:
        r["traffic"] = traffic
        r["traffic_unit"] = "HBM-side bytes per launch (FETCH_SIZE*2 + WRITE_SIZE, KiB->B), vs algorithmic %d" % round(d["bytes"] / d["launches"])
        r["traffic_source"] = src
    return r


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may really use: the smallest of the machine's count, the affinity mask and the cgroup
    CPU quota (a GPU box hands a one-GPU job a share of its 256 hardware threads: running 256 OpenMP threads inside a
    16-core quota measured 0.09 frames/s where 16 threads give 3.6)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def latency_stats(eng, n=200):
    """Device time of single steps (one hipEvent pair each on the engine's stream, graph replay):
    median and p99 over n steps (SURVEY.md §8d timing method)."""
    import numpy as np
    t = np.array([eng.time_steps(1) for _ in range(n)])
    return dict(n=n, median_ms=round(float(np.median(t)), 4), p99_ms=round(float(np.percentile(t, 99)), 4),
                min_ms=round(float(t.min()), 4))


def pcie_inclusive(eng, host_frames, steps=20):
    """Wall-clock frames/s when the boundary is handed HOST buffers: yh_set_input_u8 (host -> device copy of the uint8 frames
    into the input buffer the running step does not read, on the engine's copy stream) + yh_evaluate, `steps` times back to
    back, one sync at the end - the copy of batch k+1 runs underneath step k. Never the headline value."""
    n = host_frames.shape[0]
    steps = steps if n > 4 else 100
    eng.set_input(host_frames); eng.evaluate(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.set_input(host_frames); eng.evaluate()
    eng.sync()
    return round(n * steps / (time.perf_counter() - t0), 2)


def host_to_detections_latency(eng, host_frames, reps=100):
    """Wall-clock latency of one synchronous request: host frames in (yh_set_input_u8), step, sync,
    detections of frame 0 read back to the host (boxes, scores, classes; no masks)."""
    import numpy as np
    t = []
    for _ in range(reps + 5):
        t0 = time.perf_counter()
        eng.set_input(host_frames); eng.evaluate(); eng.sync(); eng.detections(0, want_masks=False)
        t.append(time.perf_counter() - t0)
    t = np.array(t[5:]) * 1e3
    return dict(n=reps, median_ms=round(float(np.median(t)), 4), p99_ms=round(float(np.percentile(t, 99)), 4))


def accuracy_vs_oracle(eng_out, orc_out, cut_band=0.02):
    """Engine vs CPU oracle on the same frame and weights (the restated acceptance target of SURVEY.md §8c; NOT
    parity with CPU tflite). Order-insensitive and over ALL detections of both sides:
      mask_iou_all  per class, the union of the engine's masks against the union of the oracle's masks,
                    intersections and unions summed over the classes either side detected - a detection only one
                    side has counts against it with every pixel of its mask;
      matched / unmatched_*  detections paired by (class, prior);
      mask_iou_matched  the survivor statistic round 1 reported (matched pairs only), kept for comparison;
      mask_iou_above_cut / unmatched_above_cut  the same union figure and count WITHOUT the detections that sit on the
                    top-k cut: when both lists are full (max_dets each), the last places go to whichever candidates are
                    ahead in the 4th decimal of the score, and f16 summation order decides that - detections of either
                    side whose score is within cut_band (2 %) of the lowest score kept are left out of this pair of figures
                    (and only of this pair)."""
    import numpy as np
    (ed, em), (od, om) = eng_out, orc_out
    ek = {(d["class_id"], d["prior"]): i for i, d in enumerate(ed)}
    inter = union = matched = 0
    for j, d in enumerate(od):
        i = ek.get((d["class_id"], d["prior"]))
        if i is None:
            continue
        matched += 1
        a, b = em[i] > 0, om[j] > 0
        inter += int(np.logical_and(a, b).sum()); union += int(np.logical_or(a, b).sum())
    shape = em.shape[1:] if len(ed) else om.shape[1:]

    def union_iou(keep_e, keep_o):
        ia = ua = 0
        for c in sorted({ed[i]["class_id"] for i in keep_e} | {od[j]["class_id"] for j in keep_o}):
            ue, uo = np.zeros(shape, bool), np.zeros(shape, bool)
            for i in keep_e:
                if ed[i]["class_id"] == c:
                    ue |= em[i] > 0
            for j in keep_o:
                if od[j]["class_id"] == c:
                    uo |= om[j] > 0
            ia += int((ue & uo).sum()); ua += int((ue | uo).sum())
        return round(ia / ua, 5) if ua else None
    all_e, all_o = list(range(len(ed))), list(range(len(od)))
    out = dict(oracle_dets=len(od), engine_dets=len(ed), matched_class_and_prior=matched,
               unmatched_oracle=len(od) - matched, unmatched_engine=len(ed) - matched,
               mask_iou_all=union_iou(all_e, all_o),
               mask_iou_matched=round(inter / union, 5) if union else None)
    if len(ed) and len(ed) == len(od) and all("score" in d for d in ed + od):   # both lists full: there is a cut
        cut = max(min(d["score"] for d in ed), min(d["score"] for d in od)) * (1.0 + cut_band)
        key_e, key_o = [(d["class_id"], d["prior"]) for d in ed], [(d["class_id"], d["prior"]) for d in od]
        above = {key_e[i] for i in all_e if ed[i]["score"] > cut} | {key_o[j] for j in all_o if od[j]["score"] > cut}
        # (a detection above the band on one side and inside it on the other stays in on both sides)
        ke, ko = [i for i in all_e if key_e[i] in above], [j for j in all_o if key_o[j] in above]
        out.update(mask_iou_above_cut=union_iou(ke, ko), unmatched_above_cut=len(above - set(key_e)) + len(above - set(key_o)),
                   dets_above_cut=len(above))
    return out


def parse_blob(blob):
    """Canonical weight blob ("YHW1") -> list of (weight [cout][kh][kw][cin] f32, bias [cout] f32)."""
    import numpy as np
    b = np.ascontiguousarray(blob, np.uint8)
    assert bytes(b[:4]) == b"YHW1"
    n = int(b[4:8].view(np.uint32)[0])
    off, out = 16, []
    for _ in range(n):
        cout, cin, kh, kw = (int(v) for v in b[off:off + 16].view(np.uint32))
        off += 16
        ne = cout * kh * kw * cin
        w = b[off:off + 2 * ne].view(np.float16).astype(np.float32).reshape(cout, kh, kw, cin)
        off += (2 * ne + 15) & ~15
        bias = b[off:off + 4 * cout].view(np.float32).copy()
        off += (4 * cout + 15) & ~15
        out.append((w, bias))
    assert off == b.size
    return out


def torch_cpu_forward(torch, convs, frames_u8, backbone=50):
    """The same layer table as a plain torch-CPU f32 forward on the same weights: the stronger CPU
    number SURVEY.md §8(d) asks for beside the oracle. Returns (loc, conf, mask, proto) as the engine lays them out."""
    F = torch.nn.functional
    it = iter([(torch.from_numpy(w).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last), torch.from_numpy(b)) for w, b in convs])

    def conv(x, stride=1, pad=0, act=True, res=None, wb=None):
        w, b = wb if wb is not None else next(it)
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
        if res is not None:
            y = y + res
        return F.relu(y) if act else y
    mean = torch.tensor([123.68, 116.78, 103.94]).view(1, 3, 1, 1)
    std = torch.tensor([58.40, 57.12, 57.38]).view(1, 3, 1, 1)
    x = ((torch.from_numpy(frames_u8).permute(0, 3, 1, 2).float() - mean) / std).contiguous(memory_format=torch.channels_last)
    x = F.max_pool2d(conv(x, 2, 3), 3, 2, 1)
    feats = []
    for L, nb in enumerate((3, 4, 23, 3) if backbone == 101 else (3, 4, 6, 3)):
        for blk in range(nb):
            stride = 2 if (blk == 0 and L > 0) else 1
            a = conv(x)
            bt = conv(a, stride, 1)
            wb3 = next(it)                        # canonical order: conv3, then the projection (evaluated first)
            res = conv(x, stride, 0, act=False) if blk == 0 else x
            x = conv(bt, res=res, wb=wb3)
        feats.append(x)
    lat5 = conv(feats[3], act=False)
    up = lambda t, like: F.interpolate(t, size=like.shape[-2:], mode="bilinear", align_corners=False)
    lat4 = conv(feats[2], act=False, res=up(lat5, feats[2]))
    lat3 = conv(feats[1], act=False, res=up(lat4, feats[1]))
    p5, p4, p3 = conv(lat5, 1, 1), conv(lat4, 1, 1), conv(lat3, 1, 1)
    p6 = conv(p5, 2, 1, act=False)
    p7 = conv(p6, 2, 1, act=False)
    q = p3
    for _ in range(3):
        q = conv(q, 1, 1)
    q = conv(F.interpolate(q, scale_factor=2, mode="bilinear", align_corners=False), 1, 1)
    proto = conv(q)
    head = [next(it) for _ in range(4)]
    loc, cf, mk = [], [], []
    n = x.shape[0]
    for p in (p3, p4, p5, p6, p7):
        t = conv(p, 1, 1, wb=head[0])
        loc.append(conv(t, 1, 1, act=False, wb=head[1]).permute(0, 2, 3, 1).reshape(n, -1, 4))
        cf.append(conv(t, 1, 1, act=False, wb=head[2]).permute(0, 2, 3, 1).reshape(n, -1, head[2][0].shape[0] // 3))
        mk.append(torch.tanh(conv(t, 1, 1, act=False, wb=head[3])).permute(0, 2, 3, 1).reshape(n, -1, 32))
    return torch.cat(loc, 1), torch.cat(cf, 1), torch.cat(mk, 1), proto.permute(0, 2, 3, 1).contiguous()


def acceptance_frame(size):
    """The reference's own test image (data/frc_balls.png, committed as a fixture under tests/golden/) resized to the
    configuration's input size - the frame SURVEY.md §8(c)'s restated acceptance target is quoted on. None if absent."""
    path = os.path.join(ROOT, "tests", "golden", "frc_balls.png")
    try:
        import numpy as np
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR))[None])
    except Exception:   # noqa: BLE001 - the bench then reports the synthetic frame only
        return None


def cpu_baseline(seed, frames_u8, budget_s=12.0, max_frames=64, eng_out=None, backbone=50, fp8_layers=None, accept=None):
    """The CPU oracle ("port": oracle/orc_net.c + orc_detect.c, the checker, never the product) timed on ALL of
    this host's cores on a bounded sample of the same workload: whole frames (forward + tail) until ~budget_s
    seconds have elapsed; beside it one frame at the reference's 4 threads (src/yolact.rs:34), one at 16, and a
    torch-CPU f32 forward of the same layer table on the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    usable, host = usable_cores(), os.cpu_count() or 1
    net = O.Net(backbone, frames_u8.shape[1], 81, seed=seed)
    pri = net.priors()

    def one(nthreads):
        t = time.perf_counter()
        h = net.forward(frames_u8[:1], f16=True, nthreads=nthreads)
        out = O.detect(h[0][0], h[1][0], h[2][0], h[3][0], pri)
        return time.perf_counter() - t, out
    # one frame at each candidate thread count (the reference pins its interpreter to 4: src/yolact.rs:34), then the
    # bounded sample at the count that was fastest: more threads than the quota or the memory system feeds are slower
    cand = sorted({c for c in (4, 16, 32, 64, usable) if c <= usable} | {min(4, usable)})
    per, first = {}, None
    one(cand[-1])   # warm-up: page in the weights, spin up the OpenMP pool
    for c in cand:
        dt1, out = one(c)
        per[c] = round(1.0 / dt1, 4)
        first = first or out
    cores = max(per, key=per.get)
    done, t0 = 0, time.perf_counter()
    while done < max_frames and (done == 0 or time.perf_counter() - t0 < budget_s):
        one(cores)
        done += 1
    dt = time.perf_counter() - t0
    r = dict(value=round(done / dt, 4), unit="frames/s", cores=cores, kind="port",
             sample=f"{done} frame(s) {frames_u8.shape[1]}x{frames_u8.shape[2]}x3, full forward + detection tail, "
                    f"f16-storage oracle (OpenMP, {cores} threads: the fastest of {cand} tried), {dt:.1f} s of CPU work",
             cpu_model=cpu_model(), host_cores=host, usable_cores=usable, frames_per_s_by_threads={str(k): v for k, v in per.items()},
             value_4_threads=per.get(min(4, usable)))
    try:   # the stronger CPU number: torch's own convolutions (oneDNN) on the same weights and frame, f32
        import torch
        torch.set_num_threads(usable)
        convs = parse_blob(net.blob)
        with torch.no_grad():
            torch_cpu_forward(torch, convs, frames_u8[:1], backbone)                      # warm-up (primitive caches)
            reps, tt = 0, time.perf_counter()
            while reps < 20 and (reps == 0 or time.perf_counter() - tt < budget_s / 3):
                th = torch_cpu_forward(torch, convs, frames_u8[:1], backbone)
                reps += 1
            tt = (time.perf_counter() - tt) / reps
        f32 = net.forward(frames_u8[:1], f16=False, nthreads=cores)
        tt = max(tt, 1e-9)
        r["torch_cpu"] = dict(value=round(1.0 / tt, 4), unit="frames/s", threads=torch.get_num_threads(), dtype="f32",
                              sample=f"{reps} forward(s) of the same frame, forward only (no detection tail), torch {torch.__version__} CPU",
                              max_abs_diff_vs_oracle_f32=[round(float(np.abs(a.numpy() - b).max()), 6) for a, b in zip(th, f32)])
    except Exception as e:   # noqa: BLE001 - reported in the line, never fatal for the GPU measurement
        r["torch_cpu"] = dict(error=str(e)[:200])
    if accept is not None and not fp8_layers:
        # the restated acceptance target: the reference's test image at this input size, engine vs oracle
        frame, dets = accept
        h = net.forward(frame, f16=True, nthreads=cores)
        r["engine_vs_oracle_frc_balls"] = accuracy_vs_oracle(dets, O.detect(h[0][0], h[1][0], h[2][0], h[3][0], pri))
        r["engine_vs_oracle_frc_balls"]["frame"] = f"tests/golden/frc_balls.png resized to {frame.shape[1]}x{frame.shape[2]} (bilinear)"
    if eng_out is not None and not fp8_layers:
        r["engine_vs_oracle_same_frame"] = accuracy_vs_oracle(eng_out, first)
        r["engine_vs_oracle_same_frame"]["frame"] = "synthetic frame 0 of the bench (uniform noise): the CPU sample's frame"
    elif eng_out is not None:
        # configs[4]: the engine ran its fp8 form; the oracle's fp8 mode with the engine's calibrated scales is its
        # checker, and the gap to the f16 oracle is a reported property of the configuration
        r["engine_fp8_vs_oracle_f16_same_frame"] = accuracy_vs_oracle(eng_out, first)
        lay = {}
        for name, sc in fp8_layers:
            for nm in ([f"{name}{l}" for l in range(5)] if name in ("head_t", "head_out") else [name]):
                lay[nm] = sc
        net.set_fp8(lay)
        h8 = net.forward(frames_u8[:1], f16=True, nthreads=cores)
        net.set_fp8(None)
        r["engine_vs_oracle_same_frame"] = accuracy_vs_oracle(eng_out, O.detect(h8[0][0], h8[1][0], h8[2][0], h8[3][0], pri))
        r["engine_vs_oracle_same_frame"]["oracle_mode"] = f"fp8 forward mode, {len(fp8_layers)} E4M3 layers, the engine's calibrated scales"
    return r


def pool_accuracy(per_frame):
    """Pools accuracy_vs_oracle records of several frames: counts add; the IoU figures are pooled over pixels the way one frame's
    are over classes (a frame's union-of-masks intersection and union cannot be recovered from its ratio, so the pooled figure is
    the detection-weighted mean of the frames' figures - stated as such)."""
    out = dict(frames=len(per_frame))
    for k in ("oracle_dets", "engine_dets", "matched_class_and_prior", "unmatched_oracle", "unmatched_engine"):
        out[k] = sum(r[k] for r in per_frame)
    for k in ("mask_iou_all", "mask_iou_matched"):
        w = [(r[k], max(r["oracle_dets"], r["engine_dets"])) for r in per_frame if r.get(k) is not None]
        out[k + "_mean"] = round(sum(v * n for v, n in w) / max(1, sum(n for _, n in w)), 5) if w else None
        out[k + "_per_frame"] = [r.get(k) for r in per_frame]
    out["matched_fraction_of_oracle"] = round(out["matched_class_and_prior"] / max(1, out["oracle_dets"]), 4)
    return out


def fp8_vs_oracles(seed, frames_u8, eng_outs, fp8_layers, backbone, nthreads):
    """configs[4]: the fp8 engine's detections on its frames against (a) the oracle's fp8 forward mode with the engine's own
    calibrated scales - its checker - and (b) the f16 oracle - the price of the precision. The checker, never
    the product. frames_u8 [k][S][S][3], eng_outs = the engine's (detections, masks) of each: frame 0's figures as before, and
    the figures pooled over all k frames (one noise frame's detection list is a handful of decisions near the threshold)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    net = O.Net(backbone, frames_u8.shape[1], 81, seed=seed)
    pri = net.priors()
    lay = {}
    for name, sc in fp8_layers:
        for nm in ([f"{name}{l}" for l in range(5)] if name in ("head_t", "head_out") else [name]):
            lay[nm] = sc
    out, a16, a8 = {}, [], []
    for k in range(frames_u8.shape[0]):
        t0 = time.perf_counter()
        h16 = net.forward(frames_u8[k:k + 1], f16=True, nthreads=nthreads)
        if k == 0:
            out["oracle_f16_forward_s"] = round(time.perf_counter() - t0, 2)
        a16.append(accuracy_vs_oracle(eng_outs[k], O.detect(h16[0][0], h16[1][0], h16[2][0], h16[3][0], pri)))
        net.set_fp8(lay)
        h8 = net.forward(frames_u8[k:k + 1], f16=True, nthreads=nthreads)
        net.set_fp8(None)
        a8.append(accuracy_vs_oracle(eng_outs[k], O.detect(h8[0][0], h8[1][0], h8[2][0], h8[3][0], pri)))
    out["engine_fp8_vs_oracle_f16"] = a16[0]
    out["engine_vs_oracle_fp8_mode"] = a8[0]
    out["engine_vs_oracle_fp8_mode"]["oracle_mode"] = f"fp8 forward mode, {len(fp8_layers)} E4M3 layers, the engine's calibrated scales"
    if len(a16) > 1:
        out["engine_fp8_vs_oracle_f16_all_frames"] = pool_accuracy(a16)
        out["engine_vs_oracle_fp8_mode_all_frames"] = pool_accuracy(a8)
    return out


def family_rooflines(prof):
    """Every kernel symbol of a step with its own roofline figures (what bound it, achieved, share of the step)."""
    _, by = dominant_kernel(prof)
    total = sum(d["ms"] for d in by.values())
    out = []
    for sym, d in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
        if d["flops"] <= 0 and d["bytes"] <= 0:
            continue
        tf, gbs = d["flops"] / (d["ms"] * 1e-3) / 1e12, d["bytes"] / (d["ms"] * 1e-3) / 1e9
        peak = MFMA_FP8_DENSE_PEAK_TFLOPS if sym.startswith("conv_igemm_fp8") else MFMA_F16_DENSE_PEAK_TFLOPS
        out.append(dict(kernel=sym, family=kernel_family(sym), launches=d["launches"], ms=round(d["ms"], 4), share_of_step=round(d["ms"] / total, 3),
                        tflops=round(tf, 1), frac_mfma=round(tf / peak, 3), gbs=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 3)))
    return out


def tflite_record(ya, invokes=120):
    """The reference's own model family (MobileNetV2-FPN YOLACT, uint8, 224 x 224: data/README.md:5-16) through the .tflite
    executor (yh_tfl_*): the reference's literal workload - one 640 x 480 camera frame = two tiles, src/yolact.rs:192-234 - as
    ms per classify(), plus launches per invoke and the share of CONV_2D launches on the int8 matrix pipes. The real
    FRC_model.tflite is absent (.MISSING_LARGE_BLOBS); the model is the 136-op stand-in with the log's op census
    (tests/tfl_models.mobilenetv2_yolact), serialised by tests/tfl_builder.py."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import tfl_builder as B
        import tfl_models as M
    except Exception as e:   # noqa: BLE001
        return dict(error=f"model builder unavailable: {e}"[:200])
    rng = np.random.default_rng(0)
    model = M.mobilenetv2_yolact(rng)
    eng = ya.TfliteEngine(bytes(B.serialize(model)))
    frame = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
    t = []
    for _ in range(invokes):
        f = frame.copy()
        t0 = time.perf_counter(); eng.classify_frame(f, 640, 480, ya.COMPAT_SANE); t.append(time.perf_counter() - t0)
    t = np.array(t[20:]) * 1e3
    rec = dict(workload="classify(640x480 packed u32 frame) through a 136-op MobileNetV2-FPN-YOLACT uint8 .tflite stand-in: resize, two 224x224 tiles in one batch-2 pass, "
                        "28x28 argmax post-process, x8 upsample, resize back - all on the device, host frame in and out (src/yolact.rs:192-234)",
               ops=len(model.ops), ms_per_frame_median=round(float(np.median(t)), 4), ms_per_frame_p99=round(float(np.percentile(t, 99)), 4),
               frames_per_s=round(1e3 / float(np.median(t)), 1), reference_published="~2 x 50 ms per frame on Pi 4 + Coral USB (data/README.md:12; out.log:429-430)")
    try:
        plan = eng.plan_summary()
        rec.update(plan)
    except Exception:   # noqa: BLE001 - older library without the summary call
        pass
    eng.close()
    return rec


VERBOSE = False


def progress(msg):
    """--verbose: phase markers on stderr (the JSON line stays the only thing on stdout)."""
    if VERBOSE:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def run_config(ya, torch, dist, rank, world, local_rank, batch, steps, warmup, seed, size, blob_dev_ptr, blob_nbytes, ring=4, backbone=50,
               precision="f16", tune=None, fp8_per_tensor=False):
    eng = ya.Engine(input_size=size, backbone=backbone, max_batch=batch, use_graph=True, device=local_rank,
                    precision=ya.PRECISION_FP8 if precision == "fp8" else ya.PRECISION_F16, tune=tune, fp8_per_tensor=fp8_per_tensor)
    eng.load_weights_device(blob_dev_ptr, blob_nbytes)
    g = torch.Generator(device=f"cuda:{local_rank}")
    start, _ = shard_frames(world * batch, world, rank)
    g.manual_seed(0x594F4C41 + start)   # counter-based per-shard seed
    bufs = [torch.randint(0, 256, (batch, size, size, 3), dtype=torch.uint8, device=f"cuda:{local_rank}", generator=g)
            for _ in range(ring)]
    torch.cuda.synchronize()
    if precision == "fp8":   # per-tensor activation scales from the bench frames themselves (outside the timed region)
        eng.set_input_device(bufs[0].data_ptr(), batch)
        eng.fp8_calibrate()
        eng.sync()

    def step(i):
        eng.set_input_device(bufs[i % ring].data_ptr(), batch)
        eng.evaluate()
    progress(f"batch {batch}: warmup")
    for i in range(warmup):
        step(i)
    eng.sync()
    progress(f"batch {batch}: timed steps")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist, dt, f"cuda:{local_rank}")
    if dist is not None:
        dist.barrier()
    progress(f"batch {batch}: timed steps done, per-launch profile")
    prof = eng.profile(with_tail=True, reps=3) if rank == 0 else None
    progress(f"batch {batch}: detections")
    ndet = sum(len(eng.detections(f, want_masks=False)[0]) for f in range(min(batch, 4))) if rank == 0 else 0
    flops = eng.flops_per_frame()
    aux = {}
    if rank == 0:
        host = bufs[0].cpu().numpy()
        aux["host_frame"] = host[:1]
        progress(f"batch {batch}: latency_stats")
        aux["latency"] = latency_stats(eng)
        progress(f"batch {batch}: pcie_inclusive")
        # frames in PINNED host memory (what a capture pipeline hands over: a true DMA) and, beside it, in pageable memory
        # (the runtime stages those through its own pinned buffers with the calling thread)
        pinned = bufs[0].cpu().pin_memory()
        aux["pcie_inclusive_fps"] = pcie_inclusive(eng, pinned.numpy())
        aux["pcie_inclusive_fps_pageable"] = pcie_inclusive(eng, host)
        del pinned
        if batch == 1:
            progress(f"batch {batch}: host_to_detections_latency")
            aux["host_to_detections_latency"] = host_to_detections_latency(eng, host)
        progress(f"batch {batch}: final frame-0 detections")
        eng.set_input_device(bufs[0].data_ptr(), batch); eng.evaluate(); eng.sync()
        aux["dets_frame0"] = eng.detections(0, want_masks=True)
        if batch <= 8:   # (configs[4]'s share: every frame's detections, for the pooled accuracy figures)
            aux["host_frames"] = host
            aux["dets_all"] = [aux["dets_frame0"]] + [eng.detections(f, want_masks=True) for f in range(1, batch)]
        acc_img = acceptance_frame(size)
        if acc_img is not None:
            eng.set_input(acc_img); eng.evaluate(); eng.sync()
            aux["accept"] = (acc_img, eng.detections(0, want_masks=True))
        aux["fp8_layers"] = eng.fp8_channel_scales() if precision == "fp8" else None   # (name, one activation scale per input channel)
    if dist is not None:
        dist.barrier()
    eng.close()
    progress(f"batch {batch}: engine closed")
    return dt, prof, flops, ndet, aux


def single_process_main(a, ya, torch, tune):
    """--single-process: the sharded path as ONE process drives it through the library's own group API (yh_group_*: one host
    worker thread + stream per device, contiguous frame blocks, one weight replication, no per-step collective) - what a
    host that is not torch (the reference's Rust process, src/main.rs:63-75) would run. Same timing contract: resident
    frames, W warm-up steps, exactly K timed steps between two syncs of every member, one JSON line."""
    ndev = torch.cuda.device_count()
    devices = list(range(a.gpus)) if ndev >= a.gpus else [0] * a.gpus
    g = ya.Group(devices, input_size=a.size, backbone=a.backbone, max_batch=a.batch, use_graph=True,
                 precision=ya.PRECISION_FP8 if a.precision == "fp8" else ya.PRECISION_F16, tune=tune)
    blob = g.members[0].generate_weights(a.seed)
    g.load_weights(blob)
    ring = 2
    bufs = []
    for i, d in enumerate(devices):
        gen = torch.Generator(device=f"cuda:{d}")
        gen.manual_seed(0x594F4C41 + i * a.batch)
        bufs.append([torch.randint(0, 256, (a.batch, a.size, a.size, 3), dtype=torch.uint8, device=f"cuda:{d}", generator=gen) for _ in range(ring)])
    for d in set(devices):
        torch.cuda.synchronize(d)
    counts = [a.batch] * a.gpus
    if a.precision == "fp8":
        g.members[0].set_input_device(bufs[0][0].data_ptr(), a.batch)
        g.fp8_calibrate()

    def step(k):
        g.evaluate_device([b[k % ring].data_ptr() for b in bufs], counts)
    for k in range(a.warmup):
        step(k)
    g.sync()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    g.sync()
    dt = time.perf_counter() - t0
    ndet = sum(len(g.detections(f, want_masks=False)[0]) for f in range(min(4, a.gpus * a.batch)))
    e0 = g.members[0]
    e0.n = a.batch
    prof = e0.profile(with_tail=True, reps=3)
    flops = e0.flops_per_frame()
    fps = a.gpus * a.batch * a.steps / dt
    line = {
        "metric": f"frames/sec YOLACT-{a.size} (ResNet-{a.backbone}-FPN, 32 prototypes) " + ("fp16" if a.precision == "f16" else "fp8") + ", forward + detection tail",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": f"YOLACT-{a.size} R{a.backbone}-FPN batch={a.batch} per GPU, {a.size}x{a.size}x3 uint8 frames resident in HBM, hipGraph steady state, "
                               f"ONE process: yh_group over devices {devices}" + (" (members share device 0: a rehearsal, not a measurement)" if ndev < a.gpus else ""),
                   "batch_per_gpu": a.batch, "global_batch": a.batch * a.gpus, "input": [a.size, a.size, 3],
                   "weights": f"seeded synthetic (seed {a.seed}), BN folded", "weights_replication": g.weights_replication(),
                   "detections_first_frames": ndet, "host": "single process, one worker thread per device (yh_group_evaluate_device)"},
        "net_tflops": round(fps * flops / 1e12, 2), "gflop_per_frame": round(flops / 1e9, 2),
        "roofline": roofline_of(prof, a.batch),
    }
    g.close()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (64: configs[2]/[3]; 1: configs[1])")
    ap.add_argument("--size", type=int, default=550)
    ap.add_argument("--backbone", type=int, default=50, choices=(50, 101), help="50: YOLACT-550 R50 (default); 101 with --size 700: configs[4] geometry in f16")
    ap.add_argument("--precision", default="f16", choices=("f16", "fp8"),
                    help="fp8: E4M3 operands on the block-scaled fp8 MFMA for the K-heavy 3x3 layers (configs[4]: --backbone 101 --size 700 --precision fp8)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the cpu_baseline sample")
    ap.add_argument("--no-batch1", action="store_true")
    ap.add_argument("--no-configs4", action="store_true", help="skip the configs[4] sub-record (YOLACT-700 R101 fp8 at this rank's share of 64 frames over 8 GPUs)")
    ap.add_argument("--configs4-batch", type=int, default=8, help="frames per GPU of the configs[4] sub-record (64 frames / 8 GPUs)")
    ap.add_argument("--no-tflite", action="store_true", help="skip the tflite sub-record (the reference's own model family through the uint8 .tflite executor)")
    ap.add_argument("--fp8-per-tensor", action="store_true", help="fp8 engines: one activation scale per tensor (round 3's scheme) instead of one per input channel - A/B of the accuracy figures")
    ap.add_argument("--verbose", action="store_true", help="phase markers on stderr")
    ap.add_argument("--tune", default="", help="comma-separated yh_tuning fields for A/B measurements, e.g. tailfork=1,k1tile=0 (default: none)")
    ap.add_argument("--torch-broadcast", action="store_true",
                    help="replicate the weights with torch.distributed.broadcast instead of the library's own RCCL broadcast (yh_rank_broadcast_weights)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive --gpus N devices from ONE process through the library's group API (yh_group_*) instead of one rank per GPU; "
                         "with fewer visible devices than N the members share device 0 (a rehearsal)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="control-flow rehearsal of the N > 1 path on a single-GPU box: every rank uses cuda:0 and the "
                         "weight broadcast goes over gloo (RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--rccl-library", default="",
                    help="rehearsal only, with --rehearse-on-one-gpu: take the library's own RCCL call (yh_rank_broadcast_weights) against THIS "
                         "librccl file - the stand-in of tests/rccl_standin/, which lets two ranks share a device (real RCCL refuses that); "
                         "torch.distributed stays on gloo. Not a measurement")
    a = ap.parse_args()
    global VERBOSE
    VERBOSE = a.verbose

    import numpy as np
    import torch
    import yolact_amd as ya
    tune = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)} or None
    if a.single_process:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP kernel library has no CPU fallback")
        return single_process_main(a, ya, torch, tune)
    rank, world, local_rank = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if a.rehearse_on_one_gpu:
        local_rank = 0
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernel library has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    # weights: generated once on rank 0, replicated with ONE RCCL broadcast over xGMI
    make_src = lambda: ya.Engine(input_size=a.size, backbone=a.backbone, max_batch=1, use_graph=False, device=local_rank)   # noqa: E731
    src0 = make_src()
    nbytes = src0.weights_nbytes()
    if a.rccl_library:
        if not a.rehearse_on_one_gpu:
            raise SystemExit("--rccl-library is a rehearsal switch: use it with --rehearse-on-one-gpu")
        rc = ya.load_library().yh_debug_rccl_library(a.rccl_library.encode())
        if rc != 0:
            raise SystemExit(f"yh_debug_rccl_library: {rc}")
    use_library = not a.torch_broadcast and (not a.rehearse_on_one_gpu or bool(a.rccl_library))
    how, src = replicate_weights(ya, torch, dist, rank, world, local_rank, src0, a.seed, use_library=use_library, make_engine=make_src)
    # (src is src0 unless this rank's library call is stuck - then src0 is never touched again, not even closed: replicate_weights)
    blob_ptr = src.weights_device_ptr()

    dt, prof, flops, ndet, aux = run_config(ya, torch, dist, rank, world, local_rank, a.batch, a.steps, a.warmup,
                                              a.seed, a.size, blob_ptr, nbytes, backbone=a.backbone, precision=a.precision, tune=tune, fp8_per_tensor=a.fp8_per_tensor)
    extra = {}
    if a.batch != 1 and not a.no_batch1:   # configs[1]: batch=1 latency case, same run
        s1 = max(a.steps * 4, 40)
        dt1, prof1, _, _, aux1 = run_config(ya, torch, dist, rank, world, local_rank, 1, s1, max(a.warmup, 5), a.seed, a.size,
                                         blob_ptr, nbytes, backbone=a.backbone, precision=a.precision, tune=tune, fp8_per_tensor=a.fp8_per_tensor)
        if rank == 0:
            extra["batch1"] = dict(workload=f"YOLACT-{a.size} R{a.backbone}-FPN batch=1 {a.precision} {a.size}x{a.size}x3 per GPU (configs[1])",
                                   value=round(world * s1 / dt1, 2), unit="frames/s", ms_per_step=round(dt1 / s1 * 1e3, 4),
                                   net_tflops=round(world * s1 / dt1 * flops / 1e12, 2), roofline=roofline_of(prof1, 1),
                                   kernel_families=family_rooflines(prof1), launches_per_step=len(prof1),
                                   latency=aux1["latency"], pcie_inclusive_fps=aux1["pcie_inclusive_fps"], pcie_inclusive_fps_pageable=aux1["pcie_inclusive_fps_pageable"],
                                   host_to_detections_latency=aux1.get("host_to_detections_latency"),
                                   weights_replication=how + " (the headline record's blob: the batch-1 engine loads it device to device)")
    if a.batch != 1 and not a.no_configs4 and not (a.backbone == 101 and a.precision == "fp8"):
        # BASELINE.json configs[4]: YOLACT-700 ResNet-101, fp8 operands on the fp8 MFMA, batch 64 across 8 GPUs = 8 frames per
        # GPU - this rank's share, timed in the same run (same contract: resident frames, graph replay, max over ranks)
        make_src4 = lambda: ya.Engine(input_size=700, backbone=101, max_batch=1, use_graph=False, device=local_rank)   # noqa: E731
        # (the library's RCCL path again only if it worked for the headline engine's weights; should THIS call stick on a rank, that rank
        # takes its weights in a fresh engine from the factory and every rank still meets in the fallback broadcast: ADVICE r4)
        how4, src4 = replicate_weights(ya, torch, dist, rank, world, local_rank, make_src4(), a.seed,
                                       use_library=use_library and (world == 1 or how.startswith("yh_rank_broadcast_weights")), make_engine=make_src4)
        s4 = max(a.steps, 20)
        dt4, prof4, flops4, _, aux4 = run_config(ya, torch, dist, rank, world, local_rank, a.configs4_batch, s4, max(a.warmup, 3), a.seed, 700,
                                                  src4.weights_device_ptr(), src4.weights_nbytes(), backbone=101, precision="fp8", tune=tune, fp8_per_tensor=a.fp8_per_tensor)
        src4.close()
        if rank == 0:
            fps4 = world * a.configs4_batch * s4 / dt4
            rec = dict(workload=f"YOLACT-700 R101-FPN fp8 (E4M3 operands on the {len(aux4['fp8_layers'])} K-heavy 3x3 launches, f16 elsewhere), "
                                f"batch={a.configs4_batch} per GPU (configs[4]: 64 frames over 8 GPUs), 700x700x3 uint8 frames resident in HBM, hipGraph steady state",
                       value=round(fps4, 2), unit="frames/s", n_gpus=world, steps=s4, ms_per_step=round(dt4 / s4 * 1e3, 4), dtype="fp8",
                       net_tflops=round(fps4 * flops4 / 1e12, 2), gflop_per_frame=round(flops4 / 1e9, 2), weights_replication=how4,
                       roofline=roofline_of(prof4, a.configs4_batch), kernel_families=family_rooflines(prof4), latency=aux4["latency"],
                       pcie_inclusive_fps=aux4["pcie_inclusive_fps"], pcie_inclusive_fps_pageable=aux4["pcie_inclusive_fps_pageable"])
            fp8f = [k for k in rec["kernel_families"] if k["kernel"].startswith("conv_igemm_fp8")]
            if fp8f:   # the fp8 launches as one family, against the 5 PFLOP/s dense fp8 peak
                ms = sum(k["ms"] for k in fp8f); fl = sum(k["tflops"] * k["ms"] for k in fp8f)
                rec["fp8_launches"] = dict(launches=sum(k["launches"] for k in fp8f), ms=round(ms, 4), tflops=round(fl / ms, 1),
                                           peak=MFMA_FP8_DENSE_PEAK_TFLOPS, frac=round(fl / ms / MFMA_FP8_DENSE_PEAK_TFLOPS, 4))
            rec["fp8_activation_scales"] = "one per tensor (yh_config.fp8_per_tensor = 1)" if a.fp8_per_tensor else "one per input channel, folded into the weights' K axis (round 4)"
            if world == 1 and not a.no_cpu_baseline:
                if "dets_all" in aux4:
                    rec["accuracy"] = fp8_vs_oracles(a.seed, aux4["host_frames"], aux4["dets_all"], aux4["fp8_layers"], 101, usable_cores())
                else:
                    rec["accuracy"] = fp8_vs_oracles(a.seed, aux4["host_frame"], [aux4["dets_frame0"]], aux4["fp8_layers"], 101, usable_cores())
                if aux4.get("accept") is not None:   # ... and on the reference's own test image at 700 x 700
                    img, dets = aux4["accept"]
                    rec["accuracy_frc_balls"] = fp8_vs_oracles(a.seed, img, [dets], aux4["fp8_layers"], 101, usable_cores())
                    rec["accuracy_frc_balls"]["frame"] = "tests/golden/frc_balls.png resized to 700x700 (bilinear)"
            extra["configs4"] = rec
    if world == 1 and rank == 0 and not a.no_tflite:
        extra["tflite"] = tflite_record(ya)
    if not _LIBRARY_CALL_STUCK:
        src.close()
    if rank != 0:
        leave_if_stuck()          # (every stuck rank, not only rank 0: right after its last collective)
        if dist is not None:
            dist.destroy_process_group()
        return
    fps = world * a.batch * a.steps / dt
    line = {
        "metric": f"frames/sec YOLACT-{a.size} (ResNet-{a.backbone}-FPN, 32 prototypes) " + ("fp16" if a.precision == "f16" else "fp8 (E4M3 operands on the K-heavy 3x3 layers, f16 elsewhere)") + ", forward + detection tail",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": f"YOLACT-{a.size} R{a.backbone}-FPN batch={a.batch} per GPU, {a.size}x{a.size}x3 uint8 frames resident in HBM, "
                               f"hipGraph steady state, frames sharded over {world} GPU(s), " + ("one GPU: no collective" if world == 1 else f"weights replicated once: {how}"),
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "input": [a.size, a.size, 3],
                   "weights": f"seeded synthetic (seed {a.seed}), BN folded", "weights_replication": how,
                   "detections_first_frames": ndet},
        "net_tflops": round(fps * flops / 1e12, 2), "gflop_per_frame": round(flops / 1e9, 2),
        "net_frac_of_mfma_peak": round(fps * flops / 1e12 / (MFMA_F16_DENSE_PEAK_TFLOPS * world), 4),   # against the f16 peak in either precision (mixed-precision step)
        "roofline": roofline_of(prof, a.batch),
        "roofline_second": roofline_of(prof, a.batch, rank=1),
        "kernel_families": family_rooflines(prof),
        "launches_per_step": len(prof),
        "latency": aux["latency"],
        "pcie_inclusive_fps": aux["pcie_inclusive_fps"],
        "pcie_inclusive_fps_pageable": aux["pcie_inclusive_fps_pageable"],
        "pcie_inclusive_note": "host frames in pinned / pageable memory through yh_set_input_u8, copy of batch k+1 under step k (two input buffers + copy stream)",
    }
    line.update(extra)
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(a.seed, aux["host_frame"], budget_s=a.cpu_budget, eng_out=aux["dets_frame0"], backbone=a.backbone,
                                            fp8_layers=aux.get("fp8_layers"), accept=aux.get("accept"))
    print(json.dumps(line), flush=True)
    leave_if_stuck()             # a helper thread is still inside the library's RCCL call: no teardown that could wait for it
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
