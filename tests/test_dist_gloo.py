"""The N > 1 path on CPU: world_size 2 over gloo (127.0.0.1). Frames shard with no data-path
collective; the one collective is the weight-blob broadcast from rank 0; step time is the MAX
over ranks. Uses the same helpers bench.py runs on RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nbytes, out_dir):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = torch.zeros(nbytes, dtype=torch.uint8)
        if rank == 0:
            rng = np.random.default_rng(7)
            blob.copy_(torch.from_numpy(rng.integers(0, 256, nbytes, dtype=np.uint8)))
        bench.broadcast_weights(dist, blob)
        start, count = bench.shard_frames(129, world, rank)
        t = bench.max_over_ranks(dist, 1.0 + rank, "cpu")
        # the branch every rank takes for the weight replication is agreed by one MIN all-reduce
        agree = [bench.all_ranks_ok(dist, True, "cpu"), bench.all_ranks_ok(dist, rank != 1, "cpu"), bench.all_ranks_ok(dist, False, "cpu")]
        np.save(os.path.join(out_dir, f"a{rank}.npy"), np.array(agree))
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([int(blob.sum()), start, count, t], np.float64))
        np.save(os.path.join(out_dir, f"b{rank}.npy"), blob.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_shard_and_timing(tmp_path):
    world, nbytes = 2, 1 << 16
    mp.spawn(_worker, args=(world, _free_port(), nbytes, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npy") for k in range(world)]
    b = [np.load(tmp_path / f"b{k}.npy") for k in range(world)]
    assert np.array_equal(b[0], b[1]) and b[0].any()            # rank 1 received rank 0's blob
    assert (r[0][1], r[0][2]) == (0, 65) and (r[1][1], r[1][2]) == (65, 64)   # contiguous blocks of 129 frames
    assert r[0][3] == r[1][3] == 2.0                            # MAX over ranks
    for k in range(world):                                      # one rank's failure sends BOTH ranks to the fallback
        assert list(np.load(tmp_path / f"a{k}.npy")) == [True, False, False]


def test_shard_frames_partitions_exactly():
    import bench
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_frames(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_roofline_helper_picks_dominant_kernel():
    import bench
    prof = [dict(name="conv_a:l1", ms=2.0, flops=4e12, bytes=1e9), dict(name="conv_a:l2", ms=1.0, flops=2e12, bytes=1e9),
            dict(name="pool:p", ms=0.5, flops=0.0, bytes=2e9)]
    r = bench.roofline_of(prof)
    assert r["kernel"] == "conv_a" and r["bound"] == "mfma" and r["launches"] == 2
    assert abs(r["achieved"] - 2000.0) < 1e-6 and abs(r["frac"] - 0.8) < 1e-9


def test_accuracy_helper_matches_by_class_and_prior():
    """bench.accuracy_vs_oracle (the engine-vs-oracle report in the bench line): detections are matched
    by (class, prior); mask_iou_matched is pooled over matched pairs only, mask_iou_all over every detection."""
    import bench
    m = np.zeros((2, 4, 4), np.uint8); m[0, :2] = 1; m[1, 1:3, 1:3] = 1
    eng = ([dict(class_id=3, prior=7), dict(class_id=5, prior=9)], m)
    om = m.copy(); om[0, 2, 0] = 1                                     # one extra pixel in the first mask
    orc = ([dict(class_id=3, prior=7), dict(class_id=5, prior=9), dict(class_id=1, prior=2)], np.concatenate([om, np.ones((1, 4, 4), np.uint8)]))
    r = bench.accuracy_vs_oracle(eng, orc)
    assert r["oracle_dets"] == 3 and r["engine_dets"] == 2 and r["matched_class_and_prior"] == 2
    assert abs(r["mask_iou_matched"] - (8 + 4) / (9 + 4)) < 1e-4
    # the figure that counts: per-class unions over ALL detections - the oracle-only detection (class 1, 16 pixels)
    # adds its whole mask to the union and nothing to the intersection
    assert r["unmatched_oracle"] == 1 and r["unmatched_engine"] == 0
    assert abs(r["mask_iou_all"] - (8 + 4 + 0) / (9 + 4 + 16)) < 1e-4
    same = bench.accuracy_vs_oracle(eng, eng)
    assert same["mask_iou_all"] == 1.0 and same["unmatched_oracle"] == 0
    e = bench.accuracy_vs_oracle(([], m[:0]), ([], m[:0]))
    assert e["mask_iou_matched"] is None and e["mask_iou_all"] is None
    # full lists whose last places swap at the cut (scores equal to the 4th decimal): the *_above_cut pair leaves the tie out
    m3 = np.concatenate([m, np.ones((1, 4, 4), np.uint8)])
    a = ([dict(class_id=3, prior=7, score=0.9), dict(class_id=5, prior=9, score=0.5), dict(class_id=8, prior=1, score=0.1001)], m3)
    b = ([dict(class_id=3, prior=7, score=0.9), dict(class_id=5, prior=9, score=0.5), dict(class_id=9, prior=2, score=0.1002)], m3)
    r = bench.accuracy_vs_oracle(a, b)
    assert r["unmatched_oracle"] == 1 and r["mask_iou_all"] < 0.5 and r["mask_iou_above_cut"] == 1.0 and r["unmatched_above_cut"] == 0 and r["dets_above_cut"] == 2
    b2 = ([dict(class_id=3, prior=7, score=0.9), dict(class_id=6, prior=9, score=0.5), dict(class_id=9, prior=2, score=0.1002)], m3)
    assert bench.accuracy_vs_oracle(a, b2)["unmatched_above_cut"] == 2          # a real disagreement stays one


def test_measured_traffic_reads_the_committed_pmc_file():
    import bench
    t, src = bench.measured_traffic("conv_igemm_f16<256,256,2,4,0,2,mfma16>", 64)
    assert src and src.startswith("profiles/") and t > 1e8
    assert bench.measured_traffic("no_such_kernel", 64) == (None, None)
    # a tile FAMILY ({symbol: launches per step}): launch-weighted mean over its symbols; None unless every member was measured
    fam = {"conv_igemm_f16<256,256,2,4,0,2,mfma16>": 5, "conv_igemm_f16<256,256,2,4,0,2,mfma16>[+1x1]": 1, "conv_igemm_f16<256,256,2,4,0,2,mfma16>[ml]": 2}
    tf, _ = bench.measured_traffic(fam, 64)
    assert tf is not None and t < tf < 8e8
    assert bench.measured_traffic(dict(fam, no_such_kernel=1), 64) == (None, None)


def test_roofline_groups_by_tile_family_and_records_say_how_weights_were_replicated():
    """the headline roofline describes the step's largest tile FAMILY (the 256 x 256 tile is three kernel
    symbols), and kernel_family() strips exactly the template flags."""
    import bench
    assert bench.kernel_family("conv_igemm_f16<256,256,2,4,0,2,mfma16>[+1x1]") == "conv_igemm_f16<256,256,2,4,0,2>"
    assert bench.kernel_family("conv_igemm_f16<256,256,2,4,0,2,mfma16>[ml]") == bench.kernel_family("conv_igemm_f16<256,256,2,4,0,2,mfma16>")
    # round 5: a single-stage streaming tile is two populations with two bounds - its 3x3 launches are a family of their own
    assert bench.kernel_family("conv_igemm_f16<128,128,2,2,0,1>[3x3]") == "conv_igemm_f16<128,128,2,2,0,1>[3x3]" != bench.kernel_family("conv_igemm_f16<128,128,2,2,0,1>")
    assert bench.kernel_family("conv_igemm_f16<128,128,2,2,0,1,dual>") == bench.kernel_family("conv_igemm_f16<128,128,2,2,0,1,resup>") == "conv_igemm_f16<128,128,2,2,0,1>"
    assert bench.kernel_family("conv_igemm_f16<128,128,2,2,0,1>[ml][3x3]") == "conv_igemm_f16<128,128,2,2,0,1>[3x3]"
    assert bench.kernel_family("conv_igemm_f16<128,128,2,2,0,2>[3x3]") == "conv_igemm_f16<128,128,2,2,0,2>"   # (pipelined tiles: one bound)
    assert bench.kernel_family("bneck_chain_f16<64,256,next>") == "bneck_chain_f16<64,256,next>" and bench.kernel_family("det_masks") == "det_masks"
    prof = [dict(name="conv_igemm_f16<128,128,2,2,0,1>:a", ms=2.0, flops=1e12, bytes=6e9),
            dict(name="conv_igemm_f16<256,256,2,4,0,2,mfma16>:p", ms=1.5, flops=1.5e12, bytes=1e9),
            dict(name="conv_igemm_f16<256,256,2,4,0,2,mfma16>[+1x1]:q", ms=1.2, flops=1.4e12, bytes=0.6e9),
            dict(name="conv_igemm_f16<256,256,2,4,0,2,mfma16>[ml]:h", ms=0.9, flops=0.9e12, bytes=0.7e9)]
    r = bench.roofline_of(prof, 64)
    assert r["kernel"] == "conv_igemm_f16<256,256,2,4,0,2>" and r["launches"] == 3 and r["bound"] == "mfma" and len(r["symbols"]) == 3
    assert abs(r["share_of_step"] - 3.6 / 5.6) < 1e-3 and abs(r["achieved"] - 3.8e12 / 3.6e-3 / 1e12) < 0.5
    # the two largest families, each with ONE bound: a tie that flips by box changes their order, not what is reported
    prof2 = prof + [dict(name="conv_igemm_f16<128,128,2,2,0,1>[3x3]:b", ms=1.0, flops=0.9e12, bytes=0.3e9)]
    r1, r2 = bench.roofline_of(prof2, 64), bench.roofline_of(prof2, 64, rank=1)
    assert r1["kernel"] == "conv_igemm_f16<256,256,2,4,0,2>" and r2["kernel"] == "conv_igemm_f16<128,128,2,2,0,1>" and r2["bound"] == "hbm" and r2["launches"] == 1
    r3 = bench.roofline_of(prof2, 64, rank=2)
    assert r3["kernel"] == "conv_igemm_f16<128,128,2,2,0,1>[3x3]" and r3["bound"] == "mfma" and bench.roofline_of(prof2, 64, rank=9) is None
    fams = bench.family_rooflines(prof)
    assert [f["kernel"] for f in fams][0] == "conv_igemm_f16<128,128,2,2,0,1>" and all("family" in f for f in fams)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("weights_replication=") >= 2 and '"weights_replication": how' in src   # batch1 and configs4 carry their own field beside the headline's


class _FakeEngine:
    """Stands in for yolact_amd.Engine in bench.replicate_weights: rank 1's library broadcast never returns in time."""
    def __init__(self, rank, nbytes, hang_s):
        self.rank, self.nbytes, self.hang_s, self.loaded = rank, nbytes, hang_s, None

    def generate_weights(self, seed):
        return (np.arange(self.nbytes, dtype=np.int64) * 7 + seed).astype(np.uint8)

    def load_weights(self, blob):
        self.loaded = np.array(blob, copy=True)

    def weights_nbytes(self):
        return self.nbytes

    def rank_broadcast_weights(self, ident, rank, world, root):
        import time
        if rank == 1:
            time.sleep(self.hang_s)          # a peer blocked inside ncclCommInitRank
        raise RuntimeError("no RCCL in this test")

    def load_weights_device(self, ptr, n):
        import ctypes
        self.loaded = np.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=np.uint8).copy()


class _FakeYa:
    @staticmethod
    def rccl_unique_id():
        return b"\0" * 128


def _worker_deadline(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench.LIBRARY_BROADCAST_DEADLINE_S = 1
        eng = _FakeEngine(rank, 4096, hang_s=20)
        how, holder = bench.replicate_weights(_FakeYa, torch, dist, rank, world, 0, eng, 3, use_library=True, device="cpu",
                                              make_engine=lambda: _FakeEngine(rank, 4096, hang_s=20))
        np.save(os.path.join(out_dir, f"w{rank}.npy"), holder.loaded)
        with open(os.path.join(out_dir, f"how{rank}.txt"), "w") as f:
            # (a stuck rank's first engine belongs to the blocked thread: the weights went into a FRESH one and the old one was not touched)
            f.write(how + "\n" + str(bench._LIBRARY_CALL_STUCK) + "\n" + str(holder is eng) + "\n" + str(eng.loaded is None))
    finally:
        dist.destroy_process_group()
    os._exit(0)   # (rank 1's helper thread is still asleep in the fake library call: what bench.main does in that case)


def test_library_broadcast_that_hangs_on_one_rank_falls_back_within_its_deadline(tmp_path):
    """bench.replicate_weights runs the library's RCCL broadcast under a deadline: here rank 0's call fails at once and rank 1's
    blocks (the hang mode INTEGRATION.md §4 describes); both ranks must agree on the torch.distributed fallback within seconds,
    end up with rank 0's weights, and the blocked rank must know that it may not tear down normally."""
    import time
    t0 = time.time()
    mp.spawn(_worker_deadline, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert time.time() - t0 < 15
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1) and w0[5] == (5 * 7 + 3) % 256
    how0, how1 = (open(tmp_path / f"how{r}.txt").read().split("\n") for r in (0, 1))
    assert how0[0].startswith("torch.distributed.broadcast") and how1[0].startswith("torch.distributed.broadcast")
    assert "failed on rank 0" in how0[0] and "did not return within 1 s on rank 1" in how1[0]
    assert how0[1] == "False" and how1[1] == "True"
    assert how0[2] == "True"                              # rank 0's call failed cleanly: its engine keeps the weights
    assert how1[2] == "False" and how1[3] == "True"       # rank 1's call is stuck: a fresh engine holds them, the first one was never touched again
