# tune.ablate on the MFMA-bound 256 x 256 launches (timing only, results are garbage): 0 shipped, 1 no activation stream, 2 no weight
# stream (zero-record descriptors: the loads issue, nothing moves - but MFMAs on zeros also run at a higher clock), 3 both, 4 no
# loader instruction at all. (The round-3 study also had a bit 3: the activation tile of the taps s = 1, 2 not fetched, on stale
# non-zero data - what sharing one fetch between the three horizontal taps would save; its numbers are in DESIGN.md §12, the built
# form in tools/study/xs_shared_activation.patch.)
for a in 0 1 2 3 4; do
  echo "ablate=$a"; python tools/profile_table.py --batch 64 --tune ablate=$a | grep -E "^YOLACT|:proto[0-3]|:p3 |head_t/rounds|head_out/ch0|\[3x3\]:l3b1_b|\[3x3\]:l4b1_b|\[3x3\]:l2b0_b"
done
