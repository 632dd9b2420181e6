/*
 * yolact_hip_debug.h - measurement, study and test surface of libyolact_hip.so: the named fields of yh_tuning, yh_set_tuning /
 * yh_get_tuning, the profiling hooks bench.py uses, the single-op entry points the parity tests call, the yh_debug_* hooks.
 * NOT part of the drop-in boundary (include/yolact_hip.h): a host that replaces src/yolact.rs needs none of it, and the C++ host
 * mirror (tiny-object-detection_amd/host/) compiles against yolact_hip.h alone. Include THIS header instead of yolact_hip.h to get
 * both; the struct it defines has the layout of the public header's opaque `yh_tuning { int32_t knob[32]; }`.
 */
#ifndef YOLACT_HIP_DEBUG_H
#define YOLACT_HIP_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_TUNING_DEFINED
/* Measurement and test knobs of ONE handle. Every field: -1 = the library's default.
 * None of them changes results beyond the stated f16 tolerance (most are bit-equivalent); they exist so that
 * every launch-plan decision can be re-measured A/B and so that tests can reach every plan with small tensors.
 * There are no process-global switches: the library reads no environment variable. */
typedef struct yh_tuning {
    int32_t plan_cus;        /* CU count the launch plans assume (default: the device's multiProcessorCount) */
    int32_t mfma16;          /* 256x256 tile on v_mfma_f32_16x16x32_f16 (1, default) or 32x32x16 (0) */
    int32_t t128x256_m16;    /* 128x256 tile: 2-stage 16x16x32 form on stride-1 layers (1) or the 3-stage ring (0) */
    int32_t small16;         /* 128x128 tiles on 16x16x32 everywhere (0) */
    int32_t bigk;            /* K from which Cout >= 128 layers use the 8-wave tiles (256); creation time only */
    int32_t tailsplit;       /* two-phase launches against wave quantisation (1) */
    int32_t chsplit;         /* 256 + 128 channel split of the 384-channel head (1) */
    int32_t k1tile;          /* single-stage streaming tiles (four workgroups per CU): 0 off, 1 1x1 layers on the 128x128 form,
                              * 2 + 64-channel 1x1, 3 + 64-channel 3x3, 4 + 128-channel 3x3, 5 / 6 + 3x3 layers with few big tiles and
                              * the head's 128-channel remainder (6) */
    int32_t k1_maxk;         /* ... for 1x1 layers with K <= this (1024) */
    int32_t splitk_minsteps; /* K-steps from which few-tile launches split K (12) */
    int32_t t64;             /* 64x64 tiles for latency-bound launches: 0 off, 1 never split K, 2 split K (2) */
    int32_t t64_maxb;        /* ... when at most this many 128x128 tiles (256) */
    int32_t t64_minsteps;    /* ... split K from this many K-steps (24) */
    int32_t stemfuse;        /* fused stem + max pool (1); creation time only */
    int32_t prefuse;         /* preprocessing inside the stem's patch loader (1); creation time only */
    int32_t headmerge;       /* the shared head as one multi-level launch per conv (1); creation time only */
    int32_t upfuse;          /* FPN top-down upsamples evaluated in the lateral conv's epilogue instead of a kernel and a
                              * tensor of their own (1); creation time only */
    int32_t k1_generic;      /* the generic softmax/candidate kernel also for 81 classes (0) */
    int32_t ablate;          /* timing only (results are garbage): bit 0 / bit 1 drop the activation / weight stream (zero-record
                              * descriptors: the loads issue, nothing moves), bit 2 issues no loader instruction at all */
    int32_t op_tile;         /* single-op entry points: force this ConvTile id (-1: the engine's choice) */
    int32_t op_kslices;      /* single-op entry points: force a split-K with this many slices */
    int32_t tfl_dot;         /* TFLite path (yh_tfl handles only), CONV_2D: 0 one lane per output element, 1 the v_dot4 kernel (Ci % 4 == 0),
                              * 2 also the int8 MFMA kernel on 64 x 64 LDS tiles where Ci % 64 == 0, 3 (default) the int8 MFMA kernel fed from
                              * registers - one wave per 16 x 16 tile of v_mfma_i32_16x16x64_i8, no LDS, a ring of 1-18 k-steps of loads in
                              * flight - where Ci % 4 == 0 (large long-K layers keep the LDS tiles); all four give the same bytes */
    int32_t tfl_graph;       /* TFLite path: 0 eager launches (default, and the faster form: 0.33 vs 0.37 ms per invoke of the 136-op
                              * model), 1 hipGraph replay of the plan (captured with a second, one-node branch: DESIGN.md §8) */
    int32_t tailfork;        /* the detection tail's K1-K3 on a second stream underneath the protonet (1); 0 keeps them on
                              * the main stream */
    int32_t dsfuse;          /* a stage's projection shortcut evaluated inside the block's last 1x1 conv (two-source K, 1);
                              * creation time only */
    int32_t headfork_maxb;   /* batches up to this size run the prediction head (and the tail's K1-K3) on the second stream
                              * beside the protonet (default: every batch size); 0: never (the tail's K1-K3 alone fork) */
    int32_t protofuse;       /* the 1x1 conv that makes the 32 prototypes evaluated in the epilogue of the 3x3 conv in front of it
                              * wherever that one runs as single 256 x 256-tile launches (1); creation time only */
    int32_t k1_min1;         /* streaming tiles: launches of at least this many tiles per CU, in QUARTERS (8 = 2 per CU), for the 1x1
                              * layers and the 64- / 128-channel 3x3 layers */
    int32_t k1_min3;         /* ... and (10 = 2.5 per CU) for the 3x3 layers with few big tiles and the head's remainder */
    int32_t chain;           /* identity bottleneck blocks of layers 1-2 as ONE launch each: 3x3 conv + 1x1 expand conv with the residual
                              * add + the next block's 1x1 reduce conv (csrc/bneck.hip; bit-identical to the separate launches). Bit 0
                              * on; bit 4 also fuse launches that only fill 64-pixel tiles (small batches: 3 % faster at batch 1-8).
                              * Default 17 (bits 0 and 4). Test switches: bit 1 layer 3's expand + next-reduce launch (bneck_xn_f16)
                              * wherever it is eligible instead of only inside its window (64-pixel tiles filling more than half a round
                              * of workgroups and at most one), bit 5 layer 1 only, bit 6 without the fused form of layer 1's FIRST block
                              * (projection shortcut), bit 7 without layer 3's launch. Bits 2-3 (round 3's persistent grid and phase
                              * stagger) are retired and ignored */
    int32_t tfl_fuse;        /* TFLite path: 1 (default) element-wise operators (QUANTIZE / RELU / RELU6 / TANH / ADD), PAD and contiguous
                              * CONCATENATION parts folded into the launch of the convolution / resize that produces their operand -
                              * same bytes, fewer launches; 0: one launch per operator, every tensor materialised (the checker) */
    int32_t tfl_group;       /* TFLite path: 1 (default) independent register-fed convolutions of one kernel form at one depth of the plan's
                              * graph as ONE launch (the prediction head's convolutions over the pyramid levels: 23 launches become 3), the
                              * plan in depth order; 0: one launch per convolution in file order. Same bytes */
} yh_tuning;

#ifdef __cplusplus
}
#endif

typedef char yh_tuning_has_the_public_headers_size[sizeof(yh_tuning) == 32 * sizeof(int32_t) ? 1 : -1];

#include "yolact_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces the run-time tuning fields of a live handle (captured graphs are dropped and re-captured on the next
 * call); YH_ESTATE if a creation-time field (bigk, stemfuse, prefuse, headmerge, upfuse, dsfuse, protofuse) differs from
 * the handle's. */
int yh_set_tuning(yh_engine* h, const yh_tuning* tune);
/* The handle's tuning with every default resolved (plan_cus = the CU count the plans really use, ...). The tfl_* fields belong to
 * yh_tfl handles (yh_tfl_create_tuned) and stay -1: an engine handle does not carry them. */
int yh_get_tuning(const yh_engine* h, yh_tuning* out);

/* ---- measurement hooks (bench.py; not part of the reference surface) ------------------------ */

/* Number of kernel launches in one forward(+tail if with_tail) for the current batch. */
int yh_profile_launch_count(const yh_engine* h, int32_t with_tail);
/* Runs `reps` forwards with a hipEvent pair around every launch on the handle's own stream and
 * writes, per launch: mean milliseconds, algorithmic FLOPs and algorithmic bytes, and a static
 * name ("kernel_symbol:layer"). Arrays must hold yh_profile_launch_count entries. */
int yh_profile_run(yh_engine* h, int32_t with_tail, int32_t reps, float* ms, double* flops,
                   double* bytes, const char** names);
/* Times `steps` back-to-back yh_evaluate (or yh_invoke if !with_tail) calls with one hipEvent
 * pair on the handle's stream; returns total milliseconds in *ms_total. */
int yh_time_steps(yh_engine* h, int32_t with_tail, int32_t steps, float* ms_total);
/* Algorithmic conv FLOPs of one frame for this architecture (2 x MAC, convs only). */
double yh_flops_per_frame(const yh_engine* h);

/* Same with the TFLite fields of a tuning struct (tfl_dot, tfl_graph, tfl_fuse, tfl_group; tune may be NULL). */
int yh_tfl_create_tuned(const void* model_bytes, size_t nbytes, int32_t device, const yh_tuning* tune, yh_tfl** out);

/* Kernel launches per invoke of the prepared plan; how many are CONV_2D and how many of those run on the int8 matrix pipes. */
int yh_tfl_plan_info(const yh_tfl* h, int32_t* launches, int32_t* conv_launches, int32_t* conv_mfma_launches);

int yh_tfl_tensor_count(const yh_tfl* h);

int yh_tfl_tensor_read(yh_tfl* h, int32_t tensor, void* dst, size_t nbytes);   /* test hook: any tensor by index */

/* Measurement hook: mean device milliseconds per frame over `reps` re-runs of the last appended frame, with the inputs and
 * the compat mode of that append (a device frame given to yh_scene_append_classified must still be valid). The re-runs
 * overwrite the outputs with the same values. */
int yh_scene_time(yh_scene* h, int32_t reps, float* ms_per_frame);

/* Test hook: copies the named intermediate tensor of the last forward (layer names of DESIGN.md:
 * "stem", "pool", "c2".."c5", "lat3".."lat5", "p3".."p7", "proto0".."proto3", "proto_up", "head_t0"..) to
 * host memory as f32 NHWC; dims receives {n,h,w,c}. Returns YH_EINVAL for unknown names. "stem" is
 * fused into the pool kernel and only materialised by engines created with debug_tensors = 1 (YH_ESTATE otherwise). */
int yh_debug_read_tensor(yh_engine* h, const char* name, float* dst_host, size_t nfloats, int32_t dims[4]);
/* Same for ONE frame of the batch (dims receives {1,h,w,c}): a batch-64 tensor is gigabytes as f32. */
int yh_debug_read_tensor_frame(yh_engine* h, const char* name, int32_t frame, float* dst_host, size_t nfloats, int32_t dims[4]);
/* Test hook: number of conv kernel launches the last yh_op_conv2d_f16 on this handle was planned as
 * (1 = single launch, 2 = two-phase or channel-split plan; the split-K reduce is not counted). */
int yh_debug_last_conv_launches(const yh_engine* h);

/* Audit hooks (profiles/r03_fault_audit.md). Text into out[cap] (YH_EOVERFLOW if truncated): every buffer of the handle
 * with [base, end) and where those sit inside their 2 MiB page; the nodes of the step for the current batch size as the
 * handle's tuning captures it (kernel symbol, grid, block, the pointers of its launch argument), sorted. */
int yh_debug_alloc_map(yh_engine* h, char* out, size_t cap);
/* Process-wide counters of the setup discipline: out[0] once-only setup sections entered (graph captures), out[1]
 * jobs run by group worker threads, out[2] times one of the former overlapped one of the latter anywhere in the process (must stay 0),
 * out[3] sections / jobs in flight right now. */
int yh_debug_setup_audit(int64_t out[4]);
/* Study hook (tools/study/cu_mask_scaling.py, profiles/r05_overlap.txt): re-creates the handle's two compute streams with
 * hipExtStreamCreateWithCUMask - bit i of mask[] enables CU i (the driver deals the bits over the XCDs round-robin) - and drops the
 * captured steps. How every kernel of a step scales with the CUs it is given is what decides whether two engines on disjoint CU
 * sets can overlap the step's memory-bound and matrix-bound phases. */
int yh_debug_set_cu_mask(yh_engine* h, const uint32_t* mask, int32_t n_words);
/* ... and one PHASE of the forward for the frames last set, `reps` times back to back, eagerly on the handle's stream (timing only: the
 * second phase reads whatever the first last left): phase 0 = backbone + FPN laterals (the memory-bound half of a batch-64 step),
 * phase 1 = the FPN's 3x3 convolutions, the protonet and the prediction head (the matrix-bound half). *ms_total: device time. */
int yh_debug_run_phase(yh_engine* h, int32_t phase, int32_t reps, float* ms_total);
/* Test hook, process-wide: allow = 1 lets yh_group_broadcast_weights / yh_group_replicate_weights treat handles that share a device as
 * separate RCCL ranks. Real RCCL refuses two ranks on one GPU, so this is only meaningful under the stand-in librccl of
 * tests/rccl_standin/ (which is how a one-GPU box executes the n > 1 collective code). Returns the previous setting. */
int yh_debug_rccl_shared_device(int32_t allow);
/* Test hook, process-wide, before the first RCCL call of the process (YH_ESTATE afterwards): open THIS file instead of searching for
 * librccl.so.1. A process that has torch loaded already holds torch's bundled librccl under that soname, so the loader's search path
 * cannot reach the stand-in there (bench.py --rccl-library); processes without torch find it through LD_LIBRARY_PATH, hook unused. */
int yh_debug_rccl_library(const char* path);
int yh_debug_graph_nodes(yh_engine* h, int32_t with_tail, char* out, size_t cap);

/* ---- single-op entry points (parity tests call kernels through the C ABI) ------------------- */

/* One NHWC f16 convolution on the MFMA implicit-GEMM kernel with fused bias (+residual) (+act).
 * All pointers are host memory; the call stages, runs and copies back (test-only convenience).
 * x: [n,h,w,cin] f16 bits; w: [cout,kh,kw,cin] f16 bits; bias: [cout] f32; residual (nullable):
 * [n,ho,wo,cout] f16 bits; y: [n,ho,wo,cout] f16 bits. act: 0 none, 1 relu, 2 tanh. */
int yh_op_conv2d_f16(yh_engine* h, const uint16_t* x, int32_t n, int32_t hh, int32_t ww, int32_t cin,
                     const uint16_t* w, const float* bias, int32_t cout, int32_t kh, int32_t kw,
                     int32_t stride, int32_t pad, const uint16_t* residual, int32_t act,
                     uint16_t* y);
/* The two-source 1x1 convolution (a bottleneck block's last conv and its projection shortcut as one accumulation):
 * y[n][ho][wo][cout] = act(bias + sum_c w[o][c] x1[n][p][q][c] + sum_c w[o][c1 + c] x2[n][p*stride2][q*stride2][c]),
 * x1: [n][ho][wo][c1], x2: [n][h2][w2][c2], w: [cout][c1 + c2] f16 bits; c1, c2 % 64 == 0, cout % 8 == 0; act 0 / 1 (relu).
 * tune.op_tile selects the tile (those of the two-source form only), tune.op_kslices a split-K. */
int yh_op_conv2d_dual_f16(yh_engine* h, const uint16_t* x1, int32_t n, int32_t ho, int32_t wo, int32_t c1,
                          const uint16_t* x2, int32_t h2, int32_t w2, int32_t c2, int32_t stride2,
                          const uint16_t* w, const float* bias, int32_t cout, int32_t act, uint16_t* y);
/* Bilinear resize (align_corners = false) of an NHWC f16 tensor, optional accumulate into dst. */
int yh_op_bilinear_f16(yh_engine* h, const uint16_t* x, int32_t n, int32_t hh, int32_t ww, int32_t c,
                       int32_t ho, int32_t wo, uint16_t* y);
/* 3x3 stride-2 pad-1 max pool of an NHWC f16 tensor. */
int yh_op_maxpool3x3s2_f16(yh_engine* h, const uint16_t* x, int32_t n, int32_t hh, int32_t ww,
                           int32_t c, uint16_t* y);
/* The shared prediction head's form of the convolution: x[n][cells][cin] holds nlev square pyramid
 * levels (edge lengths level_sizes[]) end to end per image; a k x k stride-1 'same' conv runs over all
 * of them in one launch, every tap staying inside its row's own level. y[n][cells][cout]. */
int yh_op_conv2d_levels_f16(yh_engine* h, const uint16_t* x, int32_t n, const int32_t* level_sizes, int32_t nlev, int32_t cin,
                            const uint16_t* w, const float* bias, int32_t cout, int32_t k, int32_t act, uint16_t* y);
/* The fused stem (conv 7x7 stride 2 pad 3, 3 -> 64, bias, ReLU; then max pool 3x3 stride 2 pad 1) on
 * x[n][S][S][3] f16 bits with w[64][7][7][3]; S even. Writes pool_out[n][PO][PO][64] and, if not NULL,
 * stem_out[n][SO][SO][64] (the pre-pool tensor, a test hook). */
int yh_op_stem_pool_f16(yh_engine* h, const uint16_t* x, int32_t n, int32_t S, const uint16_t* w, const float* bias,
                        uint16_t* stem_out, uint16_t* pool_out);
/* The same from raw frames rgb[n][S][S][3] uint8: the kernel's loader also does the preprocessing
 * ((v - mean) / std per channel, rounded to f16), as in production runs. */
int yh_op_stem_pool_rgb8(yh_engine* h, const uint8_t* rgb, int32_t n, int32_t S, const uint16_t* w, const float* bias,
                         uint16_t* stem_out, uint16_t* pool_out);
/* The quantiser of the fp8 forward (the producers' epilogues apply the same conversion): y[i] = OCP FP8 E4M3 code of
 * x[i] * inv_scale (x: f16 bits), round to nearest even, saturating at +-448, NaN -> 0x7F | sign. */
int yh_op_quantize_e4m3(yh_engine* h, const uint16_t* x, size_t n, float inv_scale, uint8_t* y);
/* One fp8 convolution on the 256 x 256 tile the YH_PRECISION_FP8 forward uses: x[n][hh][ww][cin] and w[cout][k][k][cin]
 * are E4M3 codes (cin % 128 == 0), f32 accumulation on the block-scaled MFMA with unit block scales,
 * y = act(acc * scale[ch] + bias[ch] + residual) rounded to f16. If reps > 0, *ms_per_launch receives the mean
 * kernel time of `reps` further launches. */
int yh_op_conv2d_fp8(yh_engine* h, const uint8_t* x, int32_t n, int32_t hh, int32_t ww, int32_t cin, const uint8_t* w,
                     const float* scale, const float* bias, int32_t cout, int32_t k, int32_t stride, int32_t pad,
                     const uint16_t* residual, int32_t act, uint16_t* y, int32_t reps, float* ms_per_launch);
/* Detection tail alone on caller-provided head outputs (host f16 bits, layouts as outputs 0..3)
 * for n frames; results are then read with yh_read_detections. Lets the tail be checked
 * bit-for-bit against the oracle on identical inputs. */
int yh_op_detect(yh_engine* h, const uint16_t* loc, const uint16_t* conf, const uint16_t* mask,
                 const uint16_t* proto, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* YOLACT_HIP_DEBUG_H */
