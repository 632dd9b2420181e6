/*
 * yolact_hip.h — C ABI of the MI355X-native YOLACT inference library (libyolact_hip.so).
 *
 * This is the drop-in boundary for the reference's L2 layer: it stands in for the
 * TFLite interpreter + EdgeTPU delegate that /root/reference/src/yolact.rs drives.  Each entry
 * point names the reference call (file:line, relative to the reference checkout) it replaces.
 * Plain C types only; no exceptions cross the boundary; no global state; every handle owns a
 * private HIP stream and binds its device on every entry, so calls may arrive from different OS
 * threads over time (tokio worker migration, src/main.rs:63-75) but never concurrently per handle.
 *
 * Status convention: every function returning int returns YH_OK (0) on success and a negative
 * YH_E* code otherwise; yh_last_error(h) then holds a human-readable message.  The reference
 * `.expect()`s every failure (src/yolact.rs:20,25,27,29,35,163); the host shim decides to panic.
 */
#ifndef YOLACT_HIP_H
#define YOLACT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_ABI_VERSION 4

enum {
    YH_OK = 0,
    YH_EINVAL = -1,     /* bad argument */
    YH_EHIP = -2,       /* HIP runtime error */
    YH_ENOMEM = -3,
    YH_EWEIGHTS = -4,   /* weight blob does not match the architecture */
    YH_ESTATE = -5,     /* call out of order (e.g. invoke before weights) */
    YH_EDIVERGE = -6,   /* strict compat: the reference's flood fill would not terminate here
                           (src/yolact.rs:57-78) */
    YH_EOVERFLOW = -7   /* a bounded device list overflowed (never silently truncated) */
};

/* Backbone depth of the YOLACT architecture. */
enum { YH_BACKBONE_R50 = 50, YH_BACKBONE_R101 = 101 };

/* Output tensor element kinds; mirrors tflite::context::ElementKind as used at src/yolact.rs:171-186. */
enum { YH_KIND_F32 = 1, YH_KIND_U8 = 3, YH_KIND_F16 = 10 };

/* Reference-compat behaviour of yh_classify_frame_u32. */
enum {
    YH_COMPAT_STRICT = 0, /* bit-for-bit what src/yolact.rs computes, quirks A8'/A9 included; returns
                             YH_EDIVERGE where the reference would loop forever */
    YH_COMPAT_SANE = 1    /* 4-connected component ids, `|` packing: what the author meant */
};

typedef struct yh_engine yh_engine; /* opaque; replaces struct Yolact<'a> (src/yolact.rs:13-15) */

/* Arithmetic of the K-heavy convolutions. F16: f16 operands everywhere (configs[1]-[3]).
 * FP8: 3x3 convolutions with >= 256 input channels read OCP E4M3 activations (written by the producing layer's
 * epilogue, one calibrated scale per tensor) and E4M3 weights (one scale per output channel) on the block-scaled
 * fp8 MFMA; everything else stays f16 (configs[4]; the reference's own model is quantised end to end,
 * data/README.md:5-10, data/FRC_model_edgetpu.log:7-19). */
enum { YH_PRECISION_F16 = 0, YH_PRECISION_FP8 = 1 };


/* Measurement / test knobs of one handle: 32 ints, every one -1 = the library's default (yh_default_config sets them so). A host leaves
 * them alone; what they are is written down in include/yolact_hip_debug.h, which names the fields of this very struct (same layout) and
 * declares the measurement, study and test entry points - nothing the reference's caller needs ("a lab notebook is not an
 * ABI"). There are no process-global switches: the library reads no environment variable. */
#ifndef YH_TUNING_DEFINED
#define YH_TUNING_DEFINED
typedef struct yh_tuning { int32_t knob[32]; } yh_tuning;
#endif

typedef struct yh_config {
    int32_t abi_version;   /* must be YH_ABI_VERSION */
    int32_t device;        /* HIP device ordinal */
    int32_t backbone;      /* YH_BACKBONE_R50 | YH_BACKBONE_R101 */
    int32_t input_size;    /* square input: 550 (YOLACT-550), 700, 224 (reference tile, yolact.rs:143-144) ... */
    int32_t max_batch;     /* frames per invoke the arena is sized for (1..256) */
    int32_t num_classes;   /* incl. background; 81 (src/yolact.rs:108: chunks(81)) */
    int32_t top_k;         /* per-class candidates kept before Fast-NMS (200) */
    int32_t max_dets;      /* detections kept per frame (100) */
    float conf_thresh;     /* 0.05 */
    float nms_thresh;      /* 0.5 */
    int32_t use_graph;     /* 1: capture the forward in a hipGraph after the first invoke */
    int32_t debug_tensors; /* 1: also materialise tensors that production runs fuse away (the pre-pool
                            * "stem" tensor, the bilinear upsamples) for yh_debug_read_tensor; 0 (default): do not */
    int32_t precision;     /* YH_PRECISION_F16 (default) | YH_PRECISION_FP8 */
    int32_t fp8_f16_layers;/* YH_PRECISION_FP8 only: groups of the K-heavy 3x3 layers that stay f16 (a hybrid: accuracy for speed, DESIGN.md
                            * §10 table): bit 0 the shared head trunk, bit 1 the protonet (proto0..3), bit 2 the FPN's pred / down convs
                            * (p3..p7), bit 3 the backbone (layers 3-4). 0 (default): all 36 of YOLACT-700 R101 run fp8 */
    int32_t fp8_per_tensor;/* YH_PRECISION_FP8 only: 0 (default) yh_fp8_calibrate sets one activation scale per INPUT CHANNEL of every E4M3
                            * tensor (folded into the consuming conv's weights along K, applied by the producer's epilogue as a vector:
                            * no extra pass); 1: one scale per tensor (round 3's scheme) */
    int32_t reserved[4];   /* zero */
    yh_tuning tune;        /* yh_default_config sets every field to -1 */
} yh_config;

/* Mirrors tflite TensorInfo {name, element_kind, dims, params{scale, zero_point}} as read at
 * src/yolact.rs:150 and :170-175. dims are NHWC-ordered, dims[0] = current batch. */
typedef struct yh_tensor_info {
    const char* name;
    int32_t kind;      /* YH_KIND_* */
    int32_t ndims;
    int32_t dims[4];
    float scale;       /* 1.0 for float tensors */
    int32_t zero_point;/* 0 for float tensors */
} yh_tensor_info;

/* One detection of the fused path (absent in the reference, src/yolact.rs:3-5). */
typedef struct yh_detection {
    int32_t class_id;   /* 0..num_classes-2 (foreground index; background removed) */
    int32_t prior;      /* prior index the detection came from */
    float score;
    float box[4];       /* x1,y1,x2,y2 relative to the frame, unclamped */
} yh_detection;

/* ---- life cycle --------------------------------------------------------------------------- */

/* edgetpu::version() (src/scene.rs:62). Static string. */
const char* yh_version(void);

/* Fills *cfg with the YOLACT-550 R50 defaults (every tuning field -1). */
void yh_default_config(yh_config* cfg);

/* FlatBufferModel::build_from_file + InterpreterBuilder::new/build + EdgeTpuContext::open_device +
 * set_num_threads + allocate_tensors (src/yolact.rs:18-35): builds the layer table for cfg,
 * allocates the whole activation arena and a private stream on cfg->device. No weights yet. */
int yh_create(const yh_config* cfg, yh_engine** out);
void yh_destroy(yh_engine* h);
const char* yh_last_error(const yh_engine* h); /* h may be NULL: last create error of this thread */

/* ---- weights ------------------------------------------------------------------------------ */

/* Size in bytes of the canonical weight blob (DESIGN.md §Weight blob) for this architecture. */
size_t yh_weights_nbytes(const yh_engine* h);
/* The canonical blob as loaded, in device memory on the handle's device (NULL before a load): what
 * yh_load_weights_device of another handle on the same device, or the RCCL broadcast below, reads. */
const void* yh_weights_device_ptr(const yh_engine* h);
/* Writes the seeded synthetic blob (He-scaled uniform, BN folded to bias) into host memory. The
 * reference's model file is absent (.MISSING_LARGE_BLOBS:1-2); this is its stand-in. */
int yh_weights_generate(const yh_engine* h, uint64_t seed, void* blob_host, size_t nbytes);
/* Loads a canonical blob from host memory: validates the per-layer header, repacks to the kernels'
 * padded KRSC panels on device. Replaces build_from_file("data/FRC_model_edgetpu.tflite"), yolact.rs:18-20. */
int yh_load_weights_host(yh_engine* h, const void* blob_host, size_t nbytes);
/* Same, blob already in device memory on this handle's device (e.g. the receive buffer of an RCCL
 * broadcast from rank 0 over xGMI: SURVEY.md §8e). */
int yh_load_weights_device(yh_engine* h, const void* blob_dev, size_t nbytes);

/* ---- fp8 precision (yh_config.precision = YH_PRECISION_FP8; BASELINE.json configs[4]) -------------------------
 * The K-heavy 3x3 convolutions read OCP E4M3 operands on the block-scaled fp8 MFMA. Activations: one scale per INPUT
 * CHANNEL of every tensor such a convolution reads (round 4; yh_config.fp8_per_tensor = 1: one per tensor), which must be
 * set before the first invoke - by calibration on representative frames, or layer by layer from stored values. The channel
 * scale s[c] is folded into the consumer's weights along K (t = w * s[c]) before their quantisation with one scale per
 * output channel (s_w = max |t| / 448), so the weights' E4M3 codes are (re)made whenever a tensor's scales are set; the
 * producer's epilogue multiplies by 1 / s[c] - a vector instead of a scalar, no extra pass. */
/* Runs the f16 forward of the frames last set and sets the scales of every fp8 input tensor to
 * max(2 max|x[.., c]|, max|x| / 16) / 448 - twice the channel's own maximum on the calibration frames and never less than a sixteenth of
 * the tensor's: headroom for frames the calibration has not seen (free in a floating-point code; round 5). YH_ESTATE
 * (and no scale changed) if that forward overflowed: a non-finite maximum would make every code of the tensor 0. */
int yh_fp8_calibrate(yh_engine* h);
/* The convolutions that read E4M3 operands, in execution order: layer name (DESIGN.md layer names: "l3b0_b", "p5",
 * "proto0", "head_t" ...) and the LARGEST channel scale of its input tensor (yh_fp8_layer_channel_scales gives them all). */
int yh_fp8_layer_count(const yh_engine* h);
int yh_fp8_layer_info(const yh_engine* h, int32_t i, const char** conv_name, float* act_scale);
/* Channels of layer i's input tensor / its channel scales (n must equal the channel count; 1.0 where none is set yet). */
int yh_fp8_layer_channels(const yh_engine* h, int32_t i);
int yh_fp8_layer_channel_scales(const yh_engine* h, int32_t i, float* scales, int32_t n);
/* Scales belong to TENSORS (allocations), not to layers: layers that read one allocation share theirs - P3..P7 live in one
 * pyramid buffer, so setting the scales of "p6" also sets them for "p7", "head_t" and "proto0" (yh_fp8_layer_info shows it).
 * Positive finite values are required. yh_invoke / yh_evaluate return YH_ESTATE, naming the layers, until EVERY E4M3
 * input tensor has its scales (by calibration or by these calls). Loading weights AGAIN on a handle discards the scales (they
 * were calibrated for the old weights); scales set before the first load are kept. yh_fp8_set_layer_scale sets the same
 * value in every channel (a per-tensor scale). */
int yh_fp8_set_layer_scale(yh_engine* h, int32_t i, float act_scale);
int yh_fp8_set_layer_channel_scales(yh_engine* h, int32_t i, const float* scales, int32_t n);

/* ---- multi-GPU: the path's one collective (north_star: "weights replicated once via RCCL
 * broadcast over xGMI, no per-step collectives"). The reference's caller is a Rust process (src/main.rs:63-75,
 * src/scene.rs:63), not torch, so the broadcast is reachable through this ABI; librccl.so is opened on first use.
 * Frames then shard over the handles with no further communication. ------------------------------------------- */
#define YH_RCCL_ID_BYTES 128
/* One process, one handle per GPU (a host thread + stream per device): copies the weights
 * loaded on handles[root] to every other handle (ncclCommInitAll + grouped ncclBroadcast of the canonical blob,
 * then the usual validation and repack on each device). */
int yh_group_broadcast_weights(yh_engine** handles, int32_t n, int32_t root);
/* One process per GPU: rank `root` has its weights loaded; every rank calls this with the same id (from
 * yh_rccl_unique_id on one rank, handed to the others by the host's own means: ncclCommInitRank + ncclBroadcast). */
int yh_rccl_unique_id(void* id_out /* YH_RCCL_ID_BYTES */);
int yh_rank_broadcast_weights(yh_engine* h, const void* id, int32_t rank, int32_t nranks, int32_t root);

/* ---- frame sharding over the GPUs of one node: one process, one engine handle per device,
 * one host worker thread + stream per handle, contiguous frame blocks, no per-step collective. The reference's caller is
 * ONE process that owns the frame loop (src/main.rs:63-75, src/scene.rs:77-92); this is what it calls with N frames to get
 * N results over 8 devices. Two members may name the same device (they then share it: the form a one-GPU box can run). */
typedef struct yh_group yh_group;
/* cfg: the per-member configuration (cfg->device is ignored; cfg->max_batch = frames per member per step). */
int yh_group_create(const yh_config* cfg, const int32_t* devices, int32_t n, yh_group** out);
void yh_group_destroy(yh_group* g);
const char* yh_group_last_error(const yh_group* g);   /* g may be NULL: last create error of this thread */
int yh_group_size(const yh_group* g);
/* Member i's own handle (weights on member 0, per-handle tuning, yh_debug_*, ...); owned by the group. */
yh_engine* yh_group_member(yh_group* g, int32_t i);
/* Loads the canonical blob on member 0 and replicates it: ONE RCCL broadcast over the distinct devices
 * (yh_group_broadcast_weights), device-to-device copies for members that share a device with an earlier member, the host as
 * the fallback when librccl cannot be used. yh_group_weights_replication says which of these ran. */
int yh_group_load_weights_host(yh_group* g, const void* blob_host, size_t nbytes);
/* The replication alone (member 0 already holds its weights, e.g. from yh_weights_generate + yh_load_weights_host). */
int yh_group_replicate_weights(yh_group* g);
const char* yh_group_weights_replication(const yh_group* g);
/* fp8 precision: member 0 calibrates on the frames last set on it, every member runs with member 0's scales. */
int yh_group_fp8_calibrate(yh_group* g);
/* n_frames u8 RGB frames [n][S][S][3] in host memory, 1 <= n_frames <= members * max_batch: member i takes a contiguous block
 * (the first n_frames % members members one frame more), copies it in underneath its previous step and enqueues
 * yh_evaluate (with_tail = 1) or yh_invoke (0). Returns when every member has enqueued its step: the frames are free again,
 * the GPUs run on; consecutive calls pipeline. */
int yh_group_evaluate(yh_group* g, const uint8_t* frames_host, int32_t n_frames, int32_t with_tail);
/* The same with frames resident on each member's own device: frames_dev[i] holds counts[i] frames (0: the member sits out). */
int yh_group_evaluate_device(yh_group* g, const uint8_t* const* frames_dev, const int32_t* counts, int32_t with_tail);
/* Captures, on the CALLING thread and one member after the other, the step every member would run for a call with n_frames frames
 * (both input buffers of each member's block size): what the first yh_group_evaluate* of that size would otherwise do before it starts
 * its workers. A host calls it for the frame counts it will use at start-up, so that no capture falls into its frame loop (the
 * reference allocates everything in Yolact::init, src/yolact.rs:28-35). Needs the weights (fp8 precision: and the scales). The group's
 * worker threads never capture and never allocate: DESIGN.md section 7. */
int yh_group_prepare(yh_group* g, int32_t n_frames, int32_t with_tail);
int yh_group_sync(yh_group* g);
/* Results by GLOBAL frame index of the last yh_group_evaluate* (waits for that member's step): as yh_read_detections. */
int yh_group_read_detections(yh_group* g, int32_t frame, int32_t* count, yh_detection* dets, int32_t dets_capacity, uint8_t* masks, size_t masks_capacity);
/* Which member holds global frame `frame` of the last evaluate, and at which index of its batch. */
int yh_group_frame_owner(const yh_group* g, int32_t frame, int32_t* member, int32_t* local);

/* ---- interpreter-shaped surface (lets classify_tile, src/yolact.rs:133-190, port unchanged) -- */

/* interpreter.inputs()[0] + tensor_info(..).dims (src/yolact.rs:149-150): {max_batch,S,S,3}. */
int yh_input_dims(const yh_engine* h, int32_t dims[4]);
/* tensor_data_mut(in).copy_from_slice (src/yolact.rs:161-162): copies n frames of u8 RGB NHWC
 * from host memory into the engine-owned input buffer. The caller's buffer is free again on return, whether it is
 * pageable or pinned (the copy itself is asynchronous on the handle's stream). */
int yh_set_input_u8(yh_engine* h, const uint8_t* rgb_host, int32_t n_frames);
/* Same with frames already resident in device memory (bench path: inputs in HBM). */
int yh_set_input_u8_device(yh_engine* h, const uint8_t* rgb_dev, int32_t n_frames);
/* interpreter.invoke() (src/yolact.rs:163): the network forward for the frames last set.
 * Asynchronous on the handle's stream; yh_sync or any output read waits for it. */
int yh_invoke(yh_engine* h);
/* allocate_tensors() at its proper time (src/yolact.rs:35): captures the step for n_frames frames (with_tail: yh_evaluate's, else
 * yh_invoke's) for both input buffers NOW, on the calling thread, instead of at the first yh_invoke / yh_evaluate of that batch size.
 * A no-op for use_graph = 0. A host that drives several handles from several threads calls this for each handle, serially, before
 * those threads start (a capture is once-only work deep in the HIP runtime; the library never does it on a group's worker thread). */
int yh_prepare(yh_engine* h, int32_t n_frames, int32_t with_tail);
int yh_sync(yh_engine* h);
/* interpreter.outputs().len() / tensor_info(output) (src/yolact.rs:166,170). Output order:
 *   0 loc    [n,P,4]  f16   box regressions
 *   1 conf   [n,P,C]  f16   class logits (pre-softmax)
 *   2 mask   [n,P,32] f16   tanh mask coefficients
 *   3 proto  [n,Hp,Wp,32] f16 prototypes (ReLU)
 *   4 cells  [n,H3,W3,C] f32 class logits of anchor 0 on the stride-8 level: for a 224 input this
 *            is the 28*28*81 tensor the reference reads as results[4] (src/yolact.rs:91,108). */
int yh_output_count(const yh_engine* h);
int yh_output_info(const yh_engine* h, int32_t index, yh_tensor_info* info);
/* tensor_data::<u8|f32>(output) (src/yolact.rs:173,180): copies output `index` to host memory
 * converted to f32 (the reference dequantises to Vec<f32> at :176-181). nfloats = capacity. */
int yh_output_read_f32(yh_engine* h, int32_t index, float* dst_host, size_t nfloats);
/* Device pointer of the engine-owned output buffer; valid until the next yh_invoke. */
const void* yh_output_device_ptr(const yh_engine* h, int32_t index);

/* ---- fused detection path (north_star "Yolact::evaluate") ---------------- */

/* Forward + softmax/threshold + per-class top-k + box decode + Fast-NMS + top max_dets +
 * mask assembly (mask-coeff x prototypes, logit>0 == sigmoid>0.5, box crop) for the frames last
 * set. Asynchronous; results are read with yh_read_detections. */
int yh_evaluate(yh_engine* h);
/* Copies results of the last yh_evaluate for frame `frame`: counts[0] = number of detections d,
 * dets[0..d), and (if masks != NULL) d binary masks of Hp*Wp bytes each (0/1). */
int yh_read_detections(yh_engine* h, int32_t frame, int32_t* count, yh_detection* dets,
                       int32_t dets_capacity, uint8_t* masks, size_t masks_capacity);
int yh_proto_dims(const yh_engine* h, int32_t dims[2]); /* {Hp, Wp} */
int yh_num_priors(const yh_engine* h);
/* Copies the P x 4 prior table (cx,cy,w,h) to host memory. */
int yh_read_priors(const yh_engine* h, float* dst_host, size_t nfloats);

/* ---- reference-compat path: Yolact::classify (src/yolact.rs:39-41, :192-234) --------------- */

/* In-place classify of one packed camera frame: `frame` holds width*height u32 pixels packed
 * r<<24|g<<16|b<<8 (src/scene.rs:86). Runs, all on device: unpack (yolact.rs:195-201), Triangle
 * resize to (2S x S) (yolact.rs:208), two S x S tiles as one batch of 2 (yolact.rs:213-217),
 * forward, output-4 gated argmax (yolact.rs:108-118), ids (yolact.rs:52-88), pack + x8 nearest
 * upsample (yolact.rs:127-128), stitch (yolact.rs:219-220), Triangle resize back (yolact.rs:231)
 * and repack (yolact.rs:233). Requires input_size*2 x input_size tiles, input_size % 8 == 0
 * (224 for the reference) and max_batch >= 2. Synchronous (the reference's classify is). */
int yh_classify_frame_u32(yh_engine* h, uint32_t* frame_host, int32_t width, int32_t height,
                          int32_t compat_mode);
/* The post-network half only, on caller-provided output-4 logits (n_tiles x cells x C f32, host):
 * exactly `postprocess` (src/yolact.rs:90-131) per tile. out: n_tiles x (S*S) u32. Lets the
 * reference's integer logic be checked bit-for-bit without a network in the loop. */
int yh_postprocess_cells(yh_engine* h, const float* cells_host, int32_t n_tiles, uint32_t* out_host,
                         int32_t compat_mode);
/* Device Triangle resize of an RGB8 image, the `image` crate call at src/yolact.rs:208,:231. */
int yh_resize_triangle_rgb8(yh_engine* h, const uint8_t* src_host, int32_t sw, int32_t sh,
                            uint8_t* dst_host, int32_t dw, int32_t dh);

/* ---- TFLite model path: run the reference's own model family ---------------- */
/* A user who holds data/FRC_model.tflite (uint8 per-tensor quantised MobileNetV2-FPN YOLACT,
 * data/README.md:5-16; absent from the checkout) runs it here instead of tflite+EdgeTPU. Supported
 * builtin ops: the histogram of data/FRC_model_edgetpu.log:7-19 (CONV_2D, DEPTHWISE_CONV_2D, ADD,
 * PAD, RESIZE_BILINEAR, TANH, RELU, CONCATENATION, RESHAPE, QUANTIZE) plus RELU6 and DEQUANTIZE.
 * The EdgeTPU-compiled file is one custom op and is rejected with a message naming it. */
typedef struct yh_tfl yh_tfl;
/* Parses only (no GPU needed): YH_OK or YH_EWEIGHTS with a message; never reads out of bounds. */
int yh_tfl_validate(const void* model_bytes, size_t nbytes, int32_t* n_tensors, int32_t* n_ops, char* err, size_t err_cap);
/* FlatBufferModel::build_from_file + InterpreterBuilder + allocate_tensors (src/yolact.rs:18-35). */
int yh_tfl_create(const void* model_bytes, size_t nbytes, int32_t device, yh_tfl** out);
void yh_tfl_destroy(yh_tfl* h);
const char* yh_tfl_last_error(const yh_tfl* h);
int yh_tfl_input_info(const yh_tfl* h, yh_tensor_info* info);                 /* inputs()[0], :149-150 */
int yh_tfl_output_count(const yh_tfl* h);                                      /* outputs(), :166 */
int yh_tfl_output_info(const yh_tfl* h, int32_t i, yh_tensor_info* info);      /* tensor_info(output), :170-175 */
/* Images per invoke: 1 (default) or 2 - the model is batch 1 (yolact.rs:149-150), its two tiles per frame are independent
 * (:216-217) and run as one pass of a batch plan (activations image-major; yh_tfl_classify_frame_u32 does this itself).
 * yh_tfl_set_input then takes n images, yh_tfl_output_read / yh_tfl_tensor_read return n. YH_EINVAL for a model with an
 * operator along the image axis. */
int yh_tfl_set_batch(yh_tfl* h, int32_t n_images);
int yh_tfl_set_input(yh_tfl* h, const void* data, size_t nbytes);              /* tensor_data_mut copy, :161-162 */
int yh_tfl_invoke(yh_tfl* h);                                                  /* interpreter.invoke(), :163 */
int yh_tfl_output_read(yh_tfl* h, int32_t i, void* dst, size_t nbytes);        /* tensor_data::<u8|f32>, :173,:180 */
/* Yolact::classify (src/yolact.rs:192-234) with this model in the middle: two S x S tiles, one
 * invoke each, output 4 dequantised (:177) and post-processed (:90-131), all on device. */
int yh_tfl_classify_frame_u32(yh_tfl* h, uint32_t* frame_host, int32_t width, int32_t height, int32_t compat_mode);

/* ---- scene back-end: append_scene's two compute dispatches (src/scene.rs:147-331) --------------
 * shaders/pt_cloud.comp (depth + class image -> bird's-eye height map with sigmoid bumps, ball centroids) and
 * shaders/pt_cloud_weights.comp (world positions, 8-neighbour edge lengths) as HIP kernels, one lane per pixel in
 * 8x8 workgroups ([80,60,1] at 640x480: scene.rs:245,:256). The shaders as written race and use undefined GLSL;
 * what is computed is the deterministic reading of DESIGN.md §Scene (stage-by-stage completion, exact ball means).
 * compat_mode: YH_COMPAT_STRICT keeps pack()'s `&` (pt_cloud_weights.comp:32: every neighbour position decodes to
 * world(0,0)); YH_COMPAT_SANE measures the distance to the actual neighbour. */
typedef struct yh_scene yh_scene;
int yh_scene_create(int32_t device, int32_t width, int32_t height, yh_scene** out);   /* scene.rs:150-155: the storage images */
void yh_scene_destroy(yh_scene* h);
const char* yh_scene_last_error(const yh_scene* h);
/* One frame: depth u16 [h][w] (the R16_UINT texture, scene.rs:197) and the class image [h][w][2] = (class, id) (the
 * R8G8_UINT texture, scene.rs:198), both in host memory; the buffers are free again on return (pageable or pinned).
 * The computation is asynchronous on the handle's stream. */
int yh_scene_append(yh_scene* h, const uint16_t* depth_host, const uint8_t* class_id_host, int32_t compat_mode);
/* The class image taken straight from a classified frame as yh_classify_frame_u32 leaves it (packed u32 [h][w];
 * frame_on_device = 1: a device pointer, e.g. yh_classify_device_frame - no host round trip of the class image).
 * STRICT reads it as the reference does - the low 16 bits (src/scene.rs:93), class = bits 7-0, id = bits 15-8, which
 * classify leaves zero - SANE reads class = bits 31-24, id = bits 23-16. */
int yh_scene_append_classified(yh_scene* h, const uint16_t* depth_host, const uint32_t* frame, int32_t frame_on_device, int32_t compat_mode);
/* scene.rs:284-330: height map u32 [h][w], world / connections0 / connections1 f32 [h][w][4], balls f32 [100][4]
 * = (mean x, mean y, pixel count, 0). Any pointer may be NULL. Waits for the frame. */
int yh_scene_read(yh_scene* h, uint32_t* map, float* world, float* conn0, float* conn1, float* balls);
/* Device copy of the frame the last yh_classify_frame_u32 produced (valid until the next classify on this handle). */
const uint32_t* yh_classify_device_frame(const yh_engine* h);

#ifdef __cplusplus
}
#endif
#endif /* YOLACT_HIP_H */
