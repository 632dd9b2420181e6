/*
 * oracle.h — CPU restatement of the hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 * The product (libyolact_hip.so) never links, imports or calls anything in this directory.
 *
 * Parity status:
 *   - orc_ref.c restates the reference's own Rust pre/post-processing (src/yolact.rs:52-131,
 *     :133-234, src/scene.rs:86,:93). The reference has no tests or fixtures; it is pinned by the
 *     known-answer vectors of SURVEY.md Appendix A (tests/golden/reference_kat.json).
 *   - The Triangle resampler restates image 0.24.1 (Cargo.lock:481-484), which is NOT vendored in
 *     /root/reference and was written from the published algorithm: PARITY UNPINNED (+-1 LSB).
 *   - orc_net.c / orc_detect.c restate the published YOLACT architecture (SPEC-EXTERNAL; the
 *     reference's network lives in the absent tflite model + un-vendored tflite 0.9.0 /
 *     edgetpu 0.1.0 crates): PARITY UNPINNED against the reference; primitives are pinned against
 *     torch CPU in tests/test_oracle_vs_torch.py.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- orc_ref.c : the reference's own logic ---------------- */
void orc_unpack_rgb(const uint32_t* px, size_t n, uint8_t* rgb);          /* yolact.rs:134-140,195-201 */
void orc_pack_rgb(const uint8_t* rgb, size_t n, uint32_t* px);            /* yolact.rs:213-214,230-231; scene.rs:86 */
void orc_dequant_u8(const uint8_t* q, size_t n, float scale, int32_t zero_point, float* out); /* yolact.rs:172-178 */
void orc_gated_argmax(const float* dets, int ncells, int nch, uint8_t* classes);             /* yolact.rs:108-118 */
/* Returns 0 if the reference's flood fill terminates (ids then filled), 1 if it would loop forever. */
int orc_terrible_id(const uint8_t* classes, int ncells, int grid_w, int8_t* ids);            /* yolact.rs:52-88 */
/* Correct 4-connected labelling of class-3 cells (the SANE mode; not in the reference). */
void orc_sane_id(const uint8_t* classes, int grid_h, int grid_w, int8_t* ids);
/* postprocess (yolact.rs:90-131) for one tile: cells [grid*grid*nch] -> out [(grid*8)^2] u32.
 * mode 0 strict (returns 1 where the reference diverges), 1 sane. */
int orc_postprocess_tile(const float* cells, int grid, int nch, int mode, uint32_t* out);
/* image 0.24.1 imageops::resize(.., FilterType::Triangle) on RGB8 (yolact.rs:208,:231). */
void orc_resize_triangle_rgb8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
/* classify (yolact.rs:192-234) up to the network: frame -> two SxS RGB8 tiles (tiles: 2*S*S*3 bytes). */
void orc_classify_pre(const uint32_t* frame, int w, int h, int S, uint8_t* tiles);
/* classify after the network: two tiles' cells logits -> frame overwritten. Returns 1 on divergence. */
int orc_classify_post(const float* cells2, int S, int nch, int mode, uint32_t* frame, int w, int h);
/* consumer's view (scene.rs:93): low 16 bits. */
void orc_consumer_low16(const uint32_t* frame, size_t n, uint16_t* out);

/* ---------------- orc_net.c : YOLACT network (SPEC-EXTERNAL) ---------------- */
typedef struct orc_net orc_net;
typedef struct {
    int backbone;    /* 50 | 101 */
    int input_size;  /* S */
    int num_classes; /* 81 */
} orc_net_cfg;

size_t orc_weights_nbytes(const orc_net_cfg* cfg);
int orc_weights_generate(const orc_net_cfg* cfg, uint64_t seed, void* blob, size_t nbytes);
orc_net* orc_net_create(const orc_net_cfg* cfg, const void* blob, size_t nbytes);
void orc_net_destroy(orc_net* net);
int orc_net_num_priors(const orc_net* net);
void orc_net_proto_dims(const orc_net* net, int* hp, int* wp);
void orc_net_priors(const orc_net* net, float* out); /* P*4 cx,cy,w,h */
double orc_net_flops_per_frame(const orc_net* net);
/* Forward of n RGB8 frames. f16_storage=1 rounds every layer output to binary16 (the mode the HIP
 * engine is compared with). Outputs (float, values f16-representable in f16 mode):
 *   loc [n,P,4], conf [n,P,C], mask [n,P,32], proto [n,Hp,Wp,32]. Any may be NULL. */
int orc_net_forward(orc_net* net, const uint8_t* rgb, int n, int f16_storage, int nthreads,
                    float* loc, float* conf, float* mask, float* proto);
/* Copy of a named intermediate of the last forward (e.g. "stem", "pool", "c2".."c5", "p3".."p7",
 * "proto3", "head_t0"); returns element count or -1. dims out: n,h,w,c. */
long orc_net_get(const orc_net* net, const char* name, float* out, size_t cap, int dims[4]);
/* Single ops for primitive tests (float NHWC, f32 accumulate). w: [cout,kh,kw,cin]. */
void orc_conv2d(const float* x, int n, int h, int w, int cin, const float* wt, const float* bias,
                int cout, int kh, int kw, int stride, int pad, const float* residual, int act,
                int f16_storage, int nthreads, float* y);
void orc_bilinear(const float* x, int n, int h, int w, int c, int ho, int wo, int f16_storage, float* y);
void orc_maxpool3x3s2(const float* x, int n, int h, int w, int c, float* y);
float orc_f16_round(float v);
uint16_t orc_f32_to_f16_bits(float v);
float orc_f16_bits_to_f32(uint16_t b);
float orc_spec_expf(float x);
float orc_spec_tanhf(float x);

/* ---------------- orc_detect.c : detection tail (SPEC-EXTERNAL) ---------------- */
typedef struct {
    int num_classes, top_k, max_dets;
    float conf_thresh, nms_thresh;
} orc_det_cfg;
typedef struct {
    int32_t class_id, prior;
    float score;
    float box[4];
} orc_detection;
/* One frame. masks: max_dets*hp*wp bytes (0/1), may be NULL. Returns number of detections. */
int orc_detect(const orc_det_cfg* cfg, const float* loc, const float* conf, const float* mask,
               const float* proto, const float* priors, int P, int hp, int wp,
               orc_detection* dets, uint8_t* masks);


/* ---- OCP FP8 E4M3 (orc_fp8.c): groundwork for an fp8 convolution path, pinned by exhaustive search ---- */
/* accuracy study: the net's K-heavy 3x3 convs (cin >= 256) with E4M3-rounded operands (orc_net.c) */
void orc_net_set_fp8_study(orc_net* net, int on);
/* extended study: act_mode 0 off | 1 one scale per tensor | 2 E8M0 blocks of 32 channels | 3 one scale
 * per input channel; w_mode 1 one scale per output channel | 2 E8M0 blocks of 32 along K; skip_csv: name prefixes kept f16.
 * Applies to the convolutions the engine's fp8 plan takes (3x3, cin >= 256 and % 128 == 0, cout % 256 == 0). */
void orc_net_set_fp8_study_ex(orc_net* net, int act_mode, int w_mode, const char* skip_csv);
/* fp8 forward mode (configs[4]): name the convolutions that read E4M3 operands, each with its activation scales - one per
 * INPUT CHANNEL, folded into the weights' K axis before their per-output-channel quantisation (orc_net.c: run_conv) */
void orc_net_clear_fp8(orc_net* net);
int orc_net_add_fp8_layer_ch(orc_net* net, const char* conv_name, const float* scales, int channels);
float orc_e4m3_to_f32(uint8_t b);
uint8_t orc_e4m3_from_f32(float x);              /* round to nearest even, saturating at +-448, NaN -> 0x7F */
void orc_quantize_e4m3(const float* x, long long n, float inv_scale, uint8_t* y);

/* ---- scene back-end (orc_scene.c): SURVEY.md §8f-4, shaders/pt_cloud.comp + pt_cloud_weights.comp. PARITY UNPINNED
 * (the shaders race and use undefined GLSL; the file header lists what is frozen instead). */
float orc_spec_logf(float x);
void orc_scene(const uint16_t* depth, const uint8_t* cls_id, int W, int H, int mode,
               uint32_t* map, float* world, float* conn0, float* conn1, float* balls);

#ifdef __cplusplus
}
#endif
#endif
