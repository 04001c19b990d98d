/*
 * pmp.h — C ABI of libpmp_hip.so: the MI355X (gfx950) partition-map prediction path.
 *
 * The reference (AolinFeng/PMP-VVC-TIP2023) has no FFI; its hot path is plain Python.  This ABI replaces the
 * two function seams of that path and the helpers either side of them:
 *
 *   pmp_infer*              <- Metrics.py:387-419   inference_pre_QBD  (Net_Q + Net_BD forward, head regrouping)
 *                              Model_QBD.py:59-253  the four Down-Up-CNN nets
 *                              Inference_QBD.py:194-200 driver tensor prep (chroma = maxpool2(Y) ++ U ++ V)
 *   pmp_postprocess*        <- Metrics.py:764-774   seq_post_process   = eli_structual_error (Metrics.py:612-637)
 *                              + Map2Partition.py:98-373 Map_to_Partition (per-block search)
 *   pmp_infer_postprocess*  <- both, device-resident (no logits round trip): the throughput path
 *   pmp_cut_blocks*         <- Inference_QBD.py:104-149 output_block_yuv (+ :106-109 10-bit -> 8-bit)
 *   pmp_write_partition_file<- Map2Partition.py:385-412 frame tiling + text emission, parsed by
 *                              EncAppCfg::parsePartitionMatrix (codec/.../App/EncoderApp/EncAppCfg.cpp:4234-4404)
 *   pmp_load_weights        <- Inference_QBD.py:28-46 load_pretrain_model (state_dict tensors by name)
 *
 * Conventions
 *   - Every call returns 0 (PMP_OK) or a negative error class; pmp_last_error() gives the message.  Nothing
 *     aborts or throws across the ABI; HIP errors are mapped to PMP_E_HIP.
 *   - One context per GPU (several per GPU work too).  A context is not thread-safe: one host thread at a time per context.  Different
 *     contexts are independent and may be driven from different host threads CONCURRENTLY - everything a call touches hangs off its
 *     context (streams, workspaces, weights, calibration stream and workspace, range-guard queue, error string); the only
 *     process-wide state is the pool of parked workspaces behind pmp_destroy / pmp_trim (mutex-protected) and the calling thread's
 *     context-less error string (thread-local).  tests/test_gpu_threads.py: two contexts on two threads = the serial results, bit
 *     for bit.  (pmp_debug_set_conv_variant has no state in this library; in the measurement library it is process-wide.)
 *   - There is NO CPU fallback: every compute entry point needs a context on a gfx950 device.
 *   - "block" = one 64x64 luma region with its 4-pixel top/left context: u8[68][68] luma, u8[34][34] per chroma
 *     plane (Inference_QBD.py:190-191).  One VTM CTU (128x128) = 4 blocks.
 *   - Layouts (all dense, row-major):  qt f32[n][8][8]; bt, dire f32[n][3][16][16] (layer-major, as
 *     Metrics.py:399-402 regroups the heads); hor, ver u8[n][16][16]; qt_u8 u8[n][8][8]; dire_i8 i8[n][3][16][16].
 *   - *_device variants take device pointers and run asynchronously on the context's stream
 *     (pmp_set_stream lets the caller supply it); the others take host pointers and synchronise.  pmp_synchronize makes the
 *     outputs of every *_device call made so far final (see the range guard below).
 */
#ifndef PMP_H
#define PMP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMP_OK 0
#define PMP_E_INVALID (-1)   /* bad argument / unknown net / shape mismatch */
#define PMP_E_HIP (-2)       /* HIP runtime error (message has hipGetErrorString) */
#define PMP_E_NOWEIGHTS (-3) /* weights for (comp, qp) not loaded */
#define PMP_E_IO (-4)        /* file could not be written */
#define PMP_E_NOMEM (-5)     /* device or host allocation failed */
#define PMP_E_NODEVICE (-6)  /* no gfx950 device: there is no CPU fallback */
#define PMP_E_RANGE (-7)     /* f16x3 datapath: an activation left the fp16 range and the policy is PMP_SAT_ERROR */

enum { PMP_LUMA = 0, PMP_CHROMA = 1 };
enum { PMP_NET_LUMA_Q = 0, PMP_NET_LUMA_MSBD = 1, PMP_NET_CHROMA_Q = 2, PMP_NET_CHROMA_MSBD = 3 };

typedef struct pmp_ctx pmp_ctx;

/* One state_dict entry (name without the DataParallel "module." prefix, Inference_QBD.py:23-25). */
typedef struct {
    const char *name;  /* e.g. "resblock_q1.left.0.weight" */
    int ndim;          /* 4 (OIHW conv weight) or 1 (bias) */
    int shape[4];
    int64_t offset;    /* in floats, into the blob */
} pmp_tensor_desc;

const char *pmp_version(void);

/* pmp_last_error(NULL) returns the calling thread's last context-less error (e.g. from pmp_create). */
const char *pmp_last_error(const pmp_ctx *ctx);

int pmp_create(int device_id, pmp_ctx **out);
int pmp_destroy(pmp_ctx *ctx);

/* pmp_destroy parks the context's activation workspace (up to 10 GB) for the next context created on the same device instead of
 * freeing it: a large hipMalloc right after a hipFree of that size stalls for 0.5-1.4 s now and then on MI355X (the freed memory is
 * still being cleared).  Up to two parked buffers per device (a context in overlap mode owns two workspaces); pmp_trim() returns parked memory to the driver - a host that destroys its
 * context to hand the VRAM to another library calls it right after pmp_destroy (or runs with PMP_PARK_WORKSPACE=0 in the
 * environment: nothing is parked then).  pmp_trim leaves the calling thread's current device as it is. */
int pmp_trim(void);

/* Use the caller's hipStream_t (e.g. torch's current stream); NULL restores the context's own stream. */
int pmp_set_stream(pmp_ctx *ctx, void *hip_stream);
int pmp_synchronize(pmp_ctx *ctx);

/* Blocks processed per pass; n > chunk is looped.  1..4096 (32-bit element offsets inside one activation tensor), default
 * 4096.  The activation workspace is sized for the blocks a pass actually runs, min(n, chunk), and tensors share memory once
 * their last consumer is enqueued: 2.5 MB per luma block on the default datapath (10 GB for a full 4096-block pass, 10 MB
 * for a 4-block call; bf16x6 3.75 MB, chroma 1.1 MB per block); it only grows, to what the largest pass so far needed.
 * pmp_get_workspace_bytes reports that need (the buffer behind it may be a larger one taken over from a destroyed context, see pmp_trim),
 * plus the second workspace overlap mode holds while it is on (sized for its half-call chunks; freed by pmp_set_overlap(ctx, 0)). */
int pmp_set_chunk(pmp_ctx *ctx, int blocks);
int64_t pmp_get_workspace_bytes(const pmp_ctx *ctx);

/* Overlap mode (off by default; PMP_OVERLAP=1 in the environment turns it on at pmp_create).  A call of at least 1024 blocks is cut into
 * (at least) two chunks; even chunks run on the context's stream and workspace, odd ones on a second, internal stream with a second
 * workspace, forked from and joined to the context's stream by events - one chunk's small launches (stems, 16x16 tails, HBM-bound 32x32
 * layers) then run beside the other's 64x64 convolutions.  Blocks are independent, so results do not depend on how a call is cut (bit-
 * identical records; tests/test_gpu_parity.py).  Measured -0.2 ... -1.2 % (luma) / -0.7 ... -1.7 % (chroma) on the 4096-block step.  Two launches share
 * the device then, so the per-launch durations of pmp_ktime_* (and of a profiler) no longer describe a kernel running alone: bench.py
 * keeps the mode off for its timed region and reports its effect beside it.  At JOB level (the CLI driver on 8 x 4K frames, all eight
 * files) the mode bought nothing - 1.702 s on, 1.701 s off (profiles/r05e_driver_bench.txt): the driver's pipeline already hides the
 * small launches' gaps - so the driver leaves it off too (its --overlap turns it on; round 5 had it on by default). */
int pmp_set_overlap(pmp_ctx *ctx, int on);

/* Convolution datapath.  All three are fp32-accurate (EXPERIMENTS.md, precision study); results differ in the last bits only.
 *   PMP_PRECISION_F32    v_mfma_f32_16x16x4_f32, exact fp32 fmaf chain
 *   PMP_PRECISION_BF16X6 every fp32 operand carried as 3 bf16 terms, 6 bf16 MFMA products, fp32 accumulate
 *   PMP_PRECISION_F16X3  (default) every fp32 operand carried as 2 fp16 terms (weights pre-scaled by a power of two),
 *                        3 fp16 MFMA products, fp32 accumulate; activations beyond +-65504 would saturate: guarded, see below */
#define PMP_PRECISION_F32 0
#define PMP_PRECISION_BF16X6 1
#define PMP_PRECISION_F16X3 2
int pmp_set_precision(pmp_ctx *ctx, int mode);
int pmp_get_precision(const pmp_ctx *ctx);

/* Range guard of the f16x3 datapath.  Its activations travel as two fp16 terms, so a value beyond +-65504 is clamped when it
 * is stored (the reference's nets stay below 3e3 on 8-bit content with the trained QT weights; trained MTT weights are not
 * in the reference checkout).  Every kernel that stores such a tensor raises a per-context device flag when the clamp fires
 * (a NaN raises it too).  Every pmp_infer* call snapshots the flag behind its passes - stream-ordered, into pinned host memory -
 * and the snapshot is LOOKED AT LATER, so that the *_device entry points never stall the host: by the next call of the context
 * (polled: only snapshots that have landed), and by pmp_synchronize / pmp_get_saturation / any host-pointer call (waited for).
 *   PMP_SAT_RERUN (default)  a call whose flag fired is run AGAIN on the exact fp32 MFMA datapath (fp32 range and arithmetic) into
 *                            the same output buffers, and every post-processing call that was enqueued after it is replayed in
 *                            order (a later call whose logits are in the context's own buffers - no logit pointers passed - runs again too).
 *                            Consequence for *_device callers: outputs are FINAL once pmp_synchronize (or
 *                            pmp_get_saturation) has returned - synchronising the stream yourself is enough only if you also
 *                            know the flag stayed down; inputs and outputs must stay untouched until then.  Host-pointer entry
 *                            points return final results, as before.
 *   PMP_SAT_ERROR            the call that looks at a fired snapshot returns PMP_E_RANGE (for *_device calls that may be the
 *                            NEXT call or pmp_synchronize) and forgets the calls in flight; nothing is re-run
 *   PMP_SAT_IGNORE           no snapshots, no re-runs; the caller polls
 * pmp_get_saturation settles the calls in flight and returns 1 if any inference call of this context saturated since the last
 * pmp_clear_saturation (0 otherwise, negative on error); pmp_get_saturation_reruns counts the calls that were re-run. */
/* Activation scales of the f16x3 datapath (MTT nets).  The range guard above is a safety net; what keeps a net with large activations
 * ON the default datapath is this: the MTT nets are bias-free behind their stems, and ReLU, max-pool and the two gate products
 * (Model_QBD.py:143,150) are positively homogeneous, so a tensor can travel as true * 2^-e - exactly: a power of two commutes with
 * every rounding - if its consumer knows e.  The graph has five segments (stem..trunk_M1..trunk_M2..trunk_B1 | attention 1 | x5*att0..
 * trunk_B2 | attention 2 | x4*att1..trunk_B3) with one exponent each; the changes of scale are folded into numbers the kernels multiply
 * by anyway (the stem's output scale and biases, the out_scale of the convolution whose epilogue applies a gate, the head weights), so
 * they cost nothing per call.  The exponents are chosen when a (QT, MTT) pair is first used on the f16x3 datapath: one pass of both nets
 * over 32 built-in calibration blocks (flat, checkerboards, stripes, edges, white noise, smooth random content) on the fp32 MFMA
 * datapath records the largest |value| of every MTT tensor, and a segment whose maximum exceeds 2^12 gets the exponent that brings it
 * there (16x headroom below 65504 for content harsher than the calibration set; the attention trunks take theirs where their input is
 * built from the logits, capped at 2^-6: that input is O(1) and must stay out of fp16's subnormals - an attention trunk that needs more
 * falls to the range guard).
 * Deterministic: same weights, same exponents, on every context and rank.  Exponents of zero - the synthetic uniform MTT weights, any
 * net whose activations stay below 4096 - leave the arithmetic exactly as it was.
 * WHEN: as soon as both nets of a (component, QP) are loaded while the context is on the f16x3 datapath - inside the pmp_load_weights*
 * call that completes the pair (≈ 20 ms of host time; the pass runs on a private stream and a private 44 MB workspace, beside whatever the
 * context's stream is doing) - or, for a pair loaded under another datapath, at its first f16x3 inference call.  NOT AT ALL for an MTT
 * file whose .pmpw manifest carries "act_exp": [e0..e4] (tools/calibrate_pmpw.py writes them once per model directory): those exponents
 * are taken as they are - bounded by the reader (0..30 for the trunk segments, 0..6 for the attention segments: PMP_E_INVALID beyond), and
 * only if the manifest's "act_fp" fingerprints match the tensors of the file and of the QT partner that is (or later gets) loaded; a stale
 * manifest falls back to the calibration pass, and (re)loading a QT net drops exponents that were derived with another one.
 * pmp_debug_activation_report (below) returns the exponents and the recorded maxima. */
#define PMP_SAT_RERUN 0
#define PMP_SAT_ERROR 1
#define PMP_SAT_IGNORE 2
int pmp_set_saturation_policy(pmp_ctx *ctx, int policy);
int pmp_get_saturation(pmp_ctx *ctx);
int64_t pmp_get_saturation_reruns(const pmp_ctx *ctx);
int pmp_clear_saturation(pmp_ctx *ctx);

/* Caller keeps ownership of blob/descs; the library re-packs into its kernel layouts in device memory.
 * Every tensor the net needs must be present with the reference's shape (else PMP_E_INVALID). */
int pmp_load_weights(pmp_ctx *ctx, int net_id, int qp, const float *blob, const pmp_tensor_desc *descs,
                     int ndesc);
/* The same from the product's weight container (<Comp>_{Q,BD}_<qp>.pmpw, pmp_vvc_tip2023_amd/weights.py; tools/convert_weights.py
 * makes one from a reference .pkl): for hosts without Python, e.g. the in-process VTM hook (tools/vtm_build/pmp_hook.cpp).
 * The manifest's net and QP must match the arguments. */
int pmp_load_weights_file(pmp_ctx *ctx, int net_id, int qp, const char *path);
int pmp_has_weights(const pmp_ctx *ctx, int net_id, int qp);
/* Fingerprint of a loaded net's tensors / of a tensor set (names, shapes, float bit patterns; independent of the order and layout they
 * are handed over in; weights.fingerprint() computes the same number with numpy).  What ties a manifest's "act_exp" to the tensors it
 * was calibrated on: tools/calibrate_pmpw.py stores "act_fp": [this net's, its QT partner's] beside the exponents, and
 * pmp_load_weights_file ignores exponents whose fingerprints do not match what is loaded (stale file: the QT net or the tensors changed
 * since) in favour of a calibration pass.  Host-only arithmetic; pmp_fingerprint_tensors needs no context and no GPU. */
int pmp_weights_fingerprint(const pmp_ctx *ctx, int net_id, int qp, uint64_t *out);
int pmp_fingerprint_tensors(const float *blob, const pmp_tensor_desc *descs, int ndesc, uint64_t *out);

/* ---- inference: inference_pre_QBD (Metrics.py:387-419).  block_u/block_v are ignored for PMP_LUMA. --- */
int pmp_infer(pmp_ctx *ctx, int comp, int qp, const uint8_t *block_y, const uint8_t *block_u,
              const uint8_t *block_v, int64_t n, float *qt, float *bt, float *dire);
int pmp_infer_device(pmp_ctx *ctx, int comp, int qp, const uint8_t *d_block_y, const uint8_t *d_block_u,
                     const uint8_t *d_block_v, int64_t n, float *d_qt, float *d_bt, float *d_dire);

/* ---- post-processing: seq_post_process minus the file (Metrics.py:764-774).  qt = RAW QT logits. -------
 *      Value domain: EVERY float32 bit pattern, with the reference's results (tests/golden/g3b_m2p_range.npz, made by the reference):
 *      - depth logits of any magnitude: np.round has no clamp (Map2Partition.py:104) and neither has this in effect - the rounded
 *        depth is only compared with candidate depths 0..6, so the kernel's integer copy saturates at +-100 without a difference;
 *        the float32 error sums (:307-312) run on the raw values in numpy's summation order, overflow to inf included;
 *      - non-finite logits (a saturated datapath under PMP_SAT_IGNORE can hand them over): a NaN / +inf depth is "deeper than any
 *        candidate" (numpy's `== 0` and `< 0` are False), -inf "shallower"; a NaN direction counts as 0, +-inf as +-1; a QT leaf
 *        holding one takes the FIRST candidate leaf (Python's min() over inf / NaN errors); a NaN QT logit survives max-pool, round
 *        and clamp (Metrics.py:632), no check_square_unity rule fires on its quadrant, its region gets no edges and directions 0
 *        (set_partition_vector, :348-362: neither == nor > holds) and qt_u8 carries 0 (numpy's .astype(uint8) of NaN on x86-64). */
int pmp_postprocess(pmp_ctx *ctx, int comp, const float *qt, const float *bt, const float *dire, int64_t n,
                    uint8_t *hor, uint8_t *ver, uint8_t *qt_u8, int8_t *dire_i8);
int pmp_postprocess_device(pmp_ctx *ctx, int comp, const float *d_qt, const float *d_bt, const float *d_dire,
                           int64_t n, uint8_t *d_hor, uint8_t *d_ver, uint8_t *d_qt_u8, int8_t *d_dire_i8);

/* ---- fused: blocks in, split flags out; logits stay in HBM (qt/bt/dire may be NULL). ------------------- */
int pmp_infer_postprocess(pmp_ctx *ctx, int comp, int qp, const uint8_t *block_y, const uint8_t *block_u,
                          const uint8_t *block_v, int64_t n, uint8_t *hor, uint8_t *ver, uint8_t *qt_u8,
                          int8_t *dire_i8, float *qt, float *bt, float *dire);
int pmp_infer_postprocess_device(pmp_ctx *ctx, int comp, int qp, const uint8_t *d_block_y,
                                 const uint8_t *d_block_u, const uint8_t *d_block_v, int64_t n, uint8_t *d_hor,
                                 uint8_t *d_ver, uint8_t *d_qt_u8, int8_t *d_dire_i8, float *d_qt, float *d_bt,
                                 float *d_dire);

/* ---- the same with one packed RECORD per block, the unit the multi-GPU path gathers to the rank that writes the file
 *      (counterpart of nn.DataParallel's gather, Inference_QBD.py:223-224):
 *        u8 rec[n][PMP_RECORD_BYTES] = hor[256] | ver[256] | qt_u8[64] | dire_i8[768]
 *      written directly by the post-processing kernel (no repacking pass); d_rec must be 4-byte aligned. ---- */
#define PMP_RECORD_BYTES 1344
int pmp_postprocess_records_device(pmp_ctx *ctx, int comp, const float *d_qt, const float *d_bt, const float *d_dire,
                                   int64_t n, uint8_t *d_rec);
int pmp_infer_postprocess_records_device(pmp_ctx *ctx, int comp, int qp, const uint8_t *d_block_y, const uint8_t *d_block_u,
                                         const uint8_t *d_block_v, int64_t n, uint8_t *d_rec);

/* ---- block cutter: output_block_yuv (Inference_QBD.py:104-149).  Planes y[F][H][W], u,v[F][H/2][W/2];
 *      bitdepth 8 -> uint8 samples, 10 -> uint16 samples reduced with round-half-even(x/4), clipped to 255.
 *      Writes F*(H/64)*(W/64) blocks, frame-major then row-major; right/bottom remainders are dropped. ---- */
int pmp_cut_blocks(pmp_ctx *ctx, const void *y, const void *u, const void *v, int frames, int height, int width,
                   int bitdepth, uint8_t *block_y, uint8_t *block_u, uint8_t *block_v);
int pmp_cut_blocks_device(pmp_ctx *ctx, const void *d_y, const void *d_u, const void *d_v, int frames,
                          int height, int width, int bitdepth, uint8_t *d_block_y, uint8_t *d_block_u,
                          uint8_t *d_block_v);

/* ---- PartitionMat text file (Map2Partition.py:385-412): per frame hor, ver, qt, dire[3]; one decimal
 *      integer per line.  Host-side (file I/O); inputs are per-block arrays in block order. ---------------- */
int pmp_write_partition_file(const char *path, int frames, int height, int width, const uint8_t *hor,
                             const uint8_t *ver, const uint8_t *qt_u8, const int8_t *dire_i8);
/* Same bytes into memory: returns the byte count (or a negative error).  buf == NULL returns the exact size; a buffer of
 * exactly that size is enough (cap < size: PMP_E_INVALID, nothing beyond buf[cap-1] is ever written). */
int64_t pmp_format_partition_text(int frames, int height, int width, const uint8_t *hor, const uint8_t *ver,
                                  const uint8_t *qt_u8, const int8_t *dire_i8, char *buf, int64_t cap);

/* ---- sharded emission (the reference's serial tail is the per-value text emission, Map2Partition.py:401-412).
 *      Frames are self-contained in the file (Map2Partition.py:389-412) and inside a frame every section (hor, ver, qt, dire 0..2) is
 *      row-major, so a writer that holds `block_rows` consecutive BLOCK ROWS of one frame (block_rows * (width/64) blocks, row-major)
 *      owns six contiguous byte ranges of the file.  These two calls produce them back to back - section-major, i.e. exactly the
 *      text of a frame of height 64*block_rows - and report the exact size of every (block row, section) pair in
 *      row_section_bytes[block_rows][6] (may be NULL), from which the ranks derive their file offsets with one exclusive scan
 *      (pmp_vvc_tip2023_amd/parallel.py: section_offsets) and write concurrently with pwrite.  buf == NULL: sizes only; the return
 *      value is the total byte count.  The _records form reads packed PMP_RECORD_BYTES records, what the device path produces. ---- */
int64_t pmp_format_partition_rows(int width, int block_rows, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                                  const int8_t *dire_i8, char *buf, int64_t cap, int64_t *row_section_bytes);
int64_t pmp_format_partition_rows_records(int width, int block_rows, const uint8_t *rec, char *buf, int64_t cap,
                                          int64_t *row_section_bytes);
/* The same rows as matrices (binary side channel / in-process hand-over): hor, ver u8[16*block_rows][cols], qt u8[8*block_rows][cols/2],
 * dire i8[3][16*block_rows][cols], cols = 16*(width>>6). */
int pmp_tile_partition_rows_records(int width, int block_rows, const uint8_t *rec, uint8_t *out_hor, uint8_t *out_ver, uint8_t *out_qt,
                                    int8_t *out_dire);

/* ---- binary side channel.  Same content as the text file, laid out as the arrays the patched VTM keeps
 *      after parsing (Lib/CommonLib/Rom.h:240-248), so a consumer can mmap it instead of 645 k getline+stoi calls per frame:
 *        char magic[8] = "PMPB1\0\0\0"; int32 frames, height, width, rows (= 16*(H>>6)), cols (= 16*(W>>6)), reserved[3];
 *        then per frame:  u8 hor[rows][cols] | u8 ver[rows][cols] | u8 qt[rows/2][cols/2] | i8 dire[3][rows][cols]
 *      (frame matrices, already tiled from the per-block arrays; little-endian; 40-byte header). ---- */
int pmp_write_partition_binary(const char *path, int frames, int height, int width, const uint8_t *hor, const uint8_t *ver,
                               const uint8_t *qt_u8, const int8_t *dire_i8);

/* ---- in-process hand-over: the same frame matrices straight into caller memory, in the shapes the
 *      patched VTM allocates in EncAppCfg::parsePartitionMatrix (EncAppCfg.cpp:4270-4298; Rom.h:240-248), one component:
 *        hor, ver: u8[frames][rows][cols]   qt: u8[frames][rows/2][cols/2]   dire: i8[frames][3][rows][cols]
 *      with rows = 16*(height>>6), cols = 16*(width>>6) (4x4 luma units of the picture cropped to multiples of 64).
 *      A hook that replaces the text parser copies (or points) partitionHorMat[f][comp] etc. at these rows. ---- */
int pmp_tile_partition_maps(int frames, int height, int width, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                            const int8_t *dire_i8, uint8_t *out_hor, uint8_t *out_ver, uint8_t *out_qt, int8_t *out_dire);

/* ---- per-kernel-class timing with hipEvents on the launch stream (bench.py roofline leg). -------------- */
/* mask: bit i enables kernel class i (see pmp_ktime_name); 0 disables.  Resets the accumulators. */
int pmp_ktime_enable(pmp_ctx *ctx, uint32_t mask);
int pmp_ktime_classes(void);
const char *pmp_ktime_name(int cls);
/* Synchronises, then returns launches / total milliseconds / algorithmic FLOPs accumulated for the class. */
int pmp_ktime_get(pmp_ctx *ctx, int cls, int64_t *launches, double *ms, double *flops);

/* ---- measurement hook.  The product library ships ONE form of every convolution kernel - number 2 - and accepts nothing else
 *      here (PMP_E_INVALID): it has no process-wide kernel selector.  The forms that were built, parity-tested and measured slower
 *      than or equal to the shipped ones (1, 3..9; bit-identical results) and the timing-only builds (10 and above; WRONG results)
 *      exist only in the measurement library tools/abl/libpmp_hip_abl.so (`make -C tools/abl`), where this call selects them process-wide for
 *      in-process A/B timing (tools/conv_ab.py, tools/variants_agree.py; the list is in tools/abl/conv_f16x3.hip, the numbers in EXPERIMENTS.md). ---- */
int pmp_debug_set_conv_variant(int variant);

/* ---- measurement hook (f16x3 datapath): run the 3x3 64->64 convolutions - 55 % of the luma step - in the Winograd F(2,3)-along-x
 *      form (conv_f16x3_wx.hip: 1.5x fewer MFMAs, fp32-equivalent logits within the 1e-3 tolerance, not bit-identical to the direct
 *      form).  It did not beat the direct kernels (EXPERIMENTS.md, profiles/r03_notes.txt), so like the other forms that lost their
 *      A/B it exists in the measurement library tools/abl/libpmp_hip_abl.so only (`make -C tools/abl`; tools/wx_probe.py, tools/wx_ablate.py): the
 *      product library accepts on = 0 and answers PMP_E_INVALID to anything else. ---- */
int pmp_debug_set_winograd(pmp_ctx *ctx, int on);

/* ---- test / A-B hook (f16x3 datapath, per context): launch fusion.  on = 1 (default): (a) the 16x16-resolution tails of the nets - trunk_B1/B2 +
 *      heads + attention 1 of the MTT nets, resblock_q3 .. conv_q2 of the QT nets - run as two (QT) / three (MTT) launches per net with the activations of
 *      a ResidualBlock resident in LDS, two blocks per CU (chain16.hip), and (b) the ResidualBlocks with <= 32 output channels at 32x32 - trunk_B3.1, trunk_B3.2 (+ pool),
 *      trunk_Att2.0 of the MTT nets - as one launch per block with the intermediate in LDS (rbfuse32.hip).  on = 0: launch per layer, as the
 *      other two datapaths always do; on = 2: (a) only; on = 3: (b) only.  Results are BIT-IDENTICAL in all four settings
 *      (tests/test_gpu_parity.py::test_fused_16x16_tails_are_bit_identical); only the launch count (68 -> 40 per luma pass) and the time
 *      differ.  Settles the calls in flight first. ---- */
int pmp_debug_set_fusion(pmp_ctx *ctx, int on);

/* ---- test / diagnosis hook: the f16x3 activation scales of the MTT net of (comp, qp) (see "Activation scales" above) and the calibration
 *      record behind them.  Runs the calibration now if the pair has not been used on the f16x3 datapath yet (both nets must be loaded).
 *      exps[5]: the segment exponents; seg_amax[5]: the largest |value| seen in each segment on the calibration blocks (true scale);
 *      buf (may be NULL): one text line per recorded tensor, "<name> <segment> <max |value|>\n", in launch order ("<block>.t" is the
 *      intermediate of a ResidualBlock).  Returns the number of recorded tensors or a negative error. ---- */
int pmp_debug_activation_report(pmp_ctx *ctx, int comp, int qp, int exps[5], float seg_amax[5], char *buf, int64_t cap);
/* ---- test hook: on = 0 runs the f16x3 datapath with exponents of zero whatever the calibration chose (how the range-guard tests still
 *      drive activations out of the fp16 range: with the scales on, their power-of-two stress weights simply get larger exponents);
 *      on = 1 (default) uses the calibrated exponents.  Settles the calls in flight first. ---- */
int pmp_debug_set_activation_scales(pmp_ctx *ctx, int on);

/* ---- test hook (host only, no GPU needed): the f16x3 weight packing of one OIHW conv tensor (conv_f16x3.hip).
 *      Writes the power-of-two exponent k of the scale S = 2^k to *scale_exp and, if out != NULL, the packed stream
 *      [K-step][2 splits][cout_pad/16][64 lanes][8] of fp16 bit patterns (h0, h1 with h0 + h1 ~= S*w) to out.
 *      Returns the number of uint16 elements of the stream (> cap: nothing written), or a negative error. ---- */
int64_t pmp_debug_pack_f16x3(const float *w, int cout, int cin, int k, uint16_t *out, int64_t cap, int *scale_exp);

/* ---- test hook (host only): parse a .pmpw container; reports its net id (-1 if unknown), QP, tensor count, payload floats
 *      and the sum of all tensor elements. ---- */
int pmp_debug_read_weights_file(const char *path, int *net_id, int *qp, int *ntensors, int64_t *nfloats, double *checksum);

/* ---- measurement hook: one convolution layer on random data, both datapaths.  Runs conv KxK Cin->Cout (+ReLU) on
 *      n blocks of HxW with the fp32-MFMA kernel and with the context's split kernel (f16x3 or bf16x6; bf16x6 if the
 *      context is in fp32 mode), `iters` timed launches each.
 *      Outputs: average milliseconds per launch and max |fp32 - split| / max |fp32| over the whole output. ---- */
int pmp_debug_conv_bench(pmp_ctx *ctx, int n, int h, int w, int cin, int cout, int k, int iters, double *ms_f32,
                         double *ms_x6, double *max_abs_diff, double *max_abs_ref);

#ifdef __cplusplus
}
#endif
#endif /* PMP_H */
