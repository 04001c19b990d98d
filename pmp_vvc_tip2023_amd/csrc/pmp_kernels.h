// Internal launch interface between the host-side graph (nets.cpp / pmp_api.cpp) and the HIP kernels.
//
// Activation layout in HBM ("blocked channels-last"):  act[n][c/16][y][x][c%16]  float32, channels padded to a
// multiple of 16 with zeros.  A 16-channel group of one tile row is contiguous (16 px * 64 B = 1 KiB), which is
// exactly what one wave loads or stores per instruction in the MFMA conv kernel (16 B per lane, 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abl_types.h"   // hooks/ in the product build (empty structs), abl/ in the measurement library

namespace pmp {

// ------------------------------------------------------------------------------------------------ conv (MFMA)
// out = epilogue( conv_KHxKW(x, w) [+ conv_1x1(x_sc, w_sc)] [+ res] ), stride 1, zero padding K/2.
// epilogue: optional ReLU, optional multiply by `gate`, optional 2x2 max-pool.  H, W multiples of 16;
// Cin, Csc, Cout multiples of 16 (Cout <= 64).  Weight packing: pack_conv_mfma() in weights_pack.cpp.
struct ConvMfmaArgs {
    const float *x;     // [N][Cin/16][H][W][16]
    const float *w;     // packed [Cin/16][KH*KW][Cout/16][64 lanes][4]
    const float *x_sc;  // optional second source for the 1x1 shortcut conv, [N][Csc/16][H][W][16]
    const float *w_sc;  // packed [Csc/16][1][Cout/16][64][4]
    const float *res;   // optional identity residual [N][Cout/16][H][W][16]
    const float *gate;  // optional gate (attention multiply), same shape as the un-pooled output
    float *out;         // [N][Cout/16][H(/2)][W(/2)][16]
    int N, H, W, Cin, Csc, Cout, KH, KW;
    int relu, pool;
};
hipError_t launch_conv_mfma(hipStream_t s, const ConvMfmaArgs &a);

// ------------------------------------------------------------------------------------------------ conv (bf16 x 6)
// Same operation on "split-3" activations (three bf16 planes per tensor, conv_bf16x6.hip).  *_stride = elements between
// the planes of the respective tensor.  Output goes to split-3 planes (`out`) or, if out_f32 != nullptr, to a plain
// fp32 blocked tensor (for consumers that are not MFMA convs).
struct ConvX6Args {
    const unsigned short *x; size_t x_stride;
    const unsigned short *w;                 // packed [Cin/16][ceil(taps/2)][3 splits][Cout/16][64 lanes][8 bf16]
    const unsigned short *x_sc; size_t sc_stride; const unsigned short *w_sc;
    const unsigned short *res; size_t res_stride;
    const unsigned short *gate; size_t gate_stride;
    unsigned short *out; size_t out_stride;
    float *out_f32;
    int N, H, W, Cin, Csc, Cout, KH, KW;
    int relu, pool;
    float out_scale;           // f16x3 only: 1/S of the power-of-two weight scaling (conv_f16x3.hip)
    unsigned *sat;             // f16x3 only: sticky flag raised when a stored activation exceeded the fp16 range (may be nullptr)
    AblConvArgs abl;           // empty in the product library
};
hipError_t launch_conv_x6(hipStream_t s, const ConvX6Args &a);
hipError_t launch_f32_to_split3(hipStream_t s, const float *x, unsigned short *out, size_t n, size_t plane_stride);
hipError_t launch_split3_to_f32(hipStream_t s, const unsigned short *x, float *out, size_t n, size_t plane_stride);
// Same operation on "split-2" activations (two fp16 planes) with three fp16 MFMA products per term (conv_f16x3.hip).
// Weights packed [K-step][2 splits][Cout/16][64 lanes][8 fp16], pre-multiplied by 1/out_scale.
hipError_t launch_conv_h2(hipStream_t s, const ConvX6Args &a);
hipError_t launch_f32_to_split2(hipStream_t s, const float *x, unsigned short *out, size_t n, size_t plane_stride, unsigned *sat = nullptr);
hipError_t launch_split2_to_f32(hipStream_t s, const unsigned short *x, float *out, size_t n, size_t plane_stride);

// ------------------------------------------------------------------------------------------------ 16x16 tails, LDS-resident
// chain16.hip (f16x3 datapath): the layers of a net that run at 16x16 resolution - where a block is a single tile - with
// every activation of a ResidualBlock resident in LDS, two blocks per CU (round 6: three / two kernels per tail, tail16_dev.h); bit-identical to the launch-per-layer path.  Weights: a ResidualBlock's pack_h2 streams
// (RBWeights::w0h / w2h / wsch) with 1/S of its two passes (s0 = 2^-k0, s2 = 2^-k2).
struct Chain16RB { const unsigned short *w0, *w2, *wsc; float s0, s2; };
// MTT nets: trunk_B1 + conv_B1 -> out0; cat[up2(q), out0] -> trunk_Att1, x x5 -> trunk_B2 + conv_B2 -> out1 (accumulated)
struct Chain16MsbdArgs {
    const unsigned short *x5; size_t x5_stride;     // [N][4][16][16][16] split-2
    unsigned short *xb; size_t xb_stride;           // scratch, same layout: x5 * att0, handed from the attention kernel to trunk_B2's through L2
    const float *qt; float *bt, *dire;              // raw QT logits [N][64]; outputs [N][3][256], layers 0 and 1
    Chain16RB b1[3], att[2], b2[3];
    const float *head_w[2], *head_b[2];             // conv_B1, conv_B2: [9][8][2] + [2]
    unsigned *sat;
    int N;
    float att_scale;                                // f16x3 activation scale of attention segment 1 (2^-e1; the attention input is built in the kernel)
};
hipError_t launch_msbd_branch16(hipStream_t s, const Chain16MsbdArgs &a);
// QT nets: resblock_q3 -> multi-scale pool -> resblock_q4 -> resblock_q5 + max_pool2d(2) -> resblock_q6 (8x8, fp32 direct) -> conv_q2
struct Chain16QtArgs {
    const unsigned short *x4; size_t x4_stride;     // [N][4][16][16][16] split-2: resblock_q2's pooled output
    float *x5;                                      // scratch [N][2][16][16][16] fp32: resblock_q3's output (the multi-scale pool reads fp32)
    float *qt;                                      // [N][64]
    Chain16RB q3, q4, q5;
    const float *d_w0, *d_w2, *d_wsc;               // resblock_q6, plain fp32 packing (pack_plain)
    const float *head_w, *head_b;                   // conv_q2: [9][8][1] + [1]
    unsigned *sat;
    int N;
};
hipError_t launch_qt_tail16(hipStream_t s, const Chain16QtArgs &a);

// ------------------------------------------------------------------------------------------------ a ResidualBlock at 32x32, one launch
// rbfuse32.hip (f16x3 datapath): conv3x3 + ReLU, conv3x3 + 1x1 shortcut + ReLU [+ 2x2 max-pool] of a block with <= 32 output channels,
// the intermediate resident in LDS; bit-identical to two launches of conv_f16x3.hip.  Shapes: (cin_groups, cout_groups, pool_f32) =
// (2, 1, 0) trunk_B3.1, (1, 1, 1) trunk_B3.2, (1, 2, 0) trunk_Att2.0.
struct RbFuse32Args {
    const unsigned short *x; size_t x_stride;        // [N][cin_groups][H][W][16] split-2
    const unsigned short *w0, *w2, *wsc;             // the block's pack_h2 streams (RBWeights::w0h / w2h / wsch)
    float s0, s2;                                    // 1/S of the two passes
    unsigned short *out; size_t out_stride;          // [N][cout_groups][H][W][16] split-2 ...
    float *out_f32;                                  // ... or, with pool_f32, [N][cout_groups][H/2][W/2][16] plain fp32
    unsigned *sat;
    int N, H, W, cin_groups, cout_groups, pool_f32;
    // x == nullptr (trunk_Att2.0 only): the block's input is the attention input cat[up(q), up(bt[att_layer]), up(dire[att_layer])], built in
    // the kernel from the logits q [N][64], bt / dire [N][3][256] (launch_att_input's arithmetic)
    const float *q, *bt, *dire; int att_layer;
    float att_scale;                                 // ... times the f16x3 activation scale of its segment (2^-e3)
};
hipError_t launch_rbfuse32(hipStream_t s, const RbFuse32Args &a);

// ------------------------------------------------------------------------------------------------ stems
// First layers straight from the u8 blocks (Model_QBD.py:79-80, :130-135, :177-178, :228-233).
// luma: block_y u8[N][68][68];  chroma: + block_u/v u8[N][34][34], plane 0 = 2x2 max-pool of block_y
// (Inference_QBD.py:196-200).  msbd: adds the plane built from the raw QT logits q f32[N][8][8]
// (nearest x8 / x4 upsample, zero pad top/left by 4 / 2).  Output: 32 channels, [N][2][S][S][16], S = 64 / 32.
struct StemArgs {
    const uint8_t *by, *bu, *bv;
    const float *q;      // msbd only
    const float *w;      // packed per conv: [conv][tap][cin][cout], see pack_stem()
    const float *bias;   // [32]
    float *out;
    int N;
    unsigned short *out_s3; size_t s3_stride;   // if out_s3 != nullptr the 32 channels are written as split planes ...
    int fmt;                                    // ... of format 1 (three bf16) or 2 (two fp16), see split3.h
    const unsigned short *wh; float out_scale;  // fmt 2: fp16 fragment stream (pack_stem_h2) and 1/scale -> the MFMA stem
    unsigned *sat;                              // fmt 2: saturation flag (see ConvX6Args)
};
hipError_t launch_stem(hipStream_t s, bool luma, bool msbd, const StemArgs &a);

// ------------------------------------------------------------------------------------------------ small direct conv
// Generic direct convolution for the tiny tail layers (8x8 maps, 8-channel trunks):
// out = [relu]( conv(x, w) [+ conv1x1(x_sc, w_sc)] [+ res] [+ bias] ).  Weights plain [tap][cin][cout_real].
struct ConvDirectArgs {
    const float *x; const float *w;
    const float *x_sc; const float *w_sc;
    const float *res; const float *bias;
    float *out;
    int N, H, W, Cin, CinPad, Csc, CscPad, Cout, CoutPad, KH, KW, relu;
};
hipError_t launch_conv_direct(hipStream_t s, const ConvDirectArgs &a);

// Heads (Model_QBD.py:91,:139,:145-146,:152-153): 3x3, 8 -> 1 (QT) or 8 -> 2 (MTT layer k), bias, no activation.
// QT: qt[n][y][x].  MTT layer k: bt[n][k][y][x] = ch0 (+ bt[n][k-1][y][x] if k > 0), dire[n][k][y][x] = ch1.
struct HeadArgs {
    const float *x;   // [N][1][S][S][16], channels 0..7 used
    const float *w;   // [9][8][cout]
    const float *bias;
    float *qt, *bt, *dire;
    int N, S, layer;  // layer < 0: QT head
};
hipError_t launch_head(hipStream_t s, const HeadArgs &a);

// x5 [N][2][16][16][16] -> cat[x5, up2(mp2), up4(mp4), up8(mp8)] [N][8][16][16][16]  (Model_QBD.py:84-87)
hipError_t launch_multipool_concat(hipStream_t s, const float *x5, float *x6, int N, unsigned short *x6_s3 = nullptr,
                                   size_t s3_stride = 0, int fmt = 1, unsigned *sat = nullptr);

// Attention trunk input (Model_QBD.py:140, :147): [N][1][S][S][16] with ch0 = up(q) (S/8 nearest), ch1 = bt[n][layer],
// ch2 = dire[n][layer] (both 16x16, nearest-upsampled to S), channels 3..15 zero.
hipError_t launch_att_input(hipStream_t s, const float *q, const float *bt, const float *dire, int layer, float *out,
                            int N, int S, unsigned short *out_s3 = nullptr, size_t s3_stride = 0, int fmt = 1, unsigned *sat = nullptr,
                            float scale = 1.f);     // scale: the f16x3 activation scale of the attention segment (a power of two)

// Largest |value| of an fp32 tensor (n a multiple of 4), atomicMax'ed into *slot as the bit pattern of the magnitude: the calibration
// pass of the f16x3 activation scales (calibrate.cpp: calibrate_mtt).
hipError_t launch_amax_f32(hipStream_t s, const float *x, size_t n, unsigned *slot);

// ------------------------------------------------------------------------------------------------ post-processing
// eli_structual_error + Map_to_Partition, one wavefront per block.  qt raw logits [N][64]; bt, dire [N][3][256].
// record_stride != 0: the four outputs are fields of one packed record per block (bytes between blocks), else dense arrays.
hipError_t launch_postprocess(hipStream_t s, const float *qt, const float *bt, const float *dire, int64_t N,
                              int chroma_factor, uint8_t *hor, uint8_t *ver, uint8_t *qt_u8, int8_t *dire_i8, int record_stride = 0);

// Block cutter (Inference_QBD.py:104-149).
hipError_t launch_cut_blocks(hipStream_t s, const void *y, const void *u, const void *v, int F, int H, int W,
                             int bitdepth, uint8_t *by, uint8_t *bu, uint8_t *bv);

}  // namespace pmp
