// Host-side internals of libpmp_hip.so: context, packed weights, the four forward graphs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <deque>
#include <functional>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pmp.h"
#include "pmp_hostonly.h"
#include "pmp_kernels.h"

namespace pmp {

enum KClass { K_CONV3_64 = 0, K_CONV5_64, K_CONV_OTHER, K_STEM, K_SMALL, K_POST, K_NCLASS };

struct DevBuf {  // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
};

// ResidualBlock weights (Model_QBD.py:23-44), packed for the kernel that runs the block.
struct RBWeights {
    int cin = 0, cout = 0, k = 0;      // real dims
    int cin_pad = 0, cout_pad = 0;
    float *w0 = nullptr, *w2 = nullptr, *wsc = nullptr;  // fp32 MFMA packing (or direct packing when `direct`)
    unsigned short *w0x = nullptr, *w2x = nullptr, *wscx = nullptr;  // bf16x6 split packing (conv_bf16x6.hip)
    unsigned short *w0h = nullptr, *w2h = nullptr, *wsch = nullptr;  // f16x3 split packing (conv_f16x3.hip), scaled by
    int k0 = 0, k2 = 0;                                               // 2^k0 (first conv) / 2^k2 (second conv + shortcut)
    AblRB abl;                         // empty in the product library
    bool direct = false;               // 8x8 layers run on the direct kernel
    bool has_sc = false;               // 1x1 shortcut conv (cin != cout); the packed pointers exist per datapath, this flag always
};

struct NetWeights {
    bool loaded = false;
    int net_id = -1;
    unsigned packed = 0;                               // bit p: the formats of datapath p (PMP_PRECISION_*) are on the device
    std::vector<float> host;                           // the caller's fp32 tensors (OIHW), kept for the datapaths packed later
    std::vector<std::string> names;
    std::vector<pmp_tensor_desc> descs;                // offsets into `host`, names into `names`
    std::map<std::string, RBWeights> rb;
    float *stem_w = nullptr, *stem_b = nullptr;        // packed stem convs + 32 biases
    unsigned short *stem_wh = nullptr; int stem_k = 0; // f16x3 MFMA stem: fragment stream, scaled by 2^stem_k
    float *head_w[3] = {nullptr, nullptr, nullptr};    // [9][8][cout]
    float *head_b[3] = {nullptr, nullptr, nullptr};
    // f16x3 ACTIVATION SCALES (MTT nets; include/pmp.h).  The net is bias-free behind its stems and ReLU, max-pool and the gate product are
    // positively homogeneous, so a tensor may travel as true * 2^-e - exactly, a power of two commutes with every rounding - as long as
    // whoever consumes it knows e.  One exponent per SEGMENT of the graph (Model_QBD.py:127-155):
    //   0  stem .. trunk_M1 .. trunk_M2 .. trunk_B1          1  attention trunk 1 (input built from logits)
    //   2  x5 * att0 .. trunk_B2                              3  attention trunk 2                  4  x4 * att1 .. trunk_B3
    // and the changes of scale cost nothing at run time: 2^-e0 is folded into the stem's output scale and biases (stem_b_h), 2^-e1 / 2^-e3
    // into the attention inputs where they are built, the step at a gate product into the out_scale of the convolution whose epilogue
    // multiplies (nets.cpp), the way back into the head weights (head_w_h = head_w * 2^e).  The exponents come from a calibration pass on the library's own extreme-content blocks, run once when
    // the net is first used on the f16x3 datapath (calibrate.cpp: calibrate_mtt); all zero = the arithmetic of a net without scales, bit for bit.
    int act_exp[5] = {0, 0, 0, 0, 0};
    bool calibrated = false;
    uint64_t fp = 0;                                   // fingerprint_tensors() of `host` (pmp_weights_fingerprint)
    bool act_from_file = false;                        // the exponents came from a manifest ...
    bool act_fp_known = false; uint64_t act_qt_fp = 0; // ... that says which QT partner they were calibrated with
    float *stem_b_h = nullptr;                         // f16x3: stem biases * 2^-act_exp[0]
    float *head_w_h[3] = {nullptr, nullptr, nullptr};  // f16x3: head weights * 2^act_exp[{0, 2, 4}]
    std::vector<std::string> cal_names;                // calibration record: tensors in launch order ...
    std::vector<int> cal_seg;                          // ... their segment ...
    std::vector<float> cal_amax;                       // ... and their largest |value| (true scale) on the calibration blocks
    std::vector<void *> allocs;
};

// Activation workspace: a first-fit free-list allocator over one device buffer.  Every forward runs twice - a measuring
// pass (no launches) that replays the graph's alloc/release sequence to find the peak, then the real pass, which makes the
// same calls and therefore gets the same offsets.  All launches of a context go to one stream in order, so a tensor's
// bytes may be handed out again as soon as its last consumer has been ENQUEUED (Graph::release).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, top = 0, peak = 0;
    bool measuring = false;
    std::vector<std::pair<size_t, size_t>> holes;   // (offset, bytes), sorted by offset, coalesced
    void reset() { top = 0; peak = 0; holes.clear(); }
    size_t take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        size_t best = holes.size();
        for (size_t i = 0; i < holes.size(); ++i)      // best fit: the smallest hole that is large enough
            if (holes[i].second >= bytes && (best == holes.size() || holes[i].second < holes[best].second)) best = i;
        size_t off;
        if (best != holes.size()) {
            off = holes[best].first;
            if (holes[best].second == bytes) holes.erase(holes.begin() + best);
            else { holes[best].first += bytes; holes[best].second -= bytes; }
        } else {
            off = top;
            top += bytes;
            if (top > peak) peak = top;
        }
        return off;
    }
    void give(size_t off, size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        size_t i = 0;
        while (i < holes.size() && holes[i].first < off) ++i;
        holes.insert(holes.begin() + i, std::make_pair(off, bytes));
        if (i + 1 < holes.size() && holes[i].first + holes[i].second == holes[i + 1].first) { holes[i].second += holes[i + 1].second; holes.erase(holes.begin() + i + 1); }
        if (i > 0 && holes[i - 1].first + holes[i - 1].second == holes[i].first) { holes[i - 1].second += holes[i].second; holes.erase(holes.begin() + i); }
        if (!holes.empty() && holes.back().first + holes.back().second == top) { top = holes.back().first; holes.pop_back(); }
    }
    float *ptr(size_t off) const { return measuring ? nullptr : reinterpret_cast<float *>(base + off); }
};

struct KTimeRec { hipEvent_t a, b; double flops; };

// f16x3 range guard, deferred (include/pmp.h): every inference call snapshots the device flag into a pinned host word behind its
// passes and records an event; the word is read when the event has completed - at a later call (polled), at pmp_synchronize /
// pmp_get_saturation or at the end of a host-pointer call (waited for).  A call whose flag fired is run again on the fp32 MFMA
// datapath and every post-processing call enqueued after it is replayed, in order, on the same buffers; later inference calls whose
// logits live in the context's own buffers (which the re-run has overwritten) run again too.
struct PendingCall {
    bool infer;                       // inference (has a flag snapshot) or a post-processing call recorded for replay
    bool ctx_logits;                  // infer: its logits are in c->d_logit, shared by every call that passes no logit pointers
    hipEvent_t ev;                    // infer: completes when the snapshot has landed
    unsigned *slot;                   // infer: pinned host word
    std::function<int(bool)> rerun;   // infer: the same call, on fp32 MFMA if its flag fired;  post: the same call again
};

}  // namespace pmp

struct pmp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int chunk = 4096;   // blocks per pass: the 16x16-resolution layers need >= 4096 tiles to fill 256 CUs x 3 workgroups evenly (+2.5 % over 1024)
    int precision = 2;                     // 0: fp32 MFMA, 1: bf16x6 split, 2: f16x3 split (default; both splits fp32-equivalent)
    int fuse16 = 1;                        // f16x3: run the 16x16-resolution tails LDS-resident (chain16.hip: two / three launches per net); 0 = launch per layer (pmp_debug_set_fusion: A/B and the bit-identity tests)
    int act_scales = 1;                    // f16x3: use the MTT nets' calibrated activation scales (NetWeights::act_exp); 0 = exponents of zero (pmp_debug_set_activation_scales: the range-guard tests)
    int fuse32 = 1;                        // f16x3: trunk_B3.1 / B3.2 / Att2.0 (32x32, <= 32 output channels) as one launch per ResidualBlock (rbfuse32.hip); same hook
    std::string err;
    // f16x3 range guard (include/pmp.h, pmp_set_saturation_policy): device word raised by every kernel that clamps a stored activation.
    // A 256-byte block: word 0 is the flag, bytes 64.. stay zero (the zero line of conv_f16x3_t32.hip's halo DMA)
    unsigned *d_sat = nullptr;
    unsigned *h_sat = nullptr;             // PMP_SAT_SLOTS pinned host words: flag snapshots of the calls still in flight
    uint64_t sat_seq = 0;
    std::deque<pmp::PendingCall> pending;  // calls whose flag has not been looked at yet (+ the post-processing calls after them)
    pmp::AblCtx abl;                       // empty in the product library
    int sat_policy = PMP_SAT_RERUN;
    int sat_seen = 0;                      // sticky: some inference call since pmp_clear_saturation saturated
    int64_t sat_reruns = 0;                // calls re-run on the bf16x6 datapath
    std::map<int, pmp::NetWeights> nets;  // key = net_id * 100 + qp
    pmp::Arena arena;
    pmp::DevBuf ws;                        // activation workspace (its own, or a larger one parked by a destroyed context)
    pmp::DevBuf ws2;                       // second workspace: the passes of odd chunks on `stream2` (overlap mode)
    hipStream_t stream2 = nullptr;         // created on first use
    int overlap = 0;                       // two chunks in flight on two streams (PMP_OVERLAP=1 in the environment at pmp_create)
    size_t ws_need = 0;                    // what the largest pass so far needed of it (pmp_get_workspace_bytes)
    pmp::DevBuf d_in[3], d_logit[3], d_out[4], d_frames[3];  // staging for the host-pointer entry points
    // calibration of the f16x3 activation scales (NetWeights::act_exp): while cal_on, the graph (nets.cpp, running on the fp32 datapath) folds
    // the largest |value| of every tensor it produces into d_cal[slot] and logs (name, segment) per slot
    int cal_on = 0;
    unsigned *d_cal = nullptr;             // PMP_CAL_SLOTS device words
    std::vector<std::pair<std::string, int>> cal_log;
    pmp::DevBuf d_calbuf;                  // calibration blocks and their logits
    hipStream_t cal_stream = nullptr;      // calibration runs on its own stream and workspace: it neither waits for the passes in flight on the
    pmp::DevBuf ws_cal;                    // context's stream nor touches their workspace (created on first use; 44 MB for 16-block fp32 passes)
    // kernel-class timing
    uint32_t kmask = 0;
    std::vector<pmp::KTimeRec> krec[pmp::K_NCLASS];
    std::vector<hipEvent_t> event_pool;
    int64_t klaunch[pmp::K_NCLASS] = {0};
    double kms[pmp::K_NCLASS] = {0}, kflops[pmp::K_NCLASS] = {0};
};

namespace pmp {

int set_err(pmp_ctx *c, int code, const std::string &msg);
int hip_fail(pmp_ctx *c, hipError_t e, const char *what);

// weights_pack.cpp
int load_net_weights(pmp_ctx *c, int net_id, int qp, const float *blob, const pmp_tensor_desc *descs, int ndesc);
int ensure_datapath(pmp_ctx *c, NetWeights &w, int precision);   // packs the formats of `precision` if the net does not hold them yet
void free_net_weights(NetWeights &w);
// f16x3 activation scales: stores exps in w.act_exp and (re)builds the scaled stem biases and head weights on the device
int set_activation_scales(pmp_ctx *c, NetWeights &w, const int exps[5]);
constexpr int PMP_CAL_SLOTS = 128;

// pmp_api.cpp internals that calibrate.cpp shares
int ensure(pmp_ctx *c, DevBuf &b, size_t bytes);                    // grow-only device buffer (large workspaces come from the parked pool first)
NetWeights *find_net(pmp_ctx *c, int net_id, int qp);               // loaded weights of (net, qp) or nullptr
int run_graph_fn(pmp_ctx *c, const std::function<int()> &fwd);      // a forward graph twice: measuring pass, then the real one in c->ws
int settle(pmp_ctx *c);                                             // everything asked of the context so far is done and final

// calibrate.cpp
int calibrate_mtt(pmp_ctx *c, bool luma, NetWeights &wq, NetWeights &wb);
void qt_partner_changed(pmp_ctx *c, int qt_net_id, int qp);         // a QT net was (re)loaded: its MTT partner's exponents may be stale
int calibrate_if_ready(pmp_ctx *c, int net_id, int qp);             // a (QT, MTT) pair that has just become complete, on the f16x3 datapath

// nets.cpp: forward graphs on device pointers (n <= chunk); all launches go to c->stream.
int forward_q(pmp_ctx *c, bool luma, const NetWeights &w, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
              int n, float *qt);
int forward_msbd(pmp_ctx *c, bool luma, const NetWeights &w, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
                 const float *qt, int n, float *bt, float *dire);

// timing hooks used by nets.cpp
struct KScope {
    pmp_ctx *c; int cls; bool on; hipEvent_t a, b; double flops;
    KScope(pmp_ctx *c, int cls, double flops);
    ~KScope();
};

}  // namespace pmp

#include "abl_hooks.h"   // hooks/ in the product build (no-op inlines), abl/ in the measurement library
