// HRNet plan builder + executor behind the sncal_hrnet_* C ABI.
//
// Mirrors the topology of HighResolutionNet (/root/reference/src/models/hrnet/hrnet.py:255-355 for the
// construction order / state-dict names, :437-511 for the forward) and of the line network
// (/root/reference/src/models/line/hrnet.py:30-249).  The network is lowered once into a flat list of
// ops over NHWC tensors:
//     INPUT   NCHW fp32 frames -> NHWC (channel-padded to one 16-byte k-group)
//     CONV    MFMA implicit-GEMM conv with folded BN, optional residual + ReLU   (conv.hpp)
//     UPADD   out = [relu](base + sum bilinear_up(src_i)); also used to write upsampled branches into a
//             channel slice of the head's concat tensor                             (ops.hip)
//     SOFTMAX NHWC fp32 logits -> NCHW fp32 (log-)softmax heatmaps
//     DECODE  D1 keypoint decode (decode.hip)
// Tensors get offsets inside one caller-provided workspace from a lifetime-based first-fit allocator, so a
// forward is a fixed sequence of kernel launches with no allocation.  Frames are processed in sub-batches
// (SNCAL_SUBBATCH, default 64) so that the activations of a sub-batch stay Infinity-Cache sized.
#include "common.hpp"
#include "conv.hpp"
#include "ops.hpp"
#include "head.hpp"
#include "bblock.hpp"
#include "bblockx3.hpp"
#include "bneckx3.hpp"
#include "x3.hpp"
#include "conv_tt.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

extern "C" int sncal_heatmap_decode(const float*, int, int, int, int, int, int, float*, void*);

namespace sncal {

struct ConvLayer {
    std::string name, bn;
    int cin, cout, k, stride;
    bool bias;
    // host weights (BN folded on set)
    std::vector<float> w, scale, shift;
    bool is_set = false;
    // packing / dispatch
    int cin_phys = 0, mi = 0, g = 0, chunks = 0, nblk = 0, cout_frags = 0;
    void* d_w = nullptr;
    float* d_bias = nullptr;
    void* d_w_tt = nullptr;     // second packing, for the two-team kernel (conv_tt.hip): 32 x 32 x 16 MFMA fragment order
    // C5 path: e4m3 weights of the same layer for conv_tt_kernel<true> (one scale per output channel), the scales, and
    // oscale = (calibrated scale of the layer's input tensor) x (weight scale of the channel)
    void* d_w8 = nullptr;
    std::vector<float> wscale;
    float* d_oscale = nullptr;
    int stage = 0;              // 2..4 for model.stageN.* layers, 0 otherwise
    bool fp8_on = false;
    void* d_w_x3 = nullptr;     // bf16x3 engine: hi / lo split weights in the two-team kernel's fragment order (16-channel stages)
    int x3_blk = TT_COUT;       // ... packed in output-channel blocks of 96 (tile 96 x 8 x 32) or, for widths that are no multiple of 96, 64 (64 x 12 x 32)
    bool x3_on = false;
    void* d_w_bbx = nullptr;    // bf16x3 engine, 48 -> 48 3x3 layers: pair-step packing of the fused BasicBlock (bblockx3.hip)
    void* d_w_bnp = nullptr;    // split engines, layer1's 1x1 layers (64 -> 256, 256 -> 64): A fragments of the fused Bottleneck seam (bneckx3.hip)
    // internal layers of the fused head: t_i = W0[:, col_off : col_off + cin] . branch_i  (derived at finalize)
    bool derived = false;
    int col_off = 0;
    bool derived_shift = false;     // the slice that also carries last_layer.0's folded-BN shift (split head: the direct tensor's)
};

enum OpType { OP_INPUT, OP_CONV, OP_UPADD, OP_SOFTMAX, OP_DECODE, OP_HEAD };
enum OpGroup { GRP_ALL = 0, GRP_UNFUSED = 1, GRP_FUSED = 2, GRP_SPLIT = 3 };   // head variants living side by side in the plan

struct Op {
    OpType type;
    int conv = -1;
    int in = -1, out = -1, res = -1;
    bool relu = false;
    int out_coff = 0;
    bool out_f32 = false;
    bool res_twin = false;       // bf16x3: the residual is read from res's split twin (set by layout())
    int base = -1, srcs[4] = {-1, -1, -1, -1}, nsrc = 0;
    int dims_from = -1, dims_mul = 1;     // UPADD without base: out dims = dims(dims_from) * dims_mul
    int group = GRP_ALL;
    int launch_group = -1;                // >= 0: independent convs that may share one grouped launch (consecutive ops)
    bool shared_in = false;               // ... and all members read the SAME input tensor with stride 2 (conv_shared_s2_kernel)
    int head_direct = -1, head_src[HEAD_MAX_SRC] = {-1, -1, -1, -1, -1}, head_nsrc = 0;   // OP_HEAD
    int head_fold[HEAD_MAX_FOLD] = {-1, -1}, head_nfold = 0;                              // OP_HEAD: branches folded into stage-1 K
};

struct Tensor {
    int C = 0;
    bool f32 = false;         // fp32 storage regardless of the net dtype (logits / heat)
    bool external_heat = false;
    bool fp8 = false;            // e4m3 twin (1 byte per element) of a bf16 tensor, input of an fp8 convolution
    bool split = false;          // bf16x3 engine: split twin ([16 hi | 16 lo] bf16 per 16-channel group = 4 bytes per element) of an fp32 tensor
    int twin = -1;               // index of this tensor's fp8 twin, if any
    float scale = 0.f;           // calibrated per-tensor scale of the twin: amax / 448
    int first = -1, last = -1;   // producing / last consuming op, as the allocator sees them (extended over launch groups and fusable pairs)
    int last_read = -1;          // the op that really reads the tensor last (what a fusion's "nobody else reads it" test asks)
    // per-run
    int H = 0, W = 0;
    size_t offset = 0, bytes = 0;
};

}  // namespace sncal

using namespace sncal;

struct sncal_hrnet {
    sncal_hrnet_desc desc;
    int dtype;
    int ge;          // elements per 16-byte k-group
    int esize;
    std::vector<ConvLayer> layers;
    std::map<std::string, int> layer_by_name;
    std::vector<Op> ops;
    std::vector<Tensor> tensors;
    int t_heat = -1, t_kpts_src = -1;
    int n_public = 0;                 // layers [0, n_public) are the reference's convs; the rest are internal
    int t_stem = -1, t_branch0 = -1;  // tensors whose dims decide whether the fused head applies
    int l_head0 = -1, l_head1 = -1;   // last_layer.0 / last_layer.3
    int head_direct_coff = 0, head_direct_c = 0, head_hp = 0, head_m2 = 0;
    int head_k = 0, head_ks1 = 2;     // stage-1 K of the fused head (direct + folded branch channels), its k-steps
    bool fused_enabled = true, use_fused = false;
    // exact-fp32 engine: the head in its restructured form (per-source 1x1 products at native resolution, one bilinear sum) on the
    // generic fp32 kernels -- the 784 -> 784 product at 270x480 (31 % of the reference's MACs) shrinks ninefold
    bool has_split = false;
    bool split_enabled = getenv("SNCAL_SPLIT_HEAD") ? atoi(getenv("SNCAL_SPLIT_HEAD")) != 0 : true, use_split = false;
    // wide 3x3 stride-1 convolutions (96 / 192 / 384 channels) on the two-team persistent kernel (conv_tt.hip), bf16 path
    bool use_conv_tt = getenv("SNCAL_CONV_TT") ? atoi(getenv("SNCAL_CONV_TT")) != 0 : true;
    bool use_s2p = getenv("SNCAL_S2P") ? atoi(getenv("SNCAL_S2P")) != 0 : false;     // fp16x3: 3x3 stride-2 convolutions on the pipelined persistent kernel (conv_s2p.hip): bit-identical, measured slower -> off
    struct TTPlanDev { sncal::TTItem* items = nullptr; uint32_t* first = nullptr; int n_wgs = 0; int lazy = 0; int cfg = 0; };
    std::map<int, TTPlanDev> tt_plans;    // work lists per launch (key: index of its first op), rebuilt when the layout changes
    int n_cus = 0;
    // C5: fp8 (OCP e4m3) arithmetic for the wide 3x3 stride-1 convolutions, everything else as the bf16 engine
    bool fp8 = false, fp8_calibrated = false, calibrating = false;
    bool x3_res_twin = true;     // bf16x3 engine: residuals of the two-team convolutions from the split twin (SNCAL_X3_RES_TWIN=0: from fp32)
    bool x3_producer_twins = !(getenv("SNCAL_X3_SPLIT_PASS") && atoi(getenv("SNCAL_X3_SPLIT_PASS")) != 0);   // bf16x3: twins from the producers' epilogues, not from split_f32_kernel
    bool x3_generic = false;     // bf16x3 engine: generic convolutions on the x3_t variants (packed weights [4 hi | 4 lo] bf16 per k-group)
    bool x3 = false;                          // SNCAL_BF16X3: the fp32 engine with split-bf16 arithmetic in the 3x3 stride-1 convolutions of stages 2-4
    unsigned fp8_stages = 0;                  // bit s: stage s selected (0 = all stages)
    std::vector<int> fp8_widths;              // selected channel widths (empty = all)
    unsigned* d_amax = nullptr;               // calibration: per-tensor max |x| (float bit patterns)
    std::vector<char> need_bf16;              // per tensor: some active consumer reads the bf16 tensor
    std::vector<int> producer;                // per tensor: active op that writes it
    bool fuse_bblock = getenv("SNCAL_FUSE_BBLOCK") ? atoi(getenv("SNCAL_FUSE_BBLOCK")) != 0 : true;   // 48-channel BasicBlocks as one kernel (bblock.hip), bf16 path
    // split engines, layer1 (bneckx3.hip): bit 0 = conv3 of a Bottleneck + conv1 of the next as one pass, bit 1 = block 0's downsample branch inside its conv3
    int fuse_bneck = getenv("SNCAL_FUSE_BNECK") ? atoi(getenv("SNCAL_FUSE_BNECK")) : 3;
    bool fuse_bbx3 = getenv("SNCAL_FUSE_BBX3") ? atoi(getenv("SNCAL_FUSE_BBX3")) != 0 : true;         // ... and in split arithmetic (bblockx3.hip), bf16x3 engine
    void *d_hw0 = nullptr, *d_hw1 = nullptr;
    void *d_hw0_32 = nullptr, *d_hw1_32 = nullptr;      // head32.hip packing (null when K1 is not a multiple of 16)
    void *d_hw0_32l = nullptr, *d_hw1_32l = nullptr;    // bf16x3 engine (headx3.hip): lo parts of the split weights; d_hw0_32 / d_hw1_32 then hold the hi parts
    int head_ks16 = 0;
    float *d_hb0 = nullptr, *d_hb1 = nullptr;
    int cur_group = GRP_ALL;
    bool finalized = false;
    bool equalize = true;        // fp16x3: rebalance block-internal channels by powers of two at finalize (equalize_blocks)
    bool equalize_done = false;  // ... already applied to the weights held now (re-armed by sncal_hrnet_set_conv)
    int equalized = 0;           // channels moved by the last equalize_blocks
    int subbatch = 64;
    const ConvVariant* variants = nullptr;
    int nvariants = 0;
    // cached per-(sb,H,W) layout
    int lay_sb = -1, lay_h = -1, lay_w = -1;
    size_t lay_bytes = 0;
    // profiling (sncal_hrnet_set_profiling): events recorded between launches + what each interval ran
    int profiling = 0;                    // 0 off, 1 every launch, 2 only the launches of `focus` (labels cached per op by a mode-1 run)
    std::string focus;
    int n_launch_groups = 0;
    std::vector<std::string> op_label;
    struct Interval { hipEvent_t e0, e1; std::string kernel; double flops, bytes; };
    std::vector<Interval> intervals;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::string last_kernel;
    double last_flops = 0, last_bytes = 0;
    // work tickets of the persistent kernels that deal their work dynamically (bneckx3.hip, bblockx3.hip): 64 zeroed words, re-armed by the
    // kernels themselves; launches of one network are ordered on its stream, so they share the words
    unsigned* d_tickets = nullptr;
    // range flag of the split-fp16 engine (x3.hpp x3_report): [0] wavefronts that split a value beyond +-65504, [1] workgroups of the
    // layout kernel that met a NaN / infinite input value.  Sticky until sncal_hrnet_range_status(clear = 1); allocated at finalize
    unsigned* d_range = nullptr;
    // test instrumentation (sncal_hrnet_plan_tap): copies of plan tensors taken while the executor passes an op
    struct Tap { int op, tensor; void* dst; };
    std::vector<Tap> taps;
};

namespace {

std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

struct Builder {
    sncal_hrnet& net;
    explicit Builder(sncal_hrnet& n) : net(n) {}

    int add_layer(const std::string& name, const std::string& bn, int cin, int cout, int k, int stride, bool bias) {
        ConvLayer L;
        L.name = name; L.bn = bn; L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.bias = bias;
        { const size_t q = name.find("stage"); if (q != std::string::npos && q + 5 < name.size()) L.stage = name[q + 5] - '0'; }
        net.layers.push_back(L);
        net.layer_by_name[name] = (int)net.layers.size() - 1;
        return (int)net.layers.size() - 1;
    }
    int new_tensor(int C, bool f32 = false) {
        Tensor t; t.C = C; t.f32 = f32;
        net.tensors.push_back(t);
        return (int)net.tensors.size() - 1;
    }
    int conv(const std::string& name, int in, bool relu, int res = -1, bool out_f32 = false) {
        auto it = net.layer_by_name.find(name);
        if (it == net.layer_by_name.end()) { set_error("internal: conv %s not enumerated", name.c_str()); return -1; }
        const ConvLayer& L = net.layers[it->second];
        Op op; op.type = OP_CONV; op.conv = it->second; op.in = in; op.res = res; op.relu = relu; op.out_f32 = out_f32;
        op.group = net.cur_group;
        const int cphys = out_f32 ? ((L.cout + 15) / 16) * 16 : L.cout;
        op.out = new_tensor(cphys, out_f32);
        net.ops.push_back(op);
        return op.out;
    }
    int upadd(int base, const std::vector<int>& srcs, bool relu, int C) {
        Op op; op.type = OP_UPADD; op.base = base; op.nsrc = (int)srcs.size(); op.relu = relu;
        for (size_t i = 0; i < srcs.size(); ++i) op.srcs[i] = srcs[i];
        op.group = net.cur_group;
        op.out = new_tensor(C);
        net.ops.push_back(op);
        return op.out;
    }
    void concat_part(int cat, int src, int coff, int dims_from, int dims_mul) {
        Op op; op.type = OP_UPADD; op.base = -1; op.nsrc = 1; op.srcs[0] = src; op.out = cat; op.out_coff = coff;
        op.dims_from = dims_from; op.dims_mul = dims_mul; op.group = net.cur_group;
        net.ops.push_back(op);
    }

    // ---- enumeration in the reference's registration order (hrnet.py:255-355) -------------------------
    void block_layers(const std::string& p, bool bottleneck, int inpl, int planes, bool ds) {
        if (!bottleneck) {
            add_layer(p + ".conv1", p + ".bn1", inpl, planes, 3, 1, false);
            add_layer(p + ".conv2", p + ".bn2", planes, planes, 3, 1, false);
            if (ds) add_layer(p + ".downsample.0", p + ".downsample.1", inpl, planes, 1, 1, false);
        } else {
            add_layer(p + ".conv1", p + ".bn1", inpl, planes, 1, 1, false);
            add_layer(p + ".conv2", p + ".bn2", planes, planes, 3, 1, false);
            add_layer(p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, false);
            if (ds) add_layer(p + ".downsample.0", p + ".downsample.1", inpl, planes * 4, 1, 1, false);
        }
    }

    void enumerate() {
        const sncal_hrnet_desc& d = net.desc;
        const std::string P = "model.";
        add_layer(P + "conv1", P + "bn1", 3, d.stem_width, 3, 2, false);
        add_layer(P + "conv2", P + "bn2", d.stem_width, d.stem_width, 3, 2, false);
        int inpl = 64;   // hard-coded in the reference (hrnet.py:273)
        for (int b = 0; b < d.stage1_blocks; ++b) {
            const bool ds = b == 0 && inpl != d.stage1_channels * 4;
            block_layers(fmt("%slayer1.%d", P.c_str(), b), true, inpl, d.stage1_channels, ds);
            inpl = d.stage1_channels * 4;
        }
        std::vector<int> pre{inpl};
        for (int si = 0; si < 3; ++si) {
            const int nb = d.num_branches[si];
            std::vector<int> cur(d.num_channels[si], d.num_channels[si] + nb);
            const std::string tn = fmt("%stransition%d", P.c_str(), si + 1);
            for (int i = 0; i < nb; ++i) {
                if (i < (int)pre.size()) {
                    if (cur[i] != pre[i])
                        add_layer(fmt("%s.%d.0", tn.c_str(), i), fmt("%s.%d.1", tn.c_str(), i), pre[i], cur[i], 3, 1, false);
                } else {
                    for (int j = 0; j < i + 1 - (int)pre.size(); ++j) {
                        const int cin = pre.back();
                        const int cout = (j == i - (int)pre.size()) ? cur[i] : cin;
                        add_layer(fmt("%s.%d.%d.0", tn.c_str(), i, j), fmt("%s.%d.%d.1", tn.c_str(), i, j), cin, cout, 3, 2, false);
                    }
                }
            }
            std::vector<int> inch = cur;
            for (int m = 0; m < d.num_modules[si]; ++m) {
                const std::string mn = fmt("%sstage%d.%d", P.c_str(), si + 2, m);
                for (int br = 0; br < nb; ++br)
                    for (int b = 0; b < d.num_blocks[si]; ++b) {
                        const int ch = d.num_channels[si][br];
                        block_layers(fmt("%s.branches.%d.%d", mn.c_str(), br, b), false, inch[br], ch, b == 0 && inch[br] != ch);
                        inch[br] = ch;
                    }
                for (int i = 0; i < nb; ++i)
                    for (int j = 0; j < nb; ++j) {
                        const std::string fn = fmt("%s.fuse_layers.%d.%d", mn.c_str(), i, j);
                        if (j > i) add_layer(fn + ".0", fn + ".1", inch[j], inch[i], 1, 1, false);
                        else if (j < i)
                            for (int k = 0; k < i - j; ++k) {
                                const int cout = (k == i - j - 1) ? inch[i] : inch[j];
                                add_layer(fmt("%s.%d.0", fn.c_str(), k), fmt("%s.%d.1", fn.c_str(), k), inch[j], cout, 3, 2, false);
                            }
                    }
            }
            pre = inch;
        }
        int last = 0;
        for (int c : pre) last += c;
        if (d.upscale > 1) last += d.stem_width;
        add_layer(P + "last_layer.0", P + "last_layer.1", last, last, 1, 1, true);
        add_layer(P + "last_layer.3", "", last, d.num_classes, 1, 1, true);
    }

    // ---- op graph (hrnet.py:437-511) -----------------------------------------------------------------
    int basic_block(const std::string& p, int x) {   // hrnet.py:42-58
        const int t = conv(p + ".conv1", x, true);
        int res = x;
        if (net.layer_by_name.count(p + ".downsample.0")) res = conv(p + ".downsample.0", x, false);
        return conv(p + ".conv2", t, true, res);
    }
    int bottleneck(const std::string& p, int x) {    // hrnet.py:79-99
        int t = conv(p + ".conv1", x, true);
        t = conv(p + ".conv2", t, true);
        int res = x;
        if (net.layer_by_name.count(p + ".downsample.0")) res = conv(p + ".downsample.0", x, false);
        return conv(p + ".conv3", t, true, res);
    }

    // The ops of one module's fuse section, re-ordered (the reference registers them output by output, hrnet.py:229-244): the stride-2
    // convolutions that START a fuse-down chain on the same input tensor become one launch group of consecutive ops, so that the executor
    // can run them as ONE launch that fetches the input once (conv.hpp conv_shared_s2_kernel).  A list scheduler over the section's own data
    // dependences: ops go out in the reference's order as they become ready; a group goes out as a whole, when its last member is ready
    // (members never depend on each other: a chain's first convolution reads a module input, and its accumulate operand comes from chains
    // of OTHER inputs).  Sums are accumulated in the reference's order: same bits.  SNCAL_SHARE_S2=0 keeps the reference's op order.
    void schedule_fuse_section(size_t begin) {
        static const bool off = getenv("SNCAL_SHARE_S2") && atoi(getenv("SNCAL_SHARE_S2")) == 0;
        const size_t n = net.ops.size() - begin;
        if (off || n < 3) return;
        std::vector<Op> sec(net.ops.begin() + begin, net.ops.end());
        std::map<int, int> producer;                         // tensor -> op of the section that writes it
        for (size_t i = 0; i < n; ++i) if (sec[i].out >= 0) producer[sec[i].out] = (int)i;
        auto reads = [&](const Op& o) {
            std::vector<int> r{o.in, o.res, o.base, o.dims_from};
            for (int k = 0; k < o.nsrc; ++k) r.push_back(o.srcs[k]);
            return r;
        };
        std::vector<int> grp(n, -1);                          // group key per op: existing launch groups keep theirs
        std::map<int, std::vector<int>> shared;               // input tensor -> chain-starting stride-2 convolutions
        for (size_t i = 0; i < n; ++i) {
            const Op& o = sec[i];
            if (o.launch_group >= 0) { grp[i] = o.launch_group; continue; }
            if (o.type == OP_CONV && net.layers[o.conv].stride == 2 && net.layers[o.conv].k == 3 && !producer.count(o.in)) shared[o.in].push_back((int)i);
        }
        for (auto& kv : shared) {
            if (kv.second.size() < 2) continue;
            for (size_t k = 0; k < kv.second.size(); k += 3) {      // launches take up to three members
                if (kv.second.size() - k < 2) break;
                const int gid = net.n_launch_groups++;
                for (size_t q = k; q < std::min(kv.second.size(), k + 3); ++q) { grp[kv.second[q]] = gid; sec[kv.second[q]].launch_group = gid; sec[kv.second[q]].shared_in = true; }
            }
        }
        std::vector<char> done(n, 0);
        auto ready = [&](size_t i) {
            for (int t : reads(sec[i])) { auto it = t >= 0 ? producer.find(t) : producer.end(); if (it != producer.end() && it->second != (int)i && !done[it->second]) return false; }
            return true;
        };
        std::vector<Op> order;
        while (order.size() < n) {
            bool progressed = false;
            for (size_t i = 0; i < n && !progressed; ++i) {
                if (done[i] || !ready(i)) continue;
                std::vector<size_t> members{i};
                if (grp[i] >= 0) {
                    members.clear();
                    bool all = true;
                    for (size_t q = 0; q < n; ++q) if (grp[q] == grp[i]) { members.push_back(q); all = all && !done[q] && ready(q); }
                    if (!all) continue;
                }
                for (size_t q : members) { order.push_back(sec[q]); done[q] = 1; }
                progressed = true;
            }
            if (!progressed) {                               // (cannot happen with HRNet's fuse layers; keep the reference's order rather than loop)
                for (size_t i = 0; i < n; ++i) { sec[i].launch_group = net.ops[begin + i].launch_group; sec[i].shared_in = false; }
                return;
            }
        }
        std::copy(order.begin(), order.end(), net.ops.begin() + begin);
    }

    bool build() {
        const sncal_hrnet_desc& d = net.desc;
        const std::string P = "model.";
        enumerate();
        const int t_in = new_tensor(net.ge);
        { Op op; op.type = OP_INPUT; op.out = t_in; net.ops.push_back(op); }
        const int t_stem = conv(P + "conv1", t_in, true);
        int x = conv(P + "conv2", t_stem, true);
        for (int b = 0; b < d.stage1_blocks; ++b) x = bottleneck(fmt("%slayer1.%d", P.c_str(), b), x);
        std::vector<int> ys{x};
        for (int si = 0; si < 3; ++si) {
            const int nb = d.num_branches[si];
            const std::string tn = fmt("%stransition%d", P.c_str(), si + 1);
            std::vector<int> xs;
            for (int i = 0; i < nb; ++i) {
                if (i < (int)ys.size()) {
                    if (net.layer_by_name.count(fmt("%s.%d.0", tn.c_str(), i))) xs.push_back(conv(fmt("%s.%d.0", tn.c_str(), i), ys[i], true));
                    else xs.push_back(ys[i]);
                } else {
                    int t = ys.back();
                    for (int j = 0; j < i + 1 - (int)ys.size(); ++j) t = conv(fmt("%s.%d.%d.0", tn.c_str(), i, j), t, true);
                    xs.push_back(t);
                }
            }
            for (int m = 0; m < d.num_modules[si]; ++m) {
                const std::string mn = fmt("%sstage%d.%d", P.c_str(), si + 2, m);
                // Branch 0 keeps block order (its conv pairs are pattern-matched into the fused BasicBlock kernel of the
                // bf16 path).  The other branches are emitted depth-major: the same-depth convs of branches 1..nb-1 are
                // independent and adjacent, so the executor can put them into ONE grouped launch (conv.hpp).
                const bool group_convs = !(getenv("SNCAL_GROUP_CONVS") && atoi(getenv("SNCAL_GROUP_CONVS")) == 0);     // read per net: tests toggle it
                bool plain = true;
                for (int br = 0; br < nb; ++br)
                    for (int b = 0; b < d.num_blocks[si]; ++b)
                        if (net.layer_by_name.count(fmt("%s.branches.%d.%d.downsample.0", mn.c_str(), br, b))) plain = false;
                if (group_convs && plain && nb > 2) {
                    for (int b = 0; b < d.num_blocks[si]; ++b) xs[0] = basic_block(fmt("%s.branches.0.%d", mn.c_str(), b), xs[0]);
                    for (int b = 0; b < d.num_blocks[si]; ++b) {
                        std::vector<int> t(nb);
                        const int g1 = net.n_launch_groups++;
                        for (int br = 1; br < nb; ++br) { t[br] = conv(fmt("%s.branches.%d.%d.conv1", mn.c_str(), br, b), xs[br], true); net.ops.back().launch_group = g1; }
                        const int g2 = net.n_launch_groups++;
                        for (int br = 1; br < nb; ++br) { xs[br] = conv(fmt("%s.branches.%d.%d.conv2", mn.c_str(), br, b), t[br], true, xs[br]); net.ops.back().launch_group = g2; }
                    }
                } else {
                    for (int br = 0; br < nb; ++br)
                        for (int b = 0; b < d.num_blocks[si]; ++b) xs[br] = basic_block(fmt("%s.branches.%d.%d", mn.c_str(), br, b), xs[br]);
                }
                std::vector<int> out(nb);
                const size_t fuse_begin = net.ops.size();
                for (int i = 0; i < nb; ++i) {                       // hrnet.py:229-244
                    int acc = xs[i];
                    const bool has_up = i < nb - 1;
                    for (int j = 0; j < i; ++j) {                    // fuse-down chains end with an accumulate
                        const std::string fn = fmt("%s.fuse_layers.%d.%d", mn.c_str(), i, j);
                        int t = xs[j];
                        for (int k = 0; k < i - j; ++k) {
                            const bool lastk = k == i - j - 1;
                            if (!lastk) t = conv(fmt("%s.%d.0", fn.c_str(), k), t, true);
                            else acc = conv(fmt("%s.%d.0", fn.c_str(), k), t, /*relu=*/!has_up && j == i - 1, acc);
                        }
                    }
                    if (has_up) {
                        std::vector<int> ups;
                        const bool grp = !(getenv("SNCAL_GROUP_CONVS") && atoi(getenv("SNCAL_GROUP_CONVS")) == 0) && nb - i - 1 >= 2;
                        const int gid = grp ? net.n_launch_groups++ : -1;      // the 1x1 convs of one fuse-up sum are independent
                        for (int j = i + 1; j < nb; ++j) {
                            ups.push_back(conv(fmt("%s.fuse_layers.%d.%d.0", mn.c_str(), i, j), xs[j], false));
                            net.ops.back().launch_group = gid;
                        }
                        acc = upadd(acc, ups, true, net.tensors[xs[i]].C);
                    }
                    out[i] = acc;
                }
                schedule_fuse_section(fuse_begin);
                xs = out;
            }
            ys = xs;
        }
        // head, reference formulation: upsample + concat + two 1x1 convs (hrnet.py:489-510; line/hrnet.py:236-248)
        net.n_public = (int)net.layers.size();
        net.t_stem = t_stem; net.t_branch0 = ys[0];
        net.l_head0 = net.layer_by_name[P + "last_layer.0"]; net.l_head1 = net.layer_by_name[P + "last_layer.3"];
        int catC = 0;
        for (int t : ys) catC += net.tensors[t].C;
        if (d.upscale > 1) catC += d.stem_width;
        net.cur_group = GRP_UNFUSED;
        const int cat = new_tensor(catC);
        int coff = 0;
        if (d.upscale > 1) { concat_part(cat, t_stem, coff, ys[0], d.upscale); coff += d.stem_width; }
        for (int t : ys) { concat_part(cat, t, coff, ys[0], d.upscale); coff += net.tensors[t].C; }
        const int hid = conv(P + "last_layer.0", cat, true);
        const int logits = conv(P + "last_layer.3", hid, false, -1, true);
        // head, fused formulation (head.hip): per-branch 1x1 products at native resolution + one fused kernel
        net.cur_group = GRP_FUSED;
        net.head_hp = ((catC + 31) / 32) * 32;
        net.head_m2 = (d.num_classes + 15) / 16;
        {
            Op hop; hop.type = OP_HEAD; hop.group = GRP_FUSED; hop.out = logits;
            int col = 0;
            std::vector<int> gathered;
            if (d.upscale > 1) { hop.head_direct = t_stem; net.head_direct_coff = 0; net.head_direct_c = d.stem_width; col = d.stem_width; gathered = ys; }
            else { hop.head_direct = ys[0]; net.head_direct_coff = 0; net.head_direct_c = net.tensors[ys[0]].C; col = net.head_direct_c; gathered.assign(ys.begin() + 1, ys.end()); }
            // narrow branches are upsampled inside the head kernel and appended to the stage-1 K dimension (their
            // columns of last_layer.0 follow the direct tensor's in concat order); the wide ones go through
            // t_i = W0_i . b_i at native resolution and are gathered
            net.head_k = net.head_direct_c;
            size_t first = 0;
            while (first < gathered.size() && hop.head_nfold < HEAD_MAX_FOLD && gathered.size() - first > 2 &&
                   net.head_k + net.tensors[gathered[first]].C <= 224 && net.tensors[gathered[first]].C % 8 == 0 && net.head_k % 8 == 0) {
                hop.head_fold[hop.head_nfold++] = gathered[first];
                net.head_k += net.tensors[gathered[first]].C;
                col += net.tensors[gathered[first]].C;
                ++first;
            }
            net.head_ks1 = net.head_k <= 64 ? 2 : net.head_k <= 160 ? 5 : 7;
            for (size_t gi = first; gi < gathered.size(); ++gi) {
                const int t = gathered[gi];
                const std::string nm = fmt("head.t%d", hop.head_nsrc);
                const int li = add_layer(nm, "", net.tensors[t].C, net.head_hp, 1, 1, false);
                net.layers[li].derived = true; net.layers[li].col_off = col;
                col += net.tensors[t].C;
                hop.head_src[hop.head_nsrc++] = conv(nm, t, false);
            }
            net.ops.push_back(hop);
        }
        // head, split formulation for the exact-fp32 engine: W0 . concat(up(b_i)) = sum_i up(W0_i . b_i) (a 1x1 convolution commutes
        // with bilinear interpolation): every source's 1x1 product at ITS OWN resolution (generic fp32 conv kernel; the direct
        // tensor's carries the folded-BN shift), one upsample_add with ReLU, then last_layer.3.  Same arithmetic type as the
        // reference formulation, different summation order (fp32 rounding level); 8.8 instead of 79.7 GMAC per frame at 960x540
        net.cur_group = GRP_SPLIT;
        {
            const int direct = d.upscale > 1 ? t_stem : ys[0];
            std::vector<int> rest;
            if (d.upscale > 1) rest = ys; else rest.assign(ys.begin() + 1, ys.end());
            if (rest.size() <= 4) {
                int col = 0;
                const ConvLayer& H0 = net.layers[net.l_head0];
                const int ld = add_layer("headx.d", "", net.tensors[direct].C, H0.cout, 1, 1, false);
                net.layers[ld].derived = true; net.layers[ld].col_off = col; net.layers[ld].derived_shift = true;
                col += net.tensors[direct].C;
                const int t_d = conv("headx.d", direct, false);
                std::vector<int> prods;
                for (size_t gi = 0; gi < rest.size(); ++gi) {
                    const std::string nm = fmt("headx.t%d", (int)gi);
                    const int li = add_layer(nm, "", net.tensors[rest[gi]].C, H0.cout, 1, 1, false);
                    net.layers[li].derived = true; net.layers[li].col_off = col;
                    col += net.tensors[rest[gi]].C;
                    prods.push_back(conv(nm, rest[gi], false));
                }
                const int hidden = upadd(t_d, prods, true, H0.cout);
                net.ops.back().group = GRP_SPLIT;
                Op op; op.type = OP_CONV; op.conv = net.l_head1; op.in = hidden; op.relu = false; op.out_f32 = true; op.group = GRP_SPLIT;
                op.out = logits;
                net.ops.push_back(op);
                net.has_split = true;
            }
        }
        net.cur_group = GRP_ALL;
        { Op op; op.type = OP_SOFTMAX; op.in = logits; op.out = new_tensor(d.num_classes, true);
          net.tensors[op.out].external_heat = true; net.t_heat = op.out; net.ops.push_back(op); }
        { Op op; op.type = OP_DECODE; op.in = net.t_heat; net.ops.push_back(op); }
        // C5: every wide 3x3 stride-1 convolution may run in fp8 -> its input tensor gets an e4m3 twin (allocated only while
        // the layer is selected, see layout())
        if (net.x3)
            for (const Op& op : net.ops) {
                if (op.type != OP_CONV || op.group != GRP_ALL) continue;
                const ConvLayer& L = net.layers[op.conv];
                if (L.k == 3 && L.stride == 1 && L.stage >= 2 && L.cin % 16 == 0 && L.cout % 16 == 0 && net.tensors[op.in].C == L.cin && net.tensors[op.in].twin < 0) {
                    const int tw = new_tensor(L.cin);
                    net.tensors[tw].split = true;
                    net.tensors[op.in].twin = tw;
                }
            }
        if (net.fp8)
            for (const Op& op : net.ops) {
                if (op.type != OP_CONV) continue;
                const ConvLayer& L = net.layers[op.conv];
                if (L.k == 3 && L.stride == 1 && L.cin % 32 == 0 && L.cout % TT_COUT == 0 && net.tensors[op.in].C == L.cin && net.tensors[op.in].twin < 0) {
                    const int tw = new_tensor(L.cin);
                    net.tensors[tw].fp8 = true;
                    net.tensors[op.in].twin = tw;
                }
            }
        // lifetimes
        for (size_t i = 0; i < net.ops.size(); ++i) {
            const Op& op = net.ops[i];
            auto use = [&](int t) { if (t >= 0) net.tensors[t].last = (int)i; };
            use(op.in); use(op.res); use(op.base);
            for (int s = 0; s < op.nsrc; ++s) use(op.srcs[s]);
            if (op.out >= 0) { Tensor& t = net.tensors[op.out]; if (t.first < 0) t.first = (int)i; t.last = std::max(t.last, (int)i); }
            if (op.dims_from >= 0) use(op.dims_from);
        }
        return true;
    }
};

// choose (MI, G) for a layer: maximise useful/padded work x operand reuse among the instantiated variants
void choose_packing(sncal_hrnet& net, ConvLayer& L) {
    const int ge = net.ge;
    const int cout_frags = (L.cout + 15) / 16;
    double best = -1;
    static const int force_mi = getenv("SNCAL_FORCE_MI") ? atoi(getenv("SNCAL_FORCE_MI")) : 0;   // tuning aid
    for (int v = 0; v < net.nvariants; ++v) {
        const ConvVariant& V = net.variants[v];
        if (V.ks != L.k || V.stride != L.stride) continue;
        if (force_mi && L.k == 3 && L.stride == 1 && V.mi != force_mi && cout_frags % force_mi == 0) continue;
        { static const int force_mi_s2 = getenv("SNCAL_FORCE_MI_S2") ? atoi(getenv("SNCAL_FORCE_MI_S2")) : 0;
          if (force_mi_s2 && L.k == 3 && L.stride == 2 && L.cin_phys >= 48 && V.mi != force_mi_s2 && cout_frags % force_mi_s2 == 0) continue; }
        { static const int force_g_s2 = getenv("SNCAL_FORCE_G_S2") ? atoi(getenv("SNCAL_FORCE_G_S2")) : 0;
          if (force_g_s2 && L.k == 3 && L.stride == 2 && L.cin_phys >= 48 && V.g != force_g_s2) continue; }
        { static const int force_g = getenv("SNCAL_FORCE_G") ? atoi(getenv("SNCAL_FORCE_G")) : 0;
          if (force_g && L.k == 3 && L.stride == 1 && L.cin_phys >= 96 && V.g != force_g) continue;
          static const int force_g48 = getenv("SNCAL_FORCE_G48") ? atoi(getenv("SNCAL_FORCE_G48")) : 0;
          if (force_g48 && L.k == 3 && L.stride == 1 && L.cin_phys == 48 && L.cout == 48 && V.g != force_g48) continue; }
        {   // tuning aid: SNCAL_FORCE_PACK="k,stride,cin,cout,mi,g" pins the packing of the layers of that shape (cout 0 = any)
            static const char* fp = getenv("SNCAL_FORCE_PACK");
            int fk, fs, fci, fco, fmi, fg;
            if (fp && sscanf(fp, "%d,%d,%d,%d,%d,%d", &fk, &fs, &fci, &fco, &fmi, &fg) == 6 && L.k == fk && L.stride == fs && L.cin_phys == fci &&
                (fco == 0 || L.cout == fco) && (V.mi != fmi || V.g != fg)) continue;
        }
        const int chunks = (L.cin_phys + V.g * ge - 1) / (V.g * ge);
        const int nks = conv_nks(V.ks, V.g);
        const double k_eff = (double)(L.k * L.k * L.cin_phys / ge) / (double)(chunks * nks * 4);
        const int nblk = (cout_frags + V.mi - 1) / V.mi;
        const double m_eff = (double)cout_frags / (nblk * V.mi);
        const double reuse = (double)(V.mi * 4) / (V.mi + 4);           // MFMAs per LDS fragment read (NI=4 nominal)
        const double per_chunk = (double)nks / (nks + 1.0);              // amortisation of the per-chunk sync/load
        // two workgroups per CU (<= 80 KB of LDS each) hide the staging rounds; judged on a nominal 2-wide tile
        const size_t stage2 = conv_stage_bytes(V.ks, V.stride, V.ni, V.mi, V.g, 2);
        // three resident workgroups (<= 53 KB, <= 168 VGPRs) measured 4 % faster on the latency-bound 48-channel class
        const double occ = stage2 > 80 * 1024 ? 0.55 : (stage2 <= 53 * 1024 && conv_wgs_per_cu(V.ks, V.ni, V.mi, V.g) == 3) ? 1.1 : 1.0;
        const double score = k_eff * m_eff * (0.55 + 0.45 * reuse / 2.4) * per_chunk * occ;
        if (score > best + 1e-9) { best = score; L.mi = V.mi; L.g = V.g; }
    }
    L.cout_frags = cout_frags;
    L.nblk = (cout_frags + L.mi - 1) / L.mi;
    L.chunks = (L.cin_phys + L.g * ge - 1) / (L.g * ge);
}

inline uint16_t f2bf(float f) {   // round-to-nearest-even
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int pack_layer(sncal_hrnet& net, ConvLayer& L) {
    const int ge = net.ge, KS = L.k, G = L.g, MI = L.mi;
    const int nks = conv_nks(KS, G);
    const size_t n16 = (size_t)L.nblk * L.chunks * nks * MI * 64;       // 16-byte vectors
    std::vector<uint8_t> host(n16 * 16, 0);
    for (int nb = 0; nb < L.nblk; ++nb)
        for (int c = 0; c < L.chunks; ++c)
            for (int s = 0; s < nks; ++s)
                for (int mi = 0; mi < MI; ++mi)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, g = lane >> 4;
                        const int kg = 4 * s + g;
                        const int tap = kg / G, cgi = kg % G;
                        const int co = (nb * MI + mi) * 16 + m;
                        uint8_t* dst = host.data() + ((((size_t)(nb * L.chunks + c) * nks + s) * MI + mi) * 64 + lane) * 16;
                        if (tap >= KS * KS || co >= L.cout) continue;
                        for (int e = 0; e < ge; ++e) {
                            const int ci = (c * G + cgi) * ge + e;
                            if (ci >= L.cin) continue;
                            const float v = L.w[(((size_t)co * L.cin + ci) * KS + tap / KS) * KS + tap % KS] * L.scale[co];
                            if (net.dtype == SNCAL_BF16) { const uint16_t b = f2bf(v); memcpy(dst + e * 2, &b, 2); }
                            else if (net.x3_generic) {          // [4 hi | 4 lo]: hi = rne16(w), lo = rne16(w - hi) (x3.hpp)
                                uint16_t h, l;
                                x3_split_host(v, &h, &l);
                                memcpy(dst + e * 2, &h, 2); memcpy(dst + 8 + e * 2, &l, 2);
                            }
                            else memcpy(dst + e * 4, &v, 4);
                        }
                    }
    std::vector<float> bias((size_t)L.nblk * MI * 16, 0.f);
    for (int co = 0; co < L.cout; ++co) bias[co] = L.shift[co];
    if (L.d_w) { (void)hipFree(L.d_w); L.d_w = nullptr; }
    if (L.d_bias) { (void)hipFree(L.d_bias); L.d_bias = nullptr; }
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w, host.size()));
    SNCAL_CHECK_HIP(hipMalloc((void**)&L.d_bias, bias.size() * sizeof(float)));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w, host.data(), host.size(), hipMemcpyHostToDevice));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    return SNCAL_OK;
}

// Packing of a wide 3x3 stride-1 layer for the two-team kernel (conv_tt.hip): per (96-channel block nb, 32-channel chunk c)
// one 54 KB stage [tap 9][channel half 2][32-row block 3][lane 64] x 8 bf16, the A fragments of v_mfma_f32_32x32x16_bf16:
// lane l holds output channel nb * 96 + mb * 32 + (l & 31), input channels c * 32 + h * 16 + (l >> 5) * 8 + 0..7 of the tap.
bool tt_shape_ok(const sncal_hrnet& net, const ConvLayer& L) {
    return net.dtype == SNCAL_BF16 && L.k == 3 && L.stride == 1 && L.cin == L.cin_phys && L.cin % TT_CIN == 0 &&
           L.cout % TT_COUT == 0 && L.cout <= 480;
}

// bf16x3 engine: [nb][16-channel chunk c][tap 9][part: hi, lo][32-row block 3][lane 64] x 8 bf16 -- the two-team kernel's stage layout
// with the stage's two K = 16 steps holding the hi and the lo parts of the SAME 16 input channels: lane l holds output channel
// nb * 96 + mb * 32 + (l & 31) (zero rows above the layer's width: a 48-channel layer runs as one padded 96-channel block), input
// channels c * 16 + (l >> 5) * 8 + 0..7 of the tap; hi = bf16(w), lo = bf16(w - hi), w = folded weight in fp32.
bool x3_shape_ok(const sncal_hrnet& net, const ConvLayer& L) {
    return net.x3 && net.dtype == SNCAL_F32 && L.k == 3 && L.stride == 1 && L.stage >= 2 && L.cin == L.cin_phys && L.cin % 16 == 0 &&
           L.cout % 16 == 0 && L.cout <= 480;
}

int pack_layer_x3(sncal_hrnet& net, ConvLayer& L) {
    if (L.d_w_x3) { (void)hipFree(L.d_w_x3); L.d_w_x3 = nullptr; }
    if (!x3_shape_ok(net, L)) return SNCAL_OK;
    static const bool blk64 = !(getenv("SNCAL_X3_BLK64") && atoi(getenv("SNCAL_X3_BLK64")) == 0);
    L.x3_blk = (L.cout % TT_COUT == 0 || !blk64) ? TT_COUT : 64;       // 48 channels: one padded 64-channel block (25 % zero rows) instead of 96 (50 %)
    const int MBk = L.x3_blk / 32;
    const int chunks = L.cin / 16, nblk = (L.cout + L.x3_blk - 1) / L.x3_blk;
    std::vector<uint16_t> host((size_t)nblk * chunks * 9 * 2 * MBk * 64 * 8, 0);
    for (int nb = 0; nb < nblk; ++nb)
        for (int c = 0; c < chunks; ++c)
            for (int s = 0; s < 9; ++s)
                for (int mb = 0; mb < MBk; ++mb)
                    for (int lane = 0; lane < 64; ++lane) {
                        // MFMA row r = lane & 31 computes channel x3_row_channel(r) of its block: quads 2 p, 2 p + 1 of a lane's accumulator
                        // registers are then 8 consecutive channels (conv_tt_body.inc x3_quad_channel, the twin-only epilogue)
                        const int r = lane & 31, rq = r >> 3, rh = (r >> 2) & 1, ri = r & 3;
                        const int co = nb * L.x3_blk + mb * 32 + 16 * (rq >> 1) + 8 * rh + 4 * (rq & 1) + ri;
                        if (co >= L.cout) continue;
                        uint16_t* hi = host.data() + ((((((size_t)nb * chunks + c) * 9 + s) * 2 + 0) * MBk + mb) * 64 + lane) * 8;
                        uint16_t* lo = host.data() + ((((((size_t)nb * chunks + c) * 9 + s) * 2 + 1) * MBk + mb) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) {
                            const int ci = c * 16 + (lane >> 5) * 8 + e;
                            const float w = L.w[(((size_t)co * L.cin + ci) * 3 + s / 3) * 3 + s % 3] * L.scale[co];
                            x3_split_host(w, &hi[e], &lo[e]);
                        }
                    }
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w_x3, host.size() * 2));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w_x3, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    return SNCAL_OK;
}

// bf16x3 engine: the fused 48-channel BasicBlock's own packing of a 48 -> 48 layer (bblockx3.hpp: 14 pair-steps of 6 KB)
int pack_layer_bbx3(sncal_hrnet& net, ConvLayer& L) {
    if (L.d_w_bbx) { (void)hipFree(L.d_w_bbx); L.d_w_bbx = nullptr; }
    if (!x3_shape_ok(net, L) || L.cin != 48 || L.cout != 48) return SNCAL_OK;
    std::vector<uint16_t> host;
    bbx3_pack_weights(L.w.data(), L.scale.data(), [](float v, uint16_t* hi, uint16_t* lo) { x3_split_host(v, hi, lo); }, host);
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w_bbx, host.size() * 2));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w_bbx, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    return SNCAL_OK;
}

// split engines: a 1x1 layer of layer1 in the fused Bottleneck seam's fragment order (bneckx3.hpp)
int pack_layer_bnp(sncal_hrnet& net, ConvLayer& L) {
    if (L.d_w_bnp) { (void)hipFree(L.d_w_bnp); L.d_w_bnp = nullptr; }
    const bool shape = L.k == 1 && L.stride == 1 && ((L.cin == BNP_MID && L.cout == BNP_WIDE) || (L.cin == BNP_WIDE && L.cout == BNP_MID));
    if (!net.x3 || net.dtype != SNCAL_F32 || !shape || L.derived) return SNCAL_OK;
    std::vector<uint16_t> host;
    bnp_pack_weights(L.w.data(), L.scale.data(), L.cout, L.cin, [](float v, uint16_t* hi, uint16_t* lo) { x3_split_host(v, hi, lo); }, host);
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w_bnp, host.size() * 2));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w_bnp, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    return SNCAL_OK;
}

int pack_layer_tt(sncal_hrnet& net, ConvLayer& L) {
    if (L.d_w_tt) { (void)hipFree(L.d_w_tt); L.d_w_tt = nullptr; }
    if (!tt_shape_ok(net, L)) return SNCAL_OK;
    const int chunks = L.cin / TT_CIN, nblk = L.cout / TT_COUT;
    std::vector<uint16_t> host((size_t)nblk * chunks * 9 * 2 * 3 * 64 * 8, 0);
    for (int nb = 0; nb < nblk; ++nb)
        for (int c = 0; c < chunks; ++c)
            for (int s = 0; s < 9; ++s)
                for (int h = 0; h < 2; ++h)
                    for (int mb = 0; mb < 3; ++mb)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = nb * TT_COUT + mb * 32 + (lane & 31);
                            uint16_t* dst = host.data() + ((((((size_t)nb * chunks + c) * 9 + s) * 2 + h) * 3 + mb) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const int ci = c * TT_CIN + h * 16 + (lane >> 5) * 8 + e;
                                dst[e] = f2bf(L.w[(((size_t)co * L.cin + ci) * 3 + s / 3) * 3 + s % 3] * L.scale[co]);
                            }
                        }
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w_tt, host.size() * 2));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w_tt, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    return SNCAL_OK;
}

// float -> OCP e4m3fn (1-4-3, bias 7, max 448, no infinities), round to nearest even, saturating
inline uint8_t f2fp8(float f) {
    if (!(f == f)) return 0x7f;
    const uint8_t sign = f < 0 ? 0x80 : 0;
    float a = std::fabs(f);
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0.0009765625f) return sign;                    // below half the smallest subnormal (2^-9 / 2): zero
    int e;
    float m = std::frexp(a, &e);                            // a = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                                      // biased exponent of 1.xxx * 2^(e-1)
    int q;
    if (E >= 1) {                                           // normal: 3 mantissa bits
        const float x = (m * 2.f - 1.f) * 8.f;
        q = (int)std::nearbyint(x);
        if (q == 8) { q = 0; ++E; }
        if (E > 15 || (E == 15 && q > 6)) return sign | 0x7e;
        return sign | (uint8_t)(E << 3) | (uint8_t)q;
    }
    q = (int)std::nearbyint(a * 512.f);                     // subnormal: multiples of 2^-9
    if (q >= 8) return sign | 0x08;
    return sign | (uint8_t)q;
}

// Packing of a wide 3x3 stride-1 layer for the fp8 variant of the two-team kernel: per (96-channel block nb, 64-channel chunk c)
// one 54 KB stage [tap 9][32-row block 3][half 2][lane 64] x 16 e4m3, the A operand of v_mfma_scale_f32_32x32x64_f8f6f4: lane l
// holds output channel nb * 96 + mb * 32 + (l & 31), input channels c * 64 + 32 (l >> 5) + 16 half + 0..15 of the tap (zeros
// beyond Cin).  One scale per output channel: wscale = max |w| / 448 over the folded weights of the channel.
int pack_layer_fp8(sncal_hrnet& net, ConvLayer& L) {
    if (L.d_w8) { (void)hipFree(L.d_w8); L.d_w8 = nullptr; }
    if (!net.fp8 || !tt_shape_ok(net, L)) return SNCAL_OK;
    const int chunks = (L.cin + 63) / 64, nblk = L.cout / TT_COUT;
    L.wscale.assign(L.cout, 1.f);
    for (int co = 0; co < L.cout; ++co) {
        float mx = 0.f;
        for (size_t i = 0; i < (size_t)L.cin * 9; ++i) mx = std::max(mx, std::fabs(L.w[(size_t)co * L.cin * 9 + i] * L.scale[co]));
        L.wscale[co] = mx > 0.f ? mx / 448.f : 1.f;
    }
    std::vector<uint8_t> host((size_t)nblk * chunks * 9 * 3 * 2 * 64 * 16, 0);
    for (int nb = 0; nb < nblk; ++nb)
        for (int c = 0; c < chunks; ++c)
            for (int s = 0; s < 9; ++s)
                for (int mb = 0; mb < 3; ++mb)
                    for (int half = 0; half < 2; ++half)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = nb * TT_COUT + mb * 32 + (lane & 31);
                            uint8_t* dst = host.data() + (((((((size_t)nb * chunks + c) * 9 + s) * 3 + mb) * 2 + half) * 64) + lane) * 16;
                            for (int e = 0; e < 16; ++e) {
                                const int ci = c * 64 + 32 * (lane >> 5) + 16 * half + e;
                                if (ci < L.cin) dst[e] = f2fp8(L.w[(((size_t)co * L.cin + ci) * 3 + s / 3) * 3 + s % 3] * L.scale[co] / L.wscale[co]);
                            }
                        }
    SNCAL_CHECK_HIP(hipMalloc(&L.d_w8, host.size()));
    SNCAL_CHECK_HIP(hipMemcpy(L.d_w8, host.data(), host.size(), hipMemcpyHostToDevice));
    if (!L.d_oscale) SNCAL_CHECK_HIP(hipMalloc((void**)&L.d_oscale, (size_t)L.cout * 4));
    return SNCAL_OK;
}

// stage-1 / stage-2 A fragments + biases of the fused head (head.hip), bf16 only
int pack_head(sncal_hrnet& net) {
    if (net.dtype != SNCAL_BF16 && !net.x3) return SNCAL_OK;
    const ConvLayer& H0 = net.layers[net.l_head0];
    const ConvLayer& H1 = net.layers[net.l_head1];
    if (!H1.is_set) { set_error("conv %s has no weights", H1.name.c_str()); return SNCAL_ERR_STATE; }
    const int HP = net.head_hp, NQ = HP / 32, M2 = net.head_m2, Cd = net.head_direct_c, coff = net.head_direct_coff;
    const int K1 = net.head_k, KS1 = net.head_ks1;      // stage-1 K: the first K1 concat columns (direct + folded branches)
    if (Cd > 64 || M2 > 4 || K1 > KS1 * 32) return SNCAL_OK;      // fused kernel does not apply; the reference formulation is used
    std::vector<uint16_t> w0((size_t)NQ * 2 * KS1 * 64 * 8, 0), w1((size_t)NQ * M2 * 64 * 8, 0);
    std::vector<float> b0(HP, 0.f), b1((size_t)std::max(M2 * 16, 64), 0.f);      // head32's decode epilogue reads 64 bias slots whatever C
    for (int q = 0; q < NQ; ++q)
        for (int f = 0; f < 2; ++f)
            for (int ks = 0; ks < KS1; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 15, gk = lane >> 4;
                    const int ch = q * 32 + (m >> 2) * 8 + f * 4 + (m & 3);      // row permutation, see head.hip
                    if (ch >= H0.cout) continue;
                    uint16_t* dst = w0.data() + ((((size_t)(q * 2 + f) * KS1 + ks) * 64) + lane) * 8;
                    for (int e = 0; e < 8; ++e) {
                        const int k = ks * 32 + gk * 8 + e;
                        if (k < K1) dst[e] = f2bf(H0.w[(size_t)ch * H0.cin + coff + k] * H0.scale[ch]);
                    }
                }
    for (int co = 0; co < H0.cout; ++co) b0[co] = H0.shift[co];
    for (int q = 0; q < NQ; ++q)
        for (int mi = 0; mi < M2; ++mi)
            for (int lane = 0; lane < 64; ++lane) {
                const int cls = mi * 16 + (lane & 15), gk = lane >> 4;
                if (cls >= H1.cout) continue;
                uint16_t* dst = w1.data() + (((size_t)(q * M2 + mi) * 64) + lane) * 8;
                for (int e = 0; e < 8; ++e) {
                    const int k = q * 32 + gk * 8 + e;
                    if (k < H1.cin) dst[e] = f2bf(H1.w[(size_t)cls * H1.cin + k] * H1.scale[cls]);
                }
            }
    for (int c = 0; c < H1.cout; ++c) b1[c] = H1.shift[c];
    for (void** q : {&net.d_hw0, &net.d_hw1, &net.d_hw0_32, &net.d_hw1_32, &net.d_hw0_32l, &net.d_hw1_32l}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    net.head_ks16 = 0;
    if (net.x3) {             // bf16x3 engine: the 32 x 32 x 16 layouts with every weight split into bf16 hi + bf16 lo (headx3.hip)
        if (K1 % 16 == 0) {
            const int KS16 = K1 / 16, RB = (M2 * 16 + 31) / 32;
            std::vector<uint16_t> v0h((size_t)NQ * KS16 * 64 * 8, 0), v0l(v0h.size(), 0), v1h((size_t)NQ * RB * 2 * 64 * 8, 0), v1l(v1h.size(), 0);
            auto put = [](float w, uint16_t& h, uint16_t& l) { x3_split_host(w, &h, &l); };
            for (int q = 0; q < NQ; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = h32_row_channel(lane & 31), kb = (lane >> 5) * 8;
                    const int ch = q * 32 + row;
                    for (int ks = 0; ks < KS16 && ch < H0.cout; ++ks) {
                        const size_t o = (((size_t)q * KS16 + ks) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) put(H0.w[(size_t)ch * H0.cin + coff + ks * 16 + kb + e] * H0.scale[ch], v0h[o + e], v0l[o + e]);
                    }
                    for (int rb = 0; rb < RB; ++rb)
                        for (int h = 0; h < 2; ++h) {
                            const int cls = rb * 32 + row;
                            if (cls >= H1.cout) continue;
                            const size_t o = ((((size_t)q * RB + rb) * 2 + h) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const int k = q * 32 + h * 16 + kb + e;
                                if (k < H1.cin) put(H1.w[(size_t)cls * H1.cin + k] * H1.scale[cls], v1h[o + e], v1l[o + e]);
                            }
                        }
                }
            for (auto pr : {std::make_pair(&net.d_hw0_32, &v0h), std::make_pair(&net.d_hw0_32l, &v0l), std::make_pair(&net.d_hw1_32, &v1h), std::make_pair(&net.d_hw1_32l, &v1l)}) {
                SNCAL_CHECK_HIP(hipMalloc(pr.first, pr.second->size() * 2));
                SNCAL_CHECK_HIP(hipMemcpy(*pr.first, pr.second->data(), pr.second->size() * 2, hipMemcpyHostToDevice));
            }
            net.head_ks16 = KS16;
        }
        for (int co = 0; co < H0.cout; ++co) b0[co] = H0.shift[co];
        for (int c = 0; c < H1.cout; ++c) b1[c] = H1.shift[c];
        if (net.d_hb0) { (void)hipFree(net.d_hb0); net.d_hb0 = nullptr; }
        if (net.d_hb1) { (void)hipFree(net.d_hb1); net.d_hb1 = nullptr; }
        SNCAL_CHECK_HIP(hipMalloc((void**)&net.d_hb0, b0.size() * 4));
        SNCAL_CHECK_HIP(hipMalloc((void**)&net.d_hb1, b1.size() * 4));
        SNCAL_CHECK_HIP(hipMemcpy(net.d_hb0, b0.data(), b0.size() * 4, hipMemcpyHostToDevice));
        SNCAL_CHECK_HIP(hipMemcpy(net.d_hb1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
        return SNCAL_OK;
    }
    if (K1 % 16 == 0) {       // head32.hip: A fragments of v_mfma_f32_32x32x16_bf16 -- lane l holds row h32_row_channel(l & 31) of the 32-row
        const int KS16 = K1 / 16, RB = (M2 * 16 + 31) / 32;                // block, k = 16 ks + 8 (l >> 5) + 0..7
        std::vector<uint16_t> v0((size_t)NQ * KS16 * 64 * 8, 0), v1((size_t)NQ * RB * 2 * 64 * 8, 0);
        for (int q = 0; q < NQ; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = h32_row_channel(lane & 31), kb = (lane >> 5) * 8;
                const int ch = q * 32 + row;
                for (int ks = 0; ks < KS16 && ch < H0.cout; ++ks) {
                    uint16_t* dst = v0.data() + (((size_t)q * KS16 + ks) * 64 + lane) * 8;
                    for (int e = 0; e < 8; ++e) dst[e] = f2bf(H0.w[(size_t)ch * H0.cin + coff + ks * 16 + kb + e] * H0.scale[ch]);
                }
                for (int rb = 0; rb < RB; ++rb)
                    for (int h = 0; h < 2; ++h) {
                        const int cls = rb * 32 + row;
                        if (cls >= H1.cout) continue;
                        uint16_t* dst = v1.data() + ((((size_t)q * RB + rb) * 2 + h) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) {
                            const int k = q * 32 + h * 16 + kb + e;
                            if (k < H1.cin) dst[e] = f2bf(H1.w[(size_t)cls * H1.cin + k] * H1.scale[cls]);
                        }
                    }
            }
        SNCAL_CHECK_HIP(hipMalloc(&net.d_hw0_32, v0.size() * 2));
        SNCAL_CHECK_HIP(hipMalloc(&net.d_hw1_32, v1.size() * 2));
        SNCAL_CHECK_HIP(hipMemcpy(net.d_hw0_32, v0.data(), v0.size() * 2, hipMemcpyHostToDevice));
        SNCAL_CHECK_HIP(hipMemcpy(net.d_hw1_32, v1.data(), v1.size() * 2, hipMemcpyHostToDevice));
        net.head_ks16 = KS16;
    }
    if (net.d_hb0) { (void)hipFree(net.d_hb0); net.d_hb0 = nullptr; }
    if (net.d_hb1) { (void)hipFree(net.d_hb1); net.d_hb1 = nullptr; }
    SNCAL_CHECK_HIP(hipMalloc(&net.d_hw0, w0.size() * 2));
    SNCAL_CHECK_HIP(hipMalloc(&net.d_hw1, w1.size() * 2));
    SNCAL_CHECK_HIP(hipMalloc((void**)&net.d_hb0, b0.size() * 4));
    SNCAL_CHECK_HIP(hipMalloc((void**)&net.d_hb1, b1.size() * 4));
    SNCAL_CHECK_HIP(hipMemcpy(net.d_hw0, w0.data(), w0.size() * 2, hipMemcpyHostToDevice));
    SNCAL_CHECK_HIP(hipMemcpy(net.d_hw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    SNCAL_CHECK_HIP(hipMemcpy(net.d_hb0, b0.data(), b0.size() * 4, hipMemcpyHostToDevice));
    SNCAL_CHECK_HIP(hipMemcpy(net.d_hb1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
    return SNCAL_OK;
}

// shape inference + workspace layout for a sub-batch of `sb` frames of HxW
inline bool op_active(const sncal_hrnet& net, const Op& op) {
    const int head = net.use_fused ? GRP_FUSED : net.use_split ? GRP_SPLIT : GRP_UNFUSED;
    return op.group == GRP_ALL || op.group == head;
}

bool tt_eligible(const sncal_hrnet& net, const Op& op, int sb);

int layout(sncal_hrnet& net, int sb, int H, int W) {
    if (net.lay_sb == sb && net.lay_h == H && net.lay_w == W) return SNCAL_OK;
    for (auto& kv : net.tt_plans) { (void)hipFree(kv.second.items); (void)hipFree(kv.second.first); }
    net.tt_plans.clear();
    std::vector<Tensor>& T = net.tensors;
    {   // does the fused head apply?  (bf16 path; the direct tensor must already sit at head resolution)
        auto half = [](int v) { return (v + 2 - 3) / 2 + 1; };
        const int sh = half(H), sw = half(W), bh = half(sh), bw = half(sw);
        const bool dims_ok = net.desc.upscale == 1 || (sh == bh * net.desc.upscale && sw == bw * net.desc.upscale);
        net.use_fused = net.fused_enabled && net.dtype == SNCAL_BF16 && dims_ok && net.d_hw0 != nullptr;
        if (net.x3 && net.fused_enabled && dims_ok && net.desc.upscale == 2 && net.d_hw0_32l && net.head_ks16 == 13 && net.head_m2 == 4 &&
            !(getenv("SNCAL_HEADX3") && atoi(getenv("SNCAL_HEADX3")) == 0)) {
            // bf16x3: the fused split-arithmetic head (headx3.hip) when its gather boxes fit: branches 2 and 3 against the head's width
            const int w2 = half(half(bw)), w3 = half(w2);
            const float sx2 = sw > 1 ? (float)(w2 - 1) / (float)(sw - 1) : 0.f, sx3 = sw > 1 ? (float)(w3 - 1) / (float)(sw - 1) : 0.f;
            net.use_fused = 2 * ((int)(sx2 * 31) + 3) <= 16 && 2 * ((int)(sx3 * 31) + 3) <= 16;
        }
        net.use_split = !net.use_fused && net.split_enabled && net.has_split && net.dtype == SNCAL_F32 && dims_ok;
    }
    for (const Op& op : net.ops) {
        if (!op_active(net, op)) continue;
        switch (op.type) {
            case OP_INPUT: T[op.out].H = H; T[op.out].W = W; break;
            case OP_CONV: {
                const ConvLayer& L = net.layers[op.conv];
                const int pad = L.k / 2;
                T[op.out].H = (T[op.in].H + 2 * pad - L.k) / L.stride + 1;
                T[op.out].W = (T[op.in].W + 2 * pad - L.k) / L.stride + 1;
                break;
            }
            case OP_UPADD:
                if (op.base >= 0) { T[op.out].H = T[op.base].H; T[op.out].W = T[op.base].W; }
                else { T[op.out].H = T[op.dims_from].H * op.dims_mul; T[op.out].W = T[op.dims_from].W * op.dims_mul; }
                break;
            case OP_SOFTMAX: T[op.out].H = T[op.in].H; T[op.out].W = T[op.in].W; break;
            case OP_HEAD: T[op.out].H = T[op.head_direct].H; T[op.out].W = T[op.head_direct].W; break;
            case OP_DECODE: break;
        }
    }
    // C5: which layers run in fp8 = selected by sncal_hrnet_set_fp8_layers AND served by the two-team kernel at this size (the only
    // kernel that reads an e4m3 twin): twins, producers' outputs and the dispatch below all key on this ONE predicate, so a selected
    // layer that falls back to the generic kernel (SNCAL_CONV_TT=0, an odd channel offset, a size limit) simply stays bf16
    for (ConvLayer& L : net.layers) { L.fp8_on = false; L.x3_on = false; }
    for (const Op& op : net.ops) {
        if (op.type != OP_CONV || !op_active(net, op)) continue;
        ConvLayer& L = net.layers[op.conv];
        L.x3_on = net.x3 && L.d_w_x3 != nullptr && T[op.in].twin >= 0 && tt_eligible(net, op, sb);
        bool w_ok = net.fp8_widths.empty();
        for (int w : net.fp8_widths) w_ok = w_ok || w == L.cout;
        const bool s_ok = net.fp8_stages == 0 || ((net.fp8_stages >> L.stage) & 1u);
        L.fp8_on = net.fp8 && net.fp8_calibrated && !net.calibrating && L.d_w8 != nullptr && w_ok && s_ok && L.stage >= 2 &&
                   T[op.in].twin >= 0 && tt_eligible(net, op, sb);
    }
    for (Tensor& t : T) { t.first = -1; t.last = -1; }
    for (size_t i = 0; i < net.ops.size(); ++i) {       // lifetimes over the active ops
        const Op& op = net.ops[i];
        if (!op_active(net, op)) continue;
        auto use = [&](int t) { if (t >= 0) T[t].last = std::max(T[t].last, (int)i); };
        use(op.in); use(op.res); use(op.base); use(op.dims_from); use(op.head_direct);
        for (int s2 = 0; s2 < op.nsrc; ++s2) use(op.srcs[s2]);
        for (int s2 = 0; s2 < op.head_nsrc; ++s2) use(op.head_src[s2]);
        for (int s2 = 0; s2 < op.head_nfold; ++s2) use(op.head_fold[s2]);
        if (op.out >= 0) { if (T[op.out].first < 0) T[op.out].first = (int)i; T[op.out].last = std::max(T[op.out].last, (int)i); }
    }
    for (Tensor& t : T) t.last_read = t.last;
    {   // Two consecutive convolutions of which the second reads the first one's output may run as ONE kernel (forward_impl: the fused
        // BasicBlocks, layer1's Bottleneck seams conv3 + next conv1, block 0's downsample tail).  That kernel reads the FIRST op's inputs
        // while it already writes the SECOND op's output, so for every such pair -- a superset of what the executor really fuses: the
        // predicates there depend on packing and sizes -- the first op's inputs outlive the second op and the second op's output exists
        // from the first op on.  (Until round 4 the second output was placed at step i + 1, after the first op's inputs had been
        // released: that they never overlapped was an accident of the first-fit geometry -- ADVICE r4.)
        int prev = -1;
        for (size_t i = 0; i < net.ops.size(); ++i) {
            const Op& op = net.ops[i];
            if (!op_active(net, op)) continue;
            if (prev >= 0 && op.type == OP_CONV && net.ops[prev].type == OP_CONV) {
                const Op& a = net.ops[prev];
                if (a.out >= 0 && (op.in == a.out || op.res == a.out)) {
                    auto keep = [&](int t) { if (t >= 0 && T[t].first >= 0) T[t].last = std::max(T[t].last, (int)i); };
                    keep(a.in); keep(a.res); keep(a.out);
                    if (op.out >= 0 && T[op.out].first > prev) T[op.out].first = prev;
                }
            }
            prev = (int)i;
        }
    }
    {   // the members of a launch group run concurrently: a tensor one of them reads must outlive ALL of them, and their
        // outputs must all exist from the first member on
        std::vector<int> gfirst(net.ops.size()), glast(net.ops.size());
        for (size_t i = 0; i < net.ops.size(); ++i) { gfirst[i] = glast[i] = (int)i; }
        for (size_t i = 0; i < net.ops.size();) {
            size_t j = i + 1;
            if (net.ops[i].launch_group >= 0)
                while (j < net.ops.size() && net.ops[j].launch_group == net.ops[i].launch_group) ++j;
            for (size_t k = i; k < j; ++k) { gfirst[k] = (int)i; glast[k] = (int)j - 1; }
            i = j;
        }
        for (Tensor& t : T) { if (t.first >= 0) t.first = gfirst[t.first]; if (t.last >= 0) t.last = glast[t.last]; }
    }
    {   // C5: an e4m3 twin lives exactly as long as its bf16 tensor, and only while an fp8 convolution reads it;
        // who writes each tensor, and whether anybody still reads the bf16 version (residuals, fuse layers, bf16 convs)
        net.producer.assign(T.size(), -1);
        net.need_bf16.assign(T.size(), 0);
        std::vector<char> twin_used(T.size(), 0);
        for (const Op& op : net.ops)
            if (op_active(net, op) && op.in >= 0 && op.type == OP_CONV && (net.layers[op.conv].fp8_on || net.layers[op.conv].x3_on) && T[op.in].twin >= 0)
                twin_used[op.in] = 1;
        for (size_t i = 0; i < net.ops.size(); ++i) {
            Op& op = net.ops[i];
            if (!op_active(net, op)) continue;
            if (op.out >= 0 && net.producer[op.out] < 0) net.producer[op.out] = (int)i;
            auto bf = [&](int t) { if (t >= 0) net.need_bf16[t] = 1; };
            // bf16x3: the second convolution of a BasicBlock takes its residual from the block input's split twin (the first convolution
            // read it), so that inside a chain of blocks nobody needs -- and no epilogue writes -- the fp32 form
            op.res_twin = net.x3_res_twin && op.type == OP_CONV && op.res >= 0 && net.layers[op.conv].x3_on && T[op.res].twin >= 0 && twin_used[op.res] &&
                          op.out_coff == 0 && T[op.out].C == net.layers[op.conv].cout && T[op.res].C == net.layers[op.conv].cout;
            if (!op.res_twin) bf(op.res);
            bf(op.base); bf(op.dims_from); bf(op.head_direct);
            for (int k = 0; k < op.nsrc; ++k) bf(op.srcs[k]);
            for (int k = 0; k < op.head_nsrc; ++k) bf(op.head_src[k]);
            for (int k = 0; k < op.head_nfold; ++k) bf(op.head_fold[k]);
            if (op.in >= 0) {
                const bool f8 = op.type == OP_CONV && (net.layers[op.conv].fp8_on || net.layers[op.conv].x3_on) && T[op.in].twin >= 0;
                if (f8) twin_used[op.in] = 1; else bf(op.in);
            }
        }
        for (size_t t = 0; t < T.size(); ++t)
            if (T[t].twin >= 0) {
                Tensor& w = T[T[t].twin];
                if (twin_used[t]) { w.first = T[t].first; w.last = T[t].last; } else { w.first = w.last = -1; }
            }
    }
    for (Tensor& t : T) if (t.twin >= 0) { T[t.twin].H = t.H; T[t.twin].W = t.W; }
    // first-fit allocator over op order
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;
    size_t top = 0;
    auto alloc = [&](size_t bytes) -> size_t {
        bytes = (bytes + 255) & ~(size_t)255;
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].size >= bytes) {
                const size_t off = free_list[i].off;
                free_list[i].off += bytes; free_list[i].size -= bytes;
                if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
                return off;
            }
        const size_t off = top; top += bytes; return off;
    };
    auto release = [&](size_t off, size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        free_list.push_back({off, bytes});
        std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_list.size();)
            if (free_list[i].off + free_list[i].size == free_list[i + 1].off) { free_list[i].size += free_list[i + 1].size; free_list.erase(free_list.begin() + i + 1); }
            else ++i;
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) { top = free_list.back().off; free_list.pop_back(); }
    };
    for (size_t i = 0; i < net.ops.size(); ++i) {
        for (size_t t = 0; t < T.size(); ++t)
            if (T[t].first == (int)i) {
                T[t].bytes = (size_t)sb * T[t].H * T[t].W * T[t].C * (T[t].f32 ? 4 : T[t].fp8 ? 1 : net.esize);
                T[t].offset = alloc(T[t].bytes);
            }
        for (size_t t = 0; t < T.size(); ++t)
            if (T[t].last == (int)i && T[t].first >= 0) release(T[t].offset, T[t].bytes);
    }
    // peak = max end offset
    size_t peak = 0;
    for (const Tensor& t : T) if (t.first >= 0) peak = std::max(peak, t.offset + ((t.bytes + 255) & ~(size_t)255));
    net.lay_bytes = peak;
    net.lay_sb = sb; net.lay_h = H; net.lay_w = W;
    return SNCAL_OK;
}

// parameters, kernel variant and dynamic LDS size of one convolution op (no launch)
int prepare_conv(sncal_hrnet& net, const Op& op, int sb, char* ws, ConvParams& p, const ConvVariant*& bestv, size_t& best_lds) {
    const ConvLayer& L = net.layers[op.conv];
    const Tensor& ti = net.tensors[op.in];
    const Tensor& to = net.tensors[op.out];
    memset(&p, 0, sizeof(p));
    p.range = net.d_range;
    p.in = ws + ti.offset; p.out = ws + to.offset;
    p.res = op.res >= 0 ? ws + net.tensors[op.res].offset : nullptr;
    p.w = L.d_w; p.bias = L.d_bias;
    p.N = sb; p.Hin = ti.H; p.Win = ti.W; p.Cin = ti.C;
    p.Hout = to.H; p.Wout = to.W; p.cout_frags = L.cout_frags; p.cout = L.cout;
    p.out_cstride = to.C; p.out_coff = op.out_coff;
    p.cin_chunks = L.chunks; p.relu = op.relu ? 1 : 0; p.out_f32 = op.out_f32 ? 1 : 0;
    // one IMAGE of the input / output tensor is a buffer-descriptor range in the kernel (int byte counts, out-of-range sentinel 0x80000000)
    if ((size_t)ti.H * ti.W * ti.C * 4 >= (1u << 31) || (size_t)to.H * to.W * to.C * 4 >= (1u << 31)) {
        set_error("conv %s: one image of a tensor reaches 2 GB (%dx%dx%d -> %dx%dx%d): unsupported", L.name.c_str(), ti.H, ti.W, ti.C, to.H, to.W, to.C);
        return SNCAL_ERR_UNSUPPORTED;
    }
    // pick NI / tile shape / sub-tiles per weight chunk for this spatial size
    bestv = nullptr;
    best_lds = 0;
    int best_twf = 1; double best_score = -1;
    static const int force_ni = getenv("SNCAL_FORCE_NI") ? atoi(getenv("SNCAL_FORCE_NI")) : 0;   // tuning aids
    static const double three_gain = getenv("SNCAL_THREE_GAIN") ? atof(getenv("SNCAL_THREE_GAIN")) : 1.15;
    bool has_forced = false;
    for (int v = 0; v < net.nvariants; ++v) {
        const ConvVariant& V = net.variants[v];
        if (V.ks == L.k && V.stride == L.stride && V.mi == L.mi && V.g == L.g && V.ni == force_ni) has_forced = true;
    }
    for (int v = 0; v < net.nvariants; ++v) {
        const ConvVariant& V = net.variants[v];
        if (V.ks != L.k || V.stride != L.stride || V.mi != L.mi || V.g != L.g) continue;
        if (force_ni && has_forced && L.k == 3 && L.stride == 1 && V.ni != force_ni) continue;
        const int F = 4 * V.ni;
        const size_t wchunk = (size_t)conv_nks(V.ks, V.g) * V.mi * 1024;
        static const int force_twf = getenv("SNCAL_FORCE_TWF") ? atoi(getenv("SNCAL_FORCE_TWF")) : 0;   // tuning aid
        for (int twf = 1; twf <= F; twf *= 2) {
            if (F % twf) continue;
            if (force_twf && L.k == 3 && L.stride == 1 && L.cin >= 96 && twf != force_twf) continue;
            const int th = F / twf;
            const size_t lds = conv_stage_bytes(V.ks, V.stride, V.ni, V.mi, V.g, twf);
            if (lds > 160 * 1024 || lds - wchunk > 64 * 1024) continue;   // halo tiles are capped at 64 DMA pieces
            const long ty = (to.H + th - 1) / th, tx = (to.W + 16 * twf - 1) / (16 * twf);
            const double eff = (double)to.H * to.W / ((double)ty * th * tx * 16 * twf);
            const long blocks = ty * tx * sb * L.nblk;
            const int per_cu = (int)std::min<size_t>(conv_resident_wgs(V.ks, V.ni, V.mi, V.g), (160 * 1024) / lds);
            const double fill = std::min(1.0, (double)blocks / (256.0 * per_cu));
            const double reuse = (double)(V.mi * V.ni) / (V.mi + V.ni);       // MFMAs per LDS fragment read
            // exposed staging latency is hidden by co-resident workgroups only (see conv.hpp)
            const double overlap = per_cu >= 3 ? three_gain : per_cu >= 2 ? 1.0 : 0.55;
            const double score = eff * (0.3 + 0.7 * fill) * std::pow(reuse, 0.6) * overlap;
            if (score > best_score + 1e-9) { best_score = score; bestv = &V; best_twf = twf; best_lds = lds; }
        }
    }
    if (!bestv) { set_error("no conv variant for %s (k=%d s=%d mi=%d g=%d)", L.name.c_str(), L.k, L.stride, L.mi, L.g); return SNCAL_ERR_STATE; }
    const int th = 4 * bestv->ni / best_twf;
    p.twf = best_twf;
    p.twf_log2 = 0;
    while ((1 << p.twf_log2) < best_twf) ++p.twf_log2;
    p.halo_w_magic = 0xFFFFFFFFu / (unsigned)((16 * best_twf - 1) * L.stride + L.k) + 1u;
    p.tiles_x = (to.W + 16 * best_twf - 1) / (16 * best_twf);
    p.tiles_y = (to.H + th - 1) / th;
    {   // LDS-transposed epilogue: the fp32 tile of the 4 waves is staged in the (grown, if that keeps two
        // workgroups per CU) dynamic LDS; needs whole 8-channel groups
        static const int epi = getenv("SNCAL_EPI_LDS") ? atoi(getenv("SNCAL_EPI_LDS")) : 1;
        const int wgs = conv_resident_wgs(L.k, bestv->ni, L.mi, L.g);
        const size_t need = (size_t)4 * conv_epi_frags(L.k, bestv->ni, L.mi, L.g) * 16 * (L.mi * 16 + 4) * 4;
        // bf16: whole 8-channel groups; fp32 / bf16x3 engines (epilogue F): whole 4-channel groups (SNCAL_EPI_F32=0: the direct epilogue)
        static const int epi32 = getenv("SNCAL_EPI_F32") ? atoi(getenv("SNCAL_EPI_F32")) : 1;
        const bool shape_ok = net.dtype == SNCAL_BF16 ? (!op.out_f32 && L.cout % 8 == 0 && to.C % 8 == 0 && op.out_coff % 8 == 0)
                                                     : (epi32 && L.cout % 4 == 0 && to.C % 4 == 0 && op.out_coff % 4 == 0);
        const size_t now_per_cu = std::min<size_t>(wgs, (160 * 1024) / best_lds);
        const bool fits = need <= best_lds || need <= (160 * 1024) / now_per_cu || need <= 52 * 1024;
        p.epi_lds = (epi && shape_ok && fits) ? 1 : 0;
        if (p.epi_lds && need > best_lds) best_lds = need;
    }
    { static const int abl = getenv("SNCAL_ABLATE") ? atoi(getenv("SNCAL_ABLATE")) : 0; p.ablate = abl; }
    p.w_bytes = (unsigned)((size_t)L.nblk * L.chunks * conv_nks(L.k, L.g) * L.mi * 1024);
    { static const int extra = getenv("SNCAL_EXTRA_LDS") ? atoi(getenv("SNCAL_EXTRA_LDS")) : 0; best_lds = std::min<size_t>(best_lds + extra, 160 * 1024); }
    { static const bool dbg = getenv("SNCAL_CONV_DEBUG") != nullptr;
      if (dbg) fprintf(stderr, "[conv] %-44s %dx%d cin %d cout %d: NI%d MI%d G%d twf %d lds %zu grid %dx%d epi_lds %d\n", L.name.c_str(), to.H, to.W, L.cin, L.cout,
                       bestv->ni, L.mi, L.g, best_twf, best_lds, p.tiles_x * p.tiles_y * sb, L.nblk, p.epi_lds); }
    p.nblk = L.nblk;
    p.n_work = (unsigned)(p.tiles_x * p.tiles_y * sb * L.nblk);
    p.per_xcd = (p.n_work + 7) / 8;
    p.nblk_magic = conv_magic((unsigned)L.nblk); p.tiles_x_magic = conv_magic((unsigned)p.tiles_x); p.tiles_y_magic = conv_magic((unsigned)p.tiles_y);
    return SNCAL_OK;
}

void conv_profile_entry(sncal_hrnet& net, const Op& op, int sb, const ConvVariant* bestv, bool add) {
    const ConvLayer& L = net.layers[op.conv];
    const Tensor& ti = net.tensors[op.in];
    const Tensor& to = net.tensors[op.out];
    net.last_kernel = fmt("conv<%s,k%d,s%d,NI%d,MI%d,G%d>", net.dtype == SNCAL_BF16 ? "bf16" : net.x3_generic ? SNCAL_X3_NAME : "f32", L.k, L.stride, bestv ? bestv->ni : 0, L.mi, L.g);
    static const bool detail = getenv("SNCAL_PROFILE_DETAIL") != nullptr;      // tuning aid: one profile row per layer shape
    if (detail) net.last_kernel += fmt("@%dx%d:%d->%d%s", to.H, to.W, L.cin, L.cout, op.res >= 0 ? "+res" : "");
    const double px = (double)sb * to.H * to.W;
    if (!add) { net.last_flops = 0; net.last_bytes = 0; }
    net.last_flops += 2.0 * px * L.cout * L.cin * L.k * L.k;
    net.last_bytes += (double)sb * ti.H * ti.W * ti.C * net.esize + px * L.cout * (op.out_f32 ? 4 : net.esize) * (op.res >= 0 ? 2 : 1) +
                      (double)L.cout * L.cin * L.k * L.k * net.esize;
}

// ---- two-team persistent kernel for the wide 3x3 stride-1 convolutions (conv_tt.hip) --------------------------------
bool tt_eligible(const sncal_hrnet& net, const Op& op, int sb) {
    if (!net.use_conv_tt || op.type != OP_CONV) return false;
    const ConvLayer& L = net.layers[op.conv];
    const Tensor& ti = net.tensors[op.in];
    const Tensor& to = net.tensors[op.out];
    if (net.x3) {                                   // bf16x3 engine: fp32 tensors, split twin in, fp32 out
        if (net.dtype != SNCAL_F32 || !L.d_w_x3 || ti.C != L.cin || ti.twin < 0 || to.C % 8 || op.out_coff % 8) return false;
        // every whole-tensor byte count the kernel forms (input twin, fp32 output, output twin, residual) stays below 2^31: its epilogue
        // relies on voffset 0x80000000 + soffset being out of range for masked lanes, which only holds for ranges below that
        const size_t in_bytes = (size_t)sb * ti.H * ti.W * ti.C * 4, out_bytes = (size_t)sb * to.H * to.W * to.C * 4;
        const size_t twin_bytes = (size_t)sb * to.H * to.W * L.cout * 4;
        return in_bytes < (1u << 31) && out_bytes < (1u << 31) && twin_bytes < (1u << 31) &&
               (size_t)((L.cout + TT_COUT - 1) / TT_COUT) * (L.cin / 16) * 9 * 6 * 1024 < (1u << 31);
    }
    if (net.dtype != SNCAL_BF16 || op.out_f32) return false;
    if (!L.d_w_tt || ti.C != L.cin) return false;                                   // packed at finalize for the eligible shapes
    if (to.C % 8 || op.out_coff % 8) return false;
    const size_t in_bytes = (size_t)sb * ti.H * ti.W * ti.C * 2, out_elems = (size_t)sb * to.H * to.W * to.C;
    return in_bytes < (1u << 31) && out_elems < (1ull << 32) && (size_t)L.nblk * L.chunks * 9 * 6 * 1024 < (1u << 31);
}

// does the active op that produced tensor t write its e4m3 twin itself (an fp8 convolution on the two-team kernel)?
bool twin_written_by_producer(const sncal_hrnet& net, int t, int sb) {
    static const bool no_twin_out = getenv("SNCAL_FP8_NO_TWIN_OUT") != nullptr;      // debugging aid: every twin through the quantise kernel
    if (no_twin_out) return false;
    const int pi = t >= 0 && t < (int)net.producer.size() ? net.producer[t] : -1;
    if (pi < 0) return false;
    const Op& po = net.ops[pi];
    if (po.type == OP_CONV && net.layers[po.conv].x3_on && tt_eligible(net, po, sb)) {        // bf16x3: the producer's epilogue writes the split twin
        static const bool x3_split_always = getenv("SNCAL_X3_NO_TWIN_OUT") != nullptr;            // (dense outputs only)
        return !x3_split_always && po.out_coff == 0 && net.tensors[po.out].C == net.layers[po.conv].cout;
    }
    if (net.x3 && net.x3_producer_twins && net.dtype == SNCAL_F32) {
        // bf16x3: the generic split-arithmetic convolution and the fp32 fuse sum write the twin of a dense output in their epilogues
        if (po.type == OP_CONV && net.x3_generic && !net.layers[po.conv].x3_on && po.out_coff == 0 && !po.out_f32 &&
            net.tensors[po.out].C == net.layers[po.conv].cout && net.layers[po.conv].cout % 16 == 0) return true;
        if (po.type == OP_UPADD && po.out_coff == 0 && net.tensors[po.out].C % 16 == 0) return true;
    }
    return po.type == OP_CONV && net.layers[po.conv].fp8_on && tt_eligible(net, po, sb);
}

// bf16x3: the split twin a generic producer (convolution / fuse sum) writes for tensor t, or null; *skip_f32 = nobody reads the fp32 form
void* producer_twin(const sncal_hrnet& net, int t, int sb, char* ws, bool* skip_f32) {
    *skip_f32 = false;
    if (!net.x3 || t < 0) return nullptr;
    const Tensor& to = net.tensors[t];
    if (to.twin < 0 || net.tensors[to.twin].first < 0 || !twin_written_by_producer(net, t, sb)) return nullptr;
    *skip_f32 = !net.need_bf16[t];
    return ws + net.tensors[to.twin].offset;
}

void tt_member(const sncal_hrnet& net, const Op& op, int sb, char* ws, TTMember& m) {
    const ConvLayer& L = net.layers[op.conv];
    const Tensor& ti = net.tensors[op.in];
    const Tensor& to = net.tensors[op.out];
    memset(&m, 0, sizeof(m));
    m.in = ws + ti.offset; m.out = ws + to.offset; m.res = op.res >= 0 ? ws + net.tensors[op.res].offset : nullptr;
    m.w = L.d_w_tt; m.bias = L.d_bias;
    m.N = sb; m.H = ti.H; m.W = ti.W; m.Cin = L.cin; m.chunks = L.cin / TT_CIN;
    m.cout = L.cout; m.out_cstride = to.C; m.out_coff = op.out_coff; m.relu = op.relu ? 1 : 0;
    m.w_bytes = (unsigned)((size_t)(L.cout / TT_COUT) * m.chunks * 9 * 6 * 1024);
    m.in_bytes = (unsigned)((size_t)sb * ti.H * ti.W * ti.C * 2);
    m.hp1_magic = 0xFFFFFFFFu / (unsigned)(ti.H + 1) + 1u;
    if (L.x3_on) {           // bf16x3: split twin in (a bf16 tensor of 2 C pseudo-channels), 16-channel stages, hi / lo weights, fp32 out
        m.in = ws + net.tensors[ti.twin].offset;
        m.in_bytes = (unsigned)((size_t)sb * ti.H * ti.W * ti.C * 4);
        m.Cin = 2 * L.cin;
        m.chunks = L.cin / 16;
        m.w = L.d_w_x3;
        m.w_bytes = (unsigned)((size_t)((L.cout + L.x3_blk - 1) / L.x3_blk) * m.chunks * 9 * 2 * (L.x3_blk / 32) * 1024);
        // outputs: the split twin when a bf16x3 convolution reads this tensor next, the fp32 tensor when anybody else does
        const bool twin_out = to.twin >= 0 && net.tensors[to.twin].first >= 0 && twin_written_by_producer(net, op.out, sb);
        m.out8 = twin_out ? ws + net.tensors[to.twin].offset : nullptr;
        if (twin_out && !net.need_bf16[op.out]) m.out = nullptr;
        if (op.res_twin) { m.res = ws + net.tensors[net.tensors[op.res].twin].offset; m.res_split = 1; }
    }
    if (L.fp8_on) {          // C5: e4m3 twin in, 64-channel stages, e4m3 weights; outputs: bf16 if anybody reads it, twin if an fp8 conv follows
        m.in = ws + net.tensors[ti.twin].offset;
        m.in_bytes = (unsigned)((size_t)sb * ti.H * ti.W * ti.C);
        m.chunks = (L.cin + 63) / 64;
        m.w = L.d_w8;
        m.w_bytes = (unsigned)((size_t)(L.cout / TT_COUT) * m.chunks * 9 * 6 * 1024);
        m.oscale = L.d_oscale;
        const bool twin_out = to.twin >= 0 && net.tensors[to.twin].first >= 0;
        m.out8 = twin_out ? ws + net.tensors[to.twin].offset : nullptr;
        m.out8_inv_scale = twin_out && to.scale > 0.f ? 1.0f / to.scale : 1.0f;
        if (!net.need_bf16[op.out]) m.out = nullptr;
        if (getenv("SNCAL_FP8_NO_TWIN_OUT")) { m.out8 = nullptr; m.out = ws + to.offset; }
    }
}

// The work items of the member convolutions as eight queues, one per XCD.  Workgroup b runs on XCD b % 8 (observed
// dispatch rule, used for speed only): every member's items -- tile-major, the 96-channel blocks of a tile adjacent --
// are cut into 8 contiguous slices, one per XCD, so that neighbouring tiles (shared halo rows) and the blocks of one
// tile (same input) meet in one L2; inside an XCD's queue the members follow each other, most expensive first
// (longest-processing-time order: the teams, which take the next item when they finish one, end within one cheap item
// of each other).  Rounds 2-4 dealt the items to the teams HERE (static lists); a workgroup whose CU was held by a
// camera-solve wavefront then started when the first other workgroup had finished, and the launch lasted twice as long.
int tt_build_plan(sncal_hrnet& net, const TTMember* mem, int n, sncal_hrnet::TTPlanDev& out, hipStream_t stream, int tile_h = TT_TH, int cout_blk = TT_COUT) {
    if (!net.n_cus) {
        int dev = 0, cus = 0;
        SNCAL_CHECK_HIP(hipGetDevice(&dev));
        SNCAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        net.n_cus = cus > 0 ? cus : 256;
    }
    const int n_wgs = net.n_cus;
    const int n_xcd = n_wgs >= 8 ? 8 : 1;                  // (fewer than 8 workgroups: every workgroup reads queue b % 8, so only queue 0.. exist)
    std::vector<std::vector<TTItem>> per_xcd(8);
    int order[TT_MAX_MEMBERS] = {0, 1, 2};
    std::sort(order, order + n, [&](int a, int b) { return mem[a].chunks > mem[b].chunks; });
    // (Interleaving the members' items, so that the memory-heavy 96-channel tiles do not all run at the tail of the launch, measured
    // 1.4 % SLOWER than member after member: 193.3 vs 190.6 us per grouped launch; the two teams of a workgroup walking the classes in
    // OPPOSITE order measured 1.7 % slower on the fp16x3 launches, round 4.)
    for (int oi = 0; oi < n; ++oi) {
        const TTMember& m = mem[order[oi]];
        const int tiles_y = (m.N * (m.H + 1) + tile_h - 1) / tile_h, tiles_x = (m.W + TT_TW - 1) / TT_TW, nblk = (m.cout + cout_blk - 1) / cout_blk;
        const long total = (long)tiles_y * tiles_x * nblk;
        for (long i = 0; i < total; ++i) {
            const int x = n_xcd == 8 ? (int)(i * 8 / total) : 0;
            TTItem it;
            it.member = (uint16_t)order[oi]; it.nb = (uint16_t)(i % nblk);
            const long tile = i / nblk;
            it.col0 = (int32_t)(tile % tiles_x) * TT_TW; it.row0 = (int32_t)(tile / tiles_x) * tile_h; it.pad_ = 0;
            per_xcd[x].push_back(it);
        }
    }
    // (Measured and not kept, round 5: the tickets of a member's slice S-way interleaved -- S = 4, 8, 16, 64 -- so that the items that share
    // halo lines are not drawn at the same instant: 32.87 / 33.06 / 32.95 / 33.10 ms per step of two-team launches against 32.85 in
    // tile-major order, FETCH_SIZE 434 against 410 MB per launch.  The queue's extra HBM reads over round 4's static deal -- 412 against
    // 310 MB per grouped launch by PMC -- are not concurrent misses of neighbouring tickets.)
    std::vector<TTItem> flat;
    std::vector<uint32_t> first(9, 0);
    for (int x = 0; x < 8; ++x) { first[x] = (uint32_t)flat.size(); flat.insert(flat.end(), per_xcd[x].begin(), per_xcd[x].end()); }
    first[8] = (uint32_t)flat.size();
    if (flat.empty()) flat.push_back(TTItem{0, 0, 0, 0, 0});
    SNCAL_CHECK_HIP(hipMalloc((void**)&out.items, flat.size() * sizeof(TTItem)));
    SNCAL_CHECK_HIP(hipMalloc((void**)&out.first, first.size() * 4));
    // on the forward's OWN stream, then a wait for that stream only: a plain hipMemcpy runs on the legacy null stream, which synchronises with
    // the pipeline's CU-masked (blocking) solve streams -- a new layout (the tail batch of a directory) then waited for every solve in flight
    SNCAL_CHECK_HIP(hipMemcpyAsync(out.items, flat.data(), flat.size() * sizeof(TTItem), hipMemcpyHostToDevice, stream));
    SNCAL_CHECK_HIP(hipMemcpyAsync(out.first, first.data(), first.size() * 4, hipMemcpyHostToDevice, stream));
    SNCAL_CHECK_HIP(hipStreamSynchronize(stream));       // (the host vectors die here; once per layout)
    out.n_wgs = n_wgs;
    // fewer than 8 pairs of items per workgroup in an XCD's list: the kernel's three-pairs-ahead ticket pipeline would starve most workgroups
    out.lazy = flat.size() < (size_t)3 * (size_t)n_wgs ? 1 : 0;      // fewer than 1.5 pairs per workgroup
    return SNCAL_OK;
}

// ticket words of a network: [0, 9) and [16, 25) the fused blocks, [32, 48) the two-team kernel, [48, 57) layer1's seams, [64, 73) the
// pipelined stride-2 kernel
constexpr int TICKET_WORDS = 96;
static int ensure_tickets(sncal_hrnet* net, hipStream_t stream) {
    if (net->d_tickets) return SNCAL_OK;
    SNCAL_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&net->d_tickets), TICKET_WORDS * sizeof(unsigned)));
    SNCAL_CHECK_HIP(hipMemsetAsync(net->d_tickets, 0, TICKET_WORDS * sizeof(unsigned), stream));
    return SNCAL_OK;
}

// the ops [ops, ops + n) (independent, all eligible, all bf16 or all fp8) as ONE launch of the two-team kernel
int run_conv_tt(sncal_hrnet& net, const Op* ops, int n, int key, int sb, char* ws, hipStream_t stream) {
    TTParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.range = net.d_range;
    const bool fp8 = net.layers[ops[0].conv].fp8_on, x3 = net.layers[ops[0].conv].x3_on;
    const sncal::LaunchEvents armed = sncal::launch_events();        // the profiling event pair belongs to the convolution launch,
    sncal::launch_events() = sncal::LaunchEvents{};                  // not to the calibration / quantisation helpers in front of it
    for (int i = 0; i < n; ++i) {
        if (net.calibrating && net.tensors[ops[i].in].twin >= 0) {        // C5 calibration: max |x| of every candidate input tensor
            const Tensor& ti = net.tensors[ops[i].in];
            const int rc = launch_absmax_bf16(ws + ti.offset, (size_t)sb * ti.H * ti.W * ti.C, net.d_amax + ops[i].in, stream);
            if (rc) return rc;
        }
        if (x3 && !twin_written_by_producer(net, ops[i].in, sb)) {       // bf16x3: the fp32 input's split twin, unless its producer wrote it
            const Tensor& ti = net.tensors[ops[i].in];
            const int rc = launch_split_f32(ws + ti.offset, ws + net.tensors[ti.twin].offset, (size_t)sb * ti.H * ti.W * ti.C, stream, net.d_range);
            if (rc) return rc;
        }
        if (fp8 && !twin_written_by_producer(net, ops[i].in, sb)) {      // first fp8 conv of a chain: quantise its input here
            const Tensor& ti = net.tensors[ops[i].in];
            const int rc = launch_quantize_fp8(ws + ti.offset, ws + net.tensors[ti.twin].offset, (size_t)sb * ti.H * ti.W * ti.C, ti.scale, stream);
            if (rc) return rc;
        }
        tt_member(net, ops[i], sb, ws, tp.m[i]);
    }
    sncal::launch_events() = armed;
    const bool cfg64 = x3 && n == 1 && net.layers[ops[0].conv].x3_blk == 64;   // bf16x3, 48-channel branch: tile 64 x 12 x 32
    if (fp8) key += 1 << 30;                                               // fp8 plans have their own stage counts
    auto it = net.tt_plans.find(key);
    if (it == net.tt_plans.end() || it->second.n_wgs == 0) {       // static per layout: built on the first forward
        sncal_hrnet::TTPlanDev pd;
        // Small launches of the split-arithmetic engine (round 5): below two 8-row items per team the launch lasts as long as its longest item
        // while most teams hold short ones or nothing -- the 96 x 4 x 32 tile (conv_tt.hip, c31) halves the items instead
        int tile_h = cfg64 ? 12 : TT_TH;
        pd.cfg = cfg64 ? 1 : 0;
        if (x3 && !cfg64) {
            const int per_team = getenv("SNCAL_TT_SMALL_ITEMS") ? atoi(getenv("SNCAL_TT_SMALL_ITEMS")) : 2;      // (read per plan: tests run both tiles in one process; 0 = never)
            long items = 0;
            for (int i = 0; i < n; ++i)
                items += (long)((tp.m[i].N * (tp.m[i].H + 1) + TT_TH - 1) / TT_TH) * ((tp.m[i].W + TT_TW - 1) / TT_TW) * ((tp.m[i].cout + TT_COUT - 1) / TT_COUT);
            int cus = net.n_cus;
            if (!cus) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); if (cus <= 0) cus = 256; }
            if (items < (long)per_team * 2 * cus) { tile_h = 4; pd.cfg = 2; }
        }
        const int rc = tt_build_plan(net, tp.m, n, pd, stream, tile_h, cfg64 ? 64 : TT_COUT);
        if (rc) return rc;
        it = net.tt_plans.insert({key, pd}).first;
    }
    const int cfg = it->second.cfg;
    tp.items = it->second.items; tp.xcd_first = it->second.first; tp.lazy = it->second.lazy;
    { const int rc = ensure_tickets(&net, stream); if (rc) return rc; }
    tp.queue = net.d_tickets + 32;
    // tuning aid: SNCAL_TT_TRACE=<file> dumps the per-team phase timestamps of the LAST launch with 3 members
    // (SNCAL_TT_TRACE_CFG64=1: of the last launch of the 64-channel tile instead)
    static const char* trace_file = getenv("SNCAL_TT_TRACE");
    static const bool trace_cfg64 = getenv("SNCAL_TT_TRACE_CFG64") && atoi(getenv("SNCAL_TT_TRACE_CFG64")) != 0;
    static const int trace_nth = getenv("SNCAL_TT_TRACE_NTH") ? atoi(getenv("SNCAL_TT_TRACE_NTH")) : -1;      // only the n-th such launch of the process
    static int trace_seen = 0;
    unsigned long long* d_trace = nullptr;
    const size_t n_trace = (size_t)it->second.n_wgs * 2 * 256;
    if (trace_file && (trace_cfg64 ? cfg64 : n == 3) && (trace_nth < 0 || trace_seen++ == trace_nth) && hipMalloc(&d_trace, n_trace * 8) == hipSuccess) { (void)hipMemsetAsync(d_trace, 0, n_trace * 8, stream); tp.trace = d_trace; }
    { static const int abl = getenv("SNCAL_TT_ABLATE") ? atoi(getenv("SNCAL_TT_ABLATE")) : 0; tp.ablate = abl; }
    launch_conv_tt(tp, it->second.n_wgs, fp8 ? 1 : x3 ? 2 : 0, stream, cfg);
    SNCAL_CHECK_LAUNCH();
    if (d_trace) {
        std::vector<unsigned long long> h(n_trace);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_trace, n_trace * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, n_trace, f); fclose(f); }
    }
    static const bool fp8_debug = getenv("SNCAL_FP8_DEBUG") != nullptr;      // tuning aid: range of every fp8 launch's bf16 outputs
    if (fp8 && fp8_debug) {
        (void)hipStreamSynchronize(stream);
        for (int i = 0; i < n; ++i) {
            const Tensor& to = net.tensors[ops[i].out];
            const Tensor& ti = net.tensors[ops[i].in];
            const size_t ne = (size_t)sb * to.H * to.W * to.C;
            std::vector<uint16_t> h(ne);
            double mx = 0, sum = 0; size_t bad = 0;
            unsigned long long ck8 = 0, ckin = 0;
            if (tp.m[i].out8) { std::vector<uint8_t> h8(ne); (void)hipMemcpy(h8.data(), tp.m[i].out8, ne, hipMemcpyDeviceToHost); for (size_t k = 0; k < ne; ++k) ck8 = ck8 * 1315423911ull + h8[k]; }
            { const size_t ni = (size_t)sb * ti.H * ti.W * ti.C; std::vector<uint8_t> h8(ni); (void)hipMemcpy(h8.data(), tp.m[i].in, ni, hipMemcpyDeviceToHost); for (size_t k = 0; k < ni; ++k) ckin = ckin * 1315423911ull + h8[k]; }
            fprintf(stderr, "[ck] %s in %016llx out8 %016llx\n", net.layers[ops[i].conv].name.c_str(), ckin, ck8);
            if (tp.m[i].out) {
                (void)hipMemcpy(h.data(), tp.m[i].out, ne * 2, hipMemcpyDeviceToHost);
                for (size_t k = 0; k < ne; ++k) { uint32_t u = (uint32_t)h[k] << 16; float f; memcpy(&f, &u, 4); if (!(std::fabs(f) < 1e3f)) { if (bad < 48) fprintf(stderr, "   bad %g at n %zu y %zu x %zu c %zu\n", f, k / ((size_t)to.H * to.W * to.C), (k / ((size_t)to.W * to.C)) % to.H, (k / to.C) % to.W, k % to.C); ++bad; } else { mx = std::max(mx, (double)std::fabs(f)); sum += std::fabs(f); } }
            }
            fprintf(stderr, "[fp8] %-44s in_scale %.4g out_scale %.4g bf16_out %d twin_out %d  |out| max %.4g mean %.4g bad %zu\n", net.layers[ops[i].conv].name.c_str(),
                    ti.scale, to.scale, tp.m[i].out != nullptr, tp.m[i].out8 != nullptr, mx, ne ? sum / ne : 0.0, bad);
        }
    }
    if (net.profiling) {
        for (int i = 0; i < n; ++i) conv_profile_entry(net, ops[i], sb, nullptr, i > 0);
        net.last_kernel = fp8 ? "conv_tt<fp8,k3,s1,8x32x96>" : cfg64 ? "conv_tt<" SNCAL_X3_NAME ",k3,s1,12x32x64>" : cfg == 2 ? "conv_tt<" SNCAL_X3_NAME ",k3,s1,4x32x96>" : x3 ? "conv_tt<" SNCAL_X3_NAME ",k3,s1,8x32x96>" : "conv_tt<bf16,k3,s1,8x32x96>";
        static const bool detail = getenv("SNCAL_PROFILE_DETAIL") != nullptr;      // tuning aid: one profile row per launch kind
        if (detail) net.last_kernel += fmt("@%d members%s%s%s", n, tp.m[0].res ? "+res" : "", tp.m[0].out ? "+f32" : "", tp.m[0].out8 ? "+twin" : "");
    }
    return SNCAL_OK;
}

int run_conv(sncal_hrnet& net, const Op& op, int sb, char* ws, hipStream_t stream) {
    const ConvLayer& L = net.layers[op.conv];
    if (tt_eligible(net, op, sb)) return run_conv_tt(net, &op, 1, (int)(&op - net.ops.data()) * 4096 + sb, sb, ws, stream);
    ConvParams p;
    const ConvVariant* bestv = nullptr;
    size_t best_lds = 0;
    const int rc = prepare_conv(net, op, sb, ws, p, bestv, best_lds);
    if (rc) return rc;
    if (net.x3_generic) {        // bf16x3: the split twin for the two-team convolution that reads this output; fp32 only if somebody reads it
        bool skip_f32 = false;
        p.out_twin = producer_twin(net, op.out, sb, ws, &skip_f32);
        if (p.out_twin && skip_f32) p.out = nullptr;
    }
    // tuning aid: SNCAL_CONV_TRACE=<layer name> dumps per-workgroup phase timestamps of that layer's last launch
    static const char* trace_name = getenv("SNCAL_CONV_TRACE");
    unsigned long long* d_trace = nullptr; size_t n_trace = 0;
    if (trace_name && L.name == trace_name) {
        n_trace = (size_t)8 * ((p.n_work + 7) / 8) * 16;
        if (hipMalloc(&d_trace, n_trace * 8) == hipSuccess) { (void)hipMemsetAsync(d_trace, 0, n_trace * 8, stream); p.trace = d_trace; }
    }
    const bool s2p = net.x3_generic && net.use_s2p && !d_trace && !p.ablate && bestv->ks == 3 && bestv->stride == 2 && bestv->ni == 2 && bestv->g == 3 &&
                     (bestv->mi == 6 || bestv->mi == 3);
    if (s2p) {               // the one-member case of the pipelined stride-2 kernel
        ConvSharedParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.p[0] = p; sp.mi[0] = bestv->mi;
        sp.first[0] = 0;
        for (int i = 1; i < 4; ++i) sp.first[i] = (unsigned)p.nblk;
        sp.n = 1;
        sp.tiles = (unsigned)(p.tiles_x * p.tiles_y * sb);
        sp.tiles_per_xcd = (sp.tiles + 7) / 8;
        { const int rc2 = ensure_tickets(&net, stream); if (rc2) return rc2; }
        const int rc2 = launch_conv_s2p_x3(sp, net.d_tickets + 64, stream);
        if (rc2) return rc2;
        if (net.profiling) { conv_profile_entry(net, op, sb, bestv, false); static const bool detail = getenv("SNCAL_PROFILE_DETAIL") != nullptr; if (!detail) net.last_kernel = "conv_s2p<" SNCAL_X3_NAME ",k3,s2>"; }
        return SNCAL_OK;
    }
    bestv->launch(p, dim3(8 * p.per_xcd), best_lds, stream);
    SNCAL_CHECK_LAUNCH();
    if (d_trace) {
        std::vector<unsigned long long> h(n_trace);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_trace, n_trace * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_trace);
        if (FILE* f = fopen(getenv("SNCAL_CONV_TRACE_FILE") ? getenv("SNCAL_CONV_TRACE_FILE") : "conv_trace.bin", "wb")) { fwrite(h.data(), 8, n_trace, f); fclose(f); }
    }
    if (net.profiling) conv_profile_entry(net, op, sb, bestv, false);
    return SNCAL_OK;
}

// the members of a launch group (independent convs, consecutive ops): one grouped launch when they agree on the kernel
// variant and that variant has a grouped instantiation; otherwise *done = false and the caller runs them one by one
int run_conv_group(sncal_hrnet& net, const Op* ops, int n, int sb, char* ws, hipStream_t stream, bool* done) {
    *done = false;
    if (n < 2 || n > 3) return SNCAL_OK;
    {
        bool all_tt = true;
        int couts = 0;
        for (int i = 0; i < n; ++i) {
            all_tt = all_tt && tt_eligible(net, ops[i], sb) && net.layers[ops[i].conv].fp8_on == net.layers[ops[0].conv].fp8_on;
            couts += (net.layers[ops[i].conv].cout + TT_COUT - 1) / TT_COUT * TT_COUT;
            all_tt = all_tt && !(net.layers[ops[i].conv].x3_on && net.layers[ops[i].conv].x3_blk != TT_COUT);
        }
        static const bool fp8_singles = getenv("SNCAL_FP8_SINGLES") != nullptr;          // debugging aid
        bool any_fp8 = false;
        for (int i = 0; i < n; ++i) any_fp8 = any_fp8 || net.layers[ops[i].conv].fp8_on;
        bool all_fp8 = true;
        for (int i = 0; i < n; ++i) all_fp8 = all_fp8 && net.layers[ops[i].conv].fp8_on;
        // a group that mixes fp8 and bf16 members (layer selection by width) cannot share one launch: *done stays false and the
        // caller runs the members one by one (run_conv)
        if (any_fp8 && (!all_fp8 || fp8_singles)) return SNCAL_OK;
        if (all_tt && couts <= TT_COUT_MAX) {      // (the tables' last 16 floats carry the ticket queue's slots)
            const int rc = run_conv_tt(net, ops, n, (int)(ops - net.ops.data()) * 4096 + sb, sb, ws, stream);
            *done = rc == SNCAL_OK;
            return rc;
        }
    }
    if (ops[0].shared_in) {
        // the chain-starting stride-2 convolutions of one input tensor (schedule_fuse_section): one launch, tile-major (conv.hpp)
        if (!net.x3_generic) return SNCAL_OK;                 // (the other engines run them one by one)
        ConvSharedParams sp;
        memset(&sp, 0, sizeof(sp));
        const ConvVariant* vs[3] = {nullptr, nullptr, nullptr};
        size_t lds_s = 0;
        unsigned ipt = 0;
        for (int i = 0; i < n; ++i) {
            size_t l = 0;
            const int rc = prepare_conv(net, ops[i], sb, ws, sp.p[i], vs[i], l);
            if (rc) return rc;
            bool skip_f32 = false;
            sp.p[i].out_twin = producer_twin(net, ops[i].out, sb, ws, &skip_f32);
            if (sp.p[i].out_twin && skip_f32) sp.p[i].out = nullptr;
            const ConvVariant* v = vs[i];
            const bool ok = ops[i].in == ops[0].in && v->ks == 3 && v->stride == 2 && v->ni == 2 && v->g == 3 && (v->mi == 6 || v->mi == 3) &&
                            sp.p[i].tiles_x == sp.p[0].tiles_x && sp.p[i].tiles_y == sp.p[0].tiles_y && sp.p[i].twf == sp.p[0].twf;
            if (!ok) return SNCAL_OK;                          // *done stays false: one by one
            sp.mi[i] = v->mi;
            sp.first[i] = ipt;
            ipt += (unsigned)sp.p[i].nblk;
            lds_s = std::max(lds_s, l);
        }
        for (int i = n; i < 4; ++i) sp.first[i] = ipt;
        sp.n = n;
        sp.tiles = (unsigned)(sp.p[0].tiles_x * sp.p[0].tiles_y * sb);
        sp.tiles_per_xcd = (sp.tiles + 7) / 8;
        if (net.use_s2p && !sp.p[0].ablate) {
            { const int rc2 = ensure_tickets(&net, stream); if (rc2) return rc2; }
            const int rc2 = launch_conv_s2p_x3(sp, net.d_tickets + 64, stream);
            if (rc2) return rc2;
        } else {
            launch_conv_shared_s2_x3(sp, 8u * sp.tiles_per_xcd * ipt, lds_s, stream);
            SNCAL_CHECK_LAUNCH();
        }
        if (net.profiling) {
            for (int i = 0; i < n; ++i) conv_profile_entry(net, ops[i], sb, vs[i], i > 0);
            const Tensor& ti = net.tensors[ops[0].in];
            net.last_bytes -= (double)(n - 1) * sb * ti.H * ti.W * ti.C * net.esize;      // the members' common input counts once
            static const bool detail = getenv("SNCAL_PROFILE_DETAIL") != nullptr;
            if (!detail) net.last_kernel = net.use_s2p ? "conv_s2p_shared<" SNCAL_X3_NAME ",k3,s2>" : "conv_shared_s2<" SNCAL_X3_NAME ",k3,NI2,G3>";     // (its own row: not the first member's variant)
        }
        *done = true;
        return SNCAL_OK;
    }
    ConvGroupParams gp;
    memset(&gp, 0, sizeof(gp));
    const ConvVariant* v0 = nullptr;
    size_t lds = 0;
    ConvParams mp[3];
    double cost[3];
    for (int i = 0; i < n; ++i) {
        const ConvVariant* v = nullptr;
        size_t l = 0;
        const int rc = prepare_conv(net, ops[i], sb, ws, mp[i], v, l);
        if (rc) return rc;
        if (net.x3_generic) {
            bool skip_f32 = false;
            mp[i].out_twin = producer_twin(net, ops[i].out, sb, ws, &skip_f32);
            if (mp[i].out_twin && skip_f32) mp[i].out = nullptr;
        }
        if (i == 0) v0 = v;
        if (v != v0 || !v->launch_group) return SNCAL_OK;
        lds = std::max(lds, l);
        cost[i] = (double)net.layers[ops[i].conv].chunks;            // K-chunks per work item
    }
    int order[3] = {0, 1, 2};
    std::sort(order, order + n, [&](int a, int b) { return cost[a] > cost[b]; });      // longest items first
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
        gp.p[i] = mp[order[i]];
        gp.per_xcd[i] = (gp.p[i].n_work + 7) / 8;
        blocks += gp.per_xcd[i];
    }
    gp.n = n;
    v0->launch_group(gp, dim3(8 * blocks), lds, stream);
    SNCAL_CHECK_LAUNCH();
    if (net.profiling) for (int i = 0; i < n; ++i) conv_profile_entry(net, ops[i], sb, v0, i > 0);
    *done = true;
    return SNCAL_OK;
}

}  // namespace

extern "C" int sncal_hrnet_create(const sncal_hrnet_desc* desc, int dtype, sncal_hrnet** out) {
    SNCAL_CHECK_ARG(desc && out, "sncal_hrnet_create: null pointer");
    SNCAL_CHECK_ARG(dtype == SNCAL_F32 || dtype == SNCAL_BF16 || dtype == SNCAL_FP8 || dtype == SNCAL_BF16X3, "sncal_hrnet_create: dtype %d", dtype);
    SNCAL_CHECK_ARG(desc->stem_width == 64, "stem_width must be 64 (layer1 input is hard-coded, hrnet.py:273)");
    SNCAL_CHECK_ARG(desc->num_classes >= 2 && desc->num_classes <= 64, "num_classes %d out of range", desc->num_classes);
    SNCAL_CHECK_ARG(desc->upscale == 1 || desc->upscale == 2, "upscale must be 1 or 2");
    for (int s = 0; s < 3; ++s) {
        SNCAL_CHECK_ARG(desc->num_branches[s] == s + 2, "stage%d must have %d branches", s + 2, s + 2);
        SNCAL_CHECK_ARG(desc->num_modules[s] >= 1 && desc->num_blocks[s] >= 1, "stage%d: modules/blocks", s + 2);
        for (int b = 0; b < desc->num_branches[s]; ++b)
            SNCAL_CHECK_ARG(desc->num_channels[s][b] > 0 && desc->num_channels[s][b] % 16 == 0,
                            "stage%d branch %d: width %d must be a multiple of 16", s + 2, b, desc->num_channels[s][b]);
    }
    SNCAL_CHECK_ARG(desc->stage1_blocks >= 1 && desc->stage1_channels % 16 == 0, "stage1 config");
    sncal_hrnet* net = new sncal_hrnet();
    net->desc = *desc;
    net->fp8 = dtype == SNCAL_FP8;            // C5: the bf16 engine with e4m3 arithmetic in the wide 3x3 stride-1 convolutions
    if (net->fp8) dtype = SNCAL_BF16;
    net->x3 = dtype == SNCAL_BF16X3;          // the fp32 engine with split-bf16 3x3 convolutions
    if (net->x3) dtype = SNCAL_F32;
    net->dtype = dtype;
    net->ge = dtype == SNCAL_BF16 ? 8 : 4;
    net->esize = dtype == SNCAL_BF16 ? 2 : 4;
    // bf16x3 engine: the generic kernel too multiplies in split-bf16 arithmetic (x3_t, conv.hpp); SNCAL_X3_GENERIC=0 keeps its exact-fp32 variants
    net->x3_res_twin = !(getenv("SNCAL_X3_RES_TWIN") && atoi(getenv("SNCAL_X3_RES_TWIN")) == 0);
    net->x3_generic = net->x3 && !(getenv("SNCAL_X3_GENERIC") && atoi(getenv("SNCAL_X3_GENERIC")) == 0);
    net->variants = dtype == SNCAL_BF16 ? conv_variants_bf16(&net->nvariants) : net->x3_generic ? conv_variants_x3(&net->nvariants) : conv_variants_f32(&net->nvariants);
    if (const char* e = getenv("SNCAL_SUBBATCH")) { const int v = atoi(e); if (v > 0) net->subbatch = v; }
    if (const char* e = getenv("SNCAL_FUSED_HEAD")) net->fused_enabled = atoi(e) != 0;
    Builder b(*net);
    if (!b.build()) { delete net; return SNCAL_ERR_STATE; }
    *out = net;
    return SNCAL_OK;
}

extern "C" void sncal_hrnet_destroy(sncal_hrnet* net) {
    if (!net) return;
    for (auto& kv : net->tt_plans) { (void)hipFree(kv.second.items); (void)hipFree(kv.second.first); }
    for (ConvLayer& L : net->layers) { if (L.d_w) (void)hipFree(L.d_w); if (L.d_bias) (void)hipFree(L.d_bias); if (L.d_w_tt) (void)hipFree(L.d_w_tt); if (L.d_w8) (void)hipFree(L.d_w8); if (L.d_w_x3) (void)hipFree(L.d_w_x3); if (L.d_w_bbx) (void)hipFree(L.d_w_bbx); if (L.d_w_bnp) (void)hipFree(L.d_w_bnp); if (L.d_oscale) (void)hipFree(L.d_oscale); }
    for (hipEvent_t e : net->event_pool) (void)hipEventDestroy(e);
    if (net->d_tickets) (void)hipFree(net->d_tickets);
    if (net->d_range) (void)hipFree(net->d_range);
    if (net->d_amax) (void)hipFree(net->d_amax);
    for (void* q : {net->d_hw0, net->d_hw1, net->d_hw0_32, net->d_hw1_32, net->d_hw0_32l, net->d_hw1_32l, (void*)net->d_hb0, (void*)net->d_hb1}) if (q) (void)hipFree(q);
    delete net;
}

extern "C" int sncal_hrnet_num_convs(const sncal_hrnet* net) { return net ? net->n_public : 0; }

extern "C" int sncal_hrnet_conv_info(const sncal_hrnet* net, int idx, char* name, int name_cap, char* bn_name, int bn_cap,
                                     int* cin, int* cout, int* ksize, int* stride, int* has_bias) {
    SNCAL_CHECK_ARG(net && idx >= 0 && idx < net->n_public, "sncal_hrnet_conv_info: index %d", idx);
    const ConvLayer& L = net->layers[idx];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", L.name.c_str());
    if (bn_name && bn_cap > 0) snprintf(bn_name, bn_cap, "%s", L.bn.c_str());
    if (cin) *cin = L.cin;
    if (cout) *cout = L.cout;
    if (ksize) *ksize = L.k;
    if (stride) *stride = L.stride;
    if (has_bias) *has_bias = L.bias ? 1 : 0;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_set_conv(sncal_hrnet* net, int idx, const float* h_weight, const float* h_scale, const float* h_shift) {
    SNCAL_CHECK_ARG(net && idx >= 0 && idx < net->n_public, "sncal_hrnet_set_conv: index %d", idx);
    SNCAL_CHECK_ARG(h_weight && h_shift, "sncal_hrnet_set_conv: null weights");
    ConvLayer& L = net->layers[idx];
    const size_t nw = (size_t)L.cout * L.cin * L.k * L.k;
    L.w.assign(h_weight, h_weight + nw);
    L.scale.assign(L.cout, 1.0f);
    if (h_scale) L.scale.assign(h_scale, h_scale + L.cout);
    L.shift.assign(h_shift, h_shift + L.cout);
    L.is_set = true;
    net->finalized = false;
    net->equalize_done = false;
    return SNCAL_OK;
}

// The split-fp16 engine (fp16x3) carries every operand as fp16 hi + fp16 lo: 22 significand bits for |v| in [2^-3, 65504], an ABSOLUTE
// resolution of 2^-25 below 2^-3 (lo is subnormal there), a hard clamp at +-65504 above (x3.hpp).  The reference's predict() is fp32 with
// no such limits (src/models/hrnet/metamodel.py:127-134), and a trained checkpoint may fold a near-dead BatchNorm channel
// (running_var ~ 0 -> scale gamma / sqrt(eps) = 316 gamma) into its weights.  So the engine refuses what it cannot represent instead of
// clamping it silently (x3_split_host saturates): any folded weight beyond 65504, or a layer whose weights sit so low that most of its
// weight mass has lost more than half of the 22 bits (|w| < 2^-14: hi itself is subnormal).  The caller falls back to dtype fp32
// (load_model does it by itself and says so).  SNCAL_X3_RANGE_CHECK=0 switches the refusal off (tests of the run-time range flag).
static int x3_range_check(const sncal_hrnet& net) {
#if SNCAL_X3_F16
    if (!net.x3) return SNCAL_OK;
    static const bool off = getenv("SNCAL_X3_RANGE_CHECK") && atoi(getenv("SNCAL_X3_RANGE_CHECK")) == 0;
    if (off) return SNCAL_OK;
    for (const ConvLayer& L : net.layers) {
        if (!L.is_set || L.w.empty()) continue;
        const size_t per = L.w.size() / (size_t)L.cout;
        double mx = 0, mass = 0, low = 0;
        int mx_co = 0;
        bool outside = false;                                             // a folded weight beyond the range, infinite or NaN
        for (int co = 0; co < L.cout && !outside; ++co) {
            const double sc = L.scale.empty() ? 1.0 : (double)L.scale[co];
            for (size_t i = 0; i < per; ++i) {
                const double v = std::fabs((double)L.w[(size_t)co * per + i] * sc);
                if (!(v <= 65504.0)) { mx = v; mx_co = co; outside = true; break; }
                if (v > mx) { mx = v; mx_co = co; }
                mass += v;
                if (v < 6.103515625e-05) low += v;                        // 2^-14: fp16's smallest normal
            }
        }
        if (outside) {
            set_error("fp16x3 engine: folded weight %.6g of conv %s (output channel %d, BatchNorm scale %.6g) is outside the fp16 range "
                      "(65504): this checkpoint needs dtype='fp32'", mx, L.name.c_str(), mx_co, L.scale.empty() ? 1.0 : (double)L.scale[mx_co]);
            return SNCAL_ERR_RANGE;
        }
        if (mass > 0 && low > 0.5 * mass) {
            set_error("fp16x3 engine: %.0f %% of the folded weight mass of conv %s lies below 2^-14 (largest weight %.3g): fp16 halves keep fewer "
                      "than 11 of fp32's 24 bits there: this checkpoint needs dtype='fp32'", 100.0 * low / mass, L.name.c_str(), mx);
            return SNCAL_ERR_RANGE;
        }
    }
#endif
    return SNCAL_OK;
}

// Power-of-two rebalancing of block-internal channels for the split-fp16 engine.  fp16 halves carry 22 significand bits only for |v| in
// [2^-3, 65504] and an ABSOLUTE 2^-25 below: a product w.x loses relative precision 2^-25 (1/|w| + 1/|x|), smallest when the weight and
// the activation it meets are of one size.  A trained checkpoint need not be balanced -- a BatchNorm with a small gamma in front of a
// convolution with large weights is the same function as the reverse (the reference computes in fp32 and cannot tell,
// src/models/hrnet/metamodel.py:127-134) -- and measured on a four-decade spread the engine drifted to |dlogp| 5e-3 with no flag
// (tests/test_range_guard_gpu.py).  Inside a block the balance is free to choose, EXACTLY: the output of conv1 + bn1 + ReLU of a BasicBlock
// (conv1 / conv2 of a Bottleneck) feeds one convolution only (src/models/hrnet/hrnet.py:42-58, 79-99), ReLU commutes with a positive factor,
// so row c of the producer (folded scale and shift) x 1/q_c and column c of the consumer x q_c, q_c a power of two, is the same network bit
// for bit in fp32.  m_c = size of the consumer column's large folded weights (90th percentile over its output channels of the largest tap:
// ONE outlier row -- a near-dead BatchNorm behind the consumer -- must not drag every column with it; that row is x3_range_check's to
// refuse), a_c = |shift_c| + |row c of the producer|_2 = size of the activation for unit-size inputs, l_c = round(log2(a_c / m_c) / 2) says
// how far apart the two are; an ordinary checkpoint (Kaiming-size weights, unit-size activations) sits at l = 2, the operating point every
// golden and parity workload of the build was measured at.  Channels with |l_c - 2| >= 4 are brought back to it (q_c = 2^(l_c - 2));
// everything else -- every channel of the build's own workloads -- is left untouched, bit for bit.  Tensors with several consumers
// (module outputs, residual streams) are not rebalanced: there the two range guards apply.  Host only (no HIP call).
// Rounds 5's host mirror did this in Python (HRNetHeatmap._equalize_blocks); it lives here so that every caller of the C ABI gets it.
static int equalize_blocks(sncal_hrnet& net) {
    net.equalized = 0;
    net.equalize_done = true;
#if SNCAL_X3_F16
    if (!net.x3 || !net.equalize) return 0;
    const int MIN_LOG2 = 4, CENTRE = 2;
    auto split_name = [](const std::string& n, std::string& stem, std::string& leaf) {
        const size_t p = n.rfind('.');
        if (p == std::string::npos) { stem.clear(); leaf = n; } else { stem = n.substr(0, p); leaf = n.substr(p + 1); }
    };
    for (int i = 0; i + 1 < net.n_public; ++i) {
        ConvLayer& P = net.layers[i];
        ConvLayer& C = net.layers[i + 1];
        std::string stem, leaf, nstem, nleaf;
        split_name(P.name, stem, leaf);
        split_name(C.name, nstem, nleaf);
        const bool pair = (leaf == "conv1" && nleaf == "conv2") || (leaf == "conv2" && nleaf == "conv3");
        if (P.bn.empty() || stem != nstem || stem == "model" || !pair) continue;
        if (!P.is_set || !C.is_set || P.w.empty() || C.w.empty() || C.cin != P.cout) continue;
        const int nch = P.cout, taps2 = C.k * C.k;
        const size_t per1 = (size_t)P.cin * P.k * P.k;
        std::vector<double> col(C.cout);
        for (int c = 0; c < nch; ++c) {
            for (int co = 0; co < C.cout; ++co) {                 // consumer column c: largest tap of every output channel, folded
                double mx = 0;
                const float* w = &C.w[((size_t)co * C.cin + c) * taps2];
                for (int t = 0; t < taps2; ++t) mx = std::max(mx, std::fabs((double)w[t]));
                col[co] = mx * std::fabs((double)C.scale[co]);
            }
            std::sort(col.begin(), col.end());
            const double pos = 0.9 * (C.cout - 1);                // torch.quantile's linear interpolation
            const int lo = (int)std::floor(pos), hi = std::min(lo + 1, C.cout - 1);
            const double m = col[lo] + (col[hi] - col[lo]) * (pos - lo);
            double ss = 0;                                        // producer row c: size of its output
            const double sc = (double)P.scale[c];
            for (size_t j = 0; j < per1; ++j) { const double v = (double)P.w[(size_t)c * per1 + j] * sc; ss += v * v; }
            const double a = std::fabs((double)P.shift[c]) + std::sqrt(ss);
            if (!(m > 0) || !(a > 0) || !std::isfinite(m) || !std::isfinite(a)) continue;
            double lg = std::nearbyint(0.5 * std::log2(a / m)) - CENTRE;      // distance from the balance of an ordinary checkpoint
            if (std::fabs(lg) < MIN_LOG2) continue;
            lg = std::max(-60.0, std::min(60.0, lg));
            const float q = (float)std::exp2(lg), iq = (float)std::exp2(-lg);
            P.scale[c] *= iq;
            P.shift[c] *= iq;
            for (int co = 0; co < C.cout; ++co) {
                float* w = &C.w[((size_t)co * C.cin + c) * taps2];
                for (int t = 0; t < taps2; ++t) w[t] *= q;
            }
            ++net.equalized;
        }
    }
#endif
    return net.equalized;
}

extern "C" int sncal_hrnet_set_equalize(sncal_hrnet* net, int enable) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_set_equalize: null");
    net->equalize = enable != 0;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_equalize(sncal_hrnet* net, int* moved) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_equalize: null");
    for (int i = 0; i < net->n_public; ++i)
        if (!net->layers[i].is_set || net->layers[i].w.empty()) { set_error("sncal_hrnet_equalize: conv %s has no weights (call it between sncal_hrnet_set_conv and sncal_hrnet_finalize)", net->layers[i].name.c_str()); return SNCAL_ERR_STATE; }
    if (!net->equalize_done) equalize_blocks(*net);
    if (moved) *moved = net->equalized;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_get_conv(const sncal_hrnet* net, int idx, float* h_weight, float* h_scale, float* h_shift) {
    SNCAL_CHECK_ARG(net && idx >= 0 && idx < net->n_public, "sncal_hrnet_get_conv: index %d", idx);
    const ConvLayer& L = net->layers[idx];
    if (!L.is_set || L.w.empty()) { set_error("sncal_hrnet_get_conv: conv %s holds no host weights (they are released by sncal_hrnet_finalize)", L.name.c_str()); return SNCAL_ERR_STATE; }
    if (h_weight) std::copy(L.w.begin(), L.w.end(), h_weight);
    if (h_scale) std::copy(L.scale.begin(), L.scale.end(), h_scale);
    if (h_shift) std::copy(L.shift.begin(), L.shift.end(), h_shift);
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_finalize(sncal_hrnet* net) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_finalize: null");
    for (int i = 0; i < net->n_public; ++i)
        if (!net->layers[i].is_set) { set_error("conv %s has no weights", net->layers[i].name.c_str()); return SNCAL_ERR_STATE; }
    if (!net->equalize_done) equalize_blocks(*net);   // fp16x3: before the head slices are derived and the range check reads the folded weights
    // physical Cin of every conv = channel count of its input tensor
    for (const Op& op : net->ops)
        if (op.type == OP_CONV) net->layers[op.conv].cin_phys = net->tensors[op.in].C;
    {   // internal layers of the fused head are slices of last_layer.0 (BN scale folded, no shift)
        const ConvLayer& H0 = net->layers[net->l_head0];
        if (!H0.is_set) { set_error("conv %s has no weights", H0.name.c_str()); return SNCAL_ERR_STATE; }
        for (ConvLayer& L : net->layers) {
            if (!L.derived) continue;
            L.w.assign((size_t)L.cout * L.cin, 0.f);
            L.scale.assign(L.cout, 1.f); L.shift.assign(L.cout, 0.f);
            for (int co = 0; co < H0.cout; ++co) {
                L.scale[co] = H0.scale[co];
                if (L.derived_shift) L.shift[co] = H0.shift[co];
                for (int ci = 0; ci < L.cin; ++ci) L.w[(size_t)co * L.cin + ci] = H0.w[(size_t)co * H0.cin + L.col_off + ci];
            }
            L.is_set = true;
        }
        const int rc = pack_head(*net);
        if (rc) return rc;
    }
    {   const int rc = x3_range_check(*net);            // split-fp16 engine: the folded weights must live in fp16's range (SNCAL_ERR_RANGE)
        if (rc) return rc;
    }
    for (ConvLayer& L : net->layers) {
        if (!L.is_set) { set_error("conv %s has no weights", L.name.c_str()); return SNCAL_ERR_STATE; }
        choose_packing(*net, L);
        if (L.mi == 0) { set_error("no kernel variant for conv %s (k=%d s=%d)", L.name.c_str(), L.k, L.stride); return SNCAL_ERR_STATE; }
        int rc = pack_layer(*net, L);
        if (rc) return rc;
        rc = pack_layer_tt(*net, L);
        if (rc) return rc;
        rc = pack_layer_fp8(*net, L);
        if (rc) return rc;
        rc = pack_layer_x3(*net, L);
        if (rc) return rc;
        rc = pack_layer_bbx3(*net, L);
        if (rc) return rc;
        rc = pack_layer_bnp(*net, L);
        if (rc) return rc;
        std::vector<float>().swap(L.w);
    }
    if (net->x3 && !net->d_range) {
        SNCAL_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&net->d_range), 2 * sizeof(unsigned)));
        SNCAL_CHECK_HIP(hipMemset(net->d_range, 0, 2 * sizeof(unsigned)));
    }
    net->finalized = true;
    return SNCAL_OK;
}

// The run-time half of the fp16 range guard (the load-time half is x3_range_check): how many wavefronts have split an activation beyond
// +-65504 (x3.hpp: the value was clamped, the results of those forwards are NOT the reference's) and how many workgroups met a NaN /
// infinite input value, since the counters were last cleared.  Synchronises `stream`.  Engines without the clamp report zeros.
extern "C" int sncal_hrnet_range_status(sncal_hrnet* net, unsigned* overflow, unsigned* nonfinite, int clear, void* stream_) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_range_status: null net");
    unsigned h[2] = {0u, 0u};
    if (net->d_range) {
        hipStream_t stream = as_stream(stream_);
        SNCAL_CHECK_HIP(hipMemcpyAsync(h, net->d_range, sizeof(h), hipMemcpyDeviceToHost, stream));
        if (clear) SNCAL_CHECK_HIP(hipMemsetAsync(net->d_range, 0, sizeof(h), stream));
        SNCAL_CHECK_HIP(hipStreamSynchronize(stream));
    }
    if (overflow) *overflow = h[0];
    if (nonfinite) *nonfinite = h[1];
    if (h[0] || h[1]) {
        set_error("fp16x3 engine: %u wavefront(s) split an activation beyond the fp16 range (clamped to +-65504) and %u workgroup(s) met a NaN / infinite "
                  "input value since the last check: these forwards are not the reference's fp32 result -- use dtype fp32 for this checkpoint / input", h[0], h[1]);
        return SNCAL_ERR_RANGE;
    }
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_output_size(const sncal_hrnet* net, int H, int W, int* out_h, int* out_w) {
    SNCAL_CHECK_ARG(net && H >= 32 && W >= 32, "sncal_hrnet_output_size: bad arguments");
    auto half = [](int v) { return (v + 2 - 3) / 2 + 1; };
    const int h4 = half(half(H)), w4 = half(half(W));
    if (out_h) *out_h = h4 * net->desc.upscale;
    if (out_w) *out_w = w4 * net->desc.upscale;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_workspace(const sncal_hrnet* cnet, int B, int H, int W, size_t* bytes) {
    SNCAL_CHECK_ARG(cnet && bytes && B >= 0 && H >= 32 && W >= 32, "sncal_hrnet_workspace: bad arguments");
    sncal_hrnet* net = const_cast<sncal_hrnet*>(cnet);
    const int sb = std::max(1, std::min(B, net->subbatch));
    const int rc = layout(*net, sb, H, W);
    if (rc) return rc;
    *bytes = net->lay_bytes;
    return SNCAL_OK;
}

namespace {
hipEvent_t next_event(sncal_hrnet& net) {
    if (net.events_used == net.event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        net.event_pool.push_back(e);
    }
    return net.event_pool[net.events_used++];
}
}  // namespace

static int forward_impl(sncal_hrnet* net, const float* d_x, const unsigned char* d_x8, int B, int H, int W, float* d_heat,
                        float* d_kpts, int img_h, int img_w, void* d_ws, size_t ws_bytes, void* stream_);

extern "C" int sncal_hrnet_set_fp8_layers(sncal_hrnet* net, const char* spec) {
    SNCAL_CHECK_ARG(net && spec, "sncal_hrnet_set_fp8_layers: null");
    SNCAL_CHECK_ARG(net->fp8, "sncal_hrnet_set_fp8_layers: the network was not created with SNCAL_FP8");
    unsigned stages = 0;
    std::vector<int> widths;
    bool none = false;
    std::string tok;
    const std::string sp = std::string(spec) + ",";
    for (char ch : sp) {
        if (ch != ',') { if (ch != ' ') tok += ch; continue; }
        if (tok.empty()) continue;
        if (tok == "all") { stages = 0; widths.clear(); }
        else if (tok == "none") none = true;
        else if (tok.size() == 6 && tok.compare(0, 5, "stage") == 0 && tok[5] >= '2' && tok[5] <= '4') stages |= 1u << (tok[5] - '0');
        else if (tok.size() >= 2 && tok[0] == 'c' && atoi(tok.c_str() + 1) > 0) widths.push_back(atoi(tok.c_str() + 1));
        else { set_error("sncal_hrnet_set_fp8_layers: token '%s' (use all, none, stage2..stage4, c<width>)", tok.c_str()); return SNCAL_ERR_ARG; }
        tok.clear();
    }
    net->fp8_stages = none ? (1u << 31) : stages;       // bit 31 matches no stage: nothing selected
    net->fp8_widths = widths;
    net->lay_sb = -1;                                   // twins / lifetimes depend on the selection
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_calibrate_fp8(sncal_hrnet* net, const float* d_x, int B, int H, int W, void* d_ws, size_t ws_bytes, void* stream_) {
    SNCAL_CHECK_ARG(net && d_x && d_ws, "sncal_hrnet_calibrate_fp8: null");
    SNCAL_CHECK_ARG(net->fp8, "sncal_hrnet_calibrate_fp8: the network was not created with SNCAL_FP8");
    hipStream_t stream = as_stream(stream_);
    const size_t nt = net->tensors.size();
    if (!net->d_amax) SNCAL_CHECK_HIP(hipMalloc((void**)&net->d_amax, nt * 4));
    SNCAL_CHECK_HIP(hipMemsetAsync(net->d_amax, 0, nt * 4, stream));
    // keypoints into the (unused) head of the workspace would alias activations: decode into a scratch buffer of our own
    float* d_kp = nullptr;
    SNCAL_CHECK_HIP(hipMalloc((void**)&d_kp, (size_t)B * (net->desc.num_classes - 1) * 3 * 4));
    net->calibrating = true; net->lay_sb = -1;          // set only around the forward: every exit path below sees it cleared
    const int rc = forward_impl(net, d_x, nullptr, B, H, W, nullptr, d_kp, H, W, d_ws, ws_bytes, stream_);
    net->calibrating = false; net->lay_sb = -1;
    if (rc) { (void)hipFree(d_kp); return rc; }
    std::vector<float> amax(nt);
    SNCAL_CHECK_HIP(hipStreamSynchronize(stream));
    SNCAL_CHECK_HIP(hipMemcpy(amax.data(), net->d_amax, nt * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_kp);
    for (size_t t = 0; t < nt; ++t) net->tensors[t].scale = amax[t] > 0.f ? amax[t] / 448.f : 1.f;
    // per-layer output scales: (scale of the layer's input tensor) x (weight scale of the channel)
    for (const Op& op : net->ops) {
        if (op.type != OP_CONV) continue;
        ConvLayer& L = net->layers[op.conv];
        if (!L.d_w8 || net->tensors[op.in].twin < 0) continue;
        std::vector<float> os(L.cout);
        for (int co = 0; co < L.cout; ++co) os[co] = net->tensors[op.in].scale * L.wscale[co];
        SNCAL_CHECK_HIP(hipMemcpy(L.d_oscale, os.data(), os.size() * 4, hipMemcpyHostToDevice));
    }
    net->fp8_calibrated = true;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_set_profiling(sncal_hrnet* net, int enable) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_set_profiling: null");
    SNCAL_CHECK_ARG(enable >= 0 && enable <= 2, "sncal_hrnet_set_profiling: mode %d", enable);
    if (enable == 2) {                  // focus = the kernel variant with the largest total in the profile recorded so far
        std::map<std::string, double> tot;
        for (const auto& iv : net->intervals) {
            SNCAL_CHECK_HIP(hipEventSynchronize(iv.e1));
            float ms = 0;
            SNCAL_CHECK_HIP(hipEventElapsedTime(&ms, iv.e0, iv.e1));
            tot[iv.kernel] += ms;
        }
        SNCAL_CHECK_ARG(!tot.empty(), "sncal_hrnet_set_profiling: mode 2 needs a mode-1 profile of at least one forward first");
        net->focus.clear();
        double best = -1;
        for (const auto& kv : tot) if (kv.second > best) { best = kv.second; net->focus = kv.first; }
    }
    net->profiling = enable;
    net->intervals.clear();
    net->events_used = 0;
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_get_profile(sncal_hrnet* net, sncal_kernel_stat* out, int cap, int* count) {
    SNCAL_CHECK_ARG(net && count, "sncal_hrnet_get_profile: null");
    std::map<std::string, sncal_kernel_stat> agg;
    for (const auto& iv : net->intervals) {
        SNCAL_CHECK_HIP(hipEventSynchronize(iv.e1));
        float ms = 0;
        SNCAL_CHECK_HIP(hipEventElapsedTime(&ms, iv.e0, iv.e1));
        sncal_kernel_stat& st = agg[iv.kernel];
        if (st.launches == 0) { memset(&st, 0, sizeof(st)); snprintf(st.kernel, sizeof(st.kernel), "%s", iv.kernel.c_str()); }
        st.flops += iv.flops; st.bytes += iv.bytes; st.ms += ms; st.launches += 1;
    }
    *count = (int)agg.size();
    int i = 0;
    for (const auto& kv : agg) { if (out && i < cap) out[i] = kv.second; ++i; }
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_plan_num_ops(const sncal_hrnet* net) { return net ? (int)net->ops.size() : 0; }
extern "C" int sncal_hrnet_plan_num_tensors(const sncal_hrnet* net) { return net ? (int)net->tensors.size() : 0; }

extern "C" int sncal_hrnet_plan_op(const sncal_hrnet* net, int idx, sncal_plan_op* out) {
    SNCAL_CHECK_ARG(net && out && idx >= 0 && idx < (int)net->ops.size(), "sncal_hrnet_plan_op: index %d", idx);
    const Op& op = net->ops[idx];
    memset(out, 0, sizeof(*out));
    out->type = (int)op.type; out->active = op_active(*net, op) ? 1 : 0; out->conv = op.conv;
    out->in = op.in; out->res = op.res; out->out = op.out; out->base = op.base; out->nsrc = op.nsrc;
    for (int i = 0; i < 4; ++i) out->src[i] = op.srcs[i];
    out->head_direct = op.head_direct; out->head_nsrc = op.head_nsrc; out->head_nfold = op.head_nfold;
    for (int i = 0; i < 5; ++i) out->head_src[i] = i < HEAD_MAX_SRC ? op.head_src[i] : -1;
    for (int i = 0; i < 2; ++i) out->head_fold[i] = i < HEAD_MAX_FOLD ? op.head_fold[i] : -1;
    out->relu = op.relu ? 1 : 0; out->out_coff = op.out_coff; out->out_f32 = op.out_f32 ? 1 : 0;
    if (op.conv >= 0) {
        const ConvLayer& L = net->layers[op.conv];
        snprintf(out->name, sizeof(out->name), "%s", L.name.c_str());
        out->cin = L.cin; out->cout = L.cout; out->ksize = L.k; out->stride = L.stride; out->col_off = L.col_off;
        out->fp8 = L.fp8_on ? 1 : L.x3_on ? 2 : (net->x3_generic && op.type == OP_CONV) ? 3 : 0;
    }
    out->res_twin = op.res_twin ? 1 : 0;
    if (idx < (int)net->op_label.size()) snprintf(out->kernel, sizeof(out->kernel), "%s", net->op_label[idx].c_str());
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_plan_tensor(const sncal_hrnet* net, int id, sncal_plan_tensor* out) {
    SNCAL_CHECK_ARG(net && out && id >= 0 && id < (int)net->tensors.size(), "sncal_hrnet_plan_tensor: id %d", id);
    SNCAL_CHECK_ARG(net->lay_sb > 0, "sncal_hrnet_plan_tensor: no layout yet (call sncal_hrnet_workspace or a forward first)");
    const Tensor& t = net->tensors[id];
    memset(out, 0, sizeof(*out));
    out->C = t.C; out->H = t.H; out->W = t.W; out->dtype = t.f32 ? 0 : t.fp8 ? 2 : (net->dtype == SNCAL_BF16 ? 1 : 0);
    out->twin = t.twin; out->alive = t.first >= 0 ? 1 : 0; out->scale = t.scale;
    if (t.fp8)                      // the calibrated scale is kept on the bf16 tensor the twin belongs to
        for (const Tensor& o : net->tensors) if (o.twin == id) out->scale = o.scale;
    out->sub_batch = net->lay_sb;
    out->bytes = (size_t)net->lay_sb * t.H * t.W * t.C * (t.f32 ? 4 : t.fp8 ? 1 : net->esize);
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_plan_tap(sncal_hrnet* net, int op_idx, int tensor_id, void* d_dst) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_plan_tap: null");
    if (op_idx < 0) { net->taps.clear(); return SNCAL_OK; }
    SNCAL_CHECK_ARG(op_idx < (int)net->ops.size() && tensor_id >= 0 && tensor_id < (int)net->tensors.size() && d_dst,
                    "sncal_hrnet_plan_tap: op %d tensor %d", op_idx, tensor_id);
    net->taps.push_back({op_idx, tensor_id, d_dst});
    return SNCAL_OK;
}

// Every forward starts from zeroed ticket words on ITS stream (256 bytes): the kernels re-arm the words themselves, but a launch that
// failed or was torn down half way (device reset by another client, a killed process sharing nothing but the driver) must not leave the
// next forward a counter that skips or repeats work.  A network handle is single-stream: forwards of ONE handle issued on two streams
// at once would share these words (include/sncal.h says so); use one handle per stream (the weights are small against 288 GB).
static int rearm_tickets(sncal_hrnet* net, hipStream_t stream) {
    const int rc = ensure_tickets(net, stream);
    if (rc) return rc;
    SNCAL_CHECK_HIP(hipMemsetAsync(net->d_tickets, 0, TICKET_WORDS * sizeof(unsigned), stream));
    return SNCAL_OK;
}

extern "C" int sncal_hrnet_forward(sncal_hrnet* net, const float* d_x, int B, int H, int W, float* d_heat, float* d_kpts,
                                   int img_h, int img_w, void* d_ws, size_t ws_bytes, void* stream_) {
    return forward_impl(net, d_x, nullptr, B, H, W, d_heat, d_kpts, img_h, img_w, d_ws, ws_bytes, stream_);
}

extern "C" int sncal_hrnet_forward_u8(sncal_hrnet* net, const unsigned char* d_x, int B, int H, int W, float* d_heat,
                                      float* d_kpts, int img_h, int img_w, void* d_ws, size_t ws_bytes, void* stream_) {
    return forward_impl(net, nullptr, d_x, B, H, W, d_heat, d_kpts, img_h, img_w, d_ws, ws_bytes, stream_);
}

static int forward_impl(sncal_hrnet* net, const float* d_x, const unsigned char* d_x8, int B, int H, int W, float* d_heat,
                        float* d_kpts, int img_h, int img_w, void* d_ws, size_t ws_bytes, void* stream_) {
    SNCAL_CHECK_ARG(net, "sncal_hrnet_forward: null net");
    if (!net->finalized) { set_error("sncal_hrnet_forward: weights not finalized"); return SNCAL_ERR_STATE; }
    if (net->fp8 && !net->fp8_calibrated && !net->calibrating) { set_error("sncal_hrnet_forward: fp8 network without calibration (sncal_hrnet_calibrate_fp8)"); return SNCAL_ERR_STATE; }
    SNCAL_CHECK_ARG(B >= 0 && H >= 32 && W >= 32, "sncal_hrnet_forward: bad shape B=%d H=%d W=%d", B, H, W);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG((d_x || d_x8) && d_ws, "sncal_hrnet_forward: null input / workspace");
    SNCAL_CHECK_ARG(d_heat || d_kpts, "sncal_hrnet_forward: need d_heat or d_kpts");
    SNCAL_CHECK_ARG(!(d_kpts && net->desc.head_softmax), "sncal_hrnet_forward: keypoint decode needs a log-softmax head");
    hipStream_t stream = as_stream(stream_);
    const int SB = std::max(1, std::min(B, net->subbatch));
    int rc = layout(*net, SB, H, W);
    if (rc) return rc;
    if (ws_bytes < net->lay_bytes) { set_error("workspace too small: %zu < %zu", ws_bytes, net->lay_bytes); return SNCAL_ERR_WORKSPACE; }
    char* ws = reinterpret_cast<char*>(d_ws);
    const Tensor& th = net->tensors[net->t_heat];
    const int C = net->desc.num_classes;
    rc = rearm_tickets(net, stream);
    if (rc) return rc;
    for (int b0 = 0; b0 < B; b0 += SB) {
        const int sb = std::min(SB, B - b0);
        float* heat = d_heat ? d_heat + (size_t)b0 * C * th.H * th.W : reinterpret_cast<float*>(ws + th.offset);
        bool skip_next = false, decoded = false;
        bool head_decoded = false;       // the head kernel produced the decode's partial maxima instead of logits
        int skip_group = 0;                                  // remaining members of a launch group that already ran
        for (size_t oi = 0; oi < net->ops.size(); ++oi) {
            const Op& op = net->ops[oi];
            if (!op_active(*net, op)) continue;
            auto run_taps = [&]() -> int {                       // test instrumentation: first sub-batch only, stream-ordered copies
                if (net->taps.empty() || b0 != 0) return SNCAL_OK;
                for (const sncal_hrnet::Tap& tp : net->taps) {
                    if (tp.op != (int)oi) continue;
                    const Tensor& tt = net->tensors[tp.tensor];
                    if (tt.first < 0) { set_error("sncal_hrnet_plan_tap: tensor %d is not allocated at this layout", tp.tensor); return SNCAL_ERR_STATE; }
                    const void* src = tt.external_heat ? (const void*)heat : (const void*)(ws + tt.offset);
                    const size_t nb = (size_t)sb * tt.H * tt.W * tt.C * (tt.f32 ? 4 : tt.fp8 ? 1 : net->esize);
                    SNCAL_CHECK_HIP(hipMemcpyAsync(tp.dst, src, nb, hipMemcpyDeviceToDevice, stream));
                }
                return SNCAL_OK;
            };
            if (skip_next) { skip_next = false; rc = run_taps(); if (rc) return rc; continue; }      // second conv of a fused BasicBlock
            if (skip_group > 0) { --skip_group; rc = run_taps(); if (rc) return rc; continue; }
            net->last_kernel.clear(); net->last_flops = 0; net->last_bytes = 0;
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (net->op_label.size() != net->ops.size()) net->op_label.assign(net->ops.size(), std::string());
            if (net->profiling == 1 || (net->profiling == 2 && net->op_label[oi] == net->focus)) {
                ev0 = next_event(*net); ev1 = next_event(*net); sncal::launch_events() = sncal::LaunchEvents{ev0, ev1};
            }
            switch (op.type) {
                case OP_INPUT:
                    if (d_x8) rc = launch_u8hwc_to_nhwc(net->dtype, d_x8 + (size_t)b0 * 3 * H * W, ws + net->tensors[op.out].offset, sb, H, W, stream);
                    else rc = launch_nchw_to_nhwc(net->dtype, d_x + (size_t)b0 * 3 * H * W, ws + net->tensors[op.out].offset, sb, 3, H, W, stream, net->d_range ? net->d_range + 1 : nullptr);
                    break;
                case OP_CONV: {
                    if (op.launch_group >= 0) {              // same-depth convs of the parallel branches: one grouped launch
                        int n = 1;
                        while (oi + n < net->ops.size() && net->ops[oi + n].launch_group == op.launch_group && net->ops[oi + n].type == OP_CONV &&
                               op_active(*net, net->ops[oi + n])) ++n;
                        bool done = false;
                        rc = run_conv_group(*net, &net->ops[oi], n, sb, ws, stream, &done);
                        if (rc) return rc;
                        if (done) { skip_group = n - 1; break; }
                    }
                    // BasicBlock of the 48-channel branch: conv1 + conv2 (+ residual) fused when the next active op
                    // is its second convolution and both layers carry the (MI = 3, G = 3) packing
                    const Op* op2 = nullptr;
                    if (net->fuse_bblock && net->dtype == SNCAL_BF16 && op.relu && op.res < 0 && !op.out_f32 && oi + 1 < net->ops.size()) {
                        const Op& nx = net->ops[oi + 1];
                        if (nx.type == OP_CONV && op_active(*net, nx) && nx.in == op.out && nx.res == op.in && nx.relu && !nx.out_f32 &&
                            nx.out_coff == 0 && op.out_coff == 0) {
                            const ConvLayer& A = net->layers[op.conv]; const ConvLayer& Bl = net->layers[nx.conv];
                            auto ok = [&](const ConvLayer& L) { return L.k == 3 && L.stride == 1 && L.cin == 48 && L.cout == 48 && L.cin_phys == 48 &&
                                                                       L.mi == 3 && L.g == 3 && L.chunks == 2 && L.nblk == 1; };
                            if (ok(A) && ok(Bl) && net->tensors[op.in].C == 48 && net->tensors[nx.out].C == 48) op2 = &nx;
                        }
                    }
                    // ... and in the bf16x3 engine: conv1 -> mid tile in LDS as hi / lo planes -> conv2 + residual (bblockx3.hip); the mid tensor and
                    // its twin are not written at all
                    const Op* opx = nullptr;
                    if (net->fuse_bbx3 && net->x3 && op.relu && op.res < 0 && !op.out_f32 && oi + 1 < net->ops.size() && net->layers[op.conv].x3_on &&
                        net->layers[op.conv].d_w_bbx && tt_eligible(*net, op, sb)) {
                        const Op& nx = net->ops[oi + 1];
                        if (nx.type == OP_CONV && op_active(*net, nx) && nx.in == op.out && nx.res == op.in && nx.relu && !nx.out_f32 && op.out_coff == 0 &&
                            net->layers[nx.conv].x3_on && net->layers[nx.conv].d_w_bbx && tt_eligible(*net, nx, sb) && net->tensors[op.in].C == 48 &&
                            net->tensors[op.in].twin >= 0 && net->tensors[net->tensors[op.in].twin].first >= 0) opx = &nx;
                    }
                    // split engines, layer1: conv3 (+ residual, ReLU) of a Bottleneck and conv1 (+ ReLU) of the next one in one pass over the
                    // pixels (bneckx3.hip): the 256-channel tensor between them is written once (the next residual) and not read back
                    const Op* opn = nullptr;
                    const Op* opd = nullptr;        // ... and block 0's tail: this op is the downsample branch, the next one the conv3 that adds it
                    if ((net->fuse_bneck & 2) && net->x3 && !op.relu && op.res < 0 && !op.out_f32 && op.out_coff == 0 && oi + 1 < net->ops.size() &&
                        net->layers[op.conv].d_w_bnp && net->layers[op.conv].cout == BNP_WIDE && !net->layers[op.conv].x3_on && op.launch_group < 0) {
                        const Op& nx = net->ops[oi + 1];
                        if (nx.type == OP_CONV && op_active(*net, nx) && nx.res == op.out && nx.relu && !nx.out_f32 && nx.out_coff == 0 && nx.launch_group < 0 &&
                            net->layers[nx.conv].d_w_bnp && net->layers[nx.conv].cout == BNP_WIDE && !net->layers[nx.conv].x3_on &&
                            net->tensors[op.in].C == BNP_MID && net->tensors[nx.in].C == BNP_MID && net->tensors[nx.out].C == BNP_WIDE &&
                            net->tensors[op.in].H == net->tensors[nx.in].H && net->tensors[op.in].W == net->tensors[nx.in].W &&
                            net->tensors[op.out].last_read == (int)oi + 1) {      // nobody else reads the downsample branch
                            bool skip_a = false;
                            if (!producer_twin(*net, nx.out, sb, ws, &skip_a)) opd = &nx;
                        }
                    }
                    if ((net->fuse_bneck & 1) && net->x3 && op.relu && op.res >= 0 && !op.out_f32 && op.out_coff == 0 && oi + 1 < net->ops.size() &&
                        net->layers[op.conv].d_w_bnp && net->layers[op.conv].cout == BNP_WIDE && !net->layers[op.conv].x3_on) {
                        const Op& nx = net->ops[oi + 1];
                        const Tensor& t_in = net->tensors[op.in];
                        const Tensor& t_res = net->tensors[op.res];
                        const Tensor& t_y = net->tensors[op.out];
                        if (nx.type == OP_CONV && op_active(*net, nx) && nx.in == op.out && nx.res < 0 && nx.relu && !nx.out_f32 && nx.out_coff == 0 &&
                            nx.launch_group < 0 && net->layers[nx.conv].d_w_bnp && net->layers[nx.conv].cout == BNP_MID && !net->layers[nx.conv].x3_on &&
                            t_in.C == BNP_MID && t_res.C == BNP_WIDE && t_y.C == BNP_WIDE && net->tensors[nx.out].C == BNP_MID &&
                            t_res.H == t_y.H && t_res.W == t_y.W) {
                            bool skip_a = false, skip_b = false;         // neither output may owe somebody a split twin (they feed generic kernels)
                            if (!producer_twin(*net, op.out, sb, ws, &skip_a) && !producer_twin(*net, nx.out, sb, ws, &skip_b)) opn = &nx;
                        }
                    }
                    if (opd) {
                        const Tensor& t_y = net->tensors[opd->out];
                        BneckPairParams bp;
                        memset(&bp, 0, sizeof(bp));
                        bp.range = net->d_range;
                        bp.h2 = reinterpret_cast<const float*>(ws + net->tensors[opd->in].offset);
                        bp.x0 = reinterpret_cast<const float*>(ws + net->tensors[op.in].offset);
                        bp.y = reinterpret_cast<float*>(ws + t_y.offset);
                        bp.w3 = net->layers[opd->conv].d_w_bnp; bp.b3 = net->layers[opd->conv].d_bias;
                        bp.wds = net->layers[op.conv].d_w_bnp; bp.bds = net->layers[op.conv].d_bias;
                        bp.P = (long long)sb * t_y.H * t_y.W;
                        if (!net->n_cus) {
                            int dev = 0, cus = 0;
                            SNCAL_CHECK_HIP(hipGetDevice(&dev));
                            SNCAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                            net->n_cus = cus > 0 ? cus : 256;
                        }
                        rc = ensure_tickets(net, stream);
                        if (rc) return rc;
                        bp.ticket = net->d_tickets;
                        rc = launch_bneck_pair_x3(bp, net->n_cus, stream);
                        if (net->profiling) {
                            net->last_kernel = "bneck_tail_ds_x3";
                            net->last_flops = 2.0 * 2.0 * (double)bp.P * BNP_MID * BNP_WIDE;
                            net->last_bytes = (double)bp.P * 4.0 * (BNP_MID + BNP_MID + BNP_WIDE) + 2.0 * BNP_W_BYTES;
                        }
                        skip_next = true;
                    } else if (opn) {
                        const Tensor& t_y = net->tensors[op.out];
                        BneckPairParams bp;
                        memset(&bp, 0, sizeof(bp));
                        bp.range = net->d_range;
                        bp.h2 = reinterpret_cast<const float*>(ws + net->tensors[op.in].offset);
                        bp.res = reinterpret_cast<const float*>(ws + net->tensors[op.res].offset);
                        bp.y = reinterpret_cast<float*>(ws + t_y.offset);
                        bp.h1 = reinterpret_cast<float*>(ws + net->tensors[opn->out].offset);
                        bp.w3 = net->layers[op.conv].d_w_bnp; bp.b3 = net->layers[op.conv].d_bias;
                        bp.w1 = net->layers[opn->conv].d_w_bnp; bp.b1 = net->layers[opn->conv].d_bias;
                        bp.P = (long long)sb * t_y.H * t_y.W;
                        if (!net->n_cus) {
                            int dev = 0, cus = 0;
                            SNCAL_CHECK_HIP(hipGetDevice(&dev));
                            SNCAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                            net->n_cus = cus > 0 ? cus : 256;
                        }
                        rc = ensure_tickets(net, stream);
                        if (rc) return rc;
                        bp.ticket = net->d_tickets;
                        rc = launch_bneck_pair_x3(bp, net->n_cus, stream);
                        if (net->profiling) {
                            net->last_kernel = "bneck_seam_x3";
                            net->last_flops = 2.0 * 2.0 * (double)bp.P * BNP_MID * BNP_WIDE;
                            net->last_bytes = (double)bp.P * 4.0 * (BNP_MID + BNP_WIDE + BNP_WIDE + BNP_MID) + 2.0 * BNP_W_BYTES;
                        }
                        skip_next = true;
                    } else if (opx) {
                        const Tensor& ti = net->tensors[op.in];
                        const Tensor& to = net->tensors[opx->out];
                        const sncal::LaunchEvents armed = sncal::launch_events();
                        sncal::launch_events() = sncal::LaunchEvents{};
                        if (!twin_written_by_producer(*net, op.in, sb)) {         // the fp32 input's split twin, unless its producer wrote it
                            rc = launch_split_f32(ws + ti.offset, ws + net->tensors[ti.twin].offset, (size_t)sb * ti.H * ti.W * ti.C, stream, net->d_range);
                            if (rc) return rc;
                        }
                        sncal::launch_events() = armed;
                        BBlockX3Params bp;
                        memset(&bp, 0, sizeof(bp));
                        bp.range = net->d_range;
                        bp.x = ws + net->tensors[ti.twin].offset;
                        const bool twin_out = to.twin >= 0 && net->tensors[to.twin].first >= 0 && twin_written_by_producer(*net, opx->out, sb);
                        bp.out_twin = twin_out ? ws + net->tensors[to.twin].offset : nullptr;
                        bp.out = (!twin_out || net->need_bf16[opx->out]) ? reinterpret_cast<float*>(ws + to.offset) : nullptr;
                        bp.w1 = net->layers[op.conv].d_w_bbx; bp.b1 = net->layers[op.conv].d_bias;
                        bp.w2 = net->layers[opx->conv].d_w_bbx; bp.b2 = net->layers[opx->conv].d_bias;
                        bp.N = sb; bp.H = ti.H; bp.W = ti.W; bp.out_cstride = to.C; bp.out_coff = opx->out_coff;
                        rc = ensure_tickets(net, stream);
                        if (rc) return rc;
                        bp.ticket = net->d_tickets + 16;
                        rc = launch_bblockx3(bp, stream);
                        if (net->profiling) {
                            net->last_kernel = "bblockx3_fused";
                            const double px = (double)sb * ti.H * ti.W;
                            net->last_flops = 2.0 * 2.0 * px * 48 * 48 * 9;
                            net->last_bytes = px * 48 * 4 * (1.0 + (bp.out_twin ? 1.0 : 0.0) + (bp.out ? 1.0 : 0.0)) + 2.0 * BBX_W_BYTES;
                        }
                        skip_next = true;
                    } else if (op2) {
                        const Tensor& ti = net->tensors[op.in];
                        BBlockParams bp;
                        bp.x = ws + ti.offset; bp.out = ws + net->tensors[op2->out].offset;
                        bp.w1 = net->layers[op.conv].d_w; bp.b1 = net->layers[op.conv].d_bias;
                        bp.w2 = net->layers[op2->conv].d_w; bp.b2 = net->layers[op2->conv].d_bias;
                        bp.N = sb; bp.H = ti.H; bp.W = ti.W; bp.tiles_x = bp.tiles_y = 0; bp.trace = nullptr;
                        rc = ensure_tickets(net, stream);
                        if (rc) return rc;
                        bp.ticket = net->d_tickets + 48;
                        rc = launch_bblock48(bp, stream);
                        if (net->profiling) {
                            net->last_kernel = "bblock48_fused";
                            const double px = (double)sb * ti.H * ti.W;
                            net->last_flops = 2.0 * 2.0 * px * 48 * 48 * 9;
                            net->last_bytes = 2.0 * px * 48 * 2 + 2.0 * 48 * 48 * 9 * 2;
                        }
                        skip_next = true;
                    } else {
                        rc = run_conv(*net, op, sb, ws, stream);
                    }
                    break;
                }
                case OP_UPADD: {
                    const Tensor& to = net->tensors[op.out];
                    UpsampleAddParams p;
                    memset(&p, 0, sizeof(p));
                    p.range = net->d_range;
                    p.base = op.base >= 0 ? ws + net->tensors[op.base].offset : nullptr;
                    p.nsrc = op.nsrc;
                    int C0 = to.C;
                    for (int s = 0; s < op.nsrc; ++s) {
                        const Tensor& ts = net->tensors[op.srcs[s]];
                        p.src[s] = ws + ts.offset; p.Hs[s] = ts.H; p.Ws[s] = ts.W;
                        p.sy[s] = to.H > 1 ? (float)(ts.H - 1) / (float)(to.H - 1) : 0.f;
                        p.sx[s] = to.W > 1 ? (float)(ts.W - 1) / (float)(to.W - 1) : 0.f;
                        C0 = ts.C;
                    }
                    p.out = ws + to.offset; p.N = sb; p.H = to.H; p.W = to.W; p.C = C0;
                    p.out_cstride = to.C; p.out_coff = op.out_coff; p.relu = op.relu ? 1 : 0;
                    if (net->x3) {
                        bool skip_f32 = false;
                        p.out_twin = producer_twin(*net, op.out, sb, ws, &skip_f32);
                        if (p.out_twin && skip_f32) p.out = nullptr;
                    }
                    rc = launch_upsample_add(net->dtype, p, stream);
                    break;
                }
                case OP_SOFTMAX: {
                    const Tensor& tl = net->tensors[op.in];
                    // nobody wants the heatmap (predict() / the pipeline): log-softmax and the keypoint decode run fused and the
                    // (B,C,h,w) tensor is never written; its workspace slot serves as the (much smaller) scratch
                    static const bool fuse_decode = !(getenv("SNCAL_FUSE_DECODE") && atoi(getenv("SNCAL_FUSE_DECODE")) == 0);
                    if (head_decoded) {        // head32.hip already holds log-softmax + the tiles' maxima: the decode's second half only
                        int rp, cp;
                        head32_decode_parts(tl.H, tl.W, &rp, &cp);
                        const float* parts = reinterpret_cast<const float*>(ws + tl.offset);        // the logits tensor's slot holds them
                        rc = launch_kp_finish(parts, rp, parts + (size_t)sb * (C - 1) * tl.H * rp, cp, C, sb, tl.H, tl.W, img_h, img_w,
                                              d_kpts + (size_t)b0 * (C - 1) * 3, stream);
                        decoded = true;
                        if (net->profiling) { net->last_kernel = "kp_finish"; net->last_bytes = (double)head32_decode_scratch(sb, C, tl.H, tl.W); }
                        break;
                    }
                    if (fuse_decode && !d_heat && d_kpts && !net->desc.head_softmax &&
                        logsoftmax_decode_scratch(sb, C, tl.H, tl.W) <= (size_t)sb * C * th.H * th.W * sizeof(float)) {
                        rc = launch_logsoftmax_decode(reinterpret_cast<const float*>(ws + tl.offset), tl.C, C, sb, tl.H, tl.W, img_h, img_w,
                                                      heat, d_kpts + (size_t)b0 * (C - 1) * 3, stream);
                        decoded = true;
                        if (net->profiling) { net->last_kernel = "logsoftmax_decode_fused"; net->last_bytes = (double)sb * tl.H * tl.W * tl.C * 4; }
                        break;
                    }
                    rc = launch_softmax_nchw(reinterpret_cast<const float*>(ws + tl.offset), tl.C, C, (size_t)sb * tl.H * tl.W,
                                             (size_t)tl.H * tl.W, net->desc.head_softmax ? 0 : 1, heat, stream);
                    break;
                }
                case OP_HEAD: {
                    const Tensor& td = net->tensors[op.head_direct];
                    const Tensor& to = net->tensors[op.out];
                    HeadParams hp;
                    memset(&hp, 0, sizeof(hp));
                    hp.range = net->d_range;
                    hp.direct = ws + td.offset; hp.Cd = td.C;
                    hp.w0 = net->d_hw0; hp.bias0 = net->d_hb0; hp.w1 = net->d_hw1; hp.bias1 = net->d_hb1;
                    hp.w0_32 = net->d_hw0_32; hp.w1_32 = net->d_hw1_32; hp.ks16 = net->head_ks16;
                    hp.nsrc = op.head_nsrc;
                    for (int s2 = 0; s2 < op.head_nsrc; ++s2) {
                        const Tensor& ts = net->tensors[op.head_src[s2]];
                        hp.src[s2] = ws + ts.offset; hp.Hs[s2] = ts.H; hp.Ws[s2] = ts.W;
                        hp.sy[s2] = to.H > 1 ? (float)(ts.H - 1) / (float)(to.H - 1) : 0.f;
                        hp.sx[s2] = to.W > 1 ? (float)(ts.W - 1) / (float)(to.W - 1) : 0.f;
                    }
                    hp.nfold = op.head_nfold; hp.ks1 = net->head_ks1;
                    for (int s2 = 0; s2 < op.head_nfold; ++s2) {
                        const Tensor& tf = net->tensors[op.head_fold[s2]];
                        hp.fold[s2] = ws + tf.offset; hp.Cf[s2] = tf.C; hp.Hf[s2] = tf.H; hp.Wf[s2] = tf.W;
                        hp.fsy[s2] = to.H > 1 ? (float)(tf.H - 1) / (float)(to.H - 1) : 0.f;
                        hp.fsx[s2] = to.W > 1 ? (float)(tf.W - 1) / (float)(to.W - 1) : 0.f;
                    }
                    hp.logits = reinterpret_cast<float*>(ws + to.offset);
                    hp.N = sb; hp.H = to.H; hp.W = to.W; hp.HP = net->head_hp; hp.NQ = net->head_hp / 32; hp.LC = to.C;
                    {   // nobody wants the heatmap: log-softmax and the decode's maxima inside the head kernel (head32.hip), neither logits nor
                        // log-probabilities are written; the logits tensor's own workspace slot (alive from here to the softmax op) holds the
                        // partial maxima instead
                        static const bool fuse_dec = !(getenv("SNCAL_FUSE_DECODE") && atoi(getenv("SNCAL_FUSE_DECODE")) == 0) &&
                                                     !(getenv("SNCAL_HEAD_DECODE") && atoi(getenv("SNCAL_HEAD_DECODE")) == 0);
                        head_decoded = false;
                        hp.w0_32_lo = net->d_hw0_32l; hp.w1_32_lo = net->d_hw1_32l;
                        if (fuse_dec && !d_heat && d_kpts && !net->desc.head_softmax && C > 32 && C <= 64 && (net->x3 ? headx3_applies(hp) : head32_applies(hp)) &&
                            head32_decode_scratch(sb, C, to.H, to.W) <= to.bytes && th.H == to.H && th.W == to.W) {
                            int rp, cp;
                            head32_decode_parts(to.H, to.W, &rp, &cp);
                            hp.dec_row = hp.logits; hp.dec_col = hp.logits + (size_t)sb * (C - 1) * to.H * rp; hp.dec_C = C;
                            head_decoded = true;
                        }
                    }
                    if (net->x3) {
                        hp.w0_32_lo = net->d_hw0_32l; hp.w1_32_lo = net->d_hw1_32l;
                        if (!launch_headx3(hp, stream)) { set_error("bf16x3 head: configuration not served by headx3 (set SNCAL_HEADX3=0)"); return SNCAL_ERR_STATE; }
                        SNCAL_CHECK_LAUNCH();
                        rc = SNCAL_OK;
                    } else
                    rc = launch_head_fused(hp, net->head_m2, stream);
                    if (net->profiling) {
                        net->last_kernel = net->x3 ? "headx3_fused" : "head_fused";
                        const double px = (double)sb * to.H * to.W;
                        net->last_flops = 2.0 * px * net->head_hp * (net->head_k + net->head_m2 * 16);
                        net->last_bytes = px * (td.C * 2 + to.C * 4);
                        for (int s2 = 0; s2 < op.head_nsrc; ++s2) { const Tensor& ts = net->tensors[op.head_src[s2]]; net->last_bytes += (double)sb * ts.H * ts.W * ts.C * 2; }
                        for (int s2 = 0; s2 < op.head_nfold; ++s2) { const Tensor& tf = net->tensors[op.head_fold[s2]]; net->last_bytes += (double)sb * tf.H * tf.W * tf.C * 2; }
                    }
                    break;
                }
                case OP_DECODE:
                    if (d_kpts && !decoded) rc = sncal_heatmap_decode(heat, sb, C, th.H, th.W, img_h, img_w, d_kpts + (size_t)b0 * (C - 1) * 3, stream_);
                    break;
            }
            if (rc) return rc;
            rc = run_taps();
            if (rc) return rc;
            if (net->profiling) {
                sncal::LaunchEvents& le = sncal::launch_events();
                const bool launched = !le.start && !le.stop;          // the op's launch consumed the pair
                le = sncal::LaunchEvents{};
                if (!launched || !ev0 || !ev1) continue;
                if (net->last_kernel.empty()) {
                    const char* names[] = {"nchw_to_nhwc", "conv", "upsample_add", "softmax_nchw", "kp_decode", "head_fused"};
                    net->last_kernel = names[op.type];
                    if (op.type == OP_UPADD) {
                        const Tensor& to = net->tensors[op.out];
                        double b = 0;
                        for (int s2 = 0; s2 < op.nsrc; ++s2) { const Tensor& ts = net->tensors[op.srcs[s2]]; b += (double)sb * ts.H * ts.W * ts.C * net->esize; }
                        const int C0 = op.nsrc ? net->tensors[op.srcs[0]].C : to.C;
                        net->last_bytes = b + (double)sb * to.H * to.W * C0 * net->esize * (op.base >= 0 ? 2 : 1);
                    } else if (op.type == OP_SOFTMAX || op.type == OP_DECODE) {
                        net->last_bytes = (double)sb * C * th.H * th.W * 4 * (op.type == OP_SOFTMAX ? 2 : 1);
                    } else if (op.type == OP_INPUT) {
                        net->last_bytes = (double)sb * H * W * (3 * 4 + net->ge * net->esize);
                    }
                }
                if (op.type == OP_DECODE && (!d_kpts || decoded)) continue;
                if (net->profiling == 1) net->op_label[oi] = net->last_kernel;
                net->intervals.push_back({ev0, ev1, net->last_kernel, net->last_flops, net->last_bytes});
            }
        }
    }
    return SNCAL_OK;
}
