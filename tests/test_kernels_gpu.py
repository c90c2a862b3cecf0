"""GPU: EVERY launch of a forward against torch fp32 on the operands that launch really read (per-kernel oracle tests).

The whole-network goldens (tests/test_hrnet_gpu.py) pin the fp32 engine to the reference capture; the bf16 / fp8 engines run
DIFFERENT kernels (two-team conv_tt in bf16 and e4m3, the fused 48-channel BasicBlock, the restructured head32, generic
stride-2 / 1x1 variants, upsample_add) that a drift bound over 300 layers cannot check at tap resolution.  Here the plan's
taps (sncal_hrnet_plan_tap, include/sncal.h) copy the input / residual / output tensors of every op out of the workspace
while a REAL forward runs -- grouped launches, stacked frames, partial tiles, sub-batch of B frames as they are -- and each op
is recomputed with torch (fp32 conv2d / index arithmetic) from the SAME rounded operands:

    conv (+BN shift, +residual, ReLU)       hrnet.py:42-58, 79-99, 195-214, 357-391    all kernels that serve it
    fused BasicBlock (48 channels)          hrnet.py:42-58                              bblock48: conv1 -> bf16 -> conv2 + x
    fuse sums (1x1 + bilinear up, add)      hrnet.py:229-244                            upsample_add
    head (upsample, concat, 1x1, 1x1)       hrnet.py:316-329, 489-510                   head32 / head_fused, restructured
    e4m3 convolutions (C5)                  same layers, per-tensor / per-channel scales conv_tt<fp8>, its bf16 and e4m3 outputs

Tolerance = one ulp of the output type (bf16: 2^-7 |y|, the worst-case ulp of a value; e4m3: 2^-3 |y| + one subnormal step)
plus 1e-3 of fp32 accumulation-order slack -- a wrong tap, a wrong tile-edge pixel, a row of the neighbouring frame or a
mis-scaled channel is two orders of magnitude above it.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BN_EPS = 1e-5


# ---- reference-side weights: the fold of hrnet.py's load_state_dict, restated (fp64 fold, fp32 storage) ---------------------
X3_T = torch.float16          # 16-bit type of the split twins: the build's (sncal_x3_name()); set in run_case
# split-arithmetic slack as a fraction of sum |x w|: fp16 splits drop ~2^-22 of a product (measured worst case 1.1e-6 of sum |x w|,
# the fp32 accumulation itself), bf16 splits ~2^-16 (measured below 1e-5)
X3_SLACK = 2e-6


def folded(sd, name, bn, has_bias):
    w = sd[name + '.weight'].to(torch.float64)
    cout = w.shape[0]
    b = sd[name + '.bias'].to(torch.float64) if has_bias else torch.zeros(cout, dtype=torch.float64)
    if bn:
        g, beta = sd[bn + '.weight'].to(torch.float64), sd[bn + '.bias'].to(torch.float64)
        mu, var = sd[bn + '.running_mean'].to(torch.float64), sd[bn + '.running_var'].to(torch.float64)
        scale = g / torch.sqrt(var + BN_EPS)
        shift = beta + (b - mu) * scale
    else:
        scale, shift = torch.ones(cout, dtype=torch.float64), b
    return w.to(torch.float32), scale.to(torch.float32), shift.to(torch.float32)


class Weights:
    def __init__(self, net, sd, dev):
        self.units = {u[0]: u for u in net.conv_units()}
        self.sd, self.dev = sd, dev
        self.cache = {}

    def get(self, op):
        """(w_scaled fp32 (cout,cin,k,k) = w * BN scale in fp32 -- what the packers round --, shift fp32 (cout))."""
        name = op['name']
        if name in self.cache:
            return self.cache[name]
        if name in self.units:
            _, bn, cin, cout, k, stride, has_bias = self.units[name]
            w, sc, sh = folded(self.sd, name, bn, has_bias)
            ws = w * sc[:, None, None, None]
        else:                                   # head-internal slice of last_layer.0: columns col_off.., BN scale folded, no shift
            _, bn, cin0, cout0, _, _, has_bias = self.units['model.last_layer.0']
            w, sc, _ = folded(self.sd, 'model.last_layer.0', bn, has_bias)
            ws = torch.zeros((op['cout'], op['cin'], 1, 1), dtype=torch.float32)
            ws[:cout0] = w[:, op['col_off']:op['col_off'] + op['cin']] * sc[:, None, None, None]
            sh = torch.zeros(op['cout'], dtype=torch.float32)
            if name == 'headx.d':               # split head (fp32 engine): the direct tensor's product carries last_layer.0's shift
                sh[:cout0] = folded(self.sd, 'model.last_layer.0', bn, has_bias)[2]
        out = (ws.to(self.dev), sh.to(self.dev))
        self.cache[name] = out
        return out


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def e4m3_decode(u8):
    return u8.view(torch.float8_e4m3fn).to(torch.float32)


def e4m3_round(t):
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def nchw(t):            # (N,H,W,C) any dtype -> (N,C,H,W) fp32
    return t.to(torch.float32).permute(0, 3, 1, 2).contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def conv_ref(x_nchw, w, stride):
    k = w.shape[-1]
    return F.conv2d(x_nchw, w, None, stride=stride, padding=k // 2)


def lerp_tables(out_size, in_size, dev):
    """ops.hip lerp_idx / head32.hip: PyTorch's align_corners=True index in fp32.  Returns i0, i1 (long), w0, w1 (fp32)."""
    scale = (torch.tensor(float(in_size - 1), dtype=torch.float32) / torch.tensor(float(out_size - 1), dtype=torch.float32)) \
        if out_size > 1 else torch.tensor(0.0, dtype=torch.float32)
    o = torch.arange(out_size, dtype=torch.float32)
    src = scale * o
    i0 = src.to(torch.int64).clamp(max=in_size - 1)
    i1 = i0 + (i0 < in_size - 1).to(torch.int64)
    w1 = src - i0.to(torch.float32)
    w0 = 1.0 - w1
    return i0.to(dev), i1.to(dev), w0.to(dev), w1.to(dev)


def bilinear_up(src_nhwc, H, W, weight_round):
    """src (N,Hs,Ws,C) fp32 -> (N,H,W,C): sum of the four taps x weights; weight_round rounds the four PRODUCT weights (bf16 in
    upsample_add's bf16 path and in the head's gather, identity in the head's folded branches and in the fp32 engine)."""
    N, Hs, Ws, C = src_nhwc.shape
    dev = src_nhwc.device
    y0, y1, wy0, wy1 = lerp_tables(H, Hs, dev)
    x0, x1, wx0, wx1 = lerp_tables(W, Ws, dev)
    out = torch.zeros((N, H, W, C), dtype=torch.float32, device=dev)
    for yi, wy in ((y0, wy0), (y1, wy1)):
        rows = src_nhwc[:, yi]                                   # (N,H,Ws,C)
        for xi, wx in ((x0, wx0), (x1, wx1)):
            wgt = weight_round(wx[None, :] * wy[:, None])        # (H,W): lx.w * ly.w as the kernels form it
            out += rows[:, :, xi] * wgt[None, :, :, None]
    return out


def check(name, got, ref, rel, abs_, stats, kernel, flip_slack=None, flip_fraction=0.0):
    """|got - ref| <= rel |ref| + abs_ everywhere.  flip_slack (same shape, optional): where an INTERNAL bf16 rounding sits between
    two matrix products (the head's hidden vector) kernel and reference may round a few hidden values to different neighbours;
    then at most `flip_fraction` of the elements may exceed the tight tolerance, none may exceed tight + flip_slack, and the mean
    error stays far below either (a systematic error -- one pixel position of every tile -- fails all three)."""
    err = (got - ref).abs()
    tol = rel * ref.abs() + abs_
    bad = err > tol
    if flip_slack is not None:
        frac = float(bad.float().mean())
        assert frac <= flip_fraction, f'{kernel} / {name}: {frac:.2e} of the elements beyond the tight tolerance (allowed {flip_fraction:.0e})'
        assert float(err.mean()) <= 0.02 * float(tol.mean()), f'{kernel} / {name}: mean |err| {float(err.mean()):.3e}'
        tol = tol + flip_slack
        bad = err > tol
    worst = float((err / tol).max())
    row = stats.setdefault(kernel, dict(ops=0, elements=0, worst_err_over_tol=0.0, max_abs_err=0.0, worst_op=''))
    row['ops'] += 1
    row['elements'] += int(err.numel())
    row['max_abs_err'] = max(row['max_abs_err'], float(err.max()))
    if worst > row['worst_err_over_tol']:
        row['worst_err_over_tol'], row['worst_op'] = worst, name
    if bad.any():
        idx = [int(v) for v in torch.nonzero(bad)[0]]
        raise AssertionError(f'{kernel} / {name}: {int(bad.sum())} of {bad.numel()} elements beyond tolerance; first at {idx}: '
                             f'got {float(got[tuple(idx)])} want {float(ref[tuple(idx)])}; max |err| {float(err.max())}')


BF16_ULP = 2.0 ** -7          # largest ulp / |y| of a bf16 value
E4M3_ULP = 2.0 ** -3


def run_case(sncal, cuda, cfg, sd, x, dtype, fp8_layers=None, want_heat=True):
    """One forward with every op tapped; returns (ops, tensors dict (op idx, tensor id) -> torch tensor, net)."""
    global X3_T
    global X3_SLACK
    X3_T = torch.float16 if sncal._lib.lib().sncal_x3_name() == b'fp16x3' else torch.bfloat16
    X3_SLACK = 2e-6 if X3_T == torch.float16 else 3e-5
    net = sncal.HRNetHeatmap(cfg, dtype=dtype, device=cuda)
    net.load_state_dict(sd)
    if dtype == 'fp8':
        net.calibrate_fp8(x)
        net.set_fp8_layers(fp8_layers or 'all')
    net.set_profiling(1)                                    # labels: which kernel served which op
    net.forward(x, want_heat=want_heat, decode_size=(540, 960) if net.cfg.get('head', 'logsoftmax') == 'logsoftmax' else None)
    torch.cuda.synchronize()
    ops = net.plan_ops()
    net.set_profiling(0)
    taps = {}
    for op in ops:
        if not op['active'] or op['type'] == 'decode':
            continue
        ids = [op['in'], op['res'], op['out'], op['base'], op['head_direct']] + op['src'] + op['head_src'] + op['head_fold']
        for t in ids:
            if t is None or t < 0:
                continue
            info = net.plan_tensor(t)
            cand = [t] + ([info['twin']] if info['twin'] >= 0 else [])
            for tid in cand:
                if net.plan_tensor(tid)['alive'] and (op['idx'], tid) not in taps:
                    taps[(op['idx'], tid)] = net.tap(op['idx'], tid)
    net.forward(x, want_heat=want_heat, decode_size=(540, 960) if net.cfg.get('head', 'logsoftmax') == 'logsoftmax' else None)
    torch.cuda.synchronize()
    net.clear_taps()
    return ops, taps, net


def verify_plan(sncal, cuda, cfg, sd, x, dtype, fp8_layers=None, tag=''):
    ops, taps, net = run_case(sncal, cuda, cfg, sd, x, dtype, fp8_layers)
    W = Weights(net, sd, cuda)
    f32_engine = dtype in ('fp32', 'fp16x3')
    act_round = (lambda t: t) if f32_engine else bf16r
    rel_out = 1e-5 if f32_engine else BF16_ULP
    abs_out = 2e-4 if f32_engine else 1e-3
    stats = {}
    B = min(x.shape[0], net.plan_tensor(0)['sub_batch'])
    x = x[:B]
    by_idx = {op['idx']: op for op in ops}
    label = ''
    fused_second = set()

    def T(op, tid):
        return taps[(op['idx'], tid)]

    for op in ops:
        if not op['active']:
            continue
        if op['kernel']:
            label = op['kernel']                      # grouped members / the second conv of a fused block carry no label of their own
        if op['type'] == 'input':
            got = T(op, op['out']).to(torch.float32)
            ref = torch.zeros_like(got)
            ref[..., :3] = act_round(nhwc(x))
            check('input layout', got, ref, 0.0, 0.0, stats, 'nchw_to_nhwc')
        elif op['type'] == 'conv':
            if op['idx'] in fused_second:
                continue
            w, shift = W.get(op)
            ti = net.plan_tensor(op['in'])
            if label == 'bblock48_fused' and op['kernel'] == 'bblock48_fused':
                nxt = by_idx[op['idx'] + 1]
                assert nxt['type'] == 'conv' and nxt['in'] == op['out'] and nxt['res'] == op['in'], 'fused block shape'
                fused_second.add(nxt['idx'])
                w2, shift2 = W.get(nxt)
                xin = nchw(taps[(nxt['idx'], op['in'])])                       # x as it stood when the fused launch was done
                mid = bf16r(torch.relu(conv_ref(xin, bf16r(w), 1) + shift[None, :, None, None]))
                ref = conv_ref(mid, bf16r(w2), 1) + shift2[None, :, None, None] + xin
                ref = torch.relu(ref)
                got = nchw(taps[(nxt['idx'], nxt['out'])])
                # the intermediate tile is rounded to bf16 between the two convolutions: where kernel and reference round a mid value to
                # different neighbours (their fp32 sums differ in the last bits) the output moves by ulp(mid) x |w2| ~ 4e-4 per flip
                check(op['name'] + ' + conv2', got, ref, rel_out, 4e-3, stats, 'bblock48_fused')
                continue
            if label == 'bblockx3_fused' and op['kernel'] == 'bblockx3_fused':
                # fp16x3, fused 48-channel BasicBlock (bblockx3.hip): x (split twin) -> conv1 + shift, ReLU -> mid (split in LDS, never in
                # HBM) -> conv2 + shift + x, ReLU -> split twin and / or fp32.  Every product good to ~2^-16: 3e-5 of sum |x w| per
                # convolution, the first one's bound carried through |w2|
                nxt = by_idx[op['idx'] + 1]
                assert nxt['type'] == 'conv' and nxt['in'] == op['out'] and nxt['res'] == op['in'], 'fused block shape'
                fused_second.add(nxt['idx'])
                w2, shift2 = W.get(nxt)
                xin = _split_twin_value(taps[(nxt['idx'], ti['twin'])])
                mid = torch.relu(conv_ref(xin, w, 1) + shift[None, :, None, None])
                y = torch.relu(conv_ref(mid, w2, 1) + shift2[None, :, None, None] + xin)
                slack = X3_SLACK * conv_ref(mid.abs(), w2.abs(), 1) + conv_ref(X3_SLACK * conv_ref(xin.abs(), w.abs(), 1), w2.abs(), 1)
                to = net.plan_tensor(nxt['out'])
                name = f"{op['name']} + conv2 {to['H']}x{to['W']} 48->48->48"
                checked = False
                if to['alive'] and _bf16_written(net, ops, nxt):
                    got = nchw(taps[(nxt['idx'], nxt['out'])])[:, nxt['out_coff']:nxt['out_coff'] + 48]
                    check(name, got, y, rel_out, abs_out + slack, stats, 'bblockx3_fused')
                    checked = True
                if to['twin'] >= 0 and net.plan_tensor(to['twin'])['alive'] and (nxt['idx'], to['twin']) in taps:
                    check(name + ' [split twin out]', _split_twin_value(taps[(nxt['idx'], to['twin'])]), y, 1e-5 + 2.0 ** -16, abs_out + slack, stats, 'bblockx3_fused split out')
                    checked = True
                assert checked, name
                continue
            if label == 'bneck_tail_ds_x3' and op['kernel'] == 'bneck_tail_ds_x3':
                # fp16x3, layer1 block 0 (bneckx3.hip): this op is the downsample branch, the next one the conv3 that adds it; one launch
                # computes ReLU(W3 . h2 + shift3 + Wds . x0 + shift_ds), the 256-channel branch tensor is never written
                nxt = by_idx[op['idx'] + 1]
                assert nxt['type'] == 'conv' and nxt['res'] == op['out'] and nxt['relu'] and not op['relu'], 'fused tail shape'
                fused_second.add(nxt['idx'])
                w3, shift3 = W.get(nxt)
                x0 = nchw(T(op, op['in']))[:, :op['cin']]
                h2 = nchw(taps[(nxt['idx'], nxt['in'])])[:, :nxt['cin']]
                y = torch.relu(conv_ref(h2, w3, 1) + shift3[None, :, None, None] + conv_ref(x0, w, 1) + shift[None, :, None, None])
                slack = X3_SLACK * (conv_ref(h2.abs(), w3.abs(), 1) + conv_ref(x0.abs(), w.abs(), 1))
                to = net.plan_tensor(nxt['out'])
                got = nchw(taps[(nxt['idx'], nxt['out'])])
                check(f"{nxt['name']} + downsample {to['H']}x{to['W']} 64+64->256", got, y, rel_out, abs_out + slack, stats, 'bneck_tail_ds_x3')
                continue
            if op['fp8']:
                tw = net.plan_tensor(ti['twin'])
                xin = e4m3_decode(T(op, ti['twin'])).permute(0, 3, 1, 2).contiguous()            # codes
                wmax = w.abs().amax(dim=(1, 2, 3))
                wscale = torch.where(wmax > 0, wmax / 448.0, torch.ones_like(wmax))
                wq = e4m3_round(w / wscale[:, None, None, None])
                oscale = (torch.tensor(tw['scale'], dtype=torch.float32, device=cuda) * wscale)  # fp32 product, as the host forms it
                y = conv_ref(xin, wq, 1) * oscale[None, :, None, None] + shift[None, :, None, None]
                # v_mfma_scale_f32_32x32x64_f8f6f4 does NOT accumulate in full fp32 (tools/dev/fp8_acc_probe.hip, measured): the 64
                # products of a K block are summed in groups of 8 whose members are aligned to the group's largest product and
                # TRUNCATED 2^-13 below it (a 2^8 product keeps 2^-5 neighbours exactly and drops 2^-6 ones entirely).  Each term can
                # lose up to 2^-13 of its group's maximum: <= 7 x 2^-14 of the sum of |x w| on average, one-sided.  Measured on these
                # layers: up to 1.5e-4 = 2^-12.7 of sum |x w|; allowed: 2^-11 of it (a wrong tap moves the sum by ~2^-6 of it)
                fp8_slack = 2.0 ** -11 * conv_ref(xin.abs(), wq.abs(), 1) * oscale[None, :, None, None]
            elif op.get('x3'):
                # fp16x3: fp32 operands split into bf16 hi + lo, hi.hi + hi.lo + lo.hi in fp32: every product is good to ~2^-16 of
                # itself, so the sum is good to a few 1e-5 of sum |x w| (measured below 1e-5); a wrong tap is 2^-6 of it
                raw = T(op, ti['twin'])                       # the operand the kernel reads: the split twin, hi + lo = x to 2^-17
                pr = raw.view(X3_T).reshape(raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3] // 16, 2, 16).to(torch.float32)
                xin = (pr[..., 0, :] + pr[..., 1, :]).reshape(raw.shape).permute(0, 3, 1, 2).contiguous()
                y = conv_ref(xin, w, op['stride']) + shift[None, :, None, None]
                fp8_slack = X3_SLACK * conv_ref(xin.abs(), w.abs(), op['stride'])
            else:
                xin = nchw(T(op, op['in']))[:, :op['cin']]
                y = conv_ref(xin, act_round(w), op['stride']) + shift[None, :, None, None]
                # fp16x3 engine, generic kernel (x3_t): fp32 operands split in registers, same arithmetic and the same bound as above
                fp8_slack = X3_SLACK * conv_ref(xin.abs(), w.abs(), op['stride']) if op.get('x3g') else 0.0
            if op['res'] >= 0 and op.get('res_twin'):
                # fp16x3, inside a BasicBlock chain: the block input lives only as the split twin its first convolution read
                raw = T(op, net.plan_tensor(op['res'])['twin'])
                pr = raw.view(X3_T).reshape(raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3] // 16, 2, 16).to(torch.float32)
                y = y + (pr[..., 0, :] + pr[..., 1, :]).reshape(raw.shape).permute(0, 3, 1, 2)
            elif op['res'] >= 0:
                y = y + nchw(T(op, op['res']))[:, op['out_coff']:op['out_coff'] + op['cout']]
            if op['relu']:
                y = torch.relu(y)
            to = net.plan_tensor(op['out'])
            kern = 'conv_tt<fp8,k3,s1,8x32x96>' if op['fp8'] else ('conv_tt<fp16x3,k3,s1,12x32x64>' if op['cout'] % 96 else 'conv_tt<fp16x3,k3,s1,8x32x96>') if op.get('x3') else label
            name = f"{op['name']} {to['H']}x{to['W']} {op['cin']}->{op['cout']}" + ('+res' if op['res'] >= 0 else '')
            checked = False
            gen_twin = bool(op.get('x3g')) and _producer_twin(net, op, taps)
            if gen_twin:
                check(name + ' [split twin out]', _split_twin_value(T(op, to['twin'])), y, 1e-5 + 2.0 ** -16, abs_out + fp8_slack, stats, kern + ' split out')
                checked = True
            if to['alive'] and not ((op['fp8'] or op.get('x3') or gen_twin) and not _bf16_written(net, ops, op)):
                got = nchw(T(op, op['out']))[:, op['out_coff']:op['out_coff'] + op['cout']]
                if op['out_f32']:
                    check(name, got, y, 1e-5 if f32_engine else 2.0 ** -9, abs_out, stats, kern)     # fp32 logits of bf16 operands
                else:
                    check(name, got, y, rel_out, abs_out + fp8_slack, stats, kern)
                checked = True
            if op.get('x3') and to['twin'] >= 0 and net.plan_tensor(to['twin'])['alive'] and (op['idx'], to['twin']) in taps:
                # split twin written by the epilogue: [16 hi | 16 lo] bf16 per 16-channel group; hi + lo reproduces y to 2^-17
                raw = T(op, to['twin'])                                                  # fp32-typed storage, (N,H,W,C)
                pr = raw.view(X3_T).reshape(raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3] // 16, 2, 16).to(torch.float32)
                got_t = (pr[..., 0, :] + pr[..., 1, :]).reshape(raw.shape).permute(0, 3, 1, 2)
                check(name + ' [split twin out]', got_t, y, 1e-5 + 2.0 ** -16, abs_out + fp8_slack, stats, kern + ' split out')
                checked = True
            if op['fp8'] and to['twin'] >= 0 and net.plan_tensor(to['twin'])['alive']:
                tw_o = net.plan_tensor(to['twin'])
                got8 = e4m3_decode(T(op, to['twin'])).permute(0, 3, 1, 2) * tw_o['scale']
                ref8 = e4m3_round(bf16r(y) / tw_o['scale']) * tw_o['scale']
                check(name + ' [e4m3 out]', got8, ref8, E4M3_ULP, tw_o['scale'] * 2.0 ** -9 + 1e-3 + fp8_slack, stats, kern + ' e4m3 out')
                checked = True
            assert checked, name
        elif op['type'] == 'upsample_add':
            to = net.plan_tensor(op['out'])
            wr = (lambda t: t) if f32_engine else bf16r
            acc = None
            C0 = None
            for s in op['src']:
                src = T(op, s).to(torch.float32)
                C0 = src.shape[-1]
                up = bilinear_up(src, to['H'], to['W'], wr)
                acc = up if acc is None else acc + up
            if op['base'] >= 0:
                acc = acc + T(op, op['base']).to(torch.float32)
            if op['relu']:
                acc = torch.relu(acc)
            nm = f"fuse sum -> {to['H']}x{to['W']}x{C0} ({len(op['src'])} sources)"
            tw = dtype == 'fp16x3' and _producer_twin(net, op, taps)
            if tw:          # fp16x3: the sum's split twin for the two-team convolution that reads it; the fp32 form only if somebody reads that
                check(nm + ' [split twin out]', _split_twin_value(T(op, to['twin'])).permute(0, 2, 3, 1), acc, 1e-5 + 2.0 ** -16, abs_out, stats, 'upsample_add split out')
            if not tw or _bf16_written(net, ops, op):
                got = T(op, op['out']).to(torch.float32)[..., op['out_coff']:op['out_coff'] + C0]
                check(nm, got, acc, rel_out, abs_out, stats, 'upsample_add')
        elif op['type'] == 'head':
            head_reference(net, op, T, W, by_idx, stats, cuda, exact=f32_engine)
        elif op['type'] == 'softmax':
            logits = T(op, op['in'])                                       # (N,H,W,Cpad) fp32
            C = net.num_classes
            to = net.plan_tensor(op['out'])
            got = T(op, op['out']).reshape(-1).view(logits.shape[0], C, to['H'], to['W'])     # the heat tensor is NCHW
            lg = logits[..., :C].permute(0, 3, 1, 2)
            ref = torch.softmax(lg, dim=1) if net.cfg.get('head') == 'softmax' else torch.log_softmax(lg, dim=1)
            check('softmax head', got, ref, 1e-5, 1e-5, stats, 'softmax_nchw')
    stats['_case'] = dict(tag=tag, dtype=dtype, fp8_layers=fp8_layers, frames=int(B), input=list(x.shape[2:]))
    return stats


def _split_twin_value(raw):
    """(N,H,W,C) fp32-typed storage of a split twin -> (N,C,H,W) fp32: hi + lo."""
    pr = raw.view(X3_T).reshape(raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3] // 16, 2, 16).to(torch.float32)
    return (pr[..., 0, :] + pr[..., 1, :]).reshape(raw.shape).permute(0, 3, 1, 2)


def _producer_twin(net, op, taps):
    """fp16x3: does this generic convolution / fuse sum write the split twin of its (dense) output?  (hrnet.cpp producer_twin)"""
    to = net.plan_tensor(op['out'])
    if to['twin'] < 0 or not net.plan_tensor(to['twin'])['alive'] or (op['idx'], to['twin']) not in taps:
        return False
    return op['out_coff'] == 0 and to['C'] % 16 == 0 and (op['type'] == 'upsample_add' or (op.get('x3g') and to['C'] == op['cout'] and not op['out_f32']))


def _bf16_written(net, ops, op):
    """An fp8 convolution writes its bf16 output only when somebody reads it (residuals, fuse layers, bf16 convolutions)."""
    t = op['out']
    for o in ops:
        if not o['active'] or o['idx'] <= op['idx']:
            continue
        readers = [-1 if o.get('res_twin') else o['res'], o['base'], o['head_direct']] + o['src'] + o['head_src'] + o['head_fold']
        if t in readers:
            return True
        if o['in'] == t and not (o['type'] == 'conv' and (o['fp8'] or o.get('x3'))):
            return True
    return False


def head_reference(net, op, T, W, by_idx, stats, dev, exact=False):
    """head32.hip / head.hip on their own operands: hidden = relu(b0 + W0[:, :K1] . [direct | up(narrow branches)] + sum_s up(t_s)),
    logits = W1 . bf16(hidden) + b1, with the roundings the kernel applies (bf16 blends of the folded branches, bf16 gather
    weights, bf16 hidden vector)."""
    to = net.plan_tensor(op['out'])
    H, Wd = to['H'], to['W']
    rnd = (lambda t: t) if exact else bf16r          # fp16x3 engine (headx3.hip): fp32 operands, every product good to ~2^-16
    direct = T(op, op['head_direct']).to(torch.float32)                    # (N,H,W,Cd)
    parts = [direct]
    for f in op['head_fold']:
        parts.append(rnd(bilinear_up(T(op, f).to(torch.float32), H, Wd, lambda t: t)))
    kin = torch.cat(parts, dim=-1)                                          # (N,H,W,K1)
    K1 = kin.shape[-1]
    units = W.units
    _, bn0, cin0, cout0, _, _, hb0 = units['model.last_layer.0']
    w0, sc0, sh0 = folded(W.sd, 'model.last_layer.0', bn0, hb0)
    w0s = rnd((w0 * sc0[:, None, None, None])[:, :K1, 0, 0].to(dev))      # (784,K1)
    hid = kin.reshape(-1, K1) @ w0s.t() + sh0.to(dev)[None]
    hid = hid.reshape(kin.shape[0], H, Wd, cout0)
    for s in op['head_src']:
        t = T(op, s).to(torch.float32)[..., :cout0]
        hid = hid + bilinear_up(t, H, Wd, rnd)
    hid = rnd(torch.relu(hid))
    _, _, cin1, cout1, _, _, hb1 = units['model.last_layer.3']
    w1, sc1, sh1 = folded(W.sd, 'model.last_layer.3', '', hb1)
    w1s = rnd(w1[:, :, 0, 0].to(dev))
    ref = hid.reshape(-1, cout0) @ w1s.t() + sh1.to(dev)[None]
    ref = ref.reshape(kin.shape[0], H, Wd, cout1)
    got = T(op, op['out'])[..., :cout1]
    if exact:
        check(f'head -> logits {H}x{Wd} (split-fp16)', got, ref, 1e-4, 5e-4, stats, 'headx3_fused')
        return
    # the hidden vector is rounded to bf16 BEFORE the 784-term second product: a hidden value whose fp32 sums differ in the last bits
    # between kernel and reference rounds to the other neighbour (one bf16 ulp of that hidden unit x |w1|).  Per element: up to eight
    # flips of the pixel's largest hidden value; measured 2e-5 .. 2e-3 of the elements leave the tight tolerance that way
    hmax = hid.amax(dim=-1, keepdim=True)                                   # (N,H,W,1)
    ulp = torch.exp2(torch.floor(torch.log2(hmax.clamp(min=2.0 ** -20))) - 7.0)
    flip = 8.0 * ulp * w1s.abs().amax(dim=1)[None, None, None, :]
    check(f'head -> logits {H}x{Wd}', got, ref, 2.0 ** -9, 4e-3, stats, 'head_fused', flip_slack=flip, flip_fraction=5e-3)


def _weights(cfg):
    import bench
    return bench.seeded_weights(cfg, seed=1)


def _report(stats, name):
    print('KERNEL-PARITY', json.dumps(stats))
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open(os.path.join('gpurun_out', f'kernel_parity_{name}.json'), 'w') as f:
            json.dump(stats, f, indent=1)
    except OSError:
        pass


def _frames(B, H, W, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((B, 3, H, W), generator=g, dtype=torch.float32).to(dev)


def test_every_launch_of_the_bf16_engine_w48_540p(sncal, cuda):
    """The benchmarked engine at the metric's size, 3 frames: branch maps 135x240 / 68x120 / 34x60 / 17x30 (partial 32-pixel tiles on
    every branch, two stacked-frame boundaries inside the tiles)."""
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(3, 540, 960, 11, cuda), 'bf16', tag='w48 540p')
    _report(stats, 'bf16_w48_540p')
    for k in ('conv_tt<bf16,k3,s1,8x32x96>', 'bblock48_fused', 'head_fused', 'upsample_add'):
        assert k in stats and stats[k]['ops'] > 0, (k, list(stats))
    assert any(k.startswith('conv<bf16,k3,s2') for k in stats) and any(k.startswith('conv<bf16,k1,s1') for k in stats)
    assert stats['conv_tt<bf16,k3,s1,8x32x96>']['ops'] == 144 and stats['bblock48_fused']['ops'] == 32


def test_every_launch_of_the_bf16_engine_w48_1080p(sncal, cuda):
    """C5's shapes (270x480 / 135x240 / 68x120 / 34x60 branches, 540x960 head), 2 frames."""
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(2, 1080, 1920, 12, cuda), 'bf16', tag='w48 1080p')
    _report(stats, 'bf16_w48_1080p')
    assert stats['conv_tt<bf16,k3,s1,8x32x96>']['ops'] == 144 and stats['bblock48_fused']['ops'] == 32


def test_every_launch_of_the_bf16_engine_odd_sizes(sncal, cuda):
    """270x500 input: 68x125 / 34x63 / 17x32 / 9x16 maps -- widths that are not multiples of the tiles, a 9-row branch (stacked
    frames interleave inside one 8-row tile), and the stem-interpolation head path."""
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(5, 270, 500, 13, cuda), 'bf16', tag='w48 270x500')
    _report(stats, 'bf16_w48_270x500')
    assert stats['conv_tt<bf16,k3,s1,8x32x96>']['ops'] == 144


@pytest.mark.parametrize('layers', ['all', 'stage4,c192,c384'])
def test_every_launch_of_the_fp8_engine_w48_540p(sncal, cuda, layers):
    """C5 arithmetic: the e4m3 convolutions against torch fp32 on the e4m3 codes they read (per-tensor input scale x per-channel
    weight scale folded into the reference), their bf16 outputs and their e4m3 twin outputs; everything else as the bf16 engine."""
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(3, 540, 960, 14, cuda), 'fp8', fp8_layers=layers, tag='w48 540p fp8 ' + layers)
    _report(stats, 'fp8_w48_540p_' + layers.replace(',', '_'))
    k = 'conv_tt<fp8,k3,s1,8x32x96>'
    assert k in stats and k + ' e4m3 out' in stats
    n8 = stats[k]['ops'] + 0
    assert (stats[k + ' e4m3 out']['ops'] > 0) and (n8 > 0)
    if layers == 'all':
        assert stats[k]['ops'] + stats[k + ' e4m3 out']['ops'] >= 144          # every one of the 144 wide convolutions checked on at least one output
        assert stats[k + ' e4m3 out']['ops'] >= 120                            # all but the last convolution of each chain hand an e4m3 twin on


def test_every_launch_of_the_fp8_engine_w48_1080p(sncal, cuda):
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(2, 1080, 1920, 15, cuda), 'fp8', fp8_layers='all', tag='w48 1080p fp8')
    _report(stats, 'fp8_w48_1080p')
    assert 'conv_tt<fp8,k3,s1,8x32x96>' in stats


def test_every_launch_of_the_line_network_bf16(sncal, cuda):
    """L1: the line network (no upscale, 720-channel head, Softmax) -- the head's other configuration."""
    sd = _weights('line_hrnet_w48')
    stats = verify_plan(sncal, cuda, 'line_hrnet_w48', sd, _frames(3, 540, 960, 16, cuda), 'bf16', tag='line w48 540p')
    _report(stats, 'bf16_line_w48_540p')
    assert 'head_fused' in stats and 'softmax_nchw' in stats


def test_every_launch_of_the_fp32_engine_w18(sncal, cuda):
    """The exact engine's kernels one by one (fp32 MFMA generic conv, fp32 upsample / concat, reference-formulation head)."""
    from oracle import hrnet_ref as hr
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 3, 4.0)
    stats = verify_plan(sncal, cuda, 'hrnet_w18', sd, _frames(3, 135, 240, 17, cuda), 'fp32', tag='w18 135x240 fp32')
    _report(stats, 'fp32_w18_135x240')
    assert any(k.startswith('conv<f32') for k in stats)


@pytest.mark.parametrize('small_items', ['2', '0'])
def test_every_launch_of_the_fp16x3_engine_w48_540p(sncal, cuda, monkeypatch, small_items):
    """The fp32-class engine: fp32 tensors everywhere, the 3x3 stride-1 convolutions of stages 2-4 (wide branches and the 48-channel
    branch as fused BasicBlocks, bblockx3.hip) in split-fp16 arithmetic -- each against torch fp32 on the
    split twin it reads (hi + lo; written by the producing convolution's epilogue or by split_f32_kernel), its fp32 output and the
    split twin it hands on.  Twice: three frames are a SMALL launch and take the two-team kernel's 96 x 4 x 32 tile by default
    (SNCAL_TT_SMALL_ITEMS = 2 items per team, hrnet.cpp run_conv_tt); 0 forces the 96 x 8 x 32 tile of the large batches."""
    monkeypatch.setenv('SNCAL_TT_SMALL_ITEMS', small_items)
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(3, 540, 960, 18, cuda), 'fp16x3', tag='w48 540p fp16x3 small tile ' + small_items)
    _report(stats, 'fp16x3_w48_540p' + ('' if small_items == '2' else '_tile8'))
    k, kb = 'conv_tt<fp16x3,k3,s1,8x32x96>', 'bblockx3_fused'   # 144 wide convolutions + 32 fused 48-channel blocks, each checked on its fp32 output, its twin, or both
    n = lambda key: stats.get(key, {'ops': 0})['ops']
    assert n(k) >= 12 and n(k + ' split out') >= 120                                         # (fp32 outputs: module ends only)
    assert n(k) + n(k + ' split out') >= 144 and n(kb) + n(kb + ' split out') >= 32 and n(kb) >= 8 and n(kb + ' split out') >= 24
    # layer1 (bneckx3.hip): block 0's conv3 with its downsample branch inside, and the seams conv3 + next conv1 of blocks 1 | 2 and 2 | 3
    # (two checks each: the 256-channel block output and the next block's 64-channel conv1 output)
    assert n('bneck_tail_ds_x3') == 1 and n('bneck_seam_x3') == 4, {k: v.get('ops') for k, v in stats.items()}


def test_every_launch_of_the_fp16x3_engine_with_the_pipelined_stride2_kernel(sncal, cuda, monkeypatch):
    """SNCAL_S2P=1: the 3x3 stride-2 convolutions (transitions and fuse-down chains, hrnet.py:183-214, 357-391) on the pipelined
    persistent kernel (conv_s2p.hip; parked: measured slower than the generic kernel, off by default) -- every launch against torch fp32
    like the default path, at W48 540p (96- and 48-channel n-blocks, shared launches) and W32 270p."""
    monkeypatch.setenv('SNCAL_S2P', '1')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', _weights('hrnet_w48'), _frames(3, 540, 960, 18, cuda), 'fp16x3', tag='w48 540p fp16x3 s2p')
    n = lambda key: sum(v.get('ops', 0) for k, v in stats.items() if k.startswith(key))
    assert n('conv_s2p<') >= 20 and n('conv_s2p_shared<') >= 8, {k: v.get('ops') for k, v in stats.items()}      # (members of a shared launch carry the first member's label)
    verify_plan(sncal, cuda, 'hrnet_w32', _weights('hrnet_w32'), _frames(2, 270, 480, 19, cuda), 'fp16x3', tag='w32 270p fp16x3 s2p')     # (whichever of its stride-2 layers pack at G = 3)


def test_every_launch_of_the_fp16x3_engine_w32_270p(sncal, cuda):
    """BASELINE config C2's shapes (HRNet-W32, 480x270): branch widths 32 / 64 / 128 / 256 all run as 64-channel blocks of the
    64 x 12 x 32 tile (the 32-channel branch half padded), maps 68x120 / 34x60 / 17x30 / 9x15 (a 9-row branch inside 12-row tiles)."""
    sd = _weights('hrnet_w32')
    stats = verify_plan(sncal, cuda, 'hrnet_w32', sd, _frames(5, 270, 480, 19, cuda), 'fp16x3', tag='w32 270p fp16x3')
    _report(stats, 'fp16x3_w32_270p')
    k48 = 'conv_tt<fp16x3,k3,s1,12x32x64>'
    n = lambda key: stats.get(key, {'ops': 0})['ops']
    assert n(k48) + n(k48 + ' split out') >= 200, {k: v['ops'] for k, v in stats.items()}
    assert any(k.startswith('conv<fp16x3,k3,s2') for k in stats) and any(k.startswith('conv<fp16x3,k1,s1') for k in stats)


def test_every_launch_of_the_fp16x3_engine_w48_1080p_and_odd_sizes(sncal, cuda):
    """C5's shapes (2 frames of 1920x1080) and a 270x500 input (68x125 / 34x63 / 17x32 / 9x16 maps, stem-interpolation head path)."""
    sd = _weights('hrnet_w48')
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(2, 1080, 1920, 20, cuda), 'fp16x3', tag='w48 1080p fp16x3')
    _report(stats, 'fp16x3_w48_1080p')
    assert 'conv_tt<fp16x3,k3,s1,8x32x96> split out' in stats and stats['bblockx3_fused split out']['ops'] >= 24 and stats['bblockx3_fused']['ops'] >= 8
    stats = verify_plan(sncal, cuda, 'hrnet_w48', sd, _frames(5, 270, 500, 21, cuda), 'fp16x3', tag='w48 270x500 fp16x3')
    _report(stats, 'fp16x3_w48_270x500')
    assert 'conv_tt<fp16x3,k3,s1,8x32x96> split out' in stats and stats['bblockx3_fused split out']['ops'] >= 24


def test_every_launch_of_the_fp16x3_engine_w18_and_line_net(sncal, cuda):
    """W18 (18 / 36 / 72 / 144 channels: no two-team tiles, every convolution on the generic split-arithmetic kernel) and the line
    network's head configuration."""
    from oracle import hrnet_ref as hr
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 3, 4.0)
    stats = verify_plan(sncal, cuda, 'hrnet_w18', sd, _frames(3, 135, 240, 22, cuda), 'fp16x3', tag='w18 135x240 fp16x3')
    _report(stats, 'fp16x3_w18_135x240')
    assert any(k.startswith('conv<fp16x3') for k in stats)
    sd = _weights('line_hrnet_w48')
    stats = verify_plan(sncal, cuda, 'line_hrnet_w48', sd, _frames(2, 540, 960, 23, cuda), 'fp16x3', tag='line w48 540p fp16x3')
    _report(stats, 'fp16x3_line_w48_540p')
