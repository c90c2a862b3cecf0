#!/usr/bin/env python3
"""Benchmark of the per-frame calibration hot path on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

One step = one pass of the hot path over one batch of frames already resident in HBM (BASELINE config C3:
HRNet-W48, 960x540, batch 64 per GPU): NCHW->NHWC, all convolutions / fuse / head kernels, log-softmax,
keypoint decode, and the batched camera solve (CameraCreator 'iterative_voter' with the make_submit.py
parameters).  The solve runs twice per step: on the keypoints decoded from the network output (real data
dependency; random-init weights give few confident points) and on a resident batch of synthetic
projected-template keypoints (the realistic solve workload).  With N > 1 every rank processes
its own 64 frames (weak scaling, frames are independent) and one RCCL all_gather of the per-frame records
closes the step.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The step uses three streams (network, solves, RCCL's internal one).  HIP multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that land on one queue run in submission order, and
# measured on MI355X the RCCL stream then shares a queue with the network stream: the all_gather's wait for the
# solves stalls the next step's convolutions, 43.1 -> 54.5 ms per step.  With 8 queues the collective costs nothing.
# Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME = 2 * 253910384640          # conv MACs of the reference's direct formulation x2
PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
BATCH = 64


def seeded_weights(cfg, seed):
    """Random-init weights of the W48 architecture (no checkpoints ship with the reference).  Same recipe as
    the test-suite generator, restated here so that the timed path never imports oracle/."""
    import sncal_amd
    rng = np.random.Generator(np.random.PCG64(seed))
    net = sncal_amd.HRNetHeatmap(cfg, dtype='bf16', device='cpu')
    sd = {}

    def uni(shape, lo, hi):
        return torch.from_numpy((rng.random(shape) * (hi - lo) + lo).astype(np.float32))
    for name, bn, cin, cout, k, stride, has_bias in net.conv_units():
        b = float(np.sqrt(6.0 / (cin * k * k)))
        sd[name + '.weight'] = uni((cout, cin, k, k), -b, b)
        if has_bias:
            sd[name + '.bias'] = uni((cout,), -0.1, 0.1)
        if bn:
            closing = bn.endswith(('bn2', 'bn3', 'downsample.1'))
            g = (0.15, 0.35) if closing else ((0.25, 0.45) if 'fuse_layers' in bn else (0.8, 1.2))
            sd[bn + '.weight'] = uni((cout,), *g)
            sd[bn + '.bias'] = uni((cout,), -0.1, 0.1)
            sd[bn + '.running_mean'] = uni((cout,), -0.1, 0.1)
            sd[bn + '.running_var'] = uni((cout,), 0.7, 1.4)
    return sd


def cpu_baseline(sd, cfg_name, n_frames=2, n_solve=6):
    """The oracle (CPU restatement of the reference's algorithm) timed on this host's cores: torch-CPU fp32
    HRNet forward + numpy decode on `n_frames` frames, numpy solve on `n_solve` synthetic frames."""
    from oracle import decode as od
    from oracle import hrnet_ref as hr
    from oracle import solve as osolve
    import sncal_amd
    cfg = hr.load_config(cfg_name)
    cores = torch.get_num_threads()
    x = torch.rand((1, 3, 540, 960))
    hr.forward(sd, x, cfg)                                    # warm-up (oneDNN primitive creation)
    t0 = time.time()
    for _ in range(n_frames):
        logp = hr.forward(sd, x, cfg)
    t_net = (time.time() - t0) / n_frames
    t0 = time.time()
    od.keypoint_decode(logp.numpy(), (540, 960))
    t_dec = time.time() - t0
    kps = sncal_amd.synth.synthetic_keypoints(n_solve, seed=123)
    oc = osolve.CameraCreatorOracle()
    t0 = time.time()
    for k in kps:
        oc(k, None)
    t_solve = (time.time() - t0) / n_solve
    return {'value': round(1.0 / (t_net + t_dec + t_solve), 4), 'unit': 'frames/s', 'cores': int(cores), 'kind': 'port',
            'sample': f'{n_frames} frames HRNet-W48 960x540 fp32 torch-CPU forward ({t_net:.2f} s/frame) + numpy decode '
                      f'({t_dec * 1e3:.0f} ms/frame) + {n_solve} frames numpy camera solve ({t_solve * 1e3:.0f} ms/frame), '
                      'single process, stages serial'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--lanes', type=int, default=1,
                    help='split the batch into this many independent sub-batches on their own streams (default 1; see DESIGN.md 5: '
                         '2 lanes fill kernel tails and launch gaps, +7 %% frames/s, but per-kernel HIP-event durations then '
                         'measure a shared GPU, so the roofline object is only meaningful at 1)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # SNCAL_BENCH_FORCE_DIST=1 (testing aid): take the multi-GPU code path -- RCCL init, the per-step all_gather on the
    # side stream, barrier, max-over-ranks -- even with one rank, so that it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get('SNCAL_BENCH_FORCE_DIST') == '1'
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    import sncal_amd
    cfg_name = 'hrnet_w48'
    sd = seeded_weights(cfg_name, seed=1)
    B = args.batch
    L = max(1, args.lanes)
    if B % L:
        raise SystemExit(f'--batch {B} is not a multiple of --lanes {L}')
    nets = []
    for _ in range(L):
        net = sncal_amd.HRNetHeatmap(cfg_name, dtype=args.dtype, device=dev)
        net.load_state_dict(sd)
        nets.append(net)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    x = torch.rand((B, 3, 540, 960), device=dev, generator=gen)          # synthetic frames, resident in HBM
    kp_synth = torch.from_numpy(sncal_amd.synth.synthetic_keypoints(B, seed=77 + rank)).to(dev)
    cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2],
                                 algorithm='iterative_voter', lines_file=None, max_rmse=55.0, max_rmse_rel=5.0,
                                 min_points=5, min_focal_length=10.0, min_points_per_plane=6,
                                 min_points_for_refinement=6, reliable_thresh=57)
    pipes = [sncal_amd.CalibrationPipeline(n, cc, decode_size=(540, 960)) for n in nets]
    lane_streams = [None] if L == 1 else [torch.cuda.Stream(device=dev) for _ in range(L)]
    bl = B // L
    xs = [x[i * bl:(i + 1) * bl] for i in range(L)]
    kps = [kp_synth[i * bl:(i + 1) * bl].contiguous() for i in range(L)]
    last = {}
    diag_nosolve = os.environ.get('SNCAL_BENCH_DIAG') == 'nosolve'
    diag_noprof = os.environ.get('SNCAL_BENCH_DIAG') == 'noprof'

    def step():
        # forward + decode on the main stream; both solves on the pipeline's side stream (they overlap the next
        # step's convolutions); every solve is complete before the closing fence of the timed region
        # multi-GPU: the single collective of the path (per-frame records to every rank, RCCL over xGMI) rides on
        # the side stream behind the solves
        for i in range(L):
            if diag_nosolve:                             # diagnosis only: network + decode, no solves (not a valid bench line)
                nets[i].forward(xs[i], want_heat=False, decode_size=(540, 960))
                continue
            if lane_streams[i] is None:
                out = pipes[i].submit(xs[i], extra_keypoints=kps[i], gather=use_dist)
            else:
                with torch.cuda.stream(lane_streams[i]):
                    out = pipes[i].submit(xs[i], extra_keypoints=kps[i], gather=use_dist)
            last[i] = out[2]

    def fence():
        for i in range(L):
            if lane_streams[i] is None:
                pipes[i].join()
            else:
                with torch.cuda.stream(lane_streams[i]):
                    pipes[i].join()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: the last warm-up step is profiled launch by launch (kernel time shares, and which variant dominates); the
    # timed region then times only that variant's launches -- the per-dispatch events cost ~4 us each, and the roofline
    # object needs the dominant kernel's duration, not those of the other ~150 launches of a step
    for _ in range(max(args.warmup - 1, 0)):
        step()
    fence()
    for n in nets:
        n.set_profiling(1)
    step()
    fence()
    warm = {}
    for n in nets:
        for q in n.get_profile():
            m = warm.setdefault(q['kernel'], dict(q, ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ('ms', 'launches', 'flops', 'bytes'):
                m[k] += q[k]
    for n in nets:
        n.set_profiling(0 if diag_noprof else 2)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    solve_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    merged = {}
    for n in nets:
        for q in n.get_profile():
            m = merged.setdefault(q['kernel'], dict(q, ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ('ms', 'launches', 'flops', 'bytes'):
                m[k] += q[k]
        n.set_profiling(False)
    prof = list(merged.values())
    if os.environ.get('SNCAL_BENCH_DIAG'):               # diagnosis runs print the step time only
        print('diag', os.environ['SNCAL_BENCH_DIAG'], round(dt / args.steps * 1e3, 3), 'ms/step')
        return
    # solve-stage time, measured separately after the timed region (torch events see torch's current stream,
    # which is the stream libsncal launches on)
    rec_syn = cc.solve_device(kp_synth)
    ev[0].record()
    for _ in range(3):
        cc.solve_device(kp_synth, out=rec_syn)
    ev[1].record()
    torch.cuda.synchronize()
    solve_ms = ev[0].elapsed_time(ev[1]) / 3

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    n_cam = sum(1 for r in cc.records(rec_syn) if r.status != 0)

    if rank == 0:
        assert len(prof) == 1, [p['kernel'] for p in prof]      # focus mode: the dominant variant only
        dom = prof[0]
        warm = list(warm.values())
        total_ms = sum(p['ms'] for p in warm)
        warm_dom = next(p for p in warm if p['kernel'] == dom['kernel'])
        ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        traffic = None      # HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/r01_pmc_hbm_traffic.md)
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
                traffic = json.load(f).get(dom['kernel'])
        except OSError:
            pass
        peak = PEAK_TFLOPS[args.dtype]
        out = {
            'metric': 'frames/sec (HRNet-W48 960x540 + PnP)', 'value': round(world * B * args.steps / dt, 2),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic (uniform-noise frames, random-init HRNet-W48; solve also driven by projected-template keypoints)',
            'config': {'workload': 'C3: HRNet-W48 960x540, batch 64 per GPU, heatmap + decode + batched camera solve (iterative_voter)',
                       'frames_per_gpu': B, 'lanes': L, 'parallelism': f'frames sharded over {world} GPU(s), one all_gather per step' if world > 1 else 'single GPU',
                       'solve_ms_per_batch': round(solve_ms, 3), 'cameras_found': f'{n_cam}/{B}',
                       'network_tflops_reference_formulation': round(world * B * args.steps / dt * FLOP_PER_FRAME / 1e12, 1),
                       'kernel_time_share_last_warmup_step': {p['kernel']: round(p['ms'] / total_ms, 4) for p in sorted(warm, key=lambda q: -q['ms'])[:8]}},
            'roofline': {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(ach / peak, 4), 'traffic': traffic, 'launches': dom['launches'],
                         'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
                         'flops_per_launch': round(dom['flops'] / dom['launches'], 0),
                         'share_of_gpu_time': round(warm_dom['ms'] / total_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, cfg_name)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
