#!/usr/bin/env python3
"""Benchmark of the per-frame calibration hot path on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c4] [--dtype fp16x3|bf16|fp32|fp8] [--size 540p|1080p]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

One step = one pass of the hot path over one batch of frames already resident in HBM (BASELINE config C3: HRNet-W48,
960x540, batch 64 per GPU): NCHW->NHWC, all convolutions / fuse / head kernels, log-softmax + keypoint decode, and the
batched camera solve (CameraCreator 'iterative_voter' with the make_submit.py parameters) ON THE KEYPOINTS THE NETWORK
DECODED -- the data dependency of the real path.  --workload c4 adds the line network (HRNet-W48 stride 4), its two-peak
decode and the on-device line join in front of the same solve (BASELINE config C4).

Data.  No trained checkpoint ships with the reference and random-init weights give flat heatmaps whose argmax cannot
drive the solve.  The workload therefore keeps the random-init W48 network and installs a designed signal path THROUGH IT
(sncal_amd.synth.peaked_state_dict(deep=True) = deep_state_dict / stamped_frames): frames carry per-keypoint code stamps at the
projections of the pitch template through sampled broadcast cameras; the code of keypoint class k travels through channel k of
every backbone tensor (stem, Bottlenecks, transitions, every BasicBlock of every branch, every fuse layer) and reaches the head
through the upsampled branch channels, on top of the random network's noise.  The heatmaps come out peaked on known cells for the
visible keypoints -- what a trained network hands to HRNetPredictionTransform / CameraCreator -- and every arithmetic error of a
fast engine inside the backbone acts on the decoded keypoints.  Every convolution runs at its full size on dense data.
Four distinct batches of such frames (--frame-sets) are resident in HBM and cycled step by step, so the steps solve different keypoint
sets (`config.frame_sets.by_set`: solve time and cameras found per batch -- one of the four holds a frame whose fits crawl to the
reference's 20000 iterations, the others solve in under 2 ms); with the driver's --warmup 5 --steps 20 the last step is batch 0, the
batch rounds 2-5 ran every step, so `drain_ms` and the parity fields stay comparable.

Engine.  The benchmarked engine is `fp16x3` (default, and load_model's default): fp32 tensors and fp32 accumulation, every product
formed on the 16-bit matrix pipe from split operands (x = hi + lo, two fp16; hi.hi + hi.lo + lo.hi: the dropped term is ~2^-22 of a
product) -- the fastest engine of the build that returns the fp32 engine's keypoint indices (the reference's predict() is fp32,
metamodel.py:127-134; north_star asks for bit-identical indices): all usable keypoints of the benchmarked frames (`parity` on the
line) and 39 642 of 39 642 on 2048 frames with 2045 of 2045 cameras identical (tools/parity_large.py, NOTES/design_history_r1_r5.md §10).  `roofline.peak`
is the 16-bit dense MFMA peak / 3 (three executed products per reference product); `roofline.frac_of_16bit_dense_peak` prices the
same reference-formulation work against the undivided hardware peak.  The bf16 throughput engine (2.6x faster, 1-3 % of the usable
keypoints move by one cell on this workload) and the exact-fp32 engine are timed on the same frames outside the timed region and
ride on the line as `bf16` and `fp32`.

Parity of the benchmarked path rides on the same line (`parity`, computed OUTSIDE the timed region on the same frames):
the benchmarked engine against this build's exact-fp32 engine (the one pinned to the reference goldens by
tests/test_hrnet_gpu.py): keypoint-index agreement, and the relative difference of the solved cameras' reprojection
error; plus the fp32 engine's own frames/s.  With N > 1 every rank processes its own 64 frames per step (weak scaling, frames
are independent); the per-frame records stay on the rank and ONE RCCL all_gather of all K steps' records closes the timed
region (north_star: "a single RCCL gather over xGMI at the end").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# The step uses the network's stream and three solve streams (+ RCCL's internal one, active only for the one gather after the
# last step).  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that land on one queue run
# in submission order (measured in round 2 with a per-step gather: RCCL's stream shared a queue with the network stream and the
# step went 43.1 -> 54.5 ms).  With 8 queues no two of them alias.  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# conv MACs of the reference's direct formulation x2
FLOP_KEYPOINT_NET = 2 * 253910384640
FLOP_LINE_NET = 2 * 185690000000
FLOP_KEYPOINT_NET_1080P = 2 * 1014180000000
PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3, 'f32': 157.3, 'fp8': 5000.0, 'fp16x3': 2500.0}   # dense MFMA peaks (fp8: block-scaled K=64/128 forms), MI355X_MICROARCH.md
# fp16x3 executes three 16-bit products per reference product: `roofline.frac` prices the ALGORITHMIC (reference-formulation) FLOPs against
# the undivided 16-bit dense peak -- useful work per chip; `frac_of_split_ceiling` = the same over peak / 3 -- how close the kernel is to
# the most this arithmetic could ever deliver (= issue-slot use of the matrix pipe)
SPLIT_PRODUCTS = {'fp16x3': 3.0}
BATCH = 64
# make_submit.py:45-50.  refine_camera's LM runs under the reference's own criterion (camera.py:116: 20000 iterations, 1e-5) -- the
# library default; rounds 1-4 capped it at 200 here because one crawling fit keeps ONE wavefront busy for up to ~600 ms and the single
# side stream joined every batch on its slowest frame.  The pipeline now solves up to four batches side by side (pipeline.py) and the
# persistent kernels take their work from tickets, so a crawl costs the CU it sits on and nothing else.
# SNCAL_BENCH_REFINE_CAP=<n> (diagnosis only; reported on the line as a non-reference setting) restores a cap.
REFINE_CAP = int(os.environ.get('SNCAL_BENCH_REFINE_CAP', '0'))
SOLVER_KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter', lines_file=None,
                 max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0, min_points_per_plane=6,
                 min_points_for_refinement=6, reliable_thresh=57)
if REFINE_CAP > 0:
    SOLVER_KW['refine_max_iters'] = REFINE_CAP


def seeded_weights(cfg, seed):
    """Random-init weights of the architecture (no checkpoints ship with the reference).  Same recipe as the
    test-suite generator, restated here so that the timed path never imports oracle/."""
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


# ---- CPU baseline: the oracle on this host's cores, bounded sample --------------------------------
def usable_cores():
    """Cores this process may really use: physical cores, capped by the affinity mask and by the cgroup CPU quota (the GPU
    box shows 128 physical cores and a quota of 16: 128 torch threads on it run 4x SLOWER than 16)."""
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except ImportError:
        phys = os.cpu_count() or 1
    n, why = int(phys), f'{phys} physical'
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, why = aff, f'{phys} physical, affinity {aff}'
    except AttributeError:
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max' and int(q) // int(per) >= 1 and int(q) // int(per) < n:
            n, why = int(q) // int(per), f'{phys} physical, cgroup quota {int(q) // int(per)}'
    except (OSError, ValueError):
        pass
    return max(1, n), why


def cpu_baseline(sd, cfg_name, frames, kpts, budget_s=45.0, nb=8):
    """torch-CPU fp32 forward on the cores this process may use (usable_cores), at batch 1 (warm-up 2, median of up to 5) and at
    batch `nb` (warm-up 1, median of up to 3) - the faster of the two counts; numpy decode; numpy camera solve single-process
    and on a process pool of that many workers (the reference uses 16 workers,
    make_submit.py:25).  `frames` (>=8,3,540,960) CPU tensor, `kpts` decoded keypoints (n,57,3) to solve."""
    from oracle import decode as od
    from oracle import hrnet_ref as hr
    cores, why = usable_cores()
    cfg = hr.load_config(cfg_name)
    torch.set_num_threads(cores)

    def timed(x, warm, runs_max, t_budget):
        t_start = time.perf_counter()
        with torch.no_grad():
            for _ in range(warm):                                    # warm-up (oneDNN primitive creation, thread pool)
                out = hr.forward(sd, x, cfg)
            runs = []
            while len(runs) < runs_max and (len(runs) < 2 or time.perf_counter() - t_start < t_budget):
                t0 = time.perf_counter()
                out = hr.forward(sd, x, cfg)
                runs.append(time.perf_counter() - t0)
        return float(np.median(runs)) / x.shape[0], len(runs), out
    t1, n1, logp = timed(frames[:1].contiguous(), 2, 5, budget_s * 0.4)         # batch 1: warm-up 2 + median of 5
    tb, nbr, logpb = (timed(frames[:nb].contiguous(), 1, 3, budget_s * 0.6) if nb > 1 else (t1, n1, logp))
    t_net = min(t1, tb)
    x = frames[:nb]
    net_txt = f'batch 1: {t1:.3f} s/frame (median of {n1}), batch {nb}: {tb:.3f} s/frame (median of {nbr})'
    logp = logpb
    lp = logp.numpy()
    t0 = time.perf_counter()
    od.keypoint_decode(lp, (540, 960))
    t_dec = (time.perf_counter() - t0) / x.shape[0]
    # camera solve: N independent worker processes (plain scripts: they never import torch or touch the GPU runtime), every
    # worker solves `per` frames after one untimed call; pool rate = frames / slowest worker's solve time
    import subprocess
    import tempfile
    kp = np.asarray(kpts, dtype=np.float32)
    worker = os.path.join(ROOT, 'oracle', 'solve_worker.py')
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'kpts.npy')
        np.save(path, kp)
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')

        def run_pool(n, per, limit_s=180.0):
            # the oracle runs refine_camera under the same criterion as the GPU leg (the reference's 20000 iterations unless the
            # diagnosis cap is set); a crawling fit costs the pure-numpy port minutes, so the pool is bounded: workers still running
            # at the limit are stopped (their own PIDs) and the rate is quoted on the frames of the workers that finished
            procs = [subprocess.Popen([sys.executable, worker, path, str(i * per), str(per)] + ([str(REFINE_CAP)] if REFINE_CAP > 0 else []), stdout=subprocess.PIPE, env=env)
                     for i in range(n)]
            t_end = time.perf_counter() + limit_s
            outs = []
            for p in procs:
                try:
                    outs.append(p.communicate(timeout=max(0.1, t_end - time.perf_counter()))[0].decode().split())
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.communicate()
                    outs.append(None)
            done = [o for o in outs if o]
            if not done:
                raise RuntimeError('no solve worker finished inside the limit')
            return [float(o[0]) for o in done], sum(int(o[1]) for o in done), len(done)
        t1, _, _ = run_pool(1, 4)
        t_solve1 = t1[0] / 4
        workers = max(1, min(cores, 64))
        per = 3
        try:
            tw, found, nd = run_pool(workers, per)
            fps_pool = nd * per / max(tw)
            pool_txt = (f'{workers}-process pool {fps_pool:.0f} frames/s ({found}/{nd * per} cameras'
                        + (f'; {workers - nd} workers stopped at the limit' if nd < workers else '') + ')')
        except Exception as e:                                       # a host that cannot start the workers: single-process figure
            fps_pool = 1.0 / t_solve1
            pool_txt = f'pool unavailable ({type(e).__name__})'
    stages = {'forward': 1.0 / t_net, 'decode': 1.0 / t_dec, 'solve': fps_pool}
    return {'value': round(min(stages.values()), 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle (CPU restatement) on {cores} cores ({why}): HRNet-W48 {frames.shape[3]}x{frames.shape[2]} fp32 torch-CPU forward, '
                      f'{net_txt} -> {1 / t_net:.2f} frames/s; numpy decode '
                      f'{t_dec * 1e3:.0f} ms/frame; numpy camera solve on the decoded keypoints {t_solve1 * 1e3:.0f} ms/frame single '
                      f'process, {pool_txt}; value = slowest stage of the pipelined three (BASELINE.md 4.5)',
            'stages_fps': {k: round(v, 3) for k, v in stages.items()},
            'serial_fps': round(1.0 / (t_net + t_dec + 1.0 / fps_pool), 4)}


def engine_leg(sncal_amd, cfg_name, sd, x, cc, dev, dtype, peak, what, steps=3, flop_frame=None):
    """One more engine on the SAME frames and the SAME step (forward + fused decode + solve of the decoded keypoints), outside the
    timed region of the main leg, measured like it: warm-up, one step profiled launch by launch to find the dominant kernel, then
    `steps` timed steps with HIP events on that kernel's launches only.  Returns (object for the line, keypoints, camera records)."""
    net = sncal_amd.HRNetHeatmap(cfg_name, dtype=dtype, device=dev)
    net.load_state_dict(sd)
    pipe = sncal_amd.CalibrationPipeline(net, cc, decode_size=(540, 960))
    out = pipe.submit(x)
    pipe.join()
    torch.cuda.synchronize()
    net.set_profiling(1)
    pipe.submit(x)
    pipe.join()
    torch.cuda.synchronize()
    warm = net.get_profile()
    net.set_profiling(2)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = pipe.submit(x)
    pipe.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom = net.get_profile()
    net.set_profiling(0)
    fps = steps * x.shape[0] / dt
    obj = {'value': round(fps, 2), 'unit': 'frames/s', 'ms_per_step': round(dt / steps * 1e3, 2), 'steps': steps, 'frames': int(x.shape[0]),
           'dtype': dtype, 'what': what,
           'timed_region': 'as the headline: `steps` steps + the drain of the last batch\'s solve at the reference\'s refine criterion (one crawling fit: ~60 ms per run)'}
    if len(dom) == 1 and dom[0]['ms'] > 0:
        d = dom[0]
        tot = sum(p['ms'] for p in warm)
        wd = next((p for p in warm if p['kernel'] == d['kernel']), None)
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        obj['roofline'] = {'bound': 'mfma', 'kernel': d['kernel'], 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': round(ach / peak, 4), 'launches': d['launches'],
                           **({'frac_of_split_ceiling': round(ach / peak * SPLIT_PRODUCTS[dtype], 4)} if dtype in SPLIT_PRODUCTS else {}),
                           'avg_launch_us': round(d['ms'] * 1e3 / d['launches'], 2),
                           'share_of_gpu_time': round(wd['ms'] / tot, 4) if wd and tot else None}
        if flop_frame:
            obj['network_tflops_reference_formulation'] = round(fps * flop_frame / 1e12, 1)
            obj['network_frac_of_peak'] = round(fps * flop_frame / 1e12 / peak, 4)
    kp = out[0].cpu().numpy()
    recs = cc.records(out[1])
    del pipe, net
    return obj, kp, recs


def lanes_leg(sncal_amd, cfg_name, sd, x, cc, dev, dtype, lanes=2, steps=5, warm=2):
    """The same step with the batch cut into `lanes` independent sub-batches, each with its own engine, workspace and stream (a
    deployment option of the pipeline, bench.py --lanes): kernels of one lane fill the tails and the under-occupied launches of the
    other.  Measured outside the timed region of the main leg; per-kernel durations then describe a shared GPU, which is why the
    driver line itself (and its roofline object) stays at one lane."""
    B = x.shape[0] // lanes * lanes
    bl = B // lanes
    nets = []
    for _ in range(lanes):
        n = sncal_amd.HRNetHeatmap(cfg_name, dtype=dtype, device=dev)
        n.load_state_dict(sd)
        nets.append(n)
    pipes = [sncal_amd.CalibrationPipeline(n, cc, decode_size=(540, 960)) for n in nets]
    streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]

    def step():
        for i in range(lanes):
            with torch.cuda.stream(streams[i]):
                pipes[i].submit(x[i * bl:(i + 1) * bl])

    def fence():
        for i in range(lanes):
            with torch.cuda.stream(streams[i]):
                pipes[i].join()
        torch.cuda.synchronize()

    for _ in range(warm):
        step()
    fence()
    marks = []
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(streams[0]):
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        step()
    fence()
    dt = time.perf_counter() - t0
    steady = marks[0].elapsed_time(marks[-1]) / (len(marks) - 1) if len(marks) > 1 else None
    del pipes, nets
    return {'value': round(steps * B / dt, 2), 'unit': 'frames/s', 'ms_per_step': round(dt / steps * 1e3, 2), 'lanes': lanes, 'frames': int(B),
            'steps': steps, 'dtype': dtype,
            'steady_state_ms_per_step': round(steady, 3) if steady else None,
            'steady_state_frames_per_s': round(B / steady * 1e3, 1) if steady else None,
            'drain_ms': round(dt * 1e3 - steady * steps, 1) if steady else None,
            'what': f'the same step with the {B} frames as {lanes} independent sub-batches on their own streams (bench.py --lanes {lanes}); '
                    'not the driver line: per-kernel durations of a shared GPU do not make a roofline'}


def input_stage_leg(sncal_amd, net, cc, dev, steps=6, B=BATCH):
    """N3 (make_submit.py:56-67: cv2.imread + ToTensor on the host): frames/s of the JPEG input stage on this box -- JpegDecoder.decode
    alone (Huffman decode on `threads` host threads, dequantise + IDCT + colour conversion on the GPU) and the whole pipeline fed from JPEG
    bytes (the host decodes batch k + 1 while the GPU runs batch k).  The frame is the 960x540 4:2:0 test frame of tests/golden
    (libjpeg-turbo byte-equal); its pixels mean nothing to the network, so the solve finds no cameras: an input-stage figure, outside the
    timed region, not a second bench line."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'jpeg_cases.npz'))
    blob = g['jpg.full'].tobytes()
    cores, why = usable_cores()
    threads = max(1, min(cores, 32))
    dec = sncal_amd.JpegDecoder(540, 960, max_batch=B, threads=threads, device=dev)
    bufs = [torch.empty((B, 540, 960, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    blobs = [blob] * B
    for k in range(2):
        dec.decode(blobs, bufs[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        dec.decode(blobs, bufs[k & 1])
    torch.cuda.synchronize()
    t_dec = (time.perf_counter() - t0) / steps
    pipe = sncal_amd.CalibrationPipeline(net, cc, decode_size=(540, 960))
    for k in range(2):
        pipe.submit(dec.decode(blobs, bufs[k & 1]))
    pipe.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        pipe.submit(dec.decode(blobs, bufs[k & 1]))
    pipe.join()
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / steps
    dec.close()
    return {'jpeg_bytes_per_frame': len(blob), 'frame': '960x540 4:2:0 baseline JPEG (tests/golden/jpeg_cases.npz jpg.full)', 'host_threads': threads, 'cores': why,
            'decode_alone_frames_per_s': round(B / t_dec, 1), 'decode_alone_ms_per_batch': round(t_dec * 1e3, 2),
            'pipeline_from_jpeg_frames_per_s': round(B / t_pipe, 1), 'pipeline_from_jpeg_ms_per_step': round(t_pipe * 1e3, 2), 'steps': steps,
            'what': 'JpegDecoder.decode alone, and decode + forward + decode + solve with the host Huffman stage of batch k + 1 beside the GPU work of batch k '
                    '(Huffman on the host threads this process may use; the rest on the GPU)'}


def parity_of(kp32, r32, kpf, rf, versus):
    """Keypoints / cameras of one engine against the exact-fp32 engine's on the same frames (all frames with two cameras)."""
    same = (kp32[..., :2] == kpf[..., :2]).all(-1)                       # (B,57) identical (x, y) indices
    usable = kp32[..., 2] >= 0.2                                         # rows any of the solver's thresholds can take
    move = np.abs(kp32[..., :2] - kpf[..., :2]).max(-1)
    both = [i for i in range(len(r32)) if r32[i].status != 0 and rf[i].status != 0]
    deltas = [abs(rf[i].rmse - r32[i].rmse) / r32[i].rmse for i in both if r32[i].rmse > 0]
    return {'vs': versus,
            'workload': 'deep path: the keypoint codes travel through every backbone tensor (synth.deep_state_dict)',
            'frames': int(kp32.shape[0]),
            'index_agreement': round(float(same[usable].mean()) if usable.any() else 1.0, 6),
            'usable_keypoints': int(usable.sum()),
            'moved_usable_keypoints': int((~same[usable]).sum()), 'moved_usable_max_px': float(move[usable].max()) if usable.any() else 0.0,
            'index_agreement_all_rows': round(float(same.mean()), 6),
            'conf_abs_delta_max_usable': round(float(np.abs(kp32[..., 2] - kpf[..., 2])[usable].max()) if usable.any() else 0.0, 6),
            # one-sided?  signed mean of (engine - fp32) confidence on the usable rows, and how many rows cross one of the solver's
            # thresholds (0.5 / 0.35 / 0.2, prediction.py:245-257) in each direction
            'conf_signed_delta_mean_usable': float(f'{float((kpf[..., 2] - kp32[..., 2])[usable].mean()):.3e}') if usable.any() else 0.0,
            'threshold_crossings': {'up': int(sum(((kp32[..., 2] < th) & (kpf[..., 2] >= th)).sum() for th in (0.5, 0.35, 0.2))),
                                    'down': int(sum(((kp32[..., 2] >= th) & (kpf[..., 2] < th)).sum() for th in (0.5, 0.35, 0.2)))},
            'cameras_both': len(both), 'cameras_fp32': sum(r.status != 0 for r in r32), 'cameras_engine': sum(r.status != 0 for r in rf),
            'rmse_rel_delta_max': float(f'{max(deltas):.3e}') if deltas else None,
            'rmse_rel_delta_median': float(f'{float(np.median(deltas)):.3e}') if deltas else None,
            'frames_rmse_rel_delta_le_1e-4': int(sum(d <= 1e-4 for d in deltas)),
            'solve_parity': 'vs the build\'s own oracle only: OpenCV parity unpinned (cv2 not installable offline)'}


def parity_leg(sncal_amd, cfg_name, sd, x, cc, kp_fast, rec_fast, dev, steps=3, flop_frame=None, main_dtype='bf16', steps_fp32=None):
    """(parity of the benchmarked engine, fp32 object, fp16x3 object).  fp32 = the reference's own arithmetic (HRNetMetaModel.predict
    is fp32, metamodel.py:127-134) on the exact-fp32 MFMA engine, load_model's default; fp16x3 = the fp32-class engine (split-fp16
    3x3 convolutions), with its own parity against fp32."""
    VS = ('exact-fp32 engine of this build (pinned to the reference goldens by tests/test_hrnet_gpu.py, kernel by kernel to torch fp32 by '
          'tests/test_kernels_gpu.py)')
    fp32, kp32, r32 = engine_leg(sncal_amd, cfg_name, sd, x, cc, dev, 'fp32', PEAK_TFLOPS['fp32'],
                                 'the same step on the exact-fp32 MFMA engine (v_mfma_f32_16x16x4_f32): the reference\'s own arithmetic (load_model(dtype=\'fp32\'))',
                                 steps_fp32 or steps, flop_frame)
    rf = cc.records(cc.solve_device(kp_fast))      # both keypoint sets through the same solve call (no line points): like for like in every workload
    parity = parity_of(kp32, r32, kp_fast.cpu().numpy(), rf, VS)
    # the other fast engine of the build on the same frames, with its own parity: fp16x3 when bf16 / fp8 is benchmarked, bf16 when fp16x3 is
    WHAT = {'fp16x3': 'the same step on the fp32-class engine: fp32 tensors and accumulation, every convolution and the head in split-fp16 arithmetic '
                      '(x = hi + lo fp16; hi.hi + hi.lo + lo.hi on the 16-bit matrix pipe; frac against the undivided 16-bit dense peak, frac_of_split_ceiling against peak / 3)',
            'bf16': 'the same step on the bf16 throughput engine (bf16 tensors, fp32 accumulation): opt-in, NOT index-identical to fp32 -- see its parity'}
    other = 'bf16' if main_dtype == 'fp16x3' else 'fp16x3'
    x3, kp3, r3 = engine_leg(sncal_amd, cfg_name, sd, x, cc, dev, other, PEAK_TFLOPS[other], WHAT[other], steps, flop_frame)
    x3['parity'] = parity_of(kp32, r32, kp3, r3, VS)
    return parity, fp32, other, x3


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-exec under torch.distributed.run with one rank per
    GPU (frames sharded over the GPUs of one node, ONE RCCL all_gather after the last step), so that the command the driver
    runs verbatim measures N GPUs.  Fails loudly when fewer than N devices are visible."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    if not args.dry_dist:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f'--gpus {args.gpus}: only {n_dev} GPU(s) visible; refusing to report a smaller job under that label')
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_dist(rank, world):
    """--dry-dist: the multi-rank skeleton of a run on CPU (gloo) -- rendezvous, two steps of rank-coded records logged on the rank,
    the path's one collective at the end (sncal_amd.dist.RecordLog, the object the pipeline uses), barrier, max-over-ranks -- without
    any GPU work."""
    import torch.distributed as dist
    from sncal_amd.dist import RecordLog, pack_records
    if world > 1:
        dist.init_process_group('gloo')
    per = 2
    log = RecordLog()
    t0 = time.perf_counter()
    for step in range(2):
        kp = torch.full((per, 57, 3), float(rank), dtype=torch.float32)
        rec = torch.full((per, 136), rank, dtype=torch.uint8)
        log.add(pack_records(kp, rec))
    allrec = log.gather()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    seen = sorted(set(int(v) for v in allrec[:, -1].tolist()))
    if rank == 0:
        print(json.dumps({'metric': 'dry-dist (launcher / collective self-test, not a measurement)', 'value': None, 'n_gpus': world,
                          'gathered_ranks': seen, 'records': int(allrec.shape[0]), 'record_bytes': int(allrec.shape[1]),
                          'backend': 'gloo'}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='fp16x3', choices=['bf16', 'fp32', 'fp8', 'fp16x3'],
                    help="fp8: e4m3 arithmetic in the wide 3x3 convolutions (BASELINE config C5; calibrated on the first frames), the rest bf16")
    ap.add_argument('--size', default='540p', choices=['540p', '1080p'], help='input frames 960x540 (the metric) or 1920x1080 (config C5)')
    ap.add_argument('--fp8-layers', default='all', help="layer selection of --dtype fp8 (sncal_hrnet_set_fp8_layers)")
    ap.add_argument('--workload', default='c3', choices=['c3', 'c4'],
                    help='c3: HRNet-W48 keypoint net + decode + solve (the metric\'s configuration); c4: + the W48 line net, '
                         'its two-peak decode and the device line join in front of the solve')
    ap.add_argument('--line-workload', default='designed', choices=['designed', 'random'],
                    help="--workload c4: 'designed' = the line network carries the same deep signal path as the keypoint network and answers with the "
                         "lines through the stamped keypoints (synth.line_deep_state_dict), joined at get_line_data's own prob_thre 0.2: line points "
                         "consistent with the frame, as a trained line network gives; 'random' = the raw random-init line network at the export "
                         "CLI's prob_thre 0 (rounds 2-5): 28 garbage line points per frame, a stress case for the solve (most fits crawl)")
    ap.add_argument('--frame-sets', type=int, default=0,
                    help='distinct batches of synthetic frames resident in HBM, cycled step by step (default: 4 at 540p, 1 at 1080p): the '
                         'steps then solve different keypoint sets -- the solve time and the cameras found are not one sample')
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--lanes', type=int, default=1,
                    help='split the batch into this many independent sub-batches on their own streams (default 1; see DESIGN.md §5: '
                         '2 lanes fill kernel tails and launch gaps, but per-kernel HIP-event durations then measure a shared GPU, '
                         'so the roofline object is only meaningful at 1)')
    ap.add_argument('--dump-gather', default=None,
                    help='testing aid (tests/test_dist_gpu.py): rank 0 saves its own packed records and the gathered tensor of the last step to this .npz')
    ap.add_argument('--dry-dist', action='store_true',
                    help='launcher / collective self-test on CPU (gloo): the N ranks exchange rank-coded records through the '
                         'path\'s single all_gather and rank 0 prints n_gpus and the ranks it saw; no GPU work, not a bench line')
    args = ap.parse_args()

    self_launch(args)                                   # `python bench.py --gpus N`, N > 1, outside torchrun: start the N ranks
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the rank count of the launcher and --gpus must agree')
    if args.dry_dist:
        return dry_dist(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    if torch.cuda.device_count() < world or local >= torch.cuda.device_count():
        raise SystemExit(f'--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU, no oversubscription)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # everything below runs on a non-blocking stream of the bench's own, never on the legacy null stream: the pipeline's CU-masked
    # solve streams are blocking streams (they synchronise with the null stream), see sncal_amd/pipeline.py
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    # SNCAL_BENCH_FORCE_DIST=1 (testing aid): take the multi-GPU code path -- RCCL init, the per-step all_gather on the
    # side stream, barrier, max-over-ranks -- even with one rank, so that it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get('SNCAL_BENCH_FORCE_DIST') == '1'
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    import sncal_amd
    cfg_name = 'hrnet_w48'
    sd = sncal_amd.synth.peaked_state_dict(seeded_weights(cfg_name, seed=1), deep=True)
    B = args.batch
    L = max(1, args.lanes)
    if B % L:
        raise SystemExit(f'--batch {B} is not a multiple of --lanes {L}')
    c4 = args.workload == 'c4'
    sd_line = seeded_weights('line_hrnet_w48', seed=2) if c4 else None
    if c4 and args.line_workload == 'designed':
        sd_line = sncal_amd.synth.line_deep_state_dict(sd_line)
    line_prob_thre = 0.2 if args.line_workload == 'designed' else 0.0
    nets, lnets = [], []
    for _ in range(L):
        net = sncal_amd.HRNetHeatmap(cfg_name, dtype=args.dtype, device=dev)
        net.load_state_dict(sd)
        nets.append(net)
        if c4:
            ln = sncal_amd.HRNetHeatmap('line_hrnet_w48', dtype=args.dtype, device=dev)
            ln.load_state_dict(sd_line)
            lnets.append(ln)
    HW = (540, 960) if args.size == '540p' else (1080, 1920)
    n_distinct = B if args.size == '540p' else min(B, 16)        # 1080p: 16 distinct frames, repeated (host memory / start-up time)
    # several distinct batches, all resident in HBM before the timed region, cycled step by step (set 0 = the frames of rounds 2-5)
    n_sets = args.frame_sets if args.frame_sets > 0 else (4 if args.size == '540p' else 1)
    reps = (B + n_distinct - 1) // n_distinct
    F_sets, X_sets, E_sets = [], [], []
    for si in range(n_sets):
        f_cpu, e = sncal_amd.synth.stamped_frames(n_distinct, seed=1000 + rank + 7919 * si, size=HW)     # synthetic frames, then resident in HBM
        f_cpu = torch.from_numpy(f_cpu)
        F_sets.append(f_cpu)
        X_sets.append(f_cpu.to(dev).repeat(reps, 1, 1, 1)[:B].contiguous())
        E_sets.append(np.tile(e, (reps, 1, 1))[:B])
    frames_cpu, x, expect = F_sets[0], X_sets[0], E_sets[0]
    cur = {'n': 0, 'last': 0}
    if args.dtype == 'fp8':
        for n in nets:
            n.calibrate_fp8(x[:4])
            n.set_fp8_layers(args.fp8_layers)
    cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **SOLVER_KW)
    pipes = [sncal_amd.CalibrationPipeline(nets[i], cc, decode_size=(540, 960), line_net=lnets[i] if c4 else None, line_prob_thre=line_prob_thre)
             for i in range(L)]
    lane_streams = [None] if L == 1 else [torch.cuda.Stream(device=dev) for _ in range(L)]
    bl = B // L
    XS_sets = [[X[i * bl:(i + 1) * bl] for i in range(L)] for X in X_sets]
    xs = XS_sets[0]
    last = {}
    diag = os.environ.get('SNCAL_BENCH_DIAG')
    diag_nosolve = diag == 'nosolve'
    diag_noprof = diag == 'noprof'

    step_marks = []                                      # one event per step at the head of its forward: the steady-state step time

    def step():
        si = cur['n'] % n_sets                            # this step's batch of frames
        cur['n'] += 1
        cur['last'] = si
        xs = XS_sets[si]
        if L == 1 and not diag_nosolve:
            step_marks.append(torch.cuda.Event(enable_timing=True))
            step_marks[-1].record()
        # forward + decode on the main stream; the solve of THESE keypoints on the pipeline's side stream (it overlaps the
        # next step's convolutions); every solve is complete before the closing fence of the timed region.
        # multi-GPU: the step only LOGS its packed records on the rank (pipeline.log); the single collective of the path
        # (all steps' records to every rank, RCCL over xGMI) is issued once, behind the last step (close_run)
        for i in range(L):
            if diag_nosolve:                             # diagnosis only: network + decode, no solves (not a valid bench line)
                nets[i].forward(xs[i], want_heat=False, decode_size=(540, 960))
                continue
            if lane_streams[i] is None:
                out = pipes[i].submit(xs[i], gather=use_dist)
            else:
                with torch.cuda.stream(lane_streams[i]):
                    out = pipes[i].submit(xs[i], gather=use_dist)
            last[i] = out

    def fence():
        for i in range(L):
            if lane_streams[i] is None:
                pipes[i].join()
                if use_dist and len(pipes[i].log):
                    last['gathered', i] = (pipes[i].log.local(), pipes[i].gather_all())     # the path's ONE collective: every step's records
            else:
                with torch.cuda.stream(lane_streams[i]):
                    pipes[i].join()
                    if use_dist and len(pipes[i].log):
                        last['gathered', i] = (pipes[i].log.local(), pipes[i].gather_all())
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
    step_marks.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # everything below describes the LAST step's batch of frames (its keypoints and records are what `last` holds)
    frames_cpu, x, expect, xs = F_sets[cur['last']], X_sets[cur['last']], E_sets[cur['last']], XS_sets[cur['last']]
    # steady state: head of step 0's forward to head of the last step's forward; what is left of dt is the pipeline's drain -- the
    # solve of the LAST batch has nothing to overlap with inside a timed region that must end with every solve complete
    steady_ms = step_marks[0].elapsed_time(step_marks[-1]) / (len(step_marks) - 1) if len(step_marks) > 1 else None
    merged = {}
    for n in nets:
        for q in n.get_profile():
            m = merged.setdefault(q['kernel'], dict(q, ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ('ms', 'launches', 'flops', 'bytes'):
                m[k] += q[k]
        n.set_profiling(False)
    prof = list(merged.values())
    if args.dump_gather and use_dist and rank == 0:      # the path's one collective, as it closed the timed region: K steps' records at once
        from sncal_amd.dist import pack_records
        np.savez(args.dump_gather, local=last['gathered', 0][0].cpu().numpy(), gathered=last['gathered', 0][1].cpu().numpy(),
                 last_step=pack_records(last[0][0], last[0][1]).cpu().numpy(), world=world, rank=rank, steps=args.steps)
    if diag:                                             # diagnosis runs print the step time only
        print('diag', diag, round(dt / args.steps * 1e3, 3), 'ms/step', 'steady', round(steady_ms, 3) if steady_ms else None)
        return
    # solve-stage time on the decoded keypoints, measured separately after the timed region (torch events see torch's
    # current stream, which is the stream libsncal launches on)
    kp_fast = torch.cat([last[i][0] for i in range(L)], dim=0)
    rec_fast = torch.cat([last[i][1] for i in range(L)], dim=0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    rec_tmp = cc.solve_device(kp_fast)
    ev[0].record()
    for _ in range(3):
        cc.solve_device(kp_fast, out=rec_tmp)
    ev[1].record()
    torch.cuda.synchronize()
    solve_ms = ev[0].elapsed_time(ev[1]) / 3
    # ... and of every batch of frames the steps cycled through (one forward + one synchronous solve each, outside the timed region): the
    # solve's latency is set by its slowest Levenberg-Marquardt fit, which differs from batch to batch
    by_set = None
    if rank == 0 and L == 1 and n_sets > 1 and not c4:
        by_set = []
        for si in range(n_sets):
            _, kps = nets[0].forward(XS_sets[si][0], want_heat=False, decode_size=(540, 960))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rs = cc.solve_device(kps)
            e1.record()
            torch.cuda.synchronize()
            by_set.append({'solve_ms': round(e0.elapsed_time(e1), 1), 'cameras_found': sum(1 for r in cc.records(rs) if r.status != 0)})
    # what the solves cost the step: the same K steps of network + decode alone (no solve, no gather), outside the timed region
    solver_note = None
    if rank == 0 and L == 1 and not diag:
        for n in nets:
            n.set_profiling(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            nets[0].forward(xs[0], want_heat=False, decode_size=(540, 960))
        torch.cuda.synchronize()
        ns_ms = (time.perf_counter() - t1) / args.steps * 1e3
        solver_note = {'refine_max_iters': REFINE_CAP if REFINE_CAP > 0 else 20000, 'refine_eps': 1e-5,
                       'criterion': ('DIAGNOSIS: SNCAL_BENCH_REFINE_CAP set, not the reference criterion' if REFINE_CAP > 0 else
                                     'the reference\'s own (baseline/camera.py:116 solvePnPRefineLM criteria (20000, 1e-5)) = library default'),
                       'solve_streams': pipes[0].max_in_flight // 2, 'solve_streams_cu_masked': bool(pipes[0].masked),
                       'solve_cus_per_xcd': sncal_amd.pipeline.SOLVE_CUS_PER_XCD if pipes[0].masked else None,
                       'nosolve_ms_per_step': round(ns_ms, 3),
                       'step_over_nosolve': round(dt / args.steps * 1e3 / ns_ms, 4),
                       'steady_state_ms_per_step': round(steady_ms, 3) if steady_ms else None,
                       'steady_state_over_nosolve': round(steady_ms / ns_ms, 4) if steady_ms else None,
                       'drain_ms': round(dt * 1e3 - steady_ms * args.steps, 1) if steady_ms else None,
                       'note': 'ms_per_step = (K steps + drain) / K: the timed region ends with every solve complete, so the last batch\'s solve '
                               '(one crawling Levenberg-Marquardt fit of 20000 iterations sets its latency) is exposed once per run; '
                               'steady_state = head of forward to head of forward inside the region'}
        if REFINE_CAP == 0:             # and what a 200-iteration cap (rounds 1-4) would change on these frames: solve time, cameras that move
            cc_cap = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **dict(SOLVER_KW, refine_max_iters=200))
            ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev2[0].record()
            rec_cap = cc_cap.solve_device(kp_fast)
            ev2[1].record()
            torch.cuda.synchronize()
            ra, rb = cc.records(rec_cap), cc.records(rec_tmp)
            both = [(a, b) for a, b in zip(ra, rb) if a.status != 0 and b.status != 0]
            solver_note['cap_200_comparison'] = {
                'solve_ms_per_batch_at_200': round(ev2[0].elapsed_time(ev2[1]), 1),
                'none_ness_changes': sum(1 for a, b in zip(ra, rb) if (a.status == 0) != (b.status == 0)),
                'cameras_rmse_rel_delta_gt_1e-4': sum(1 for a, b in both if abs(a.rmse - b.rmse) > 1e-4 * max(b.rmse, 1e-12))}

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    recs = cc.records(rec_fast)
    n_cam = sum(1 for r in recs if r.status != 0)
    kpf = kp_fast.cpu().numpy()
    vis = expect[..., 2] > 0
    conf_vis = kpf[..., 2][vis]
    near = np.abs(kpf[..., :2] - expect[..., :2]).max(-1) <= 8.0       # deep path: a peak sits next to a coarse-grid node (4 / 8 px grids)
    hit = float(near[vis & (kpf[..., 2] >= 0.2)].mean()) if (vis & (kpf[..., 2] >= 0.2)).any() else 0.0

    if rank == 0:
        # focus mode: every net timed only the variant that led ITS profile; with several nets / lanes they may differ -- the line reports the largest
        assert 1 <= len(prof) <= len(nets), [p['kernel'] for p in prof]
        dom = max(prof, key=lambda q: q['ms'])
        warm = list(warm.values())
        total_ms = sum(p['ms'] for p in warm)
        warm_dom = next(p for p in warm if p['kernel'] == dom['kernel'])
        ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        traffic = None      # HBM bytes per launch from a separate rocprofv3 --pmc pass (tools/pmc_pass.sh -> profiles/pmc_traffic.json)
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
                traffic = json.load(f).get(dom['kernel']) if args.size == '540p' else None      # the PMC pass ran the 960x540 shapes
        except OSError:
            pass
        peak = PEAK_TFLOPS['fp8' if 'fp8' in dom['kernel'] else 'bf16' if args.dtype == 'fp8' else args.dtype]
        flop_frame = (FLOP_KEYPOINT_NET if args.size == '540p' else FLOP_KEYPOINT_NET_1080P) + (FLOP_LINE_NET if c4 else 0)
        dtype_label = args.dtype + (f' (fp8 layers: {args.fp8_layers})' if args.dtype == 'fp8' else '')
        wl = (f'C5: HRNet-W48 1920x1080, batch {B} per GPU, {dtype_label}, heatmap 540x960 + decode + batched camera solve (iterative_voter)'
              if args.size == '1080p' else 'C4: HRNet-W48 keypoint net + HRNet-W48 line net, 960x540, batch 64 per GPU, decodes + device line join + batched camera solve (iterative_voter)'
                                     + (' [line network: designed signal path, line join at prob_thre 0.2]' if args.line_workload == 'designed' else ' [line network: raw random init, prob_thre 0: garbage line points, solve stress case]')
              if c4 else 'C3: HRNet-W48 960x540, batch 64 per GPU, heatmap + decode + batched camera solve (iterative_voter) on the decoded keypoints')
        out = {
            'metric': 'frames/sec (HRNet-W48 960x540 + PnP)', 'value': round(world * B * args.steps / dt, 2),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic (noise frames stamped with per-keypoint codes at the pitch template\'s projections through sampled cameras; '
                    'random-init HRNet-W48 with the deep signal path of synth.deep_state_dict: the codes travel through every backbone tensor -> peaked heatmaps; see bench.py docstring)',
            'config': {'workload': wl, 'frames_per_gpu': B, 'lanes': L,
                       'parallelism': f'frames sharded over {world} GPU(s), records stay on the rank, ONE all_gather of all {args.steps} steps\' records closes the timed region' if world > 1 else 'single GPU',
                       'solve_ms_per_batch': round(solve_ms, 3), 'cameras_found': f'{n_cam}/{B}', 'solver': solver_note,
                       'frame_sets': {'n': n_sets, 'what': 'distinct batches of frames resident in HBM, cycled step by step; the parity / hit-rate / cameras_found fields describe the last step\'s batch',
                                      'by_set': by_set},
                       'decoded_within_8px_of_stamp': round(hit, 4), 'visible_keypoint_conf_median': round(float(np.median(conf_vis)), 4),
                       'network_tflops_reference_formulation': round(world * B * args.steps / dt * flop_frame / 1e12, 1),
                       'kernel_time_share_last_warmup_step': {p['kernel']: round(p['ms'] / total_ms, 4) for p in sorted(warm, key=lambda q: -q['ms'])[:8]}},
            'roofline': {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(ach / peak, 4),
                         **({'frac_of_split_ceiling': round(ach / peak * SPLIT_PRODUCTS[args.dtype], 4),
                             'split_note': 'fp16x3: three executed 16-bit products per reference product; frac = algorithmic FLOPs / undivided 16-bit dense peak, '
                                           'frac_of_split_ceiling = the same / (peak / 3)'} if args.dtype in SPLIT_PRODUCTS and 'x3' in dom['kernel'] else {}),
                         'traffic': traffic, 'launches': dom['launches'],
                         'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
                         'flops_per_launch': round(dom['flops'] / dom['launches'], 0),
                         'share_of_gpu_time': round(warm_dom['ms'] / total_ms, 4)},
        }
        # the next kernels of the step (last warm-up step, every launch timed): fraction of the roof that bounds each + PMC traffic ratio
        kernels = {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
                pmc = json.load(f) if args.size == '540p' else {}
        except OSError:
            pmc = {}
        for p in sorted(warm, key=lambda q: -q['ms'])[:6]:
            if not p['ms'] or not p['launches']:
                continue
            k = p['kernel']
            tf = p['flops'] / (p['ms'] * 1e-3) / 1e12
            gbs = p['bytes'] / (p['ms'] * 1e-3) / 1e9
            row = {'share_of_gpu_time': round(p['ms'] / total_ms, 4), 'launches_per_step': p['launches'], 'avg_launch_us': round(p['ms'] * 1e3 / p['launches'], 1),
                   'tflops': round(tf, 1), 'frac_mfma': round(tf / (PEAK_TFLOPS['fp8'] if 'fp8' in k else PEAK_TFLOPS['f32' if 'f32' in k else 'bf16']), 4),
                   **({'frac_of_split_ceiling': round(tf / PEAK_TFLOPS['fp16x3'] * 3.0, 4)} if 'x3' in k else {}),
                   'algorithmic_gb_s': round(gbs, 0), 'frac_hbm': round(gbs / 8000.0, 4)}
            if k in pmc:
                row['pmc_bytes_per_launch'] = pmc[k]
                row['pmc_over_algorithmic'] = round(pmc[k] / (p['bytes'] / p['launches']), 3) if p['bytes'] else None
                row['pmc_gb_s'] = round(pmc[k] / (p['ms'] * 1e-3 / p['launches']) / 1e9, 0)
            kernels[k] = row
        out['kernels'] = kernels
        if world == 1 and not args.no_parity:
            for n in nets[1:] + lnets:
                n._ws = None
            npar = B if args.size == '540p' else min(B, 16)
            out['parity'], out['fp32'], other, out_other = parity_leg(sncal_amd, cfg_name, sd, x[:npar], cc, kp_fast[:npar], rec_fast[:npar], dev,
                                                                   flop_frame=FLOP_KEYPOINT_NET if args.size == '540p' else FLOP_KEYPOINT_NET_1080P,
                                                                   main_dtype=args.dtype, steps_fp32=args.steps if args.size == '540p' else None,
                                                                   steps=args.steps if args.size == '540p' else 3)
            out[other] = out_other
            if args.size == '540p' and not c4 and L == 1 and args.dtype != 'fp8':
                out['lanes2'] = lanes_leg(sncal_amd, cfg_name, sd, x, cc, dev, args.dtype, steps=args.steps)      # (the headline's step count: the same drain share)
        if world == 1 and not args.no_parity and args.size == '540p' and not c4 and args.dtype != 'fp8':
            try:
                out['input_stage'] = input_stage_leg(sncal_amd, nets[0], cc, dev)
            except Exception as e:                                   # (a box without the golden file: the line stays valid)
                out['input_stage'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, cfg_name, frames_cpu, kpf, nb=8 if args.size == '540p' else 2)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
