"""Frame pipeline: the production loop of /root/reference/src/utils/make_submit.py:42-75 on one GPU.

make_submit pushes each prediction row to a 16-process CPU pool; here the network runs on the caller's
stream and the batched camera solve runs on a side stream, ordered by events, so the solve of batch i
overlaps the convolutions of batch i+1 (the solve is latency-bound and only occupies 64 wavefronts).

Streams only overlap when they sit on different hardware queues: HIP maps streams onto GPU_MAX_HW_QUEUES (default
4) queues, and with the multi-GPU gather a third stream (RCCL's) joins; measured, it then aliases the network
stream and the step grows from 43 to 54 ms.  Export GPU_MAX_HW_QUEUES=8 before the process touches the GPU
(bench.py and submit.py do).

Round 5: a POOL of solve streams.  At the reference's refine criterion (camera.py:116: 20000 iterations, 1e-5) one
crawling Levenberg-Marquardt fit keeps a single wavefront busy for up to ~600 ms, and a launch ends with its slowest
wavefront; on ONE side stream the solve of batch k + 1 queued behind that launch and the whole pipeline ran at the
pace of the slowest fit.  Batch k now goes to solve stream k mod P (P = SOLVE_STREAMS, default 4,
shared by every pipeline of a device), so up to P batches are being solved while the network runs on; results are
still delivered in submission order (`join`, `cameras`, and the gather: collectives are issued in submission order
and run in that order on RCCL's own stream).  What a crawling wavefront still costs is the CU it sits on, which is
why the persistent convolution kernels take their work from tickets (csrc/conv_tt_body.inc, bblockx3.hip, bneckx3.hip).
"""
import os

import torch

from . import _lib

SOLVE_STREAMS = max(1, int(os.environ.get('SNCAL_SOLVE_STREAMS', '4')))
_POOLS = {}      # device index -> [streams, next]: one pool per device (lanes share it: the hardware queues are few)


def _solve_pool(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _POOLS:
        _POOLS[key] = [[torch.cuda.Stream(device=device) for _ in range(SOLVE_STREAMS)], 0]
    return _POOLS[key]


class CalibrationPipeline:
    def __init__(self, net, calibrator, decode_size=(540, 960), line_net=None, line_sigma=3.0, line_scale=4,
                 line_prob_thre=0.0):
        """line_net (optional, BASELINE config C4): the line model runs on the same frames; its two-peak decode
        (EHMPredictionTransform.mask_heat_points_gauss, sigma as export_line_result.py:147-149), the line equations
        and the 30 line-intersection keypoints (export_line_result.py:51-131 with its CLI defaults scale 4,
        prob_thre 0; prediction.py:105-124) stay on the device and feed the solver directly -- the reference goes
        through a pickle file between two scripts."""
        self.net = net
        self.calibrator = calibrator
        self.decode_size = decode_size
        self.line_net = line_net
        self.line_sigma, self.line_scale, self.line_prob_thre = float(line_sigma), float(line_scale), float(line_prob_thre)
        self.device = net.device
        self._pool = _solve_pool(self.device)
        self.max_in_flight = 2 * len(self._pool[0])
        self._pending = []

    def _next_solve_stream(self):
        streams, k = self._pool
        self._pool[1] = (k + 1) % len(streams)
        return streams[k]

    def submit(self, frames: torch.Tensor, names=None, extra_keypoints: torch.Tensor = None, gather: bool = False):
        """frames (B,3,H,W) fp32 (ToTensor's output) or (B,H,W,3) uint8 BGR (cv2.imread's / JpegDecoder.decode's
        output) on the GPU.  Enqueues forward+decode on the current stream and the solve(s)
        on the side stream; returns (kpts, records[, extra_records][, all_ranks_records]) device tensors
        (asynchronous).  gather=True (multi-GPU): the one collective of the path -- every rank's
        per-frame records to every rank -- is enqueued on the SIDE stream behind the solves, so the next batch's
        convolutions on the main stream never wait for it."""
        main = torch.cuda.current_stream(self.device)
        _, kpts = self.net.forward(frames, want_heat=False, decode_size=self.decode_size)
        if self.line_net is not None:
            from .lines import lines_to_points_device
            from .transforms import EHMPredictionTransform
            heat_l, _ = self.line_net.forward(frames, want_heat=True)
            peaks = EHMPredictionTransform.mask_heat_points_gauss(heat_l, sigma=self.line_sigma)
            d_lp = lines_to_points_device(peaks, scale=self.line_scale, prob_thre=self.line_prob_thre)
        else:
            lp = self.calibrator.line_points_array(names)
            d_lp = None
            if lp is not None:              # pinned staging buffer: the upload is a real asynchronous copy on `main`
                d_lp = torch.from_numpy(lp).pin_memory().to(self.device, non_blocking=True)
        # everything the solve reads (keypoints, line points) is produced on `main` ABOVE this record
        ready = torch.cuda.Event()
        ready.record(main)
        extra_ready = None
        if extra_keypoints is not None:     # produced by the caller, on whatever stream is current for them: order it too
            extra_ready = torch.cuda.Event()
            extra_ready.record(main)
        side = self._next_solve_stream()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            kpts.record_stream(side)
            if d_lp is not None:
                d_lp.record_stream(side)
            rec = self.calibrator.solve_device(kpts, d_lp)
            out = [kpts, rec]
            if extra_keypoints is not None:
                side.wait_event(extra_ready)
                extra_keypoints.record_stream(side)
                out.append(self.calibrator.solve_device(extra_keypoints))
            if gather:
                from .dist import pack_records, gather_records
                out.append(gather_records(pack_records(*out)))
        done = torch.cuda.Event()
        done.record(side)
        self.last_done = done               # completion of THIS batch's solves (submit.py drains batch k - 1 on it)
        self._pending.append(done)
        if len(self._pending) > self.max_in_flight:          # bound the number of batches in flight (oldest first: in submission order)
            self._pending.pop(0).synchronize()
        return tuple(out)

    def join(self):
        """Make the current stream wait for every enqueued solve."""
        main = torch.cuda.current_stream(self.device)
        for ev in self._pending:
            main.wait_event(ev)
        self._pending.clear()

    def cameras(self, records: torch.Tensor):
        """records from submit() -> list of Optional[Camera] (synchronises)."""
        from .prediction import camera_from_record
        self.join()
        torch.cuda.current_stream(self.device).synchronize()
        return [camera_from_record(r, self.calibrator.img_size) for r in self.calibrator.records(records)]
