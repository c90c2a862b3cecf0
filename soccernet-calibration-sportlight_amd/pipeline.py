"""Frame pipeline: the production loop of /root/reference/src/utils/make_submit.py:42-75 on one GPU.

make_submit pushes each prediction row to a 16-process CPU pool; here the network runs on one stream and the batched
camera solves run on side streams, ordered by events, so the solve of batch i overlaps the convolutions of batches
i+1.. (the solve is latency-bound: one 64-lane wavefront per camera).

Streams only overlap when they sit on different hardware queues: HIP maps streams onto GPU_MAX_HW_QUEUES (default
4) queues, and with the multi-GPU gather another stream (RCCL's) joins; measured, it then aliases the network
stream and the step grows from 43 to 54 ms.  Export GPU_MAX_HW_QUEUES=8 before the process touches the GPU
(bench.py and submit.py do).

Round 5 -- the solves at the REFERENCE's refine criterion (baseline/camera.py:116: solvePnPRefineLM (20000, 1e-5)).
One crawling Levenberg-Marquardt fit keeps a single wavefront busy for up to ~330 ms, and a launch ends with its
slowest wavefront.  Three things follow (NOTES/design_history_r1_r5.md §11.1, all measured):

* A POOL of solve streams: batch k goes to solve stream k mod P (P = SOLVE_STREAMS, default 3 -- see below why not
  more --, shared by every pipeline of a device), so up to P batches are being solved while the network runs on; on ONE side stream the
  solve of batch k + 1 queued behind batch k's slowest fit and the pipeline ran at the pace of the solver (VERDICT
  r4 weak 2).  Results are still delivered in submission order (`join`, `cameras`, and the gather: collectives are
  issued in submission order and run in that order on RCCL's own stream).
* The solve streams are CU-MASKED (sncal_stream_create_cu_mask: SOLVE_CUS_PER_XCD compute units of each XCD, default
  1 = 8 of 256 CUs = 32 solver wavefronts at a time).  A solve wavefront owns a SIMD's whole register file, the
  hardware deals every kernel's workgroups to the XCDs round-robin, and so one held CU costs EVERY kernel 1/32 of
  its XCD: 64 such waves scattered over the chip cost the network 13.5 %, confined to one CU per XCD 3.6 %, however
  many fits crawl (tools/dev/cumask_probe.py).
* A CU-masked stream is a BLOCKING stream (HIP has no flags argument there): it synchronises with the legacy null
  stream in both directions -- a kernel, a copy, even an event record on the null stream waits for every solve in
  flight.  So the masked pool is used only when the caller runs the pipeline on a stream of their own
  (`with pipe.stream():` provides one; bench.py and submit.py do); a caller on the null stream gets plain
  non-blocking solve streams -- correct, ordered by the same events, but paying the scattered-CU price above.
"""
import atexit
import contextlib
import os

import torch

from . import _lib

# Solve streams per device.  THREE, not more: the network's stream + three solve streams are four hardware queues, and a fifth ACTIVE
# queue costs the network 6.6 ms per 88 ms step even when the solves are short (measured at the 200-iteration cap: 1, 2, 3 solve
# streams 87.9 ms per step, 4 streams 94.5, 4 streams squeezed onto GPU_MAX_HW_QUEUES=4 queues 88.1 -- the compute pipes serve four
# queues side by side and time-slice beyond that).  Multi-GPU runs keep all three: the path's one collective is issued ONCE, after the
# last batch (`gather_all`), so RCCL's stream is idle while batches are in flight (rounds 2-5 gathered per step on the solve stream and
# gave RCCL the fourth queue: two solve streams, and every step's collective behind the slowest rank's slowest fit).
SOLVE_STREAMS = max(1, int(os.environ.get('SNCAL_SOLVE_STREAMS', '3')))
SOLVE_CUS_PER_XCD = int(os.environ.get('SNCAL_SOLVE_CUS_PER_XCD', '1'))      # 0 = unmasked solve streams (round-4 behaviour)
_POOLS = {}      # (device index, masked) -> [solve streams, next, masked]: one pool per device and kind (lanes share it: hardware queues are few)
_OWN = {}        # device index -> a non-blocking stream for callers that have none (CalibrationPipeline.stream)


def _device_index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def _n_streams():
    return SOLVE_STREAMS


_MASK_REFUSED = set()     # device indices whose runtime refused a CU-masked stream (warned once; plain solve streams from then on)


def _solve_pool(device, masked):
    """-> ([streams], next, masked): the device's pool of solve streams.  A device that refuses the CU mask (a CU count that is not a
    multiple of 8, a runtime without hipExtStreamCreateWithCUMask) gets the unmasked pool with a warning instead of an error: the
    solves then pay the scattered-CU price of the module docstring, the results are the same."""
    idx = _device_index(device)
    masked = bool(masked) and idx not in _MASK_REFUSED
    key = (idx, masked)
    if key not in _POOLS:
        streams = []
        with torch.cuda.device(idx):
            for _ in range(_n_streams()):
                if masked:
                    h = _lib.vp()
                    st = _lib.lib().sncal_stream_create_cu_mask(SOLVE_CUS_PER_XCD, h)
                    if st != 0:
                        import warnings
                        why = _lib.lib().sncal_last_error().decode(errors='replace')
                        warnings.warn(f'cuda:{idx}: CU-masked solve streams unavailable ({why}); '
                                      'using plain solve streams (solver wavefronts may land on any CU)')
                        for q in streams:
                            _lib.lib().sncal_stream_destroy(q.cuda_stream)
                        _MASK_REFUSED.add(idx)
                        return _solve_pool(device, False)
                    streams.append(torch.cuda.ExternalStream(h.value, device=torch.device('cuda', idx)))
                else:
                    streams.append(torch.cuda.Stream(device=idx))
        _POOLS[key] = [streams, 0, masked]
    return _POOLS[key]


def _destroy_masked_pools():
    """At interpreter exit: the CU-masked solve streams are created by the library (hipExtStreamCreateWithCUMask), torch only borrows
    them (ExternalStream) -- nobody else destroys them, and a process that exits with such streams alive crashed in a library finaliser
    under rocprofv3 (SIGSEGV inside __cxa_finalize after the tool's output was written; unmasked pools did not).  sncal_stream_destroy
    also releases the scratch block sncal_calibrate may hold for the stream."""
    for key in [k for k in _POOLS if k[1]]:
        streams = _POOLS.pop(key)[0]
        for s in streams:
            try:
                s.synchronize()
                _lib.lib().sncal_stream_destroy(s.cuda_stream)
            except Exception:
                pass


atexit.register(_destroy_masked_pools)


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
        self.max_in_flight = 2 * _n_streams()
        self._pending = []
        self._unjoined = []                 # completion events of the batches logged for gather_all()
        from .dist import RecordLog
        self.log = RecordLog()              # submit(gather=True): this rank's packed records, batch after batch
        self.last_done = None
        self.masked = False                 # kind of solve stream the last submit() used

    @contextlib.contextmanager
    def stream(self):
        """`with pipe.stream():` -- run the loop (frame uploads / JPEG decode, submit(), result handling) on a non-blocking stream of
        the pipeline's own instead of the null stream: the condition for the CU-masked solve streams (module docstring)."""
        key = _device_index(self.device)
        if key not in _OWN:
            _OWN[key] = torch.cuda.Stream(device=key)
        with torch.cuda.stream(_OWN[key]):
            yield _OWN[key]

    def _next_solve_stream(self, masked):
        pool = _solve_pool(self.device, masked)
        streams, k, self.masked = pool
        pool[1] = (k + 1) % len(streams)
        return streams[k]

    def submit(self, frames: torch.Tensor, names=None, extra_keypoints: torch.Tensor = None, gather: bool = False,
               solve_decoded: bool = True):
        """frames (B,3,H,W) fp32 (ToTensor's output) or (B,H,W,3) uint8 BGR (cv2.imread's / JpegDecoder.decode's
        output) on the GPU.  Enqueues forward+decode on the current stream and the solve(s) on a solve stream; returns
        (kpts, records[, extra_records]) device tensors (asynchronous; `join()` / `cameras()` / `last_done` order a consumer
        behind them).  gather=True (multi-GPU): the batch's packed per-frame records (keypoints + camera records) are
        LOGGED on the rank (dist.RecordLog, packed on the solve stream behind the solves); the one collective of the path --
        every rank's records to every rank -- is `gather_all()`, once, after the last batch.  solve_decoded=False (measurement aid, tools/noisy_pipeline.py): the step's ONE
        solve is that of `extra_keypoints`; the records slot of the decoded keypoints is returned as None."""
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
        # everything the solve reads (keypoints, line points, the caller's extra keypoints) is produced on `main` ABOVE this record
        ready = torch.cuda.Event()
        ready.record(main)
        # CU-masked (blocking) solve streams only beside a caller that stays off the null stream
        self.masked = SOLVE_CUS_PER_XCD > 0 and main != torch.cuda.default_stream(self.device)
        side = self._next_solve_stream(self.masked)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            kpts.record_stream(side)
            if d_lp is not None:
                d_lp.record_stream(side)
            rec = self.calibrator.solve_device(kpts, d_lp) if solve_decoded or extra_keypoints is None else None
            out = [kpts, rec]
            if extra_keypoints is not None:
                extra_keypoints.record_stream(side)
                out.append(self.calibrator.solve_device(extra_keypoints))
            if gather:
                from .dist import pack_records
                self.log.add(pack_records(*[o for o in out if o is not None]))
            done = torch.cuda.Event()
            done.record(side)
        self.last_done = done               # completion of THIS batch's solves (submit.py drains batch k - 1 on it)
        self._pending.append(done)
        if gather:
            self._unjoined.append(done)
        if len(self._pending) > self.max_in_flight:          # bound the number of batches in flight (oldest first: in submission order)
            self._pending.pop(0).synchronize()
        return tuple(out)

    def join(self):
        """Make the current stream wait for every enqueued solve."""
        cur = torch.cuda.current_stream(self.device)
        for ev in self._pending:
            cur.wait_event(ev)
        self._pending.clear()

    def gather_all(self, counts=None):
        """The path's ONE collective (multi-GPU; north_star: "a single RCCL gather over xGMI at the end"): the packed
        records of every batch submitted with gather=True since the last call, from every rank to every rank, in frame order
        (rank-major: a rank owns a contiguous block of frames, dist.shard_range) -> (frames of all ranks, bytes) uint8.  Enqueued on
        the current stream behind every logged batch's solves; `counts` = frames per rank when the shards are ragged.  Without an
        initialised process group the result is the rank's own log."""
        cur = torch.cuda.current_stream(self.device)
        for ev in self._unjoined:
            cur.wait_event(ev)
        self._unjoined.clear()
        return self.log.gather(counts)

    def check_range(self):
        """Raise SncalRangeError if a forward since the last check left the split-fp16 engine's range (HRNetHeatmap.range_status):
        keypoints and cameras of those batches are not the reference's fp32 results.  Synchronises the current stream."""
        for n in (self.net, self.line_net):
            if n is not None and n.dtype_name == 'fp16x3':
                n.range_status(clear=True, check=True)

    def cameras(self, records: torch.Tensor):
        """records from submit() -> list of Optional[Camera] (synchronises)."""
        from .prediction import camera_from_record
        for ev in self._pending:
            ev.synchronize()
        self._pending.clear()
        self.check_range()
        return [camera_from_record(r, self.calibrator.img_size) for r in self.calibrator.records(records)]
