import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
B = 64
knet = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='bf16', device=dev); knet.load_state_dict(seeded_weights('hrnet_w48', 1))
lnet = sncal_amd.HRNetHeatmap('line_hrnet_w48', dtype='bf16', device=dev); lnet.load_state_dict(seeded_weights('line_hrnet_w48', 2))
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter', lines_file=None,
                             max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
pipe = sncal_amd.CalibrationPipeline(knet, cc, line_net=lnet)
x = torch.rand((B, 3, 540, 960), device=dev)
kp = torch.from_numpy(sncal_amd.synth.synthetic_keypoints(B, seed=77)).to(dev)
for _ in range(2): pipe.submit(x, extra_keypoints=kp)
pipe.join(); torch.cuda.synchronize()
t0 = time.time(); K = 5
for _ in range(K): pipe.submit(x, extra_keypoints=kp)
pipe.join(); torch.cuda.synchronize()
dt = (time.time() - t0) / K
print(f'C4 (W48 keypoint + W48 line net + line join + 2 solves), B={B}: {dt*1e3:.1f} ms/step, {B/dt:.1f} frames/s, {B/dt*879.2e9/1e12:.1f} TFLOP/s (reference formulation)')
lnet.set_profiling(True)
lnet.forward(x, want_heat=True); torch.cuda.synchronize()
prof = sorted(lnet.get_profile(), key=lambda p: -p['ms'])
print('line net alone: %.1f ms' % sum(p['ms'] for p in prof))
for p in prof[:6]: print('   %6.2f ms  n=%3d  %s' % (p['ms'], p['launches'], p['kernel']))
