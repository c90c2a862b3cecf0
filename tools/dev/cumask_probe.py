"""CU-masked side streams (hipExtStreamCreateWithCUMask): the network (fp16x3 W48, 64 frames) beside k fat spinning waves confined to
a CU mask.  Masks: bits i, i + 8, ... are the CUs of XCD i % 8 (KFD interleaves the mask over the XCCs).  GPU box."""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench, sncal_amd
hip = ctypes.CDLL('libamdhip64.so')
H = ctypes.CDLL(os.path.join(ROOT, 'tools', 'scratch', 'libholder.so'))
H.holder_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev); net.load_state_dict(sd)
frames, _ = sncal_amd.synth.stamped_frames(64, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * (w + 1)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
K = 6
NS = torch.cuda.Stream()          # the network's own NON-BLOCKING stream: a CU-masked stream is a blocking one (it synchronises with the null stream)
def run(stream=None, variant=1, k=0):
    with torch.cuda.stream(NS):
        for _ in range(2): net.forward(x, want_heat=False, decode_size=(540, 960))
    torch.cuda.synchronize()
    if stream is not None:
        H.holder_launch(variant, k, K * 100.0, ctypes.c_void_p(stream.cuda_stream))
        time.sleep(0.02)
    t0 = time.perf_counter()
    with torch.cuda.stream(NS):
        for _ in range(K): net.forward(x, want_heat=False, decode_size=(540, 960))
    NS.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    torch.cuda.synchronize()
    return dt
print(f'alone: {run():.2f} ms per step')
plain = torch.cuda.Stream()
cases = {'one CU per XCD (bits 0-7)': list(range(8)), 'eight CUs of XCD 0 (bits 0,8,..,56)': list(range(0, 64, 8)),
         'two CUs per XCD (bits 0-15)': list(range(16)), 'one CU of XCDs 0,2,4,6': [0, 2, 4, 6]}
for k in (8, 64):
    print(f'unmasked stream, k={k} fat waves: {run(plain, 1, k):.2f}')
    for name, bits in cases.items():
        s = masked_stream(bits)
        print(f'mask {name}, k={k} fat waves: {run(s, 1, k):.2f}', flush=True)
print(f'alone again: {run():.2f}')
# how long do 64 fat waves of 10 ms take on the 8-CU mask (32 wave slots)?  expect 2 rounds
s = masked_stream(list(range(8)))
torch.cuda.synchronize(); t0 = time.perf_counter()
H.holder_launch(1, 64, 10.0, ctypes.c_void_p(s.cuda_stream)); s.synchronize()
print(f'64 fat waves x 10 ms on the one-CU-per-XCD mask: {(time.perf_counter() - t0) * 1e3:.1f} ms')
